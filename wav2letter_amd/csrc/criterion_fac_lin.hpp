// criterion_fac_lin.hpp -- ForceAlignmentCriterion forward / backward scans for N <= 32 labels and targets of up to 320
// positions (the ASG letter recipes: N = 30, L <= 300), ONE wavefront per utterance; included by criterion_fac.hip.
//
// Replaces fl::lib::{cpu,cuda}::ForceAlignmentCriterion<float> (un-vendored Flashlight; call sites
// recipes/slimIPL/src/Train.cpp:408-410, :1675; math SURVEY.md App. B.1; CPU restatement oracle/criterion_oracle.c; the
// arithmetic of fac_fwd_lin is modelled op for op in oracle/asg_linear_domain.py::fac_kernel_model).
//
// The log-domain kernels (fac_fwd_blk) pay ~16 dependent fp64 / transcendental operations per position and frame plus one
// workgroup barrier per frame (~0.27 us per frame).  Here the lattice lives in a scaled LINEAR domain:
//   h_t[i] = alpha_t[i] exp(A[y_i][y_i])  ->  h_t[i] = c_t[y_i] (h_{t-1}[i] + kappa[i] h_{t-1}[i-1])
//   c_t[n] = 2^(z_t[n] - max_n z_t[n]),  z_t[n] = (x_t[n] + A[n][n]) log2 e:   ONE row of N values per frame (the 300 positions
//            share 30 labels), computed by lanes 0..N-1 with an integer / fraction split (fp32 exp2 of the fraction, fp64 ldexp of
//            the integer part: no underflow however far a label lies below the frame's best), staged in LDS, gathered by label;
//   kappa[i] = exp(A[y_i][y_{i-1}] - A[y_{i-1}][y_{i-1}])
// so a position costs ONE fp64 fma and ONE fp64 multiply per frame.  Lane l holds positions l P .. l P + P - 1 as fp64
// mantissas with one integer exponent e_l; the left neighbour's last position arrives by DPP (wave_shr:1) scaled by
// 2^(e_{l-1} - e_l).  Every kFacRenorm frames -- and at once when some lane's largest mantissa has left [2^-200, 2^300] -- the
// lanes renormalise: positions that can no longer reach the end (i < S - (T - t), the reference's `low`) are
// zeroed (in a tight alignment the lagging, useless positions would otherwise outgrow the lattice front -- the only path that
// finishes -- by thousands of bits); a lane with mass takes e_l = max(own, e_l' - kFacDecay * (lanes with mass between l' and
// l)), so what arrives from the left is at most 2^kFacDecay larger than what the lane holds; an empty lane copies the exponent
// of the nearest lane with mass on its left (the study of oracle/asg_linear_domain.py: ONE exponent per frame is not enough,
// one per lane is).  What a lane flushes lies 1000+ bits below a feasible position a few labels away: harmless while one
// stay / advance decision gains less than ~200 bits.  The kernel MEASURES that -- the largest per-frame spread of the label
// scores plus the largest |log2 kappa| -- and flags an utterance beyond kFacSafeBits in ws.redo[b]: the log-domain kernel
// (fac_fwd_blk), launched behind it, recomputes exactly the flagged utterances and returns at once for the others.
// The forward leaves the stay share w1[t][i] = h_{t-1}[i] / (h_{t-1}[i] + kappa[i] h_{t-1}[i-1])
// (fp32, [B][T][L]) -- all the backward scan needs:
//   dalpha_{t-1}[i] = dalpha_t[i] w1[t][i] + dalpha_t[i+1] (1 - w1[t][i+1])
// fac_bwd_wave: the same five fp32 operations per position as fac_bwd_blk, but the neighbour term is a DPP lane shift inside
// the one wave instead of an LDS row + workgroup barrier per frame.
#pragma once
#include "common.hpp"

namespace w2l {

constexpr int kFacLinChunk = 8;    // frames per chunk (prefetch, LDS label rows, batched w1 stores)
constexpr int kFacRenorm = 4;      // frames between two renormalisations (divides kFacLinChunk)
constexpr int kFacDecay = 600;     // bits per lane WITH MASS in the exponent scan
constexpr int kFacEmptyExp = -(1 << 30);
constexpr float kFacSafeBits = 160.f;   // per-decision gain up to which the per-lane exponents are exact (tests/test_asg_linear_domain.py)

// inclusive maximum scan over the 64 lanes (DPP row shifts + the two row broadcasts of a wave scan)
__device__ __forceinline__ int wave_scan_max_i(int v, int lane) {
  const int NEGB = kFacEmptyExp;
  int t;
  t = __builtin_amdgcn_update_dpp(NEGB, v, 0x111, 0xf, 0xf, false); v = max(v, t);   // row_shr:1
  t = __builtin_amdgcn_update_dpp(NEGB, v, 0x112, 0xf, 0xf, false); v = max(v, t);   // row_shr:2
  t = __builtin_amdgcn_update_dpp(NEGB, v, 0x114, 0xf, 0xf, false); v = max(v, t);   // row_shr:4
  t = __builtin_amdgcn_update_dpp(NEGB, v, 0x118, 0xf, 0xf, false); v = max(v, t);   // row_shr:8
  t = __builtin_amdgcn_update_dpp(NEGB, v, 0x142, 0xa, 0xf, false);   // row_bcast:15: lane 15 of the previous row -> rows 1, 3
  if (lane & 16) v = max(v, t);
  t = __builtin_amdgcn_update_dpp(NEGB, v, 0x143, 0xc, 0xf, false);   // row_bcast:31: lane 31 -> rows 2, 3
  if (lane & 32) v = max(v, t);
  return v;
}

__device__ __forceinline__ int dpp_wave_shr1_i(int v, int fill) { return __builtin_amdgcn_update_dpp(fill, v, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ float dpp_wave_shl1_f(float v, float fill) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v), 0x130, 0xf, 0xf, false));
}

template <int P>
__global__ __launch_bounds__(64) void fac_fwd_lin(int T, int N, int L, int scaleMode, const float* __restrict__ x,
                                                  const int* __restrict__ target, const int* __restrict__ targetSize,
                                                  const float* __restrict__ trans, float* __restrict__ loss, FacWs ws) {
  __shared__ double sC[2][kFacLinChunk][32];   // label rows c_t[n] of two chunks
  const int b = blockIdx.x, lane = threadIdx.x;
  const int S = targetSize[b];
  const float sc = scale_of(scaleMode, T, S);
  if (lane == 0) ws.scale[b] = sc;
  if (S <= 0) {
    if (lane == 0) { loss[b] = 0.f; ws.redo[b] = 0; }
    return;
  }
  const int* y = target + (size_t)b * L;
  const float* xb = x + (size_t)b * T * N;
  float* w1b = ws.w1 + (size_t)b * T * L;
  const float NEG = -INFINITY;
  const float L2E = 1.44269504088896341f;

  int yi[P];
  bool valid[P];
  double kap[P], h[P];
  float gainBits = 0.f;   // largest |log2 kappa| of this lane's positions
  float spread = 0.f;     // largest (frame maximum - own label's score), lanes 0..N-1
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const int i = lane * P + p;
    valid[p] = i < S;
    yi[p] = valid[p] ? y[i] : 0;
    const int yp = (valid[p] && i > 0) ? y[i - 1] : 0;
    const float dk = (valid[p] && i > 0) ? trans[(size_t)yi[p] * N + yp] - trans[(size_t)yp * N + yp] : NEG;
    kap[p] = (double)__expf(dk);   // exp(-inf) = 0 for position 0 and beyond the target
    h[p] = 0.0;
    if (valid[p] && i > 0) {
      const float kb = fabsf(dk) * L2E;
      gainBits = fmaxf(gainBits, kb == kb ? kb : INFINITY);   // (a NaN or infinite transition: leave it to the log-domain kernel)
    }
  }
  const bool rowLane = lane < N;                                   // lanes 0..N-1 compute the label row of a frame
  const float adl = rowLane ? trans[(size_t)lane * N + lane] * L2E : 0.f;

  float xc[kFacLinChunk], xn[kFacLinChunk];
#pragma unroll
  for (int s = 0; s < kFacLinChunk; ++s) xc[s] = (rowLane && s < T) ? xb[(size_t)s * N + lane] : 0.f;

  const char* sCb = (const char*)&sC[0][0][0];
  int off0[P], off1[P];   // LDS byte offsets of the gathers in the two row buffers
#pragma unroll
  for (int p = 0; p < P; ++p) { off0[p] = yi[p] * 8; off1[p] = yi[p] * 8 + kFacLinChunk * 32 * 8; }

  double zsum = 0.0;   // sum_t max_n z_t[n] (base 2)
  int el = 0, dl = 0;  // this lane's exponent; e_{l-1} - e_l

  // label rows of one chunk -> LDS buffer `buf` (frame s of the chunk in row s)
  auto rows = [&](const float (&xv)[kFacLinChunk], int t0, int buf) {
#pragma unroll
    for (int s = 0; s < kFacLinChunk; ++s) {
      const float z = rowLane ? fmaf(xv[s], L2E, adl) : NEG;
      const float zm = wave_max_rows<2>(z);
      if (rowLane && t0 + s < T) spread = fmaxf(spread, zm - z);
      float zr = fmaxf(z - zm, -4000.f);
      const float zi = __builtin_rintf(zr);
      const float fr = __builtin_amdgcn_exp2f(zr - zi);           // in [2^-0.5, 2^0.5]
      const double c = __builtin_amdgcn_ldexp((double)fr, (int)zi);
      if (rowLane) sC[buf][s][lane] = c;
      if (t0 + s < T) zsum += (double)zm;
    }
  };

  auto renorm = [&](int t) {
    const int low = S - (T - t);   // positions below `low` cannot reach the end any more
    double mx = 0.0;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      if (lane * P + p < low) h[p] = 0.0;
      mx = fmax(mx, h[p]);
    }
    const bool has = mx > 0.0;
    const int cand = has ? el + __builtin_amdgcn_frexp_exp(mx) : kFacEmptyExp;
    // number of lanes with mass up to and including this one
    const unsigned long long hm = __ballot(has);
    const int cnt = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(hm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)hm, 0u)) + (has ? 1 : 0);
    const int run = wave_scan_max_i(has ? cand + kFacDecay * cnt : kFacEmptyExp, lane);
    const int en = run > kFacEmptyExp ? run - kFacDecay * cnt : el;
    int sh = el - en;
    sh = sh < -2200 ? -2200 : (sh > 2200 ? 2200 : sh);
#pragma unroll
    for (int p = 0; p < P; ++p) h[p] = __builtin_amdgcn_ldexp(h[p], sh);
    el = en;
    int d = dpp_wave_shr1_i(en, en) - en;
    d = d < -2200 ? -2200 : (d > 2200 ? 2200 : d);
    dl = d;
  };

  rows(xc, 0, 0);
  int buf = 0;
  for (int t0 = 0; t0 < T; t0 += kFacLinChunk) {
#pragma unroll
    for (int s = 0; s < kFacLinChunk; ++s) {
      const int tn = t0 + kFacLinChunk + s;
      xn[s] = (rowLane && tn < T) ? xb[(size_t)tn * N + lane] : 0.f;
    }
    float wst[kFacLinChunk][P];
#pragma unroll
    for (int s = 0; s < kFacLinChunk; ++s) {
      const int t = t0 + s;
#pragma unroll
      for (int p = 0; p < P; ++p) wst[s][p] = 0.f;
      if (t < T) {   // wave-uniform
        double c[P];
#pragma unroll
        for (int p = 0; p < P; ++p) c[p] = *(const double*)(sCb + (buf ? off1[p] : off0[p]) + s * 32 * 8);
        if (t == 0) {
          if (lane == 0) h[0] = c[0];   // alpha_0[0] = x_0[y_0]
        } else {
          const double hin = __builtin_amdgcn_ldexp(lane_shift_up_dpp(h[P - 1], 0.0), dl);
          // in place from the last position down (h[p-1] is still the previous frame's); positions beyond the target need no
          // mask: kappa = 0 there and their h starts at 0, so tot = 0 and w = 0 / 2^-1000 = 0
#pragma unroll
          for (int p = P - 1; p >= 0; --p) {
            const double prev = p == 0 ? hin : h[p - 1];
            const double tot = fma(prev, kap[p], h[p]);
            // v_rcp_f64 is accurate to 2^29 ulp = 2^-23 relative: fp32 accuracy, what w1 is stored in (the log-domain kernel
            // took v_rcp_f32 here)
            const double rc = __builtin_amdgcn_rcp(fmax(tot, 0x1p-1000));
            wst[s][p] = (float)(h[p] * rc);
            h[p] = c[p] * tot;
          }
        }
        if ((s % kFacRenorm) == 0) {   // t0 is a multiple of the chunk: frame 0 and every kFacRenorm-th after it
          renorm(t);
        } else {
          // early trigger: some lane's largest mantissa left [2^-200, 2^300] (frames that push the forced path hundreds of
          // nats under the frame's best label; a value that crossed several lanes).  Positive doubles order like their high
          // words: two integer v_max3 per frame.
          unsigned mh = 0;
#pragma unroll
          for (int p = 0; p < P; ++p) mh = max(mh, (unsigned)(__double_as_longlong(h[p]) >> 32));
          if (__any(mh != 0u && (mh < ((1023u - 200u) << 20) || mh > ((1023u + 300u) << 20)))) renorm(t);
        }
      }
    }
#pragma unroll
    for (int s = 0; s < kFacLinChunk; ++s) {
      const int t = t0 + s;
#pragma unroll
      for (int p = 0; p < P; ++p)
        if (valid[p] && t >= 1 && t < T) w1b[(size_t)t * L + lane * P + p] = wst[s][p];
    }
#pragma unroll
    for (int s = 0; s < kFacLinChunk; ++s) asm volatile("" : "+v"(xn[s]));   // one vmcnt drain per chunk
    buf ^= 1;
    rows(xn, t0 + kFacLinChunk, buf);   // (a single wave: the LDS executes its operations in order -- no barrier)
#pragma unroll
    for (int s = 0; s < kFacLinChunk; ++s) xc[s] = xn[s];
  }
  // an utterance whose decisions gain more than kFacSafeBits is recomputed by the log-domain kernel
  {
    const float sp = wave_max(rowLane ? spread : 0.f), gb = wave_max(gainBits);
    if (lane == 0) ws.redo[b] = (sp + gb <= kFacSafeBits) ? 0 : 1;   // (NaN compares false -> 1)
  }
  // loss = scale * alpha_{T-1}[S-1],  alpha = (zsum + e_l + log2 h) ln 2 - A[y][y]
  const int il = S - 1;
#pragma unroll
  for (int p = 0; p < P; ++p) {
    if (lane * P + p == il) {
      float out = -INFINITY;
      if (h[p] > 0.0) {
        const int ex = __builtin_amdgcn_frexp_exp(h[p]);
        const float mant = (float)__builtin_amdgcn_frexp_mant(h[p]);
        const double l2 = zsum + (double)el + (double)ex + (double)__builtin_amdgcn_logf(mant);
        out = (float)((double)sc * (l2 * 0.69314718055994530942 - (double)trans[(size_t)yi[p] * N + yi[p]]));
      }
      loss[b] = out;
    }
  }
}

// backward scan of one utterance in one wave: consumes w1[t][i], leaves g * dalpha_t[i] in ws.dal for fac_scatter_k
template <int P>
__global__ __launch_bounds__(64) void fac_bwd_wave(int T, int N, int L, const int* __restrict__ target,
                                                   const int* __restrict__ targetSize, const float* __restrict__ grad,
                                                   float* __restrict__ transGrad, FacWs ws) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int S = targetSize[b];
  if (S <= 0) return;  // the scatter kernel zero-fills this utterance's gradient
  const int* y = target + (size_t)b * L;
  const float* __restrict__ w1b = ws.w1 + (size_t)b * T * L;
  float* __restrict__ dalb = ws.dal + (size_t)b * T * L;
  const float g = ws.scale[b] * grad[b];

  int yi[P], yp[P];
  bool valid[P];
  float da[P], accS[P], accP[P];
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const int i = lane * P + p;
    valid[p] = i < S;
    yi[p] = valid[p] ? y[i] : 0;
    yp[p] = (valid[p] && i > 0) ? y[i - 1] : 0;
    da[p] = (i == S - 1) ? 1.f : 0.f;
    accS[p] = 0.f;
    accP[p] = 0.f;
  }
  float wc[kFacLinChunk][P], wn[kFacLinChunk][P];
#pragma unroll
  for (int s = 0; s < kFacLinChunk; ++s) {
    const int t = T - 1 - s;
#pragma unroll
    for (int p = 0; p < P; ++p) wc[s][p] = (t >= 1 && valid[p]) ? w1b[(size_t)t * L + lane * P + p] : 0.f;
  }
  for (int thi = T - 1; thi >= 0; thi -= kFacLinChunk) {
#pragma unroll
    for (int s = 0; s < kFacLinChunk; ++s) {
      const int t = thi - kFacLinChunk - s;
#pragma unroll
      for (int p = 0; p < P; ++p) wn[s][p] = (t >= 1 && valid[p]) ? w1b[(size_t)t * L + lane * P + p] : 0.f;
    }
    float dst[kFacLinChunk][P];
#pragma unroll
    for (int s = 0; s < kFacLinChunk; ++s) {
      const int t = thi - s;
#pragma unroll
      for (int p = 0; p < P; ++p) dst[s][p] = g * da[p];
      if (t >= 1) {   // wave-uniform
        float st[P], adv[P];
#pragma unroll
        for (int p = 0; p < P; ++p) {
          st[p] = da[p] * wc[s][p];
          adv[p] = da[p] - st[p];
          accS[p] += st[p];
          accP[p] += adv[p];
        }
        const float fromNext = dpp_wave_shl1_f(adv[0], 0.f);   // advance term of position (lane + 1) P
#pragma unroll
        for (int p = 0; p < P; ++p) da[p] = st[p] + (p + 1 < P ? adv[p + 1 < P ? p + 1 : 0] : fromNext);
      }
    }
#pragma unroll
    for (int s = 0; s < kFacLinChunk; ++s) {
      const int t = thi - s;
#pragma unroll
      for (int p = 0; p < P; ++p)
        if (t >= 0 && lane * P + p < L) dalb[(size_t)t * L + lane * P + p] = dst[s][p];
    }
#pragma unroll
    for (int s = 0; s < kFacLinChunk; ++s)
#pragma unroll
      for (int p = 0; p < P; ++p) asm volatile("" : "+v"(wn[s][p]));
#pragma unroll
    for (int s = 0; s < kFacLinChunk; ++s)
#pragma unroll
      for (int p = 0; p < P; ++p) wc[s][p] = wn[s][p];
  }
  float* tg = ws.tgpart ? ws.tgpart + (size_t)b * N * N : transGrad;
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const int i = lane * P + p;
    if (i < S) {
      if (accS[p] != 0.f) atomicAdd(&tg[(size_t)yi[p] * N + yi[p]], g * accS[p]);
      if (i > 0 && accP[p] != 0.f) atomicAdd(&tg[(size_t)yi[p] * N + yp[p]], g * accP[p]);
    }
  }
}

}  // namespace w2l
