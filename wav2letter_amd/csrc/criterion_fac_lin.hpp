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
#include <type_traits>

namespace w2l {

constexpr int kFacLinChunk = 8;    // frames per chunk (prefetch, LDS label rows, batched w1 stores)
constexpr int kFacRenorm = 4;      // frames between two renormalisations (divides kFacLinChunk)
constexpr int kFacDecay = 600;     // bits per lane WITH MASS in the exponent scan
constexpr int kFacEmptyExp = -(1 << 30);
// exp(d) as an fp64 value for any float d: the fraction by the fp32 exp2, the integer part by ldexp; exp(-inf) = 0.  (__expf is 0 /
// inf beyond -87 / +88 nats: with transition rows that wide -- kappa is exp of a DIFFERENCE of two transitions -- a forced alignment
// through such a step came out -inf / NaN while the log-domain reference stays finite: round 5, tools/exp/asg_wide_transitions.py)
__device__ __forceinline__ double fac_exp_wide(float d) {
  if (!(d > -INFINITY)) return d != d ? (double)d : 0.0;   // -inf -> 0 (position 0, positions beyond the target); NaN stays NaN
  const double z = fmin(fmax((double)d * 1.4426950408889634, -1070.0), 1020.0);
  const double zi = rint(z);
  return __builtin_amdgcn_ldexp((double)__builtin_amdgcn_exp2f((float)(z - zi)), (int)zi);
}

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
    kap[p] = fac_exp_wide(dk);   // exp(-inf) = 0 for position 0 and beyond the target
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


// ------------------------------------------------------------------------------------------------------------------------
// fac_fwd_blin: the same linear-domain recursion, ONE POSITION PER THREAD with ITS OWN exponent, one workgroup per utterance.
// What tools/micro/clock_probe.hip measured on MI355X (profiles/r04_run3_clock_probe.log): a wave that is alone on its SIMD
// issues one VALU instruction every ~6.5 cycles whether or not the instructions depend on each other (fp32, fp64 and DPP
// alike), and an LDS write -> s_barrier -> LDS read hand-over costs ~65 cycles.  A frame of a serial scan therefore costs
// (instructions ONE wave issues) x 6.5 cycles: fac_fwd_lin above, five positions per lane, issues ~160 instructions per frame
// (0.88 ms at T = 2000 -- measured slower than the log-domain fac_fwd_blk, 0.49 ms) -- while with one position per thread the
// waves of the workgroup issue side by side and a frame is ~25 instructions plus one barrier:
//   position i (thread i):  (m_nb, e_nb) <- LDS record of position i-1 (previous frame)       c <- LDS label row[y_i]
//     E = max(e, e_nb);  tot = ldexp(m, e - E) + kappa_i ldexp(m_nb, e_nb - E);  w1 = m' / tot;  h = c tot
//     (m, e) <- (frexp_mant(h), E + frexp_exp(h))   -> LDS record of position i (this frame);  s_barrier
//   one more wave (the "row wave") computes the label row c_{t+1}[n] = 2^(z - max z) of the NEXT frame meanwhile
//   (integer / fraction split, fp64 ldexp: no underflow) and sums the frame maxima.
// An exponent per position is the `group = 1` case of oracle/asg_linear_domain.py::fac_forward_linear: every position keeps the
// full fp64 range on its own -- no renormalisation scan, no pruning, no range check, no fallback.
struct FacRec { double m; int e; int pad; };
constexpr int kFacBlinChunk = 16;

template <int NW>
__global__ __launch_bounds__(64 * (NW + 1)) void fac_fwd_blin(int T, int N, int L, int scaleMode, const float* __restrict__ x,
                                                              const int* __restrict__ target, const int* __restrict__ targetSize,
                                                              const float* __restrict__ trans, float* __restrict__ loss, FacWs ws) {
  constexpr int NT = 64 * NW;
  __shared__ FacRec sH[2][NT + 1];   // sH[buf][i + 1] = position i after the frame of parity buf; [0] = position -1 (no mass)
  __shared__ double sC[2][32];       // label row of the frame of parity buf
  __shared__ double sZ;              // sum of the frame maxima (base 2), left by the row wave
  const int b = blockIdx.x, tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool rowWave = wave == NW;
  const int S = targetSize[b];
  const float sc = scale_of(scaleMode, T, S);
  if (tid == 0) { ws.scale[b] = sc; ws.redo[b] = 0; }
  if (S <= 0) {
    if (tid == 0) loss[b] = 0.f;
    return;
  }
  const float NEG = -INFINITY;
  const float L2E = 1.44269504088896341f;
  const int* y = target + (size_t)b * L;
  const float* xb = x + (size_t)b * T * N;
  float* w1b = ws.w1 + (size_t)b * T * L;
  const int lane = tid & 63;
  if (tid == 0) { sH[0][0].m = 0.0; sH[0][0].e = kFacEmptyExp; sH[1][0].m = 0.0; sH[1][0].e = kFacEmptyExp; }

  if (rowWave) {
    // ---- the row wave: label rows one frame ahead
    const bool act = lane < N;
    const float adl = act ? trans[(size_t)lane * N + lane] * L2E : 0.f;
    float xc[kFacBlinChunk], xn[kFacBlinChunk];
#pragma unroll
    for (int s = 0; s < kFacBlinChunk; ++s) xc[s] = (act && s < T) ? xb[(size_t)s * N + lane] : 0.f;
    double zsum = 0.0;
    auto row = [&](float xv, int t) {   // row of frame t -> sC[t & 1]
      const float z = act ? fmaf(xv, L2E, adl) : NEG;
      const float zm = wave_max_rows<2>(z);
      const float zr = fmaxf(z - zm, -4000.f);
      const float zi = __builtin_rintf(zr);
      const double c = __builtin_amdgcn_ldexp((double)__builtin_amdgcn_exp2f(zr - zi), (int)zi);
      if (act) sC[t & 1][lane] = c;
      zsum += (double)zm;
    };
    row(xc[0], 0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    for (int t0 = 0; t0 < T; t0 += kFacBlinChunk) {
#pragma unroll
      for (int s = 0; s < kFacBlinChunk; ++s) {
        const int tn = t0 + kFacBlinChunk + s;
        xn[s] = (act && tn < T) ? xb[(size_t)tn * N + lane] : 0.f;
      }
#pragma unroll
      for (int s = 0; s < kFacBlinChunk; ++s) {
        const int t = t0 + s;
        if (t < T) {   // uniform
          if (t + 1 < T) row(s + 1 < kFacBlinChunk ? xc[s + 1 < kFacBlinChunk ? s + 1 : 0] : xn[0], t + 1);
          if (t + 1 == T && lane == 0) sZ = zsum;
          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
      }
#pragma unroll
      for (int s = 0; s < kFacBlinChunk; ++s) asm volatile("" : "+v"(xn[s]));
#pragma unroll
      for (int s = 0; s < kFacBlinChunk; ++s) xc[s] = xn[s];
    }
    return;
  }

  // ---- position threads
  const int i = tid;
  const bool valid = i < S;
  const int yi = valid ? y[i] : 0;
  const int yp = (valid && i > 0) ? y[i - 1] : 0;
  const float dk = (valid && i > 0) ? trans[(size_t)yi * N + yp] - trans[(size_t)yp * N + yp] : NEG;
  const double kap = fac_exp_wide(dk);   // 0 for position 0 and beyond the target
  double m = 0.0;
  int e = kFacEmptyExp;
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // row 0 is in sC[0]
  for (int t0 = 0; t0 < T; t0 += kFacBlinChunk) {
    float wst[kFacBlinChunk];
#pragma unroll
    for (int s = 0; s < kFacBlinChunk; ++s) {
      const int t = t0 + s;
      wst[s] = 0.f;
      if (t < T) {   // uniform
        const double c = sC[t & 1][yi];
        double h;
        int E;
        if (t == 0) {
          h = i == 0 ? c : 0.0;   // alpha_0[0] = x_0[y_0]
          E = 0;
        } else {
          const FacRec nb = sH[(t - 1) & 1][i];   // position i - 1 after frame t - 1
          E = max(e, nb.e);
          const double ms = __builtin_amdgcn_ldexp(m, e - E);
          const double tot = fma(__builtin_amdgcn_ldexp(nb.m, nb.e - E), kap, ms);
          // v_rcp_f64: 2^-23 relative, the accuracy of the fp32 w1 (the log-domain kernel takes v_rcp_f32 here)
          wst[s] = (float)(ms * __builtin_amdgcn_rcp(fmax(tot, 0x1p-1000)));
          h = valid ? c * tot : 0.0;
        }
        m = __builtin_amdgcn_frexp_mant(h);
        e = h > 0.0 ? E + __builtin_amdgcn_frexp_exp(h) : kFacEmptyExp;
        FacRec out;
        out.m = m; out.e = e; out.pad = 0;
        sH[t & 1][i + 1] = out;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      }
    }
#pragma unroll
    for (int s = 0; s < kFacBlinChunk; ++s) {
      const int t = t0 + s;
      if (valid && t >= 1 && t < T) w1b[(size_t)t * L + i] = wst[s];
    }
  }
  if (i == S - 1) {   // loss = scale * alpha_{T-1}[S-1],  alpha = (zsum + e + log2 m) ln 2 - A[y][y]
    float out = -INFINITY;
    if (m > 0.0) {
      const double l2 = sZ + (double)e + (double)__builtin_amdgcn_logf((float)m);
      out = (float)((double)sc * (l2 * 0.69314718055994530942 - (double)trans[(size_t)yi * N + yi]));
    }
    loss[b] = out;
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// fac_rows_k + fac_fwd_blin2: the label rows do not depend on the recursion, so they come from a parallel pre-pass over all
// (utterance, frame) pairs -- c_t[n] as fp64 in the workspace ([B][T][32]) and the frame maxima [B][T] -- and the scan's position
// threads gather c_t[y_i] straight from there, prefetched a chunk of 16 frames ahead (the way fac_fwd_blk gathers x_t[y_i]).
// Against fac_fwd_blin: no row wave (one wave less on the CU's four SIMDs), no LDS row, three instructions less per frame.
constexpr int kFacRowsPerWave = 8;
constexpr int kFacRowsWaves = 4;   // waves per workgroup (16000 one-wave workgroups at B = 64, T = 2000 were launch-rate bound: 20 us)
// max over the 32 lanes of this lane's half of the wave (every lane gets it): the in-row DPP reduction of wave_max_rows, then the
// two rows of the half through a 16-lane swap
__device__ __forceinline__ float half_wave_max(float v) {
  asm volatile(
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0"
      : "+v"(v));
  return fmaxf(v, __shfl_xor(v, 16));
}
// Two frames per pass (round 6): N <= 32 labels fill half a wave, so lanes 0 .. 31 take frame t and lanes 32 .. 63 frame t + 1 --
// half the instructions per frame, 512-byte stores; the values are those of the one-frame-per-pass kernel bit for bit (a maximum
// does not round).  21 -> 12 us at B = 64, T = 2000.
__global__ __launch_bounds__(64 * kFacRowsWaves) static void fac_rows_k(int T, int N, const float* __restrict__ x, const float* __restrict__ trans,
                                                 double* __restrict__ crow, float* __restrict__ zmax, float* __restrict__ zspr = nullptr,
                                                 const int* __restrict__ tsTarget = nullptr, int tsL = 0, int* __restrict__ tsOut = nullptr,
                                                 float* __restrict__ zeroBuf = nullptr, unsigned zeroCount = 0) {
  // (zeroBuf: the transition-gradient partials of the backward pass, cleared here -- backward starts without a fill launch)
  if (zeroBuf) {
    const unsigned stride = gridDim.x * gridDim.y * blockDim.x;
    for (unsigned k = (blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x; k < zeroCount; k += stride) zeroBuf[k] = 0.f;
  }
  // (tsOut: the ASG criterion's fused sequence -- the utterance's target size, batch_target_size_k's count of the labels in front of
  //  the first negative one capped at T, by the first wave of the utterance's first block: no launch of its own in front of the scans)
  if (tsOut && blockIdx.x == 0 && threadIdx.x < 64) {
    const int* y = tsTarget + (size_t)blockIdx.y * tsL;
    int first = tsL;
    for (int i = (int)threadIdx.x; i < tsL; i += 64)
      if (y[i] < 0) { first = i; break; }
    for (int off = 32; off > 0; off >>= 1) first = min(first, __shfl_xor(first, off));
    if (threadIdx.x == 0) tsOut[blockIdx.y] = first < T ? first : T;
  }
  const int b = blockIdx.y, lane = threadIdx.x & 63, half = lane >> 5, n = lane & 31;
  const int t0 = (blockIdx.x * kFacRowsWaves + (threadIdx.x >> 6)) * kFacRowsPerWave;
  const float L2E = 1.44269504088896341f;
  const bool act = n < N;
  const float adl = act ? trans[(size_t)n * N + n] * L2E : 0.f;
  const float* xb = x + (size_t)b * T * N;
  float xv[kFacRowsPerWave / 2];
#pragma unroll
  for (int s = 0; s < kFacRowsPerWave / 2; ++s) xv[s] = (act && t0 + 2 * s + half < T) ? xb[(size_t)(t0 + 2 * s + half) * N + n] : 0.f;
#pragma unroll
  for (int s = 0; s < kFacRowsPerWave / 2; ++s) {
    const int t = t0 + 2 * s + half;
    const float z = act ? fmaf(xv[s], L2E, adl) : -INFINITY;
    const float zm = half_wave_max(z);
    const float zr = fmaxf(z - zm, -4000.f);
    const float zi = __builtin_rintf(zr);
    const double c = __builtin_amdgcn_ldexp((double)__builtin_amdgcn_exp2f(zr - zi), (int)zi);
    float sp = 0.f;
    if (zspr) sp = half_wave_max(act ? (z == z ? zm - z : INFINITY) : 0.f);   // the frame's spread (bits), for the range check of fac_fwd_plin; a NaN score makes it NaN
    if (t < T) {
      crow[((size_t)b * T + t) * 32 + n] = act ? c : 0.0;
      if (n == 0) {
        zmax[(size_t)b * T + t] = zm;
        if (zspr) zspr[(size_t)b * T + t] = zm == zm ? sp : __builtin_nanf("");
      }
    }
  }
}

template <int NW>
__global__ __launch_bounds__(64 * NW) void fac_fwd_blin2(int T, int N, int L, int scaleMode, const int* __restrict__ target,
                                                         const int* __restrict__ targetSize, const float* __restrict__ trans,
                                                         float* __restrict__ loss, FacWs ws) {
  constexpr int NT = 64 * NW;
  __shared__ FacRec sH[2][NT + 1];   // sH[buf][i + 1] = position i after the frame of parity buf; [0] = position -1 (no mass)
  const int b = blockIdx.x, tid = threadIdx.x;
  const int S = targetSize[b];
  const float sc = scale_of(scaleMode, T, S);
  if (tid == 0) { ws.scale[b] = sc; ws.redo[b] = 0; }
  if (S <= 0) {
    if (tid == 0) loss[b] = 0.f;
    return;
  }
  const float NEG = -INFINITY;
  const int* y = target + (size_t)b * L;
  float* w1b = ws.w1 + (size_t)b * T * L;
  if (tid == 0) { sH[0][0].m = 0.0; sH[0][0].e = kFacEmptyExp; sH[1][0].m = 0.0; sH[1][0].e = kFacEmptyExp; }
  const int i = tid;
  const bool valid = i < S;
  const int yi = valid ? y[i] : 0;
  const int yp = (valid && i > 0) ? y[i - 1] : 0;
  const float dk = (valid && i > 0) ? trans[(size_t)yi * N + yp] - trans[(size_t)yp * N + yp] : NEG;
  const double kap = fac_exp_wide(dk);   // 0 for position 0 and beyond the target
  const double* cb = ws.crow + (size_t)b * T * 32 + yi;
  double cc[kFacBlinChunk], cn[kFacBlinChunk];
#pragma unroll
  for (int s = 0; s < kFacBlinChunk; ++s) cc[s] = (valid && s < T) ? cb[(size_t)s * 32] : 0.0;
#pragma unroll
  for (int s = 0; s < kFacBlinChunk; ++s) asm volatile("" : "+v"(cc[s]));   // landed before the loop (see fac_fwd_plin)
  double m = 0.0;
  int e = kFacEmptyExp;
  __syncthreads();
  for (int t0 = 0; t0 < T; t0 += kFacBlinChunk) {
#pragma unroll
    for (int s = 0; s < kFacBlinChunk; ++s) {
      const int tn = t0 + kFacBlinChunk + s;
      cn[s] = (valid && tn < T) ? cb[(size_t)tn * 32] : 0.0;
    }
    float wst[kFacBlinChunk];
#pragma unroll
    for (int s = 0; s < kFacBlinChunk; ++s) {
      const int t = t0 + s;
      wst[s] = 0.f;
      if (t < T) {   // uniform
        double h;
        int E;
        if (t == 0) {
          h = i == 0 ? cc[s] : 0.0;   // alpha_0[0] = x_0[y_0]
          E = 0;
        } else {
          const FacRec nb = sH[(t - 1) & 1][i];   // position i - 1 after frame t - 1
          E = max(e, nb.e);
          const double ms = __builtin_amdgcn_ldexp(m, e - E);
          const double tot = fma(__builtin_amdgcn_ldexp(nb.m, nb.e - E), kap, ms);
          wst[s] = (float)(ms * __builtin_amdgcn_rcp(fmax(tot, 0x1p-1000)));   // (v_rcp_f64: 2^-23 relative, as the fp32 w1)
          h = cc[s] * tot;
        }
        m = __builtin_amdgcn_frexp_mant(h);
        e = h > 0.0 ? E + __builtin_amdgcn_frexp_exp(h) : kFacEmptyExp;
        FacRec out;
        out.m = m; out.e = e; out.pad = 0;
        sH[t & 1][i + 1] = out;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      }
    }
#pragma unroll
    for (int s = 0; s < kFacBlinChunk; ++s) asm volatile("" : "+v"(cn[s]));   // (before the stores: see fac_fwd_plin)
#pragma unroll
    for (int s = 0; s < kFacBlinChunk; ++s) {
      const int t = t0 + s;
      if (valid && t >= 1 && t < T) w1b[(size_t)t * L + i] = wst[s];
    }
#pragma unroll
    for (int s = 0; s < kFacBlinChunk; ++s) cc[s] = cn[s];
  }
  // loss = scale * alpha_{T-1}[S-1],  alpha = (sum_t zmax_t + e + log2 m) ln 2 - A[y][y]; the wave that holds position S - 1 sums
  if ((tid >> 6) == ((S - 1) >> 6)) {   // wave-uniform
    const float* zb = ws.zmax + (size_t)b * T;
    double zs = 0.0;
    for (int t = tid & 63; t < T; t += 64) zs += (double)zb[t];
    zs = wave_sum_f64(zs);
    if (i == S - 1) {
      float out = -INFINITY;
      if (m > 0.0) {
        const double l2 = zs + (double)e + (double)__builtin_amdgcn_logf((float)m);
        out = (float)((double)sc * (l2 * 0.69314718055994530942 - (double)trans[(size_t)yi * N + yi]));
      }
      loss[b] = out;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// fac_fwd_plin / fac_bwd_plin: the waves of an utterance as a SKEWED PIPELINE -- no workgroup barrier at all.
// Measured (profiles/r04_run4 / run6): with one position per thread a frame is ~27 instructions (~175 cycles), but the
// s_barrier that keeps 5-6 waves in lockstep costs ~400 cycles per frame (fac_fwd_blin 0.50 ms, fac_bwd_blk<5,1> 0.35 ms at
// T = 2000): the barrier IS the kernel.  A position needs its LEFT neighbour's value of the PREVIOUS frame (forward) / its RIGHT
// neighbour's advance term of the same step (backward): inside a wave that is a DPP lane shift, and between waves it is ONE
// record per frame per wave boundary.  So wave w runs a whole chunk of kPlinChunk frames behind wave w - 1 (forward; ahead of
// it in the backward scan): the boundary records go through an LDS ring of kPlinRing frames, and the waves synchronise ONCE PER
// CHUNK through a progress word (poll until the leader has finished the chunk; the leader checks that its follower is less
// than three chunks behind before it overwrites ring slots).  Per frame that leaves: 3 DPP moves, one broadcast LDS read,
// one single-lane LDS write.  (The round-2 pipeline, fac_fwd_pipe, checked a tag EVERY frame and carried the log-domain
// arithmetic: 459 cycles per frame alone; it measured no faster than the barrier version.)  Every poll is bounded: a wave that
// never sees its leader poisons the loss (NaN) instead of hanging the GPU.  Label rows: fac_rows_k (pre-pass).
constexpr int kPlinChunk = 16;
constexpr int kPlinRing = 64;
constexpr int kPlinSpinMax = 1 << 18;   // ~10 ms: a legitimate wait is a few microseconds
// Range of fac_fwd_plin: a position's value is an fp64 mantissa with its own integer exponent, but one frame's update multiplies by
// kappa (fp64, any |log2| up to ~1020) and by the label weight c_t[y] = 2^(z - max z) (fp64, 0 below 2^-1074) BEFORE it is split
// again: tot = m + kappa m_left lies in [2^-1 kappa, 2^1021), h = c tot >= 2^-(1 + |log2 kappa| + spread).  While
// (largest |log2 kappa| of the target) + (largest per-frame spread of the label scores) <= kFacPlinSafeBits nothing leaves the
// normal range.  Beyond that (transition rows ~100 nats wide: profiles/r05_run31_asg_wide_transitions_after.log had loss -inf
// against a finite oracle) the utterance is flagged in ws.redo and recomputed by the log-domain kernel fac_fwd_blk.
constexpr float kFacPlinSafeBits = 900.f;

__device__ __forceinline__ bool plin_wait_ge(const int* p, int want) {   // poll *p >= want (relaxed LDS loads), bounded
  int spins = 0, v;
#pragma clang loop unroll(disable)
  do {
    v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (v >= want) return true;
    __builtin_amdgcn_s_sleep(1);
  } while (++spins < kPlinSpinMax);
  return false;
}
__device__ __forceinline__ bool plin_wait_le(const int* p, int want) {
  int spins = 0, v;
#pragma clang loop unroll(disable)
  do {
    v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (v <= want) return true;
    __builtin_amdgcn_s_sleep(1);
  } while (++spins < kPlinSpinMax);
  return false;
}

template <int NW>
__global__ __launch_bounds__(64 * NW) void fac_fwd_plin(int T, int N, int L, int scaleMode, const int* __restrict__ target,
                                                        const int* __restrict__ targetSize, const float* __restrict__ trans,
                                                        float* __restrict__ loss, FacWs ws) {
  __shared__ FacRec ring[NW][kPlinRing];   // ring[w][t & 63]: position 64 w + 63 after frame t
  __shared__ int prog[NW];                 // last frame wave w has finished
  __shared__ int bad;
  __shared__ int gbits;                    // largest |log2 kappa| over the target (bit pattern of a non-negative float)
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int S = targetSize[b];
  const float sc = scale_of(scaleMode, T, S);
  if (tid == 0) ws.scale[b] = sc;
  if (S <= 0) {
    if (tid == 0) { loss[b] = 0.f; ws.redo[b] = 0; }
    return;
  }
  if (tid < NW) prog[tid] = -1;
  if (tid == 0) { bad = 0; gbits = 0; }
  __syncthreads();
  const int lastWave = (S - 1) >> 6;
  if (wave > lastWave) return;           // nothing to do, and nobody waits for these waves
  const bool fed = wave > 0, feeds = wave < lastWave;
  const float NEG = -INFINITY;
  const int* y = target + (size_t)b * L;
  float* w1b = ws.w1 + (size_t)b * T * L;
  const int i = tid;
  const bool valid = i < S;
  const int yi = valid ? y[i] : 0;
  const int yp = (valid && i > 0) ? y[i - 1] : 0;
  const float dk = (valid && i > 0) ? trans[(size_t)yi * N + yp] - trans[(size_t)yp * N + yp] : NEG;
  const double kap = fac_exp_wide(dk);   // 0 for position 0 and beyond the target
  {   // range check, part 1 (before the first frame: the last wave reads it when the scan is over)
    const float kb = (valid && i > 0) ? fabsf(dk) * 1.44269504088896341f : 0.f;
    const float kw = wave_max(kb == kb ? kb : INFINITY);   // (a NaN or infinite transition: leave it to the log-domain kernel)
    if (lane == 0) atomicMax(&gbits, __float_as_int(kw));
  }
  const double* cb = ws.crow + (size_t)b * T * 32 + yi;
  double cc[kPlinChunk], cn[kPlinChunk];
#pragma unroll
  for (int s = 0; s < kPlinChunk; ++s) cc[s] = (valid && s < T) ? cb[(size_t)s * 32] : 0.0;
#pragma unroll
  for (int s = 0; s < kPlinChunk; ++s) asm volatile("" : "+v"(cc[s]));   // landed before the loop: with a load pending at the loop head
                                                                        // hipcc waits vmcnt(0) at every frame's first use -- i.e. for
                                                                        // the NEXT chunk's prefetch, one memory latency per chunk
  double m = 0.0;
  int e = kFacEmptyExp;
  bool ok = true;
  const FacRec* srcRing = &ring[fed ? wave - 1 : 0][0];
  FacRec* dstRing = &ring[wave][0];
  for (int t0 = 0; t0 < T; t0 += kPlinChunk) {
    const int tlast = min(t0 + kPlinChunk, T) - 1;
    if (fed) ok = plin_wait_ge(&prog[wave - 1], tlast) && ok;                                  // the leader has finished this chunk
    if (feeds && t0 >= kPlinRing - kPlinChunk) ok = plin_wait_ge(&prog[wave + 1], t0 - (kPlinRing - kPlinChunk)) && ok;   // ring slots free
#pragma unroll
    for (int s = 0; s < kPlinChunk; ++s) {
      const int tn = t0 + kPlinChunk + s;
      cn[s] = (valid && tn < T) ? cb[(size_t)tn * 32] : 0.0;
    }
    // the leader's records of frames t0 - 1 .. t0 + 14 (position 64 wave - 1 after the frame BEFORE each of this chunk's frames):
    // all sixteen at the top of the chunk -- a broadcast LDS read per frame, none of them on the recursion's chain
    double rm[kPlinChunk];
    int re[kPlinChunk];
#pragma unroll
    for (int s = 0; s < kPlinChunk; ++s) { rm[s] = 0.0; re[s] = kFacEmptyExp; }
    if (fed) {   // uniform
#pragma unroll
      for (int s = 0; s < kPlinChunk; ++s) {
        const FacRec r = srcRing[(t0 + s - 1) & (kPlinRing - 1)];   // (t0 = 0 never gets here with s = 0 used: frame 0 has no predecessor)
        rm[s] = r.m; re[s] = r.e;
      }
    }
    float wst[kPlinChunk];
    double pm[kPlinChunk];   // this lane's (m, e) after each frame: lane 63's are published at the end of the chunk
    int pe[kPlinChunk];
    auto frames = [&](auto full) {
#pragma unroll
      for (int s = 0; s < kPlinChunk; ++s) {
        const int t = t0 + s;
        wst[s] = 0.f; pm[s] = 0.0; pe[s] = kFacEmptyExp;
        if (decltype(full)::value || t < T) {   // uniform
          double h;
          int E;
          if (t == 0) {
            h = i == 0 ? cc[s] : 0.0;   // alpha_0[0] = x_0[y_0]
            E = 0;
          } else {
            // position i - 1 after frame t - 1: the lane below (DPP wave_shr:1); lane 0 keeps the `old` operand = the record
            const long long mb = __double_as_longlong(m), rb = __double_as_longlong(rm[s]);
            const int lo = __builtin_amdgcn_update_dpp((int)rb, (int)mb, 0x138, 0xf, 0xf, false);
            const int hi = __builtin_amdgcn_update_dpp((int)(rb >> 32), (int)(mb >> 32), 0x138, 0xf, 0xf, false);
            const double nm = __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
            const int ne = __builtin_amdgcn_update_dpp(re[s], e, 0x138, 0xf, 0xf, false);
            E = max(e, ne);
            const double ms = __builtin_amdgcn_ldexp(m, e - E);
            const double tot = fma(__builtin_amdgcn_ldexp(nm, ne - E), kap, ms);
            wst[s] = (float)(ms * __builtin_amdgcn_rcp(fmax(tot, 0x1p-1000)));   // (v_rcp_f64: 2^-23 relative, as the fp32 w1)
            h = cc[s] * tot;
          }
          m = __builtin_amdgcn_frexp_mant(h);
          e = h > 0.0 ? E + __builtin_amdgcn_frexp_exp(h) : kFacEmptyExp;
          pm[s] = m; pe[s] = e;
        }
      }
    };
    if (t0 + kPlinChunk <= T) frames(std::true_type{});
    else frames(std::false_type{});
    if (feeds) {   // the chunk's records (lane 63), then the progress word: the LDS executes a wave's operations in order
      if (lane == 63) {
        FacRec* d = dstRing + (t0 & (kPlinRing - 1));   // t0 is a multiple of the chunk: the chunk's slots are contiguous
#pragma unroll
        for (int s = 0; s < kPlinChunk; ++s) { FacRec out; out.m = pm[s]; out.e = pe[s]; out.pad = 0; d[s] = out; }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (lane == 0) __hip_atomic_store(&prog[wave], tlast, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    // the prefetched chunk is consumed BEFORE this chunk's stores are issued (a use behind the stores makes hipcc wait for the
    // stores' round trip too: the counter is shared and in order)
#pragma unroll
    for (int s = 0; s < kPlinChunk; ++s) asm volatile("" : "+v"(cn[s]));
#pragma unroll
    for (int s = 0; s < kPlinChunk; ++s) {
      const int t = t0 + s;
      if (valid && t >= 1 && t < T) w1b[(size_t)t * L + i] = wst[s];
    }
#pragma unroll
    for (int s = 0; s < kPlinChunk; ++s) cc[s] = cn[s];
  }
  if (!ok && lane == 0) __hip_atomic_store(&bad, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  // loss = scale * alpha_{T-1}[S-1],  alpha = (sum_t zmax_t + e + log2 m) ln 2 - A[y][y]; the wave that holds position S - 1 sums
  if (wave == lastWave) {
    const float* zb = ws.zmax + (size_t)b * T;
    const float* zp = ws.zspr + (size_t)b * T;
    double zs = 0.0;
    float spr = 0.f;
    for (int t = lane; t < T; t += 64) {
      zs += (double)zb[t];
      const float sp = zp[t];
      spr = fmaxf(spr, sp == sp ? sp : INFINITY);
    }
    zs = wave_sum_f64(zs);
    spr = wave_max(spr);
    if (i == S - 1) {
      float out = -INFINITY;
      if (m > 0.0) {
        const double l2 = zs + (double)e + (double)__builtin_amdgcn_logf((float)m);
        out = (float)((double)sc * (l2 * 0.69314718055994530942 - (double)trans[(size_t)yi * N + yi]));
      }
      // (the leaders finished before this wave did: their verdicts are in `bad`)
      loss[b] = (ok && !__hip_atomic_load(&bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) ? out : __builtin_nanf("");
      // range check, part 2: beyond kFacPlinSafeBits the log-domain kernel behind this one recomputes the utterance
      const float gb = __int_as_float(__hip_atomic_load(&gbits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
      ws.redo[b] = (spr + gb <= kFacPlinSafeBits) ? 0 : 1;
    }
  }
}

// backward scan as the mirrored pipeline: wave w + 1 leads wave w (a position needs its RIGHT neighbour's advance term of the
// same step); consumes w1[t][i], leaves g * dalpha_t[i] in ws.dal for fac_scatter_k
template <int NW, int KC = kPlinChunk>   // KC: frames per chunk (the ring holds four chunks)
__global__ __launch_bounds__(64 * NW) void fac_bwd_plin(int T, int N, int L, const int* __restrict__ target,
                                                        const int* __restrict__ targetSize, const float* __restrict__ grad,
                                                        float* __restrict__ transGrad, FacWs ws) {
  __shared__ float ring[NW][(4 * KC)];   // ring[w][t & 63]: advance term of position 64 w at step t
  __shared__ int prog[NW];                // lowest frame wave w has finished (steps run from T - 1 down)
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int S = targetSize[b];
  if (S <= 0) return;  // the scatter kernel zero-fills this utterance's gradient
  if (tid < NW) prog[tid] = 0x3fffffff;
  __syncthreads();
  const int lastWave = (S - 1) >> 6;
  const int* y = target + (size_t)b * L;
  const float* __restrict__ w1b = ws.w1 + (size_t)b * T * L;
  float* __restrict__ dalb = ws.dal + (size_t)b * T * L;
  const float g = ws.scale[b] * grad[b];
  const int i = tid;
  const bool valid = i < S;
  if (wave > lastWave) return;   // no mass here, nobody waits for these waves (fac_scatter_k reads positions < S only)
  const int yi = valid ? y[i] : 0;
  const int yp = (valid && i > 0) ? y[i - 1] : 0;
  float da = (i == S - 1) ? 1.f : 0.f, accS = 0.f, accP = 0.f;
  const bool fed = wave < lastWave, feeds = wave > 0;   // wave + 1 supplies lane 63's right neighbour; lane 0 supplies wave - 1
  bool ok = true;
  const float* srcRing = &ring[fed ? wave + 1 : 0][0];
  float* dstRing = &ring[wave][0];
  float wc[KC], wn[KC];
#pragma unroll
  for (int s = 0; s < KC; ++s) {
    const int t = T - 1 - s;
    wc[s] = (t >= 1 && valid) ? w1b[(size_t)t * L + i] : 0.f;
  }
#pragma unroll
  for (int s = 0; s < KC; ++s) asm volatile("" : "+v"(wc[s]));   // landed before the loop (see fac_fwd_plin)
  // chunks are aligned to multiples of KC from the TOP frame: step index k = T - 1 - t, ring slot k & 63
  for (int k0 = 0; k0 < T; k0 += KC) {
    const int thi = T - 1 - k0;
    const int tlow = max(thi - KC + 1, 0);
    if (fed) ok = plin_wait_le(&prog[wave + 1], tlow) && ok;                                        // the leader has finished this chunk
    if (feeds && k0 >= (4 * KC) - KC) ok = plin_wait_le(&prog[wave - 1], thi + ((4 * KC) - KC)) && ok;   // ring slots free
#pragma unroll
    for (int s = 0; s < KC; ++s) {
      const int t = thi - KC - s;
      wn[s] = (t >= 1 && valid) ? w1b[(size_t)t * L + i] : 0.f;
    }
    float rr[KC];   // the leader's advance terms of this chunk's steps (position 64 (wave + 1))
#pragma unroll
    for (int s = 0; s < KC; ++s) rr[s] = 0.f;
    if (fed) {   // uniform
      const float* sr = srcRing + (k0 & ((4 * KC) - 1));
#pragma unroll
      for (int s = 0; s < KC; ++s) rr[s] = sr[s];
    }
    float dstv[KC], pa[KC];
#pragma unroll
    for (int s = 0; s < KC; ++s) {
      const int t = thi - s;
      dstv[s] = g * da;   // row t of g * dalpha (0 beyond S)
      pa[s] = 0.f;
      if (t >= 1) {   // uniform
        const float st = da * wc[s];
        const float adv = da - st;
        accS += st;
        accP += adv;
        pa[s] = adv;
        // advance term of position i + 1: the lane above (DPP wave_shl:1); lane 63 keeps the `old` operand = the leader's
        const float right = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(rr[s]), __float_as_int(adv), 0x130, 0xf, 0xf, false));
        da = st + right;
      }
    }
    if (feeds) {
      if (lane == 0) {
        float* d = dstRing + (k0 & ((4 * KC) - 1));
#pragma unroll
        for (int s = 0; s < KC; ++s) d[s] = pa[s];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (lane == 0) __hip_atomic_store(&prog[wave], tlow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
    for (int s = 0; s < KC; ++s) asm volatile("" : "+v"(wn[s]));   // (before the stores: see fac_fwd_plin)
#pragma unroll
    for (int s = 0; s < KC; ++s) {
      const int t = thi - s;
      if (t >= 0 && i < L) dalb[(size_t)t * L + i] = ok ? dstv[s] : __builtin_nanf("");
    }
#pragma unroll
    for (int s = 0; s < KC; ++s) wc[s] = wn[s];
  }
  float* tg = ws.tgpart ? ws.tgpart + (size_t)b * N * N : transGrad;
  if (valid) {
    if (accS != 0.f) atomicAdd(&tg[(size_t)yi * N + yi], g * accS);
    if (i > 0 && accP != 0.f) atomicAdd(&tg[(size_t)yi * N + yp], g * accP);
  }
}

}  // namespace w2l
