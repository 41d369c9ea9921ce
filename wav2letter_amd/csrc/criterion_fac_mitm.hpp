// criterion_fac_mitm.hpp -- ForceAlignmentCriterion for N <= 32 labels and targets of up to 320 positions, MEET IN THE MIDDLE
// (round 6); included by criterion_fac.hip behind criterion_fac_lin.hpp, whose pipelined scaled-linear scans (fac_fwd_plin /
// fac_bwd_plin: one position per thread with its own exponent, the waves of an utterance as a skewed pipeline) it generalises.
//
// Replaces fl::lib::{cpu,cuda}::ForceAlignmentCriterion<float> (un-vendored; call sites recipes/slimIPL/src/Train.cpp:408-410,
// :1675; math SURVEY.md App. B.1; CPU restatement oracle/criterion_oracle.c).
//
// fac_fwd_plin is T dependent frames (335 us at T = 2000, L = 300), fac_bwd_plin another T (280 us).  The lattice has a forward
// and a backward recursion that do not depend on each other, so each pass runs both, from the two ends to the middle frame
// m = (T - 1) / 2, in two workgroups per utterance (fac_mitm_fwd / fac_mitm_bwd, grid (B, 2)):
//   forward, block 0 (alpha):  h_t[i] = c_t[y_i] (h_{t-1}[i] + kappa_i h_{t-1}[i-1]),  t = 0 .. m;  w1[t][i] = share of the stay term
//   forward, block 1 (beta):   G_t[i] = c_t[y_i] g_t[i],  g_t[i] = G_{t+1}[i] + kappa_{i+1} G_{t+1}[i+1],  t = T-1 .. m;
//                              w1[t+1][i] = share of the stay term in g_t[i]   (rows m+1 .. T-1: the share that LEAVES (t, i) upward)
//   fac_mitm_finish:           Z = sum_i h_m[i] g_m[i]  ->  loss;  gamma_m[i] = h_m[i] g_m[i] / Z;  the range check
//   backward, block 0 (down):  gamma_{t-1}[i] = gamma_t[i] w1[t][i] + gamma_t[i+1] (1 - w1[t][i+1]),    t = m .. 1   (fac_bwd_plin from m)
//   backward, block 1 (up):    gamma_t[i] = gamma_{t-1}[i] w1[t][i] + gamma_{t-1}[i-1] (1 - w1[t][i-1]),  t = m+1 .. T-1
// Both backward recursions are exp-free fp32 and start from the SAME posterior gamma_m; stay / advance masses are accumulated per
// position for the transition gradient as before, g gamma_t goes to ws.dal for fac_scatter_k.  T / 2 dependent frames per pass.
// The h and G recursions are one body (fac_half_fwd<BETA>): mirrored neighbour, mirrored pipeline direction, reversed frame order.
#pragma once
#include "criterion_fac_lin.hpp"

namespace w2l {

// the middle frame: alpha takes 6 / 11 of the frames -- 168 ns per frame against the beta half's 202 (profiles/r06_run10_*)
__host__ __device__ inline int fac_mitm_mid(int T) { return (int)(((long long)(T - 1) * 6) / 11); }

// One half of the forward pass.  Frames are counted k = 0 .. nK - 1 from the half's own end of the utterance: alpha k <-> frame k,
// beta k <-> frame T - 1 - k; k = 0 is the initial row.  ring[w][k & 63] = the boundary position of wave w after frame k.
template <int NW, bool BETA>
__device__ __forceinline__ void fac_half_fwd(int T, int N, int L, const int* __restrict__ target, const int* __restrict__ targetSize,
                                             const float* __restrict__ trans, const FacWs& ws, FacRec (*ring)[kPlinRing], int* prog, int abl = 0) {
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int S = targetSize[b];
  const int m0 = fac_mitm_mid(T);
  const int nK = BETA ? T - m0 : m0 + 1;
  const int lastWave = (S - 1) >> 6;
  if (wave > lastWave) return;   // nothing to do, and nobody waits for these waves
  const bool fed = BETA ? wave < lastWave : wave > 0;
  const bool feeds = BETA ? wave > 0 : wave < lastWave;
  const int lead = BETA ? wave + 1 : wave - 1, foll = BETA ? wave - 1 : wave + 1;
  const float NEG = -INFINITY;
  const int* y = target + (size_t)b * L;
  float* w1b = ws.w1 + (size_t)b * T * L;
  const int i = tid;
  const bool valid = i < S;
  const int yi = valid ? y[i] : 0;
  float dk;
  if (!BETA) {   // kappa of the step INTO position i from i - 1
    const int yp = (valid && i > 0) ? y[i - 1] : 0;
    dk = (valid && i > 0) ? trans[(size_t)yi * N + yp] - trans[(size_t)yp * N + yp] : NEG;
  } else {       // kappa of the step OUT of position i into i + 1
    const bool nx = i + 1 < S;
    const int yn = nx ? y[i + 1] : 0;
    dk = nx ? trans[(size_t)yn * N + yi] - trans[(size_t)yi * N + yi] : NEG;
  }
  const double kap = fac_exp_wide(dk);   // 0 where there is no such step
  const double* cb = ws.crow + (size_t)b * T * 32 + yi;
  // label weight of frame k: c of the frame -- except beta's last one: g_m carries no emission of the middle frame (h_m does)
  auto cw = [&](int k) -> double {
    if (!valid || k >= nK) return 0.0;
    if (abl & 1) return 0.75;   // (probe, timing only: no label-weight loads)
    if (BETA) return k == nK - 1 ? 1.0 : cb[(size_t)(T - 1 - k) * 32];
    return cb[(size_t)k * 32];
  };
  double cc[kPlinChunk], cn[kPlinChunk];
#pragma unroll
  for (int s = 0; s < kPlinChunk; ++s) cc[s] = cw(s);
#pragma unroll
  for (int s = 0; s < kPlinChunk; ++s) asm volatile("" : "+v"(cc[s]));   // landed before the loop (see fac_fwd_plin)
  double m = 0.0;
  int e = kFacEmptyExp;
  bool ok = true;
  const FacRec* srcRing = &ring[fed ? lead : 0][0];
  FacRec* dstRing = &ring[wave][0];
  for (int k0 = 0; k0 < nK; k0 += kPlinChunk) {
    const int klast = min(k0 + kPlinChunk, nK) - 1;
    if (fed) ok = plin_wait_ge(&prog[lead], klast) && ok;                                                       // the leader has finished this chunk
    if (feeds && k0 >= kPlinRing - kPlinChunk) ok = plin_wait_ge(&prog[foll], k0 - (kPlinRing - kPlinChunk)) && ok;   // ring slots free
#pragma unroll
    for (int s = 0; s < kPlinChunk; ++s) cn[s] = cw(k0 + kPlinChunk + s);
    double rm[kPlinChunk];
    int re[kPlinChunk];
#pragma unroll
    for (int s = 0; s < kPlinChunk; ++s) { rm[s] = 0.0; re[s] = kFacEmptyExp; }
    if (fed) {   // uniform: the leader's boundary records after the frame BEFORE each of this chunk's frames
#pragma unroll
      for (int s = 0; s < kPlinChunk; ++s) {
        const FacRec r = srcRing[(k0 + s - 1) & (kPlinRing - 1)];
        rm[s] = r.m; re[s] = r.e;
      }
    }
    float wst[kPlinChunk];
    double pm[kPlinChunk];
    int pe[kPlinChunk];
    auto frames = [&](auto full) {
#pragma unroll
      for (int s = 0; s < kPlinChunk; ++s) {
        const int k = k0 + s;
        wst[s] = 0.f; pm[s] = 0.0; pe[s] = kFacEmptyExp;
        if (decltype(full)::value || k < nK) {   // uniform
          double h;
          int E;
          if (k == 0) {
            h = (BETA ? i == S - 1 : i == 0) ? cc[s] : 0.0;
            E = 0;
          } else {
            // the neighbour after frame k - 1: alpha the lane below (wave_shr:1, lane 0 keeps the ring record), beta the lane above
            // (wave_shl:1, lane 63 keeps it)
            const long long mb = __double_as_longlong(m), rb = __double_as_longlong(rm[s]);
            int lo, hi, ne;
            if (BETA) {
              lo = __builtin_amdgcn_update_dpp((int)rb, (int)mb, 0x130, 0xf, 0xf, false);
              hi = __builtin_amdgcn_update_dpp((int)(rb >> 32), (int)(mb >> 32), 0x130, 0xf, 0xf, false);
              ne = __builtin_amdgcn_update_dpp(re[s], e, 0x130, 0xf, 0xf, false);
            } else {
              lo = __builtin_amdgcn_update_dpp((int)rb, (int)mb, 0x138, 0xf, 0xf, false);
              hi = __builtin_amdgcn_update_dpp((int)(rb >> 32), (int)(mb >> 32), 0x138, 0xf, 0xf, false);
              ne = __builtin_amdgcn_update_dpp(re[s], e, 0x138, 0xf, 0xf, false);
            }
            const double nm = __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
            E = max(e, ne);
            const double ms = __builtin_amdgcn_ldexp(m, e - E);
            const double tot = fma(__builtin_amdgcn_ldexp(nm, ne - E), kap, ms);
            wst[s] = (float)(ms * __builtin_amdgcn_rcp(fmax(tot, 0x1p-1000)));   // share of the stay term (v_rcp_f64: 2^-23 relative)
            h = cc[s] * tot;
          }
          m = __builtin_amdgcn_frexp_mant(h);
          e = h > 0.0 ? E + __builtin_amdgcn_frexp_exp(h) : kFacEmptyExp;
          pm[s] = m; pe[s] = e;
        }
      }
    };
    if (k0 + kPlinChunk <= nK) frames(std::true_type{});
    else frames(std::false_type{});
    if (feeds) {
      if (lane == (BETA ? 0 : 63)) {
        FacRec* d = dstRing + (k0 & (kPlinRing - 1));
#pragma unroll
        for (int s = 0; s < kPlinChunk; ++s) { FacRec out; out.m = pm[s]; out.e = pe[s]; out.pad = 0; d[s] = out; }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (lane == 0) __hip_atomic_store(&prog[wave], klast, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
    for (int s = 0; s < kPlinChunk; ++s) asm volatile("" : "+v"(cn[s]));
#pragma unroll
    for (int s = 0; s < kPlinChunk; ++s) {
      const int k = k0 + s;
      // alpha: row k = frame k; beta: the share computed at k belongs to the step between frames T - k and T - 1 - k: row T - k
      if (valid && k >= 1 && k < nK && !(abl & 2)) w1b[(size_t)(BETA ? T - k : k) * L + i] = wst[s];
    }
#pragma unroll
    for (int s = 0; s < kPlinChunk; ++s) cc[s] = cn[s];
  }
  if (valid) {
    FacRec out;
    out.m = ok ? m : (double)__builtin_nanf("");   // a wave that never saw its leader poisons the loss instead of hanging the GPU
    out.e = e; out.pad = 0;
    (BETA ? ws.gm : ws.hm)[(size_t)b * 320 + i] = out;
  }
}

template <int NW>
__global__ __launch_bounds__(64 * NW) void fac_mitm_fwd(int T, int N, int L, const int* __restrict__ target,
                                                        const int* __restrict__ targetSize, const float* __restrict__ trans, FacWs ws, int dir0 = 0, int abl = 0) {
  __shared__ FacRec ring[NW][kPlinRing];
  __shared__ int prog[NW];
  if (targetSize[blockIdx.x] <= 0) return;   // fac_mitm_finish writes loss 0
  if (threadIdx.x < NW) prog[threadIdx.x] = -1;
  __syncthreads();
  if (blockIdx.y + dir0 == 0) fac_half_fwd<NW, false>(T, N, L, target, targetSize, trans, ws, ring, prog, abl);
  else fac_half_fwd<NW, true>(T, N, L, target, targetSize, trans, ws, ring, prog, abl);
}

// Z, loss, the middle frame's posterior and the range check (see kFacPlinSafeBits); one workgroup per utterance: every wave sums a
// share of the frame maxima / spreads (T loads from one wave were 32 dependent round trips: 16 us), wave 0 combines the records
constexpr int kFacFinishThreads = 512;
__device__ __forceinline__ void fac_mitm_finish_body(int T, int N, int L, int scaleMode, const int* __restrict__ target,
                                                      const int* __restrict__ targetSize, const float* __restrict__ trans,
                                                      float* loss, const FacWs& ws) {
  __shared__ double sZs[kFacFinishThreads / 64];
  __shared__ float sSp[kFacFinishThreads / 64];
  const int b = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int S = targetSize[b];
  const float sc = scale_of(scaleMode, T, S);
  if (threadIdx.x == 0) ws.scale[b] = sc;
  if (S <= 0) {
    if (threadIdx.x == 0) { loss[b] = 0.f; ws.redo[b] = 0; }
    return;
  }
  {
    const float* zb = ws.zmax + (size_t)b * T;
    const float* zp = ws.zspr + (size_t)b * T;
    double zs = 0.0;
    float spr = 0.f;
    for (int t = threadIdx.x; t < T; t += kFacFinishThreads) {
      zs += (double)zb[t];
      const float sp = zp[t];
      spr = fmaxf(spr, sp == sp ? sp : INFINITY);
    }
    zs = wave_sum_f64(zs);
    spr = wave_max(spr);
    if (lane == 0) { sZs[wv] = zs; sSp[wv] = spr; }
  }
  __syncthreads();
  if (wv != 0) return;
  const int* y = target + (size_t)b * L;
  constexpr int P = 5;   // positions per lane (L <= 320)
  double pm[P];
  int pe[P];
  int E = kFacEmptyExp;
  float kb = 0.f;
  bool nan = false;
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const int i = lane + 64 * p;
    pm[p] = 0.0; pe[p] = kFacEmptyExp;
    if (i < S) {
      const FacRec h = ws.hm[(size_t)b * 320 + i], g = ws.gm[(size_t)b * 320 + i];
      nan = nan || h.m != h.m || g.m != g.m;
      if (h.m > 0.0 && g.m > 0.0) { pm[p] = h.m * g.m; pe[p] = h.e + g.e; }
      if (i > 0) {
        const int yi = y[i], yp = y[i - 1];
        const float d = fabsf(trans[(size_t)yi * N + yp] - trans[(size_t)yp * N + yp]) * 1.44269504088896341f;
        kb = fmaxf(kb, d == d ? d : INFINITY);
      }
    }
    E = max(E, pe[p]);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) E = max(E, __shfl_xor(E, off));
  double z = 0.0;
#pragma unroll
  for (int p = 0; p < P; ++p) {
    int sh = pe[p] - E;
    sh = sh < -2000 ? -2000 : sh;
    pm[p] = pe[p] > kFacEmptyExp ? __builtin_amdgcn_ldexp(pm[p], sh) : 0.0;
    z += pm[p];
  }
  z = wave_sum_f64(z);
  const double zinv = z > 0.0 ? 1.0 / z : 0.0;
  float* gam = ws.gam + (size_t)b * 320;
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const int i = lane + 64 * p;
    if (i < 320) gam[i] = i < S ? (float)(pm[p] * zinv) : 0.f;
  }
  double zs = 0.0;
  float spr = 0.f;
#pragma unroll
  for (int w = 0; w < kFacFinishThreads / 64; ++w) { zs += sZs[w]; spr = fmaxf(spr, sSp[w]); }
  kb = wave_max(kb);
  nan = __any(nan) != 0;
  if (lane == 0) {
    float out = -INFINITY;
    if (z > 0.0) {
      const int ex = __builtin_amdgcn_frexp_exp(z);
      const float mant = (float)__builtin_amdgcn_frexp_mant(z);
      const double l2 = zs + (double)E + (double)ex + (double)__builtin_amdgcn_logf(mant);
      const int yl = y[S - 1];
      out = (float)((double)sc * (l2 * 0.69314718055994530942 - (double)trans[(size_t)yl * N + yl]));
    }
    loss[b] = nan ? __builtin_nanf("") : out;
    // range check: beyond kFacPlinSafeBits the log-domain kernel behind this one recomputes the utterance (loss and every w1 row)
    ws.redo[b] = (spr + kb <= kFacPlinSafeBits) ? 0 : 1;
  }
}
__global__ __launch_bounds__(kFacFinishThreads) void fac_mitm_finish(int T, int N, int L, int scaleMode, const int* __restrict__ target,
                                                      const int* __restrict__ targetSize, const float* __restrict__ trans,
                                                      float* __restrict__ loss, FacWs ws) {
  fac_mitm_finish_body(T, N, L, scaleMode, target, targetSize, trans, loss, ws);
}

// One half of the backward pass: UP = false: frames Ttop .. 0 (gamma_{t-1} from gamma_t, the leader is the wave above);
// UP = true: frames m+1 .. T-1 (gamma_t from gamma_{t-1}, the leader is the wave below).  Steps are counted k = 0, 1, ... from the
// half's starting frame; ring[w][k & 63] = the advance term of wave w's boundary position at step k.
template <int NW, bool UP>
__device__ __forceinline__ void fac_half_bwd(int T, int N, int L, const int* __restrict__ target, const int* __restrict__ targetSize,
                                             const float* __restrict__ grad, float* __restrict__ transGrad, const FacWs& ws,
                                             float (*ring)[4 * kPlinChunk], int* prog, int Ttop, bool oneHot) {
  constexpr int KC = kPlinChunk;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int S = targetSize[b];
  const int lastWave = (S - 1) >> 6;
  const int* y = target + (size_t)b * L;
  const float* __restrict__ w1b = ws.w1 + (size_t)b * T * L;
  float* __restrict__ dalb = ws.dal + (size_t)b * T * L;
  const float g = ws.scale[b] * grad[b];
  const int i = tid;
  const bool valid = i < S;
  if (wave > lastWave) return;
  const int yi = valid ? y[i] : 0;
  const int yp = (valid && i > 0) ? y[i - 1] : 0;
  float da = oneHot ? ((i == S - 1) ? 1.f : 0.f) : (valid ? ws.gam[(size_t)b * 320 + i] : 0.f);
  float accS = 0.f, accP = 0.f;
  const bool fed = UP ? wave > 0 : wave < lastWave;
  const bool feeds = UP ? wave < lastWave : wave > 0;
  const int lead = UP ? wave - 1 : wave + 1, foll = UP ? wave + 1 : wave - 1;
  // step k uses row: DOWN t = Ttop - k (while t >= 1); UP t = Ttop + 1 + k (while t <= T - 1)
  const int nSteps = UP ? T - 1 - Ttop : Ttop;
  auto row = [&](int k) { return UP ? Ttop + 1 + k : Ttop - k; };
  bool ok = true;
  const float* srcRing = &ring[fed ? lead : 0][0];
  float* dstRing = &ring[wave][0];
  float wc[KC], wn[KC];
#pragma unroll
  for (int s = 0; s < KC; ++s) wc[s] = (s < nSteps && valid) ? w1b[(size_t)row(s) * L + i] : 0.f;
#pragma unroll
  for (int s = 0; s < KC; ++s) asm volatile("" : "+v"(wc[s]));
  // DOWN writes g gamma of the frame BEFORE each step (rows Ttop .. 0: nSteps + 1 rows); UP of the frame AFTER it (rows Ttop+1 .. T-1)
  const int nRows = UP ? nSteps : nSteps + 1;
  for (int k0 = 0; k0 < nRows; k0 += KC) {
    const int klast = min(k0 + KC, nRows) - 1;
    if (fed) ok = plin_wait_ge(&prog[lead], klast) && ok;
    if (feeds && k0 >= 4 * KC - KC) ok = plin_wait_ge(&prog[foll], k0 - (4 * KC - KC)) && ok;
#pragma unroll
    for (int s = 0; s < KC; ++s) {
      const int k = k0 + KC + s;
      wn[s] = (k < nSteps && valid) ? w1b[(size_t)row(k) * L + i] : 0.f;
    }
    float rr[KC];
#pragma unroll
    for (int s = 0; s < KC; ++s) rr[s] = 0.f;
    if (fed) {
      const float* sr = srcRing + (k0 & (4 * KC - 1));
#pragma unroll
      for (int s = 0; s < KC; ++s) rr[s] = sr[s];
    }
    float dstv[KC], pa[KC];
#pragma unroll
    for (int s = 0; s < KC; ++s) {
      const int k = k0 + s;
      if (!UP) dstv[s] = g * da;
      pa[s] = 0.f;
      if (k < nSteps) {   // uniform
        const float st = da * wc[s];
        const float adv = da - st;
        accS += st;
        pa[s] = adv;
        float nb;
        if (UP) nb = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(rr[s]), __float_as_int(adv), 0x138, 0xf, 0xf, false));   // from i - 1
        else nb = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(rr[s]), __float_as_int(adv), 0x130, 0xf, 0xf, false));      // from i + 1
        accP += UP ? nb : adv;   // the advance INTO position i (from i - 1): received (up) / split off (down)
        da = st + nb;
      }
      if (UP) dstv[s] = g * da;
    }
    if (feeds) {
      if (lane == (UP ? 63 : 0)) {
        float* d = dstRing + (k0 & (4 * KC - 1));
#pragma unroll
        for (int s = 0; s < KC; ++s) d[s] = pa[s];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (lane == 0) __hip_atomic_store(&prog[wave], klast, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
    for (int s = 0; s < KC; ++s) asm volatile("" : "+v"(wn[s]));
#pragma unroll
    for (int s = 0; s < KC; ++s) {
      const int k = k0 + s;
      if (k < nRows && i < L) dalb[(size_t)(UP ? Ttop + 1 + k : Ttop - k) * L + i] = ok ? dstv[s] : __builtin_nanf("");
    }
#pragma unroll
    for (int s = 0; s < KC; ++s) wc[s] = wn[s];
  }
  // (the upward half has a partial set of its own behind the B sets of the downward halves: see fac_partial_sets)
  float* tg = ws.tgpart ? ws.tgpart + ((size_t)(UP ? gridDim.x : 0) + b) * N * N : transGrad;
  if (valid) {
    if (accS != 0.f) atomicAdd(&tg[(size_t)yi * N + yi], g * accS);
    if (i > 0 && accP != 0.f) atomicAdd(&tg[(size_t)yi * N + yp], g * accP);
  }
}

// positions of an utterance sorted by label for fac_scatter_csr_k (criterion_fac.hip, where fac_csr_k is this body as a launch)
__device__ __forceinline__ void fac_csr_body(int N, int L, const int* __restrict__ target, const int* __restrict__ targetSize,
                                                int* __restrict__ csr) {
  __shared__ int cnt[8][64];
  __shared__ int offs[65];
  const int b = blockIdx.x, i = threadIdx.x, lane = i & 63, wave = i >> 6;
  const int nWaves = (int)(blockDim.x >> 6);   // 8 in fac_csr_k; ceil(L / 64) inside the backward scan launch (thread = position either way)
  const int S = min(targetSize[b], L);
  int* pos = csr + (size_t)b * (L + 68);
  int* off = pos + L;
  if (S <= 0) return;
  const int yi = i < S ? target[(size_t)b * L + i] : -1;
  const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
  int rank = 0;
  for (int n = 0; n < N; ++n) {
    const unsigned long long m = __ballot(yi == n);
    if (lane == 0) cnt[wave][n] = __popcll(m);
    if (yi == n) rank = __popcll(m & below);
  }
  __syncthreads();
  if (i <= N) {   // off[n] = positions with a smaller label
    int less = 0;
    for (int n = 0; n < i; ++n)
      for (int w = 0; w < nWaves; ++w) less += cnt[w][n];
    offs[i] = less;
    off[i] = less;
  }
  __syncthreads();
  if (yi >= 0 && yi < N) {
    int base = offs[yi];
    for (int w = 0; w < wave; ++w) base += cnt[w][yi];
    pos[base + rank] = i;
  }
}


template <int NW>
__global__ __launch_bounds__(64 * NW) void fac_mitm_bwd(int T, int N, int L, const int* __restrict__ target,
                                                        const int* __restrict__ targetSize, const float* __restrict__ grad,
                                                        float* __restrict__ transGrad, FacWs ws, int dir0 = 0, int csrInline = 0) {
  __shared__ float ring[NW][4 * kPlinChunk];
  __shared__ int prog[NW];
  const int b = blockIdx.x;
  // csrInline (the fused ASG sequence): the utterance's upward block sorts the positions by label for the scatter launch behind this
  // one before it starts its half (the shorter one) -- no fac_csr_k launch in front of the scans
  if (csrInline && blockIdx.y + dir0 == 1) {
    fac_csr_body(N, L, target, targetSize, ws.csr);
    __syncthreads();
  }
  if (targetSize[b] <= 0) return;   // the scatter kernel zero-fills this utterance's gradient
  if (threadIdx.x < NW) prog[threadIdx.x] = -1;
  __syncthreads();
  const bool flagged = ws.redo[b] != 0;   // recomputed by fac_fwd_blk: every w1 row is a forward share -> the classic scan from T - 1
  const int m = fac_mitm_mid(T);
  if (blockIdx.y + dir0 == 0) fac_half_bwd<NW, false>(T, N, L, target, targetSize, grad, transGrad, ws, ring, prog, flagged ? T - 1 : m, flagged);
  else if (!flagged) fac_half_bwd<NW, true>(T, N, L, target, targetSize, grad, transGrad, ws, ring, prog, m, false);
}

}  // namespace w2l
