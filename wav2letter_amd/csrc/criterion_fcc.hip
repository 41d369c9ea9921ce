// criterion_fcc.hip -- FullConnectionCriterion + ViterbiPath for gfx950.
//
// Replaces Flashlight's fl::lib::cuda::FullConnectionCriterion<float> /
// ViterbiPath<float> (un-vendored; one kernel launch PER TIME STEP there, see
// SURVEY.md 2.4).  Call sites in the reference: recipes/slimIPL/src/Train.cpp:410
// (ASGLoss), :1675 (forward), :838 (viterbiPath).  Math: SURVEY.md App. B.2/B.3,
// CPU restatement: oracle/criterion_oracle.c.
//
// Small-N path (N <= 64, the ASG letter/phone case, N = 30 for LibriSpeech):
// ONE WAVEFRONT PER UTTERANCE scans T inside a single launch.
//   lane i <-> state i.  exp(A[i][j] - rowmax_i) lives in NP registers per lane.
//   Step: e_j = exp(ahat_{t-1}[j]) (one v_exp per lane); s_i = sum_j EA[i][j] e_j
//   with e_j broadcast by v_readlane into an SGPR operand of v_fmac (no LDS, no
//   barrier); ahat_t[i] = x_t[i] + rowmax_i + log s_i - c_t, c_t = max_i (DPP).
//   The running offset sum_t c_t is kept in fp64 so the fp32 recursion never
//   carries O(T) magnitudes (ASG = FCC - FAC cancellation, SURVEY 7 hard part 2b).
//   Emission rows are read as coalesced [t][0..N) rows, prefetched 8 steps ahead.
// Workspace keeps ahat [B][T][N] and log s [B][T][N] (fp32) for backward, which
// then needs neither the emissions nor any exp-domain re-summation:
//   w_ij = EA[i][j] e_j / s_i.
#include "common.hpp"
#include "criterion_asg_fused.hpp"

namespace w2l {

constexpr int kChunk = 16;  // frames per prefetch chunk; a chunk's loads are consumed and its rows stored at the chunk
                           // boundary: ONE vmcnt drain per chunk (on gfx9 that counter also holds the frames' global stores)
// The scaled-exp scans hold s_t[i] = sum_j exp(A[i][j] - rowmax_i) exp(ahat_{t-1}[j]) as a sum of fp32 products of two factors in
// (0, 1]: a factor below 1e-38 is flushed, so what the sum can have lost is at most NP * 1.2e-38.  While every sum of an utterance
// stays above kFccMinSum that loss is below 1e-8 relative (and, in the backward scan, every weight exp(A) e / s that an underflow
// zeroed is below 1e-10): the scan is exact to fp32.  An utterance with a smaller sum is flagged in ws.redo and recomputed in the
// log domain, as the reference computes every utterance.
constexpr float kFccMinSum = 1e-28f;
constexpr int kDtChunks = 64;  // time chunks of the (parallel) transition-gradient kernel (16 chunks: 85 us behind the scan at T = 2000)

struct FccWs {
  float* ahat;   // [B][T][N]
  float* logs;   // [B][T][N]
  float* r;      // [B][T][N]  r_t = dalpha_t / s_t of the backward scan, for fcc_dtrans_small (its own buffer: written
                 //            over `logs` the stores alias the scan's prefetch loads and cost a vmcnt(0) per frame)
  float* scale;  // [B]
  float* tgpart; // [B][kDtChunks][N][N] transition-gradient partials (utterance x time chunk)
  float* r2;     // [B][T][N]  N <= 31, meet in the middle: b_t q'_t of the forward pass's beta half (frames above the middle)
  double* half;  // [B][2]  ... : base-2 scale sums of the alpha half (frames 0 .. m) and of the beta half (m+1 .. T-1)
  float* bm;     // [B][32] ... : b_m of the beta half, by state
  float* ginv;   // [B]     ... : 1 / sum_i u_m[i] b_m[i]
  int* redo;     // [B]  1 = this utterance is outside what the fp32 scaled-domain scan holds exactly -- N <= 31: transition rows spread
                 //      over more than kFccSafeSpread nats (fcc_fwd_dpp sets it); 32 <= N <= 64: some state's sum fell under
                 //      kFccMinSum (fcc_fwd_small sets it) -- and runs on the log-domain pair fcc_fwd_log / fcc_bwd_log instead
};

__host__ __device__ inline FccWs fcc_ws(void* ws, int B, int T, int N) {
  FccWs w;
  char* p = (char*)ws;
  size_t btn = align_up((size_t)B * T * N * sizeof(float), 256);
  w.ahat = (float*)p; p += btn;
  w.logs = (float*)p; p += btn;
  w.r = (float*)p; p += btn;
  w.scale = (float*)p; p += align_up((size_t)B * sizeof(float), 256);
  w.tgpart = (float*)p; p += align_up((size_t)B * kDtChunks * N * N * sizeof(float), 256);
  w.redo = (int*)p; p += align_up((size_t)B * sizeof(int), 256);
  w.r2 = nullptr; w.half = nullptr; w.bm = nullptr; w.ginv = nullptr;
  if (N <= 31) {
    w.r2 = (float*)p; p += btn;
    w.half = (double*)p; p += align_up((size_t)B * 2 * sizeof(double), 256);
    w.bm = (float*)p; p += align_up((size_t)B * 32 * sizeof(float), 256);
    w.ginv = (float*)p;
  }
  return w;
}

}  // namespace w2l

#include "criterion_asg_dpp.hpp"   // N <= 31: scaled linear domain on DPP row rotations (fcc_fwd_dpp, fcc_bwd_dpp, vit_fwd_dpp, vit_psi_k, vit_walk_k)
#include "criterion_asg_mitm.hpp"  // N <= 31, the product: the same scans from both ends to the middle frame (fcc_mitm_fwd, fcc_mitm_bwd)

namespace w2l {

// N <= 31 runs the DPP kernels; the probe library can put the previous generation back for A/B work (W2L_ASG_OLD=1)
inline bool asg_dpp_path(int N) {
  static const bool old = tune_env("W2L_ASG_OLD") != nullptr;
  return N <= 31 && !old;
}
// ... and, for A/B work, the round-4 full-length scans of criterion_asg_dpp.hpp instead of the meet-in-the-middle pair (W2L_ASG_NOMITM=1)
// probe, timing only (results are wrong): W2L_MITM_ONLY=0 / 1 launches one half of every meet-in-the-middle kernel alone
inline int mitm_only() {
  static const int v = [] { const char* e = tune_env("W2L_MITM_ONLY"); return e ? (e[0] == '1' ? 1 : 0) : -1; }();
  return v;
}
inline bool asg_mitm_path() {
  static const bool off = tune_env("W2L_ASG_NOMITM") != nullptr || tune_env("W2L_FCC_1WAVE") != nullptr;
  return !off;
}

template <int NP>
__global__ __launch_bounds__(64) void fcc_fwd_small(int T, int N, int scaleMode,
                                                    const float* __restrict__ x,
                                                    const int* __restrict__ targetSize,
                                                    const float* __restrict__ trans,
                                                    float* __restrict__ loss, FccWs ws, const int* __restrict__ redo = nullptr) {
  const int b = blockIdx.x;
  const int lane = threadIdx.x;
  if (redo && !redo[b]) return;   // launched behind fcc_fwd_dpp: only the utterances it flagged
  const bool act = lane < N;
  const float NEG = -INFINITY;

  float EA[NP];
  float rowmax = NEG;
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    float a = (act && j < N) ? trans[(size_t)lane * N + j] : NEG;
    EA[j] = a;
    rowmax = fmaxf(rowmax, a);
  }
  if (!act) rowmax = 0.f;
#pragma unroll
  for (int j = 0; j < NP; ++j) EA[j] = (act && j < N) ? __expf(EA[j] - rowmax) : 0.f;

  const float* xb = x + (size_t)b * T * N;
  float* ahb = ws.ahat + (size_t)b * T * N;
  float* lsb = ws.logs + (size_t)b * T * N;

  float xc[kChunk], xn[kChunk];
#pragma unroll
  for (int u = 0; u < kChunk; ++u) xc[u] = (act && u < T) ? xb[(size_t)u * N + lane] : 0.f;

  float ah = 0.f;
  float smin = INFINITY;   // smallest sum of this lane's state over the frames (range check, off the chain)
  double C = 0.0;
  for (int t0 = 0; t0 < T; t0 += kChunk) {
#pragma unroll
    for (int u = 0; u < kChunk; ++u) {
      int tn = t0 + kChunk + u;
      xn[u] = (act && tn < T) ? xb[(size_t)tn * N + lane] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < kChunk; ++u) {
      const int t = t0 + u;
      if (t < T) {  // wave-uniform
        float a, ls = 0.f;
        if (t == 0) {
          a = act ? xc[u] : NEG;
        } else {
          float e = act ? __expf(ah) : 0.f;
          // two packed accumulators: v_pk_fma_f32 takes the pair (e_j, e_j+1) from an SGPR pair, 16 packed FMAs
          // instead of 32 scalar ones behind the 32 v_readlane broadcasts
          typedef float f32x2_t __attribute__((ext_vector_type(2)));
          f32x2_t s0 = {0.f, 0.f}, s1 = {0.f, 0.f};   // (four chains measured slower: 0.60 vs 0.54 ms at T = 2000)
#pragma unroll
          for (int j = 0; j < NP; j += 4) {
            const f32x2_t e01 = {readlane(e, j), readlane(e, j + 1)}, e23 = {readlane(e, j + 2), readlane(e, j + 3)};
            const f32x2_t a01 = {EA[j], EA[j + 1]}, a23 = {EA[j + 2], EA[j + 3]};
            s0 = __builtin_elementwise_fma(a01, e01, s0);
            s1 = __builtin_elementwise_fma(a23, e23, s1);
          }
          const float sraw = (s0.x + s0.y) + (s1.x + s1.y);
          smin = act ? (sraw >= smin ? smin : sraw) : smin;   // (a NaN sum lands in smin and fails the check below)
          float s = fmaxf(sraw, 1e-37f);
          ls = fast_logf(s);
          a = act ? (xc[u] + rowmax + ls) : NEG;
        }
        float c = wave_max_rows<(NP > 32 ? 4 : 2)>(a);
        ah = a - c;
        C += (double)c;
        if (act) {
          ahb[(size_t)t * N + lane] = ah;
          lsb[(size_t)t * N + lane] = ls;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < kChunk; ++u) xc[u] = xn[u];
  }
  float e = act ? __expf(ah) : 0.f;
  float tot = wave_sum(e);
  float sc = scale_of(scaleMode, T, targetSize[b]);
  // range check: every sum of every state stayed above kFccMinSum (and was a number), else the log-domain kernel behind this one
  // recomputes the utterance.  (Launched as the fallback of an older generation -- `redo` given -- there is nothing behind it.)
  const bool low = __any(act && !(smin >= kFccMinSum)) != 0;
  if (lane == 0) {
    loss[b] = (float)((double)sc * (C + (double)__logf(tot)));
    ws.scale[b] = sc;
    if (!redo) ws.redo[b] = low ? 1 : 0;
  }
}

template <int NP>
__global__ __launch_bounds__(64) void fcc_bwd_small(int T, int N, const float* __restrict__ trans,
                                                    const float* __restrict__ grad,
                                                    float* __restrict__ inputGrad, FccWs ws, const int* __restrict__ redo = nullptr) {
  const int b = blockIdx.x;
  const int lane = threadIdx.x;
  if (redo && !redo[b]) return;
  if (!redo && ws.redo[b]) return;   // flagged by fcc_fwd_small: fcc_bwd_log, launched behind this kernel, differentiates it
  const bool act = lane < N;
  const float NEG = -INFINITY;

  // rowmax_i in lane i, then EAT[i] = exp(A[i][lane] - rowmax_i) (column `lane`)
  float rowmax = NEG;
#pragma unroll
  for (int j = 0; j < NP; ++j) {  // unrolled: NP independent loads, one wait (the dynamic loop serialised N round trips)
    const float a = (act && j < N) ? trans[(size_t)lane * N + j] : NEG;
    rowmax = fmaxf(rowmax, a);
  }
  if (!act) rowmax = 0.f;
  float EAT[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    float rm = readlane(rowmax, i);
    EAT[i] = (act && i < N) ? __expf(trans[(size_t)i * N + lane] - rm) : 0.f;
  }

  const float* __restrict__ ahb = ws.ahat + (size_t)b * T * N;
  const float* __restrict__ lsb = ws.logs + (size_t)b * T * N;
  float* __restrict__ rbw = ws.r + (size_t)b * T * N;
  float* __restrict__ dxb = inputGrad + (size_t)b * T * N;
  const float g = ws.scale[b] * grad[b];

  // d loss / d alpha_{T-1} = softmax(ahat_{T-1})
  float e = act ? __expf(ahb[(size_t)(T - 1) * N + lane]) : 0.f;
  float da = e / wave_sum(e);

  float lc[kChunk], an[kChunk], ln[kChunk], ac[kChunk];
  // chunk c covers steps t = thi - u, u = 0..kChunk-1; needs logs[t], ahat[t-1]
#pragma unroll
  for (int u = 0; u < kChunk; ++u) {
    int t = T - 1 - u;
    lc[u] = (act && t >= 1) ? lsb[(size_t)t * N + lane] : 0.f;
    ac[u] = (act && t >= 1) ? ahb[(size_t)(t - 1) * N + lane] : NEG;
  }
  for (int thi = T - 1; thi >= 1; thi -= kChunk) {
#pragma unroll
    for (int u = 0; u < kChunk; ++u) {
      int t = thi - kChunk - u;
      ln[u] = (act && t >= 1) ? lsb[(size_t)t * N + lane] : 0.f;
      an[u] = (act && t >= 1) ? ahb[(size_t)(t - 1) * N + lane] : NEG;
    }
    float dxs[kChunk], rs[kChunk];  // this chunk's rows of g * dalpha and r: stored after the chunk
#pragma unroll
    for (int u = 0; u < kChunk; ++u) {
      const int t = thi - u;
      dxs[u] = 0.f; rs[u] = 0.f;
      if (t >= 1) {  // wave-uniform
        float r = act ? da * __expf(-lc[u]) : 0.f;  // da_t[i] / s_t[i]
        float ep = act ? __expf(ac[u]) : 0.f;       // e_{t-1}[j]
        dxs[u] = g * da;
        rs[u] = r;                         // r_t for fcc_dtrans_small
        // only the matrix-vector product is on the serial chain (two packed chains, SGPR-pair operands); the
        // transition gradient sum_t r_t[i] e_{t-1}[j] has no dependence between time steps and is accumulated by a
        // separate, fully parallel kernel (it cost the scan 32 more broadcasts and 16 more packed FMAs per step)
        typedef float f32x2_t __attribute__((ext_vector_type(2)));
        f32x2_t n0 = {0.f, 0.f}, n1 = {0.f, 0.f};
#pragma unroll
        for (int k = 0; k < NP; k += 4) {
          const f32x2_t r01 = {readlane(r, k), readlane(r, k + 1)}, r23 = {readlane(r, k + 2), readlane(r, k + 3)};
          const f32x2_t t01 = {EAT[k], EAT[k + 1]}, t23 = {EAT[k + 2], EAT[k + 3]};
          n0 = __builtin_elementwise_fma(t01, r01, n0);
          n1 = __builtin_elementwise_fma(t23, r23, n1);
        }
        da = ep * ((n0.x + n0.y) + (n1.x + n1.y));
      }
    }
#pragma unroll
    for (int u = 0; u < kChunk; ++u) {
      const int t = thi - u;
      if (act && t >= 1) {
        dxb[(size_t)t * N + lane] = dxs[u];
        rbw[(size_t)t * N + lane] = rs[u];
      }
    }
#pragma unroll
    for (int u = 0; u < kChunk; ++u) { asm volatile("" : "+v"(ln[u]), "+v"(an[u])); }  // consume the prefetched chunk here
#pragma unroll
    for (int u = 0; u < kChunk; ++u) { lc[u] = ln[u]; ac[u] = an[u]; }
  }
  if (act) dxb[lane] = g * da;
}

// ---------------------------------------------------------------- the log-domain pair behind the N <= 31 scans (round 5)
// The utterances the linear-domain scans flag (transition rows more than 30 nats wide) used to re-run on fcc_*_small above,
// which are scaled-exp recursions too -- exp(A - rowmax) and exp(alpha - c_t) as separate fp32 factors, a floor of 1e-37 under
// the sums -- and clamp a state that is more than ~87 nats behind: with emissions AND transitions tens of nats wide the posterior
// moved to another path (loss within 1e-4, gradients off by O(1): profiles/r05_run32_criterion_fuzz.log).  These two evaluate
// every term as ONE exponential of a sum that is <= 0 by construction, as the reference's recursion does:
//   L_t[i]  = log sum_j exp(ahat_{t-1}[j] + A[i][j])  (max over j first),      alpha_t[i] = x_t[i] + L_t[i],  ahat = alpha - c_t
//   w_t[i][j] = exp(ahat_{t-1}[j] + A[i][j] - L_t[i]) in (0, 1],   dalpha_{t-1}[j] = sum_i dalpha_t[i] w_t[i][j],
//   dA[i][j] = sum_t dalpha_t[i] w_t[i][j]   (accumulated by the same loop: no r_t hand-over to the fcc_dtrans kernels).
// One wave per flagged utterance, lane = state, N <= NP (32 or 64); `ahat` and `logs` (= L_t) keep the workspace meaning of the
// other kernels.  NP = 64 (round 6): the fallback of the 32-64-label scans fcc_*_small<64>, which flag what they cannot hold.
template <int NP, bool MITM = false>
__device__ __forceinline__ void fcc_fwd_log_body(int T, int N, int scaleMode, const float* __restrict__ x,
                                                 const int* __restrict__ targetSize, const float* __restrict__ trans,
                                                 float* loss, const FccWs& ws) {
  const int b = blockIdx.x, lane = threadIdx.x;
  if (!ws.redo[b]) {
    if (MITM) fcc_mitm_finish(b, lane, T, N, scaleMode, targetSize, loss, ws);   // loss of the two linear-domain halves
    return;
  }
  const bool act = lane < N;
  const float NEG = -INFINITY;
  float Ar[NP];   // row `lane` of the transitions
#pragma unroll
  for (int j = 0; j < NP; ++j) Ar[j] = (act && j < N) ? trans[(size_t)lane * N + j] : NEG;
  const float* xb = x + (size_t)b * T * N;
  float* ahb = ws.ahat + (size_t)b * T * N;
  float* lsb = ws.logs + (size_t)b * T * N;
  float ah = 0.f;
  double C = 0.0;
  // frames in chunks of kChunk as in fcc_fwd_small: the next chunk's emissions are loaded at the top of a chunk and consumed at its
  // end, so that no frame waits on the memory counter (which also holds the frames' stores)
  float xc[kChunk], xn[kChunk];
#pragma unroll
  for (int u = 0; u < kChunk; ++u) xc[u] = (act && u < T) ? xb[(size_t)u * N + lane] : 0.f;
  for (int t0 = 0; t0 < T; t0 += kChunk) {
#pragma unroll
    for (int u = 0; u < kChunk; ++u) {
      const int tn = t0 + kChunk + u;
      xn[u] = (act && tn < T) ? xb[(size_t)tn * N + lane] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < kChunk; ++u) {
      const int t = t0 + u;
      if (t < T) {   // wave-uniform
        float a, L = 0.f;
        if (t == 0) {
          a = act ? xc[u] : NEG;
        } else {
          float v[NP];
          float m = NEG;
#pragma unroll
          for (int j = 0; j < NP; ++j) { v[j] = readlane(ah, j) + Ar[j]; m = fmaxf(m, v[j]); }   // (-inf for j >= N and for dead states)
          float sum = 0.f;
#pragma unroll
          for (int j = 0; j < NP; ++j) sum += fast_expf(v[j] - m);   // every term <= 1, the maximum's is 1
          L = m + fast_logf(sum);
          a = act && m > NEG ? xc[u] + L : NEG;
        }
        const float c = wave_max_rows<(NP > 32 ? 4 : 2)>(a);
        ah = a - c;
        C += (double)c;
        if (act) {
          ahb[(size_t)t * N + lane] = ah;
          lsb[(size_t)t * N + lane] = L;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < kChunk; ++u) asm volatile("" : "+v"(xn[u]));
#pragma unroll
    for (int u = 0; u < kChunk; ++u) xc[u] = xn[u];
  }
  const float e = act ? __expf(ah) : 0.f;
  const float tot = wave_sum(e);
  const float sc = scale_of(scaleMode, T, targetSize[b]);
  if (lane == 0) {
    loss[b] = (float)((double)sc * (C + (double)__logf(tot)));
    ws.scale[b] = sc;
  }
}
template <int NP, bool MITM = false>
__global__ __launch_bounds__(64) void fcc_fwd_log(int T, int N, int scaleMode, const float* __restrict__ x,
                                                  const int* __restrict__ targetSize, const float* __restrict__ trans,
                                                  float* __restrict__ loss, FccWs ws) {
  fcc_fwd_log_body<NP, MITM>(T, N, scaleMode, x, targetSize, trans, loss, ws);
}

template <int NP>
__global__ __launch_bounds__(64) void fcc_bwd_log(int T, int N, const float* __restrict__ trans, const float* __restrict__ grad,
                                                  float* __restrict__ inputGrad, FccWs ws) {
  const int b = blockIdx.x, lane = threadIdx.x;
  if (!ws.redo[b]) return;
  const bool act = lane < N;
  const float NEG = -INFINITY;
  float Ac[NP];   // column `lane` of the transitions: Ac[i] = A[i][lane]
#pragma unroll
  for (int i = 0; i < NP; ++i) Ac[i] = (act && i < N) ? trans[(size_t)i * N + lane] : NEG;
  const float* ahb = ws.ahat + (size_t)b * T * N;
  const float* lsb = ws.logs + (size_t)b * T * N;
  float* dxb = inputGrad + (size_t)b * T * N;
  const float g = ws.scale[b] * grad[b];
  float acc[NP];   // acc[i] = sum_t dalpha_t[i] w_t[i][lane]
#pragma unroll
  for (int i = 0; i < NP; ++i) acc[i] = 0.f;
  const float e = act ? __expf(ahb[(size_t)(T - 1) * N + lane]) : 0.f;
  float da = e / wave_sum(e);   // d loss / d alpha_{T-1} = softmax(ahat_{T-1})
  const int li = act ? lane : 0;
  // chunks of kChunk steps from the top frame (as fcc_bwd_small): step t needs L_t (lane i) and ahat_{t-1} (lane j)
  float lc[kChunk], ac[kChunk], ln[kChunk], an[kChunk];
#pragma unroll
  for (int u = 0; u < kChunk; ++u) {
    const int t = T - 1 - u;
    lc[u] = lsb[(size_t)max(t, 0) * N + li];
    ac[u] = ahb[(size_t)max(t - 1, 0) * N + li];
  }
  for (int thi = T - 1; thi >= 1; thi -= kChunk) {
#pragma unroll
    for (int u = 0; u < kChunk; ++u) {
      const int t = thi - kChunk - u;
      ln[u] = lsb[(size_t)max(t, 0) * N + li];
      an[u] = ahb[(size_t)max(t - 1, 0) * N + li];
    }
    float dxs[kChunk];
#pragma unroll
    for (int u = 0; u < kChunk; ++u) {
      const int t = thi - u;
      dxs[u] = 0.f;
      if (t >= 1) {   // wave-uniform
        dxs[u] = g * da;
        const float apj = act ? ac[u] : NEG;
        float nd = 0.f;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
          const float di = readlane(da, i);
          const float w = fast_expf(apj + Ac[i] - readlane(lc[u], i));   // w_t[i][lane]; exp(-inf) = 0 for i >= N, dead states
          const float dw = di > 0.f ? di * w : 0.f;                      // (0 x anything: a state without posterior mass hands nothing on)
          acc[i] += dw;
          nd += dw;
        }
        da = nd;
      }
    }
#pragma unroll
    for (int u = 0; u < kChunk; ++u) {
      const int t = thi - u;
      if (act && t >= 1) dxb[(size_t)t * N + lane] = dxs[u];
    }
#pragma unroll
    for (int u = 0; u < kChunk; ++u) asm volatile("" : "+v"(ln[u]), "+v"(an[u]));
#pragma unroll
    for (int u = 0; u < kChunk; ++u) { lc[u] = ln[u]; ac[u] = an[u]; }
  }
  if (act) dxb[lane] = g * da;
  // the transition-gradient partial of this utterance: time chunk 0 (the fcc_dtrans kernels zero the other chunks of a flagged utterance)
  float* tg = ws.tgpart + (size_t)b * kDtChunks * N * N;
#pragma unroll
  for (int i = 0; i < NP; ++i)
    if (act && i < N) tg[(size_t)i * N + lane] = g * acc[i];
}

// transition gradient of one (utterance, time chunk): part[i][j] = g * EA[i][j] * sum_{t in chunk} r_t[i] e_{t-1}[j]
// lane = j; r_t (left in the log-s slots by the scan) is broadcast per i.  No dependence between steps.
// LIN: the workspace of the DPP scans -- `ahat` holds u_t itself (scaled linear domain), r_t = b_t q_t.
template <int NP, bool LIN = false>
__global__ __launch_bounds__(64) void fcc_dtrans_small(int T, int N, const float* __restrict__ trans,
                                                       const float* __restrict__ grad, FccWs ws) {
  const int b = blockIdx.x, c = blockIdx.y;
  const bool lin = LIN;
  const int lane = threadIdx.x;
  const bool act = lane < N;
  if (ws.redo[b]) {   // a flagged utterance ran on fcc_fwd_log / fcc_bwd_log: its whole transition gradient is in chunk 0
    if (c > 0 && act) {
      float* tz = ws.tgpart + ((size_t)b * kDtChunks + c) * N * N;
      for (int i = 0; i < N; ++i) tz[(size_t)i * N + lane] = 0.f;
    }
    return;
  }
  const float* ahb = ws.ahat + (size_t)b * T * N;
  const float* rb = ws.r + (size_t)b * T * N;
  const int per = (T - 1 + kDtChunks - 1) / kDtChunks;
  const int t0 = 1 + c * per;
  int t1 = t0 + per;
  if (t1 > T) t1 = T;
  float acc[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) acc[i] = 0.f;
  for (int t = t0; t < t1; ++t) {
    const float r = act ? rb[(size_t)t * N + lane] : 0.f;
    const float ev = act ? ahb[(size_t)(t - 1) * N + lane] : 0.f;
    const float e = lin ? ev : (act ? __expf(ev) : 0.f);
#pragma unroll
    for (int i = 0; i < NP; ++i) acc[i] = fmaf(readlane(r, i), e, acc[i]);
  }
  const float g = ws.scale[b] * grad[b];
  float* tg = ws.tgpart + ((size_t)b * kDtChunks + c) * N * N;
  // rowmax_i (the scan's exp(A - rowmax) normalisation) in lane i
  float rowmax = -INFINITY;
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    const float a = (act && j < N) ? trans[(size_t)lane * N + j] : -INFINITY;
    rowmax = fmaxf(rowmax, a);
  }
  if (!act) rowmax = 0.f;
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const float rm = readlane(rowmax, i);
    if (act && i < N) tg[(size_t)i * N + lane] = g * __expf(trans[(size_t)i * N + lane] - rm) * acc[i];
  }
}

// The same partial for the N <= 31 path on the matrix pipe: sum_t r_t[i] e_{t-1}[j] is a [N x frames] . [frames x N] product -- one
// v_mfma_f32_32x32x2_f32 per two frames (A: lane (i, k) = r_{t+k}[i], B: lane (k, j) = e_{t+k-1}[j]) instead of 30 broadcasts + 30
// FMAs per frame: 39 -> ~8 us behind the backward scan at B = 64, T = 2000 (it is on the criterion's critical path).
__global__ __launch_bounds__(64) void fcc_dtrans_mfma(int T, int N, const float* __restrict__ trans, const float* __restrict__ grad, FccWs ws) {
  typedef float f32x16_t __attribute__((ext_vector_type(16)));
  const int b = blockIdx.x, c = blockIdx.y;
  const bool lin = true;
  const int lane = threadIdx.x, li = lane & 31, lh = lane >> 5;
  const bool act = li < N;
  if (ws.redo[b]) {   // a flagged utterance ran on fcc_fwd_log / fcc_bwd_log: its whole transition gradient is in chunk 0
    if (c > 0 && lane < N) {
      float* tz = ws.tgpart + ((size_t)b * kDtChunks + c) * N * N;
      for (int i = 0; i < N; ++i) tz[(size_t)i * N + lane] = 0.f;
    }
    return;
  }
  const float* ahb = ws.ahat + (size_t)b * T * N;
  const float* rb = ws.r + (size_t)b * T * N;
  const int per = (T - 1 + kDtChunks - 1) / kDtChunks;
  const int t0 = 1 + c * per;
  int t1 = t0 + per;
  if (t1 > T) t1 = T;
  f32x16_t acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int t = t0; t < t1; t += 8) {   // four frame pairs per round: eight loads in flight
    float rv[4], ev[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int tt = t + 2 * u + lh;
      const bool ok = act && tt < t1;
      const int tc = tt < t1 ? tt : t0;
      rv[u] = rb[(size_t)tc * N + (act ? li : 0)];
      ev[u] = ahb[(size_t)(tc - 1) * N + (act ? li : 0)];
      if (!ok) rv[u] = 0.f;
      if (!lin) ev[u] = __expf(ev[u]);
      if (!ok) ev[u] = 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(rv[u], ev[u], acc, 0, 0, 0);
  }
  const float g = ws.scale[b] * grad[b];
  float* tg = ws.tgpart + ((size_t)b * kDtChunks + c) * N * N;
  // rowmax_i (the scan's exp(A - rowmax) normalisation) in lane i; a lane of the C layout holds column j = li of rows (r & 3) + 8 (r >> 2) + 4 lh
  float rowmax = -INFINITY;
  {
    float tv[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) tv[j] = (lane < N && j < N) ? trans[(size_t)lane * N + j] : -INFINITY;   // all loads issued at once
#pragma unroll
    for (int j = 0; j < 32; ++j) rowmax = fmaxf(rowmax, tv[j]);
    if (lane >= N) rowmax = 0.f;
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i0 = (r & 3) + 8 * (r >> 2);
    const float rm = lh ? readlane(rowmax, i0 + 4) : readlane(rowmax, i0);
    const int i = i0 + 4 * lh;
    if (act && i < N) tg[(size_t)i * N + li] = g * __expf(trans[(size_t)i * N + li] - rm) * acc[r];
  }
}

// out[k] = sum_b part[b * stride][k]  (deterministic order)
__global__ void reduce_over_b(int B, size_t n, const float* __restrict__ part, float* __restrict__ out, int stride = 1) {
  size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int b = 0;
  for (; b + 7 < B; b += 8) {   // eight loads in flight (a handful of blocks: the kernel is one load latency per iteration)
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(b + u) * stride * n + k];
    s0 += v[0]; s1 += v[1]; s2 += v[2]; s3 += v[3];
    s0 += v[4]; s1 += v[5]; s2 += v[6]; s3 += v[7];
  }
  for (; b < B; ++b) s0 += part[(size_t)b * stride * n + k];
  out[k] = (s0 + s1) + (s2 + s3);
}

// part[b][0][k] = sum_c part[b][c][k]  (time chunks of one utterance, fixed order; grid.y = b)
__global__ void reduce_chunks(int C, size_t n, float* __restrict__ part) {
  size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  float* p = part + (size_t)blockIdx.y * C * n;
  float s0 = 0.f, s1 = 0.f;
  int c = 0;
  for (; c + 7 < C; c += 8) {   // eight loads in flight
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(c + u) * n + k];
    s0 += v[0]; s1 += v[1]; s0 += v[2]; s1 += v[3]; s0 += v[4]; s1 += v[5]; s0 += v[6]; s1 += v[7];
  }
  for (; c < C; ++c) s0 += p[(size_t)c * n + k];
  p[k] = s0 + s1;
}

// ---------------------------------------------------------------- Viterbi
struct VitWs {
  unsigned char* psi;  // [B][T][N]
};

constexpr int kBtChunk = 256;  // backtrace chunk (time steps staged in LDS)

template <int NP>
__global__ __launch_bounds__(64) void viterbi_small(int T, int N, const float* __restrict__ x,
                                                    const float* __restrict__ trans,
                                                    int* __restrict__ path, unsigned char* psiAll) {
  __shared__ unsigned char sPsi[kBtChunk * 64];
  __shared__ int sPath[kBtChunk];
  const int b = blockIdx.x;
  const int lane = threadIdx.x;
  const bool act = lane < N;
  const float NEG = -INFINITY;

  float A[NP];
#pragma unroll
  for (int j = 0; j < NP; ++j) A[j] = (act && j < N) ? trans[(size_t)lane * N + j] : NEG;

  const float* xb = x + (size_t)b * T * N;
  unsigned char* psi = psiAll + (size_t)b * T * N;

  float xc[kChunk], xn[kChunk];
#pragma unroll
  for (int u = 0; u < kChunk; ++u) xc[u] = (act && u < T) ? xb[(size_t)u * N + lane] : 0.f;
  float delta = NEG;
  for (int t0 = 0; t0 < T; t0 += kChunk) {
#pragma unroll
    for (int u = 0; u < kChunk; ++u) {
      int tn = t0 + kChunk + u;
      xn[u] = (act && tn < T) ? xb[(size_t)tn * N + lane] : 0.f;
    }
    int args[kChunk];
#pragma unroll
    for (int u = 0; u < kChunk; ++u) {
      const int t = t0 + u;
      args[u] = 0;
      if (t < T) {
        if (t == 0) {
          delta = act ? xc[u] : NEG;
        } else {
          // strict '>' scan over j upward: first maximum wins (oracle order).  Four independent scans over quarter
          // ranges, combined in range order with the same strict '>' (an earlier range keeps ties): the same
          // argmax as one 31-link compare/select chain at a quarter of its dependency depth.
          constexpr int Q = NP / 4;
          float bq[4];
          int aq[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) { bq[q] = readlane(delta, q * Q) + A[q * Q]; aq[q] = q * Q; }
#pragma unroll
          for (int jj = 1; jj < Q; ++jj) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int j = q * Q + jj;
              const float v = readlane(delta, j) + A[j];
              const bool gt = v > bq[q];
              bq[q] = gt ? v : bq[q];
              aq[q] = gt ? j : aq[q];
            }
          }
          float best = bq[0];
          int arg = aq[0];
#pragma unroll
          for (int q = 1; q < 4; ++q) {
            const bool gt = bq[q] > best;
            best = gt ? bq[q] : best;
            arg = gt ? aq[q] : arg;
          }
          delta = act ? best + xc[u] : NEG;
          args[u] = arg;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < kChunk; ++u) {
      const int t = t0 + u;
      if (act && t >= 1 && t < T) psi[(size_t)t * N + lane] = (unsigned char)args[u];
    }
#pragma unroll
    for (int u = 0; u < kChunk; ++u) asm volatile("" : "+v"(xn[u]));  // consume the prefetched chunk here
#pragma unroll
    for (int u = 0; u < kChunk; ++u) xc[u] = xn[u];
  }
  // final state: first argmax over i
  float m = wave_max(delta);
  unsigned long long eq = __ballot(act && delta == m);
  int cur = __ffsll((long long)eq) - 1;
  __syncthreads();  // drains this wave's psi stores before they are re-read
  int* pb = path + (size_t)b * T;
  // backtrace, chunks of kBtChunk steps staged through LDS
  for (int thi = T - 1; thi >= 0; thi -= kBtChunk) {
    int tlo = thi - kBtChunk + 1;
    if (tlo < 0) tlo = 0;
    int nsteps = thi - tlo + 1;
    for (int k = lane; k < nsteps * N; k += 64) sPsi[k] = psi[(size_t)tlo * N + k];
    __syncthreads();
    if (lane == 0) {
      for (int t = thi; t >= tlo; --t) {
        sPath[t - tlo] = cur;
        if (t >= 1) cur = sPsi[(t - tlo) * N + cur];
      }
    }
    cur = __builtin_amdgcn_readfirstlane(cur);
    __syncthreads();
    for (int k = lane; k < nsteps; k += 64) pb[tlo + k] = sPath[k];
    __syncthreads();
  }
}

// large-N path (criterion_fcc_big.hip)
bool fcc_big_supported(int B, int T, int N);
size_t fcc_big_workspace_size(int B, int T, int N);
int fcc_big_forward(int B, int T, int N, int scaleMode, const float* input, const int* targetSize,
                    const float* trans, float* loss, void* workspace, hipStream_t s);
int fcc_big_backward(int B, int T, int N, const float* trans, const float* grad, float* inputGrad,
                     float* transGrad, void* workspace, hipStream_t s);
bool viterbi_big_supported(int B, int T, int N);
size_t viterbi_big_workspace_size(int B, int T, int N);
int viterbi_big_compute(int B, int T, int N, const float* input, const float* trans, int* path,
                        void* workspace, hipStream_t s);
const int* fcc_big_range_flags(int B, int T, int N, const void* workspace);

}  // namespace w2l

using namespace w2l;


W2L_API size_t w2l_fcc_workspace_size(int B, int T, int N) {
  if (B <= 0 || T <= 0 || N <= 0) return 0;
  if (N > 64) return fcc_big_supported(B, T, N) ? fcc_big_workspace_size(B, T, N) : 0;
  size_t btn = align_up((size_t)B * T * N * sizeof(float), 256);
  size_t sz = 3 * btn + align_up((size_t)B * sizeof(float), 256) +
              align_up((size_t)B * kDtChunks * N * N * sizeof(float), 256) + align_up((size_t)B * sizeof(int), 256);
  if (N <= 31)   // the meet-in-the-middle scans (criterion_asg_mitm.hpp): r2, half, bm, ginv
    sz += btn + align_up((size_t)B * 2 * sizeof(double), 256) + align_up((size_t)B * 32 * sizeof(float), 256) + align_up((size_t)B * sizeof(float), 256);
  return sz;
}

W2L_API int w2l_fcc_forward(int B, int T, int N, int scaleMode, const float* input,
                            const int* targetSize, const float* trans, float* loss,
                            void* workspace, w2l_stream_t stream) {
  if (B <= 0 || T <= 0 || N <= 0 || !input || !targetSize || !trans || !loss || !workspace)
    return W2L_EINVAL;
  if (N > 64) {
    if (!fcc_big_supported(B, T, N)) return W2L_EUNSUPPORTED;
    return fcc_big_forward(B, T, N, scaleMode, input, targetSize, trans, loss, workspace, (hipStream_t)stream);
  }
  hipStream_t s = (hipStream_t)stream;
  FccWs ws = fcc_ws(workspace, B, T, N);
  if (asg_dpp_path(N) && asg_mitm_path()) {
    // product: alpha over frames 0 .. m and beta over T-1 .. m in two workgroups per utterance; the loss from the middle frame
    // (and the log-domain recursion for the utterances the range check flagged) by the second launch
    hipLaunchKernelGGL(fcc_mitm_fwd, dim3(B, mitm_only() < 0 ? 2 : 1), dim3(128), mitm_excl(B, (const void*)fcc_mitm_fwd), s, T, N, input, trans, ws, mitm_only() < 0 ? 0 : mitm_only());
    W2L_LAUNCH_CHECK();
    hipLaunchKernelGGL((fcc_fwd_log<32, true>), dim3(B), dim3(64), 0, s, T, N, scaleMode, input, targetSize, trans, loss, ws);
  } else if (asg_dpp_path(N)) {
    static const bool oneWave = tune_env("W2L_FCC_1WAVE") != nullptr;   // probe: the one-wave scan (A/B)
    if (oneWave) hipLaunchKernelGGL(fcc_fwd_dpp, dim3(B), dim3(64), 0, s, T, N, scaleMode, input, targetSize, trans, loss, ws);
    else hipLaunchKernelGGL(fcc_fwd_dpp2, dim3(B), dim3(128), 0, s, T, N, scaleMode, input, targetSize, trans, loss, ws);
    W2L_LAUNCH_CHECK();
    // the log-domain kernel for the utterances fcc_fwd_dpp flagged (returns at once for the others)
    hipLaunchKernelGGL(fcc_fwd_log<32>, dim3(B), dim3(64), 0, s, T, N, scaleMode, input, targetSize, trans, loss, ws);
  } else if (N <= 32) {
    // the scaled-exp scan flags the utterances whose sums left its range (kFccMinSum); the log-domain kernel recomputes those
    hipLaunchKernelGGL(fcc_fwd_small<32>, dim3(B), dim3(64), 0, s, T, N, scaleMode, input, targetSize, trans, loss, ws);
    W2L_LAUNCH_CHECK();
    hipLaunchKernelGGL(fcc_fwd_log<32>, dim3(B), dim3(64), 0, s, T, N, scaleMode, input, targetSize, trans, loss, ws);
  } else {
    hipLaunchKernelGGL(fcc_fwd_small<64>, dim3(B), dim3(64), 0, s, T, N, scaleMode, input, targetSize, trans, loss, ws);
    W2L_LAUNCH_CHECK();
    hipLaunchKernelGGL(fcc_fwd_log<64>, dim3(B), dim3(64), 0, s, T, N, scaleMode, input, targetSize, trans, loss, ws);
  }
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}

W2L_API int w2l_fcc_backward(int B, int T, int N, const float* trans, const float* grad,
                             float* inputGrad, float* transGrad, void* workspace,
                             w2l_stream_t stream) {
  return w2l::fcc_backward_impl(B, T, N, trans, grad, inputGrad, transGrad, workspace, (hipStream_t)stream, false);
}

// partialsOnly (the ASG criterion's fused backward sequence, N <= 64): the per-utterance transition-gradient partials stay in the
// workspace ([B][kDtChunks][N][N], chunk 0 of every utterance reduced over its chunks) and the caller's combine launch sums them
// over the utterances in reduce_over_b's order (asg_bwd_combine_k) -- one launch fewer on this criterion's chain
int w2l::fcc_backward_impl(int B, int T, int N, const float* trans, const float* grad, float* inputGrad, float* transGrad,
                           void* workspace, hipStream_t stream, bool partialsOnly) {
  if (B <= 0 || T <= 0 || N <= 0 || !trans || !grad || !inputGrad || !transGrad || !workspace)
    return W2L_EINVAL;
  if (N > 64) {
    if (!fcc_big_supported(B, T, N)) return W2L_EUNSUPPORTED;
    return fcc_big_backward(B, T, N, trans, grad, inputGrad, transGrad, workspace, (hipStream_t)stream);
  }
  hipStream_t s = (hipStream_t)stream;
  FccWs ws = fcc_ws(workspace, B, T, N);
  const bool dpp = asg_dpp_path(N);
  if (dpp && asg_mitm_path()) {
    hipLaunchKernelGGL(fcc_mitm_bwd, dim3(B, mitm_only() < 0 ? 2 : 1), dim3(128), mitm_excl(B, (const void*)fcc_mitm_bwd), s, T, N, trans, grad, inputGrad, ws, mitm_only() < 0 ? 0 : mitm_only());
    W2L_LAUNCH_CHECK();
    hipLaunchKernelGGL(fcc_bwd_log<32>, dim3(B), dim3(64), 0, s, T, N, trans, grad, inputGrad, ws);
  } else if (dpp) {
    static const bool oneWave = tune_env("W2L_FCC_1WAVE") != nullptr;
    if (oneWave) hipLaunchKernelGGL(fcc_bwd_dpp, dim3(B), dim3(64), 0, s, T, N, trans, grad, inputGrad, ws);
    else hipLaunchKernelGGL(fcc_bwd_dpp2, dim3(B), dim3(128), 0, s, T, N, trans, grad, inputGrad, ws);
    W2L_LAUNCH_CHECK();
    hipLaunchKernelGGL(fcc_bwd_log<32>, dim3(B), dim3(64), 0, s, T, N, trans, grad, inputGrad, ws);
  } else if (N <= 32) {
    hipLaunchKernelGGL(fcc_bwd_small<32>, dim3(B), dim3(64), 0, s, T, N, trans, grad, inputGrad, ws);
    W2L_LAUNCH_CHECK();
    hipLaunchKernelGGL(fcc_bwd_log<32>, dim3(B), dim3(64), 0, s, T, N, trans, grad, inputGrad, ws);
  } else {
    hipLaunchKernelGGL(fcc_bwd_small<64>, dim3(B), dim3(64), 0, s, T, N, trans, grad, inputGrad, ws);
    W2L_LAUNCH_CHECK();
    hipLaunchKernelGGL(fcc_bwd_log<64>, dim3(B), dim3(64), 0, s, T, N, trans, grad, inputGrad, ws);
  }
  W2L_LAUNCH_CHECK();
  if (dpp && !tune_env("W2L_FCC_DTRANS_OLD"))
    hipLaunchKernelGGL(fcc_dtrans_mfma, dim3(B, kDtChunks), dim3(64), 0, s, T, N, trans, grad, ws);
  else if (dpp)
    hipLaunchKernelGGL((fcc_dtrans_small<32, true>), dim3(B, kDtChunks), dim3(64), 0, s, T, N, trans, grad, ws);
  else if (N <= 32)
    hipLaunchKernelGGL(fcc_dtrans_small<32>, dim3(B, kDtChunks), dim3(64), 0, s, T, N, trans, grad, ws);
  else
    hipLaunchKernelGGL(fcc_dtrans_small<64>, dim3(B, kDtChunks), dim3(64), 0, s, T, N, trans, grad, ws);
  W2L_LAUNCH_CHECK();
  size_t n = (size_t)N * N;
  hipLaunchKernelGGL(reduce_chunks, dim3((unsigned)((n + 255) / 256), (unsigned)B), dim3(256), 0, s, kDtChunks, n, ws.tgpart);
  W2L_LAUNCH_CHECK();
  if (partialsOnly) return W2L_OK;
  hipLaunchKernelGGL(reduce_over_b, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, B, n, ws.tgpart, transGrad, kDtChunks);
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}

const float* w2l::fcc_transgrad_partials(void* workspace, int B, int T, int N, int* stride) {
  if (N > 64) return nullptr;
  *stride = kDtChunks;
  return fcc_ws(workspace, B, T, N).tgpart;
}

W2L_API int w2l_fcc_range_flags(int B, int T, int N, const void* workspace, int* flags, w2l_stream_t stream) {
  if (B <= 0 || T <= 0 || N <= 0 || !workspace || !flags) return W2L_EINVAL;
  const int* src;
  if (N > 64) {
    if (!fcc_big_supported(B, T, N)) return W2L_EUNSUPPORTED;
    src = fcc_big_range_flags(B, T, N, workspace);
  } else {
    src = fcc_ws((void*)workspace, B, T, N).redo;
  }
  W2L_HIP_CHECK(hipMemcpyAsync(flags, src, (size_t)B * sizeof(int), hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return W2L_OK;
}

W2L_API size_t w2l_viterbi_workspace_size(int B, int T, int N) {
  if (B <= 0 || T <= 0 || N <= 0) return 0;
  if (N > 64) return viterbi_big_supported(B, T, N) ? viterbi_big_workspace_size(B, T, N) : 0;
  // N <= 31: the delta rows [B][T][N] fp32 + back-pointer bytes + composed chunk maps (vit_fwd_dpp -> vit_psi_k -> vit_walk_k);
  // else the back-pointer bytes of viterbi_small
  return N <= 31 ? vit_dpp_ws_bytes(B, T, N) : align_up((size_t)B * T * N, 256);
}

W2L_API int w2l_viterbi_compute(int B, int T, int N, const float* input, const float* trans,
                                int* path, void* workspace, w2l_stream_t stream) {
  if (B <= 0 || T <= 0 || N <= 0 || !input || !trans || !path || !workspace) return W2L_EINVAL;
  if (N > 64) {
    if (!viterbi_big_supported(B, T, N)) return W2L_EUNSUPPORTED;
    return viterbi_big_compute(B, T, N, input, trans, path, workspace, (hipStream_t)stream);
  }
  hipStream_t s = (hipStream_t)stream;
  if (asg_dpp_path(N)) {
    hipLaunchKernelGGL(vit_fwd_dpp, dim3(B), dim3(64), 0, s, T, N, input, trans, (float*)workspace);
    W2L_LAUNCH_CHECK();
    const VitBtWs bw = vit_bt_ws(workspace, B, T, N);
    const int nC = vit_chunks(T);
    if (nC > 0) {
      hipLaunchKernelGGL(vit_psi_k, dim3((unsigned)nC, (unsigned)B), dim3(64), 0, s, T, N, trans, (const float*)workspace, bw);
      W2L_LAUNCH_CHECK();
    }
    const int stage = (size_t)T * 32 + (size_t)nC * 36 + 64 <= (size_t)150 * 1024;   // the psi bytes of an utterance in LDS
    const size_t shm = (((stage ? (size_t)T * 32 : 0) + (size_t)nC * 32 + 15) & ~(size_t)15) + (size_t)(nC + 1) * sizeof(int);
    if (shm > 64 * 1024) W2L_HIP_CHECK(hipFuncSetAttribute((const void*)vit_walk_k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    hipLaunchKernelGGL(vit_walk_k, dim3(B), dim3(64), shm, s, T, N, stage, (const float*)workspace, bw, path);
  } else if (N <= 32)
    hipLaunchKernelGGL(viterbi_small<32>, dim3(B), dim3(64), 0, s, T, N, input, trans, path, (unsigned char*)workspace);
  else
    hipLaunchKernelGGL(viterbi_small<64>, dim3(B), dim3(64), 0, s, T, N, input, trans, path, (unsigned char*)workspace);
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}
