// criterion_fac.hip -- ForceAlignmentCriterion (forward / backward / viterbi), gfx950.
//
// Replaces fl::lib::cuda::ForceAlignmentCriterion<float> (un-vendored Flashlight;
// the reference's call sites: recipes/slimIPL/src/Train.cpp:408-410, :1675).
// Math: SURVEY.md App. B.1; CPU restatement: oracle/criterion_oracle.c.
//
// One WORKGROUP per utterance scans T in a single launch: up to 4 wavefronts, thread k owns
// target positions k, k + NT, ... (P = ceil(L / NT) <= 2 as a template parameter), so every
// per-frame global access (w1 / dalpha rows) is coalesced and the serial chain per frame is one
// position deep.  The hand-off alpha_{t-1}[i-1] goes through a double-buffered LDS row: ONE
// s_barrier per time step (a step is ~300 cycles instead of the ~3400 of the earlier
// one-wave / 8-positions-per-lane mapping: 3.2 ms -> 0.3 ms at T = 2000, L = 300).  alpha is carried in fp64 registers; the two-way
// log-sum-exp correction log(1 + exp(-|d|)) is evaluated in fp32 (|error| < 1e-7
// per step), which keeps ASG = FCC - FAC inside the 1e-4 parity bar without any
// per-step renormalisation.  The forward stores only the "stay" posterior
// w1[t][i] = exp(s_stay - lse) (fp32, [B][T][L]); backward is then exp-free:
//   dalpha_{t-1}[i] = dalpha_t[i] w1[t][i] + dalpha_t[i+1] (1 - w1[t][i+1])
// and leaves g * dalpha_t[i] in place of w1; the emission gradient (a scatter of dalpha rows
// by label) is then a separate, fully parallel kernel over (b, frame chunk) instead of an LDS
// atomic + barrier pair inside every serial step.
// Emission values x[t][y_i] are gathered straight from the coalesced [B][T][N]
// rows (L1/L2 resident: the row is 120 B at N = 30), prefetched kFacChunk steps ahead.
#include "common.hpp"
#include "criterion_asg_fused.hpp"
#include <cstring>
#include <cstdlib>

namespace w2l {

constexpr int kFacChunk = 16;  // frames per prefetch chunk: ONE vmcnt drain (loads AND the frames' stores) per chunk
constexpr int kFacMaxN = 2048;  // LDS row buffer for the input-gradient scatter

struct FacRec;   // (criterion_fac_lin.hpp) fp64 mantissa + integer exponent of one lattice position

struct FacWs {
  float* w1;     // [B][T][L]  soft-max weights of the forward scan
  float* dal;    // [B][T][L]  g * dalpha of the backward scan (a SEPARATE buffer: written in place of w1, the stores
                 //            alias the scan's prefetch loads and hipcc drains vmcnt(0) -- stores included -- every frame)
  float* scale;  // [B]
  float* tgpart; // [B][N][N] (only when it is small enough, else NULL -> atomics)
  unsigned char* bp;  // viterbi back pointers [B][T][L]
  double* crow;  // [B][T][32] label rows c_t[n] of fac_rows_k (N <= 32 only, else NULL)
  float* zmax;   // [B][T]     frame maxima (base 2) of the label scores
  float* zspr;   // [B][T]     frame maximum - frame minimum of the label scores (base 2): the range check of fac_fwd_plin
  FacRec* hm;    // [B][320]   meet in the middle (criterion_fac_mitm.hpp): h_m of the alpha half, per position
  FacRec* gm;    // [B][320]   ... g_m of the beta half
  float* gam;    // [B][320]   ... the middle frame's posterior gamma_m = h_m g_m / Z
  int* csr;      // [B][L + 68] N <= 64: the utterance's positions sorted by label + the label offsets (fac_csr_k, for fac_scatter_csr_k)
  int* redo;     // [B]  set by fac_fwd_lin / fac_fwd_plin: the utterance's dynamics exceed what the scaled linear domain holds
                 //      exactly -> fac_fwd_blk (log domain), launched behind it, recomputes that utterance
};

__host__ __device__ inline bool fac_use_partials(int B, int N) {
  return (size_t)B * N * N <= ((size_t)1 << 22);
}
// transition-gradient partials per utterance: the two halves of the meet-in-the-middle backward pass (N <= 32) accumulate into
// sets of their own -- two workgroups adding into one set would make the sum depend on their timing (run-to-run differences in
// the last bit: tests/test_gpu_fl_compat.py::test_train_binary_on_list_files holds training to bit-identical reruns)
__host__ __device__ inline int fac_partial_sets(int N) { return N <= 32 ? 2 : 1; }

__host__ __device__ inline FacWs fac_ws(void* ws, int B, int T, int N, int L) {
  FacWs w;
  char* p = (char*)ws;
  w.w1 = (float*)p; p += align_up((size_t)B * T * L * sizeof(float), 256);
  w.dal = (float*)p; p += align_up((size_t)B * T * L * sizeof(float), 256);
  w.scale = (float*)p; p += align_up((size_t)B * sizeof(float), 256);
  w.redo = (int*)p; p += align_up((size_t)B * sizeof(int), 256);
  w.bp = (unsigned char*)w.w1;  // viterbi reuses the w1 region (needs B*T*L bytes)
  w.tgpart = fac_use_partials(B, N) ? (float*)p : nullptr;
  if (w.tgpart) p += align_up((size_t)fac_partial_sets(N) * B * N * N * sizeof(float), 256);
  w.crow = nullptr; w.zmax = nullptr; w.zspr = nullptr; w.hm = nullptr; w.gm = nullptr; w.gam = nullptr; w.csr = nullptr;
  if (N <= 64) { w.csr = (int*)p; p += align_up((size_t)B * (L + 68) * sizeof(int), 256); }
  if (N <= 32) {
    w.crow = (double*)p; p += align_up((size_t)B * T * 32 * sizeof(double), 256);
    w.zmax = (float*)p; p += align_up((size_t)B * T * sizeof(float), 256);
    w.zspr = (float*)p; p += align_up((size_t)B * T * sizeof(float), 256);
    w.hm = (FacRec*)p; p += align_up((size_t)B * 320 * 16, 256);
    w.gm = (FacRec*)p; p += align_up((size_t)B * 320 * 16, 256);
    w.gam = (float*)p;
  }
  return w;
}

}  // namespace w2l

#include "criterion_fac_lin.hpp"   // N <= 32, L <= 320: scaled linear domain, one wave per utterance (fac_fwd_lin, fac_bwd_wave)
#include "criterion_fac_mitm.hpp"  // ... the product: the pipelined scans from both ends to the middle frame (fac_mitm_fwd, fac_mitm_bwd)

namespace w2l {

// the probe library can put the previous generation (fac_*_blk) back for A/B work (W2L_ASG_OLD=1)
inline bool fac_lin_path(int N, int L) {
  static const bool old = tune_env("W2L_ASG_OLD") != nullptr;
  return N <= 32 && L <= 320 && !old;
}
// product: the meet-in-the-middle pair; the probe library runs the round-4 / round-5 full-length scans under W2L_ASG_NOMITM=1 or any
// of the generation switches (W2L_FAC_GEN, W2L_FAC_BWD, W2L_FAC_BWD32)
inline int fac_mitm_only() {   // probe, timing only: one half of the meet-in-the-middle kernels alone (W2L_MITM_ONLY=0 / 1)
  static const int v = [] { const char* e = tune_env("W2L_MITM_ONLY"); return e ? (e[0] == '1' ? 1 : 0) : -1; }();
  return v;
}
inline int fac_mitm_abl() {   // probe, timing only: W2L_FAC_ABL bit 0 = no label-weight loads, bit 1 = no w1 stores in fac_mitm_fwd
  static const int v = [] { const char* e = tune_env("W2L_FAC_ABL"); return e ? atoi(e) : 0; }();
  return v;
}
inline bool fac_mitm_path() {
  static const bool off = tune_env("W2L_ASG_NOMITM") || tune_env("W2L_FAC_GEN") || tune_env("W2L_FAC_BWD") || tune_env("W2L_FAC_BWD32");
  return !off;
}

template <int P>
__global__ __launch_bounds__(64) void fac_fwd(int T, int N, int L, int scaleMode,
                                              const float* __restrict__ x,
                                              const int* __restrict__ target,
                                              const int* __restrict__ targetSize,
                                              const float* __restrict__ trans,
                                              float* __restrict__ loss, FacWs ws) {
  const int b = blockIdx.x;
  const int lane = threadIdx.x;
  const int S = targetSize[b];
  const float sc = scale_of(scaleMode, T, S);
  if (lane == 0) ws.scale[b] = sc;
  if (S <= 0) {
    if (lane == 0) loss[b] = 0.f;
    return;
  }
  const int* y = target + (size_t)b * L;
  const float* xb = x + (size_t)b * T * N;
  float* w1b = ws.w1 + (size_t)b * T * L;
  const double NEG = -INFINITY;

  int yi[P];
  float selfT[P], prevT[P];
  double alpha[P];
#pragma unroll
  for (int p = 0; p < P; ++p) {
    int i = lane * P + p;
    bool v = i < S;
    yi[p] = v ? y[i] : 0;
    int yp = (v && i > 0) ? y[i - 1] : 0;
    selfT[p] = v ? trans[(size_t)yi[p] * N + yi[p]] : 0.f;
    prevT[p] = (v && i > 0) ? trans[(size_t)yi[p] * N + yp] : 0.f;
    alpha[p] = NEG;
  }

  float xc[kFacChunk][P], xn[kFacChunk][P];
#pragma unroll
  for (int u = 0; u < kFacChunk; ++u)
#pragma unroll
    for (int p = 0; p < P; ++p)
      xc[u][p] = (u < T && lane * P + p < S) ? xb[(size_t)u * N + yi[p]] : 0.f;

  for (int t0 = 0; t0 < T; t0 += kFacChunk) {
#pragma unroll
    for (int u = 0; u < kFacChunk; ++u) {
      int tn = t0 + kFacChunk + u;
#pragma unroll
      for (int p = 0; p < P; ++p)
        xn[u][p] = (tn < T && lane * P + p < S) ? xb[(size_t)tn * N + yi[p]] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < kFacChunk; ++u) {
      const int t = t0 + u;
      if (t < T) {
        if (t == 0) {
          if (lane == 0) alpha[0] = (double)xc[u][0];
        } else {
          double carry = lane_shift_up_dpp(alpha[P - 1], NEG);  // alpha_{t-1}[lane*P - 1]
          double prevA = carry;
#pragma unroll
          for (int p = 0; p < P; ++p) {
            const int i = lane * P + p;
            double cur = alpha[p];
            double s1 = cur + (double)selfT[p];
            double s2 = prevA + (double)prevT[p];
            double m = fmax(s1, s2);
            double na = NEG;
            float w = 0.f;
            if (i < S && m != NEG) {
              float d = (float)(fmin(s1, s2) - m);   // <= 0, may be -inf
              float ed = __expf(d);
              float den = 1.f + ed;
              na = m + (double)fast_logf(den) + (double)xc[u][p];
              float inv = 1.f / den;
              w = (s1 >= s2) ? inv : ed * inv;
            }
            if (i < S) w1b[(size_t)t * L + i] = w;
            prevA = cur;
            alpha[p] = na;
          }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < kFacChunk; ++u)
#pragma unroll
      for (int p = 0; p < P; ++p) xc[u][p] = xn[u][p];
  }
  // loss = scale * alpha[T-1][S-1]
  const int il = S - 1;
#pragma unroll
  for (int p = 0; p < P; ++p)
    if (lane * P + p == il) loss[b] = (float)((double)sc * alpha[p]);
}

template <int P>
__global__ __launch_bounds__(64) void fac_bwd(int T, int N, int L, const int* __restrict__ target,
                                              const int* __restrict__ targetSize,
                                              const float* __restrict__ grad,
                                              float* __restrict__ inputGrad,
                                              float* __restrict__ transGrad, FacWs ws) {
  __shared__ float row[kFacMaxN];
  const int b = blockIdx.x;
  const int lane = threadIdx.x;
  const int S = targetSize[b];
  float* dxb = inputGrad + (size_t)b * T * N;
  if (S <= 0) {
    for (size_t k = lane; k < (size_t)T * N; k += 64) dxb[k] = 0.f;
    return;
  }
  const int* y = target + (size_t)b * L;
  const float* w1b = ws.w1 + (size_t)b * T * L;
  const float g = ws.scale[b] * grad[b];

  int yi[P], yp[P];
  float da[P], accS[P], accP[P];
#pragma unroll
  for (int p = 0; p < P; ++p) {
    int i = lane * P + p;
    bool v = i < S;
    yi[p] = v ? y[i] : 0;
    yp[p] = (v && i > 0) ? y[i - 1] : 0;
    da[p] = (i == S - 1) ? 1.f : 0.f;
    accS[p] = 0.f;
    accP[p] = 0.f;
  }
  for (int k = lane; k < N; k += 64) row[k] = 0.f;
  __syncthreads();

  float wc[kFacChunk][P], wn[kFacChunk][P];
#pragma unroll
  for (int u = 0; u < kFacChunk; ++u) {
    int t = T - 1 - u;
#pragma unroll
    for (int p = 0; p < P; ++p)
      wc[u][p] = (t >= 1 && lane * P + p < S) ? w1b[(size_t)t * L + lane * P + p] : 0.f;
  }
  for (int thi = T - 1; thi >= 0; thi -= kFacChunk) {
#pragma unroll
    for (int u = 0; u < kFacChunk; ++u) {
      int t = thi - kFacChunk - u;
#pragma unroll
      for (int p = 0; p < P; ++p)
        wn[u][p] = (t >= 1 && lane * P + p < S) ? w1b[(size_t)t * L + lane * P + p] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < kFacChunk; ++u) {
      const int t = thi - u;
      if (t >= 0) {
        // scatter-add dalpha_t[i] into the emission-gradient row of frame t
#pragma unroll
        for (int p = 0; p < P; ++p)
          if (lane * P + p < S && da[p] != 0.f) atomicAdd(&row[yi[p]], g * da[p]);
        __syncthreads();
        for (int k = lane; k < N; k += 64) {
          dxb[(size_t)t * N + k] = row[k];
          row[k] = 0.f;
        }
        __syncthreads();
        if (t >= 1) {
          // advance part handed to position i-1: da_t[i] * (1 - w1[t][i])
          float adv[P];
#pragma unroll
          for (int p = 0; p < P; ++p) {
            float w = wc[u][p];
            float st = da[p] * w;
            adv[p] = da[p] - st;
            accS[p] += st;
            accP[p] += adv[p];
            da[p] = st;
          }
          float fromNext = lane_shift_down(adv[0], 0.f);  // adv of position (lane+1)*P
#pragma unroll
          for (int p = 0; p < P; ++p) da[p] += (p + 1 < P) ? adv[p + 1 < P ? p + 1 : 0] : fromNext;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < kFacChunk; ++u)
#pragma unroll
      for (int p = 0; p < P; ++p) wc[u][p] = wn[u][p];
  }
  float* tg = ws.tgpart ? ws.tgpart + (size_t)b * N * N : transGrad;
#pragma unroll
  for (int p = 0; p < P; ++p) {
    int i = lane * P + p;
    if (i < S) {
      if (accS[p] != 0.f) atomicAdd(&tg[(size_t)yi[p] * N + yi[p]], g * accS[p]);
      if (i > 0 && accP[p] != 0.f) atomicAdd(&tg[(size_t)yi[p] * N + yp[p]], g * accP[p]);
    }
  }
}

// ---------------------------------------------------------------- workgroup-per-utterance kernels
template <int NW, int P>
__device__ __forceinline__ void fac_fwd_blk_body(int T, int N, int L, int scaleMode,
                                                 const float* __restrict__ x,
                                                 const int* __restrict__ target,
                                                 const int* __restrict__ targetSize,
                                                 const float* __restrict__ trans,
                                                 float* loss, const FacWs& ws, const int* __restrict__ redo = nullptr) {
  constexpr int NT = 64 * NW;
  __shared__ double sA[2][NT * P + 1];  // sA[buf][i + 1] = alpha[i]; sA[buf][0] = -inf (position -1)
  const int b = blockIdx.x, tid = threadIdx.x;
  if (redo && !redo[b]) return;   // launched behind fac_fwd_lin: only the utterances it flagged
  const int S = targetSize[b];
  const float sc = scale_of(scaleMode, T, S);
  if (tid == 0) ws.scale[b] = sc;
  if (S <= 0) {
    if (tid == 0) loss[b] = 0.f;
    return;
  }
  const int* y = target + (size_t)b * L;
  const float* xb = x + (size_t)b * T * N;
  float* w1b = ws.w1 + (size_t)b * T * L;
  const double NEG = -INFINITY;

  int yi[P];
  float selfT[P], prevT[P];
  double alpha[P];
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const int i = tid + NT * p;
    const bool v = i < S;
    yi[p] = v ? y[i] : 0;
    const int yp = (v && i > 0) ? y[i - 1] : 0;
    selfT[p] = v ? trans[(size_t)yi[p] * N + yi[p]] : 0.f;
    prevT[p] = (v && i > 0) ? trans[(size_t)yi[p] * N + yp] : 0.f;
    alpha[p] = NEG;
  }
  float xc[kFacChunk][P], xn[kFacChunk][P];
#pragma unroll
  for (int u = 0; u < kFacChunk; ++u)
#pragma unroll
    for (int p = 0; p < P; ++p) xc[u][p] = (u < T && tid + NT * p < S) ? xb[(size_t)u * N + yi[p]] : 0.f;

  if (tid == 0) { sA[0][0] = NEG; sA[1][0] = NEG; }
  for (int t0 = 0; t0 < T; t0 += kFacChunk) {
#pragma unroll
    for (int u = 0; u < kFacChunk; ++u) {
      const int tn = t0 + kFacChunk + u;
#pragma unroll
      for (int p = 0; p < P; ++p) xn[u][p] = (tn < T && tid + NT * p < S) ? xb[(size_t)tn * N + yi[p]] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < kFacChunk; ++u) {
      const int t = t0 + u;
      if (t < T) {  // uniform
        if (t == 0) {
          if (tid == 0) alpha[0] = (double)xc[u][0];
        } else {
          // all neighbour reads first, then branch-free arithmetic: the P positions of a lane are independent
          // dependency chains and interleave (the branchy form ran them one after the other, each behind its own
          // LDS round trip: ~500 cycles per position); 1/den through v_rcp_f32 (den in [1, 2])
          double prevA[P];
#pragma unroll
          for (int p = 0; p < P; ++p) prevA[p] = sA[(t - 1) & 1][tid + NT * p];  // alpha_{t-1}[i-1]
#pragma unroll
          for (int p = 0; p < P; ++p) {
            const int i = tid + NT * p;
            const double s1 = alpha[p] + (double)selfT[p];
            const double s2 = prevA[p] + (double)prevT[p];
            const double m = fmax(s1, s2);
            const bool live = i < S && m != NEG;
            const float d = (float)(fmin(s1, s2) - m);  // <= 0, -inf for a dead branch, NaN only when !live
            const float ed = fast_expf(d);
            const float den = 1.f + ed;
            const double na = m + (double)fast_logf(den) + (double)xc[u][p];
            const float inv = __builtin_amdgcn_rcpf(den);
            const float w = (s1 >= s2) ? inv : ed * inv;
            if (i < S) w1b[(size_t)t * L + i] = live ? w : 0.f;
            alpha[p] = live ? na : NEG;
          }
        }
#pragma unroll
        for (int p = 0; p < P; ++p) sA[t & 1][tid + NT * p + 1] = alpha[p];
        __syncthreads();
      }
    }
    // consume the whole prefetched chunk HERE: hipcc places `s_waitcnt vmcnt(0)` at the first use of a loaded register,
    // and on gfx9 that counter also holds the frames' global stores -- left to the first use, every frame of the next
    // chunk drained its predecessors' stores (one L2 round trip per frame)
#pragma unroll
    for (int u = 0; u < kFacChunk; ++u)
#pragma unroll
      for (int p = 0; p < P; ++p) asm volatile("" : "+v"(xn[u][p]));
#pragma unroll
    for (int u = 0; u < kFacChunk; ++u)
#pragma unroll
      for (int p = 0; p < P; ++p) xc[u][p] = xn[u][p];
  }
#pragma unroll
  for (int p = 0; p < P; ++p)
    if (tid + NT * p == S - 1) loss[b] = (float)((double)sc * alpha[p]);
}
template <int NW, int P>
__global__ __launch_bounds__(64 * NW) void fac_fwd_blk(int T, int N, int L, int scaleMode,
                                                       const float* __restrict__ x,
                                                       const int* __restrict__ target,
                                                       const int* __restrict__ targetSize,
                                                       const float* __restrict__ trans,
                                                       float* __restrict__ loss, FacWs ws, const int* __restrict__ redo = nullptr) {
  fac_fwd_blk_body<NW, P>(T, N, L, scaleMode, x, target, targetSize, trans, loss, ws, redo);
}

// The tail of the ASG criterion's fused forward sequence (fac_forward_asg below) in ONE launch: fac_mitm_finish (Z, loss, the middle
// frame's posterior, the range check), the log-domain recomputation of an utterance the check flagged (fac_fwd_blk<8, 1>: any L <= 512;
// a path no recipe input takes), and ASG's own subtraction  minuend[b] = minuend[b] - loss[b]  (minuend = FullConnectionCriterion's
// loss: the caller has joined that stream in front of this launch).  Two launches and an axpy fewer on the forward chain.
__global__ __launch_bounds__(kFacFinishThreads) void fac_mitm_finish_all(int T, int N, int L, int scaleMode, const float* __restrict__ x,
                                                                         const int* __restrict__ target, const int* __restrict__ targetSize,
                                                                         const float* __restrict__ trans, float* loss, FacWs ws,
                                                                         float* minuend) {
  static_assert(kFacFinishThreads == 512, "fac_fwd_blk<8, 1> is a 512-thread body");
  const int b = blockIdx.x;
  fac_mitm_finish_body(T, N, L, scaleMode, target, targetSize, trans, loss, ws);
  __syncthreads();   // the flag and the loss of thread 0, visible to the workgroup
  if (*(volatile int*)(ws.redo + b) != 0) {   // workgroup-uniform
    fac_fwd_blk_body<8, 1>(T, N, L, scaleMode, x, target, targetSize, trans, loss, ws, nullptr);
    __syncthreads();
  }
  if (minuend && threadIdx.x == 0) minuend[b] = minuend[b] - *(volatile float*)(loss + b);
}

// backward scan: consumes w1[t][i], leaves g * dalpha_t[i] in its place; transition gradients per position
template <int NW, int P>
__global__ __launch_bounds__(64 * NW) void fac_bwd_blk(int T, int N, int L, const int* __restrict__ target,
                                                       const int* __restrict__ targetSize,
                                                       const float* __restrict__ grad,
                                                       float* __restrict__ transGrad, FacWs ws) {
  constexpr int NT = 64 * NW;
  __shared__ float sAdv[2][NT * P + 1];  // sAdv[buf][i] = dalpha_t[i] * (1 - w1[t][i]); entry NT*P stays 0
  const int b = blockIdx.x, tid = threadIdx.x;
  const int S = targetSize[b];
  if (S <= 0) return;  // the scatter kernel zero-fills this utterance's gradient
  const int* y = target + (size_t)b * L;
  const float* __restrict__ w1b = ws.w1 + (size_t)b * T * L;
  float* __restrict__ dalb = ws.dal + (size_t)b * T * L;
  const float g = ws.scale[b] * grad[b];

  int yi[P], yp[P];
  float da[P], accS[P], accP[P];
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const int i = tid + NT * p;
    const bool v = i < S;
    yi[p] = v ? y[i] : 0;
    yp[p] = (v && i > 0) ? y[i - 1] : 0;
    da[p] = (i == S - 1) ? 1.f : 0.f;
    accS[p] = 0.f;
    accP[p] = 0.f;
  }
  if (tid == 0) { sAdv[0][NT * P] = 0.f; sAdv[1][NT * P] = 0.f; }

  float wc[kFacChunk][P], wn[kFacChunk][P];
#pragma unroll
  for (int u = 0; u < kFacChunk; ++u) {
    const int t = T - 1 - u;
#pragma unroll
    for (int p = 0; p < P; ++p) wc[u][p] = (t >= 1 && tid + NT * p < S) ? w1b[(size_t)t * L + tid + NT * p] : 0.f;
  }
  for (int thi = T - 1; thi >= 0; thi -= kFacChunk) {
#pragma unroll
    for (int u = 0; u < kFacChunk; ++u) {
      const int t = thi - kFacChunk - u;
#pragma unroll
      for (int p = 0; p < P; ++p) wn[u][p] = (t >= 1 && tid + NT * p < S) ? w1b[(size_t)t * L + tid + NT * p] : 0.f;
    }
    float dst[kFacChunk][P];  // this chunk's rows of g * dalpha: stored after the chunk (no store in flight inside it)
#pragma unroll
    for (int u = 0; u < kFacChunk; ++u) {
      const int t = thi - u;
#pragma unroll
      for (int p = 0; p < P; ++p) dst[u][p] = g * da[p];  // row t of g * dalpha (0 beyond S)
      if (t >= 0) {  // uniform
        if (t >= 1) {
          float st[P];
#pragma unroll
          for (int p = 0; p < P; ++p) {
            st[p] = da[p] * wc[u][p];
            const float adv = da[p] - st[p];
            accS[p] += st[p];
            accP[p] += adv;
            sAdv[t & 1][tid + NT * p] = adv;
          }
          // LDS-only barrier: __syncthreads() also drains vmcnt(0) here -- the frame's global stores and the prefetch
          // loads -- which made every frame wait for a store round trip (708 cycles per frame for five fp32 operations)
          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
          for (int p = 0; p < P; ++p) da[p] = st[p] + sAdv[t & 1][tid + NT * p + 1];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < kFacChunk; ++u) {
      const int t = thi - u;
#pragma unroll
      for (int p = 0; p < P; ++p)
        if (t >= 0 && tid + NT * p < L) dalb[(size_t)t * L + tid + NT * p] = dst[u][p];
    }
#pragma unroll
    for (int u = 0; u < kFacChunk; ++u)
#pragma unroll
      for (int p = 0; p < P; ++p) asm volatile("" : "+v"(wn[u][p]));  // one vmcnt drain per chunk (see fac_fwd_blk)
#pragma unroll
    for (int u = 0; u < kFacChunk; ++u)
#pragma unroll
      for (int p = 0; p < P; ++p) wc[u][p] = wn[u][p];
  }
  float* tg = ws.tgpart ? ws.tgpart + (size_t)b * N * N : transGrad;
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const int i = tid + NT * p;
    if (i < S) {
      if (accS[p] != 0.f) atomicAdd(&tg[(size_t)yi[p] * N + yi[p]], g * accS[p]);
      if (i > 0 && accP[p] != 0.f) atomicAdd(&tg[(size_t)yi[p] * N + yp[p]], g * accP[p]);
    }
  }
}

#ifdef W2L_PROBE  // measured not faster than fac_*_blk: kept for A/B work only
// ---------------------------------------------------------------- skewed wave pipeline (no workgroup barrier per frame)
// fac_*_blk above pay ONE s_barrier + LDS round trip per frame (~580-700 cycles per frame at 8 waves: the frame's own
// arithmetic is ~150).  But a position only needs its LEFT neighbour's alpha of the previous frame (forward) / its RIGHT
// neighbour's advance term of the same step (backward): inside a wave that is a DPP lane shift, and across waves it is
// ONE value per frame per wave boundary.  So the waves of an utterance run skewed instead of in lockstep: wave w-1 stays
// at least a frame ahead of wave w (forward; w+1 ahead of w in the backward scan) and hands its boundary value over
// through an LDS mailbox ring (value, then frame tag; the reader polls the tag -- in steady state the value has been
// there for a frame), with back-pressure every kMbRing/2 frames so that a leader never laps its follower.
// One position per lane (L <= 64 NW).  Same arithmetic, same order as fac_*_blk: bit-identical results.
// MEASURED (profiles/r02_run16_fac_pipeline_negative.log): correct, and NOT faster -- 517 us against 487 us forward at 5 waves.
// The premise was wrong: ONE wave with no neighbour at all (L = 60) already needs 459 cycles per frame -- the frame is a
// chain of ~16 dependent fp64 / transcendental operations -- and the barrier version adds only ~125 cycles to that at 5
// waves.  What the first versions of this kernel taught on the way: `volatile` LDS accesses make hipcc drain vmcnt(0)
// around each one (2150 cycles per frame), a store in flight costs the same in front of every poll loop (1100), and a
// `while` poll is unrolled 16x with a ~100-instruction exit cascade (740) -- hence relaxed workgroup atomics, stores
// after the chunk, `unroll(disable)`.  Kept in the probe library (W2L_FAC_PIPE=1); the product runs fac_*_blk.
constexpr int kMbRing = 16;
constexpr int kMbSpinMax = 1 << 22;   // a poll that never succeeds ends the kernel with a poisoned loss instead of hanging the GPU

// mailbox accesses: relaxed workgroup-scope atomics = plain ds_read / ds_write.  (`volatile` made hipcc drain vmcnt(0) -- the
// frame's global stores -- around every access: 2150 cycles per frame.)
template <class T> __device__ __forceinline__ T mb_load(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
template <class T> __device__ __forceinline__ void mb_store(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// poll *p until pred(value) (bounded; hipcc unrolls a `while` poll 16x with a ~100-instruction exit cascade: keep it a 4-instruction loop)
template <class Pred> __device__ __forceinline__ bool mb_poll(const int* p, Pred pred) {
  int spins = 0;
  int vv;
#pragma clang loop unroll(disable)
  do { vv = mb_load(p); } while (!pred(vv) && ++spins < kMbSpinMax);
  return pred(vv);
}
__device__ __forceinline__ int dpp_wave_shr1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xF, 0xF, false); }   // lane l <- lane l-1
__device__ __forceinline__ int dpp_wave_shl1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x130, 0xF, 0xF, false); }   // lane l <- lane l+1

template <int NW>
__global__ __launch_bounds__(64 * NW) void fac_fwd_pipe(int T, int N, int L, int scaleMode,
                                                        const float* __restrict__ x,
                                                        const int* __restrict__ target,
                                                        const int* __restrict__ targetSize,
                                                        const float* __restrict__ trans,
                                                        float* __restrict__ loss, FacWs ws) {
  __shared__ double mbVal[NW + 1][kMbRing];   // row NW: where the lanes that publish nothing write
  __shared__ int mbTag[NW + 1][kMbRing];
  __shared__ int prog[NW + 1];   // prog[w] = last frame wave w has finished (-1: none)
  __shared__ int bad;
  __shared__ double sink[64];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int S = targetSize[b];
  const float sc = scale_of(scaleMode, T, S);
  if (tid == 0) ws.scale[b] = sc;
  if (S <= 0) {
    if (tid == 0) loss[b] = 0.f;
    return;
  }
  for (int e = tid; e < (NW + 1) * kMbRing; e += 64 * NW) (&mbTag[0][0])[e] = -1;
  if (tid <= NW) prog[tid] = -1;
  if (tid == 0) bad = 0;
  __syncthreads();
  const int* y = target + (size_t)b * L;
  const float* xb = x + (size_t)b * T * N;
  float* w1b = ws.w1 + (size_t)b * T * L;
  const double NEG = -INFINITY;
  const int i = tid;
  const bool v = i < S;
  const int yi = v ? y[i] : 0;
  const int yp = (v && i > 0) ? y[i - 1] : 0;
  const float selfT = v ? trans[(size_t)yi * N + yi] : 0.f;
  const float prevT = (v && i > 0) ? trans[(size_t)yi * N + yp] : 0.f;
  double alpha = NEG;
  const int lastWave = (S - 1) >> 6;   // waves beyond the target's last position have nothing to do and nobody waits for them
  if (wave > lastWave) return;
  const bool feeds = wave < lastWave;  // somebody consumes this wave's boundary value
  const bool fed = wave > 0;
  const int src = fed ? wave - 1 : NW;
  // every lane runs the mailbox accesses (the reads are broadcasts, the writes of lanes != 63 go to a sink): no divergent
  // region in the frame; the value for frame t+1 is read one frame early -- in steady state the leader is frames ahead
  double* const pubVal = lane == 63 && feeds ? &mbVal[wave][0] : &sink[lane] - 0;
  int* const pubTag = lane == 63 && feeds ? &mbTag[wave][0] : &mbTag[NW][0];
  const int pubStep = lane == 63 && feeds ? 1 : 0;
  int pfTag = -2;
  double pfVal = NEG;

  float xc[kFacChunk], xn[kFacChunk];
#pragma unroll
  for (int u = 0; u < kFacChunk; ++u) xc[u] = (u < T && v) ? xb[(size_t)u * N + yi] : 0.f;

  for (int t0 = 0; t0 < T; t0 += kFacChunk) {
#pragma unroll
    for (int u = 0; u < kFacChunk; ++u) {
      const int tn = t0 + kFacChunk + u;
      xn[u] = (tn < T && v) ? xb[(size_t)tn * N + yi] : 0.f;
    }
    float wst[kFacChunk];   // this chunk's rows of w1: stored after the chunk -- a store in flight makes hipcc's vmcnt(0) in front
                            // of every poll loop wait for its round trip (the frame's arithmetic is 5x shorter)
#pragma unroll
    for (int u = 0; u < kFacChunk; ++u) {
      const int t = t0 + u;
      wst[u] = 0.f;
      if (t < T) {  // uniform
        if (t == 0) {
          if (tid == 0) alpha = (double)xc[u];
        } else {
          // alpha_{t-1}[i-1]: the lane below, or the previous wave's last lane through the mailbox
          const long long ab = __double_as_longlong(alpha);
          const int lo = dpp_wave_shr1((int)(ab & 0xffffffffll)), hi = dpp_wave_shr1((int)(ab >> 32));
          double prevA = __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
          if (fed) {  // uniform
            const int slot = (t - 1) & (kMbRing - 1);
            if (pfTag != t - 1) {   // uniform (a broadcast value): the early read came before the leader's write
              if (!mb_poll(&mbTag[src][slot], [&](int g) { return g == t - 1; })) mb_store(&bad, 1);
              pfVal = mb_load(&mbVal[src][slot]);
            }
            if (lane == 0) prevA = pfVal;
            pfTag = mb_load(&mbTag[src][t & (kMbRing - 1)]);   // frame t's boundary value, for frame t + 1
            pfVal = mb_load(&mbVal[src][t & (kMbRing - 1)]);
          } else if (lane == 0) {
            prevA = NEG;
          }
          const double s1 = alpha + (double)selfT;
          const double s2 = prevA + (double)prevT;
          const double m = fmax(s1, s2);
          const bool live = v && m != NEG;
          const float d = (float)(fmin(s1, s2) - m);
          const float ed = fast_expf(d);
          const float den = 1.f + ed;
          const double na = m + (double)fast_logf(den) + (double)xc[u];
          const float inv = __builtin_amdgcn_rcpf(den);
          const float w = (s1 >= s2) ? inv : ed * inv;
          wst[u] = live ? w : 0.f;
          alpha = live ? na : NEG;
        }
        if (feeds) {  // uniform
          // never overwrite a slot the follower has not read: every kMbRing / 2 frames make sure it finished frame t - kMbRing / 2
          if ((t & (kMbRing / 2 - 1)) == 0 && t >= kMbRing / 2)
            if (!mb_poll(&prog[wave + 1], [&](int g) { return g >= t - kMbRing / 2; })) mb_store(&bad, 1);
        }
        {
          const int slot = (t & (kMbRing - 1)) * pubStep;
          mb_store(pubVal + slot, alpha);
          asm volatile("" ::: "memory");   // value, then tag: the LDS executes a wave's operations in order
          mb_store(pubTag + slot, t);
        }
        if (fed && (t & (kMbRing / 2 - 1)) == 0 && lane == 0) mb_store(&prog[wave], t);   // (read frame t-1 of the leader: its slots up to t-1 are free)
      }
    }
#pragma unroll
    for (int u = 0; u < kFacChunk; ++u) {
      const int t = t0 + u;
      if (t >= 1 && t < T && v) w1b[(size_t)t * L + i] = wst[u];
    }
#pragma unroll
    for (int u = 0; u < kFacChunk; ++u) asm volatile("" : "+v"(xn[u]));   // one vmcnt drain per chunk (see fac_fwd_blk)
#pragma unroll
    for (int u = 0; u < kFacChunk; ++u) xc[u] = xn[u];
  }
  if (i == S - 1) loss[b] = mb_load(&bad) ? __builtin_nanf("") : (float)((double)sc * alpha);
}

template <int NW>
__global__ __launch_bounds__(64 * NW) void fac_bwd_pipe(int T, int N, int L, const int* __restrict__ target,
                                                        const int* __restrict__ targetSize,
                                                        const float* __restrict__ grad,
                                                        float* __restrict__ transGrad, FacWs ws) {
  __shared__ float mbVal[NW + 1][kMbRing];
  __shared__ int mbTag[NW + 1][kMbRing];
  __shared__ int prog[NW + 1];   // prog[w + 1] = lowest frame wave w has finished (scans run from T-1 down)
  __shared__ float sink[64];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int S = targetSize[b];
  if (S <= 0) return;  // the scatter kernel zero-fills this utterance's gradient
  for (int e = tid; e < (NW + 1) * kMbRing; e += 64 * NW) (&mbTag[0][0])[e] = -1;
  if (tid <= NW) prog[tid] = 0x3fffffff;
  __syncthreads();
  const int* y = target + (size_t)b * L;
  const float* __restrict__ w1b = ws.w1 + (size_t)b * T * L;
  float* __restrict__ dalb = ws.dal + (size_t)b * T * L;
  const float g = ws.scale[b] * grad[b];
  const int i = tid;
  const bool v = i < S;
  const int yi = v ? y[i] : 0;
  const int yp = (v && i > 0) ? y[i - 1] : 0;
  float da = (i == S - 1) ? 1.f : 0.f, accS = 0.f, accP = 0.f;
  const int lastWave = (S - 1) >> 6;
  if (wave > lastWave) return;          // (fac_scatter_k reads positions < S only)
  const bool feeds = wave > 0;          // wave w-1 consumes this wave's lane-0 advance term
  const bool fed = wave < lastWave;     // this wave's lane 63 needs wave w+1's
  const int src = fed ? wave + 1 : NW;
  float* const pubVal = lane == 0 && feeds ? &mbVal[wave][0] : &sink[lane];
  int* const pubTag = lane == 0 && feeds ? &mbTag[wave][0] : &mbTag[NW][0];
  const int pubStep = lane == 0 && feeds ? 1 : 0;
  int pfTag = -2;
  float pfVal = 0.f;

  float wc[kFacChunk], wn[kFacChunk];
#pragma unroll
  for (int u = 0; u < kFacChunk; ++u) {
    const int t = T - 1 - u;
    wc[u] = (t >= 1 && v) ? w1b[(size_t)t * L + i] : 0.f;
  }
  for (int thi = T - 1; thi >= 0; thi -= kFacChunk) {
#pragma unroll
    for (int u = 0; u < kFacChunk; ++u) {
      const int t = thi - kFacChunk - u;
      wn[u] = (t >= 1 && v) ? w1b[(size_t)t * L + i] : 0.f;
    }
    float dst[kFacChunk];
#pragma unroll
    for (int u = 0; u < kFacChunk; ++u) {
      const int t = thi - u;
      dst[u] = g * da;
      if (t >= 1) {  // uniform
        const float st = da * wc[u];
        const float adv = da - st;
        accS += st;
        accP += adv;
        const int k = T - 1 - t;   // step counter (frames run downwards)
        if (feeds && (k & (kMbRing / 2 - 1)) == 0 && k >= kMbRing / 2)
          mb_poll(&prog[wave], [&](int gg) { return gg <= t + kMbRing / 2; });   // prog[wave] = progress of wave - 1
        {
          const int slot = (t & (kMbRing - 1)) * pubStep;
          mb_store(pubVal + slot, adv);
          asm volatile("" ::: "memory");
          mb_store(pubTag + slot, t);
        }
        float right = __int_as_float(dpp_wave_shl1(__float_as_int(adv)));   // advance term of position i + 1
        if (fed) {  // uniform
          const int slot = t & (kMbRing - 1);
          if (pfTag != t) {
            mb_poll(&mbTag[src][slot], [&](int gg) { return gg == t; });
            pfVal = mb_load(&mbVal[src][slot]);
          }
          if (lane == 63) right = pfVal;
          pfTag = mb_load(&mbTag[src][(t - 1) & (kMbRing - 1)]);   // the step after this one
          pfVal = mb_load(&mbVal[src][(t - 1) & (kMbRing - 1)]);
        } else if (lane == 63) {
          right = 0.f;
        }
        da = st + right;
        if (fed && (k & (kMbRing / 2 - 1)) == 0 && lane == 63) mb_store(&prog[wave + 1], t);
      }
    }
#pragma unroll
    for (int u = 0; u < kFacChunk; ++u) {
      const int t = thi - u;
      if (t >= 0 && i < L) dalb[(size_t)t * L + i] = dst[u];
    }
#pragma unroll
    for (int u = 0; u < kFacChunk; ++u) asm volatile("" : "+v"(wn[u]));
#pragma unroll
    for (int u = 0; u < kFacChunk; ++u) wc[u] = wn[u];
  }
  float* tg = ws.tgpart ? ws.tgpart + (size_t)b * N * N : transGrad;
  if (v) {
    if (accS != 0.f) atomicAdd(&tg[(size_t)yi * N + yi], g * accS);
    if (i > 0 && accP != 0.f) atomicAdd(&tg[(size_t)yi * N + yp], g * accP);
  }
}

#endif  // W2L_PROBE

// emission gradient: dx[b][t][n] = sum_{i : y_i = n} (g dalpha_t)[i], frames in chunks of TCH.
// grid (ceil(T / TCH), B), 256 threads, dynamic LDS TCH * N floats.
__global__ __launch_bounds__(256) void fac_scatter_k(int T, int N, int L, int TCH, const int* __restrict__ target,
                                                    const int* __restrict__ targetSize,
                                                    const float* __restrict__ dal, float* __restrict__ dx) {
  extern __shared__ float rows[];
  const int b = blockIdx.y, t0 = blockIdx.x * TCH;
  int tc = T - t0;
  if (tc > TCH) tc = TCH;
  const int S = targetSize[b];
  const int n = tc * N;
  for (int k = threadIdx.x; k < n; k += 256) rows[k] = 0.f;
  __syncthreads();
  if (S > 0) {
    const int* y = target + (size_t)b * L;
    const float* dr = dal + ((size_t)b * T + t0) * L;
    // thread = lattice position (its label in a register), loop over the chunk's frames: coalesced row reads, no index
    // arithmetic per element (one division per element made this kernel VALU-bound: 70 us for 154 MB at the C4 shape)
    for (int i = threadIdx.x; i < S; i += 256) {
      float* const col = rows + y[i];
      const float* src = dr + i;
      for (int t8 = 0; t8 < tc; t8 += 8) {   // eight row loads in flight, THEN the LDS atomics (an atomic orders the loads behind it)
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = t8 + u < tc ? src[(size_t)(t8 + u) * L] : 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (v[u] != 0.f) atomicAdd(col + (t8 + u) * N, v[u]);
      }
    }
  }
  __syncthreads();
  float* out = dx + ((size_t)b * T + t0) * N;
  for (int k = threadIdx.x; k < n; k += 256) out[k] = rows[k];
}

// The same sum for small label sets (N <= 64: the letter recipes) WITHOUT atomics (round 6): a tile of kScF frames of the g dalpha rows
// is staged in LDS (coalesced), the utterance's positions are sorted by label once per workgroup (stable: position order inside a
// label), and thread (frame, label) adds its label's positions in that order -- every row element is read once, no LDS-atomic
// conflicts (300 positions over ~28 labels: ten-way), run-to-run identical by construction.  70 -> ~40 us at B = 64, T = 2000, L = 300.
constexpr int kScF = 32;
// positions of an utterance sorted by label, stable: pos[b][off[n] .. off[n + 1]) = the positions with label n in position order.
// One workgroup per utterance, thread = position (L <= 512): a counting sort on wave ballots -- per label one ballot gives the
// wave's count and every lane's rank inside it; the waves' counts and the label offsets are combined through LDS.
__global__ __launch_bounds__(512) void fac_csr_k(int N, int L, const int* __restrict__ target, const int* __restrict__ targetSize,
                                                int* __restrict__ csr) {
  fac_csr_body(N, L, target, targetSize, csr);   // (criterion_fac_mitm.hpp: the fused ASG sequence runs it inside its backward scan launch)
}

template <int NP>   // NP = ceil(L / 64): row segments per lane
__global__ __launch_bounds__(256) void fac_scatter_csr_k(int T, int N, int L, const int* __restrict__ targetSize,
                                                        const int* __restrict__ csr, const float* __restrict__ dal,
                                                        float* __restrict__ dx) {
  extern __shared__ float smSc[];
  const int b = blockIdx.y, t0 = blockIdx.x * kScF, tid = threadIdx.x;
  const int tc = min(kScF, T - t0);
  const int S = min(targetSize[b], L);
  const int Lp = L | 1;                      // odd row pitch: the frame index does not pick the bank
  float* tile = smSc;                          // [kScF][Lp]
  int* pos = (int*)(smSc + kScF * Lp);         // [L] positions sorted by label
  int* off = pos + L;                        // [N + 1]
  float* out = dx + ((size_t)b * T + t0) * N;
  if (S <= 0) {
    for (int k = tid; k < tc * N; k += 256) out[k] = 0.f;
    return;
  }
  {
    const int* cp = csr + (size_t)b * (L + 68);
    for (int i = tid; i < S; i += 256) pos[i] = cp[i];
    if (tid <= N) off[tid] = cp[L + tid];
    // stage the rows: wave w takes frames w, w + 4, ...; lanes over positions.  Every load of the wave's eight frames is issued
    // before the first LDS store (written as load -> store per element the loop is one memory round trip per element: 70 us)
    const float* dr = dal + ((size_t)b * T + t0) * L;
    const int w = tid >> 6, lane = tid & 63;
    float v[kScF / 4][NP];
#pragma unroll
    for (int q = 0; q < kScF / 4; ++q)
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int f = w + 4 * q, i = lane + 64 * p;
        v[q][p] = (f < tc && i < S) ? dr[(size_t)f * L + i] : 0.f;
      }
#pragma unroll
    for (int q = 0; q < kScF / 4; ++q)
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int f = w + 4 * q, i = lane + 64 * p;
        if (f < tc && i < S) tile[f * Lp + i] = v[q][p];
      }
  }
  __syncthreads();
  const int f = tid >> 3, nq = tid & 7;
  if (f < tc) {
    const float* row = tile + f * Lp;
    for (int n = nq; n < N; n += 8) {
      float acc = 0.f;
      for (int k = off[n]; k < off[n + 1]; ++k) acc += row[pos[k]];
      out[f * N + n] = acc;
    }
  }
}

__global__ void reduce_over_b_fac(int B, size_t n, const float* __restrict__ part, float* __restrict__ out) {
  size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  float s = 0.f;
  int b = 0;
  for (; b + 7 < B; b += 8) {   // eight loads in flight; the additions keep their order
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(b + u) * n + k];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; b < B; ++b) s += part[(size_t)b * n + k];
  out[k] = s;
}

// The ASG criterion's backward tail in one launch (fac_backward_asg): the first blocks reduce ForceAlignmentCriterion's transition-
// gradient partials and subtract them from FullConnectionCriterion's transition gradient (reduce_over_b_fac), the others
// subtract its input gradient from FullConnectionCriterion's (w2l_axpy with alpha = -1: y + (-1) x == y - x) -- one launch behind
// the join of the two streams instead of a reduce in front of it and two axpy behind it.
constexpr int kAsgCombineReduceThreads = 256;
__global__ __launch_bounds__(256) void asg_bwd_combine_k(int parts, unsigned nn, const float* __restrict__ part, float* __restrict__ dTrans,
                                                         float* __restrict__ dEm, const float* __restrict__ dx2, size_t n, unsigned redBlocks,
                                                         const float* __restrict__ fccPart = nullptr, int fccB = 0, int fccStride = 0) {
  if (blockIdx.x < redBlocks) {
    const unsigned k = blockIdx.x * 256u + threadIdx.x;
    if (k >= nn) return;
    float prev;
    if (fccPart) {   // FullConnectionCriterion's sum over the utterances, in reduce_over_b's order (criterion_fcc.hip)
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
      int b = 0;
      for (; b + 7 < fccB; b += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = fccPart[(size_t)(b + u) * fccStride * nn + k];
        s0 += v[0]; s1 += v[1]; s2 += v[2]; s3 += v[3];
        s0 += v[4]; s1 += v[5]; s2 += v[6]; s3 += v[7];
      }
      for (; b < fccB; ++b) s0 += fccPart[(size_t)b * fccStride * nn + k];
      prev = (s0 + s1) + (s2 + s3);
    } else {
      prev = dTrans[k];
    }
    float s = 0.f;
    int b = 0;
    for (; b + 7 < parts; b += 8) {   // (reduce_over_b_fac's order)
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(b + u) * nn + k];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; b < parts; ++b) s += part[(size_t)b * nn + k];
    dTrans[k] = prev - s;
    return;
  }
  const size_t n4 = n >> 2, stride = (size_t)(gridDim.x - redBlocks) * 256;
  for (size_t i = (size_t)(blockIdx.x - redBlocks) * 256 + threadIdx.x; i < n4; i += stride) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    v4f a = *(const v4f*)(dEm + 4 * i);
    const v4f x = __builtin_nontemporal_load((const v4f*)(dx2 + 4 * i));   // read once
    a -= x;
    *(v4f*)(dEm + 4 * i) = a;
  }
  for (size_t e = (n4 << 2) + (size_t)(blockIdx.x - redBlocks) * 256 + threadIdx.x; e < n; e += stride) dEm[e] -= dx2[e];
}

constexpr int kFacBt = 128;  // backtrace chunk

template <int P>
__global__ __launch_bounds__(64) void fac_vit(int T, int N, int L, const float* __restrict__ x,
                                              const int* __restrict__ target,
                                              const int* __restrict__ targetSize,
                                              const float* __restrict__ trans,
                                              int* __restrict__ bestPaths, unsigned char* bpAll) {
  extern __shared__ unsigned char sBp[];  // kFacBt * L bytes, then kFacBt ints
  const int b = blockIdx.x;
  const int lane = threadIdx.x;
  const int S = targetSize[b];
  int* pb = bestPaths + (size_t)b * T;
  if (S <= 0) {
    for (int t = lane; t < T; t += 64) pb[t] = -1;
    return;
  }
  const int* y = target + (size_t)b * L;
  const float* xb = x + (size_t)b * T * N;
  unsigned char* bp = bpAll + (size_t)b * T * L;
  const float NEG = -INFINITY;
  int yi[P];
  float selfT[P], prevT[P], alpha[P];
#pragma unroll
  for (int p = 0; p < P; ++p) {
    int i = lane * P + p;
    bool v = i < S;
    yi[p] = v ? y[i] : 0;
    int ypv = (v && i > 0) ? y[i - 1] : 0;
    selfT[p] = v ? trans[(size_t)yi[p] * N + yi[p]] : 0.f;
    prevT[p] = (v && i > 0) ? trans[(size_t)yi[p] * N + ypv] : 0.f;
    alpha[p] = NEG;
  }
  if (lane == 0) alpha[0] = xb[yi[0]];
  for (int t = 1; t < T; ++t) {
    float carry = lane_shift_up(alpha[P - 1], NEG);
    float prevA = carry;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int i = lane * P + p;
      float cur = alpha[p];
      float s1 = cur + selfT[p];
      float s2 = prevA + prevT[p];
      float xv = (i < S) ? xb[(size_t)t * N + yi[p]] : 0.f;
      bool adv = s2 > s1;  // ties: stay wins (oracle order)
      float na = (adv ? s2 : s1) + xv;
      if (i < S) bp[(size_t)t * L + i] = adv ? 1 : 0;
      prevA = cur;
      alpha[p] = (i < S) ? na : NEG;
    }
  }
  __syncthreads();
  int* sPath = (int*)(sBp + (size_t)kFacBt * L);
  int cur = S - 1;
  for (int thi = T - 1; thi >= 0; thi -= kFacBt) {
    int tlo = thi - kFacBt + 1;
    if (tlo < 0) tlo = 0;
    int nsteps = thi - tlo + 1;
    for (int k = lane; k < nsteps * L; k += 64) sBp[k] = bp[(size_t)tlo * L + k];
    __syncthreads();
    if (lane == 0) {
      for (int t = thi; t >= tlo; --t) {
        sPath[t - tlo] = y[cur];
        if (t >= 1 && sBp[(t - tlo) * L + cur]) --cur;
      }
    }
    cur = __builtin_amdgcn_readfirstlane(cur);
    __syncthreads();
    for (int k = lane; k < nsteps; k += 64) pb[tlo + k] = sPath[k];
    __syncthreads();
  }
}

}  // namespace w2l

using namespace w2l;


W2L_API size_t w2l_fac_workspace_size(int B, int T, int N, int L) {
  if (B <= 0 || T <= 0 || N <= 0 || L <= 0) return 0;
  size_t sz = 2 * align_up((size_t)B * T * L * sizeof(float), 256) + align_up((size_t)B * sizeof(float), 256) +
              align_up((size_t)B * sizeof(int), 256);
  if (fac_use_partials(B, N)) sz += align_up((size_t)fac_partial_sets(N) * B * N * N * sizeof(float), 256);
  if (N <= 64) sz += align_up((size_t)B * (L + 68) * sizeof(int), 256);
  if (N <= 32) sz += align_up((size_t)B * T * 32 * sizeof(double), 256) + 2 * align_up((size_t)B * T * sizeof(float), 256) +
                     2 * align_up((size_t)B * 320 * 16, 256) + align_up((size_t)B * 320 * sizeof(float), 256);
  return sz;
}

#define W2L_FAC_DISPATCH(KERNEL, SHMEM, ...)                                                   \
  do {                                                                                         \
    if (L <= 64) hipLaunchKernelGGL(KERNEL<1>, dim3(B), dim3(64), SHMEM, s, __VA_ARGS__);      \
    else if (L <= 128) hipLaunchKernelGGL(KERNEL<2>, dim3(B), dim3(64), SHMEM, s, __VA_ARGS__); \
    else if (L <= 256) hipLaunchKernelGGL(KERNEL<4>, dim3(B), dim3(64), SHMEM, s, __VA_ARGS__); \
    else hipLaunchKernelGGL(KERNEL<8>, dim3(B), dim3(64), SHMEM, s, __VA_ARGS__);              \
  } while (0)

#define W2L_FAC_BLK_DISPATCH(KERNEL, ...)                                                      \
  do {                                                                                         \
    if (L <= 64) hipLaunchKernelGGL((KERNEL<1, 1>), dim3(B), dim3(64), 0, s, __VA_ARGS__);       \
    else if (L <= 128) hipLaunchKernelGGL((KERNEL<2, 1>), dim3(B), dim3(128), 0, s, __VA_ARGS__); \
    else if (L <= 256) hipLaunchKernelGGL((KERNEL<4, 1>), dim3(B), dim3(256), 0, s, __VA_ARGS__); \
    else hipLaunchKernelGGL((KERNEL<4, 2>), dim3(B), dim3(256), 0, s, __VA_ARGS__);              \
  } while (0)

W2L_API int w2l_fac_forward(int B, int T, int N, int L, int scaleMode, const float* input,
                            const int* target, const int* targetSize, const float* trans,
                            float* loss, void* workspace, w2l_stream_t stream) {
  if (B <= 0 || T <= 0 || N <= 0 || L <= 0 || !input || !target || !targetSize || !trans || !loss || !workspace)
    return W2L_EINVAL;
  if (L > 512) return W2L_EUNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  FacWs ws = fac_ws(workspace, B, T, N, L);
  if (fac_lin_path(N, L)) {
    // scaled linear domain (criterion_fac_lin.hpp).  Product: label rows by a pre-pass, one position per thread with its own
    // exponent, the waves of an utterance as a skewed pipeline that synchronises once per 16 frames (fac_fwd_plin).  The probe
    // library runs the other generations for A/B work: W2L_FAC_GEN = blin2 (the same with one workgroup barrier per frame),
    // blin (rows by a sixth wave), wave (one wave per utterance, five positions per lane, flags what it cannot hold for the
    // log-domain kernel behind it); W2L_ASG_OLD=1 = the round-3 log-domain kernels.
    static const int gen = [] {
      const char* e = tune_env("W2L_FAC_GEN");
      if (!e) return 0;
      return !strcmp(e, "blin2") ? 1 : !strcmp(e, "blin") ? 2 : !strcmp(e, "wave") ? 3 : 0;
    }();
    const int nw = (L + 63) / 64;
    if (fac_mitm_path()) {
      hipLaunchKernelGGL(fac_rows_k, dim3((unsigned)((T + kFacRowsPerWave * kFacRowsWaves - 1) / (kFacRowsPerWave * kFacRowsWaves)), (unsigned)B), dim3(64 * kFacRowsWaves), 0, s, T, N, input,
                         trans, ws.crow, ws.zmax, ws.zspr, (const int*)nullptr, 0, (int*)nullptr);
      W2L_LAUNCH_CHECK();
#define W2L_FAC_M_GO(NWV) hipLaunchKernelGGL((fac_mitm_fwd<NWV>), dim3(B, fac_mitm_only() < 0 ? 2 : 1), dim3(64 * NWV), mitm_excl(B, (const void*)fac_mitm_fwd<NWV>), s, T, N, L, target, targetSize, trans, ws, fac_mitm_only() < 0 ? 0 : fac_mitm_only(), fac_mitm_abl())
      switch (nw) {
        case 1: W2L_FAC_M_GO(1); break;
        case 2: W2L_FAC_M_GO(2); break;
        case 3: W2L_FAC_M_GO(3); break;
        case 4: W2L_FAC_M_GO(4); break;
        default: W2L_FAC_M_GO(5); break;
      }
#undef W2L_FAC_M_GO
      W2L_LAUNCH_CHECK();
      hipLaunchKernelGGL(fac_mitm_finish, dim3(B), dim3(kFacFinishThreads), 0, s, T, N, L, scaleMode, target, targetSize, trans, loss, ws);
      W2L_LAUNCH_CHECK();
      // the utterances the range check flagged: the whole forward scan again in the log domain (returns at once for the others)
      if (L > 256) hipLaunchKernelGGL((fac_fwd_blk<8, 1>), dim3(B), dim3(512), 0, s, T, N, L, scaleMode, input, target, targetSize, trans, loss, ws, (const int*)ws.redo);
      else W2L_FAC_BLK_DISPATCH(fac_fwd_blk, T, N, L, scaleMode, input, target, targetSize, trans, loss, ws, (const int*)ws.redo);
      W2L_LAUNCH_CHECK();
      return W2L_OK;
    }
    if (gen == 3) {
#define W2L_FAC_LIN_GO(PP) hipLaunchKernelGGL(fac_fwd_lin<PP>, dim3(B), dim3(64), 0, s, T, N, L, scaleMode, input, target, targetSize, trans, loss, ws)
      switch (nw) {
        case 1: W2L_FAC_LIN_GO(1); break;
        case 2: W2L_FAC_LIN_GO(2); break;
        case 3: W2L_FAC_LIN_GO(3); break;
        case 4: W2L_FAC_LIN_GO(4); break;
        default: W2L_FAC_LIN_GO(5); break;
      }
#undef W2L_FAC_LIN_GO
      W2L_LAUNCH_CHECK();
      if (L > 256) hipLaunchKernelGGL((fac_fwd_blk<8, 1>), dim3(B), dim3(512), 0, s, T, N, L, scaleMode, input, target, targetSize, trans, loss, ws, (const int*)ws.redo);
      else W2L_FAC_BLK_DISPATCH(fac_fwd_blk, T, N, L, scaleMode, input, target, targetSize, trans, loss, ws, (const int*)ws.redo);
      W2L_LAUNCH_CHECK();
      return W2L_OK;
    }
    if (gen == 2) {
#define W2L_FAC_BLIN_GO(NWV) hipLaunchKernelGGL((fac_fwd_blin<NWV>), dim3(B), dim3(64 * (NWV + 1)), 0, s, T, N, L, scaleMode, input, target, targetSize, trans, loss, ws)
      switch (nw) {
        case 1: W2L_FAC_BLIN_GO(1); break;
        case 2: W2L_FAC_BLIN_GO(2); break;
        case 3: W2L_FAC_BLIN_GO(3); break;
        case 4: W2L_FAC_BLIN_GO(4); break;
        default: W2L_FAC_BLIN_GO(5); break;
      }
#undef W2L_FAC_BLIN_GO
      W2L_LAUNCH_CHECK();
      return W2L_OK;
    }
    hipLaunchKernelGGL(fac_rows_k, dim3((unsigned)((T + kFacRowsPerWave * kFacRowsWaves - 1) / (kFacRowsPerWave * kFacRowsWaves)), (unsigned)B), dim3(64 * kFacRowsWaves), 0, s, T, N, input,
                       trans, ws.crow, ws.zmax, ws.zspr);
    W2L_LAUNCH_CHECK();
#define W2L_FAC_P_GO(K, NWV) hipLaunchKernelGGL((K<NWV>), dim3(B), dim3(64 * NWV), 0, s, T, N, L, scaleMode, target, targetSize, trans, loss, ws)
#define W2L_FAC_P_SWITCH(K)               \
    switch (nw) {                         \
      case 1: W2L_FAC_P_GO(K, 1); break;  \
      case 2: W2L_FAC_P_GO(K, 2); break;  \
      case 3: W2L_FAC_P_GO(K, 3); break;  \
      case 4: W2L_FAC_P_GO(K, 4); break;  \
      default: W2L_FAC_P_GO(K, 5); break; \
    }
    if (gen == 1) { W2L_FAC_P_SWITCH(fac_fwd_blin2) } else { W2L_FAC_P_SWITCH(fac_fwd_plin) }
#undef W2L_FAC_P_SWITCH
#undef W2L_FAC_P_GO
    W2L_LAUNCH_CHECK();
    if (gen == 0) {
      // fac_fwd_plin flags the utterances whose label-score spread + transition ratios could carry a position's fp64 value out
      // of range (kFacPlinSafeBits); the log-domain kernel recomputes exactly those and returns at once for the others
      if (L > 256) hipLaunchKernelGGL((fac_fwd_blk<8, 1>), dim3(B), dim3(512), 0, s, T, N, L, scaleMode, input, target, targetSize, trans, loss, ws, (const int*)ws.redo);
      else W2L_FAC_BLK_DISPATCH(fac_fwd_blk, T, N, L, scaleMode, input, target, targetSize, trans, loss, ws, (const int*)ws.redo);
      W2L_LAUNCH_CHECK();
    }
    return W2L_OK;
  }
  static const int waveMode = [] { const char* e = tune_env("W2L_FAC_WAVE"); return e ? atoi(e) : 0; }();
  if (waveMode && L <= 384) {  // experiment: one wave per utterance, ceil(L/64) positions per lane, DPP neighbour exchange
    const int P = (L + 63) / 64;
#define W2L_FAC_WAVE_GO(PP) hipLaunchKernelGGL(fac_fwd<PP>, dim3(B), dim3(64), 0, s, T, N, L, scaleMode, input, target, targetSize, trans, loss, ws)
    switch (P) {
      case 1: W2L_FAC_WAVE_GO(1); break;
      case 2: W2L_FAC_WAVE_GO(2); break;
      case 3: W2L_FAC_WAVE_GO(3); break;
      case 4: W2L_FAC_WAVE_GO(4); break;
      case 5: W2L_FAC_WAVE_GO(5); break;
      default: W2L_FAC_WAVE_GO(6); break;
    }
#undef W2L_FAC_WAVE_GO
    W2L_LAUNCH_CHECK();
    return W2L_OK;
  }
  // workgroup shapes measured at B = 64, T = 2000, L = 300 (profiles/r01_run56_fac_shapes.log): one position per lane
  // wins the forward scan -- 8 waves x 1: 0.60 ms, 4 x 2: 0.77, 2 x 3: 0.87, 1 x 5: 1.20 -- the backward scan (a
  // handful of fp32 operations per step, barrier-bound) is fastest with 4 waves x 2
#ifdef W2L_PROBE
  if (tune_env("W2L_FAC_PIPE")) {   // probe build: skewed wave pipeline (one position per lane, no per-frame workgroup barrier) -- measured NOT faster
    const int nw = (L + 63) / 64;
#define W2L_FAC_PIPE_GO(NWV) hipLaunchKernelGGL((fac_fwd_pipe<NWV>), dim3(B), dim3(64 * NWV), 0, s, T, N, L, scaleMode, input, target, targetSize, trans, loss, ws)
    switch (nw) {
      case 1: W2L_FAC_PIPE_GO(1); break;
      case 2: W2L_FAC_PIPE_GO(2); break;
      case 3: W2L_FAC_PIPE_GO(3); break;
      case 4: W2L_FAC_PIPE_GO(4); break;
      case 5: W2L_FAC_PIPE_GO(5); break;
      case 6: W2L_FAC_PIPE_GO(6); break;
      case 7: W2L_FAC_PIPE_GO(7); break;
      default: W2L_FAC_PIPE_GO(8); break;
    }
#undef W2L_FAC_PIPE_GO
    W2L_LAUNCH_CHECK();
    return W2L_OK;
  }
#endif
  if (L > 256) hipLaunchKernelGGL((fac_fwd_blk<8, 1>), dim3(B), dim3(512), 0, s, T, N, L, scaleMode, input, target, targetSize, trans, loss, ws);
  else W2L_FAC_BLK_DISPATCH(fac_fwd_blk, T, N, L, scaleMode, input, target, targetSize, trans, loss, ws);
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}

W2L_API int w2l_fac_backward(int B, int T, int N, int L, const int* target, const int* targetSize,
                             const float* grad, float* inputGrad, float* transGrad,
                             void* workspace, w2l_stream_t stream) {
  if (B <= 0 || T <= 0 || N <= 0 || L <= 0 || !target || !targetSize || !grad || !inputGrad || !transGrad || !workspace)
    return W2L_EINVAL;
  if (L > 512 || N > 32768) return W2L_EUNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  FacWs ws = fac_ws(workspace, B, T, N, L);
  size_t n = (size_t)N * N;
  const bool mitm = fac_lin_path(N, L) && fac_mitm_path();
  const int sets = mitm ? 2 : 1;   // (see fac_partial_sets)
  if (N <= 64) {   // positions by label for the scatter kernel (tiny; ahead of the scans)
    hipLaunchKernelGGL(fac_csr_k, dim3(B), dim3(512), 0, s, N, L, target, targetSize, ws.csr);
    W2L_LAUNCH_CHECK();
  }
  if (ws.tgpart) W2L_HIP_CHECK(hipMemsetAsync(ws.tgpart, 0, (size_t)sets * B * n * sizeof(float), s));
  else W2L_HIP_CHECK(hipMemsetAsync(transGrad, 0, n * sizeof(float), s));
  // product (N <= 32, L <= 320): the pipelined backward scan fac_bwd_plin; probe: W2L_FAC_BWD = wave (one wave per utterance),
  // blk51 (five waves x one position, a workgroup barrier per frame), blk42 (the round-3 shape)
  static const int bgen = [] {
    const char* e = tune_env("W2L_FAC_BWD");
    if (!e) return 0;
    return !strcmp(e, "wave") ? 1 : !strcmp(e, "blk51") ? 2 : !strcmp(e, "blk42") ? 3 : 0;
  }();
  if (mitm) {
#define W2L_FAC_M_GO(NWV) hipLaunchKernelGGL((fac_mitm_bwd<NWV>), dim3(B, fac_mitm_only() < 0 ? 2 : 1), dim3(64 * NWV), mitm_excl(B, (const void*)fac_mitm_bwd<NWV>), s, T, N, L, target, targetSize, grad, transGrad, ws, fac_mitm_only() < 0 ? 0 : fac_mitm_only())
    switch ((L + 63) / 64) {
      case 1: W2L_FAC_M_GO(1); break;
      case 2: W2L_FAC_M_GO(2); break;
      case 3: W2L_FAC_M_GO(3); break;
      case 4: W2L_FAC_M_GO(4); break;
      default: W2L_FAC_M_GO(5); break;
    }
#undef W2L_FAC_M_GO
  } else if (fac_lin_path(N, L) && bgen == 0 && tune_env("W2L_FAC_BWD32")) {   // probe: 32 frames per chunk (half the per-chunk round trips)
#define W2L_FAC_PB_GO(NWV) hipLaunchKernelGGL((fac_bwd_plin<NWV, 32>), dim3(B), dim3(64 * NWV), 0, s, T, N, L, target, targetSize, grad, transGrad, ws)
    switch ((L + 63) / 64) {
      case 1: W2L_FAC_PB_GO(1); break;
      case 2: W2L_FAC_PB_GO(2); break;
      case 3: W2L_FAC_PB_GO(3); break;
      case 4: W2L_FAC_PB_GO(4); break;
      default: W2L_FAC_PB_GO(5); break;
    }
#undef W2L_FAC_PB_GO
  } else if (fac_lin_path(N, L) && bgen == 0) {
#define W2L_FAC_PB_GO(NWV) hipLaunchKernelGGL((fac_bwd_plin<NWV>), dim3(B), dim3(64 * NWV), 0, s, T, N, L, target, targetSize, grad, transGrad, ws)
    switch ((L + 63) / 64) {
      case 1: W2L_FAC_PB_GO(1); break;
      case 2: W2L_FAC_PB_GO(2); break;
      case 3: W2L_FAC_PB_GO(3); break;
      case 4: W2L_FAC_PB_GO(4); break;
      default: W2L_FAC_PB_GO(5); break;
    }
#undef W2L_FAC_PB_GO
  } else if (fac_lin_path(N, L) && bgen == 1) {
#define W2L_FAC_LIN_GO(PP) hipLaunchKernelGGL(fac_bwd_wave<PP>, dim3(B), dim3(64), 0, s, T, N, L, target, targetSize, grad, transGrad, ws)
    switch ((L + 63) / 64) {
      case 1: W2L_FAC_LIN_GO(1); break;
      case 2: W2L_FAC_LIN_GO(2); break;
      case 3: W2L_FAC_LIN_GO(3); break;
      case 4: W2L_FAC_LIN_GO(4); break;
      default: W2L_FAC_LIN_GO(5); break;
    }
#undef W2L_FAC_LIN_GO
  } else if (bgen == 2 && L > 256 && L <= 320) {
    hipLaunchKernelGGL((fac_bwd_blk<5, 1>), dim3(B), dim3(320), 0, s, T, N, L, target, targetSize, grad, transGrad, ws);
  } else
#ifdef W2L_PROBE
  if (tune_env("W2L_FAC_PIPE")) {
    const int nw = (L + 63) / 64;
#define W2L_FAC_PIPE_GO(NWV) hipLaunchKernelGGL((fac_bwd_pipe<NWV>), dim3(B), dim3(64 * NWV), 0, s, T, N, L, target, targetSize, grad, transGrad, ws)
    switch (nw) {
      case 1: W2L_FAC_PIPE_GO(1); break;
      case 2: W2L_FAC_PIPE_GO(2); break;
      case 3: W2L_FAC_PIPE_GO(3); break;
      case 4: W2L_FAC_PIPE_GO(4); break;
      case 5: W2L_FAC_PIPE_GO(5); break;
      case 6: W2L_FAC_PIPE_GO(6); break;
      case 7: W2L_FAC_PIPE_GO(7); break;
      default: W2L_FAC_PIPE_GO(8); break;
    }
#undef W2L_FAC_PIPE_GO
  } else
#endif
  {
    W2L_FAC_BLK_DISPATCH(fac_bwd_blk, T, N, L, target, targetSize, grad, transGrad, ws);
  }
  W2L_LAUNCH_CHECK();
  if (N <= 64) {
    const size_t shmem = (size_t)kScF * (L | 1) * sizeof(float) + (size_t)(L + N + 1) * sizeof(int);   // <= 68 KiB at L = 512
    const dim3 grid((unsigned)((T + kScF - 1) / kScF), (unsigned)B);
#define W2L_FAC_SC_GO(NPV)                                                                                                       \
  do {                                                                                                                           \
    static bool attr[64] = {};                                                                                                   \
    if (shmem > 64 * 1024 && first_on_device(attr))                                                                              \
      W2L_HIP_CHECK(hipFuncSetAttribute((const void*)fac_scatter_csr_k<NPV>, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024)); \
    hipLaunchKernelGGL(fac_scatter_csr_k<NPV>, grid, dim3(256), shmem, s, T, N, L, targetSize, ws.csr, ws.dal, inputGrad);       \
  } while (0)
    switch ((L + 63) / 64) {
      case 1: W2L_FAC_SC_GO(1); break;
      case 2: W2L_FAC_SC_GO(2); break;
      case 3: W2L_FAC_SC_GO(3); break;
      case 4: W2L_FAC_SC_GO(4); break;
      case 5: W2L_FAC_SC_GO(5); break;
      case 6: W2L_FAC_SC_GO(6); break;
      case 7: W2L_FAC_SC_GO(7); break;
      default: W2L_FAC_SC_GO(8); break;
    }
#undef W2L_FAC_SC_GO
    W2L_LAUNCH_CHECK();
  } else {
    int tch = 32768 / N;  // <= 128 KiB of LDS rows
    if (tch > 16) tch = 16;
    if (tch < 1) tch = 1;
    const size_t shmem = (size_t)tch * N * sizeof(float);
    if (shmem > 64 * 1024)
      W2L_HIP_CHECK(hipFuncSetAttribute((const void*)fac_scatter_k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    hipLaunchKernelGGL(fac_scatter_k, dim3((unsigned)((T + tch - 1) / tch), (unsigned)B), dim3(256), shmem, s, T, N, L, tch,
                       target, targetSize, ws.dal, inputGrad);
    W2L_LAUNCH_CHECK();
  }
  if (ws.tgpart) {
    hipLaunchKernelGGL(reduce_over_b_fac, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, sets * B, n, ws.tgpart, transGrad);
    W2L_LAUNCH_CHECK();
  }
  return W2L_OK;
}

// ---- the ASG criterion's fused sequence (criterion_asg_fused.hpp)
namespace w2l {

bool fac_asg_fused_ok(int B, int T, int N, int L) {
  return B > 0 && T > 0 && L > 0 && fac_lin_path(N, L) && fac_mitm_path() && fac_mitm_only() < 0 && fac_use_partials(B, N);
}

int fac_forward_asg(int B, int T, int N, int L, int scaleMode, const float* input, const int* target, int* ts, const float* trans,
                    float* loss2, float* minuend, void* workspace, hipStream_t s, AsgHook hook, void* arg) {
  if (!fac_asg_fused_ok(B, T, N, L)) return W2L_EUNSUPPORTED;
  if (!input || !target || !ts || !trans || !loss2 || !workspace) return W2L_EINVAL;
  FacWs ws = fac_ws(workspace, B, T, N, L);
  hipLaunchKernelGGL(fac_rows_k, dim3((unsigned)((T + kFacRowsPerWave * kFacRowsWaves - 1) / (kFacRowsPerWave * kFacRowsWaves)), (unsigned)B), dim3(64 * kFacRowsWaves), 0, s, T, N, input,
                     trans, ws.crow, ws.zmax, ws.zspr, target, L, ts, ws.tgpart, (unsigned)((size_t)2 * B * N * N));
  W2L_LAUNCH_CHECK();
  if (hook) hook(arg, ASG_TARGET_SIZES_QUEUED);
#define W2L_FAC_M_GO(NWV) hipLaunchKernelGGL((fac_mitm_fwd<NWV>), dim3(B, 2), dim3(64 * NWV), mitm_excl(B, (const void*)fac_mitm_fwd<NWV>), s, T, N, L, target, (const int*)ts, trans, ws, 0, 0)
  switch ((L + 63) / 64) {
    case 1: W2L_FAC_M_GO(1); break;
    case 2: W2L_FAC_M_GO(2); break;
    case 3: W2L_FAC_M_GO(3); break;
    case 4: W2L_FAC_M_GO(4); break;
    default: W2L_FAC_M_GO(5); break;
  }
#undef W2L_FAC_M_GO
  W2L_LAUNCH_CHECK();
  if (hook && minuend) hook(arg, ASG_NEED_FCC_LOSS);
  hipLaunchKernelGGL(fac_mitm_finish_all, dim3(B), dim3(kFacFinishThreads), 0, s, T, N, L, scaleMode, input, target, (const int*)ts, trans, loss2, ws, minuend);
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}

int fac_backward_asg(int B, int T, int N, int L, const int* target, const int* ts, const float* grad, float* dEm, float* dTrans,
                     float* dx2, void* workspace, bool partialsClear, const float* fccPart, int fccStride, hipStream_t s, hipStream_t sOut,
                     AsgHook hook, void* arg) {
  if (!fac_asg_fused_ok(B, T, N, L)) return W2L_EUNSUPPORTED;
  if (!target || !ts || !grad || !dEm || !dTrans || !dx2 || !workspace) return W2L_EINVAL;
  FacWs ws = fac_ws(workspace, B, T, N, L);
  const size_t n = (size_t)N * N;
  // the transition-gradient partials: cleared by forward's label-row launch; a second backward pass on the same forward fills them here
  if (!partialsClear) W2L_HIP_CHECK(hipMemsetAsync(ws.tgpart, 0, (size_t)2 * B * n * sizeof(float), s));
#define W2L_FAC_M_GO(NWV) hipLaunchKernelGGL((fac_mitm_bwd<NWV>), dim3(B, 2), dim3(64 * NWV), mitm_excl(B, (const void*)fac_mitm_bwd<NWV>), s, T, N, L, target, ts, grad, dTrans, ws, 0, 1)
  switch ((L + 63) / 64) {
    case 1: W2L_FAC_M_GO(1); break;
    case 2: W2L_FAC_M_GO(2); break;
    case 3: W2L_FAC_M_GO(3); break;
    case 4: W2L_FAC_M_GO(4); break;
    default: W2L_FAC_M_GO(5); break;
  }
#undef W2L_FAC_M_GO
  W2L_LAUNCH_CHECK();
  {
    const size_t shmem = (size_t)kScF * (L | 1) * sizeof(float) + (size_t)(L + N + 1) * sizeof(int);   // < 64 KiB at L <= 320
    const dim3 grid((unsigned)((T + kScF - 1) / kScF), (unsigned)B);
#define W2L_FAC_SC_GO(NPV) hipLaunchKernelGGL(fac_scatter_csr_k<NPV>, grid, dim3(256), shmem, s, T, N, L, ts, (const int*)ws.csr, (const float*)ws.dal, dx2)
    switch ((L + 63) / 64) {
      case 1: W2L_FAC_SC_GO(1); break;
      case 2: W2L_FAC_SC_GO(2); break;
      case 3: W2L_FAC_SC_GO(3); break;
      case 4: W2L_FAC_SC_GO(4); break;
      default: W2L_FAC_SC_GO(5); break;
    }
#undef W2L_FAC_SC_GO
    W2L_LAUNCH_CHECK();
  }
  if (hook) hook(arg, ASG_NEED_FCC_GRADS);
  {
    const size_t tot = (size_t)B * T * N;
    const unsigned redBlocks = (unsigned)((n + 255) / 256);
    size_t axBlocks = ((tot >> 2) + 255) / 256;
    if (axBlocks > 4096) axBlocks = 4096;
    if (axBlocks < 1) axBlocks = 1;
    hipLaunchKernelGGL(asg_bwd_combine_k, dim3(redBlocks + (unsigned)axBlocks), dim3(256), 0, sOut, 2 * B, (unsigned)n, (const float*)ws.tgpart, dTrans,
                       dEm, (const float*)dx2, tot, redBlocks, fccPart, B, fccStride);
    W2L_LAUNCH_CHECK();
  }
  return W2L_OK;
}

}  // namespace w2l

W2L_API int w2l_fac_range_flags(int B, int T, int N, int L, const void* workspace, int* flags, w2l_stream_t stream) {
  if (B <= 0 || T <= 0 || N <= 0 || L <= 0 || !workspace || !flags) return W2L_EINVAL;
  if (L > 512) return W2L_EUNSUPPORTED;
  if (!fac_lin_path(N, L)) {   // every utterance runs on the log-domain kernels: nothing is ever handed over
    W2L_HIP_CHECK(hipMemsetAsync(flags, 0, (size_t)B * sizeof(int), (hipStream_t)stream));
    return W2L_OK;
  }
  const FacWs ws = fac_ws((void*)workspace, B, T, N, L);
  W2L_HIP_CHECK(hipMemcpyAsync(flags, ws.redo, (size_t)B * sizeof(int), hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return W2L_OK;
}

W2L_API int w2l_fac_viterbi(int B, int T, int N, int L, const float* input, const int* target,
                            const int* targetSize, const float* trans, int* bestPaths,
                            void* workspace, w2l_stream_t stream) {
  if (B <= 0 || T <= 0 || N <= 0 || L <= 0 || !input || !target || !targetSize || !trans || !bestPaths || !workspace)
    return W2L_EINVAL;
  if (L > 512) return W2L_EUNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  FacWs ws = fac_ws(workspace, B, T, N, L);
  size_t shmem = (size_t)kFacBt * L + kFacBt * sizeof(int);
  W2L_FAC_DISPATCH(fac_vit, shmem, T, N, L, input, target, targetSize, trans, bestPaths, ws.bp);
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}
