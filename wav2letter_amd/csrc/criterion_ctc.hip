// criterion_ctc.hip -- ConnectionistTemporalClassificationCriterion for gfx950.
//
// Replaces Flashlight's CTC criterion (CUDA build: warp-ctc; CPU build:
// flashlight/lib/sequence/criterion/cpu/ConnectionistTemporalClassificationCriterion.cpp,
// un-vendored).  Reference call sites: recipes/slimIPL/src/Train.cpp:406-407, :1675;
// blank is the LAST class (Train.cpp:248-251).  Math: SURVEY.md App. B.4; CPU
// restatement: oracle/criterion_oracle.c (== torch ctc_loss(blank=N-1)).
//
// At the north-star size (N = 9998 word pieces) the criterion is HBM-bound on
// streaming the [B][T][N] emissions, so it is split by access pattern:
//   ctc_rows_lse   one workgroup per (b,t) row: the row is read ONCE into
//                  registers (coalesced, 16 B/lane where alignment allows) and
//                  reduced to lse[b][t]; the <= 2L+1 label log-probs the lattice
//                  needs are gathered from the register-resident row's source.
//   ctc_scan       two wavefronts per utterance, side by side: the alpha scan and
//                  the beta scan over the 2L+1 extended labels (positions blocked
//                  over lanes, fp64 carries with fp32 log-sum-exp corrections);
//                  the alpha wave also writes the loss.
//   ctc_rows_grad  one workgroup per row: grad = g*(softmax(x) - occupancy):
//                  streams x once more, writes grad once, then subtracts the
//                  <= 2L+1 occupancies gamma[t][s] = exp(alpha+beta-lp-logZ) of
//                  that frame.
// Algorithmic HBM bytes: 4BTN (fwd) + 8BTN (bwd) = 12*B*T*N.
#include "common.hpp"

namespace w2l {

constexpr int kRowThreads = 256;
constexpr int kRowMaxPer = 48;  // register-resident row: N <= 256*48 = 12288

struct CtcWs {
  float* lse;     // [B][T]
  float* lp;      // [B][T][S]   label log-probs
  double* alpha;  // [B][T][S]
  double* beta;   // [B][T][S]   (beta includes lp[t][s], as alpha does)
  float* scale;   // [B]
  float* nll;     // [B]  (-log likelihood, unscaled)
  int S;          // 2L+1 for the padded L
};

__host__ __device__ inline CtcWs ctc_ws(void* ws, int B, int T, int N, int L) {
  (void)N;
  CtcWs w;
  w.S = 2 * L + 1;
  char* p = (char*)ws;
  w.lse = (float*)p; p += align_up((size_t)B * T * sizeof(float), 256);
  w.lp = (float*)p; p += align_up((size_t)B * T * w.S * sizeof(float), 256);
  w.alpha = (double*)p; p += align_up((size_t)B * T * w.S * sizeof(double), 256);
  w.beta = (double*)p; p += align_up((size_t)B * T * w.S * sizeof(double), 256);
  w.scale = (float*)p; p += align_up((size_t)B * sizeof(float), 256);
  w.nll = (float*)p;
  return w;
}

__device__ __forceinline__ float block_reduce_max(float v, float* sm) {
  v = wave_max(v);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) sm[w] = v;
  __syncthreads();
  float r = sm[0];
  for (int k = 1; k < (int)(blockDim.x >> 6); ++k) r = fmaxf(r, sm[k]);
  __syncthreads();
  return r;
}
__device__ __forceinline__ float block_reduce_sum(float v, float* sm) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) sm[w] = v;
  __syncthreads();
  float r = sm[0];
  for (int k = 1; k < (int)(blockDim.x >> 6); ++k) r += sm[k];
  __syncthreads();
  return r;
}

// Row loader: elements [0,N) of a row that is only 4-byte aligned. Thread tid owns
// element indices  head: tid (< nh);  body: nh + 4*(tid + 256*k) .. +3;  tail scalars.
// nh = number of leading scalars so that the body is 16-byte aligned.
struct RowSplit { int nh, nbody4, ntail; };
__device__ __forceinline__ RowSplit row_split(const float* row, int N) {
  RowSplit r;
  int mis = (int)(((uintptr_t)row >> 2) & 3);
  r.nh = mis ? 4 - mis : 0;
  if (r.nh > N) r.nh = N;
  r.nbody4 = (N - r.nh) >> 2;
  r.ntail = (N - r.nh) & 3;
  return r;
}

// lse[b][t] and the label log-probs lp[b][t][s] = x[ext_s] - lse
__global__ __launch_bounds__(kRowThreads) void ctc_rows_lse(int T, int N, int L,
                                                            const float* __restrict__ x,
                                                            const int* __restrict__ target,
                                                            const int* __restrict__ targetSize,
                                                            CtcWs ws) {
  __shared__ float sm[8];
  const size_t r = blockIdx.x;  // row = b*T + t
  const int b = (int)(r / T);
  const float* row = x + r * N;
  const int tid = threadIdx.x;
  RowSplit sp = row_split(row, N);
  const float4* body = (const float4*)(row + sp.nh);

  float4 v[kRowMaxPer / 4];
  float m = -INFINITY;
#pragma unroll
  for (int k = 0; k < kRowMaxPer / 4; ++k) {
    int idx = tid + kRowThreads * k;
    if (idx < sp.nbody4) {
      v[k] = body[idx];
      m = fmaxf(m, fmaxf(fmaxf(v[k].x, v[k].y), fmaxf(v[k].z, v[k].w)));
    }
  }
  float hv = -INFINITY;
  if (tid < sp.nh) hv = row[tid];
  else if (tid >= 64 && tid - 64 < sp.ntail) hv = row[sp.nh + 4 * sp.nbody4 + (tid - 64)];
  m = fmaxf(m, hv);
  m = block_reduce_max(m, sm);
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < kRowMaxPer / 4; ++k) {
    int idx = tid + kRowThreads * k;
    if (idx < sp.nbody4)
      s += (__expf(v[k].x - m) + __expf(v[k].y - m)) + (__expf(v[k].z - m) + __expf(v[k].w - m));
  }
  if (hv != -INFINITY) s += __expf(hv - m);
  s = block_reduce_sum(s, sm);
  const float lse = m + __logf(s);
  if (tid == 0) ws.lse[r] = lse;
  // gather label log-probs
  const int Lb = targetSize[b];
  const int S = 2 * Lb + 1;
  const int* y = target + (size_t)b * L;
  float* lp = ws.lp + r * ws.S;
  for (int si = tid; si < S; si += kRowThreads) {
    int lab = (si & 1) ? y[si >> 1] : (N - 1);
    lp[si] = row[lab] - lse;
  }
}

// generic-N fallback (N > 256*kRowMaxPer): two passes over the row
__global__ __launch_bounds__(kRowThreads) void ctc_rows_lse_big(int T, int N, int L,
                                                                const float* __restrict__ x,
                                                                const int* __restrict__ target,
                                                                const int* __restrict__ targetSize,
                                                                CtcWs ws) {
  __shared__ float sm[8];
  const size_t r = blockIdx.x;
  const int b = (int)(r / T);
  const float* row = x + r * N;
  const int tid = threadIdx.x;
  float m = -INFINITY;
  for (int n = tid; n < N; n += kRowThreads) m = fmaxf(m, row[n]);
  m = block_reduce_max(m, sm);
  float s = 0.f;
  for (int n = tid; n < N; n += kRowThreads) s += __expf(row[n] - m);
  s = block_reduce_sum(s, sm);
  const float lse = m + __logf(s);
  if (tid == 0) ws.lse[r] = lse;
  const int Lb = targetSize[b];
  const int S = 2 * Lb + 1;
  const int* y = target + (size_t)b * L;
  float* lp = ws.lp + r * ws.S;
  for (int si = tid; si < S; si += kRowThreads) {
    int lab = (si & 1) ? y[si >> 1] : (N - 1);
    lp[si] = row[lab] - lse;
  }
}

// log-sum-exp of three fp64 lattice values with fp32 corrections.  The correction terms use the FULL-precision fp32 expf /
// logf (ocml, < 1 ulp, unbiased), not the 1-ulp hardware v_exp_f32 / v_log_f32 the ASG scans use: the hardware functions
// round faithfully, not to nearest, and their ~1e-7 per-step bias adds up LINEARLY over a long scan -- at T = 700 frames and
// 300 labels the occupancies were off by 1e-4 (run j1).  The two scans of an utterance now run side by side with a deep
// prefetch, so the few extra instructions per step are affordable.
__device__ __forceinline__ double lse3(double a, double b, double c) {
  double m = fmax(a, fmax(b, c));
  if (m == -INFINITY) return m;
  float s = expf((float)(a - m)) + expf((float)(b - m)) + expf((float)(c - m));
  return m + (double)logf(s);   // s in [1, 3]
}

// alpha OR beta over the extended label sequence (blockIdx.y = 0: alpha, 1: beta): one wavefront per (utterance,
// direction), positions blocked over lanes (P per lane).  The two scans of an utterance are independent, so they run
// as two waves side by side (they used to be one wave doing alpha, then beta + occupancies: 364 us at T' = 188,
// 1.9 us per frame, memory-latency bound on a 4-step prefetch that every store of the scan drained).  Each scan now
// reads ONE array (the label log-probs lp[t][s]) through a register double buffer D steps deep (2 D P floats = 128
// registers whatever P is) and writes its lattice row (fp64) fire-and-forget; the occupancies
// gamma[t][s] = exp(alpha + beta - lp - logZ) have no dependence between frames and are taken in ctc_rows_grad.
template <int P, int D>
__global__ __launch_bounds__(64) void ctc_scan(int T, int N, int L, int scaleMode,
                                               const int* __restrict__ target,
                                               const int* __restrict__ targetSize,
                                               float* __restrict__ loss, CtcWs ws) {
  const int b = blockIdx.x;
  const bool isBeta = blockIdx.y == 1;
  const int lane = threadIdx.x;
  const int Lb = targetSize[b];
  const int S = 2 * Lb + 1;
  const int SW = ws.S;
  const int* y = target + (size_t)b * L;
  const float* lp = ws.lp + (size_t)b * T * SW;
  const double NEG = -INFINITY;

  if (!isBeta) {
    double* al = ws.alpha + (size_t)b * T * SW;
    bool skipPrev[P];  // may come from s-2
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int si = lane * P + p;
      const int e0 = (si & 1) ? ((si >> 1) < Lb ? y[si >> 1] : -1) : (N - 1);
      const int em2 = (si >= 2 && (si & 1) && ((si - 2) >> 1) < Lb) ? y[(si - 2) >> 1] : -2;
      skipPrev[p] = (si < S) && (si & 1) && si >= 2 && e0 != em2;
    }
    double a[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int si = lane * P + p;
      a[p] = (si < S && si < 2) ? (double)lp[si] : NEG;
      if (si < S) al[si] = a[p];
    }
    float lc[D][P], ln[D][P];
#pragma unroll
    for (int u = 0; u < D; ++u)
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const int si = lane * P + p, t = 1 + u;
        lc[u][p] = (t < T && si < S) ? lp[(size_t)t * SW + si] : 0.f;
      }
    for (int t0 = 1; t0 < T; t0 += D) {
#pragma unroll
      for (int u = 0; u < D; ++u)
#pragma unroll
        for (int p = 0; p < P; ++p) {
          const int si = lane * P + p, t = t0 + D + u;
          ln[u][p] = (t < T && si < S) ? lp[(size_t)t * SW + si] : 0.f;
        }
#pragma unroll
      for (int u = 0; u < D; ++u) {
        const int t = t0 + u;
        if (t < T) {
          const double c1 = lane_shift_up_dpp(a[P - 1], NEG);                                    // alpha[lane*P - 1]
          const double c2 = P >= 2 ? lane_shift_up_dpp(a[P >= 2 ? P - 2 : 0], NEG) : lane_shift_up_dpp(c1, NEG);  // alpha[lane*P - 2]
          double pm1 = c1, pm2 = c2;
          double* alt = al + (size_t)t * SW;
#pragma unroll
          for (int p = 0; p < P; ++p) {
            const int si = lane * P + p;
            const double cur = a[p];
            const double v = lse3(cur, pm1, skipPrev[p] ? pm2 : NEG);
            double na = NEG;
            if (si < S && v != NEG) na = v + (double)lc[u][p];
            if (si < S) alt[si] = na;
            pm2 = pm1;
            pm1 = cur;
            a[p] = na;
          }
        }
      }
#pragma unroll
      for (int u = 0; u < D; ++u)
#pragma unroll
        for (int p = 0; p < P; ++p) lc[u][p] = ln[u][p];
    }
    // log-likelihood = lse(alpha[T-1][S-1], alpha[T-1][S-2])
    double v1 = NEG, v2 = NEG;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int si = lane * P + p;
      if (si == S - 1) v1 = a[p];
      if (si == S - 2) v2 = a[p];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      v1 = fmax(v1, __shfl_xor(v1, off));
      v2 = fmax(v2, __shfl_xor(v2, off));
    }
    const double ll = lse3(v1, v2, NEG);
    if (lane == 0) {
      const float sc = scale_of(scaleMode, T, Lb);
      loss[b] = (float)(-(double)sc * ll);
      ws.scale[b] = sc;
      ws.nll[b] = (float)(-ll);
    }
    return;
  }

  // ---- beta[t][s] (includes lp[t][s], like alpha): beta[T-1][s] = lp[T-1][s] for s >= S-2,
  //      beta[t-1][s] = lse(beta[t][s], beta[t][s+1], beta[t][s+2] if allowed) + lp[t-1][s]
  double* bt = ws.beta + (size_t)b * T * SW;
  bool skipNext[P];  // may go to s+2
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const int si = lane * P + p;
    const int e0 = (si & 1) ? ((si >> 1) < Lb ? y[si >> 1] : -1) : (N - 1);
    const int ep2 = ((si & 1) && si + 2 < S) ? y[(si + 2) >> 1] : -2;
    skipNext[p] = (si < S) && (si & 1) && si + 2 < S && e0 != ep2;
  }
  double be[P];
  {
    const float* lpt = lp + (size_t)(T - 1) * SW;
    double* btt = bt + (size_t)(T - 1) * SW;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int si = lane * P + p;
      be[p] = (si < S && si >= S - 2) ? (double)lpt[si] : NEG;
      if (si < S) btt[si] = be[p];
    }
  }
  float lc[D][P], ln[D][P];  // lc[u] = lp[thi - 1 - u]
#pragma unroll
  for (int u = 0; u < D; ++u)
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int si = lane * P + p, t = T - 2 - u;
      lc[u][p] = (t >= 0 && si < S) ? lp[(size_t)t * SW + si] : 0.f;
    }
  for (int thi = T - 1; thi >= 1; thi -= D) {
#pragma unroll
    for (int u = 0; u < D; ++u)
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const int si = lane * P + p, t = thi - 1 - D - u;
        ln[u][p] = (t >= 0 && si < S) ? lp[(size_t)t * SW + si] : 0.f;
      }
#pragma unroll
    for (int u = 0; u < D; ++u) {
      const int t = thi - u;  // beta_t -> beta_{t-1}
      if (t >= 1) {
        const double n1 = lane_shift_down_dpp(be[0], NEG);                                        // beta[(lane+1)*P]
        const double n2 = P >= 2 ? lane_shift_down_dpp(be[P >= 2 ? 1 : 0], NEG) : lane_shift_down_dpp(n1, NEG);  // beta[(lane+1)*P + 1]
        double nb[P];
        double* btt = bt + (size_t)(t - 1) * SW;
#pragma unroll
        for (int p = P - 1; p >= 0; --p) {
          const int si = lane * P + p;
          const double b1 = (p + 1 < P) ? be[p + 1 < P ? p + 1 : 0] : n1;
          const double b2 = (p + 2 < P) ? be[p + 2 < P ? p + 2 : 0] : ((p + 1 < P) ? n1 : n2);
          const double v = lse3(be[p], b1, skipNext[p] ? b2 : NEG);
          nb[p] = (si < S && v != NEG) ? v + (double)lc[u][p] : NEG;
          if (si < S) btt[si] = nb[p];
        }
#pragma unroll
        for (int p = 0; p < P; ++p) be[p] = nb[p];
      }
    }
#pragma unroll
    for (int u = 0; u < D; ++u)
#pragma unroll
      for (int p = 0; p < P; ++p) lc[u][p] = ln[u][p];
  }
}

// grad row = g * softmax(x); then subtract g * gamma at the frame's labels
__global__ __launch_bounds__(kRowThreads) void ctc_rows_grad(int T, int N, int L,
                                                             const float* __restrict__ x,
                                                             const int* __restrict__ target,
                                                             const int* __restrict__ targetSize,
                                                             const float* __restrict__ grad,
                                                             float* __restrict__ dx, CtcWs ws) {
  const size_t r = blockIdx.x;
  const int b = (int)(r / T);
  const float* row = x + r * N;
  float* out = dx + r * N;
  const int tid = threadIdx.x;
  const float g = ws.scale[b] * grad[b];
  const float lse = ws.lse[r];
  RowSplit sp = row_split(row, N);  // dx has the same alignment as x modulo 16 B iff bases agree
  const bool same = ((((uintptr_t)row) ^ ((uintptr_t)out)) & 15) == 0;
  if (same) {
    const float4* body = (const float4*)(row + sp.nh);
    float4* obody = (float4*)(out + sp.nh);
    for (int idx = tid; idx < sp.nbody4; idx += kRowThreads) {
      float4 v = body[idx];
      v.x = g * __expf(v.x - lse); v.y = g * __expf(v.y - lse);
      v.z = g * __expf(v.z - lse); v.w = g * __expf(v.w - lse);
      obody[idx] = v;
    }
    if (tid < sp.nh) out[tid] = g * __expf(row[tid] - lse);
    else if (tid >= 64 && tid - 64 < sp.ntail) {
      int n = sp.nh + 4 * sp.nbody4 + (tid - 64);
      out[n] = g * __expf(row[n] - lse);
    }
  } else {
    for (int n = tid; n < N; n += kRowThreads) out[n] = g * __expf(row[n] - lse);
  }
  __syncthreads();  // drains the row stores (vmcnt(0)) before the label fix-up
  const int Lb = targetSize[b];
  const int S = 2 * Lb + 1;
  const int* y = target + (size_t)b * L;
  // occupancy of label position s at this frame: exp(alpha + beta - lp - logZ) (alpha and beta both carry lp[t][s])
  const float* lpr = ws.lp + r * ws.S;
  const double* alr = ws.alpha + r * ws.S;
  const double* ber = ws.beta + r * ws.S;
  const float nll = ws.nll[b];
  if (nll != INFINITY) {   // an infeasible target (logZ = -inf) has no occupancy: its loss is +inf, its gradient g * softmax
    const double ll = -(double)nll;
    for (int si = tid; si < S; si += kRowThreads) {
      const int lab = (si & 1) ? y[si >> 1] : (N - 1);
      const double av = alr[si], bv = ber[si];
      if (av != -INFINITY && bv != -INFINITY) {
        const float v = __expf((float)(av + bv - (double)lpr[si] - ll));
        if (v != 0.f) atomicAdd(&out[lab], -g * v);
      }
    }
  }
}

__global__ __launch_bounds__(kRowThreads) void ctc_rows_argmax(int N, const float* __restrict__ x,
                                                               int* __restrict__ path) {
  __shared__ float smv[4];
  __shared__ int smi[4];
  const size_t r = blockIdx.x;
  const float* row = x + r * N;
  const int tid = threadIdx.x;
  float best = -INFINITY;
  int arg = 0x7fffffff;
  for (int n = tid; n < N; n += kRowThreads) {
    float v = row[n];
    if (v > best) { best = v; arg = n; }  // ascending n per thread: first max kept
  }
  // wave: max value, then smallest index among lanes holding it
  float m = wave_max(best);
  int cand = (best == m) ? arg : 0x7fffffff;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) cand = min(cand, __shfl_xor(cand, off));
  if ((tid & 63) == 0) { smv[tid >> 6] = m; smi[tid >> 6] = cand; }
  __syncthreads();
  if (tid == 0) {
    float bm = smv[0];
    int bi = smi[0];
    for (int k = 1; k < kRowThreads / 64; ++k) {
      if (smv[k] > bm || (smv[k] == bm && smi[k] < bi)) { bm = smv[k]; bi = smi[k]; }
    }
    path[r] = bi;
  }
}

// one wavefront per utterance (a thread per utterance walked the L labels with dependent, uncoalesced loads: 49 us at
// L = 300 in front of every criterion call -- profiles/r02_run16_asg_timeline.log)
__global__ __launch_bounds__(64) void batch_target_size_k(int B, int L, int maxSize, const int* __restrict__ target,
                                                          int* __restrict__ targetSize, int ctc) {
  const int b = blockIdx.x, lane = threadIdx.x;
  if (b >= B) return;
  const int* y = target + (size_t)b * L;
  int first = L;   // index of the first negative label
  for (int i = lane; i < L; i += 64)
    if (y[i] < 0) { first = i; break; }
  for (int off = 32; off > 0; off >>= 1) first = min(first, __shfl_xor(first, off));
  const int n = first;
  if (!ctc) {
    if (lane == 0) targetSize[b] = n < maxSize ? n : maxSize;
    return;
  }
  int R = 0;
  for (int i = 1 + lane; i < n; i += 64) R += (y[i] == y[i - 1]);
  for (int off = 32; off > 0; off >>= 1) R += __shfl_xor(R, off);
  if (lane == 0) {
    const int m = (n + R < maxSize ? n + R : maxSize) - R;
    targetSize[b] = m < 0 ? 0 : m;
  }
}

}  // namespace w2l

using namespace w2l;

W2L_API int w2l_batch_target_size(int B, int L, int maxSize, const int* target, int* targetSize,
                                  w2l_stream_t stream) {
  if (B <= 0 || L <= 0 || !target || !targetSize) return W2L_EINVAL;
  hipLaunchKernelGGL(batch_target_size_k, dim3(B), dim3(64), 0, (hipStream_t)stream, B, L, maxSize, target, targetSize, 0);
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}

W2L_API int w2l_batch_ctc_target_size(int B, int L, int T, const int* target, int* targetSize,
                                      w2l_stream_t stream) {
  if (B <= 0 || L <= 0 || !target || !targetSize) return W2L_EINVAL;
  hipLaunchKernelGGL(batch_target_size_k, dim3(B), dim3(64), 0, (hipStream_t)stream, B, L, T, target, targetSize, 1);
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}

W2L_API size_t w2l_ctc_workspace_size(int B, int T, int N, int L) {
  if (B <= 0 || T <= 0 || N <= 0 || L < 0) return 0;
  size_t S = 2 * (size_t)L + 1;
  return align_up((size_t)B * T * sizeof(float), 256) + align_up((size_t)B * T * S * sizeof(float), 256) +
         2 * align_up((size_t)B * T * S * sizeof(double), 256) + 2 * align_up((size_t)B * sizeof(float), 256);
}

W2L_API int w2l_ctc_forward(int B, int T, int N, int L, int scaleMode, const float* input,
                            const int* target, const int* targetSize, float* loss,
                            void* workspace, w2l_stream_t stream) {
  if (B <= 0 || T <= 0 || N <= 1 || L <= 0 || !input || !target || !targetSize || !loss || !workspace)
    return W2L_EINVAL;
  if (2 * L + 1 > 64 * 32) return W2L_EUNSUPPORTED;   // L <= 1023 label positions per utterance (32 per lane)
  hipStream_t s = (hipStream_t)stream;
  CtcWs ws = ctc_ws(workspace, B, T, N, L);
  const unsigned rows = (unsigned)((size_t)B * T);
  if (N <= kRowThreads * kRowMaxPer)
    hipLaunchKernelGGL(ctc_rows_lse, dim3(rows), dim3(kRowThreads), 0, s, T, N, L, input, target, targetSize, ws);
  else
    hipLaunchKernelGGL(ctc_rows_lse_big, dim3(rows), dim3(kRowThreads), 0, s, T, N, L, input, target, targetSize, ws);
  W2L_LAUNCH_CHECK();
  const int S = 2 * L + 1;
  const dim3 grid((unsigned)B, 2), blk(64);   // (utterance, alpha | beta)
  if (S <= 64) hipLaunchKernelGGL((ctc_scan<1, 32>), grid, blk, 0, s, T, N, L, scaleMode, target, targetSize, loss, ws);
  else if (S <= 128) hipLaunchKernelGGL((ctc_scan<2, 32>), grid, blk, 0, s, T, N, L, scaleMode, target, targetSize, loss, ws);
  else if (S <= 256) hipLaunchKernelGGL((ctc_scan<4, 16>), grid, blk, 0, s, T, N, L, scaleMode, target, targetSize, loss, ws);
  else if (S <= 512) hipLaunchKernelGGL((ctc_scan<8, 8>), grid, blk, 0, s, T, N, L, scaleMode, target, targetSize, loss, ws);
  else if (S <= 1024) hipLaunchKernelGGL((ctc_scan<16, 4>), grid, blk, 0, s, T, N, L, scaleMode, target, targetSize, loss, ws);
  else hipLaunchKernelGGL((ctc_scan<32, 2>), grid, blk, 0, s, T, N, L, scaleMode, target, targetSize, loss, ws);
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}

W2L_API int w2l_ctc_backward(int B, int T, int N, int L, const float* input, const int* target,
                             const int* targetSize, const float* grad, float* inputGrad,
                             void* workspace, w2l_stream_t stream) {
  if (B <= 0 || T <= 0 || N <= 1 || L <= 0 || !input || !target || !targetSize || !grad || !inputGrad || !workspace)
    return W2L_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  CtcWs ws = ctc_ws(workspace, B, T, N, L);
  const unsigned rows = (unsigned)((size_t)B * T);
  hipLaunchKernelGGL(ctc_rows_grad, dim3(rows), dim3(kRowThreads), 0, s, T, N, L, input, target, targetSize, grad, inputGrad, ws);
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}

W2L_API int w2l_ctc_viterbi(int B, int T, int N, const float* input, int* path, w2l_stream_t stream) {
  if (B <= 0 || T <= 0 || N <= 0 || !input || !path) return W2L_EINVAL;
  hipLaunchKernelGGL(ctc_rows_argmax, dim3((unsigned)((size_t)B * T)), dim3(kRowThreads), 0, (hipStream_t)stream, N, input, path);
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}
