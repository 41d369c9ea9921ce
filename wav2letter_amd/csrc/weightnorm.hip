// weightnorm.hip -- fl::WeightNorm reparameterisation w = v * g / ||v|| and its backward,
// plus SpecAugment masking.  Reference: WN token (recipes/joint_training_vox_populi/cpc/
// SequentialBuilder.cpp:379-386), parameter order v, g, bias (recipes/utilities/
// convlm_serializer/Utils.cpp:112-143); SAUG token (SequentialBuilder.cpp:602-613).
// Both internal weight layouts ([kw*Cin][Cout] for "WN 3 C", [in][out] for "WN 0 L") are
// [K][Nout] matrices normalised per column, so one kernel pair serves both.
#include "common.hpp"
#include "gemm.hpp"  // sk_scratch: the shared per-stream scratch

namespace w2l {

// per-column sums of squares / dot products over the K rows of [K][N] matrices, two deterministic passes (row-block
// partials in the stream scratch, then a fixed-order sum): the one-pass version ran N/64 <= 29 workgroups for the 174 MB
// weight of the last conv_glu layer (2.6 ms a call, 35 ms per C4 step; profiles/r01_run49_c4_kernel_stats.csv)
constexpr int kWnMaxParts = 128;

template <int V>
__global__ __launch_bounds__(256) void wn_colreduce_partial_k(const float* __restrict__ a, const float* __restrict__ b,
                                                              float* __restrict__ partial, int K, int N, int rowsPerBlock) {
  typedef float vec_t __attribute__((ext_vector_type(V)));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = (blockIdx.x * 64 + lane) * V;
  __shared__ float sm[4][64 * V];
  const int k0 = blockIdx.y * rowsPerBlock;
  int k1 = k0 + rowsPerBlock;
  if (k1 > K) k1 = K;
  vec_t acc0 = 0.f, acc1 = 0.f;
  if (n < N) {
    int k = k0 + wave;
    for (; k + 4 < k1; k += 8) {
      const vec_t a0 = *(const vec_t*)(a + (size_t)k * N + n), b0 = *(const vec_t*)(b + (size_t)k * N + n);
      const vec_t a1 = *(const vec_t*)(a + (size_t)(k + 4) * N + n), b1 = *(const vec_t*)(b + (size_t)(k + 4) * N + n);
      acc0 += a0 * b0; acc1 += a1 * b1;
    }
    for (; k < k1; k += 4) acc0 += *(const vec_t*)(a + (size_t)k * N + n) * *(const vec_t*)(b + (size_t)k * N + n);
  }
  const vec_t acc = acc0 + acc1;
#pragma unroll
  for (int v = 0; v < V; ++v) sm[wave][lane * V + v] = acc[v];
  __syncthreads();
  for (int c = threadIdx.x; c < 64 * V; c += 256) {
    const int nn = blockIdx.x * 64 * V + c;
    if (nn < N) partial[(size_t)blockIdx.y * N + nn] = (sm[0][c] + sm[1][c]) + (sm[2][c] + sm[3][c]);
  }
}

__global__ __launch_bounds__(256) void wn_colreduce_finish_k(const float* __restrict__ partial, float* __restrict__ out, int parts,
                                                             int N, int sqrtOut) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + lane;
  __shared__ float sm[4][64];
  float s = 0.f;
  if (n < N)
    for (int p = wave; p < parts; p += 4) s += partial[(size_t)p * N + n];
  sm[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && n < N) {
    const float t = (sm[0][lane] + sm[1][lane]) + (sm[2][lane] + sm[3][lane]);
    out[n] = sqrtOut ? sqrtf(t) : t;
  }
}

static int wn_colreduce(const float* a, const float* b, float* out, int K, int N, int sqrtOut, hipStream_t s) {
  int rowsPerBlock = (K + kWnMaxParts - 1) / kWnMaxParts;
  if (rowsPerBlock < 16) rowsPerBlock = 16;
  const int parts = (K + rowsPerBlock - 1) / rowsPerBlock;
  float* partial = sk_scratch(s, kSkScratchBytes);
  if (!partial || (size_t)parts * N * sizeof(float) > kSkScratchBytes) return W2L_EHIP;
  const bool al = ((((uintptr_t)a) | ((uintptr_t)b)) & 15) == 0 && N % 4 == 0;
  if (al) {
    dim3 grid((unsigned)((N + 255) / 256), (unsigned)parts);
    hipLaunchKernelGGL(wn_colreduce_partial_k<4>, grid, dim3(256), 0, s, a, b, partial, K, N, rowsPerBlock);
  } else {
    dim3 grid((unsigned)((N + 63) / 64), (unsigned)parts);
    hipLaunchKernelGGL(wn_colreduce_partial_k<1>, grid, dim3(256), 0, s, a, b, partial, K, N, rowsPerBlock);
  }
  W2L_LAUNCH_CHECK();
  hipLaunchKernelGGL(wn_colreduce_finish_k, dim3((unsigned)((N + 63) / 64)), dim3(256), 0, s, partial, out, parts, N, sqrtOut);
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}

__global__ __launch_bounds__(256) void wn_apply_k(const float* __restrict__ v, const float* __restrict__ g,
                                                  const float* __restrict__ norm, float* __restrict__ w,
                                                  size_t total, int N) {
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    int n = (int)(e % N);
    w[e] = v[e] * (g[n] / norm[n]);
  }
}

// dv = g/norm * (dw - v * dot/norm^2), dg = dot/norm   (dot = sum_k v*dw per column)
__global__ __launch_bounds__(256) void wn_bwd_k(const float* __restrict__ v, const float* __restrict__ g,
                                                const float* __restrict__ norm, const float* __restrict__ dot,
                                                const float* __restrict__ dw, float* __restrict__ dv,
                                                float* __restrict__ dg, size_t total, int N) {
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    int n = (int)(e % N);
    float nn = norm[n];
    dv[e] = g[n] / nn * (dw[e] - v[e] * dot[n] / (nn * nn));
    if (e < (size_t)N) dg[e] = dot[e] / norm[e];
  }
}

// fl::SpecAugment (SAUG token; recipes/sota/2019/am_arch/am_tds_ctc.arch:1, builder
// recipes/joint_training_vox_populi/cpc/SequentialBuilder.cpp:602-613) on frame-major features x[B][T][F], fill 0.
// Semantics as recalled from the un-vendored fl/contrib/modules/SpecAugment.cpp (restated in oracle/nn_oracle.c):
// ONE set of masks per batch (af::span over the batch dim), widths drawn from [0, fMaskF) / [0, min(tMaskT, T*p)),
// af::seq ends inclusive (a draw of f masks f + 1 rows), no time warping.  Draw k = hash32(k, seed, 1000) mod range.
__global__ __launch_bounds__(256) void specaug_k(float* __restrict__ x, int T, int F, int fMaskF, int nFMask,
                                                 int tMaskT, float tMaskP, int nTMask, uint32_t seed) {
  const int b = blockIdx.y;
  __shared__ int f0s[8], f1s[8], t0s[8], t1s[8];
  if (threadIdx.x < 8) {
    int k = threadIdx.x;
    f0s[k] = t0s[k] = 0;
    f1s[k] = t1s[k] = -1;
    if (k < nFMask && fMaskF > 0) {
      int fw = (int)(hash32(4 * k, seed, 1000) % (uint32_t)fMaskF);
      int f0 = (int)(hash32(4 * k + 1, seed, 1000) % (uint32_t)(F - fw));
      f0s[k] = f0; f1s[k] = f0 + fw;
    }
    int tmax = (int)((float)T * tMaskP);
    if (tmax > tMaskT) tmax = tMaskT;
    if (tmax > T) tmax = T;
    if (k < nTMask && tmax > 0) {
      int tw = (int)(hash32(4 * k + 2, seed, 1000) % (uint32_t)tmax);
      int t0 = (int)(hash32(4 * k + 3, seed, 1000) % (uint32_t)(T - tw));
      t0s[k] = t0; t1s[k] = t0 + tw;
    }
  }
  __syncthreads();
  float* xb = x + (size_t)b * T * F;
  const size_t n = (size_t)T * F;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
    int t = (int)(e / F), f = (int)(e - (size_t)t * F);
    bool m = false;
#pragma unroll
    for (int k = 0; k < 8; ++k) m = m || (f >= f0s[k] && f <= f1s[k]) || (t >= t0s[k] && t <= t1s[k]);
    if (m) xb[e] = 0.f;
  }
}

}  // namespace w2l

using namespace w2l;

// w[K][N] = v * g / ||v||_col ; norm[N] kept for backward
W2L_API int w2l_weightnorm_forward(const float* v, const float* g, float* w, float* norm, int K, int N,
                                   w2l_stream_t stream) {
  if (!v || !g || !w || !norm || K <= 0 || N <= 0) return W2L_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  { const int st = wn_colreduce(v, v, norm, K, N, 1, s); if (st) return st; }
  size_t total = (size_t)K * N;
  unsigned grid = (unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(wn_apply_k, dim3(grid), dim3(256), 0, s, v, g, norm, w, total, N);
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}

// dot[N] is scratch
W2L_API int w2l_weightnorm_backward(const float* v, const float* g, const float* norm, const float* dw,
                                    float* dv, float* dg, float* dot, int K, int N, w2l_stream_t stream) {
  if (!v || !g || !norm || !dw || !dv || !dg || !dot || K <= 0 || N <= 0) return W2L_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  { const int st = wn_colreduce(v, dw, dot, K, N, 0, s); if (st) return st; }
  size_t total = (size_t)K * N;
  unsigned grid = (unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(wn_bwd_k, dim3(grid), dim3(256), 0, s, v, g, norm, dot, dw, dv, dg, total, N);
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}

W2L_API int w2l_specaugment_inplace(float* x, int B, int T, int F, int fMaskF, int nFMask, int tMaskT,
                                    float tMaskP, int nTMask, uint32_t seed, w2l_stream_t stream) {
  if (!x || B <= 0 || T <= 0 || F <= 0 || nFMask > 8 || nTMask > 8 || nFMask < 0 || nTMask < 0 || fMaskF < 0 || tMaskT < 0 ||
      F < fMaskF /* the reference throws: "Invalid input frequency channels" */) return W2L_EINVAL;
  size_t n = (size_t)T * F;
  unsigned gx = (unsigned)((n + 255) / 256 > 256 ? 256 : (n + 255) / 256);
  hipLaunchKernelGGL(specaug_k, dim3(gx, (unsigned)B), dim3(256), 0, (hipStream_t)stream, x, T, F, fMaskF, nFMask,
                     tMaskT, tMaskP, nTMask, seed);
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}
