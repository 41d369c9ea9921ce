// criterion_linseg.hip -- LinearSegmentationCriterion (LinSegCriterion) support kernels (gfx950).
//
// Reference: every ASG recipe trains its first --linseg updates with `LinSegCriterion(numClasses, scalemode)`
// sharing the ASG transition parameter (recipes/slimIPL/src/Train.cpp:589-617, :1866-1883; --linseg=1 in
// recipes/conv_glu/librispeech/train.cfg:15).  The class lives in Flashlight [UNVENDORED]:
// LinearSegmentationCriterion::forward(input, target) = AutoSegmentationCriterion::forward(input,
// getLinearTarget(target, T)), where getLinearTarget stretches each utterance's label string over the T frames:
//     newTarget[b][t] = target[b][t * L_b / T]          (integer division; L_b = leading non-negative entries)
//     a row with L_b == 0 or L_b > T is filled with -1  ("make ASG think L == 0")
// The stretched target has length T, so ForceAlignmentCriterion on it has exactly ONE alignment (state t at frame t):
//     FAC = s_b * ( sum_t x[t][y_t] + sum_{t >= 1} trans[y_t][y_{t-1}] ),   s_b = scale(mode, T, target size T)
// The general FAC kernels keep alpha as [B][T][L] and stop at L = 512; this single-path case is a gather-sum and a
// scatter, done here.  FullConnectionCriterion is unchanged (w2l_fcc_*).
// Parity: unpinned in /root/reference (no LinSeg test there); tests/ check against the oracle's ASG on the target
// stretched by the numpy restatement of getLinearTarget (oracle/pyoracle.py).
#include <hip/hip_runtime.h>

#include "common.hpp"

namespace w2l {

__global__ __launch_bounds__(256) void linear_target_k(int L, int T, const int* __restrict__ target, int* __restrict__ lin) {
  const int b = blockIdx.x;
  const int* y = target + (size_t)b * L;
  __shared__ int firstNeg;
  if (threadIdx.x == 0) firstNeg = L;
  __syncthreads();
  for (int i = threadIdx.x; i < L; i += 256)
    if (y[i] < 0) atomicMin(&firstNeg, i);
  __syncthreads();
  const int TN = firstNeg;
  int* o = lin + (size_t)b * T;
  const bool bad = TN == 0 || TN > T;
  for (int t = threadIdx.x; t < T; t += 256) o[t] = bad ? -1 : y[(int)(((long long)t * TN) / T)];
}

// one workgroup per utterance: fp64 gather-sum over the T frames, fixed-order tree
__global__ __launch_bounds__(256) void fac_fullpath_fwd_k(int T, int N, int scaleMode, const float* __restrict__ x,
                                                          const int* __restrict__ path, const float* __restrict__ trans,
                                                          float* __restrict__ loss) {
  const int b = blockIdx.x;
  const int* y = path + (size_t)b * T;
  const float* xb = x + (size_t)b * T * N;
  __shared__ double sm[256];
  __shared__ int bad;
  if (threadIdx.x == 0) bad = 0;
  __syncthreads();
  double acc = 0.0;
  for (int t = threadIdx.x; t < T; t += 256) {
    const int c = y[t];
    if (c < 0 || c >= N) { bad = 1; continue; }
    acc += (double)xb[(size_t)t * N + c];
    if (t > 0) {
      const int p = y[t - 1];
      if (p >= 0 && p < N) acc += (double)trans[(size_t)c * N + p];
    }
  }
  sm[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[b] = bad ? 0.f : scale_of(scaleMode, T, T) * (float)sm[0];
}

// inputGrad[b][t][:] = 0 except [y_t] = s_b * grad[b]; rows of a -1 path are all zero
__global__ __launch_bounds__(256) void fac_fullpath_dx_k(int T, int N, int scaleMode, const int* __restrict__ path,
                                                         const float* __restrict__ grad, float* __restrict__ dx) {
  const int b = blockIdx.y;
  const size_t row = (size_t)b * T + blockIdx.x;
  const int c = path[row];
  const bool ok = path[(size_t)b * T] >= 0;
  const float v = ok ? scale_of(scaleMode, T, T) * grad[b] : 0.f;
  float* o = dx + row * N;
  for (int n = threadIdx.x; n < N; n += 256) o[n] = (n == c) ? v : 0.f;
}

// transGrad[i][j] = sum_b s_b grad[b] * #{t >= 1 : y_t = i, y_{t-1} = j}: one thread per entry, utterances in order
// (deterministic); meant for letter-sized N (ASG token sets, N ~ 30)
__global__ __launch_bounds__(256) void fac_fullpath_dtrans_gather_k(int B, int T, int N, int scaleMode,
                                                                    const int* __restrict__ path,
                                                                    const float* __restrict__ grad, float* __restrict__ dt) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= N * N) return;
  const int i = e / N, j = e - i * N;
  float acc = 0.f;
  for (int b = 0; b < B; ++b) {
    const int* y = path + (size_t)b * T;
    if (y[0] < 0) continue;
    int cnt = 0;
    int prev = y[0];
    for (int t = 1; t < T; ++t) {
      const int c = y[t];
      cnt += (c == i && prev == j) ? 1 : 0;
      prev = c;
    }
    acc += scale_of(scaleMode, T, T) * grad[b] * (float)cnt;
  }
  dt[e] = acc;
}

// large N: scatter with float atomics into a zeroed matrix (order of the adds is not fixed)
__global__ __launch_bounds__(256) void fac_fullpath_dtrans_scatter_k(int T, int N, int scaleMode, const int* __restrict__ path,
                                                                     const float* __restrict__ grad, float* __restrict__ dt) {
  const int b = blockIdx.y;
  const int t = blockIdx.x * 256 + threadIdx.x + 1;
  const int* y = path + (size_t)b * T;
  if (t >= T || y[0] < 0) return;
  const int c = y[t], p = y[t - 1];
  if (c < 0 || c >= N || p < 0 || p >= N) return;
  atomicAdd(&dt[(size_t)c * N + p], scale_of(scaleMode, T, T) * grad[b]);
}

}  // namespace w2l

using namespace w2l;

W2L_API int w2l_linear_target(int B, int L, int T, const int* target, int* linTarget, w2l_stream_t stream) {
  if (B <= 0 || L <= 0 || T <= 0 || !target || !linTarget) return W2L_EINVAL;
  hipLaunchKernelGGL(linear_target_k, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, L, T, target, linTarget);
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}

W2L_API int w2l_fac_fullpath_forward(int B, int T, int N, int scaleMode, const float* input, const int* path,
                                     const float* trans, float* loss, w2l_stream_t stream) {
  if (B <= 0 || T <= 0 || N <= 0 || !input || !path || !trans || !loss) return W2L_EINVAL;
  hipLaunchKernelGGL(fac_fullpath_fwd_k, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, T, N, scaleMode, input, path,
                     trans, loss);
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}

W2L_API int w2l_fac_fullpath_backward(int B, int T, int N, int scaleMode, const int* path, const float* grad,
                                      float* inputGrad, float* transGrad, w2l_stream_t stream) {
  if (B <= 0 || T <= 0 || N <= 0 || !path || !grad || !inputGrad || !transGrad) return W2L_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(fac_fullpath_dx_k, dim3((unsigned)T, (unsigned)B), dim3(256), 0, s, T, N, scaleMode, path, grad, inputGrad);
  W2L_LAUNCH_CHECK();
  if (N <= 64) {
    hipLaunchKernelGGL(fac_fullpath_dtrans_gather_k, dim3((unsigned)((N * N + 255) / 256)), dim3(256), 0, s, B, T, N, scaleMode,
                       path, grad, transGrad);
  } else {
    W2L_HIP_CHECK(hipMemsetAsync(transGrad, 0, (size_t)N * N * sizeof(float), s));
    hipLaunchKernelGGL(fac_fullpath_dtrans_scatter_k, dim3((unsigned)((T + 254) / 255), (unsigned)B), dim3(256), 0, s, T, N,
                       scaleMode, path, grad, transGrad);
  }
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}
