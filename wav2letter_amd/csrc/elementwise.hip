// elementwise.hip -- HBM-bound glue of the TDS / conv_glu stacks, fused so that
// every activation tensor is read and written as few times as the data flow allows:
//   residual + dropout + LayerNorm statistics in one pass, LayerNorm apply,
//   LayerNorm backward (reduce + apply, with the ReLU/dropout mask of the producer
//   folded into the apply), GLU forward/backward, layout transposes, SGD.
// Reference modules: fl::LayerNorm / fl::ReLU / fl::Dropout / fl::GatedLinearUnit /
// fl::Reorder (arch grammar: recipes/joint_training_vox_populi/cpc/SequentialBuilder.cpp
// :358-377, :423-428, :388-394, :467-473, :106-119); TDS data flow:
// recipes/streaming_convnets/inference/inference/module/nn/TDSBlock.cpp:58-70.
// All kernels are float4-vectorised grid-stride loops (coalesced 16 B/lane).
#include "common.hpp"
#include "gemm.hpp"  // sk_scratch: the shared per-stream scratch (sumsq partials)

namespace w2l {

constexpr int kEwThreads = 256;

static inline unsigned ew_grid(size_t n4) {
  size_t g = (n4 + kEwThreads - 1) / kEwThreads;
  if (g > 256 * 16) g = 256 * 16;
  if (g < 1) g = 1;
  return (unsigned)g;
}

constexpr int kLnMaxParts = 64;   // partial sums per group (utterance-sized groups)
constexpr size_t kLnSmall = 8192;  // groups of at most this many floats: one block, one pass

// Deterministic two-level reduction: every block leaves ONE (a, b) partial in
// part[2*(g*gridDim.x + blockIdx.x)]; consumers add the gridDim.x partials of their group in
// index order (no atomics: 32 groups x thousands of waves on 64 addresses was the old cost).
__device__ __forceinline__ void block_partial2(double a, double b, double* part) {
  __shared__ double sm[2][kEwThreads / 64];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    a += __shfl_xor(a, off);
    b += __shfl_xor(b, off);
  }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { sm[0][w] = a; sm[1][w] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double sa = 0, sb = 0;
#pragma unroll
    for (int i = 0; i < kEwThreads / 64; ++i) { sa += sm[0][i]; sb += sm[1][i]; }
    part[0] = sa;
    part[1] = sb;
  }
}
__device__ __forceinline__ void sum_parts(const double* __restrict__ part, int g, int nparts, double& a, double& b) {
  a = 0; b = 0;
  for (int i = 0; i < nparts; ++i) { a += part[2 * ((size_t)g * nparts + i)]; b += part[2 * ((size_t)g * nparts + i) + 1]; }
}

// ---- r = dropout(a) + x (a updated in place to its dropped value), part[g][bx] = (sum r, sum r^2)
// groups are contiguous chunks of `inner` elements (LayerNorm axes {0,1,2}: one per utterance)
__global__ __launch_bounds__(kEwThreads) void residual_dropout_stats_k(
    float* __restrict__ a, const float* __restrict__ x, float* __restrict__ r, double* __restrict__ part,
    size_t inner, uint32_t thr, float keepScale, uint32_t seed, uint32_t stream) {
  const int g = blockIdx.y;
  const size_t base = (size_t)g * inner;
  const size_t n4 = inner >> 2;
  double s = 0, ss = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const size_t e = base + 4 * i;
    float4 av = *(const float4*)(a + e);
    float4 rv = x ? *(const float4*)(x + e) : make_float4(0.f, 0.f, 0.f, 0.f);
    if (thr) {
      av.x = keep_elem(e, seed, stream, thr) ? av.x * keepScale : 0.f;
      av.y = keep_elem(e + 1, seed, stream, thr) ? av.y * keepScale : 0.f;
      av.z = keep_elem(e + 2, seed, stream, thr) ? av.z * keepScale : 0.f;
      av.w = keep_elem(e + 3, seed, stream, thr) ? av.w * keepScale : 0.f;
      if (r != a || !x) *(float4*)(a + e) = av;   // (r stored over a: the sum below is what lands there -- one store, not two)
    }
    rv.x += av.x; rv.y += av.y; rv.z += av.z; rv.w += av.w;
    if (r != a || x) *(float4*)(r + e) = rv;  // plain LayerNorm (r aliases a, no residual): nothing to write
    const float p1 = (rv.x + rv.y) + (rv.z + rv.w);
    const float p2 = (rv.x * rv.x + rv.y * rv.y) + (rv.z * rv.z + rv.w * rv.w);
    s += (double)p1;
    ss += (double)p2;
  }
  block_partial2(s, ss, part + 2 * ((size_t)g * gridDim.x + blockIdx.x));
}

// y = gamma * (r - mu) * rstd + beta ; mu/rstd from the partials (sum, sumsq), biased variance + eps.
// writes mean/rstd (fp32) for the backward pass.
__global__ __launch_bounds__(kEwThreads) void ln_apply_k(const float* __restrict__ r, float* __restrict__ y,
                                                        const double* __restrict__ part,
                                                        float* __restrict__ meanRstd, size_t inner,
                                                        const float* __restrict__ gammaBeta, float eps) {
  const int g = blockIdx.y;
  double s1, s2;
  sum_parts(part, g, gridDim.x, s1, s2);
  const double mu = s1 / (double)inner;
  double var = s2 / (double)inner - mu * mu;
  if (var < 0) var = 0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  const float muf = (float)mu;
  if (blockIdx.x == 0 && threadIdx.x == 0) { meanRstd[2 * g] = muf; meanRstd[2 * g + 1] = rstd; }
  const float gam = gammaBeta[0] * rstd, bet = gammaBeta[1];
  const size_t base = (size_t)g * inner, n4 = inner >> 2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    float4 v = *(const float4*)(r + base + 4 * i);
    v.x = (v.x - muf) * gam + bet; v.y = (v.y - muf) * gam + bet;
    v.z = (v.z - muf) * gam + bet; v.w = (v.w - muf) * gam + bet;
    *(float4*)(y + base + 4 * i) = v;
  }
}

// Small groups (per-frame LayerNorm, axes {1,2}: inner = H*C <= kLnSmall): ONE block per group,
// ONE pass -- the group is held in registers between the statistics and the apply.
constexpr int kLnSmallV = (int)(kLnSmall / 4 / kEwThreads);  // float4 per thread at most
// NJ = the group's 16-byte chunks per thread, rounded up (the host picks the instance: a 2160-float frame takes 3, not 8)
template <int NJ>
__global__ __launch_bounds__(kEwThreads) void residual_ln_small_k(
    float* __restrict__ a, const float* __restrict__ x, float* __restrict__ r, float* __restrict__ y,
    float* __restrict__ meanRstd, size_t inner, const float* __restrict__ gammaBeta, float eps,
    uint32_t thr, float keepScale, uint32_t seed, uint32_t stream) {
  __shared__ double bc[2];
  const int g = blockIdx.x;
  const size_t base = (size_t)g * inner;
  const int n4 = (int)(inner >> 2);
  // every load of the group before the first use, none behind a lane predicate (a chunk past the group re-reads its last
  // chunk and is ignored): as `if (i < n4) { load; load; use; store }` per chunk, hipcc waited for each chunk's loads in turn
  float4 v[NJ], avs[NJ], xvs[NJ];
  const int lastc = n4 - 1;
#pragma unroll
  for (int j = 0; j < NJ; ++j) avs[j] = *(const float4*)(a + base + 4 * (size_t)min((int)threadIdx.x + j * kEwThreads, lastc));
  if (x) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) xvs[j] = *(const float4*)(x + base + 4 * (size_t)min((int)threadIdx.x + j * kEwThreads, lastc));
  } else {
#pragma unroll
    for (int j = 0; j < NJ; ++j) xvs[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  double s = 0, ss = 0;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int i = threadIdx.x + j * kEwThreads;
    v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n4) {
      const size_t e = base + 4 * (size_t)i;
      float4 av = avs[j];
      float4 rv = xvs[j];
      if (thr) {
        av.x = keep_elem(e, seed, stream, thr) ? av.x * keepScale : 0.f;
        av.y = keep_elem(e + 1, seed, stream, thr) ? av.y * keepScale : 0.f;
        av.z = keep_elem(e + 2, seed, stream, thr) ? av.z * keepScale : 0.f;
        av.w = keep_elem(e + 3, seed, stream, thr) ? av.w * keepScale : 0.f;
        if (r != a || !x) *(float4*)(a + e) = av;   // (r stored over a: the sum below is what lands there -- one store, not two)
      }
      rv.x += av.x; rv.y += av.y; rv.z += av.z; rv.w += av.w;
      if (r != a || x) *(float4*)(r + e) = rv;
      v[j] = rv;
      s += (double)((rv.x + rv.y) + (rv.z + rv.w));
      ss += (double)((rv.x * rv.x + rv.y * rv.y) + (rv.z * rv.z + rv.w * rv.w));
    }
  }
  block_partial2(s, ss, bc);
  __syncthreads();
  const double mu = bc[0] / (double)inner;
  double var = bc[1] / (double)inner - mu * mu;
  if (var < 0) var = 0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  const float muf = (float)mu;
  if (threadIdx.x == 0) { meanRstd[2 * g] = muf; meanRstd[2 * g + 1] = rstd; }
  const float gam = gammaBeta[0] * rstd, bet = gammaBeta[1];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int i = threadIdx.x + j * kEwThreads;
    if (i < n4) {
      float4 o = v[j];
      o.x = (o.x - muf) * gam + bet; o.y = (o.y - muf) * gam + bet;
      o.z = (o.z - muf) * gam + bet; o.w = (o.w - muf) * gam + bet;
      *(float4*)(y + base + 4 * (size_t)i) = o;
    }
  }
}

// The same one-pass LayerNorm with ONE WAVE per group (round 6; rows of at most 64 * kLnWaveV float4 = 2304 floats: the per-frame
// LayerNorms of the streaming recipe, 1200 ... 2160 floats).  Against residual_ln_small_k (one 256-thread block per group): no LDS, no
// __syncthreads between the statistics and the apply (the two double sums cross the wave by shuffles), 44 of 256 threads no longer
// idle on a 300-chunk row, and a workgroup's four rows are independent -- 11968 x 1200: 100 -> ~60 us (profiles/r06_run37_*).
// NJ = 16-byte chunks per lane.  Values: the block kernel's, up to the order of the double-precision sums.
constexpr int kLnWaveV = 9;
template <int NJ>
__global__ __launch_bounds__(kEwThreads) void residual_ln_wave_k(
    int groups, float* __restrict__ a, const float* __restrict__ x, float* __restrict__ r, float* __restrict__ y,
    float* __restrict__ meanRstd, size_t inner, const float* __restrict__ gammaBeta, float eps,
    uint32_t thr, float keepScale, uint32_t seed, uint32_t stream) {
  const int lane = threadIdx.x & 63;
  const int g = blockIdx.x * (kEwThreads / 64) + (threadIdx.x >> 6);
  if (g >= groups) return;
  const size_t base = (size_t)g * inner;
  const int n4 = (int)(inner >> 2), lastc = n4 - 1;
  float4 v[NJ], avs[NJ], xvs[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) avs[j] = *(const float4*)(a + base + 4 * (size_t)min(lane + 64 * j, lastc));
  if (x) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) xvs[j] = *(const float4*)(x + base + 4 * (size_t)min(lane + 64 * j, lastc));
  } else {
#pragma unroll
    for (int j = 0; j < NJ; ++j) xvs[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  double s = 0, ss = 0;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int i = lane + 64 * j;
    v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n4) {
      const size_t e = base + 4 * (size_t)i;
      float4 av = avs[j];
      float4 rv = xvs[j];
      if (thr) {
        av.x = keep_elem(e, seed, stream, thr) ? av.x * keepScale : 0.f;
        av.y = keep_elem(e + 1, seed, stream, thr) ? av.y * keepScale : 0.f;
        av.z = keep_elem(e + 2, seed, stream, thr) ? av.z * keepScale : 0.f;
        av.w = keep_elem(e + 3, seed, stream, thr) ? av.w * keepScale : 0.f;
        if (r != a || !x) *(float4*)(a + e) = av;
      }
      rv.x += av.x; rv.y += av.y; rv.z += av.z; rv.w += av.w;
      if (r != a || x) *(float4*)(r + e) = rv;
      v[j] = rv;
      s += (double)((rv.x + rv.y) + (rv.z + rv.w));
      ss += (double)((rv.x * rv.x + rv.y * rv.y) + (rv.z * rv.z + rv.w * rv.w));
    }
  }
  s = wave_sum_f64(s);
  ss = wave_sum_f64(ss);
  const double mu = s / (double)inner;
  double var = ss / (double)inner - mu * mu;
  if (var < 0) var = 0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  const float muf = (float)mu;
  if (lane == 0) { meanRstd[2 * g] = muf; meanRstd[2 * g + 1] = rstd; }
  const float gam = gammaBeta[0] * rstd, bet = gammaBeta[1];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int i = lane + 64 * j;
    if (i < n4) {
      float4 o = v[j];
      o.x = (o.x - muf) * gam + bet; o.y = (o.y - muf) * gam + bet;
      o.z = (o.z - muf) * gam + bet; o.w = (o.w - muf) * gam + bet;
      *(float4*)(y + base + 4 * (size_t)i) = o;
    }
  }
}

// Same op for a group size that is not a multiple of 4 (the reference's TDSBlock golden vector: inner = 5 mel rows x
// 2 channels = 10): scalar loads, one block per group, forward only.  Not on the training hot path.
__global__ __launch_bounds__(kEwThreads) void residual_ln_scalar_k(
    float* __restrict__ a, const float* __restrict__ x, float* __restrict__ r, float* __restrict__ y,
    float* __restrict__ meanRstd, size_t inner, const float* __restrict__ gammaBeta, float eps,
    uint32_t thr, float keepScale, uint32_t seed, uint32_t stream) {
  __shared__ double bc[2];
  const int g = blockIdx.x;
  const size_t base = (size_t)g * inner;
  double s = 0, ss = 0;
  for (size_t i = threadIdx.x; i < inner; i += kEwThreads) {
    const size_t e = base + i;
    float av = a[e];
    if (thr) { av = keep_elem(e, seed, stream, thr) ? av * keepScale : 0.f; if (r != a || !x) a[e] = av; }
    const float rv = (x ? x[e] : 0.f) + av;
    r[e] = rv;
    s += (double)rv;
    ss += (double)(rv * rv);
  }
  block_partial2(s, ss, bc);
  __syncthreads();
  const double mu = bc[0] / (double)inner;
  double var = bc[1] / (double)inner - mu * mu;
  if (var < 0) var = 0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  const float muf = (float)mu;
  if (threadIdx.x == 0) { meanRstd[2 * g] = muf; meanRstd[2 * g + 1] = rstd; }
  const float gam = gammaBeta[0] * rstd, bet = gammaBeta[1];
  for (size_t i = threadIdx.x; i < inner; i += kEwThreads) y[base + i] = (r[base + i] - muf) * gam + bet;
}

// backward reduce: part[g][bx] = (sum dy, sum dy * xhat), xhat = (r - mu) * rstd
__global__ __launch_bounds__(kEwThreads) void ln_bwd_reduce_k(const float* __restrict__ r,
                                                             const float* __restrict__ dy,
                                                             const float* __restrict__ meanRstd,
                                                             double* __restrict__ part, size_t inner) {
  const int g = blockIdx.y;
  const float mu = meanRstd[2 * g], rstd = meanRstd[2 * g + 1];
  const size_t base = (size_t)g * inner, n4 = inner >> 2;
  double s1 = 0, s2 = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    float4 rv = *(const float4*)(r + base + 4 * i);
    float4 dv = *(const float4*)(dy + base + 4 * i);
    s1 += (double)((dv.x + dv.y) + (dv.z + dv.w));
    s2 += (double)((dv.x * ((rv.x - mu) * rstd) + dv.y * ((rv.y - mu) * rstd)) +
                   (dv.z * ((rv.z - mu) * rstd) + dv.w * ((rv.w - mu) * rstd)));
  }
  block_partial2(s1, s2, part + 2 * ((size_t)g * gridDim.x + blockIdx.x));
}

// backward apply: dr = gamma*rstd*(dy - S1/n - xhat*S2/n).
// Optional second output dmask = dr * (maskSrc > 0 ? maskScale : 0)  (ReLU+dropout of the
// producer branch: maskSrc is its stored post-dropout output).
// Block (0, g) also leaves the group totals in sums[2g..2g+1] for the parameter gradients.
__global__ __launch_bounds__(kEwThreads) void ln_bwd_apply_k(const float* __restrict__ r,
                                                            const float* __restrict__ dy,
                                                            const float* __restrict__ meanRstd,
                                                            const double* __restrict__ part,
                                                            double* __restrict__ sums,
                                                            const float* __restrict__ gammaBeta,
                                                            float* __restrict__ dr,
                                                            const float* __restrict__ maskSrc,
                                                            float* __restrict__ dmask, float maskScale,
                                                            size_t inner) {
  const int g = blockIdx.y;
  const float mu = meanRstd[2 * g], rstd = meanRstd[2 * g + 1];
  double S1, S2;
  sum_parts(part, g, gridDim.x, S1, S2);
  if (blockIdx.x == 0 && threadIdx.x == 0) { sums[2 * g] = S1; sums[2 * g + 1] = S2; }
  const float c1 = (float)(S1 / (double)inner), c2 = (float)(S2 / (double)inner);
  const float gr = gammaBeta[0] * rstd;
  const size_t base = (size_t)g * inner, n4 = inner >> 2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const size_t e = base + 4 * i;
    float4 rv = *(const float4*)(r + e);
    float4 dv = *(const float4*)(dy + e);
    float4 o;
    o.x = gr * (dv.x - c1 - (rv.x - mu) * rstd * c2);
    o.y = gr * (dv.y - c1 - (rv.y - mu) * rstd * c2);
    o.z = gr * (dv.z - c1 - (rv.z - mu) * rstd * c2);
    o.w = gr * (dv.w - c1 - (rv.w - mu) * rstd * c2);
    *(float4*)(dr + e) = o;
    if (dmask) {
      float4 mv = *(const float4*)(maskSrc + e);
      float4 d2;
      d2.x = mv.x > 0.f ? o.x * maskScale : 0.f;
      d2.y = mv.y > 0.f ? o.y * maskScale : 0.f;
      d2.z = mv.z > 0.f ? o.z * maskScale : 0.f;
      d2.w = mv.w > 0.f ? o.w * maskScale : 0.f;
      *(float4*)(dmask + e) = d2;
    }
  }
}

// small groups: one block per group, r and dy held in registers between reduce and apply
template <int NJ>
__global__ __launch_bounds__(kEwThreads) void ln_bwd_small_k(const float* __restrict__ r, const float* __restrict__ dy,
                                                            const float* __restrict__ meanRstd,
                                                            double* __restrict__ sums,
                                                            const float* __restrict__ gammaBeta,
                                                            float* __restrict__ dr,
                                                            const float* __restrict__ maskSrc,
                                                            float* __restrict__ dmask, float maskScale,
                                                            size_t inner) {
  __shared__ double bc[2];
  const int g = blockIdx.x;
  const float mu = meanRstd[2 * g], rstd = meanRstd[2 * g + 1];
  const size_t base = (size_t)g * inner;
  const int n4 = (int)(inner >> 2);
  // (loads first and unpredicated, as in residual_ln_small_k; the dropout mask source of the apply loop rides with them)
  float4 xh[NJ], dv[NJ], mvs[NJ];
  const int lastc = n4 - 1;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const size_t e = base + 4 * (size_t)min((int)threadIdx.x + j * kEwThreads, lastc);
    xh[j] = *(const float4*)(r + e);
    dv[j] = *(const float4*)(dy + e);
  }
  if (dmask) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) mvs[j] = *(const float4*)(maskSrc + base + 4 * (size_t)min((int)threadIdx.x + j * kEwThreads, lastc));
  } else {
#pragma unroll
    for (int j = 0; j < NJ; ++j) mvs[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  double s1 = 0, s2 = 0;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int i = threadIdx.x + j * kEwThreads;
    const float4 rv = xh[j];
    xh[j] = make_float4((rv.x - mu) * rstd, (rv.y - mu) * rstd, (rv.z - mu) * rstd, (rv.w - mu) * rstd);
    if (i < n4) {
      s1 += (double)((dv[j].x + dv[j].y) + (dv[j].z + dv[j].w));
      s2 += (double)((dv[j].x * xh[j].x + dv[j].y * xh[j].y) + (dv[j].z * xh[j].z + dv[j].w * xh[j].w));
    }
  }
  block_partial2(s1, s2, bc);
  __syncthreads();
  if (threadIdx.x == 0) { sums[2 * g] = bc[0]; sums[2 * g + 1] = bc[1]; }
  const float c1 = (float)(bc[0] / (double)inner), c2 = (float)(bc[1] / (double)inner);
  const float gr = gammaBeta[0] * rstd;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int i = threadIdx.x + j * kEwThreads;
    if (i < n4) {
      const size_t e = base + 4 * (size_t)i;
      float4 o;
      o.x = gr * (dv[j].x - c1 - xh[j].x * c2);
      o.y = gr * (dv[j].y - c1 - xh[j].y * c2);
      o.z = gr * (dv[j].z - c1 - xh[j].z * c2);
      o.w = gr * (dv[j].w - c1 - xh[j].w * c2);
      *(float4*)(dr + e) = o;
      if (dmask) {
        const float4 mv = mvs[j];
        float4 d2;
        d2.x = mv.x > 0.f ? o.x * maskScale : 0.f;
        d2.y = mv.y > 0.f ? o.y * maskScale : 0.f;
        d2.z = mv.z > 0.f ? o.z * maskScale : 0.f;
        d2.w = mv.w > 0.f ? o.w * maskScale : 0.f;
        *(float4*)(dmask + e) = d2;
      }
    }
  }
}

// one wave per group (see residual_ln_wave_k)
template <int NJ>
__global__ __launch_bounds__(kEwThreads) void ln_bwd_wave_k(int groups, const float* __restrict__ r, const float* __restrict__ dy,
                                                           const float* __restrict__ meanRstd, double* __restrict__ sums,
                                                           const float* __restrict__ gammaBeta, float* __restrict__ dr,
                                                           const float* __restrict__ maskSrc, float* __restrict__ dmask, float maskScale,
                                                           size_t inner) {
  const int lane = threadIdx.x & 63;
  const int g = blockIdx.x * (kEwThreads / 64) + (threadIdx.x >> 6);
  if (g >= groups) return;
  const float mu = meanRstd[2 * g], rstd = meanRstd[2 * g + 1];
  const size_t base = (size_t)g * inner;
  const int n4 = (int)(inner >> 2), lastc = n4 - 1;
  float4 xh[NJ], dv[NJ], mvs[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const size_t e = base + 4 * (size_t)min(lane + 64 * j, lastc);
    xh[j] = *(const float4*)(r + e);
    dv[j] = *(const float4*)(dy + e);
  }
  if (dmask) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) mvs[j] = *(const float4*)(maskSrc + base + 4 * (size_t)min(lane + 64 * j, lastc));
  } else {
#pragma unroll
    for (int j = 0; j < NJ; ++j) mvs[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  double s1 = 0, s2 = 0;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const float4 rv = xh[j];
    xh[j] = make_float4((rv.x - mu) * rstd, (rv.y - mu) * rstd, (rv.z - mu) * rstd, (rv.w - mu) * rstd);
    if (lane + 64 * j < n4) {
      s1 += (double)((dv[j].x + dv[j].y) + (dv[j].z + dv[j].w));
      s2 += (double)((dv[j].x * xh[j].x + dv[j].y * xh[j].y) + (dv[j].z * xh[j].z + dv[j].w * xh[j].w));
    }
  }
  s1 = wave_sum_f64(s1);
  s2 = wave_sum_f64(s2);
  if (lane == 0) { sums[2 * g] = s1; sums[2 * g + 1] = s2; }
  const float c1 = (float)(s1 / (double)inner), c2 = (float)(s2 / (double)inner);
  const float gr = gammaBeta[0] * rstd;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int i = lane + 64 * j;
    if (i < n4) {
      const size_t e = base + 4 * (size_t)i;
      float4 o;
      o.x = gr * (dv[j].x - c1 - xh[j].x * c2);
      o.y = gr * (dv[j].y - c1 - xh[j].y * c2);
      o.z = gr * (dv[j].z - c1 - xh[j].z * c2);
      o.w = gr * (dv[j].w - c1 - xh[j].w * c2);
      *(float4*)(dr + e) = o;
      if (dmask) {
        const float4 mv = mvs[j];
        float4 d2;
        d2.x = mv.x > 0.f ? o.x * maskScale : 0.f;
        d2.y = mv.y > 0.f ? o.y * maskScale : 0.f;
        d2.z = mv.z > 0.f ? o.z * maskScale : 0.f;
        d2.w = mv.w > 0.f ? o.w * maskScale : 0.f;
        *(float4*)(dmask + e) = d2;
      }
    }
  }
}

// scalar-parameter gradients: dgamma = sum_g S2_g, dbeta = sum_g S1_g.  One workgroup of 1024 threads, every thread a
// fixed strided slice with four loads in flight, then a fixed-order tree: deterministic.  (One wave walking the groups
// took 73-280 us per call on the per-frame LayerNorms of the streaming recipe -- 48 000 groups -- 3 % of that step.)
__global__ __launch_bounds__(1024) void ln_param_grad_k(const double* __restrict__ sums, int groups, float* __restrict__ dGammaBeta) {
  __shared__ double r1[1024], r2[1024];
  const int tid = threadIdx.x;
  double a1 = 0, a2 = 0, b1 = 0, b2 = 0, c1 = 0, c2 = 0, d1 = 0, d2 = 0;
  int g = tid;
  for (; g + 3 * 1024 < groups; g += 4 * 1024) {
    const double2 v0 = *(const double2*)(sums + 2 * (size_t)g), v1 = *(const double2*)(sums + 2 * (size_t)(g + 1024)),
                  v2 = *(const double2*)(sums + 2 * (size_t)(g + 2048)), v3 = *(const double2*)(sums + 2 * (size_t)(g + 3072));
    a1 += v0.x; a2 += v0.y; b1 += v1.x; b2 += v1.y; c1 += v2.x; c2 += v2.y; d1 += v3.x; d2 += v3.y;
  }
  for (; g < groups; g += 1024) { a1 += sums[2 * (size_t)g]; a2 += sums[2 * (size_t)g + 1]; }
  r1[tid] = (a1 + b1) + (c1 + d1);
  r2[tid] = (a2 + b2) + (c2 + d2);
  __syncthreads();
  for (int off = 512; off > 0; off >>= 1) {
    if (tid < off) { r1[tid] += r1[tid + off]; r2[tid] += r2[tid + off]; }
    __syncthreads();
  }
  if (tid == 0) {
    dGammaBeta[0] = (float)r2[0];
    dGammaBeta[1] = (float)r1[0];
  }
}

// ---- dropout in place (+ optional ReLU first), mask from the stateless hash
__global__ __launch_bounds__(kEwThreads) void dropout_k(float* __restrict__ x, size_t n, uint32_t thr,
                                                       float keepScale, uint32_t seed, uint32_t stream) {
  const size_t n4 = n >> 2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const size_t e = 4 * i;
    float4 v = *(const float4*)(x + e);
    v.x = keep_elem(e, seed, stream, thr) ? v.x * keepScale : 0.f;
    v.y = keep_elem(e + 1, seed, stream, thr) ? v.y * keepScale : 0.f;
    v.z = keep_elem(e + 2, seed, stream, thr) ? v.z * keepScale : 0.f;
    v.w = keep_elem(e + 3, seed, stream, thr) ? v.w * keepScale : 0.f;
    *(float4*)(x + e) = v;
  }
  for (size_t e = (n4 << 2) + (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x)
    x[e] = keep_elem(e, seed, stream, thr) ? x[e] * keepScale : 0.f;
}

// dx = dy * (src > 0 ? scale : 0)   (ReLU [+dropout] backward from the stored output)
__global__ __launch_bounds__(kEwThreads) void mask_bwd_k(const float* __restrict__ dy, const float* __restrict__ src,
                                                        float* __restrict__ dx, size_t n, float scale) {
  const size_t n4 = n >> 2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    float4 d = *(const float4*)(dy + 4 * i);
    float4 s = *(const float4*)(src + 4 * i);
    d.x = s.x > 0.f ? d.x * scale : 0.f; d.y = s.y > 0.f ? d.y * scale : 0.f;
    d.z = s.z > 0.f ? d.z * scale : 0.f; d.w = s.w > 0.f ? d.w * scale : 0.f;
    *(float4*)(dx + 4 * i) = d;
  }
  for (size_t e = (n4 << 2) + (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x)
    dx[e] = src[e] > 0.f ? dy[e] * scale : 0.f;
}

__global__ __launch_bounds__(kEwThreads) void fill_k(float* __restrict__ y, size_t n, float v) {
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) y[e] = v;
}

// y (+)= alpha * x
__global__ __launch_bounds__(kEwThreads) void axpy_k(float* __restrict__ y, const float* __restrict__ x, size_t n, float alpha) {
  const size_t n4 = n >> 2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    float4 a = *(const float4*)(y + 4 * i), b = *(const float4*)(x + 4 * i);
    a.x += alpha * b.x; a.y += alpha * b.y; a.z += alpha * b.z; a.w += alpha * b.w;
    *(float4*)(y + 4 * i) = a;
  }
  for (size_t e = (n4 << 2) + (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x)
    y[e] += alpha * x[e];
}

// ---- batched 2-D transpose: in [G][R][Cc] -> out [G][Cc][R]  (Reorder between the
// reference's time-fastest input (T,NFEAT,1,B) and the frame-major internal layout)
__global__ __launch_bounds__(256) void transpose_k(const float* __restrict__ in, float* __restrict__ out, int R, int Cc) {
  __shared__ float tile[32][33];
  const size_t g = blockIdx.z;
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int k = ty; k < 32; k += 8) {
    int rr = r0 + k, cc = c0 + tx;
    tile[k][tx] = (rr < R && cc < Cc) ? in[(g * R + rr) * Cc + cc] : 0.f;
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    int cc = c0 + k, rr = r0 + tx;
    if (rr < R && cc < Cc) out[(g * Cc + cc) * R + rr] = tile[tx][k];
  }
}

// ---- GLU over the last (channel) axis: x [M][2*half] -> y [M][half] = a * sigmoid(b)
__global__ __launch_bounds__(kEwThreads) void glu_fwd_k(const float* __restrict__ x, float* __restrict__ y, size_t M, int half) {
  const size_t n = M * (size_t)half;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
    size_t m = e / half;
    int c = (int)(e - m * half);
    float a = x[m * 2 * half + c], b = x[m * 2 * half + half + c];
    y[e] = a / (1.f + __expf(-b));
  }
}
__global__ __launch_bounds__(kEwThreads) void glu_bwd_k(const float* __restrict__ x, const float* __restrict__ dy,
                                                       float* __restrict__ dx, size_t M, int half) {
  const size_t n = M * (size_t)half;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
    size_t m = e / half;
    int c = (int)(e - m * half);
    float a = x[m * 2 * half + c], b = x[m * 2 * half + half + c];
    float s = 1.f / (1.f + __expf(-b));
    float d = dy[e];
    dx[m * 2 * half + c] = d * s;
    dx[m * 2 * half + half + c] = d * a * s * (1.f - s);
  }
}

// ---- optimizer: sum of squares (fp64 accumulate), then SGD with momentum + global-norm clip
__global__ __launch_bounds__(kEwThreads) void sumsq_k(const float* __restrict__ g, size_t n, double* __restrict__ out) {
  double s = 0;
  const size_t n4 = n >> 2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    float4 v = *(const float4*)(g + 4 * i);
    s += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
  }
  for (size_t e = (n4 << 2) + (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x)
    s += (double)g[e] * g[e];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  // DETERMINISTIC: one partial per workgroup (waves added in wave order), the partials added in a fixed order by
  // sumsq_finish_k.  The norm gates every update (clip coefficient, non-finite skip) on every data-parallel rank: with
  // atomicAdd(double) in arrival order the float clip coefficient could differ by an ulp between replicas and let them drift.
  __shared__ double sm[kEwThreads / 64];
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0;
#pragma unroll
    for (int w = 0; w < kEwThreads / 64; ++w) t += sm[w];
    out[blockIdx.x] = t;
  }
}

// out[0] = (accumulate ? out[0] : 0) + sum of the `parts` workgroup partials, always in the same order
__global__ __launch_bounds__(256) void sumsq_finish_k(const double* __restrict__ part, int parts, double* __restrict__ out, int accumulate) {
  __shared__ double sm[256];
  double t = 0;
  for (int i = threadIdx.x; i < parts; i += 256) t += part[i];
  sm[threadIdx.x] = t;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) sm[threadIdx.x] += sm[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = (accumulate ? out[0] : 0.0) + sm[0];
}

// acc[0] = sum g^2 over the network gradients, acc[1] = over the criterion gradients (w2l_sumsq), batch = the
// all-reduced number of utterances (device scalar riding in the gradient arena's tail) or null.
// -> acc[2] = the clip norm^2 (network, + criterion if clampCrit) or NaN if ANY gradient is non-finite,
//    acc[3] = 1 / batch (0: caller's gradScale stands), acc[4] += 1 for a skipped (non-finite) update.
__global__ void grad_guard_k(double* __restrict__ acc, const float* __restrict__ batch, int clampCrit) {
  const double tot = acc[0] + acc[1];
  const bool ok = isfinite(tot) && (!batch || (isfinite(*batch) && *batch > 0.f));
  acc[2] = ok ? acc[0] + (clampCrit ? acc[1] : 0.0) : __longlong_as_double(0x7ff8000000000000ll);
  acc[3] = (ok && batch) ? 1.0 / (double)*batch : 0.0;
  if (!ok) acc[4] += 1.0;
}

// fl::SGDOptimizer::step with momentum (no dampening/nesterov/wd, as the recipes use):
//   g' = g * gradScale * clipCoef ; v = mom*v + g' ; p -= lr * v
// clipCoef = min(1, maxNorm / (||g*gradScale|| + 1e-6))  (fl::clipGradNorm), maxNorm <= 0: off.
__global__ __launch_bounds__(kEwThreads) void sgd_k(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ v, size_t n, float lr, float mom,
                                                   float gradScale, float maxNorm, const double* __restrict__ sumsq,
                                                   int guarded) {
  float coef = gradScale;
  if (sumsq) {
    // Non-finite gradient norm (a NaN / Inf anywhere in the reduced gradient): leave parameters and momentum
    // untouched -- ALWAYS, whether or not clipping is on (--maxgradnorm=0 is the reference default).  The reference
    // aborts on a non-finite loss (Train.cpp:1686-1698) and makes all ranks skip an update together through an
    // all-reduced flag (:1651-1660); here the norm is computed from the ALL-REDUCED gradient, so every rank takes the
    // same decision without a host round trip.  w2l_trainer_grad_norm() / w2l_trainer_skipped_updates() expose it.
    if (!isfinite(sumsq[0])) return;
    if (guarded && sumsq[1] > 0.0) coef = gradScale = (float)sumsq[1];  // 1 / all-reduced batch size (grad_guard_k)
    if (maxNorm > 0.f) {
      float norm = (float)sqrt(sumsq[0]) * gradScale;
      float c = maxNorm / (norm + 1e-6f);
      if (c < 1.f) coef *= c;
    }
  }
  const size_t n4 = n >> 2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    float4 pv = *(const float4*)(p + 4 * i), gv = *(const float4*)(g + 4 * i);
    if (mom != 0.f) {
      float4 vv = *(const float4*)(v + 4 * i);
      vv.x = mom * vv.x + gv.x * coef; vv.y = mom * vv.y + gv.y * coef;
      vv.z = mom * vv.z + gv.z * coef; vv.w = mom * vv.w + gv.w * coef;
      *(float4*)(v + 4 * i) = vv;
      pv.x -= lr * vv.x; pv.y -= lr * vv.y; pv.z -= lr * vv.z; pv.w -= lr * vv.w;
    } else {
      pv.x -= lr * gv.x * coef; pv.y -= lr * gv.y * coef; pv.z -= lr * gv.z * coef; pv.w -= lr * gv.w * coef;
    }
    *(float4*)(p + 4 * i) = pv;
  }
  for (size_t e = (n4 << 2) + (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
    float gg = g[e] * coef;
    if (mom != 0.f) { float vv = mom * v[e] + gg; v[e] = vv; p[e] -= lr * vv; }
    else p[e] -= lr * gg;
  }
}

// fl::AdagradOptimizer::step (--netoptim=adagrad of recipes/sota/2019/librivox/train_am_transformer_ctc.cfg:25-26; the class is
// un-vendored Flashlight: variance += g'^2 ; p -= lr * g' / (sqrt(variance) + eps), eps = 1e-8), with the same gradient scale,
// global-norm clip and non-finite guard as sgd_k
__global__ __launch_bounds__(kEwThreads) void adagrad_k(float* __restrict__ p, const float* __restrict__ g,
                                                       float* __restrict__ var, size_t n, float lr, float eps,
                                                       float gradScale, float maxNorm, const double* __restrict__ sumsq,
                                                       int guarded) {
  float coef = gradScale;
  if (sumsq) {
    if (!isfinite(sumsq[0])) return;
    if (guarded && sumsq[1] > 0.0) coef = gradScale = (float)sumsq[1];
    if (maxNorm > 0.f) {
      float norm = (float)sqrt(sumsq[0]) * gradScale;
      float c = maxNorm / (norm + 1e-6f);
      if (c < 1.f) coef *= c;
    }
  }
  const size_t n4 = n >> 2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    float4 pv = *(const float4*)(p + 4 * i), gv = *(const float4*)(g + 4 * i), vv = *(const float4*)(var + 4 * i);
    gv.x *= coef; gv.y *= coef; gv.z *= coef; gv.w *= coef;
    vv.x += gv.x * gv.x; vv.y += gv.y * gv.y; vv.z += gv.z * gv.z; vv.w += gv.w * gv.w;
    *(float4*)(var + 4 * i) = vv;
    pv.x -= lr * gv.x / (sqrtf(vv.x) + eps); pv.y -= lr * gv.y / (sqrtf(vv.y) + eps);
    pv.z -= lr * gv.z / (sqrtf(vv.z) + eps); pv.w -= lr * gv.w / (sqrtf(vv.w) + eps);
    *(float4*)(p + 4 * i) = pv;
  }
  for (size_t e = (n4 << 2) + (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
    float gg = g[e] * coef;
    float vv = var[e] + gg * gg;
    var[e] = vv;
    p[e] -= lr * gg / (sqrtf(vv) + eps);
  }
}

// fl::AdadeltaOptimizer::step (--netoptim=adadelta --lr=0.4 of recipes/sota/2019/librispeech/train_am_transformer_ctc.cfg:23-26;
// un-vendored Flashlight class, rho = 0.9, eps = 1e-8 by the Trainer's --optimrho / --optimepsilon defaults):
//   accGrad = rho accGrad + (1 - rho) g'^2 ; delta = sqrt(accDelta + eps) / sqrt(accGrad + eps) g' ; p -= lr delta ;
//   accDelta = rho accDelta + (1 - rho) delta^2
__global__ __launch_bounds__(kEwThreads) void adadelta_k(float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ accG, float* __restrict__ accD, size_t n, float lr,
                                                        float rho, float eps, float gradScale, float maxNorm,
                                                        const double* __restrict__ sumsq, int guarded) {
  float coef = gradScale;
  if (sumsq) {
    if (!isfinite(sumsq[0])) return;
    if (guarded && sumsq[1] > 0.0) coef = gradScale = (float)sumsq[1];
    if (maxNorm > 0.f) {
      float norm = (float)sqrt(sumsq[0]) * gradScale;
      float c = maxNorm / (norm + 1e-6f);
      if (c < 1.f) coef *= c;
    }
  }
  const float om = 1.f - rho;
  auto one = [&](float& pv, float gv, float& ag, float& ad) {
    gv *= coef;
    ag = rho * ag + om * gv * gv;
    const float d = sqrtf(ad + eps) / sqrtf(ag + eps) * gv;
    pv -= lr * d;
    ad = rho * ad + om * d * d;
  };
  const size_t n4 = n >> 2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    float4 pv = *(const float4*)(p + 4 * i), gv = *(const float4*)(g + 4 * i);
    float4 ag = *(const float4*)(accG + 4 * i), ad = *(const float4*)(accD + 4 * i);
    one(pv.x, gv.x, ag.x, ad.x); one(pv.y, gv.y, ag.y, ad.y); one(pv.z, gv.z, ag.z, ad.z); one(pv.w, gv.w, ag.w, ad.w);
    *(float4*)(accG + 4 * i) = ag;
    *(float4*)(accD + 4 * i) = ad;
    *(float4*)(p + 4 * i) = pv;
  }
  for (size_t e = (n4 << 2) + (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x)
    one(p[e], g[e], accG[e], accD[e]);
}

// (shared with layernorm_images.hip)
int ln_param_grad(const double* sums, int groups, float* dGammaBeta, hipStream_t stream) {
  hipLaunchKernelGGL(ln_param_grad_k, dim3(1), dim3(1024), 0, stream, sums, groups, dGammaBeta);
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}

}  // namespace w2l

using namespace w2l;

#define W2L_S ((hipStream_t)stream)

// W2L_LN_WAVE=0 (probe library): the one-block-per-group kernels for groups of <= 2304 floats too (A/B runs)
static bool ln_wave_enabled() {
  static const bool v = [] { const char* e = tune_env("W2L_LN_WAVE"); return !(e && e[0] == '0'); }();
  return v;
}

static inline unsigned ln_parts(size_t inner) {
  size_t g = ((inner >> 2) + kEwThreads * 4 - 1) / (kEwThreads * 4);  // >= 4 float4 per thread
  if (g > (size_t)kLnMaxParts) g = kLnMaxParts;
  if (g < 1) g = 1;
  return (unsigned)g;
}

// scratch (in doubles) of the two LayerNorm entry points: [2*groups totals | 2*groups*parts partials]
W2L_API size_t w2l_layernorm_scratch_doubles(int groups, size_t inner) {
  if (groups <= 0) return 0;
  return 2 * (size_t)groups * (1 + (inner <= kLnSmall ? 0 : ln_parts(inner)));
}

// LayerNorm over `groups` contiguous chunks of `inner` elements (inner % 4 == 0):
// fused with the residual add and the dropout of the incoming branch.
//   a   [groups*inner]  branch output (dropout applied IN PLACE when p > 0)
//   x   residual input or NULL
//   r   pre-norm sum (kept for backward), y normalised output
//   stats  double[w2l_layernorm_scratch_doubles] scratch, meanRstd float[2*groups] (kept for backward)
W2L_API int w2l_residual_layernorm_forward(int groups, size_t inner, float* a, const float* x, float* r,
                                           float* y, const float* gammaBeta, float eps, double p,
                                           uint32_t seed, uint32_t rngStream, double* stats,
                                           float* meanRstd, w2l_stream_t stream) {
  if (groups <= 0 || inner == 0 || !a || !r || !y || !gammaBeta || !stats || !meanRstd) return W2L_EINVAL;
  const uint32_t thr = dropout_threshold(p);
  const float ks = (float)(1.0 / (1.0 - p));
  if (inner & 3) {  // odd group sizes (golden-vector geometry): scalar kernel, r must not alias a
    if (r == a) return W2L_EINVAL;
    hipLaunchKernelGGL(residual_ln_scalar_k, dim3((unsigned)groups), dim3(kEwThreads), 0, W2L_S, a, x, r, y, meanRstd,
                       inner, gammaBeta, eps, thr, ks, seed, rngStream);
    W2L_LAUNCH_CHECK();
    return W2L_OK;
  }
  if (inner / 4 <= 64 * (size_t)kLnWaveV && ln_wave_enabled()) {   // one wave per group
    const int nj = (int)((inner / 4 + 63) / 64);
    const dim3 grid((unsigned)((groups + kEwThreads / 64 - 1) / (kEwThreads / 64)));
#define W2L_LN_FWDW(NJ_)                                                                                                    \
  case NJ_:                                                                                                                 \
    hipLaunchKernelGGL(residual_ln_wave_k<NJ_>, grid, dim3(kEwThreads), 0, W2L_S, groups, a, x, r, y, meanRstd, inner,      \
                       gammaBeta, eps, thr, ks, seed, rngStream);                                                           \
    break
    switch (nj) {
      W2L_LN_FWDW(1); W2L_LN_FWDW(2); W2L_LN_FWDW(3); W2L_LN_FWDW(4); W2L_LN_FWDW(5); W2L_LN_FWDW(6); W2L_LN_FWDW(7); W2L_LN_FWDW(8);
      default: hipLaunchKernelGGL(residual_ln_wave_k<kLnWaveV>, grid, dim3(kEwThreads), 0, W2L_S, groups, a, x, r, y, meanRstd, inner,
                                  gammaBeta, eps, thr, ks, seed, rngStream);
    }
#undef W2L_LN_FWDW
    W2L_LAUNCH_CHECK();
    return W2L_OK;
  }
  if (inner <= kLnSmall) {
    const int nj = (int)((inner / 4 + kEwThreads - 1) / kEwThreads);
#define W2L_LN_FWD(NJ_)                                                                                                     \
  case NJ_:                                                                                                                 \
    hipLaunchKernelGGL(residual_ln_small_k<NJ_>, dim3((unsigned)groups), dim3(kEwThreads), 0, W2L_S, a, x, r, y, meanRstd,  \
                       inner, gammaBeta, eps, thr, ks, seed, rngStream);                                                    \
    break
    switch (nj) {
      W2L_LN_FWD(1); W2L_LN_FWD(2); W2L_LN_FWD(3); W2L_LN_FWD(4); W2L_LN_FWD(5); W2L_LN_FWD(6); W2L_LN_FWD(7);
      default: hipLaunchKernelGGL(residual_ln_small_k<kLnSmallV>, dim3((unsigned)groups), dim3(kEwThreads), 0, W2L_S, a, x, r, y, meanRstd,
                                  inner, gammaBeta, eps, thr, ks, seed, rngStream);
    }
#undef W2L_LN_FWD
    W2L_LAUNCH_CHECK();
    return W2L_OK;
  }
  double* part = stats + 2 * (size_t)groups;
  dim3 grid(ln_parts(inner), (unsigned)groups);
  hipLaunchKernelGGL(residual_dropout_stats_k, grid, dim3(kEwThreads), 0, W2L_S, a, x, r, part, inner, thr, ks, seed, rngStream);
  W2L_LAUNCH_CHECK();
  hipLaunchKernelGGL(ln_apply_k, grid, dim3(kEwThreads), 0, W2L_S, r, y, part, meanRstd, inner, gammaBeta, eps);
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}

// dr = LayerNorm backward of (r -> y) given dy; dGammaBeta[2] overwritten;
// if maskSrc: dmask = dr * (maskSrc > 0 ? maskScale : 0)
W2L_API int w2l_layernorm_backward(int groups, size_t inner, const float* r, const float* dy,
                                   const float* gammaBeta, const float* meanRstd, float* dr,
                                   float* dGammaBeta, const float* maskSrc, float* dmask, float maskScale,
                                   double* sums, w2l_stream_t stream) {
  if (groups <= 0 || inner == 0 || (inner & 3) || !r || !dy || !gammaBeta || !meanRstd || !dr || !sums)
    return W2L_EINVAL;
  if (inner / 4 <= 64 * (size_t)kLnWaveV && ln_wave_enabled()) {   // one wave per group
    const int nj = (int)((inner / 4 + 63) / 64);
    const dim3 grid((unsigned)((groups + kEwThreads / 64 - 1) / (kEwThreads / 64)));
#define W2L_LN_BWDW(NJ_)                                                                                                 \
  case NJ_:                                                                                                              \
    hipLaunchKernelGGL(ln_bwd_wave_k<NJ_>, grid, dim3(kEwThreads), 0, W2L_S, groups, r, dy, meanRstd, sums, gammaBeta,   \
                       dr, maskSrc, maskSrc ? dmask : nullptr, maskScale, inner);                                        \
    break
    switch (nj) {
      W2L_LN_BWDW(1); W2L_LN_BWDW(2); W2L_LN_BWDW(3); W2L_LN_BWDW(4); W2L_LN_BWDW(5); W2L_LN_BWDW(6); W2L_LN_BWDW(7); W2L_LN_BWDW(8);
      default: hipLaunchKernelGGL(ln_bwd_wave_k<kLnWaveV>, grid, dim3(kEwThreads), 0, W2L_S, groups, r, dy, meanRstd, sums, gammaBeta,
                                  dr, maskSrc, maskSrc ? dmask : nullptr, maskScale, inner);
    }
#undef W2L_LN_BWDW
    W2L_LAUNCH_CHECK();
  } else if (inner <= kLnSmall) {
    const int nj = (int)((inner / 4 + kEwThreads - 1) / kEwThreads);
#define W2L_LN_BWD(NJ_)                                                                                                  \
  case NJ_:                                                                                                              \
    hipLaunchKernelGGL(ln_bwd_small_k<NJ_>, dim3((unsigned)groups), dim3(kEwThreads), 0, W2L_S, r, dy, meanRstd, sums,   \
                       gammaBeta, dr, maskSrc, maskSrc ? dmask : nullptr, maskScale, inner);                             \
    break
    switch (nj) {
      W2L_LN_BWD(1); W2L_LN_BWD(2); W2L_LN_BWD(3); W2L_LN_BWD(4); W2L_LN_BWD(5); W2L_LN_BWD(6); W2L_LN_BWD(7);
      default: hipLaunchKernelGGL(ln_bwd_small_k<kLnSmallV>, dim3((unsigned)groups), dim3(kEwThreads), 0, W2L_S, r, dy, meanRstd, sums,
                                  gammaBeta, dr, maskSrc, maskSrc ? dmask : nullptr, maskScale, inner);
    }
#undef W2L_LN_BWD
    W2L_LAUNCH_CHECK();
  } else {
    double* part = sums + 2 * (size_t)groups;
    dim3 grid(ln_parts(inner), (unsigned)groups);
    hipLaunchKernelGGL(ln_bwd_reduce_k, grid, dim3(kEwThreads), 0, W2L_S, r, dy, meanRstd, part, inner);
    W2L_LAUNCH_CHECK();
    hipLaunchKernelGGL(ln_bwd_apply_k, grid, dim3(kEwThreads), 0, W2L_S, r, dy, meanRstd, part, sums, gammaBeta, dr,
                       maskSrc, maskSrc ? dmask : nullptr, maskScale, inner);
    W2L_LAUNCH_CHECK();
  }
  if (dGammaBeta) {
    hipLaunchKernelGGL(ln_param_grad_k, dim3(1), dim3(1024), 0, W2L_S, sums, groups, dGammaBeta);
    W2L_LAUNCH_CHECK();
  }
  return W2L_OK;
}

__global__ __launch_bounds__(kEwThreads) void dropout_copy_k(float* __restrict__ y, const float* __restrict__ x, size_t n,
                                                            uint32_t thr, float keepScale, uint32_t seed, uint32_t stream) {
  const size_t n4 = n >> 2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const size_t e = 4 * i;
    float4 v = *(const float4*)(x + e);
    v.x = keep_elem(e, seed, stream, thr) ? v.x * keepScale : 0.f;
    v.y = keep_elem(e + 1, seed, stream, thr) ? v.y * keepScale : 0.f;
    v.z = keep_elem(e + 2, seed, stream, thr) ? v.z * keepScale : 0.f;
    v.w = keep_elem(e + 3, seed, stream, thr) ? v.w * keepScale : 0.f;
    *(float4*)(y + e) = v;
  }
  for (size_t e = (n4 << 2) + (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x)
    y[e] = keep_elem(e, seed, stream, thr) ? x[e] * keepScale : 0.f;
}

// y = dropout(x), out of place (same mask as w2l_dropout_inplace): one pass instead of copy + in-place pass
W2L_API int w2l_dropout_copy(float* y, const float* x, size_t n, double p, uint32_t seed, uint32_t rngStream,
                             w2l_stream_t stream) {
  if (!x || !y || ((((uintptr_t)x) | ((uintptr_t)y)) & 15)) return W2L_EINVAL;
  const uint32_t thr = dropout_threshold(p);
  if (!n) return W2L_OK;
  hipLaunchKernelGGL(dropout_copy_k, dim3(ew_grid((n >> 2) + 1)), dim3(kEwThreads), 0, W2L_S, y, x, n, thr,
                     (float)(1.0 / (1.0 - p)), seed, rngStream);
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}

W2L_API int w2l_dropout_inplace(float* x, size_t n, double p, uint32_t seed, uint32_t rngStream,
                                w2l_stream_t stream) {
  if (!x) return W2L_EINVAL;
  const uint32_t thr = dropout_threshold(p);
  if (!thr || !n) return W2L_OK;
  hipLaunchKernelGGL(dropout_k, dim3(ew_grid((n >> 2) + 1)), dim3(kEwThreads), 0, W2L_S, x, n, thr,
                     (float)(1.0 / (1.0 - p)), seed, rngStream);
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}

W2L_API int w2l_mask_backward(const float* dy, const float* src, float* dx, size_t n, float scale,
                              w2l_stream_t stream) {
  if (!dy || !src || !dx) return W2L_EINVAL;
  if (!n) return W2L_OK;
  hipLaunchKernelGGL(mask_bwd_k, dim3(ew_grid((n >> 2) + 1)), dim3(kEwThreads), 0, W2L_S, dy, src, dx, n, scale);
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}

// Conv2D with a kh x kw kernel (kh > 1: the librivox TDS arch, recipes/sota/2019/am_arch/am_tds_ctc_librivox.arch:3) as a
// kw x 1 convolution over kh*C channels: the mel axis is unrolled into the channels of each mel row,
//   xe[r][h][dh*C + c] = x[r][h + dh - padh][c]   (zero outside [0, H));   r = (utterance, frame)
__global__ __launch_bounds__(256) void hexpand_k(const float* __restrict__ x, float* __restrict__ xe, size_t n, int H, int C, int kh, int padh) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int c = (int)(i % C);
    size_t q = i / C;
    const int dh = (int)(q % kh); q /= kh;
    const int h = (int)(q % H);
    const size_t r = q / H;
    const int hs = h + dh - padh;
    xe[i] = hs >= 0 && hs < H ? x[(r * H + hs) * C + c] : 0.f;
  }
}
// the adjoint: dx[r][h][c] = sum_dh dxe[r][h - dh + padh][dh*C + c]
__global__ __launch_bounds__(256) void hfold_k(const float* __restrict__ dxe, float* __restrict__ dx, size_t n, int H, int C, int kh, int padh) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int c = (int)(i % C);
    size_t q = i / C;
    const int h = (int)(q % H);
    const size_t r = q / H;
    float s = 0.f;
    for (int dh = 0; dh < kh; ++dh) {
      const int hd = h - dh + padh;
      if (hd >= 0 && hd < H) s += dxe[((r * H + hd) * kh + dh) * C + c];
    }
    dx[i] = s;
  }
}

W2L_API int w2l_hexpand_forward(const float* x, float* xe, size_t rows, int H, int C, int kh, int padh, w2l_stream_t stream) {
  if (!x || !xe || H < 1 || C < 1 || kh < 1 || padh < 0 || padh >= kh) return W2L_EINVAL;
  const size_t n = rows * H * kh * C;
  if (!n) return W2L_OK;
  hipLaunchKernelGGL(hexpand_k, dim3(ew_grid(n)), dim3(kEwThreads), 0, W2L_S, x, xe, n, H, C, kh, padh);
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}

W2L_API int w2l_hexpand_backward(const float* dxe, float* dx, size_t rows, int H, int C, int kh, int padh, w2l_stream_t stream) {
  if (!dxe || !dx || H < 1 || C < 1 || kh < 1 || padh < 0 || padh >= kh) return W2L_EINVAL;
  const size_t n = rows * H * C;
  if (!n) return W2L_OK;
  hipLaunchKernelGGL(hfold_k, dim3(ew_grid(n)), dim3(kEwThreads), 0, W2L_S, dxe, dx, n, H, C, kh, padh);
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}

W2L_API int w2l_axpy(float* y, const float* x, size_t n, float alpha, w2l_stream_t stream) {
  if (!y || !x) return W2L_EINVAL;
  if (!n) return W2L_OK;
  hipLaunchKernelGGL(axpy_k, dim3(ew_grid((n >> 2) + 1)), dim3(kEwThreads), 0, W2L_S, y, x, n, alpha);
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}

W2L_API int w2l_fill(float* y, size_t n, float v, w2l_stream_t stream) {
  if (!y) return W2L_EINVAL;
  if (!n) return W2L_OK;
  hipLaunchKernelGGL(fill_k, dim3(ew_grid(n)), dim3(kEwThreads), 0, W2L_S, y, n, v);
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}

W2L_API int w2l_transpose(const float* in, float* out, int G, int R, int Cc, w2l_stream_t stream) {
  if (!in || !out || G <= 0 || R <= 0 || Cc <= 0) return W2L_EINVAL;
  dim3 grid((unsigned)((Cc + 31) / 32), (unsigned)((R + 31) / 32), (unsigned)G);
  hipLaunchKernelGGL(transpose_k, grid, dim3(256), 0, W2L_S, in, out, R, Cc);
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}

W2L_API int w2l_glu_forward(const float* x, float* y, size_t M, int half, w2l_stream_t stream) {
  if (!x || !y || half <= 0) return W2L_EINVAL;
  hipLaunchKernelGGL(glu_fwd_k, dim3(ew_grid(M * half)), dim3(kEwThreads), 0, W2L_S, x, y, M, half);
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}

W2L_API int w2l_glu_backward(const float* x, const float* dy, float* dx, size_t M, int half, w2l_stream_t stream) {
  if (!x || !dy || !dx || half <= 0) return W2L_EINVAL;
  hipLaunchKernelGGL(glu_bwd_k, dim3(ew_grid(M * half)), dim3(kEwThreads), 0, W2L_S, x, dy, dx, M, half);
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}

W2L_API int w2l_sumsq(const float* g, size_t n, double* out, int zeroFirst, w2l_stream_t stream) {
  if (!g || !out) return W2L_EINVAL;
  if (!n) {
    if (zeroFirst) W2L_HIP_CHECK(hipMemsetAsync(out, 0, sizeof(double), W2L_S));
    return W2L_OK;
  }
  const unsigned grid = ew_grid((n >> 2) + 1);
  double* part = (double*)sk_scratch(W2L_S, kSkScratchBytes);   // the stream's scratch (work on a stream is serialised)
  if (!part) return W2L_EHIP;
  hipLaunchKernelGGL(sumsq_k, dim3(grid), dim3(kEwThreads), 0, W2L_S, g, n, part);
  hipLaunchKernelGGL(sumsq_finish_k, dim3(1), dim3(256), 0, W2L_S, part, (int)grid, out, zeroFirst ? 0 : 1);
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}

W2L_API int w2l_sgd_step(float* p, const float* g, float* v, size_t n, float lr, float momentum,
                         float gradScale, float maxGradNorm, const double* sumsq, w2l_stream_t stream) {
  if (!p || !g || (momentum != 0.f && !v) || (maxGradNorm > 0.f && !sumsq)) return W2L_EINVAL;
  if (!n) return W2L_OK;
  hipLaunchKernelGGL(sgd_k, dim3(ew_grid((n >> 2) + 1)), dim3(kEwThreads), 0, W2L_S, p, g, v, n, lr, momentum,
                     gradScale, maxGradNorm, sumsq, 0);
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}

W2L_API int w2l_grad_guard(double* acc, const float* batchDev, int clampCrit, w2l_stream_t stream) {
  if (!acc) return W2L_EINVAL;
  hipLaunchKernelGGL(grad_guard_k, dim3(1), dim3(1), 0, W2L_S, acc, batchDev, clampCrit);
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}

W2L_API int w2l_sgd_step_guarded(float* p, const float* g, float* v, size_t n, float lr, float momentum,
                                 float gradScale, float maxGradNorm, const double* guard, w2l_stream_t stream) {
  if (!p || !g || (momentum != 0.f && !v) || !guard) return W2L_EINVAL;
  if (!n) return W2L_OK;
  hipLaunchKernelGGL(sgd_k, dim3(ew_grid((n >> 2) + 1)), dim3(kEwThreads), 0, W2L_S, p, g, v, n, lr, momentum,
                     gradScale, maxGradNorm, guard, 1);
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}

W2L_API int w2l_adagrad_step_guarded(float* p, const float* g, float* var, size_t n, float lr, float eps, float gradScale,
                                     float maxGradNorm, const double* guard, w2l_stream_t stream) {
  if (!p || !g || !var || (maxGradNorm > 0.f && !guard)) return W2L_EINVAL;
  if (!n) return W2L_OK;
  hipLaunchKernelGGL(adagrad_k, dim3(ew_grid((n >> 2) + 1)), dim3(kEwThreads), 0, W2L_S, p, g, var, n, lr, eps, gradScale,
                     maxGradNorm, guard, guard ? 1 : 0);
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}

W2L_API int w2l_adadelta_step_guarded(float* p, const float* g, float* accGrad, float* accDelta, size_t n, float lr, float rho,
                                      float eps, float gradScale, float maxGradNorm, const double* guard, w2l_stream_t stream) {
  if (!p || !g || !accGrad || !accDelta || (maxGradNorm > 0.f && !guard)) return W2L_EINVAL;
  if (!n) return W2L_OK;
  hipLaunchKernelGGL(adadelta_k, dim3(ew_grid((n >> 2) + 1)), dim3(kEwThreads), 0, W2L_S, p, g, accGrad, accDelta, n, lr, rho, eps,
                     gradScale, maxGradNorm, guard, guard ? 1 : 0);
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}
