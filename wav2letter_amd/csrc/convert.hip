// convert.hip -- fp32 -> bf16 operand copies of the mixed-precision mode (BASELINE configs 3 / 5; fl's AMP casts the
// operands of linear / conv to half precision and keeps fp32 master weights: recipes/slimIPL/src/Train.cpp:211,
// :1681-1760, recipes/joint_training_vox_populi/cpc/Train.cpp:365, :1184).
//
// One pass over an fp32 matrix x [rows][cols] (leading dimension ldx) writes up to two bf16 images
//   rowMajor   [rows][ldRows]    rowMajor[r][c]   = bf16(x[r][c]),  columns cols .. ldRows-1 ZERO
//   transposed [cols][ldTrans]   transposed[c][r] = bf16(x[r][c]),  columns rows .. ldTrans-1 ZERO
// (round to nearest even, v_cvt_pk_bf16_f32).  The zero padding is what lets the bf16 GEMM (gemm_bf16g.hpp) run whole
// 64-k tiles without a tail: ldRows / ldTrans are the reduction lengths rounded up to 64.  The transposed image is the
// k-contiguous operand of the weight-gradient product dW = x^T dy (reduction over the rows of x) and of the forward
// product against a weight stored [in][out].
// HBM-bound: 4 bytes read, 2 (+2) bytes written per element.
#include "common.hpp"

namespace w2l {

typedef __bf16 cv_bf16x2_t __attribute__((ext_vector_type(2)));
typedef float cv_f32x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t cv_pack2(float a, float b) {
  const cv_f32x2_t v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, cv_bf16x2_t));
}

constexpr int kCvTile = 64;

// one 64 x 64 tile (bx, by) of x; 256 threads
// DROP: the images are those of dropout(x) -- element (r, c) kept (and scaled) by the library's stateless hash of r * ldx + c,
// the mask w2l_dropout_copy over the dense [rows][ldx] matrix applies (the masked copy is never materialised)
struct CvDrop { uint32_t thr, seed, stream; float scale; };
template <bool DROP = false>
__device__ __forceinline__ void cvt_tile(const float* __restrict__ x, size_t rows, int cols, size_t ldx,
                                         uint16_t* __restrict__ rowMajor, size_t ldRows,
                                         uint16_t* __restrict__ transposed, size_t ldTrans, unsigned bx, unsigned by,
                                         CvDrop dr = CvDrop{0u, 0u, 0u, 1.f}) {
  __shared__ float tile[kCvTile][kCvTile + 1];
  const size_t r0 = (size_t)by * kCvTile;
  const int c0 = (int)bx * kCvTile;
  const int tid = threadIdx.x;
  const int cq = (tid & 15) * 4, rr = tid >> 4;   // 4 columns at c0 + cq, rows rr + 16 i
  const bool vec = (((uintptr_t)x) & 15) == 0 && (ldx & 3) == 0;
  // Interior tiles (every tile but the last row / column of tiles): all four loads of the thread are issued, and have LANDED at one
  // unconditional point, before the first store.  On gfx9 stores count in vmcnt too: with load i + 1 issued behind store i, hipcc's
  // `s_waitcnt vmcnt(0)` in front of the loaded value also waited for the store's round trip -- four serialised round trips per
  // tile (ISA of the round-3 kernel); and any per-element bounds branch between the loads makes the compiler wait after each.
  const bool interior = vec && r0 + kCvTile <= rows && c0 + kCvTile <= cols;   // (uniform over the workgroup)
  if (interior) {
    const int c = c0 + cq;
    float4 q[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = *(const float4*)(x + (r0 + rr + 16 * i) * ldx + c);
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(q[i].x), "+v"(q[i].y), "+v"(q[i].z), "+v"(q[i].w));
    if (DROP) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint64_t idx = (uint64_t)(r0 + rr + 16 * i) * ldx + c;
        q[i].x = keep_elem(idx, dr.seed, dr.stream, dr.thr) ? q[i].x * dr.scale : 0.f;
        q[i].y = keep_elem(idx + 1, dr.seed, dr.stream, dr.thr) ? q[i].y * dr.scale : 0.f;
        q[i].z = keep_elem(idx + 2, dr.seed, dr.stream, dr.thr) ? q[i].z * dr.scale : 0.f;
        q[i].w = keep_elem(idx + 3, dr.seed, dr.stream, dr.thr) ? q[i].w * dr.scale : 0.f;
      }
    }
    if (transposed) {   // (the barrier drains vmcnt: no store may be in flight in front of it)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        tile[rr + 16 * i][cq] = q[i].x; tile[rr + 16 * i][cq + 1] = q[i].y;
        tile[rr + 16 * i][cq + 2] = q[i].z; tile[rr + 16 * i][cq + 3] = q[i].w;
      }
      __syncthreads();
    }
    if (rowMajor) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        *(uint2*)(rowMajor + (r0 + rr + 16 * i) * ldRows + c) = make_uint2(cv_pack2(q[i].x, q[i].y), cv_pack2(q[i].z, q[i].w));
    }
    if (transposed) {   // thread -> column c0 + (tid >> 2), 16 consecutive rows: 32 bytes (interior: inside cols and ldTrans)
      uint32_t p[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) p[e] = cv_pack2(tile[16 * (tid & 3) + 2 * e][tid >> 2], tile[16 * (tid & 3) + 2 * e + 1][tid >> 2]);
      uint4* dst = (uint4*)(transposed + (size_t)(c0 + (tid >> 2)) * ldTrans + r0 + 16 * (tid & 3));
      dst[0] = make_uint4(p[0], p[1], p[2], p[3]);
      dst[1] = make_uint4(p[4], p[5], p[6], p[7]);
    }
    return;
  } else {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const size_t r = r0 + rr + 16 * i;
    const int c = c0 + cq;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (r < rows) {
      const float* src = x + r * ldx + c;
      if (vec && c + 3 < cols) {
        const float4 q = *(const float4*)src;
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (c + e < cols) v[e] = src[e];
      }
      if (DROP) {
        const uint64_t idx = (uint64_t)r * ldx + c;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = keep_elem(idx + e, dr.seed, dr.stream, dr.thr) ? v[e] * dr.scale : 0.f;
      }
    }
    if (rowMajor && r < rows && (size_t)c < ldRows) {   // ldRows % 4 == 0 (host-checked): whole 8-byte stores
      const uint2 q = make_uint2(cv_pack2(v[0], v[1]), cv_pack2(v[2], v[3]));
      *(uint2*)(rowMajor + r * ldRows + c) = q;
    }
    if (transposed) {
#pragma unroll
      for (int e = 0; e < 4; ++e) tile[rr + 16 * i][cq + e] = v[e];
    }
  }
  }
  if (!transposed) return;
  __syncthreads();
  // transposed image: thread -> column c0 + (tid >> 2), 16 consecutive rows r0 + 16 (tid & 3) .. + 16 = 32 bytes
  const int c = c0 + (tid >> 2);
  const size_t rb = r0 + 16 * (tid & 3);
  if (c >= cols || rb >= ldTrans) return;   // ldTrans % 16 == 0 (host-checked): whole 32-byte runs
  uint32_t p[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) p[e] = cv_pack2(tile[16 * (tid & 3) + 2 * e][tid >> 2], tile[16 * (tid & 3) + 2 * e + 1][tid >> 2]);
  uint4* dst = (uint4*)(transposed + (size_t)c * ldTrans + rb);
  dst[0] = make_uint4(p[0], p[1], p[2], p[3]);
  dst[1] = make_uint4(p[4], p[5], p[6], p[7]);
}

// grid (ceil(max(cols, ldRows) / 64), ceil(max(rows, ldTrans) / 64))
__global__ __launch_bounds__(256) void cvt_bf16_k(const float* __restrict__ x, size_t rows, int cols, size_t ldx,
                                                  uint16_t* __restrict__ rowMajor, size_t ldRows,
                                                  uint16_t* __restrict__ transposed, size_t ldTrans) {
  cvt_tile(x, rows, cols, ldx, rowMajor, ldRows, transposed, ldTrans, blockIdx.x, blockIdx.y);
}

__global__ __launch_bounds__(256) void cvt_bf16_drop_k(const float* __restrict__ x, size_t rows, int cols, size_t ldx,
                                                       uint16_t* __restrict__ rowMajor, size_t ldRows,
                                                       uint16_t* __restrict__ transposed, size_t ldTrans, CvDrop dr) {
  cvt_tile<true>(x, rows, cols, ldx, rowMajor, ldRows, transposed, ldTrans, blockIdx.x, blockIdx.y, dr);
}

// several matrices in one launch (the six weights of a Transformer block are 8 us conversions each: launch-bound one at a
// time): the tiles of all of them are laid end to end over a 1-D grid
constexpr int kCvMaxMulti = 8;
struct CvMulti {
  w2l_bf16_convert_desc d[kCvMaxMulti];
  unsigned first[kCvMaxMulti + 1];   // first tile of matrix i
  unsigned gx[kCvMaxMulti];          // tiles per tile row of matrix i
  int n;
};
__global__ __launch_bounds__(256) void cvt_bf16_multi_k(CvMulti m) {
  int i = 0;
  while (i + 1 < m.n && blockIdx.x >= m.first[i + 1]) ++i;
  const unsigned t = blockIdx.x - m.first[i];
  const w2l_bf16_convert_desc& d = m.d[i];
  cvt_tile(d.x, d.rows, d.cols, d.ldx, d.rowMajor, d.ldRows, d.transposed, d.ldTrans, t % m.gx[i], t / m.gx[i]);
}

}  // namespace w2l

using namespace w2l;

// x [rows][cols] fp32 (leading dimension ldx) -> rowMajor [rows][ldRows] and / or transposed [cols][ldTrans] bf16, zero
// padded (see the file header).  Either output may be null.  ldRows >= cols and ldTrans >= rows, both multiples of 16;
// outputs 16-byte aligned.
W2L_API int w2l_bf16_convert(const float* x, size_t rows, int cols, size_t ldx, uint16_t* rowMajor, size_t ldRows,
                             uint16_t* transposed, size_t ldTrans, w2l_stream_t stream) {
  if (!x || rows == 0 || cols <= 0 || ldx < (size_t)cols || (!rowMajor && !transposed)) return W2L_EINVAL;
  if (rowMajor && (ldRows < (size_t)cols || (ldRows & 15) || (((uintptr_t)rowMajor) & 15))) return W2L_EINVAL;
  if (transposed && (ldTrans < rows || (ldTrans & 15) || (((uintptr_t)transposed) & 15))) return W2L_EINVAL;
  const size_t spanC = rowMajor && ldRows > (size_t)cols ? ldRows : (size_t)cols;
  const size_t spanR = transposed && ldTrans > rows ? ldTrans : rows;
  const dim3 grid((unsigned)((spanC + kCvTile - 1) / kCvTile), (unsigned)((spanR + kCvTile - 1) / kCvTile));
  if (grid.y > 65535u) return W2L_EUNSUPPORTED;
  hipLaunchKernelGGL(cvt_bf16_k, grid, dim3(256), 0, (hipStream_t)stream, x, rows, cols, ldx, rowMajor, ldRows, transposed, ldTrans);
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}

// the images of dropout(x) (the mask of w2l_dropout_copy / w2l_dropout_inplace with the same p, seed, rngStream over the dense
// [rows][ldx] matrix): the backward pass of a dropout layer whose masked gradient is only ever a GEMM operand -- one pass over
// the gradient instead of a masked copy + a conversion
W2L_API int w2l_bf16_convert_dropout(const float* x, size_t rows, int cols, size_t ldx, uint16_t* rowMajor, size_t ldRows,
                                     uint16_t* transposed, size_t ldTrans, double p, uint32_t seed, uint32_t rngStream,
                                     w2l_stream_t stream) {
  if (!(p > 0.0)) return w2l_bf16_convert(x, rows, cols, ldx, rowMajor, ldRows, transposed, ldTrans, stream);
  if (!x || rows == 0 || cols <= 0 || ldx < (size_t)cols || (!rowMajor && !transposed) || p >= 1.0) return W2L_EINVAL;
  if (rowMajor && (ldRows < (size_t)cols || (ldRows & 15) || (((uintptr_t)rowMajor) & 15))) return W2L_EINVAL;
  if (transposed && (ldTrans < rows || (ldTrans & 15) || (((uintptr_t)transposed) & 15))) return W2L_EINVAL;
  const size_t spanC = rowMajor && ldRows > (size_t)cols ? ldRows : (size_t)cols;
  const size_t spanR = transposed && ldTrans > rows ? ldTrans : rows;
  const dim3 grid((unsigned)((spanC + kCvTile - 1) / kCvTile), (unsigned)((spanR + kCvTile - 1) / kCvTile));
  if (grid.y > 65535u) return W2L_EUNSUPPORTED;
  const CvDrop dr{dropout_threshold(p), seed, rngStream, (float)(1.0 / (1.0 - p))};
  hipLaunchKernelGGL(cvt_bf16_drop_k, grid, dim3(256), 0, (hipStream_t)stream, x, rows, cols, ldx, rowMajor, ldRows, transposed, ldTrans, dr);
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}

static int cv_check(const w2l_bf16_convert_desc& d) {
  if (!d.x || d.rows == 0 || d.cols <= 0 || d.ldx < (size_t)d.cols || (!d.rowMajor && !d.transposed)) return W2L_EINVAL;
  if (d.rowMajor && (d.ldRows < (size_t)d.cols || (d.ldRows & 15) || (((uintptr_t)d.rowMajor) & 15))) return W2L_EINVAL;
  if (d.transposed && (d.ldTrans < d.rows || (d.ldTrans & 15) || (((uintptr_t)d.transposed) & 15))) return W2L_EINVAL;
  return W2L_OK;
}

// n (1 .. 8) conversions in one launch; each entry as the arguments of w2l_bf16_convert
W2L_API int w2l_bf16_convert_multi(int n, const w2l_bf16_convert_desc* descs, w2l_stream_t stream) {
  if (n < 1 || n > kCvMaxMulti || !descs) return W2L_EINVAL;
  CvMulti m;
  m.n = n;
  unsigned long long total = 0;
  for (int i = 0; i < n; ++i) {
    const w2l_bf16_convert_desc& d = descs[i];
    const int st = cv_check(d);
    if (st != W2L_OK) return st;
    const size_t spanC = d.rowMajor && d.ldRows > (size_t)d.cols ? d.ldRows : (size_t)d.cols;
    const size_t spanR = d.transposed && d.ldTrans > d.rows ? d.ldTrans : d.rows;
    m.d[i] = d;
    m.first[i] = (unsigned)total;
    m.gx[i] = (unsigned)((spanC + kCvTile - 1) / kCvTile);
    total += (unsigned long long)m.gx[i] * ((spanR + kCvTile - 1) / kCvTile);
    if (total >= 0x7fffffffull) return W2L_EUNSUPPORTED;
  }
  m.first[n] = (unsigned)total;
  for (int i = n; i < kCvMaxMulti; ++i) { m.d[i] = m.d[0]; m.gx[i] = 1; m.first[i + 1] = (unsigned)total; }
  hipLaunchKernelGGL(cvt_bf16_multi_k, dim3((unsigned)total), dim3(256), 0, (hipStream_t)stream, m);
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}
