// gemm_bf16.hpp -- mixed-precision GEMM for BASELINE config 3 (streaming_convnets LibriSpeech TDS-CTC, bf16): fp32
// operands in HBM (master weights and activations stay fp32, like fl's AMP keeps fp32 master parameters:
// recipes/joint_training_vox_populi/cpc/Train.cpp:1184 casts the criterion input back to f32), converted to bf16 with
// the hardware's round-to-nearest-even v_cvt_pk_bf16_f32 on the way into LDS, multiplied on
// v_mfma_f32_32x32x16_bf16 (fp32 accumulate, 16x the fp32 matrix rate), fp32 epilogue = the fp32 engine's
// (bias / ReLU / dropout / mask / addend: gemm128_epilogue, same D layout).
//
// Tile 128 x 128 x 32, 4 waves x (2 x 2) 32x32 accumulators.  LDS image is [row][k] bf16 with k contiguous (64-byte
// rows, 16-byte chunks XOR-swizzled by the row: conflict-free ds_read_b128 fragments of 8 k), double buffered; the next
// K tile's global loads are in flight under the current tile's MFMAs.  A k-contiguous source stores 4 k at a time
// (ds_write_b64); a k-row source (w of the forward, both operands of dW = x^T dy) loads k pairs and stores packed
// (k, k+1) dwords.  Long reductions with few tiles (dW) are
// split over blockIdx.z into partial slabs that a second kernel adds in slice order (deterministic).
//
// Bound: with fp32 operands a 128 x 128 x 32 tile moves 32 KiB for 1 MFLOP: the kernel is L2 / LDS-staging bound
// long before the 2.5 PFLOP/s matrix peak -- the point of this first version is the 16x cheaper multiply at fp32
// storage, not the bf16 roofline (bf16 activation storage is the next step).
#pragma once
#include "gemm.hpp"

namespace w2l {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2v_t __attribute__((ext_vector_type(2)));

constexpr int kBfRowDw = 16;                     // dwords per LDS row: 32 bf16, no padding
constexpr int kBfTileDw = 128 * kBfRowDw;        // one operand tile, in dwords
constexpr size_t kBfLdsBytes = 4 * (size_t)kBfTileDw * 4;   // A/B x two stages = 32 KiB

__device__ __forceinline__ uint32_t bf_pack2(float a, float b) {
  const f32x2v_t v = {a, b};
  const bf16x2_t h = __builtin_convertvector(v, bf16x2_t);   // v_cvt_pk_bf16_f32: round to nearest even
  return __builtin_bit_cast(uint32_t, h);
}

// LDS image of an operand tile: row i (0..127) = 32 bf16 of k = four 16-byte chunks; chunk c of row i sits at chunk
// position c ^ ((i >> 2) & 3).  With an unpadded 64-byte pitch that XOR is what makes the ds_read_b128 fragment reads
// (16 consecutive rows, one chunk each) hit 16 distinct 4-bank groups, and it keeps the k-contiguous 8-byte stores and
// the k-row 4-byte stores at no more than 4-way (a padded pitch must stay a multiple of 16 bytes for the b128 reads,
// which puts rows 4 apart on the same banks: the first version's scalar bf16 stores ran 32-way conflicted and the
// whole kernel at 133 TF/s "equivalent", profiles/r02_run11_bf16_first.log).
__device__ __forceinline__ int bf_chunk(int row, int c) { return c ^ ((row >> 2) & 3); }

// Operand op(k, i): KC = true: element at p[i*ld + k] (k contiguous), false: p[k*ld + i] ("k-rows").  vec: float4-able.
template <bool KC>
struct BfOp {
  const float* p;
  int ld, extent, K;
  bool vec;
  // 16 floats per thread of the [32 k][128 i] tile
  __device__ __forceinline__ void load(float (&r)[16], int i0, int k0, int tid) const {
    if (KC) {
      const int kq = tid & 7, rr = tid >> 3;           // 4 k at k0 + 4 kq, rows rr + 32 j
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int gi = i0 + rr + 32 * j, gk = k0 + 4 * kq;
        const float* src = p + (size_t)(gi < extent ? gi : extent - 1) * ld + gk;
        if (vec && gi < extent && gk + 3 < K) {
          const float4 v = *(const float4*)src;
          r[4 * j] = v.x; r[4 * j + 1] = v.y; r[4 * j + 2] = v.z; r[4 * j + 3] = v.w;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) r[4 * j + e] = (gi < extent && gk + e < K) ? src[e] : 0.f;
        }
      }
    } else {
      const int iq = tid & 31, kp = tid >> 5;          // 4 i at i0 + 4 iq, k pairs kp and kp + 8
      const int gi = i0 + 4 * iq;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int gk = k0 + 2 * (kp + 8 * j) + h;
          const float* src = p + (size_t)(gk < K ? gk : K - 1) * ld + gi;
          if (vec && gk < K && gi + 3 < extent) {
            const float4 v = *(const float4*)src;
            r[8 * j + 4 * h] = v.x; r[8 * j + 4 * h + 1] = v.y; r[8 * j + 4 * h + 2] = v.z; r[8 * j + 4 * h + 3] = v.w;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) r[8 * j + 4 * h + e] = (gk < K && gi + e < extent) ? src[e] : 0.f;
          }
        }
    }
  }
  __device__ __forceinline__ void store(uint32_t* lds, const float (&r)[16], int tid) const {
    if (KC) {
      const int kq = tid & 7, rr = tid >> 3;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = rr + 32 * j;
        uint32_t* d = lds + row * kBfRowDw + 4 * bf_chunk(row, kq >> 1) + 2 * (kq & 1);
        const uint2 q = make_uint2(bf_pack2(r[4 * j], r[4 * j + 1]), bf_pack2(r[4 * j + 2], r[4 * j + 3]));
        *(uint2*)d = q;
      }
    } else {
      const int iq = tid & 31, kp = tid >> 5;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int kpair = kp + 8 * j;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int row = 4 * iq + e;
          lds[row * kBfRowDw + 4 * bf_chunk(row, kpair >> 2) + (kpair & 3)] = bf_pack2(r[8 * j + e], r[8 * j + 4 + e]);
        }
      }
    }
  }
};

// grid (tilesN, tilesM, splitK); splitK > 1: partial[z][M][N] (plain row-major fp32), no epilogue
template <bool AKC, bool BKC>
__global__ __launch_bounds__(256, 2) void gemm128_bf16_kernel(BfOp<AKC> aop, BfOp<BKC> bop, GemmOut out, int ktPer, float* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smemRaw[];
  uint32_t* smem = (uint32_t*)smemRaw;
  uint32_t* As[2] = {smem, smem + 2 * kBfTileDw};
  uint32_t* Bs[2] = {smem + kBfTileDw, smem + 3 * kBfTileDw};
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  const int li = lane & 31, lh = lane >> 5;
  const int m0 = blockIdx.y * 128, n0 = blockIdx.x * 128;
  const int ktAll = (out.K + 31) / 32;
  const int ktBegin = blockIdx.z * ktPer;
  int ktEnd = ktBegin + ktPer;
  if (ktEnd > ktAll) ktEnd = ktAll;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float ra[16], rb[16];
  if (ktBegin < ktEnd) {
    aop.load(ra, m0, ktBegin * 32, tid);
    bop.load(rb, n0, ktBegin * 32, tid);
    aop.store(As[0], ra, tid);
    bop.store(Bs[0], rb, tid);
  }
  __syncthreads();
  // fragment addresses (dwords): rows wm + li (+32), logical chunk 2 ks + lh
  const int ra0 = wm + li, ra1 = ra0 + 32, rb0 = wn + li, rb1 = rb0 + 32;
  for (int kt = ktBegin; kt < ktEnd; ++kt) {
    const int cur = (kt - ktBegin) & 1;
    const bool more = kt + 1 < ktEnd;
    if (more) {
      aop.load(ra, m0, (kt + 1) * 32, tid);
      bop.load(rb, n0, (kt + 1) * 32, tid);
    }
    const uint32_t* At = As[cur];
    const uint32_t* Bt = Bs[cur];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int c = 2 * ks + lh;
      const bf16x8_t a0 = *(const bf16x8_t*)(At + ra0 * kBfRowDw + 4 * bf_chunk(ra0, c));
      const bf16x8_t a1 = *(const bf16x8_t*)(At + ra1 * kBfRowDw + 4 * bf_chunk(ra1, c));
      const bf16x8_t b0 = *(const bf16x8_t*)(Bt + rb0 * kBfRowDw + 4 * bf_chunk(rb0, c));
      const bf16x8_t b1 = *(const bf16x8_t*)(Bt + rb1 * kBfRowDw + 4 * bf_chunk(rb1, c));
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
    }
    if (more) {
      aop.store(As[cur ^ 1], ra, tid);
      bop.store(Bs[cur ^ 1], rb, tid);
    }
    __syncthreads();
  }
  if (!partial) {
    gemm128_epilogue(out, m0, n0, acc);
  } else {
    float* dst = partial + (size_t)blockIdx.z * out.M * out.N;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn + j * 32 + li;
        if (n >= out.N) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (m < out.M) dst[(size_t)m * out.N + n] = acc[i][j][r];
        }
      }
  }
}

// C[m][n] = sum_z partial[z][m][n] (slice order), n a multiple of 4 not required
__global__ __launch_bounds__(256) void gemm_bf16_sum_k(const float* __restrict__ partial, int S, size_t MN, int N, float* __restrict__ C, int ldc) {
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < MN; e += (size_t)gridDim.x * 256) {
    float s = 0.f;
    for (int z = 0; z < S; ++z) s += partial[(size_t)z * MN + e];
    const size_t m = e / N, n = e - m * N;
    C[m * ldc + n] = s;
  }
}

float* sk_scratch(hipStream_t s, size_t bytes);

template <bool AKC, bool BKC>
inline int launch128_bf16(const BfOp<AKC>& a, const BfOp<BKC>& b, GemmOut o, int epi, hipStream_t s) {
  o.epi = epi & ~EPI_ATOMIC;
  const int tilesM = (o.M + 127) / 128, tilesN = (o.N + 127) / 128, tiles = tilesM * tilesN;
  const int ktAll = (o.K + 31) / 32;
  int splitK = 1;
  float* partial = nullptr;
  // few tiles, long reduction, no epilogue (the weight gradients): split K so that ~2 workgroups per CU are busy
  if (o.epi == 0 && !o.rowPin && tiles < 256 && ktAll >= 64) {
    splitK = (512 + tiles - 1) / tiles;
    if (splitK > ktAll / 16) splitK = ktAll / 16;
    if (splitK > 16) splitK = 16;
    if (splitK > 1 && (size_t)splitK * o.M * o.N * sizeof(float) <= kSkScratchBytes) partial = sk_scratch(s, kSkScratchBytes);
    if (!partial) splitK = 1;
  }
  const int ktPer = (ktAll + splitK - 1) / splitK;
  splitK = (ktAll + ktPer - 1) / ktPer;
  prof_begin(s, 2.0 * o.M * (double)o.N * o.K, PROF_GEMM_BF16);
  hipLaunchKernelGGL((gemm128_bf16_kernel<AKC, BKC>), dim3((unsigned)tilesN, (unsigned)tilesM, (unsigned)splitK), dim3(256), kBfLdsBytes, s,
                     a, b, o, ktPer, splitK > 1 ? partial : nullptr);
  if (splitK > 1) {
    const size_t MN = (size_t)o.M * o.N;
    const unsigned grid = (unsigned)((MN + 255) / 256 > 2048 ? 2048 : (MN + 255) / 256);
    hipLaunchKernelGGL(gemm_bf16_sum_k, dim3(grid), dim3(256), 0, s, partial, splitK, MN, o.N, o.C, o.ldc);
  }
  prof_end(s);
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}

}  // namespace w2l
