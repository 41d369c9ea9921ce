// gemm_glds.hpp -- persistent fp32 MFMA GEMM with direct-to-LDS operand staging (gfx950).
//
// The second-generation main loop of the fl::Linear GEMMs (see gemm.hip for the reference call sites).
// Same 128x128x32 block tile / 4 waves x (2x2) v_mfma_f32_32x32x2_f32 as gemm128_kernel, but
//   * operand tiles go global -> LDS by `global_load_lds_dwordx4` (LDS-DMA): no staging VGPRs, no
//     ds_write pass, no transposing scalar stores between the MFMAs.  The LDS image is lane-linear
//     per wave-instruction (1 KiB pieces), so the bank-conflict-free layout is produced by permuting
//     the per-lane SOURCE address (XOR swizzle), not by padding;
//   * k-contiguous operands keep k contiguous in LDS ([i][32 k], 128-B rows) and a lane fetches FOUR
//     consecutive k of its row with one ds_read_b128.  The MFMA k-slot assignment inside a group of
//     8 k is  k = 8g + 4*(lane>>5) + q  (q = 0..3 = the four MFMA k-steps of the group) for BOTH
//     operands, which only permutes the order of the k-sum;
//   * the kernel is persistent: 2 workgroups per CU walk a static list of (tile, k-range) segments
//     (whole tiles first, then one stream-K range) and the first K tile of the NEXT segment is
//     already in flight while the current segment's last K tile is multiplied and its epilogue
//     runs -- the per-tile prologue latency (5.6 K-iterations per tile in the first-generation kernel,
//     measured on MI355X) is off the critical path;
//   * tiles are rasterised in groups of 8 tile-columns, N-fastest inside a group, and the worker id
//     is XCD-major, so the 64 workgroups of one XCD work on an 8x8 block of tiles that shares
//     8 + 8 operand panels in that XCD's L2.
// Requirements (checked by the host; otherwise gemm128_kernel runs): K % 32 == 0, 16-byte aligned
// operand pointers and leading dimensions, extents % 4 == 0.
#pragma once
#include <cstdlib>

#include "gemm.hpp"

namespace w2l {

struct GOp {
  const float* p;
  int ld;
  int extent;  // number of valid i (rows of A / columns of B)
  unsigned bytes = 0;  // size of the operand's address range (buffer-addressed variant; 0 = too large)
};

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int kGStageFloats = 2 * 128 * 32;  // A tile + B tile of one K step

// XCD-major worker id: hardware workgroup b runs on XCD b % 8; give XCD x the 64 consecutive
// logical workers [64x, 64x + 64) of a 512-wide round (bijective for any worker count)
__device__ __forceinline__ int xcd_major(int b, int workers) {
  const int q = workers / 8, r = workers % 8, x = b % 8;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + b / 8;
}

template <bool KC>
__device__ __forceinline__ void g_init_ptrs(const float* (&q)[4], const GOp& op, int i0, int wave, int lane) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (KC) {
      const int r = (wave * 4 + j) * 8 + (lane >> 3);             // tile row 0..127
      const int c = (lane & 7) ^ ((r >> 1) & 7);                  // source chunk of this LDS slot
      int gi = i0 + r;
      if (gi > op.extent - 1) gi = op.extent - 1;
      q[j] = op.p + (size_t)gi * op.ld + 4 * c;
    } else {
      const int kr = (wave * 4 + j) * 2 + (lane >> 5);            // tile k-row 0..31
      int gi = i0 + 4 * (lane & 31);
      if (gi >= op.extent) gi = (op.extent - 1) & ~3;  // a chunk that straddles the row end stays in place (its tail columns are never stored)
      q[j] = op.p + (size_t)kr * op.ld + gi;
    }
  }
}

__device__ __forceinline__ void g_issue(const float* const (&q)[4], size_t off, float* ldsTile, int wave) {
#pragma unroll
  for (int j = 0; j < 4; ++j)
    __builtin_amdgcn_global_load_lds((gptr_t)(q[j] + off), (lptr_t)(ldsTile + (wave * 4 + j) * 256), 16, 0, 0);
}

__device__ __forceinline__ void g_issue1(const float* q, size_t off, float* ldsTile, int wave, int j) {
  __builtin_amdgcn_global_load_lds((gptr_t)(q + off), (lptr_t)(ldsTile + (wave * 4 + j) * 256), 16, 0, 0);
}

// ---- buffer-addressed variant (BUF): the operand is described by a 128-bit buffer resource in SGPRs, the
// lane supplies ONE 32-bit byte offset (half the address VGPR traffic of a 64-bit global pointer) and the
// K advance rides in the scalar offset of the instruction -- no per-piece 64-bit VALU add.
template <bool KC>
__device__ __forceinline__ void g_init_offs(uint32_t (&vo)[4], const GOp& op, int i0, int wave, int lane) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (KC) {
      const int r = (wave * 4 + j) * 8 + (lane >> 3);
      const int c = (lane & 7) ^ ((r >> 1) & 7);
      int gi = i0 + r;
      if (gi > op.extent - 1) gi = op.extent - 1;
      vo[j] = ((uint32_t)gi * (uint32_t)op.ld + 4u * c) * 4u;
    } else {
      const int kr = (wave * 4 + j) * 2 + (lane >> 5);
      int gi = i0 + 4 * (lane & 31);
      if (gi >= op.extent) gi = (op.extent - 1) & ~3;  // a chunk that straddles the row end stays in place (its tail columns are never stored)
      vo[j] = ((uint32_t)kr * (uint32_t)op.ld + (uint32_t)gi) * 4u;
    }
  }
}
__device__ __forceinline__ void g_issue1_buf(__amdgpu_buffer_rsrc_t rsrc, uint32_t voff, uint32_t soff, float* ldsTile, int wave, int j) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)(ldsTile + (wave * 4 + j) * 256), 16, (int)voff, (int)soff, 0, 0);
}

// fragments of one 8-k group for the two 32-row MFMA tiles of this wave
template <bool KC>
__device__ __forceinline__ void g_frag(float (&f)[2][4], const float* tile, int w0, int g, int li, int lh) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    if (KC) {
      const int r = w0 + 32 * i + li;
      const int c = (2 * g) ^ lh ^ ((li >> 1) & 7);
      const f32x4 v = *(const f32x4*)(tile + r * 32 + 4 * c);
      f[i][0] = v[0]; f[i][1] = v[1]; f[i][2] = v[2]; f[i][3] = v[3];
    } else {
      const float* src = tile + (8 * g + 4 * lh) * 128 + w0 + 32 * i + li;
#pragma unroll
      for (int q = 0; q < 4; ++q) f[i][q] = src[q * 128];
    }
  }
}


// Epilogue with 16-byte stores.  The MFMA C layout gives a lane 16 rows of ONE column, so the direct
// epilogue (gemm128_epilogue) issues 64 global_store_dword per lane and per tile -- store-issue bound: with
// the K loop ablated the TDS fc shapes ran at 117 TF/s against 145 TF/s at 4096^3 (MI355X).  Here each
// wave turns its 64x64 sub-tile through its own 8 KiB slice of the LDS stage that the K loop has just
// released (two 32-row halves): ds_write_b32 in C layout, ds_read_b128 as 4 rows x 16 float4 per
// instruction, then 16 global_store_dwordx4 per lane.  Bias / ReLU / mask / accumulate are applied on
// the float4.  Requires a 16-byte aligned C and ldc % 4 == 0 (host-checked; `wide` false otherwise).
// quad: which 64 x 64 quadrant of the 128 x 128 tile this wave holds (-1: threadIdx.x >> 6); slice: which 8 KiB slice of
// `scratch` it turns its rows through (-1: the same) -- the 256 x 256 kernel runs two 4-wave groups side by side
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 g_bf16x2_t __attribute__((ext_vector_type(2)));
typedef float g_f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t g_pack2(float a, float b) {   // two bf16, round to nearest even (v_cvt_pk_bf16_f32): convert.hip's cv_pack2
  const g_f32x2_t v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, g_bf16x2_t));
}
// IMG (the bf16-operand kernels): out.imgRows / out.imgTrans / out.maskH are honoured (see GemmOut).  The row image leaves with the
// float4 as four bf16 (128-byte row segments per 16 lanes); for the transposed image the finished float4 goes back into its LDS slot
// and, after the half's eight passes, lane c reads column c of the 32 x 64 slice and stores 32 consecutive rows = 64 bytes of
// image row n0 + c (the two halves of a wave complete the 128-byte line in the L2).
template <bool IMG = false>
__device__ __forceinline__ void gemm128g_epilogue_wide(const GemmOut& out, int m0, int n0, const f32x16 (&acc)[2][2],
                                                       float* scratch, const float (&bv)[4], int quad = -1, int slice = -1) {
  const int EPI = out.epi;
  const int lane = threadIdx.x & 63, wave = quad >= 0 ? quad : (int)(threadIdx.x >> 6);
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  const int li = lane & 31, lh = lane >> 5;
  float* sc = scratch + (slice >= 0 ? slice : wave) * 2048;       // [32 rows][64 cols] of this wave
  const int c4 = 4 * (lane & 15), rq = lane >> 4;
  const int n = n0 + wn + c4;
  const bool fullVec = n + 3 < out.N;
  // (Register budget: 187 - 195 VGPRs with the eight prefetched operand vectors of a half below -- two workgroups leave ~120
  // registers per SIMD lane free, room for a communication kernel's waves to co-reside during the overlapped gradient all-reduce.
  // Round 1 had tried the prefetch for all 16 rows of a wave at once and seen no gain; per half, with clamped instead of predicated
  // addresses, it is worth 10 - 15 us per product that has a mask or an addend: profiles/r06_run26 ... r06_run28_*.)
  const float* accSrc = out.addend ? out.addend : out.C;
  // the mask (or, without one, the addend / accumulate operand) of a half's eight row passes is fetched BEFORE the half's LDS turn,
  // addresses clamped instead of predicated (see t160_epilogue: in the pass loop every pass paid its own memory round trip)
  const bool preMask = (EPI & EPI_MASK) != 0, preAdd = !preMask && (EPI & EPI_ACCUM);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    f32x4 pre[8];
    if (IMG && preMask && out.maskH) {
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        int m;
        const bool ok = gemm_out_row(out, m0 + wm + 32 * i + 4 * p + rq, m) && fullVec;
        const u32x2 u = __builtin_nontemporal_load((const u32x2*)(out.maskH + (ok ? (size_t)m * out.ldMaskH + n : (size_t)0)));
        pre[p][0] = __uint_as_float(u[0] << 16); pre[p][1] = __uint_as_float(u[0] & 0xffff0000u);
        pre[p][2] = __uint_as_float(u[1] << 16); pre[p][3] = __uint_as_float(u[1] & 0xffff0000u);
      }
    } else if (preMask || preAdd) {
      const float* src = preMask ? out.mask : accSrc;
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        int m;
        const bool ok = gemm_out_row(out, m0 + wm + 32 * i + 4 * p + rq, m) && fullVec;
        pre[p] = __builtin_nontemporal_load((const f32x4*)(src + (ok ? (size_t)m * out.ldc + n : (size_t)0)));   // read once: not worth a cache line next to the operand panels
      }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) sc[((r & 3) + 8 * (r >> 2) + 4 * lh) * 64 + j * 32 + li] = acc[i][j][r];
    // same wave wrote and reads: LDS operations of one wave complete in order, no barrier needed
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const int row = 4 * p + rq;
      const f32x4 v4 = *(const f32x4*)(sc + row * 64 + c4);
      int m;
      if (!gemm_out_row(out, m0 + wm + 32 * i + row, m) || n >= out.N) continue;
      float v[4] = {v4[0] + bv[0], v4[1] + bv[1], v4[2] + bv[2], v4[3] + bv[3]};
      float* dst = out.C + (size_t)m * out.ldc + n;
      if (EPI & EPI_RELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
      }
      if (EPI & EPI_DROPOUT) {
        const uint64_t idx = (uint64_t)m * out.ldc + n;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = keep_elem(idx + e, out.dropSeed, out.dropStream, out.dropThr) ? v[e] * out.dropScale : 0.f;
      }
      if (fullVec) {
        if (EPI & EPI_MASK) {
          const f32x4 mk = pre[p];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = mk[e] > 0.f ? v[e] * out.maskScale : 0.f;
        }
        if (EPI & EPI_ACCUM) {
          const f32x4 o = preAdd ? pre[p] : *(const f32x4*)(accSrc + (size_t)m * out.ldc + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += o[e];
        }
        f32x4 w4;
        w4[0] = v[0]; w4[1] = v[1]; w4[2] = v[2]; w4[3] = v[3];
        if (!IMG || out.C) { if (out.ntStore) __builtin_nontemporal_store(w4, (f32x4*)dst); else *(f32x4*)dst = w4; }
        if (IMG && out.imgRows) {
          u32x2 h;
          h[0] = g_pack2(v[0], v[1]); h[1] = g_pack2(v[2], v[3]);
          *(u32x2*)(out.imgRows + (size_t)m * out.ldImgRows + n) = h;
        }
        if (IMG && out.imgTrans) *(f32x4*)(sc + row * 64 + c4) = w4;   // back into its slot, for the column pass below
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (n + e >= out.N) continue;
          float t = v[e];
          if (EPI & EPI_MASK) t = out.mask[(size_t)m * out.ldc + n + e] > 0.f ? t * out.maskScale : 0.f;
          if (EPI & EPI_ACCUM) t += accSrc[(size_t)m * out.ldc + n + e];
          dst[e] = t;
        }
      }
    }
    if (IMG && out.imgTrans) {   // (host: N % 4 == 0, no row remap -- every stored float4 above was a full one)
      const int nn = n0 + wn + lane, mb = m0 + wm + 32 * i;
      int rv = out.M - mb;
      rv = rv > 32 ? 32 : rv;
      if (nn < out.N && rv > 0) {
        uint32_t pk[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) pk[q] = g_pack2(sc[(2 * q) * 64 + lane], sc[(2 * q + 1) * 64 + lane]);
        uint16_t* d = out.imgTrans + (size_t)nn * out.ldImgTrans + mb;
        if (rv == 32) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            u32x4 t;
            t[0] = pk[4 * q]; t[1] = pk[4 * q + 1]; t[2] = pk[4 * q + 2]; t[3] = pk[4 * q + 3];
            *(u32x4*)(d + 8 * q) = t;
          }
        } else {
          for (int rr = 0; rr < rv; ++rr) d[rr] = (uint16_t)(pk[rr >> 1] >> (16 * (rr & 1)));
        }
      }
    }
    // the second half overwrites the slice only after this wave's reads have returned (in-order LDS queue)
  }
}

struct GSeg {
  int tile, kb, ke, slab;  // slab: index of the partial slab (stream-K ranges), -1 = whole tile
  bool valid;
};

// segment `ord` of logical worker w: whole tiles w, w + workers, ... then the stream-K range w
__device__ __forceinline__ GSeg g_segment(const SkPlan& p, int w, int workers, int ord) {
  GSeg s;
  s.valid = false;
  s.tile = 0; s.kb = 0; s.ke = 0; s.slab = -1;
  if (p.ksplit) {   // aligned K split: unit u = (chunk, tile), chunk-major
    const int u = w + ord * workers;
    if (u >= p.ksplit * p.skTiles) return s;
    const int x = u / p.skTiles;
    s.tile = u - x * p.skTiles;
    s.kb = x * p.kChunk;
    s.ke = s.kb + p.kChunk < p.kTiles ? s.kb + p.kChunk : p.kTiles;
    s.slab = u;
    s.valid = true;
    return s;
  }
  const int nDp = w < p.dpTiles ? (p.dpTiles - w + workers - 1) / workers : 0;
  if (ord < nDp) {
    s.tile = w + ord * workers; s.kb = 0; s.ke = p.kTiles; s.valid = true;
    return s;
  }
  if (w >= p.skBlocks) return s;
  const int o = ord - nDp;
  if (o > 1) return s;
  long long it = sk_begin(p, w);
  const long long itEnd = sk_begin(p, w + 1);
  if (it >= itEnd) return s;
  long long firstEnd = (it / p.kTiles + 1) * (long long)p.kTiles;
  if (firstEnd > itEnd) firstEnd = itEnd;
  if (o == 1) {
    it = firstEnd;
    if (it >= itEnd) return s;
    firstEnd = itEnd;
  }
  s.tile = p.dpTiles + (int)(it / p.kTiles);
  s.kb = (int)(it % p.kTiles);
  s.ke = s.kb + (int)(firstEnd - it);
  s.slab = (s.kb == 0 && s.ke == p.kTiles) ? -1 : w * 2 + o;
  s.valid = true;
  return s;
}

// segment fields as SGPRs: left in VGPRs (the schedule's 64-bit divisions), the K-tile index makes hipcc wrap every
// buffer-addressed LDS-DMA issue in a readfirstlane "waterfall" loop for its scalar offset (8 loops per K tile in the ISA)
__device__ __forceinline__ GSeg g_pin(GSeg s) {
  s.tile = __builtin_amdgcn_readfirstlane(s.tile);
  s.kb = __builtin_amdgcn_readfirstlane(s.kb);
  s.ke = __builtin_amdgcn_readfirstlane(s.ke);
  s.slab = __builtin_amdgcn_readfirstlane(s.slab);
  s.valid = __builtin_amdgcn_readfirstlane((int)s.valid) != 0;
  return s;
}

constexpr int kGemmPrioMode = 0;   // the product's value (W2L_GEMM_PRIO overrides it in the probe library)
__device__ __forceinline__ void g_dbg_record(long long* dbg, long long t0) {
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  long long* d = dbg + 4 * (size_t)blockIdx.x;
  d[0] = t0; d[1] = wall_clock64(); d[2] = hw; d[3] = xcc;
}
// Fair shares for the two workgroups of a CU.  Left alone, one of the pair runs ~8 % faster than the other for the whole launch (the
// arbiter's tie-break is not fair), finishes 80 - 140 us early and leaves its partner alone on the CU for 10 - 19 % of the launch
// (profiles/r06_run35_gemm_workgroup_end_times.log) -- and one workgroup alone does not keep the matrix pipe busy through its
// barriers and epilogues.  mode 1: the pair swaps priority every segment (who is high first follows the LDS allocation base), so
// that over two segments both get the same share.  mode 2 (probe): the second workgroup stays low (sensitivity check).
__device__ __forceinline__ int g_lds_second() {
  unsigned la;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_LDS_ALLOC)" : "=s"(la));
  return (la & 0xfffu) != 0;
}
__device__ __forceinline__ void g_tile_prio(int mode, int second, int ord) {
  if (mode == 1) {
    if ((ord + second) & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
  } else if (mode == 2) {
    if (second) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(1);
  }
}
inline int gemm_prio_mode() {
  static const int v = [] { const char* e = tune_env("W2L_GEMM_PRIO"); return e ? atoi(e) : kGemmPrioMode; }();
  return v;
}
inline long long* gemm_dbg_ptr() {
  const char* e = tune_env("W2L_GEMM_DBG");
  return e ? (long long*)strtoull(e, nullptr, 10) : nullptr;
}

// ABL: timing-only ablations (results are garbage) selected by W2L_GEMM_ABL for the probe tool:
//   1 = no LDS-DMA, 2 = no per-K-tile barrier, 4 = no fragment reads in the loop, 8 = no epilogue,
//   16 = LDS-DMA always re-reads K tile 0 (cache-resident source), 32 = LDS-DMA of the A operand only,
//   64 = every tile is tile 0 (operands and output stay cache-resident: isolates new-panel memory effects)
template <bool AKC, bool BKC, int ABL = 0, bool BUF = false>
__global__ __launch_bounds__(256, 2) void gemm128g_kernel(GOp aop, GOp bop, GemmOut out, SkPlan plan, int workers, int wide) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  // wave id as a SCALAR: LDS-DMA destinations (M0) and piece indices stay on the SALU, no v_readfirstlane per piece
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  const int li = lane & 31, lh = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(xcd_major(blockIdx.x, workers));
  const size_t aStep = AKC ? 32 : (size_t)32 * aop.ld;
  const size_t bStep = BKC ? 32 : (size_t)32 * bop.ld;

  GSeg seg = g_pin(g_segment(plan, w, workers, 0));
  if (!seg.valid) return;
  const long long dbgT0 = plan.dbg ? wall_clock64() : 0;
  const int second = plan.prio ? g_lds_second() : 0;
  const float* qa[4];
  const float* qb[4];
  uint32_t va[4], vb[4];
  __amdgpu_buffer_rsrc_t ra, rb;
  if (BUF) {
    ra = __builtin_amdgcn_make_buffer_rsrc((void*)aop.p, 0, (int)aop.bytes, 0x00020000);
    rb = __builtin_amdgcn_make_buffer_rsrc((void*)bop.p, 0, (int)bop.bytes, 0x00020000);
  }
  int bx, by;
  sk_tile_xy(plan, (ABL & 64) ? 0 : seg.tile, bx, by);
  if (BUF) {
    g_init_offs<AKC>(va, aop, bx * 128, wave, lane);
    g_init_offs<BKC>(vb, bop, by * 128, wave, lane);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      g_issue1_buf(ra, va[j], (uint32_t)(aStep * seg.kb * 4), smem, wave, j);
      g_issue1_buf(rb, vb[j], (uint32_t)(bStep * seg.kb * 4), smem + 4096, wave, j);
    }
  } else {
    g_init_ptrs<AKC>(qa, aop, bx * 128, wave, lane);
    g_init_ptrs<BKC>(qb, bop, by * 128, wave, lane);
    if (!(ABL & 1)) {
      g_issue(qa, aStep * seg.kb, smem, wave);
      g_issue(qb, bStep * seg.kb, smem + 4096, wave);
    }
  }
  int stage = 0;
  __syncthreads();  // (drains the LDS-DMA: vmcnt(0) precedes the barrier)

  for (int ord = 0;; ++ord) {
    if (plan.prio) g_tile_prio(plan.prio, second, ord);
    const GSeg nxt = g_pin(g_segment(plan, w, workers, ord + 1));
    // bias of this lane's four output columns, fetched at the START of the tile (its latency hides under the K loop)
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (out.epi & EPI_BIAS) {
      const int nb = by * 128 + wn + 4 * (lane & 15);
#pragma unroll
      for (int e = 0; e < 4; ++e) bv[e] = nb + e < out.N ? out.bias[nb + e] : 0.f;
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    for (int kt = seg.kb; kt < seg.ke; ++kt) {
      const float* As = smem + stage * kGStageFloats;
      const float* Bs = As + 4096;
      float* An = smem + (stage ^ 1) * kGStageFloats;
      float fa[2][2][4], fb[2][2][4];
      g_frag<AKC>(fa[0], As, wm, 0, li, lh);
      g_frag<BKC>(fb[0], Bs, wn, 0, li, lh);
      if (ABL & 4) {
        g_frag<AKC>(fa[1], As, wm, 1, li, lh);
        g_frag<BKC>(fb[1], Bs, wn, 1, li, lh);
      }
      // What goes to the other stage during this iteration: the next K tile, or the first K tile of
      // the next segment, or (very last iteration of this worker) a harmless re-load of this tile.
      // Every wave has passed the barrier that ended the previous iteration, so nobody reads that stage.
      size_t offA = aStep * kt, offB = bStep * kt;
      if (kt + 1 < seg.ke) {
        offA += aStep; offB += bStep;
      } else if (nxt.valid) {
        int nbx, nby;
        sk_tile_xy(plan, (ABL & 64) ? 0 : nxt.tile, nbx, nby);
        if (BUF) {
          g_init_offs<AKC>(va, aop, nbx * 128, wave, lane);
          g_init_offs<BKC>(vb, bop, nby * 128, wave, lane);
        } else {
          g_init_ptrs<AKC>(qa, aop, nbx * 128, wave, lane);
          g_init_ptrs<BKC>(qb, bop, nby * 128, wave, lane);
        }
        offA = aStep * nxt.kb; offB = bStep * nxt.kb;
      }
      const uint32_t sOffA = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(offA * 4));
      const uint32_t sOffB = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(offB * 4));
      // 16 k-steps of 4 MFMAs; the 8 LDS-DMA pieces and the fragment reads of the next 8-k group are
      // slotted BETWEEN k-steps (one filler per gap, order pinned) so that their issue cost and latency
      // sit behind MFMA execution instead of in front of it.
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int cur = g & 1;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][0][q], fb[cur][0][q], acc[0][0], 0, 0, 0);
          acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][0][q], fb[cur][1][q], acc[0][1], 0, 0, 0);
          acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][1][q], fb[cur][0][q], acc[1][0], 0, 0, 0);
          acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][1][q], fb[cur][1][q], acc[1][1], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          const int step = 4 * g + q;
          if (q == 0) {
            if (g < 3 && !(ABL & 4)) {
              g_frag<AKC>(fa[cur ^ 1], As, wm, g + 1, li, lh);
              g_frag<BKC>(fb[cur ^ 1], Bs, wn, g + 1, li, lh);
            }
          } else {
            const int piece = step - 1 - g;  // steps 1,2,3,5,6,7,9,10 -> pieces 0..7
            if (BUF) {
              if (!(ABL & 1)) {
                if (piece < 4) g_issue1_buf(ra, va[piece], sOffA, An, wave, piece);
                else if (piece < 8) g_issue1_buf(rb, vb[piece - 4], sOffB, An + 4096, wave, piece - 4);
              }
            } else if (!(ABL & 1)) {
              if (piece < 4) g_issue1(qa[piece], (ABL & 16) ? 0 : offA, An, wave, piece);
              else if (piece < 8 && !(ABL & 32)) g_issue1(qb[piece - 4], (ABL & 16) ? 0 : offB, An + 4096, wave, piece - 4);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      stage ^= 1;
      if (!(ABL & 2)) __syncthreads();  // the stage just filled has landed (vmcnt(0)) and is visible to all waves
    }

    bool doEpi = seg.slab < 0;  // whole tile: epilogue straight from the accumulators
    int resetTicket = -1;
    if (ABL & 8) {
      float t = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) t += acc[0][0][r] + acc[0][1][r] + acc[1][0][r] + acc[1][1][r];
      if (t == 123.456f) out.C[0] = t;  // keeps every accumulator live
      doEpi = false;
    } else if (!doEpi) {
      gemm128_store_partial(plan.slabs + (size_t)seg.slab * kSlabFloats, acc);
      if (plan.counters) {
        // In-kernel slab reduction (replaces the fix-up launch).  Publish: every wave drains its slab stores, one lane
        // releases at agent scope and draws the tile's arrival ticket; the workgroup that draws the LAST ticket acquires
        // and adds ALL slabs of the tile from memory in range order (its own included: deterministic), then runs the
        // ordinary epilogue below and re-zeroes the ticket (cdna_hip_programming.md, in-launch split-K reduction recipe).
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int* flag = (int*)(smem + (stage ^ 1) * kGStageFloats);  // the stage the K loop has just released
        const int t = seg.tile - plan.dpTiles;
        if (tid == 0) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          *flag = (int)__hip_atomic_fetch_add(plan.counters + t, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        const int ticket = *flag;
        int sF, sL;
        sk_tile_ranges(plan, t, sF, sL);
        if (ticket == sL - sF) {  // uniform: last arriver
          if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          __syncthreads();  // (also: every wave has read the ticket before the stage becomes epilogue scratch)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
              for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
          for (int sr = sF; sr <= sL; ++sr) {
            const int segIdx = t - (int)(sk_begin(plan, sr) / plan.kTiles);  // ranges span <= 2 tiles: 0 or 1
            const f32x4* s4 = (const f32x4*)(plan.slabs + ((size_t)sr * 2 + segIdx) * kSlabFloats);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                  const f32x4 v = s4[((wave * 4 + i * 2 + j) * 4 + q) * 64 + lane];
                  acc[i][j][4 * q] += v[0]; acc[i][j][4 * q + 1] += v[1]; acc[i][j][4 * q + 2] += v[2]; acc[i][j][4 * q + 3] += v[3];
                }
          }
          doEpi = true;
          resetTicket = t;
        }
      }
    }
    if (doEpi) {
      // `stage` now names the buffer holding the prefetched next K tile; the other one is free
      if (wide) gemm128g_epilogue_wide(out, bx * 128, by * 128, acc, smem + (stage ^ 1) * kGStageFloats, bv);
      else gemm128_epilogue(out, bx * 128, by * 128, acc);
      if (resetTicket >= 0 && tid == 0) __hip_atomic_store(plan.counters + resetTicket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!nxt.valid) break;
    if (wide || plan.counters) __syncthreads();  // the next iteration's LDS-DMA lands in the slices the epilogue / ticket used

    seg = nxt;
    sk_tile_xy(plan, (ABL & 64) ? 0 : seg.tile, bx, by);
  }
  if (plan.dbg && tid == 0) g_dbg_record(plan.dbg, dbgT0);
}

inline bool launch128p(const GOp& a, bool akc, const GOp& b, bool bkc, GemmOut o, int epi, const SkPlan& plan0, int wide, hipStream_t s);   // gemm_p5.hpp

inline int launch128g(const GOp& a, bool akc, const GOp& b, bool bkc, GemmOut o, int epi, hipStream_t s) {
  epi &= ~EPI_ATOMIC;
  SkPlan plan = make_sk_plan(o.M, o.N, o.K, sk_enabled());
  plan.grouped = 1;
  if (plan.skBlocks > 0) {
    plan.slabs = sk_scratch(s, kSkScratchBytes);
    if (!plan.slabs) { plan = make_sk_plan(o.M, o.N, o.K, false); plan.grouped = 1; }
    const char* eFix = tune_env("W2L_GEMM_INFIX");  // read per call (tests flip it): 0 = separate fix-up launch
    const int inFix = eFix ? atoi(eFix) : 1;
    if (plan.skBlocks > 0 && inFix && plan.skTiles <= 1024) plan.counters = sk_counters(s);
  }
  int workers = plan.dpTiles < kSkSlots ? plan.dpTiles : kSkSlots;
  if (workers < plan.skBlocks) workers = plan.skBlocks;
  plan.dbg = gemm_dbg_ptr();
  plan.prio = gemm_prio_mode();
  { static const int nt = [] { const char* e = tune_env("W2L_GEMM_NTSTORE"); return e ? atoi(e) : 0; }(); o.ntStore = nt; }
  const size_t shmem = 2 * (size_t)kGStageFloats * sizeof(float);
  dim3 grid((unsigned)workers), block(256);
  o.epi = epi;
  prof_begin(s, 2.0 * o.M * (double)o.N * o.K, PROF_GEMM128, o.M, o.N, o.K, 2);
  static const int wideOn = [] { const char* e = tune_env("W2L_GEMM_WIDE"); return e ? atoi(e) : 1; }();
  const int wide = wideOn && (((uintptr_t)o.C) & 15) == 0 && o.ldc % 4 == 0 &&
                   (!o.mask || (((uintptr_t)o.mask) & 15) == 0);
  // buffer-addressed LDS-DMA is the default (+8 % at 4096^3, +5-7 % on the TDS fc shapes over 64-bit global
  // addresses, MI355X); W2L_GEMM_BUF=0 selects the global_load_lds variant for A/B runs
  static const int bufOn = [] { const char* e = tune_env("W2L_GEMM_BUF"); return e ? atoi(e) : 1; }();
#ifdef W2L_PROBE  // timing-only ablations (results are garbage): compiled into the probe library only
  if (launch128p(a, akc, b, bkc, o, epi, plan, wide, s)) { prof_end(s); W2L_LAUNCH_CHECK(); return W2L_OK; }
  static const int ablBuf = [] { const char* e = tune_env("W2L_GEMM_ABLBUF"); return e ? atoi(e) : 0; }();
  static const int abl = [] { const char* e = tune_env("W2L_GEMM_ABL"); return e ? atoi(e) : 0; }();
  if (abl && akc && !bkc) {
    switch (abl) {
      case 1: hipLaunchKernelGGL((gemm128g_kernel<true, false, 1>), grid, block, shmem, s, a, b, o, plan, workers, wide); break;
      case 3: hipLaunchKernelGGL((gemm128g_kernel<true, false, 3>), grid, block, shmem, s, a, b, o, plan, workers, wide); break;
      case 7: hipLaunchKernelGGL((gemm128g_kernel<true, false, 7>), grid, block, shmem, s, a, b, o, plan, workers, wide); break;
      case 15: hipLaunchKernelGGL((gemm128g_kernel<true, false, 15>), grid, block, shmem, s, a, b, o, plan, workers, wide); break;
      case 8: hipLaunchKernelGGL((gemm128g_kernel<true, false, 8>), grid, block, shmem, s, a, b, o, plan, workers, wide); break;
      case 16: hipLaunchKernelGGL((gemm128g_kernel<true, false, 16>), grid, block, shmem, s, a, b, o, plan, workers, wide); break;
      case 32: hipLaunchKernelGGL((gemm128g_kernel<true, false, 32>), grid, block, shmem, s, a, b, o, plan, workers, wide); break;
      case 9: hipLaunchKernelGGL((gemm128g_kernel<true, false, 9>), grid, block, shmem, s, a, b, o, plan, workers, wide); break;
      case 2: hipLaunchKernelGGL((gemm128g_kernel<true, false, 2>), grid, block, shmem, s, a, b, o, plan, workers, wide); break;
      default: hipLaunchKernelGGL((gemm128g_kernel<true, false, 4>), grid, block, shmem, s, a, b, o, plan, workers, wide); break;
    }
  } else if (ablBuf && akc && !bkc && a.bytes && b.bytes) {
    switch (ablBuf) {
      case 1: hipLaunchKernelGGL((gemm128g_kernel<true, false, 1, true>), grid, block, shmem, s, a, b, o, plan, workers, wide); break;
      case 8: hipLaunchKernelGGL((gemm128g_kernel<true, false, 8, true>), grid, block, shmem, s, a, b, o, plan, workers, wide); break;
      case 9: hipLaunchKernelGGL((gemm128g_kernel<true, false, 9, true>), grid, block, shmem, s, a, b, o, plan, workers, wide); break;
      case 64: hipLaunchKernelGGL((gemm128g_kernel<true, false, 64, true>), grid, block, shmem, s, a, b, o, plan, workers, wide); break;
      case 72: hipLaunchKernelGGL((gemm128g_kernel<true, false, 72, true>), grid, block, shmem, s, a, b, o, plan, workers, wide); break;
      default: hipLaunchKernelGGL((gemm128g_kernel<true, false, 0, true>), grid, block, shmem, s, a, b, o, plan, workers, wide); break;
    }
  } else
#endif
  if (bufOn && a.bytes && b.bytes) {
    if (akc && bkc) hipLaunchKernelGGL((gemm128g_kernel<true, true, 0, true>), grid, block, shmem, s, a, b, o, plan, workers, wide);
    else if (akc) hipLaunchKernelGGL((gemm128g_kernel<true, false, 0, true>), grid, block, shmem, s, a, b, o, plan, workers, wide);
    else if (bkc) hipLaunchKernelGGL((gemm128g_kernel<false, true, 0, true>), grid, block, shmem, s, a, b, o, plan, workers, wide);
    else hipLaunchKernelGGL((gemm128g_kernel<false, false, 0, true>), grid, block, shmem, s, a, b, o, plan, workers, wide);
  } else if (akc && bkc) hipLaunchKernelGGL((gemm128g_kernel<true, true>), grid, block, shmem, s, a, b, o, plan, workers, wide);
  else if (akc) hipLaunchKernelGGL((gemm128g_kernel<true, false>), grid, block, shmem, s, a, b, o, plan, workers, wide);
  else if (bkc) hipLaunchKernelGGL((gemm128g_kernel<false, true>), grid, block, shmem, s, a, b, o, plan, workers, wide);
  else hipLaunchKernelGGL((gemm128g_kernel<false, false>), grid, block, shmem, s, a, b, o, plan, workers, wide);
  if (plan.skBlocks > 0 && !plan.counters)
    hipLaunchKernelGGL(gemm128_fixup<0>, dim3((unsigned)plan.skTiles * 4), dim3(64), 0, s, o, plan);
  prof_end(s);
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}

}  // namespace w2l
