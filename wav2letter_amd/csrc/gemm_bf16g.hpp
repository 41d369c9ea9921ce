// gemm_bf16g.hpp -- bf16-operand MFMA GEMM for the mixed-precision mode (BASELINE configs 3 and 5, fl's
// --fl_amp_use_mixed_precision restated for bf16: recipes/slimIPL/src/Train.cpp:211, :1681-1760; the criterion input stays
// f32, recipes/joint_training_vox_populi/cpc/Train.cpp:1184).
//
//   C[M][N] (fp32) = A[M][K] . B[N][K]^T,  A and B bf16 in HBM, k contiguous, rows zero-padded to a multiple of 64 k
//
// Round 2's kernel (gemm_bf16.hpp) kept fp32 operands in HBM and converted them on the way into LDS: it was bound by
// staging 4-byte operands (190 TF/s in the config-3 step).  Here the operands ARE bf16 (activations and weight copies are
// converted once per step by convert.hip, the transposed copies the weight-gradient product needs included), so the tile
// traffic halves, the K tile doubles to 64 and the whole staging path is the fp32 engine's LDS-DMA one:
//   * a 128 x 64 bf16 operand tile is byte for byte the 128 x 32 fp32 tile of gemm_glds.hpp (128 rows of 128 bytes), so the
//     buffer-addressed `buffer_load_dwordx4 ... lds` pieces, their XOR-swizzled source chunks and the conflict-free
//     ds_read_b128 fragment reads are reused unchanged: the 16 bytes a lane reads are now 8 consecutive k of its row --
//     exactly the A / B operand of ONE v_mfma_f32_32x32x16_bf16 (lane l: row l & 31, k = 16 s + 8 (l >> 5) .. + 8);
//   * 4 waves x (2 x 2) 32 x 32 fp32 accumulators per 128 x 128 tile: the C layout is dtype-independent on gfx950, so the
//     fp32 engine's epilogues (bias / ReLU / dropout / mask / addend, 16-byte stores through LDS), its persistent
//     XCD-major schedule and its in-kernel stream-K slab reduction are shared as they are;
//   * a K tile is 16 MFMAs of 32 cycles per wave instead of 64 of 64: the next tile's eight LDS-DMA pieces are issued at the
//     TOP of the iteration (they have the whole tile, and the co-resident workgroup's, to land) and the fragment reads of
//     k-step s + 1 sit between the MFMAs of k-step s.
#pragma once
#include "gemm_glds.hpp"

namespace w2l {

typedef __bf16 bf16x8v_t __attribute__((ext_vector_type(8)));

__device__ __forceinline__ bf16x8v_t h_frag(const float* tile, int row, int s, int lh, int li) {
  // chunk (2 s + lh) of the 128-byte row, at the XOR-swizzled slot the DMA source permutation put it in (g_init_offs<true>)
  const int c = (2 * s) ^ lh ^ ((li >> 1) & 7);
  const f32x4 v = *(const f32x4*)(tile + row * 32 + 4 * c);
  return __builtin_bit_cast(bf16x8v_t, v);
}

// ---- k-MAJOR operands (round 4): a bf16 matrix stored [K][extent] (k is the ROW index, the MFMA row / column index i is
// contiguous) -- an activation x [frames][in] as the A operand of a weight gradient x^T dy, a weight w [in][out] as the B operand
// of the forward product x w -- read IN PLACE instead of from a transposed bf16 image (convert.hip wrote one per operand and
// step: half of all conversion traffic of the mixed-precision step).
//   * a K tile is 64 k-rows x 128 columns = 64 rows of 256 bytes; an LDS-DMA piece (1 KiB, one wave instruction) is 4 k-rows;
//     the 16-byte chunk c of k-row k lands at chunk position c ^ (4 (k & 3)) of its row (permutation applied to the per-lane
//     SOURCE address: the DMA destination is lane-linear);
//   * the fragment of one v_mfma_f32_32x32x16_bf16 -- lane l: 8 consecutive k of column l & 31 -- is two ds_read_b64_tr_b16: inside
//     each 16-lane group the lanes' 8-byte chunks form a 4 x 16 matrix (row = chunks of lanes 4 j .. 4 j + 3) and lane t receives
//     column t (tools/micro/tr16_probe.hip, profiles/r03_run13_tr16_probe.log): lane t = 4 j + q supplies the address of
//     [k0 + j][col0 + 4 q .. + 3], lane t receives [k0 .. k0 + 3][col0 + t].  With the permutation above the 32 lanes of a
//     half-wave (two groups: columns col0 .. + 15 and + 16 .. + 31, the same four k-rows) touch 32 distinct 8-byte slots of
//     the 256-byte bank row: conflict-free.
typedef short s16x4v_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8v_t h_frag_t(const float* tile, int col0, int s, int lane) {
  const int G = lane >> 4, t = lane & 15, j = t >> 2, q = t & 3;
  const int k0 = 16 * s + 8 * (G >> 1) + j;                 // k-row of the first read (the second: + 4); k0 & 3 == j
  const int col = col0 + 16 * (G & 1) + 4 * q;
  const int chunk = (col >> 3) ^ (4 * j), half = (col >> 2) & 1;
  const char* a0 = (const char*)tile + k0 * 256 + 16 * chunk + 8 * half;
  const s16x4v_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4v_t*)a0);
  const s16x4v_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4v_t*)(a0 + 4 * 256));
  typedef short s16x8v_t __attribute__((ext_vector_type(8)));
  const s16x8v_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8v_t, v);
}
// per-lane byte offsets of the four LDS-DMA pieces of this wave for a k-major operand: piece wave * 4 + j = k-rows 4 (wave * 4 + j) ..
// + 4 of the K tile, lane L -> k-row + (L >> 4), chunk position L & 15 <- source chunk (L & 15) ^ (4 (L >> 4)).  op.ld in floats
// (= bf16 elements / 2), op.extent = number of columns; a chunk that starts beyond the last column re-reads the last valid one
// (columns past the extent are never stored)
__device__ __forceinline__ void t_init_offs(uint32_t (&vo)[4], const GOp& op, int i0, int wave, int lane) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int kr = (wave * 4 + j) * 4 + (lane >> 4);
    const int c = (lane & 15) ^ (4 * (lane >> 4));
    int col = i0 + 8 * c;
    if (col >= op.extent) col = (op.extent - 1) & ~7;
    vo[j] = ((uint32_t)kr * (uint32_t)op.ld) * 4u + (uint32_t)col * 2u;
  }
}

// aop / bop: bf16 matrices viewed as float matrices of half the width (p, ld in floats = bf16 elements / 2); plan.kTiles
// counts 64-k tiles.  Same launch geometry as gemm128g_kernel.
// GROUPED launches (GRP): up to kHMaxGroups problems of the SAME shape (own A, B, C, bias) share one persistent grid -- tile
// index t = g * grp.tiles + local tile.  The weight gradients of a Transformer block's four C x C projections are 64 tiles
// each at M = 3008 frames: one at a time they occupy a quarter of the chip (38 us each, 164 TFLOP/s), together they fill it.
constexpr int kHMaxGroups = 4;
struct HGroups {
  const float* a[kHMaxGroups];
  const float* b[kHMaxGroups];
  float* c[kHMaxGroups];
  const float* bias[kHMaxGroups];
  int n = 0, tiles = 0;      // problems, tiles per problem
};

// TA / TB: the operand is k-MAJOR ([K][M] / [K][N], see above) instead of k-contiguous
template <bool GRP, bool TA = false, bool TB = false>
__global__ __launch_bounds__(256, 2) void gemm128h_kernel(GOp aop, GOp bop, GemmOut out, SkPlan plan, int workers, int wide, HGroups grp) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  const int li = lane & 31, lh = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(xcd_major(blockIdx.x, workers));
  // one K tile = 64 bf16: 128 bytes along a row (k-contiguous operand) / 64 rows (k-major operand)
  const uint32_t stepA = TA ? 256u * (uint32_t)aop.ld : 128u, stepB = TB ? 256u * (uint32_t)bop.ld : 128u;

  GSeg seg = g_pin(g_segment(plan, w, workers, 0));
  if (!seg.valid) return;
  uint32_t va[4], vb[4];
  int grpIdx = 0, grpNext = 0;
  if (GRP) {
    grpIdx = seg.tile / grp.tiles;
    seg.tile -= grpIdx * grp.tiles;
    aop.p = grp.a[grpIdx]; bop.p = grp.b[grpIdx];
  }
  __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)aop.p, 0, (int)aop.bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)bop.p, 0, (int)bop.bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t raN = ra, rbN = rb;
  int bx, by;
  sk_tile_xy(plan, seg.tile, bx, by);
  if (TA) t_init_offs(va, aop, bx * 128, wave, lane); else g_init_offs<true>(va, aop, bx * 128, wave, lane);
  if (TB) t_init_offs(vb, bop, by * 128, wave, lane); else g_init_offs<true>(vb, bop, by * 128, wave, lane);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    g_issue1_buf(ra, va[j], stepA * (uint32_t)seg.kb, smem, wave, j);
    g_issue1_buf(rb, vb[j], stepB * (uint32_t)seg.kb, smem + 4096, wave, j);
  }
  int stage = 0;
  __syncthreads();  // (drains the LDS-DMA: vmcnt(0) precedes the barrier)

  for (int ord = 0;; ++ord) {
    GSeg nxt = g_pin(g_segment(plan, w, workers, ord + 1));
    if (GRP) {
      if (nxt.valid) {
        grpNext = nxt.tile / grp.tiles;
        nxt.tile -= grpNext * grp.tiles;
        raN = __builtin_amdgcn_make_buffer_rsrc((void*)grp.a[grpNext], 0, (int)aop.bytes, 0x00020000);
        rbN = __builtin_amdgcn_make_buffer_rsrc((void*)grp.b[grpNext], 0, (int)bop.bytes, 0x00020000);
      }
      out.C = grp.c[grpIdx];
      out.bias = grp.bias[grpIdx];
    }
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if ((out.epi & EPI_BIAS) && out.bias) {
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
      // what goes to the other stage during this iteration: the next K tile, or the first K tile of the next segment, or
      // (very last iteration of this worker) a harmless re-load of this tile
      uint32_t offA = stepA * (uint32_t)kt, offB = stepB * (uint32_t)kt;
      bool toNext = false;   // this iteration's pieces belong to the next segment (GRP: possibly another problem's operands)
      if (kt + 1 < seg.ke) {
        offA += stepA; offB += stepB;
      } else if (nxt.valid) {
        toNext = true;
        int nbx, nby;
        sk_tile_xy(plan, nxt.tile, nbx, nby);
        if (TA) t_init_offs(va, aop, nbx * 128, wave, lane); else g_init_offs<true>(va, aop, nbx * 128, wave, lane);
        if (TB) t_init_offs(vb, bop, nby * 128, wave, lane); else g_init_offs<true>(vb, bop, nby * 128, wave, lane);
        offA = stepA * (uint32_t)nxt.kb; offB = stepB * (uint32_t)nxt.kb;
      }
      offA = (uint32_t)__builtin_amdgcn_readfirstlane((int)offA);
      offB = (uint32_t)__builtin_amdgcn_readfirstlane((int)offB);
      const __amdgpu_buffer_rsrc_t rA = (GRP && toNext) ? raN : ra, rB = (GRP && toNext) ? rbN : rb;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        g_issue1_buf(rA, va[j], offA, An, wave, j);
        g_issue1_buf(rB, vb[j], offB, An + 4096, wave, j);
      }
      bf16x8v_t fa[2][2], fb[2][2];
      fa[0][0] = TA ? h_frag_t(As, wm, 0, lane) : h_frag(As, wm + li, 0, lh, li);
      fa[0][1] = TA ? h_frag_t(As, wm + 32, 0, lane) : h_frag(As, wm + 32 + li, 0, lh, li);
      fb[0][0] = TB ? h_frag_t(Bs, wn, 0, lane) : h_frag(Bs, wn + li, 0, lh, li);
      fb[0][1] = TB ? h_frag_t(Bs, wn + 32, 0, lane) : h_frag(Bs, wn + 32 + li, 0, lh, li);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int cur = s & 1;
        if (s < 3) {
          fa[cur ^ 1][0] = TA ? h_frag_t(As, wm, s + 1, lane) : h_frag(As, wm + li, s + 1, lh, li);
          fa[cur ^ 1][1] = TA ? h_frag_t(As, wm + 32, s + 1, lane) : h_frag(As, wm + 32 + li, s + 1, lh, li);
          fb[cur ^ 1][0] = TB ? h_frag_t(Bs, wn, s + 1, lane) : h_frag(Bs, wn + li, s + 1, lh, li);
          fb[cur ^ 1][1] = TB ? h_frag_t(Bs, wn + 32, s + 1, lane) : h_frag(Bs, wn + 32 + li, s + 1, lh, li);
        }
        // keep the order written here: left alone, hipcc funnels every fragment through ONE register quad (read, wait
        // lgkmcnt(0), two MFMAs, read, ...) and exposes an LDS round trip per MFMA pair -- seen in the ISA of both kernels
        __builtin_amdgcn_sched_barrier(0);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][0], fb[cur][0], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][0], fb[cur][1], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][1], fb[cur][0], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][1], fb[cur][1], acc[1][1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      stage ^= 1;
      __syncthreads();  // the stage just filled has landed (vmcnt(0)) and is visible to all waves
    }

    bool doEpi = seg.slab < 0;  // whole tile: epilogue straight from the accumulators
    int resetTicket = -1;
    if (!doEpi) {
      gemm128_store_partial(plan.slabs + (size_t)seg.slab * kSlabFloats, acc);
      if (plan.counters) {
        // in-kernel slab reduction: the same release / ticket / acquire hand-over as gemm128g_kernel (see there)
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
        int sF = 0, sL = plan.ksplit - 1;   // aligned K split: chunk sr of this tile is slab sr * skTiles + t
        if (!plan.ksplit) sk_tile_ranges(plan, t, sF, sL);
        if (ticket == sL - sF) {  // uniform: last arriver
          if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          __syncthreads();
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
              for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
          for (int sr = sF; sr <= sL; ++sr) {
            size_t slabIdx;
            if (plan.ksplit) {
              slabIdx = (size_t)sr * plan.skTiles + t;
            } else {
              const int segIdx = t - (int)(sk_begin(plan, sr) / plan.kTiles);  // ranges span <= 2 tiles: 0 or 1
              slabIdx = (size_t)sr * 2 + segIdx;
            }
            const f32x4* s4 = (const f32x4*)(plan.slabs + slabIdx * kSlabFloats);
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
      if (wide) gemm128g_epilogue_wide<true>(out, bx * 128, by * 128, acc, smem + (stage ^ 1) * kGStageFloats, bv);
      else gemm128_epilogue(out, bx * 128, by * 128, acc);
      if (resetTicket >= 0 && tid == 0) __hip_atomic_store(plan.counters + resetTicket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!nxt.valid) break;
    if (wide || plan.counters) __syncthreads();  // the next iteration's LDS-DMA lands in the slices the epilogue / ticket used

    seg = nxt;
    if (GRP) { grpIdx = grpNext; ra = raN; rb = rbN; }
    sk_tile_xy(plan, seg.tile, bx, by);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// 256 x 256 x 64 tiles, 8 waves, whole tiles only.  Counters of the 128 x 128 kernel on the config-3 fc shape (M = 11968,
// N = K = 2160; profiles/r03_run2_gemm128h_pmc.txt): 33 % MFMA busy, 48 % of the wave cycles parked at the per-K-tile DMA
// drain, 1.95 GB of operand tiles through the L2 for 166 us.  A 256 x 256 tile halves the staged bytes per flop and, with
// one 8-wave workgroup per CU (two waves per SIMD, 128 accumulator registers each), a K tile is 32 MFMAs = 1024 MFMA
// cycles per wave: the eight 1 KiB LDS-DMA pieces a wave issues at the top of the iteration have twice the time to land.
//   * wave w: column half g = w >> 2, quadrant q = w & 3 of BOTH 128-row halves -- rows si 128 + (q >> 1) 64 + {0, 32} + li
//     (si = 0, 1), columns g 128 + (q & 1) 64 + {0, 32} + li: every 128 x 128 sub-tile (si, g) is held by four waves in
//     the fp32 engine's quadrant layout, so its epilogues are reused per sub-tile (two 4-wave groups side by side);
//   * LDS: 2 stages x (A 256 rows + B 256 rows) x 128 bytes = 128 KiB; pieces, swizzle and fragment reads as above;
//   * schedule: <= 256 persistent workers (XCD-major), worker w takes tiles w, w + workers, ...  No stream-K: at bf16 speed
//     a K tile costs 2 us and a 256 KiB partial slab round trip ~45 us of serial tail -- on every config-3 / config-5 shape
//     whole tiles won, for either tile size (profiles/r03_run3_gemm_bf16_schedules.log).  Wins over the 128 x 128 kernel
//     where the tile count fits the 256 workers well (M = 11968 x N = 1200 / 2160, the N = 9998 output layer, 8192^3: 1.10
//     PFLOP/s against 1.00); the host picks per shape (launch128h).
constexpr int kH2StageFloats = 2 * 256 * 32;
constexpr int kH2Slots = 256;

__global__ __launch_bounds__(512, 1) void gemm256h_kernel(GOp aop, GOp bop, GemmOut out, SkPlan plan, int workers, int wide) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, quad = wave & 3;
  const int wm = (quad >> 1) * 64, wn = grp * 128 + (quad & 1) * 64;
  const int li = lane & 31, lh = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(xcd_major(blockIdx.x, workers));
  constexpr uint32_t kStepBytes = 128;
  const int tiles = plan.dpTiles, kTiles = plan.kTiles;
  if (w >= tiles) return;

  uint32_t va[4], vb[4];
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)aop.p, 0, (int)aop.bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)bop.p, 0, (int)bop.bytes, 0x00020000);
  int bx, by;
  sk_tile_xy(plan, w, bx, by);
  g_init_offs<true>(va, aop, bx * 256, wave, lane);   // piece wave * 4 + j = rows 8 (wave * 4 + j) .. + 8 of the 256-row tile
  g_init_offs<true>(vb, bop, by * 256, wave, lane);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    g_issue1_buf(ra, va[j], 0, smem, wave, j);
    g_issue1_buf(rb, vb[j], 0, smem + 8192, wave, j);
  }
  int stage = 0;
  __syncthreads();

  for (int tile = w; tile < tiles; tile += workers) {
    const bool more = tile + workers < tiles;
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (out.epi & EPI_BIAS) {
      const int nb = by * 256 + wn + 4 * (lane & 15);
#pragma unroll
      for (int e = 0; e < 4; ++e) bv[e] = nb + e < out.N ? out.bias[nb + e] : 0.f;
    }
    f32x16 acc[2][2][2];
#pragma unroll
    for (int si = 0; si < 2; ++si)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[si][i][j][r] = 0.f;

    for (int kt = 0; kt < kTiles; ++kt) {
      const float* As = smem + stage * kH2StageFloats;
      const float* Bs = As + 8192;
      float* An = smem + (stage ^ 1) * kH2StageFloats;
      // to the other stage during this iteration: the next K tile, or the first K tile of the worker's next tile, or (very
      // last iteration of this worker) a harmless re-load of this one
      uint32_t off = kStepBytes * (uint32_t)kt;
      if (kt + 1 < kTiles) {
        off += kStepBytes;
      } else if (more) {
        int nbx, nby;
        sk_tile_xy(plan, tile + workers, nbx, nby);
        g_init_offs<true>(va, aop, nbx * 256, wave, lane);
        g_init_offs<true>(vb, bop, nby * 256, wave, lane);
        off = 0;
      }
      off = (uint32_t)__builtin_amdgcn_readfirstlane((int)off);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        g_issue1_buf(ra, va[j], off, An, wave, j);
        g_issue1_buf(rb, vb[j], off, An + 8192, wave, j);
      }
      bf16x8v_t fa[2][4], fb[2][2];
#pragma unroll
      for (int a = 0; a < 4; ++a) fa[0][a] = h_frag(As, (a >> 1) * 128 + wm + (a & 1) * 32 + li, 0, lh, li);
#pragma unroll
      for (int b = 0; b < 2; ++b) fb[0][b] = h_frag(Bs, wn + b * 32 + li, 0, lh, li);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int cur = s & 1;
        if (s < 3) {
#pragma unroll
          for (int a = 0; a < 4; ++a) fa[cur ^ 1][a] = h_frag(As, (a >> 1) * 128 + wm + (a & 1) * 32 + li, s + 1, lh, li);
#pragma unroll
          for (int b = 0; b < 2; ++b) fb[cur ^ 1][b] = h_frag(Bs, wn + b * 32 + li, s + 1, lh, li);
        }
        __builtin_amdgcn_sched_barrier(0);   // the six reads of k-step s + 1 are in flight under the eight MFMAs of k-step s
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b)
            acc[a >> 1][a & 1][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][a], fb[cur][b], acc[a >> 1][a & 1][b], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      stage ^= 1;
      __syncthreads();
    }

    // `stage` names the buffer holding the prefetched next K tile; the other one is free: eight 8 KiB wave slices
    float* scratch = smem + (stage ^ 1) * kH2StageFloats;
    if (wide) {
      gemm128g_epilogue_wide<true>(out, bx * 256, by * 256 + grp * 128, acc[0], scratch, bv, quad, wave);
      gemm128g_epilogue_wide<true>(out, bx * 256 + 128, by * 256 + grp * 128, acc[1], scratch, bv, quad, wave);
    } else {
      gemm128_epilogue(out, bx * 256, by * 256 + grp * 128, acc[0], quad);
      gemm128_epilogue(out, bx * 256 + 128, by * 256 + grp * 128, acc[1], quad);
    }
    if (!more) break;
    __syncthreads();   // the next iteration's LDS-DMA lands in the slices the epilogue used
    sk_tile_xy(plan, tile + workers, bx, by);
  }
}

// which tile shape: 1 = 256 x 256 (W2L_GEMM_H256=1 / 0 force it on / off for A/B runs; default by problem size)
int h256_mode();
bool sk_forced();
bool ksplit_enabled();   // W2L_GEMM_KSPLIT=0 (probe build) turns the aligned K split off

// A [M][lda], B [N][ldb] bf16 (lda, ldb in bf16 elements, even, >= Kp), Kp = K rounded up to 64: columns K .. Kp of every
// row must be ZERO in both operands (convert.hip writes them so).  W2L_EUNSUPPORTED when the schedule cannot run in-kernel.
// aView / bView (bytes; 0 = a dense image): the address range of an operand whose rows OVERLAP (row stride < Kp: the
// convolution-as-GEMM view of conv.hip, A row m = frames m .. m + kw of the activation image) -- reads past it return zeros
// ta / tb: the operand is k-MAJOR -- A stored [K][lda] (lda >= M), B stored [K][ldb] (ldb >= N), lda / ldb multiples of 8, base
// 16-byte aligned; rows k >= K are never read (the buffer ends there: out-of-range loads return 0), so a k-contiguous partner's
// zero padding is not needed on this side.  128 x 128 kernel only.
inline int launch128h(const uint16_t* A, int lda, const uint16_t* B, int ldb, GemmOut o, int epi, hipStream_t s,
                      unsigned long long aView = 0, unsigned long long bView = 0, bool ta = false, bool tb = false) {
  const int Kp = (o.K + 63) / 64 * 64;
  if ((lda & 1) || (ldb & 1) || (!ta && !aView && lda < Kp) || (!tb && !bView && ldb < Kp) || (((uintptr_t)A | (uintptr_t)B) & 3)) return W2L_EINVAL;
  if ((ta && ((lda & 7) || lda < o.M || (((uintptr_t)A) & 15) || aView)) || (tb && ((ldb & 7) || ldb < o.N || (((uintptr_t)B) & 15) || bView)))
    return W2L_EINVAL;
  const unsigned long long ab = ta ? 2ull * (unsigned long long)o.K * lda : aView ? aView : 2ull * ((unsigned long long)(o.M - 1) * lda + Kp);
  const unsigned long long bb = tb ? 2ull * (unsigned long long)o.K * ldb : bView ? bView : 2ull * ((unsigned long long)(o.N - 1) * ldb + Kp);
  if (ab >= 0x7fffffffull || bb >= 0x7fffffffull) return W2L_EUNSUPPORTED;
  epi &= ~EPI_ATOMIC;
  const double flops = 2.0 * o.M * (double)o.N * o.K;
  o.epi = epi;
  const int wide = (((uintptr_t)o.C) & 15) == 0 && o.ldc % 4 == 0 && (!o.mask || (((uintptr_t)o.mask) & 15) == 0) &&
                   (!o.addend || (((uintptr_t)o.addend) & 15) == 0);
  if (o.imgRows || o.imgTrans || o.maskH || !o.C) {   // images of the result / a bf16 mask image: the wide epilogue only
    if (!wide || (o.N & 3) || o.rowPin || (!o.C && !o.imgRows && !o.imgTrans)) return W2L_EUNSUPPORTED;
    if ((o.imgRows && ((((uintptr_t)o.imgRows) & 7) || (o.ldImgRows & 3) || o.ldImgRows < o.N)) ||
        (o.imgTrans && ((((uintptr_t)o.imgTrans) & 15) || (o.ldImgTrans & 31) || o.ldImgTrans < o.M)) ||
        (o.maskH && ((((uintptr_t)o.maskH) & 7) || (o.ldMaskH & 3) || o.ldMaskH < o.N)))
      return W2L_EINVAL;
  }
  GOp ga{(const float*)A, lda / 2, o.M, (unsigned)ab}, gb{(const float*)B, ldb / 2, o.N, (unsigned)bb};
  // Which kernel: predicted time of whole-tile schedules of the two tile sizes, fitted on MI355X over the config-3 / config-5
  // shapes (profiles/r03_run3_gemm_bf16_schedules.log).  128: a CU holds two workgroups, its busiest one works through
  // ceil(tiles / 256) tile PAIRS at ~0.55 us a K tile; 256: ceil(tiles / 256) tiles at 2.0 us a K tile.  Stream-K only for a
  // handful of tiles with a long reduction (1200 x 1200 x 11968: 106 us against 132), 128 kernel.
  const int kt = Kp / 64;
  const long long tiles1 = (long long)((o.M + 127) / 128) * ((o.N + 127) / 128);
  const long long tiles2 = (long long)((o.M + 255) / 256) * ((o.N + 255) / 256);
  // (128 tiles: a CU works through its tiles at 0.45 us per K tile with two workgroups resident -- 0.54 on long reductions, where
  // the panels fall out of the L2 --, at 0.80 with one; + 5 us of prologue / epilogue per tile)
  // (up to 512 tiles every workgroup is resident from the start and runs at 0.80 us per K tile, shared CU or not)
  const double t1 = tiles1 <= 512 ? kt * 0.80 + 5.0 : (double)((tiles1 + 255) / 256) * (kt * (kt >= 96 ? 0.54 : 0.45) + 5.0);
  const double t2 = (double)((tiles2 + 255) / 256) * (kt * 2.0 + 8.0);
  const int mode = h256_mode();
  const bool big = !ta && !tb && o.M >= 256 && o.N >= 256 && tiles2 < (1 << 30) && (mode == 1 || (mode < 0 && t2 < t1));
  SkPlan plan;
  if (big) {
    plan = make_sk_plan(o.M, o.N, Kp / 2, false, 256, 256, kH2Slots);
  } else {
    // Few tiles, long reduction (the weight gradients: 100 - 300 tiles x 190 - 750 K tiles): whole tiles leave CUs idle, and
    // stream-K ranges share nothing -- every tile streams its full-K operand panels on its own, 2.4 GB through the fabric for
    // 230 MB of operands at 1200 x 1200 x 47936 (383 us).  ALIGNED K split instead: the K axis is cut into X chunks, the same
    // cut for every tile, units (chunk, tile) dealt chunk-major -- the 64 workers of an XCD run neighbouring tiles over the SAME
    // k range and share the panels in its L2; one partial slab per unit, added in chunk order by the tile's last arriver.
    int X = 0;
    if (sk_enabled() && ksplit_enabled() && kt >= 64 && tiles1 * 2 <= 2 * kSkSlots) {
      double bestT = t1;
      for (int x = 2; x <= 8; ++x) {
        const long long units = tiles1 * x;
        const int chunk = (kt + x - 1) / x;
        if (units > 1024 || chunk < 16 || (x - 1) * chunk >= kt) continue;
        const double tx = (double)((units + kSkSlots - 1) / kSkSlots) * (chunk * 1.0 + 6.0) + 15.0 + 4.0 * x;   // + slab round trips
        if (tx < (X ? bestT : 0.8 * t1)) { bestT = tx; X = x; }   // the models are +-15 %: a split must win clearly
      }
    }
    const bool sk = X == 0 && sk_enabled() && (sk_forced() || (tiles1 <= 128 && kt >= 96));
    plan = make_sk_plan(o.M, o.N, Kp / 2, sk);
    if (X) {
      plan.skTiles = plan.dpTiles; plan.dpTiles = 0;
      plan.ksplit = X; plan.kChunk = (kt + X - 1) / X;
      plan.skBlocks = plan.skTiles * X;     // units; the worker count below
    }
    if (plan.skBlocks > 0) {
      plan.slabs = sk_scratch(s, kSkScratchBytes);
      if (plan.slabs && plan.skTiles <= 1024) plan.counters = sk_counters(s);
      if (!plan.slabs || !plan.counters) plan = make_sk_plan(o.M, o.N, Kp / 2, false);
    }
  }
  plan.grouped = 1;
  const int slots = big ? kH2Slots : kSkSlots;
  int workers = plan.dpTiles < slots ? plan.dpTiles : slots;
  if (workers < plan.skBlocks) workers = plan.skBlocks;
  if (workers > slots) workers = slots;
  static const bool attr = hipFuncSetAttribute((const void*)gemm256h_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                               2 * kH2StageFloats * (int)sizeof(float)) == hipSuccess;
  if (big && !attr) return W2L_EHIP;
  prof_begin(s, flops, PROF_GEMM_BF16);
  if (big) {
    hipLaunchKernelGGL(gemm256h_kernel, dim3((unsigned)workers), dim3(512), 2 * (size_t)kH2StageFloats * sizeof(float), s, ga, gb, o,
                       plan, workers, wide);
  } else {
    const size_t shm = 2 * (size_t)kGStageFloats * sizeof(float);
    if (ta && tb) hipLaunchKernelGGL((gemm128h_kernel<false, true, true>), dim3((unsigned)workers), dim3(256), shm, s, ga, gb, o, plan, workers, wide, HGroups{});
    else if (ta) hipLaunchKernelGGL((gemm128h_kernel<false, true, false>), dim3((unsigned)workers), dim3(256), shm, s, ga, gb, o, plan, workers, wide, HGroups{});
    else if (tb) hipLaunchKernelGGL((gemm128h_kernel<false, false, true>), dim3((unsigned)workers), dim3(256), shm, s, ga, gb, o, plan, workers, wide, HGroups{});
    else hipLaunchKernelGGL(gemm128h_kernel<false>, dim3((unsigned)workers), dim3(256), shm, s, ga, gb, o, plan, workers, wide, HGroups{});
  }
  prof_end(s);
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}

// n <= kHMaxGroups problems C_g [M][N] = A_g [M][>= Kp] . B_g [N][>= Kp]^T (+ bias_g) of one shape in one launch (whole 128 x 128
// tiles, grouped over the persistent grid).  Same operand rules as launch128h.
inline int launch128h_grouped(int n, const uint16_t* const* A, int lda, const uint16_t* const* B, int ldb, float* const* C, int ldc,
                              const float* const* bias, int M, int N, int K, hipStream_t s) {
  if (n < 1 || n > kHMaxGroups || M <= 0 || N <= 0 || K <= 0) return W2L_EINVAL;
  const int Kp = (K + 63) / 64 * 64;
  if ((lda & 1) || (ldb & 1) || lda < Kp || ldb < Kp) return W2L_EINVAL;
  const unsigned long long ab = 2ull * ((unsigned long long)(M - 1) * lda + Kp), bb = 2ull * ((unsigned long long)(N - 1) * ldb + Kp);
  if (ab >= 0x7fffffffull || bb >= 0x7fffffffull) return W2L_EUNSUPPORTED;
  HGroups g;
  int wide = (ldc % 4) == 0;
  bool anyBias = false;
  for (int i = 0; i < n; ++i) {
    if (!A[i] || !B[i] || !C[i] || ((((uintptr_t)A[i]) | ((uintptr_t)B[i])) & 3)) return W2L_EINVAL;
    g.a[i] = (const float*)A[i]; g.b[i] = (const float*)B[i]; g.c[i] = C[i]; g.bias[i] = bias ? bias[i] : nullptr;
    anyBias = anyBias || g.bias[i];
    if (((uintptr_t)C[i]) & 15) wide = 0;
  }
  for (int i = n; i < kHMaxGroups; ++i) { g.a[i] = g.a[0]; g.b[i] = g.b[0]; g.c[i] = g.c[0]; g.bias[i] = nullptr; }
  SkPlan plan = make_sk_plan(M, N, Kp / 2, false);
  plan.grouped = 1;
  g.n = n; g.tiles = plan.dpTiles;
  plan.dpTiles *= n;
  GemmOut o{};
  o.C = C[0]; o.bias = nullptr; o.M = M; o.N = N; o.K = K; o.ldc = ldc; o.epi = anyBias ? EPI_BIAS : 0;
  GOp ga{(const float*)A[0], lda / 2, M, (unsigned)ab}, gb{(const float*)B[0], ldb / 2, N, (unsigned)bb};
  const int workers = plan.dpTiles < kSkSlots ? plan.dpTiles : kSkSlots;
  prof_begin(s, 2.0 * n * M * (double)N * K, PROF_GEMM_BF16);
  hipLaunchKernelGGL(gemm128h_kernel<true>, dim3((unsigned)workers), dim3(256), 2 * (size_t)kGStageFloats * sizeof(float), s, ga, gb, o, plan,
                     workers, wide, g);
  prof_end(s);
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}

}  // namespace w2l
