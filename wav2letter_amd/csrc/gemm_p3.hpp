// gemm_p3.hpp -- fp32 MFMA GEMM, 256x128x32 block tile, 8 waves, THREE LDS stages with counted waits.
//
// Third generation of the fl::Linear GEMM main loop (see gemm.hip for the reference call sites and
// gemm_glds.hpp for the second generation, which this kernel extends).  What changes and why:
//   * gemm128g_kernel ends every K iteration with __syncthreads(), whose fence drains the LDS-DMA queue
//     (vmcnt(0)): an operand tile has exactly ONE iteration to arrive.  In the steady state that is enough,
//     but the first K tile of every new output tile misses L2 (new A panel) while all 512 workgroups switch
//     tiles in lockstep: measured per-tile overhead 7.6 K-iterations on the TDS fc shapes (K = 800: 115 TF/s
//     against 142 TF/s at 4096^3, MI355X).  Here the barrier is a raw s_barrier and the wait a counted
//     s_waitcnt vmcnt(6): the pieces of tile t+1 stay in flight across the barrier while tile t is
//     multiplied and tile t+2 is being issued -- two iterations (~16k cycles) of slack, across output-tile
//     boundaries too (the issue cursor simply runs two iterations ahead in the worker's segment list).
//   * 256x128 block tile, 8 waves (4x2) of 64x64: 48 KiB per stage = 6 LDS-DMA pieces per wave and
//     iteration instead of 8 (the buffer_load..lds issue cost was the largest single loss of the 128x128
//     kernel: 143 -> 126 TF/s with / without staging at 4096^3).  3 stages = 144 KiB: one workgroup per CU,
//     still two waves per SIMD.
// Everything else (XOR-swizzled source addresses, ds_read_b128 fragments, k-slot permutation, wide epilogue
// through the released stage, persistent workers with XCD-major grouped rasterisation, stream-K tail with
// deterministic slab fix-up) is the second generation's scheme, re-dimensioned.
#pragma once
#include "gemm_glds.hpp"

namespace w2l {

constexpr int kP3StageFloats = (256 + 128) * 32;      // A tile 256x32 + B tile 128x32
constexpr int kP3SlabFloats = 256 * 128;
constexpr int kP3Slots = 256;                         // one 512-thread workgroup per CU

inline SkPlan make_sk_plan_p3(int M, int N, int K, bool allowSk) {
  SkPlan p;
  p.tilesM = (M + 255) / 256;
  p.tilesN = (N + 127) / 128;
  p.kTiles = (K + 31) / 32;
  const int tiles = p.tilesM * p.tilesN;
  p.dpTiles = tiles; p.skTiles = 0; p.skBlocks = 0; p.slabs = nullptr; p.grouped = 1; p.counters = nullptr;
  if (!allowSk || p.kTiles < 8) return p;
  const int rounds = (tiles + kP3Slots - 1) / kP3Slots;
  const double eff = (double)tiles / ((double)rounds * kP3Slots);
  if (eff >= 0.93) return p;
  const int full = tiles / kP3Slots;
  p.dpTiles = full * kP3Slots;
  p.skTiles = tiles - p.dpTiles;
  long long iters = (long long)p.skTiles * p.kTiles;
  long long blocks = iters / 4;  // >= 4 K iterations per workgroup
  if (blocks > kP3Slots) blocks = kP3Slots;
  if (blocks < 1) blocks = 1;
  p.skBlocks = (int)blocks;
  if (iters / p.skBlocks + 1 > p.kTiles) { p.dpTiles = tiles; p.skTiles = 0; p.skBlocks = 0; }  // ranges span <= 2 tiles
  return p;
}

// per-lane byte offsets of this wave's LDS-DMA pieces: NP pieces per wave, tile of 8 NP x 8 rows (k-contiguous)
// or 32 k-rows x ROWS columns (k-rows)
template <bool KC, int ROWS, int NP>
__device__ __forceinline__ void p3_init_offs(uint32_t (&vo)[NP], const GOp& op, int i0, int wave, int lane) {
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    if (KC) {
      const int r = (wave * NP + j) * 8 + (lane >> 3);             // tile row
      const int c = (lane & 7) ^ ((r >> 1) & 7);                   // source chunk of this LDS slot (XOR swizzle)
      int gi = i0 + r;
      if (gi > op.extent - 1) gi = op.extent - 1;
      vo[j] = ((uint32_t)gi * (uint32_t)op.ld + 4u * c) * 4u;
    } else {
      // k-row pieces: 1 KiB = 256 floats; ROWS = 256: one k-row per piece, ROWS = 128: two
      const int e = (wave * NP + j) * 256 + 4 * lane;              // float index inside the [32][ROWS] tile
      const int kr = e / ROWS;
      int gi = i0 + (e - kr * ROWS);
      if (gi > op.extent - 4) gi = op.extent - 4;
      vo[j] = ((uint32_t)kr * (uint32_t)op.ld + (uint32_t)gi) * 4u;
    }
  }
}

template <bool KC, int ROWS>
__device__ __forceinline__ void p3_frag(float (&f)[2][4], const float* tile, int w0, int g, int li, int lh) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    if (KC) {
      const int r = w0 + 32 * i + li;
      const int c = (2 * g) ^ lh ^ ((li >> 1) & 7);
      const f32x4 v = *(const f32x4*)(tile + r * 32 + 4 * c);
      f[i][0] = v[0]; f[i][1] = v[1]; f[i][2] = v[2]; f[i][3] = v[3];
    } else {
      const float* src = tile + (8 * g + 4 * lh) * ROWS + w0 + 32 * i + li;
#pragma unroll
      for (int q = 0; q < 4; ++q) f[i][q] = src[q * ROWS];
    }
  }
}

// wide epilogue, 16-row passes through a 4 KiB slice per wave (8 waves share one released 48 KiB stage)
__device__ __forceinline__ void p3_epilogue_wide(const GemmOut& out, int m0, int n0, const f32x16 (&acc)[2][2], float* scratch,
                                                 int wave, int lane) {
  const int EPI = out.epi;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  const int li = lane & 31, lh = lane >> 5;
  float* sc = scratch + wave * 1024;       // [16 rows][64 cols]
  const int c4 = 4 * (lane & 15), rq = lane >> 4;
  const int n = n0 + wn + c4;
  float bv[4] = {0.f, 0.f, 0.f, 0.f};
  if (EPI & EPI_BIAS) {
#pragma unroll
    for (int e = 0; e < 4; ++e) bv[e] = n + e < out.N ? out.bias[n + e] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      // rows 16h .. 16h+15 of the 32-row MFMA tile live in registers 8h .. 8h+7
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r8 = 0; r8 < 8; ++r8) sc[((r8 & 3) + 8 * (r8 >> 2) + 4 * lh) * 64 + j * 32 + li] = acc[i][j][8 * h + r8];
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int row = 4 * p + rq;
        const f32x4 v4 = *(const f32x4*)(sc + row * 64 + c4);
        const int m = m0 + wm + 32 * i + 16 * h + row;
        if (m >= out.M || n >= out.N) continue;
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
        const float* accSrc = out.addend ? out.addend : out.C;
        if (n + 3 < out.N) {
          if (EPI & EPI_MASK) {
            const f32x4 mk = *(const f32x4*)(out.mask + (size_t)m * out.ldc + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = mk[e] > 0.f ? v[e] * out.maskScale : 0.f;
          }
          if (EPI & EPI_ACCUM) {
            const f32x4 o = *(const f32x4*)(accSrc + (size_t)m * out.ldc + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += o[e];
          }
          f32x4 w4;
          w4[0] = v[0]; w4[1] = v[1]; w4[2] = v[2]; w4[3] = v[3];
          *(f32x4*)dst = w4;
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
    }
}

// slab float4 of (acc index a = 2i+j, register quad q) of this thread: ((wave*4 + a)*4 + q)*64 + lane, wave = 0..7
__device__ __forceinline__ void p3_store_partial(float* slab, const f32x16 (&acc)[2][2], int wave, int lane) {
  f32x4* s4 = (f32x4*)slab;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 v;
        v[0] = acc[i][j][4 * q]; v[1] = acc[i][j][4 * q + 1]; v[2] = acc[i][j][4 * q + 2]; v[3] = acc[i][j][4 * q + 3];
        s4[((wave * 4 + i * 2 + j) * 4 + q) * 64 + lane] = v;
      }
}

// a cursor over the worker's (tile, K-tile) iteration stream
struct P3Cursor {
  GSeg seg;
  int ord, kt;
};
__device__ __forceinline__ void p3_cursor_init(P3Cursor& c, const SkPlan& plan, int w, int workers) {
  c.ord = 0;
  c.seg = g_segment(plan, w, workers, 0);
  c.kt = c.seg.kb;
}
// returns true when the cursor entered a new segment
__device__ __forceinline__ bool p3_cursor_next(P3Cursor& c, const SkPlan& plan, int w, int workers) {
  if (!c.seg.valid) return false;
  if (++c.kt < c.seg.ke) return false;
  c.seg = g_segment(plan, w, workers, ++c.ord);
  c.kt = c.seg.kb;
  return true;
}

template <bool AKC, bool BKC>
__global__ __launch_bounds__(512, 1) void gemm256_kernel(GOp aop, GOp bop, GemmOut out, SkPlan plan, int workers, int wide) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..7, scalar
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  const int li = lane & 31, lh = lane >> 5;
  const int w = xcd_major(blockIdx.x, workers);
  const uint32_t aStepB = (AKC ? 32u : 32u * (uint32_t)aop.ld) * 4u;   // bytes per K tile
  const uint32_t bStepB = (BKC ? 32u : 32u * (uint32_t)bop.ld) * 4u;
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)aop.p, 0, (int)aop.bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)bop.p, 0, (int)bop.bytes, 0x00020000);

  P3Cursor cc, ic;                       // compute cursor, issue cursor (runs two iterations ahead)
  p3_cursor_init(cc, plan, w, workers);
  if (!cc.seg.valid) return;
  ic = cc;
  uint32_t va[4], vb[2];
  auto set_tile = [&](int tile) {
    int tx, ty;
    sk_tile_xy(plan, tile, tx, ty);
    p3_init_offs<AKC, 256, 4>(va, aop, tx * 256, wave, lane);
    p3_init_offs<BKC, 128, 2>(vb, bop, ty * 128, wave, lane);
  };
  auto issue_piece = [&](int piece, int kt, float* stageBase) {
    if (piece < 4)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lptr_t)(stageBase + (wave * 4 + piece) * 256), 16, (int)va[piece],
                                               (int)(aStepB * (uint32_t)kt), 0, 0);
    else
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lptr_t)(stageBase + 8192 + (wave * 2 + piece - 4) * 256), 16,
                                               (int)vb[piece - 4], (int)(bStepB * (uint32_t)kt), 0, 0);
  };
  // prologue: iterations 0 and 1 of the stream into stages 0 and 1 (a missing second iteration re-loads the first)
  set_tile(ic.seg.tile);
  int ikt = ic.kt;                       // K tile the issue cursor points at (for the harmless re-load at the end)
#pragma unroll
  for (int p = 0; p < 6; ++p) issue_piece(p, ikt, smem);
  if (p3_cursor_next(ic, plan, w, workers) && ic.seg.valid) set_tile(ic.seg.tile);
  if (ic.seg.valid) ikt = ic.kt;
#pragma unroll
  for (int p = 0; p < 6; ++p) issue_piece(p, ikt, smem + kP3StageFloats);

  int sC = 0;                            // stage holding the compute cursor's K tile; the issue stage is (sC + 2) % 3
  int bx, by;
  sk_tile_xy(plan, cc.seg.tile, bx, by);
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  while (true) {
    // advance the issue cursor to iteration t+2 (its offsets are needed from the first filler slot on)
    if (p3_cursor_next(ic, plan, w, workers) && ic.seg.valid) set_tile(ic.seg.tile);
    if (ic.seg.valid) ikt = ic.kt;       // else: harmless re-load of the last tile (keeps the vmcnt count uniform)
    // the K tile of this iteration has landed once at most the 6 newest pieces (tile t+1) are outstanding
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_s_barrier();        // ... in every wave; and every wave is done reading stage (sC+2)%3
    const float* As = smem + sC * kP3StageFloats;
    const float* Bs = As + 8192;
    const int sI = sC >= 1 ? sC - 1 : 2;  // (sC + 2) % 3
    float* Is = smem + sI * kP3StageFloats;
    float fa[2][2][4], fb[2][2][4];
    p3_frag<AKC, 256>(fa[0], As, wm, 0, li, lh);
    p3_frag<BKC, 128>(fb[0], Bs, wn, 0, li, lh);
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
        if (q == 0) {
          if (g < 3) {
            p3_frag<AKC, 256>(fa[cur ^ 1], As, wm, g + 1, li, lh);
            p3_frag<BKC, 128>(fb[cur ^ 1], Bs, wn, g + 1, li, lh);
          }
        } else if (g < 2) {
          issue_piece(3 * g + q - 1, ikt, Is);  // steps 1,2,3,5,6,7 -> pieces 0..5
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    sC = sC == 2 ? 0 : sC + 1;
    const bool tileDone = cc.kt + 1 == cc.seg.ke;
    if (tileDone) {
      const GSeg done = cc.seg;
      if (done.slab < 0) {
        if (wide) {
          // every wave must be done with the stage just consumed before it becomes epilogue scratch; the
          // barrier at the top of the next iteration then orders the scratch reads before the next LDS-DMA
          __builtin_amdgcn_s_barrier();
          p3_epilogue_wide(out, bx * 256, by * 128, acc, smem + (sC == 0 ? 2 : sC - 1) * kP3StageFloats, wave, lane);
        } else {
          gemm128_epilogue(out, bx * 256, by * 128, acc, wave);
        }
      } else {
        p3_store_partial(plan.slabs + (size_t)done.slab * kP3SlabFloats, acc, wave, lane);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    }
    p3_cursor_next(cc, plan, w, workers);
    if (!cc.seg.valid) break;
    if (tileDone) sk_tile_xy(plan, cc.seg.tile, bx, by);
  }
}

// one wavefront per 64x64 quadrant of a stream-K tile (8 per tile): slabs added in range order, then the epilogue
__global__ __launch_bounds__(64) void gemm256_fixup(GemmOut out, SkPlan plan) {
  const int t = blockIdx.x >> 3;
  const int wave = blockIdx.x & 7, lane = threadIdx.x;
  const long long tb = (long long)t * plan.kTiles, te = tb + plan.kTiles;
  const long long I = (long long)plan.skTiles * plan.kTiles;
  int s = (int)(tb * plan.skBlocks / I);
  while (s + 1 < plan.skBlocks && sk_begin(plan, s + 1) <= tb) ++s;
  while (s > 0 && sk_begin(plan, s) > tb) --s;
  if (sk_begin(plan, s) <= tb && sk_begin(plan, s + 1) >= te) return;  // one range covered the tile: done in-kernel
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  for (; s < plan.skBlocks && sk_begin(plan, s) < te; ++s) {
    const long long b0 = sk_begin(plan, s);
    if (sk_begin(plan, s + 1) <= tb) continue;
    const int segIdx = t - (int)(b0 / plan.kTiles);  // ranges span <= 2 tiles: 0 or 1
    const f32x4* s4 = (const f32x4*)(plan.slabs + ((size_t)s * 2 + segIdx) * kP3SlabFloats);
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
  int bx, by;
  sk_tile_xy(plan, plan.dpTiles + t, bx, by);
  gemm128_epilogue(out, bx * 256, by * 128, acc, wave);
}

inline int launch256(const GOp& a, bool akc, const GOp& b, bool bkc, GemmOut o, int epi, hipStream_t s) {
  epi &= ~EPI_ATOMIC;
  SkPlan plan = make_sk_plan_p3(o.M, o.N, o.K, sk_enabled());
  if (plan.skBlocks > 0) {
    plan.slabs = sk_scratch(s, kSkScratchBytes);  // same 64 MiB: 256 x 2 x 128 KiB
    if (!plan.slabs) plan = make_sk_plan_p3(o.M, o.N, o.K, false);
  }
  int workers = plan.dpTiles < kP3Slots ? plan.dpTiles : kP3Slots;
  if (workers < plan.skBlocks) workers = plan.skBlocks;
  const size_t shmem = 3 * (size_t)kP3StageFloats * sizeof(float);
  static const int wideOn = [] { const char* e = tune_env("W2L_GEMM_WIDE"); return e ? atoi(e) : 1; }();
  const int wide = wideOn && (((uintptr_t)o.C) & 15) == 0 && o.ldc % 4 == 0 && (!o.mask || (((uintptr_t)o.mask) & 15) == 0) &&
                   (!o.addend || (((uintptr_t)o.addend) & 15) == 0);
  dim3 grid((unsigned)workers), block(512);
  o.epi = epi;
#define W2L_P3_LAUNCH(AK, BK)                                                                                            \
  do {                                                                                                                   \
    static bool attr = false;                                                                                            \
    if (!attr) {                                                                                                         \
      W2L_HIP_CHECK(hipFuncSetAttribute((const void*)gemm256_kernel<AK, BK>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                        (int)shmem));                                                                    \
      attr = true;                                                                                                       \
    }                                                                                                                    \
    hipLaunchKernelGGL((gemm256_kernel<AK, BK>), grid, block, shmem, s, a, b, o, plan, workers, wide);                   \
  } while (0)
  prof_begin(s, 2.0 * o.M * (double)o.N * o.K);
  if (akc && bkc) W2L_P3_LAUNCH(true, true);
  else if (akc) W2L_P3_LAUNCH(true, false);
  else if (bkc) W2L_P3_LAUNCH(false, true);
  else W2L_P3_LAUNCH(false, false);
#undef W2L_P3_LAUNCH
  if (plan.skBlocks > 0) hipLaunchKernelGGL(gemm256_fixup, dim3((unsigned)plan.skTiles * 8), dim3(64), 0, s, o, plan);
  prof_end(s);
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}

}  // namespace w2l
