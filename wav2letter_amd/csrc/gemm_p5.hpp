// gemm_p5.hpp -- EXPERIMENT (probe library, W2L_GEMM_P5=1): the 128 x 128 x 32 fp32 LDS-DMA GEMM with a PRODUCER wave.
//
// Why: the ablations of round 6 (profiles/r06_run29_*, r06_run43_*) say that what the recipe shapes lose against 4096^3 scales
// with the number of `buffer_load ... lds` instructions a wave issues between its MFMAs -- not with the barrier, not with the
// fragment reads, not with where the operands come from -- and that one workgroup alone on a CU multiplies as fast as two
// (r06_run36_*).  Here the four MFMA waves issue NO vector-memory instruction in the K loop: a fifth wave stages every piece of
// every K tile, two K tiles ahead, through three LDS stages (one workgroup per CU: 3 x 32 KiB of stages + 32 KiB of epilogue
// scratch of its own, so that the epilogue of tile n never waits for a free stage and the producer keeps prefetching tile n + 1
// through it).  One `s_barrier` per K tile (inline asm: hipcc drains vmcnt / lgkmcnt in front of a barrier it can see, which
// would make the producer wait for the stage it has just started to fill).
// Whole-tile schedules only (no stream-K), buffer addressing, the wide epilogue.
#pragma once
#include "gemm_glds.hpp"

namespace w2l {

constexpr int kP5Stages = 3;
constexpr int kP5Threads = 320;
constexpr size_t kP5Shmem = (size_t)(kP5Stages + 1) * kGStageFloats * sizeof(float);   // + one stage-sized epilogue scratch

template <bool AKC, bool BKC>
__global__ __launch_bounds__(kP5Threads, 1) void gemm128p_kernel(GOp aop, GOp bop, GemmOut out, SkPlan plan, int workers, int wide) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int w = __builtin_amdgcn_readfirstlane(xcd_major(blockIdx.x, workers));
  if (w >= plan.dpTiles) return;
  const int nT = (plan.dpTiles - w + workers - 1) / workers;
  const int kT = plan.kTiles;
  const int n = nT * kT;   // K tiles this worker walks
  float* scratch = smem + kP5Stages * kGStageFloats;

  if (wave == 4) {   // ---- producer
    const uint32_t aStepB = (AKC ? 32u : 32u * (uint32_t)aop.ld) * 4u, bStepB = (BKC ? 32u : 32u * (uint32_t)bop.ld) * 4u;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)aop.p, 0, (int)aop.bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)bop.p, 0, (int)bop.bytes, 0x00020000);
    uint32_t va[4][4], vb[4][4];
    auto issue = [&](int it) {
      const int ti = it / kT, kt = it - ti * kT;
      if (kt == 0) {
        int bx, by;
        sk_tile_xy(plan, w + ti * workers, bx, by);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          g_init_offs<AKC>(va[q], aop, bx * 128, q, lane);
          g_init_offs<BKC>(vb[q], bop, by * 128, q, lane);
        }
      }
      float* st = smem + (it % kP5Stages) * kGStageFloats;
      const uint32_t soA = (uint32_t)__builtin_amdgcn_readfirstlane((int)(aStepB * (uint32_t)kt));
      const uint32_t soB = (uint32_t)__builtin_amdgcn_readfirstlane((int)(bStepB * (uint32_t)kt));
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          g_issue1_buf(ra, va[q][j], soA, st, q, j);
          g_issue1_buf(rb, vb[q][j], soB, st + 4096, q, j);
        }
    };
    issue(0);
    if (n > 1) issue(1);
    for (int it = 0; it < n; ++it) {
      // K tile `it` has landed when at most the 32 pieces of `it + 1` are outstanding
      if (it + 1 < n) asm volatile("s_waitcnt vmcnt(32)\n\ts_barrier" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
      if (it + 2 < n) issue(it + 2);   // into the stage the consumers left at the barrier above
    }
    return;
  }

  // ---- consumers: gemm128g_kernel's K step without its LDS-DMA fillers
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  const int li = lane & 31, lh = lane >> 5;
  int it = 0;
  for (int ti = 0; ti < nT; ++ti) {
    int bx, by;
    sk_tile_xy(plan, w + ti * workers, bx, by);
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
    for (int kt = 0; kt < kT; ++kt, ++it) {
      asm volatile("s_barrier" ::: "memory");
      const float* As = smem + (it % kP5Stages) * kGStageFloats;
      const float* Bs = As + 4096;
      float fa[2][2][4], fb[2][2][4];
      g_frag<AKC>(fa[0], As, wm, 0, li, lh);
      g_frag<BKC>(fb[0], Bs, wn, 0, li, lh);
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
          if (q == 0 && g < 3) {
            g_frag<AKC>(fa[cur ^ 1], As, wm, g + 1, li, lh);
            g_frag<BKC>(fb[cur ^ 1], Bs, wn, g + 1, li, lh);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    if (wide) gemm128g_epilogue_wide(out, bx * 128, by * 128, acc, scratch, bv);
    else gemm128_epilogue(out, bx * 128, by * 128, acc);
  }
}

// true when the experiment kernel took the product (probe library, W2L_GEMM_P5=1, whole-tile plans)
inline bool launch128p(const GOp& a, bool akc, const GOp& b, bool bkc, GemmOut o, int epi, const SkPlan& plan0, int wide, hipStream_t s) {
  static const int on = [] { const char* e = tune_env("W2L_GEMM_P5"); return e ? atoi(e) : 0; }();
  if (!on || plan0.skBlocks > 0 || !a.bytes || !b.bytes) return false;
  SkPlan plan = plan0;
  const int workers = plan.dpTiles < 256 ? plan.dpTiles : 256;
  o.epi = epi;
  static bool attr = false;
  if (!attr) {
    attr = true;
    (void)hipFuncSetAttribute((const void*)gemm128p_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kP5Shmem);
    (void)hipFuncSetAttribute((const void*)gemm128p_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kP5Shmem);
    (void)hipFuncSetAttribute((const void*)gemm128p_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kP5Shmem);
    (void)hipFuncSetAttribute((const void*)gemm128p_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kP5Shmem);
  }
  dim3 grid((unsigned)workers), block(kP5Threads);
  if (akc && bkc) hipLaunchKernelGGL((gemm128p_kernel<true, true>), grid, block, kP5Shmem, s, a, b, o, plan, workers, wide);
  else if (akc) hipLaunchKernelGGL((gemm128p_kernel<true, false>), grid, block, kP5Shmem, s, a, b, o, plan, workers, wide);
  else if (bkc) hipLaunchKernelGGL((gemm128p_kernel<false, true>), grid, block, kP5Shmem, s, a, b, o, plan, workers, wide);
  else hipLaunchKernelGGL((gemm128p_kernel<false, false>), grid, block, kP5Shmem, s, a, b, o, plan, workers, wide);
  return true;
}

}  // namespace w2l
