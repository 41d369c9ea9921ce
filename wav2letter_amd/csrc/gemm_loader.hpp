// gemm_loader.hpp -- gemm128g_kernel with a dedicated LOADER wave (producer / consumer specialisation).
//
// Why: on gfx950 vector-memory loads and stores share ONE in-order counter (vmcnt).  In gemm128g_kernel every wave
// both issues LDS-DMA loads and, at the end of a tile, 16 epilogue stores; the `s_waitcnt vmcnt(0)` in front of the
// next K iteration's barrier therefore also waits for those stores to be acknowledged -- and all 512 workgroups
// finish their tiles in lockstep, so the stores of a whole round (32 MB) are in flight at that moment.  Ablations on
// MI355X (profiles/r01_run23_gemm_ablation_buf.log, K = 800): epilogue stores cost nothing when no load waits follow
// them (133 TF/s), 6 % when they do.  Here a fifth wave issues ALL LDS-DMA pieces of the workgroup (32 per K
// iteration) and is the only one that waits on vmcnt; the four compute waves never wait for vector memory inside
// the K loop (raw s_barrier + lgkmcnt only), so their stores drain in the background.  As a bonus the LDS-DMA issue
// slots leave the MFMA waves' instruction streams.
// MEASURED (profiles/r01_run24_gemm_loader_wave_ab.log): correct and bit-identical, but SLOWER -- forward 106 vs 112 TF/s
// on the K = 800 shape, 136 vs 142 at 4096^3, and 72-88 vs 113-146 TF/s when both operands are k-contiguous (one wave
// issuing 32 pieces per iteration makes the issue path itself the critical path).  Kept behind W2L_GEMM_LOADER=1.
// Same tiles, fragments, schedule (persistent segments, stream-K tail, in-kernel slab reduction) and results as
// gemm128g_kernel; every barrier is executed by all five waves (the loader walks the same control flow with the
// work masked off).
#pragma once
#include "gemm_glds.hpp"

namespace w2l {

// compute waves: LDS traffic of this wave retired, then the workgroup barrier.  One asm statement with a "memory"
// clobber: no LDS / global access of the compiler moves across it, and (unlike __syncthreads()) no vmcnt wait.
__device__ __forceinline__ void w_sync_compute() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// loader wave inside the K loop: its LDS-DMA pieces have landed
__device__ __forceinline__ void w_sync_loader() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <bool KC>
__device__ __forceinline__ void w_init_offs(uint32_t (&vo)[16], const GOp& op, int i0, int lane) {
#pragma unroll
  for (int P = 0; P < 16; ++P) {
    if (KC) {
      const int r = P * 8 + (lane >> 3);
      const int c = (lane & 7) ^ ((r >> 1) & 7);
      int gi = i0 + r;
      if (gi > op.extent - 1) gi = op.extent - 1;
      vo[P] = ((uint32_t)gi * (uint32_t)op.ld + 4u * c) * 4u;
    } else {
      const int kr = P * 2 + (lane >> 5);
      int gi = i0 + 4 * (lane & 31);
      if (gi > op.extent - 4) gi = op.extent - 4;
      vo[P] = ((uint32_t)kr * (uint32_t)op.ld + (uint32_t)gi) * 4u;
    }
  }
}

template <bool AKC, bool BKC>
__global__ __launch_bounds__(320, 2) void gemm128w_kernel(GOp aop, GOp bop, GemmOut out, SkPlan plan, int workers, int wide) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..3 compute, 4 loader
  const int w = xcd_major(blockIdx.x, workers);
  GSeg seg = g_segment(plan, w, workers, 0);
  if (!seg.valid) return;
  int stage = 0;

  if (wave == 4) {
    // ================= loader wave: every LDS-DMA piece of the workgroup; mirrors the compute waves' barriers =================
    const uint32_t aStepB = (AKC ? 32u : 32u * (uint32_t)aop.ld) * 4u;
    const uint32_t bStepB = (BKC ? 32u : 32u * (uint32_t)bop.ld) * 4u;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)aop.p, 0, (int)aop.bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)bop.p, 0, (int)bop.bytes, 0x00020000);
    uint32_t va[16], vb[16];
    auto issue_all = [&](uint32_t soA, uint32_t soB, float* stageBase) {
#pragma unroll
      for (int P = 0; P < 16; ++P) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lptr_t)(stageBase + P * 256), 16, (int)va[P], (int)soA, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lptr_t)(stageBase + 4096 + P * 256), 16, (int)vb[P], (int)soB, 0, 0);
      }
    };
    {
      int bx, by;
      sk_tile_xy(plan, seg.tile, bx, by);
      w_init_offs<AKC>(va, aop, bx * 128, lane);
      w_init_offs<BKC>(vb, bop, by * 128, lane);
      issue_all(aStepB * (uint32_t)seg.kb, bStepB * (uint32_t)seg.kb, smem);
      w_sync_loader();
    }
    for (int ord = 0;; ++ord) {
      const GSeg nxt = g_segment(plan, w, workers, ord + 1);
      for (int kt = seg.kb; kt < seg.ke; ++kt) {
        // the next K tile, or the first K tile of the next segment, or (very last iteration) a harmless re-load;
        // every wave has passed the barrier that ended the previous iteration: nobody reads that stage any more
        uint32_t soA = aStepB * (uint32_t)kt, soB = bStepB * (uint32_t)kt;
        if (kt + 1 < seg.ke) {
          soA += aStepB; soB += bStepB;
        } else if (nxt.valid) {
          int nbx, nby;
          sk_tile_xy(plan, nxt.tile, nbx, nby);
          w_init_offs<AKC>(va, aop, nbx * 128, lane);
          w_init_offs<BKC>(vb, bop, nby * 128, lane);
          soA = aStepB * (uint32_t)nxt.kb; soB = bStepB * (uint32_t)nxt.kb;
        }
        issue_all(soA, soB, smem + (stage ^ 1) * kGStageFloats);
        stage ^= 1;
        w_sync_loader();
      }
      if (seg.slab >= 0 && plan.counters) {  // the publish / ticket / acquire barriers of the compute waves
        w_sync_compute();
        w_sync_compute();
        const int ticket = *(const int*)(smem + (stage ^ 1) * kGStageFloats);
        int sF, sL;
        sk_tile_ranges(plan, seg.tile - plan.dpTiles, sF, sL);
        if (ticket == sL - sF) w_sync_compute();
      }
      if (!nxt.valid) break;
      if (wide || plan.counters) w_sync_compute();
      seg = nxt;
    }
    return;
  }

  // ================= compute waves =================
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  const int li = lane & 31, lh = lane >> 5;
  int bx, by;
  sk_tile_xy(plan, seg.tile, bx, by);
  w_sync_compute();  // the loader's first pieces have landed
  for (int ord = 0;; ++ord) {
    const GSeg nxt = g_segment(plan, w, workers, ord + 1);
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
      stage ^= 1;
      w_sync_compute();
    }

    bool doEpi = seg.slab < 0;
    int resetTicket = -1;
    if (!doEpi) {
      gemm128_store_partial(plan.slabs + (size_t)seg.slab * kSlabFloats, acc);
      if (plan.counters) {
        // in-kernel slab reduction, as in gemm128g_kernel; the loader wave takes the same three barriers
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        w_sync_compute();
        int* flag = (int*)(smem + (stage ^ 1) * kGStageFloats);
        const int t = seg.tile - plan.dpTiles;
        if (tid == 0) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          *flag = (int)__hip_atomic_fetch_add(plan.counters + t, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        w_sync_compute();
        const int ticket = *flag;
        int sF, sL;
        sk_tile_ranges(plan, t, sF, sL);
        if (ticket == sL - sF) {
          if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          w_sync_compute();
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
              for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
          for (int sr = sF; sr <= sL; ++sr) {
            const int segIdx = t - (int)(sk_begin(plan, sr) / plan.kTiles);
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
      if (wide) gemm128g_epilogue_wide(out, bx * 128, by * 128, acc, smem + (stage ^ 1) * kGStageFloats, bv);
      else gemm128_epilogue(out, bx * 128, by * 128, acc);
      if (resetTicket >= 0 && tid == 0) __hip_atomic_store(plan.counters + resetTicket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!nxt.valid) break;
    if (wide || plan.counters) w_sync_compute();  // the loader's next pieces land in the slices the epilogue / ticket used

    seg = nxt;
    sk_tile_xy(plan, seg.tile, bx, by);
  }
}

inline int launch128w(const GOp& a, bool akc, const GOp& b, bool bkc, GemmOut o, int epi, hipStream_t s) {
  epi &= ~EPI_ATOMIC;
  SkPlan plan = make_sk_plan(o.M, o.N, o.K, sk_enabled());
  plan.grouped = 1;
  if (plan.skBlocks > 0) {
    plan.slabs = sk_scratch(s, kSkScratchBytes);
    if (!plan.slabs) { plan = make_sk_plan(o.M, o.N, o.K, false); plan.grouped = 1; }
    if (plan.skBlocks > 0 && plan.skTiles <= 1024) plan.counters = sk_counters(s);
  }
  int workers = plan.dpTiles < kSkSlots ? plan.dpTiles : kSkSlots;
  if (workers < plan.skBlocks) workers = plan.skBlocks;
  const size_t shmem = 2 * (size_t)kGStageFloats * sizeof(float);
  static const int wideOn = [] { const char* e = tune_env("W2L_GEMM_WIDE"); return e ? atoi(e) : 1; }();
  const int wide = wideOn && (((uintptr_t)o.C) & 15) == 0 && o.ldc % 4 == 0 &&
                   (!o.mask || (((uintptr_t)o.mask) & 15) == 0) && (!o.addend || (((uintptr_t)o.addend) & 15) == 0);
  dim3 grid((unsigned)workers), block(320);
  o.epi = epi;
  prof_begin(s, 2.0 * o.M * (double)o.N * o.K);
  if (akc && bkc) hipLaunchKernelGGL((gemm128w_kernel<true, true>), grid, block, shmem, s, a, b, o, plan, workers, wide);
  else if (akc) hipLaunchKernelGGL((gemm128w_kernel<true, false>), grid, block, shmem, s, a, b, o, plan, workers, wide);
  else if (bkc) hipLaunchKernelGGL((gemm128w_kernel<false, true>), grid, block, shmem, s, a, b, o, plan, workers, wide);
  else hipLaunchKernelGGL((gemm128w_kernel<false, false>), grid, block, shmem, s, a, b, o, plan, workers, wide);
  if (plan.skBlocks > 0 && !plan.counters)
    hipLaunchKernelGGL(gemm128_fixup<0>, dim3((unsigned)plan.skTiles * 4), dim3(64), 0, s, o, plan);
  prof_end(s);
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}

}  // namespace w2l
