// gemm_t160.hpp -- 128x160 / 160x128 block-tile variant of the LDS-DMA GEMM (gfx950).
//
// Why: every fl::Linear of the TDS recipe has a dimension 80*c or 240*c (c = 10, 14, 18 channels: 800 / 1120 /
// 1440 and 2400 / 3360 / 4320).  128 does not divide them -- N = 800 runs as 7 tile columns = 896 (12 % of the
// MFMA work multiplies clamped duplicates), 1440 as 12 = 1536 (6.25 %) -- but 160 divides all six.  This kernel
// is gemm128g_kernel (gemm_glds.hpp: persistent workers, LDS-DMA staging with swizzled sources, stream-K tail
// with the in-kernel slab reduction) with a 160-wide tile on the side that needs it:
//   WIDE (TALL = false): tile 128 x 160, wave w owns rows [32w, 32w+32) x all 160 columns  (1 x 5 MFMA blocks)
//   TALL (TALL = true) : tile 160 x 128, wave w owns all 160 rows x columns [32w, 32w+32)  (5 x 1 MFMA blocks)
// Per K step of 2 a wave issues 5 v_mfma_f32_32x32x2_f32 from 1 + 5 operand fragments; a K tile is 80 MFMAs,
// 9 LDS-DMA pieces and 24 ds_read_b128 (or up to 96 ds_read_b32 for k-row operands) per wave.  LDS: 36 KiB per
// stage, two stages, two workgroups per CU (144 of 160 KiB).  The arithmetic intensity per staged byte is 11 %
// higher than 128x128 (71 vs 64 flop/B).
// Requirements (host-checked, otherwise gemm128g_kernel runs): as gemm_glds.hpp, plus buffer-addressable
// operands (< 2 GiB), a 16-byte aligned C with ldc % 4 == 0, and <= 1024 stream-K tiles.
#pragma once
#include <type_traits>

#include "gemm_glds.hpp"

namespace w2l {

constexpr int kT160StageFloats = (128 + 160) * 32;  // A tile + B tile of one K step (36 KiB)
constexpr int kT160SlabFloats = 128 * 160;

// byte offset of this lane's 16-byte chunk of piece p (1 KiB, lane-linear in LDS) of an operand tile of W rows (KC:
// [i][32 k], source chunk XOR-swizzled by the row) or W columns (k-rows: [32 k][W], a chunk never straddles a k-row)
template <bool KC, int W>
__device__ __forceinline__ uint32_t t160_off(const GOp& op, int i0, int p, int lane) {
  if (KC) {
    const int r = p * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    int gi = i0 + r;
    if (gi > op.extent - 1) gi = op.extent - 1;
    return ((uint32_t)gi * (uint32_t)op.ld + 4u * c) * 4u;
  } else {
    const int flat = 256 * p + 4 * lane;
    const int kr = flat / W, col = flat - kr * W;
    int gi = i0 + col;
    if (gi >= op.extent) gi = (op.extent - 1) & ~3;  // a chunk that straddles the row end stays in place (its tail columns are never stored)
    return ((uint32_t)kr * (uint32_t)op.ld + (uint32_t)gi) * 4u;
  }
}

// fragments of one 8-k group for NB consecutive 32-row MFMA blocks starting at o0
template <bool KC, int W, int NB>
__device__ __forceinline__ void t160_frag(float (&f)[NB][4], const float* tile, int o0, int g, int li, int lh) {
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    if (KC) {
      const int r = o0 + 32 * b + li;
      const int c = (2 * g) ^ lh ^ ((li >> 1) & 7);
      const f32x4 v = *(const f32x4*)(tile + r * 32 + 4 * c);
      f[b][0] = v[0]; f[b][1] = v[1]; f[b][2] = v[2]; f[b][3] = v[3];
    } else {
      const float* src = tile + (8 * g + 4 * lh) * W + o0 + 32 * b + li;
#pragma unroll
      for (int q = 0; q < 4; ++q) f[b][q] = src[q * W];
    }
  }
}

// partial accumulators of a stream-K range, MFMA register order (coalesced 16-byte stores)
__device__ __forceinline__ void t160_store_partial(float* slab, const f32x16 (&acc)[5], int wave, int lane) {
  f32x4* s4 = (f32x4*)slab;
#pragma unroll
  for (int b = 0; b < 5; ++b)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 v;
      v[0] = acc[b][4 * q]; v[1] = acc[b][4 * q + 1]; v[2] = acc[b][4 * q + 2]; v[3] = acc[b][4 * q + 3];
      s4[((wave * 5 + b) * 4 + q) * 64 + lane] = v;
    }
}

// Epilogue: each 32x32 accumulator block turns through the wave's own 4 KiB of the released LDS stage (ds_write_b32
// in C layout, ds_read_b128 as 8 rows x 8 float4), then 4 global_store_dwordx4 per lane: 128-byte row segments.
// PRE: the mask (or, without one, the addend / accumulate operand) of a block's four row passes is fetched BEFORE the block's LDS
// turn, addresses clamped instead of predicated.  Left in the pass loop, every pass paid its own memory round trip (the row checks
// are branches, and behind a branch the wait for a pending load is vmcnt(0), which on gfx9 also waits for the previous pass's
// store): 20 round trips per tile, +20 ... +45 us per product on the TDS shapes and +136 ... +460 us at M = 11968, N = 1200 /
// 2160 (profiles/r06_run26_gemm_epilogue_operand_cost.log).  Off in the column-sum instances (no such operand there; registers).
template <int MI, int NJ, bool PRE = true>
__device__ __forceinline__ void t160_epilogue(const GemmOut& out, int m0, int n0, const f32x16 (&acc)[5], float* scratch,
                                              const float (&bv)[NJ][4], int wave, int lane) {
  const int EPI = out.epi;
  const int li = lane & 31, lh = lane >> 5;
  float* sc = scratch + wave * 1024;  // [32 rows][32 cols]
  const int c4 = 4 * (lane & 7), rq = lane >> 3;
  const float* accSrc = out.addend ? out.addend : out.C;
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int b = i * NJ + j;
      const int n = n0 + 32 * j + c4;
      const bool fullVec = n + 3 < out.N;
      f32x4 pre[4];
      const bool preMask = PRE && (EPI & EPI_MASK), preAdd = PRE && !preMask && (EPI & EPI_ACCUM);
      if (preMask || preAdd) {
        const float* src = preMask ? out.mask : accSrc;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          int m;
          const bool ok = gemm_out_row(out, m0 + 32 * i + 8 * p + rq, m) && fullVec;
          pre[p] = __builtin_nontemporal_load((const f32x4*)(src + (ok ? (size_t)m * out.ldc + n : (size_t)0)));   // read once: not worth a cache line next to the operand panels
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) sc[((r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + li] = acc[b][r];
      // same wave wrote and reads: LDS operations of one wave complete in order, no barrier needed
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int row = 8 * p + rq;
        const f32x4 v4 = *(const f32x4*)(sc + row * 32 + c4);
        int m;
        if (!gemm_out_row(out, m0 + 32 * i + row, m) || n >= out.N) continue;
        float v[4] = {v4[0] + bv[j][0], v4[1] + bv[j][1], v4[2] + bv[j][2], v4[3] + bv[j][3]};
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
            const f32x4 mk = preMask ? pre[p] : *(const f32x4*)(out.mask + (size_t)m * out.ldc + n);
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
          if (out.ntStore) __builtin_nontemporal_store(w4, (f32x4*)dst); else *(f32x4*)dst = w4;
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
      // the next block overwrites the slice only after this wave's reads have returned (in-order LDS queue)
    }
}

// g_segment with every field pinned to SGPRs.  The schedule arithmetic contains a 64-bit division that the backend
// runs on the VALU; left alone, the (uniform) K-tile index lived in a VGPR and every LDS-DMA issue of the K loop was
// wrapped in a readfirstlane "waterfall" loop for its scalar offset (first build of this kernel: 9 loops per K tile).
__device__ __forceinline__ GSeg t160_segment(const SkPlan& p, int w, int workers, int ord) {
  GSeg s = g_segment(p, w, workers, ord);
  s.tile = __builtin_amdgcn_readfirstlane(s.tile);
  s.kb = __builtin_amdgcn_readfirstlane(s.kb);
  s.ke = __builtin_amdgcn_readfirstlane(s.ke);
  s.slab = __builtin_amdgcn_readfirstlane(s.slab);
  s.valid = __builtin_amdgcn_readfirstlane((int)s.valid) != 0;
  return s;
}

// CS (k-row operands only: the weight gradient x^T dy): the tiles of tile row 0 also add up the B fragments they multiply --
// out.colsum[n] = sum_k B[k][n], the bias gradient (GemmOut::colsum).  WIDE: the four waves hold the same B fragments, wave 0
// sums them (5 v_add per K step on the 1 / tilesM of the tiles that have bx == 0); TALL: every wave sums its own 32 columns.
// A lane's sum runs over the k it holds (k = 8g + 4 (lane >> 5) + q) in ascending order, the two lane halves are added at the
// end of the segment, the segments of a stream-K tile in range order by the tile's last arriver: run-to-run identical.
//
// ADIR (WIDE tiles with a k-contiguous A: the forward and backward-data products): in the 128 x 160 layout a wave multiplies only
// its OWN 32 rows of the A tile, so that operand's trip through LDS is private to the wave -- it is loaded straight into the
// fragment registers instead (four buffer_load_dwordx4 per K tile and wave: lane (li, lh) takes the 16 bytes at k = 8g + 4 lh of
// row 32 wave + li, the k-slot assignment of t160_frag), double-buffered in registers one K tile ahead.  The LDS-DMA stream of a K
// tile shrinks from 9 to 5 pieces per wave, the four ds_read_b128 of the A fragments go away, and the sum order is unchanged
// (bit-identical results).  MEASURED (profiles/r06_run53_gemm_adir_ab.log): level with the all-LDS loop on every shape of the step
// (84.52 / 84.55 ms step-weighted, twice) -- the operand stream costs what it costs whichever path its bytes take into the CU
// (DESIGN section 3.5), so the loop stays off in the product; W2L_GEMM_ADIR=1 (probe library) runs it, and a parity test holds it.
template <bool AKC, bool BKC, bool TALL, bool CS = false, bool ADIR = false>
__global__ __launch_bounds__(256, 2) void gemm160_kernel(GOp aop, GOp bop, GemmOut out, SkPlan plan, int workers) {
  static_assert(!ADIR || (AKC && !TALL && !CS), "A-direct: k-contiguous A on the 128 x 160 tile");
  constexpr int BM = TALL ? 160 : 128, BN = TALL ? 128 : 160;
  constexpr int MI = TALL ? 5 : 1, NJ = TALL ? 1 : 5;
  constexpr int PA = BM / 32, PB = BN / 32;  // LDS-DMA pieces per wave and K tile
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = TALL ? 0 : 32 * wave, wn = TALL ? 32 * wave : 0;
  const int li = lane & 31, lh = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(xcd_major(blockIdx.x, workers));
  const uint32_t aStepB = (AKC ? 32u : 32u * (uint32_t)aop.ld) * 4u;  // bytes per K tile
  const uint32_t bStepB = (BKC ? 32u : 32u * (uint32_t)bop.ld) * 4u;

  GSeg seg = t160_segment(plan, w, workers, 0);
  if (!seg.valid) return;
  const long long dbgT0 = plan.dbg ? wall_clock64() : 0;
  const int second = plan.prio ? g_lds_second() : 0;
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)aop.p, 0, (int)aop.bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)bop.p, 0, (int)bop.bytes, 0x00020000);
  uint32_t va[PA], vb[PB];
  int bx, by;
  sk_tile_xy(plan, seg.tile, bx, by);
  // ADIR: byte offset of this lane's row of the A tile (+ its half's four k); the row is clamped like a piece's
  auto adir_off = [&](int m0) {
    int gi = m0 + wm + li;
    if (gi > aop.extent - 1) gi = aop.extent - 1;
    return ((uint32_t)gi * (uint32_t)aop.ld + 4u * (uint32_t)lh) * 4u;
  };
  uint32_t vaD = 0;
  u32x4 aCur[4], aNxt[4];
  if (ADIR) {
    vaD = adir_off(bx * BM);
#pragma unroll
    for (int g = 0; g < 4; ++g) aCur[g] = __builtin_amdgcn_raw_buffer_load_b128(ra, (int)(vaD + 32u * g), (int)(aStepB * seg.kb), 0);
  } else {
#pragma unroll
    for (int j = 0; j < PA; ++j) va[j] = t160_off<AKC, BM>(aop, bx * BM, wave * PA + j, lane);
  }
#pragma unroll
  for (int j = 0; j < PB; ++j) vb[j] = t160_off<BKC, BN>(bop, by * BN, wave * PB + j, lane);
  if (!ADIR) {
#pragma unroll
    for (int j = 0; j < PA; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lptr_t)(smem + (wave * PA + j) * 256), 16, (int)va[j], (int)(aStepB * seg.kb), 0, 0);
  }
#pragma unroll
  for (int j = 0; j < PB; ++j)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lptr_t)(smem + BM * 32 + (wave * PB + j) * 256), 16, (int)vb[j], (int)(bStepB * seg.kb), 0, 0);
  int stage = 0;
  __syncthreads();  // (drains the LDS-DMA: vmcnt(0) precedes the barrier)

  for (int ord = 0;; ++ord) {
    if (plan.prio) g_tile_prio(plan.prio, second, ord);
    const GSeg nxt = t160_segment(plan, w, workers, ord + 1);
    // bias of this lane's output columns, fetched at the START of the tile (its latency hides under the K loop)
    float bv[NJ][4];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) bv[j][e] = 0.f;
    if (out.epi & EPI_BIAS) {
      const bool bvec = (((uintptr_t)out.bias) & 15) == 0;  // N % 4 == 0 on this path: a column quad is all in or all out
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int nb = by * BN + wn + 32 * j + 4 * (lane & 7);
        if (bvec) {
          const f32x4 t = *(const f32x4*)(out.bias + (nb < out.N ? nb : 0));
#pragma unroll
          for (int e = 0; e < 4; ++e) bv[j][e] = nb < out.N ? t[e] : 0.f;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) bv[j][e] = nb + e < out.N ? out.bias[nb + e] : 0.f;
        }
      }
    }
    f32x16 acc[5];
#pragma unroll
    for (int b = 0; b < 5; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
    float csum[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) csum[j] = 0.f;
    const bool csTile = CS && out.colsum && bx == 0;     // workgroup-uniform
    const bool csOn = csTile && (TALL || wave == 0);      // the waves that deliver the tile's column sums

    // The K loop, instantiated per column-sum role QS so that the loop of an ordinary tile carries neither the adds nor a branch
    // (measured: a scalar branch per K step in every tile cost the weight gradients 2-3.5 %): -1 = none, 4 = every K step (TALL:
    // a wave's own 32 columns), 0 .. 3 = the K steps q == QS of each 8-k group (WIDE: the four waves hold the same B fragments
    // and share the work; their partial sums meet in LDS after the loop).
    auto kloop = [&](auto qsTag) {
    constexpr int QS = decltype(qsTag)::value;
    for (int kt = seg.kb; kt < seg.ke; ++kt) {
      const float* As = smem + stage * kT160StageFloats;
      const float* Bs = As + BM * 32;
      float* An = smem + (stage ^ 1) * kT160StageFloats;
      float* Bn = An + BM * 32;
      float fa[2][MI][4], fb[2][NJ][4];
      if (!ADIR) t160_frag<AKC, BM, MI>(fa[0], As, wm, 0, li, lh);
      t160_frag<BKC, BN, NJ>(fb[0], Bs, wn, 0, li, lh);
      // What goes to the other stage during this iteration: the next K tile, or the first K tile of the next
      // segment, or (very last iteration of this worker) a harmless re-load of this tile.
      uint32_t soA = aStepB * (uint32_t)kt, soB = bStepB * (uint32_t)kt;
      if (kt + 1 < seg.ke) {
        soA += aStepB; soB += bStepB;
      } else if (nxt.valid) {
        int nbx, nby;
        sk_tile_xy(plan, nxt.tile, nbx, nby);
        if (ADIR) vaD = adir_off(nbx * BM);
        else {
#pragma unroll
          for (int j = 0; j < PA; ++j) va[j] = t160_off<AKC, BM>(aop, nbx * BM, wave * PA + j, lane);
        }
#pragma unroll
        for (int j = 0; j < PB; ++j) vb[j] = t160_off<BKC, BN>(bop, nby * BN, wave * PB + j, lane);
        soA = aStepB * (uint32_t)nxt.kb; soB = bStepB * (uint32_t)nxt.kb;
      }
      soA = (uint32_t)__builtin_amdgcn_readfirstlane((int)soA);
      soB = (uint32_t)__builtin_amdgcn_readfirstlane((int)soB);
      // 16 k-steps of 5 MFMAs; the 9 LDS-DMA pieces and the fragment reads of the next 8-k group are slotted
      // BETWEEN k-steps (one filler per gap, order pinned)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int cur = g & 1;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
          for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
              acc[i * NJ + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ADIR ? __uint_as_float(aCur[g][q]) : fa[cur][i][q], fb[cur][j][q], acc[i * NJ + j], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          if (QS == 4 || QS == q) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) csum[j] += fb[cur][j][q];
          }
          if (q == 0) {
            if (g < 3) {
              if (!ADIR) t160_frag<AKC, BM, MI>(fa[cur ^ 1], As, wm, g + 1, li, lh);
              t160_frag<BKC, BN, NJ>(fb[cur ^ 1], Bs, wn, g + 1, li, lh);
            }
          } else {
            const int piece = 3 * g + q - 1;  // steps 1,2,3,5,6,7,9,10,11 -> pieces 0..8
            if (piece < PA) {
              if (ADIR) aNxt[piece] = __builtin_amdgcn_raw_buffer_load_b128(ra, (int)(vaD + 32u * piece), (int)soA, 0);
              else __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lptr_t)(An + (wave * PA + piece) * 256), 16, (int)va[piece], (int)soA, 0, 0);
            } else if (piece < PA + PB)
              __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lptr_t)(Bn + (wave * PB + piece - PA) * 256), 16, (int)vb[piece - PA], (int)soB, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      stage ^= 1;
      __syncthreads();  // the stage just filled has landed (vmcnt(0)) and is visible to all waves
      if (ADIR) {
#pragma unroll
        for (int g = 0; g < 4; ++g) aCur[g] = aNxt[g];
      }
    }
    };
    if (!csTile) kloop(std::integral_constant<int, -1>{});
    else if (TALL) kloop(std::integral_constant<int, 4>{});
    else if (wave == 0) kloop(std::integral_constant<int, 0>{});
    else if (wave == 1) kloop(std::integral_constant<int, 1>{});
    else if (wave == 2) kloop(std::integral_constant<int, 2>{});
    else kloop(std::integral_constant<int, 3>{});

    bool doEpi = seg.slab < 0;  // whole tile: epilogue straight from the accumulators
    int resetTicket = -1;
    if (CS && csTile && !TALL) {   // the four waves' shares meet in the stage the K loop has released (behind the epilogue's slices)
      float* cs = smem + (stage ^ 1) * kT160StageFloats + 4096;
#pragma unroll
      for (int j = 0; j < NJ; ++j) cs[wave * 320 + j * 64 + lane] = csum[j];
      __syncthreads();
      if (wave == 0) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) csum[j] = ((cs[j * 64 + lane] + cs[320 + j * 64 + lane]) + cs[640 + j * 64 + lane]) + cs[960 + j * 64 + lane];
      }
    }
    if (CS && csOn) {   // this segment's column sums: lanes 0 .. 31 hold columns wn + 32 j + lane of the tile
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const float t = csum[j] + __shfl_xor(csum[j], 32);
        const int col = wn + 32 * j + lane;
        if (lane < 32) {
          if (doEpi) { if (by * BN + col < out.N) out.colsum[by * BN + col] = t; }
          else out.csPart[(size_t)seg.slab * 160 + col] = t;
        }
      }
    }
    if (!doEpi) {
      t160_store_partial(plan.slabs + (size_t)seg.slab * kT160SlabFloats, acc, wave, lane);
      // in-kernel slab reduction: see gemm128g_kernel
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      int* flag = (int*)(smem + (stage ^ 1) * kT160StageFloats);  // the stage the K loop has just released
      const int t = seg.tile - plan.dpTiles;
      if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        *flag = (int)__hip_atomic_fetch_add(plan.counters + t, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __syncthreads();
      const int ticket = *flag;
      int sF = 0, sL = plan.ksplit - 1;
      if (!plan.ksplit) sk_tile_ranges(plan, t, sF, sL);
      if (ticket == sL - sF) {  // uniform: last arriver
        if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __syncthreads();  // (also: every wave has read the ticket before the stage becomes epilogue scratch)
#pragma unroll
        for (int b = 0; b < 5; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) csum[j] = 0.f;
        for (int sr = sF; sr <= sL; ++sr) {
          size_t slab;
          if (plan.ksplit) {
            slab = (size_t)sr * plan.skTiles + t;   // chunk sr of this tile
          } else {
            const int segIdx = t - (int)(sk_begin(plan, sr) / plan.kTiles);  // ranges span <= 2 tiles: 0 or 1
            slab = (size_t)sr * 2 + segIdx;
          }
          const f32x4* s4 = (const f32x4*)(plan.slabs + slab * kT160SlabFloats);
#pragma unroll
          for (int b = 0; b < 5; ++b)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const f32x4 v = s4[((wave * 5 + b) * 4 + q) * 64 + lane];
              acc[b][4 * q] += v[0]; acc[b][4 * q + 1] += v[1]; acc[b][4 * q + 2] += v[2]; acc[b][4 * q + 3] += v[3];
            }
          if (CS && csOn && lane < 32) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) csum[j] += out.csPart[slab * 160 + wn + 32 * j + lane];
          }
        }
        if (CS && csOn && lane < 32) {
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            const int n = by * BN + wn + 32 * j + lane;
            if (n < out.N) out.colsum[n] = csum[j];
          }
        }
        doEpi = true;
        resetTicket = t;
      }
    }
    if (doEpi) {
      // `stage` now names the buffer holding the prefetched next K tile; the other one is free
      t160_epilogue<MI, NJ, !CS>(out, bx * BM + wm, by * BN + wn, acc, smem + (stage ^ 1) * kT160StageFloats, bv, wave, lane);
      if (resetTicket >= 0 && tid == 0) __hip_atomic_store(plan.counters + resetTicket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!nxt.valid) break;
    __syncthreads();  // the next iteration's LDS-DMA lands in the slices the epilogue / ticket used

    seg = nxt;
    sk_tile_xy(plan, seg.tile, bx, by);
  }
  if (plan.dbg && tid == 0) g_dbg_record(plan.dbg, dbgT0);
}

// MEASURED (profiles/r02_run17_gemm_aligned_ksplit_negative.log): correct, the L2 sharing is real, and it does not pay --
// 120-127 TF/s against 126-132 of the classic ranges on the six weight-gradient shapes (a wash at K = 24000, 6-8 % slower
// at K = 12000 / 6016): whole rounds of 512 units are 88-93 % full where stream-K balances to the K tile.  Probe library
// only (W2L_GEMM_KSPLIT=1).
inline bool t160_ksplit_enabled() {
  const char* e = tune_env("W2L_GEMM_KSPLIT");
  return e && atoi(e) != 0;
}

// A-direct K loop (gemm160_kernel<..., ADIR>) for the 128 x 160 products with a k-contiguous A: W2L_GEMM_ADIR=1 (probe library)
constexpr int kT160AdirDefault = 0;   // measured level with the all-LDS loop (profiles/r06_run53_gemm_adir_ab.log): kept as a probe variant
inline bool t160_adir_enabled() {
  const char* e = tune_env("W2L_GEMM_ADIR");   // read per call (the variant test flips it)
  return (e ? atoi(e) : kT160AdirDefault) != 0;
}

// 0 = not eligible / not worth it, 1 = WIDE (128x160), 2 = TALL (160x128): the variant whose padded tile area is
// smallest, if it saves at least 2 % of the 128x128 grid's padded area.  W2L_GEMM_T160: 0 = never, 2 = whenever eligible.
inline int t160_choice(const GOp& a, const GOp& b, const GemmOut& o) {
  const char* e = tune_env("W2L_GEMM_T160");
  const int mode = e ? atoi(e) : 1;
  if (!mode || !a.bytes || !b.bytes || o.N % 4 != 0) return 0;
  if ((((uintptr_t)o.C) & 15) != 0 || o.ldc % 4 != 0 || (o.mask && (((uintptr_t)o.mask) & 15) != 0) ||
      (o.addend && (((uintptr_t)o.addend) & 15) != 0))
    return 0;
  auto up = [](int v, int t) { return (double)((v + t - 1) / t * t); };
  const double p128 = up(o.M, 128) * up(o.N, 128);
  const double pw = up(o.M, 128) * up(o.N, 160), pt = up(o.M, 160) * up(o.N, 128);
  const int which = pw <= pt ? 1 : 2;
  const double best = pw <= pt ? pw : pt;
  if (mode >= 2) return mode == 3 ? 2 : which;
  return best <= 0.98 * p128 ? which : 0;
}

// csDone (may be null): set when o.colsum was produced by the launch (k-row operands on this kernel); otherwise the caller sums
inline int launch160(const GOp& a, bool akc, const GOp& b, bool bkc, GemmOut o, int epi, int which, hipStream_t s, bool* launched,
                     bool* csDone = nullptr) {
  epi &= ~EPI_ATOMIC;
  const bool tall = which == 2;
  SkPlan plan = make_sk_plan(o.M, o.N, o.K, sk_enabled(), tall ? 160 : 128, tall ? 128 : 160);
  plan.grouped = 1;
  *launched = false;
  if (plan.skBlocks > 0) {
    if (plan.skTiles > 1024) return W2L_OK;  // more stream-K tiles than arrival tickets: the 128x128 kernel takes it
    plan.slabs = sk_scratch(s, kSkScratchBytes);
    plan.counters = sk_counters(s);
    if (!plan.slabs || !plan.counters) return W2L_OK;
  }
  bool cs = o.colsum && !akc && !bkc;
  if (cs && plan.skBlocks > 0) {
    if (plan.slabs) o.csPart = plan.slabs + kSkSlabBytes / sizeof(float);
    else cs = false;
  }
  if (!cs) o.colsum = nullptr;
  int workers = plan.dpTiles < kSkSlots ? plan.dpTiles : kSkSlots;
  if (workers < plan.skBlocks) workers = plan.skBlocks;
  // aligned K split for a GEMM that is all stream-K (fewer tiles than workgroup slots: the weight gradients): the
  // smallest number of chunks that fills >= 88 % of whole rounds of 512 units, at most 1024 units (one slab each)
  if (plan.skBlocks > 0 && plan.dpTiles == 0 && plan.slabs && t160_ksplit_enabled()) {
    const int tiles = plan.skTiles;
    for (int X = 2; X <= 8; ++X) {
      const int units = tiles * X, rounds = (units + kSkSlots - 1) / kSkSlots;
      const int chunk = (plan.kTiles + X - 1) / X;
      if (units > 2 * kSkSlots || chunk < 8 || (X - 1) * chunk >= plan.kTiles) continue;
      if ((double)units / ((double)rounds * kSkSlots) < 0.88) continue;
      plan.ksplit = X;
      plan.kChunk = chunk;
      workers = units < kSkSlots ? units : kSkSlots;
      break;
    }
  }
  plan.dbg = gemm_dbg_ptr();
  plan.prio = gemm_prio_mode();
  { static const int nt = [] { const char* e = tune_env("W2L_GEMM_NTSTORE"); return e ? atoi(e) : 0; }(); o.ntStore = nt; }
  const size_t shmem = 2 * (size_t)kT160StageFloats * sizeof(float);
  dim3 grid((unsigned)workers), block(256);
  o.epi = epi;
  static bool attr[64] = {};  // per device: the 72 KiB dynamic LDS opt-in is a property of the function ON a device
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64) dev = 0;
  if (!attr[dev]) {
    attr[dev] = true;
#define W2L_T160_ATTR(A, B, T) (void)hipFuncSetAttribute((const void*)gemm160_kernel<A, B, T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem)
    W2L_T160_ATTR(true, true, false); W2L_T160_ATTR(true, false, false); W2L_T160_ATTR(false, true, false); W2L_T160_ATTR(false, false, false);
    W2L_T160_ATTR(true, true, true); W2L_T160_ATTR(true, false, true); W2L_T160_ATTR(false, true, true); W2L_T160_ATTR(false, false, true);
    (void)hipFuncSetAttribute((const void*)gemm160_kernel<false, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    (void)hipFuncSetAttribute((const void*)gemm160_kernel<false, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    (void)hipFuncSetAttribute((const void*)gemm160_kernel<true, true, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    (void)hipFuncSetAttribute((const void*)gemm160_kernel<true, false, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
#undef W2L_T160_ATTR
  }
  prof_begin(s, 2.0 * o.M * (double)o.N * o.K, PROF_GEMM128, o.M, o.N, o.K, tall ? 4 : 3);
#define W2L_T160_GO(A, B, T) hipLaunchKernelGGL((gemm160_kernel<A, B, T>), grid, block, shmem, s, a, b, o, plan, workers)
  if (cs) {
    if (!tall) hipLaunchKernelGGL((gemm160_kernel<false, false, false, true>), grid, block, shmem, s, a, b, o, plan, workers);
    else hipLaunchKernelGGL((gemm160_kernel<false, false, true, true>), grid, block, shmem, s, a, b, o, plan, workers);
    if (csDone) *csDone = true;
  } else if (!tall && akc && t160_adir_enabled()) {
    if (bkc) hipLaunchKernelGGL((gemm160_kernel<true, true, false, false, true>), grid, block, shmem, s, a, b, o, plan, workers);
    else hipLaunchKernelGGL((gemm160_kernel<true, false, false, false, true>), grid, block, shmem, s, a, b, o, plan, workers);
  } else if (!tall) {
    if (akc && bkc) W2L_T160_GO(true, true, false);
    else if (akc) W2L_T160_GO(true, false, false);
    else if (bkc) W2L_T160_GO(false, true, false);
    else W2L_T160_GO(false, false, false);
  } else {
    if (akc && bkc) W2L_T160_GO(true, true, true);
    else if (akc) W2L_T160_GO(true, false, true);
    else if (bkc) W2L_T160_GO(false, true, true);
    else W2L_T160_GO(false, false, true);
  }
#undef W2L_T160_GO
  prof_end(s);
  *launched = true;
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}

}  // namespace w2l
