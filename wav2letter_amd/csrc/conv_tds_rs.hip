// conv_tds_rs.hip -- the TDS time convolution proper (fl::TDSBlock's Conv2D kw x 1, C -> C channels with
// C = 10 / 14 / 18, kw = 21, stride 1, H = 80 mel rows; recipes/sota/2019/am_arch/am_tds_ctc.arch:3-37,
// builder recipes/joint_training_vox_populi/cpc/SequentialBuilder.cpp:254-268) forward and backward-data in the
// ROLE-SWAPPED formulation SURVEY 7 hard-part 3(b) names.
//
// Why: with N = C_out on the MFMA columns (conv_tds.hip) 10 / 14 / 18 columns sit in 16 / 16 / 32-wide tiles --
// a 62.5 / 87.5 / 56 % ceiling before the first instruction is issued -- and every 32-cycle 16x16x4 MFMA needs its
// own LDS fragment read: the loop ran issue-bound at 34 % of the fp32 MFMA peak (round-1 verdict, weak 4).
//
// Here the kw taps are cut into G groups of J consecutive taps and the GROUP index joins the output channel on
// the MFMA column axis:
//     Y[tau][(g, co)] = sum_{j < J, ci}  X[tau + j][ci] * W[g*J + j][ci][co]          (one 32x32x2 MFMA chain)
//     out[u][co]      = sum_{g < G}      Y[u + g*J][(g, co)]                          (overlap-add through LDS)
// rows tau = 32 consecutive FRAMES of one (utterance, mel row), K = (j, ci) = J*C, columns (g, co) = G*C:
//     C = 10: G = 3, J = 7 :  30 of  32 columns, K =  70 (35 MFMAs per 32x32 tile)            93.8 % of the lanes useful
//     C = 14: G = 2, J = 11 :  28 of  32 columns, K = 154 (77 MFMAs), 21 of 22 taps                  83.5 %
//             (G = 11, J = 2 would use 91.9 % of the lanes but needs 80 overlap-add registers per 70 MFMAs, and rows
//              4 apart collide inside ONE instruction when J divides 4)
//     C = 18: G = 7, J = 3 : 126 of 128 columns, K =  54 (4 column tiles x 27 MFMAs)                98.4 %
// v_mfma_f32_32x32x2_f32 issues every 64 cycles (twice the 16x16x4 budget per instruction) and
//   * the B operand (the weights of a column tile) lives in REGISTERS for the whole persistent kernel
//     (35 / 70 / 108 VGPRs): no B fragment reads at all;
//   * the A operand is one ds_read_b32 per K step and row tile, shared by all column tiles (the slab is stored
//     TIME-FASTEST, slab[(h, ci)][frame], so the 32 lanes of a fragment read 32 consecutive floats: conflict-free,
//     and every (j, ci) of the K loop is an immediate offset from one address VGPR);
//   * the overlap-add is a read-modify-write of out[(h, co)][u] in LDS by the ONE wave that owns the mel row (it walks
//     its frames in order; a wave's LDS operations execute in issue order), so there are no atomics and the sum order
//     is program order: run-to-run deterministic.  (ds_add_f32 was measured at ~0.6 us per wave-instruction on
//     MI355X -- 1300+ cycles, lanes serialised -- which made the first version 3-9x SLOWER than conv_tds.hip:
//     profiles/r02_run1_conv_rs_ds_add.log.)  Two accumulator registers can target the same output address from
//     different lanes (rows that differ by a multiple of J), so the 16 registers of a tile are coloured at compile
//     time into rounds of mutually disjoint registers: read round, add, write round, next round.
// The price of the overlap-add is a halo of (G-1)*J Y rows per time block (they only feed outputs of the
// neighbouring block): blocks are cut so that blockLen + halo is a whole number of 32-row tiles.
//
// backward-data is the same kernel on dy with tap-flipped, transposed weights (conv_tds.hip keeps the strided C2
// sub-sampling layers and every other geometry).
#include <cstdlib>
#include <type_traits>

#include "gemm.hpp"

namespace w2l { typedef __attribute__((address_space(3))) void* lptr_t; }

namespace w2l {

constexpr int kRsMaxTb = 16;

struct TdsRsP {
  const float* x;     // [B][Tin][H][C]
  const float* w;     // [kw][C][C] (forward orientation [tap][ci][co])
  const float* bias;  // [C] or null
  const float* add;   // optional addend with the layout of y, or null
  float* y;           // [B][Tout][H][C]
  int B, Tin, Tout, H, kw, padl;
  int relu, accum, flip;
  int nTb, hBlocks;
  int tStart[kRsMaxTb];  // first output frame of time block i
  int kt[kRsMaxTb];      // 32-row tiles of time block i (block length = 32*kt - halo, clipped to Tout)
  int abl;               // probe build only (W2L_TDS_RS_ABL): timing ablations, results are garbage
  int stagger;           // start offset between the co-resident workgroups of a CU, in s_sleep(127) units (~3.4 us)
};

template <int C, int G, int J, int KTMAX>
struct RsCfg {
  static constexpr int HH = 4;                          // mel rows per workgroup = waves
  static constexpr int NCT = (G * C + 31) / 32;         // column tiles
  static constexpr int NK = J * C / 2;                  // MFMA steps per column tile
  static constexpr int HALO = (G - 1) * J;
  static constexpr int NFMAX = 32 * KTMAX + J - 1;      // slab frames
  static constexpr int FT = NFMAX | 1;                  // slab row stride (floats), odd
  static constexpr int OT = (32 * KTMAX + HALO) | 1;    // out row stride
  static constexpr int ROWS = HH * C;
  static constexpr int Q = ROWS / 4;                    // float4 pieces per frame
  // staging / epilogue thread mapping: thread = (frame chunk f0 = tid / Q, piece q = tid % Q); its pieces are frames
  // f0, f0 + FSTEP, f0 + 2 FSTEP ... of piece q -- the (frame, piece) split costs one division per KERNEL, the bias of a
  // thread's four channels sits in registers, every LDS / global offset is base + constant * v
  static constexpr int FSTEP = 256 / Q;
  static constexpr int XV = (NFMAX + FSTEP - 1) / FSTEP;   // pieces per thread
  static constexpr int EV = (32 * KTMAX - HALO + FSTEP - 1) / FSTEP;
  static constexpr size_t LDS = (size_t)(ROWS * FT + (ROWS + 1) * OT) * sizeof(float);
  static constexpr int WGS = (3 * LDS <= 160 * 1024) ? 3 : (2 * LDS <= 160 * 1024 ? 2 : 1);   // workgroups per CU
  static_assert(C % 2 == 0 && ROWS % 4 == 0, "channel count");
};

// Overlap-add rounds.  Register q = 4i + jj of a 32x32 accumulator holds rows 8i + jj (lanes 0-31) and 8i + jj + 4
// (lanes 32-63).  Lanes of different tap groups add rows that differ by d*J (0 < d < G) into the same output frame,
// so two registers conflict when any of their rows differ by such a multiple; greedy colouring at compile time.
template <int G, int J>
struct RsRounds {
  static_assert(!(4 % J == 0 && 4 / J < G), "rows 4 apart (the two lane halves of one register) must not collide");
  static constexpr bool conflict(int a, int b) {
    const int ra = 8 * (a / 4) + (a % 4), rb = 8 * (b / 4) + (b % 4);
    for (int ha = 0; ha < 2; ++ha)
      for (int hb = 0; hb < 2; ++hb) {
        int d = (ra + 4 * ha) - (rb + 4 * hb);
        if (d < 0) d = -d;
        if (d != 0 && d % J == 0 && d / J < G) return true;
      }
    return false;
  }
  struct Table { int color[16]; int n; };
  static constexpr Table make() {
    Table t{};
    t.n = 0;
    for (int q = 0; q < 16; ++q) {
      int c = 0;
      for (;; ++c) {
        bool ok = true;
        for (int r = 0; r < q; ++r)
          if (t.color[r] == c && conflict(q, r)) ok = false;
        if (ok) break;
      }
      t.color[q] = c;
      if (c + 1 > t.n) t.n = c + 1;
    }
    return t;
  }
};

// the same colouring over register PAIRS (2m, 2m+1): usable when the two rows of a pair (and their +4 lane halves) can
// never meet at one output
template <int G, int J>
struct RsPairRounds {
  static constexpr bool pconflict(int m, int n) {
    for (int a = 2 * m; a < 2 * m + 2; ++a)
      for (int b = 2 * n; b < 2 * n + 2; ++b)
        if (RsRounds<G, J>::conflict(a, b)) return true;
    return false;
  }
  static constexpr bool ok() {
    for (int m = 0; m < 8; ++m)
      if (RsRounds<G, J>::conflict(2 * m, 2 * m + 1)) return false;
    return true;
  }
  struct Table { int color[8]; int n; };
  static constexpr Table make() {
    Table t{};
    t.n = 0;
    for (int m = 0; m < 8; ++m) {
      int c = 0;
      for (;; ++c) {
        bool free = true;
        for (int r = 0; r < m; ++r)
          if (t.color[r] == c && pconflict(m, r)) free = false;
        if (free) break;
      }
      t.color[m] = c;
      if (c + 1 > t.n) t.n = c + 1;
    }
    return t;
  }
};

}  // namespace w2l
#include "conv_tds_rs3.hpp"
namespace w2l {

template <int C, int G, int J, int KTMAX>
__global__ __launch_bounds__(256, (RsCfg<C, G, J, KTMAX>::WGS)) void tds_conv_rs_k(TdsRsP p, int nTiles) {
  using Cfg = RsCfg<C, G, J, KTMAX>;
  constexpr int NCT = Cfg::NCT, NK = Cfg::NK, HALO = Cfg::HALO, FT = Cfg::FT, OT = Cfg::OT, ROWS = Cfg::ROWS, Q = Cfg::Q,
                XV = Cfg::XV, FSTEP = Cfg::FSTEP, EV = Cfg::EV;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* slab = lds;               // [ROWS][FT]   slab[(hh*C + ci)][frame]
  float* outA = lds + ROWS * FT;   // [ROWS + 1][OT]  out[(hh*C + co)][HALO + u]; row ROWS absorbs the padding columns
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hf = lane >> 5;

  // ---- once per workgroup: the weights of every column tile, in MFMA B-operand order, into registers
  // step s = (j, pp): the two k of the MFMA are ci = 2pp (lanes 0-31) and 2pp + 1 (lanes 32-63); column r of
  // tile ct is (g, co) = divmod(32 ct + r, C)
  float bw[NCT][NK];
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) {
    const int n = 32 * ct + r, g = n / C, co = n - g * C;
#pragma unroll
    for (int s = 0; s < NK; ++s) {
      const int j = s / (C / 2), ci = 2 * (s % (C / 2)) + hf;
      const int tap = g * J + j;
      const bool ok = g < G && tap < p.kw;
      // forward: W[tap][ci][co]; backward-data: dx[., n] = sum dy[., k] W[kw-1-tap][n][k]
      const size_t src = !p.flip ? ((size_t)tap * C + ci) * C + co : ((size_t)(p.kw - 1 - tap) * C + co) * C + ci;
      const float t = p.w[ok ? src : 0];
      bw[ct][s] = ok ? t : 0.f;
    }
  }
  int ob[NCT];  // overlap-add base of this lane's column: out row (wave, co), shifted back by the group's g*J frames
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) {
    const int n = 32 * ct + r, g = n / C, co = n - g * C;
    ob[ct] = (g < G ? (wave * C + co) * OT + HALO - g * J : ROWS * OT) + 4 * hf;
  }
  const float* ab = slab + (wave * C + hf) * FT + r;
  const int HC = p.H * C;
  const int f0 = tid / Q, q4 = 4 * (tid - f0 * Q);
  const bool act = f0 < FSTEP;
  float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);   // the four channels (4q + k) mod C of this thread's pieces
  if (p.bias) bias4 = make_float4(p.bias[q4 % C], p.bias[(q4 + 1) % C], p.bias[(q4 + 2) % C], p.bias[(q4 + 3) % C]);
  for (int e = tid; e < (ROWS + 1) * OT; e += 256) outA[e] = 0.f;   // every tile leaves it zeroed again

  float4 xr[XV];
  auto fetch = [&](int tile) {
    const int hb = tile % p.hBlocks, tb = (tile / p.hBlocks) % p.nTb, b = tile / (p.hBlocks * p.nTb);
    const int tIn0 = p.tStart[tb] - p.padl, nf = 32 * p.kt[tb] + J - 1;
    const float* xb = p.x + ((size_t)b * p.Tin * p.H + hb * 4) * C + q4;
#pragma unroll
    for (int v = 0; v < XV; ++v) {
      const int f = f0 + FSTEP * v;
      const int ti = tIn0 + f;
      const bool ok = act && f < nf && ti >= 0 && ti < p.Tin;
      const float4 t4 = *(const float4*)(xb + (size_t)(ok ? ti : 0) * HC);
      xr[v] = ok ? t4 : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };

  int tile = blockIdx.x;
  if (tile < nTiles) fetch(tile);
  // The co-resident workgroups of a CU start together and, doing identical work, stay in lockstep: their MFMA phases
  // collide on the matrix pipes and their staging / epilogue phases collide on the LDS, nothing overlaps (ablation:
  // time = MFMA time + everything else, profiles/r02_run8_conv_rs_diet_ablation.log).  Break the symmetry once: the k-th
  // workgroup dispatched to a CU (block index / 256 in breadth-first dispatch; only speed depends on that guess) starts
  // k * stagger later, so one workgroup's MFMA phase runs beside another's memory phase from then on.
  for (int i = 0; i < p.stagger * (int)((blockIdx.x >> 8) % Cfg::WGS); ++i) __builtin_amdgcn_s_sleep(127);
  for (; tile < nTiles; tile += gridDim.x) {
    const int hb = tile % p.hBlocks, tb = (tile / p.hBlocks) % p.nTb, b = tile / (p.hBlocks * p.nTb);
    const int t0 = p.tStart[tb], kt = p.kt[tb];
    const int nf = 32 * kt + J - 1;
#ifdef W2L_PROBE
    const int abl = p.abl;   // 1: no MFMAs, 2: no overlap-add, 4: no epilogue, 8: no staging / zero fill, 16: no prefetch
#else
    constexpr int abl = 0;
#endif
    __syncthreads();  // the previous tile's epilogue has read (and re-zeroed) outA; its fragment reads of the slab are long done
    // slab <- prefetched pieces, transposed to time-fastest
    {
      float* d = slab + q4 * FT + f0;
#pragma unroll
      for (int v = 0; v < XV; ++v) {
        if (abl & 8) break;
        if (act && f0 + FSTEP * v < nf) {
          d[FSTEP * v] = xr[v].x; d[FT + FSTEP * v] = xr[v].y; d[2 * FT + FSTEP * v] = xr[v].z; d[3 * FT + FSTEP * v] = xr[v].w;
        }
      }
    }
    __syncthreads();
    {
      const int nxt = tile + gridDim.x;
      if (!(abl & 16)) fetch(nxt < nTiles ? nxt : tile);  // in flight behind this tile's MFMAs
    }

    // ---- MFMA phase.  (A software-pipelined variant -- the overlap-add of unit u-1 issued piecewise under the MFMAs of
    // unit u with a second accumulator set, A fragments re-loaded right after their last use, sched_barrier after every
    // MFMA slot -- was built and measured: no gain at C = 10 (123 vs 117 us), and at C = 18 hipcc no longer unrolled the
    // column-tile loop, indexed the weight registers dynamically and fell to 308 us: profiles/r02_run9_conv_rs_swpipe_negative.log.)
    for (int kti = 0; kti < kt; ++kti) {
      const float* at = ab + 32 * kti;
      float a[NK];
#pragma unroll
      for (int s = 0; s < NK; ++s) a[s] = at[(2 * (s % (C / 2))) * FT + s / (C / 2)];
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {
        f32x16 acc;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
        for (int s = 0; s < NK; ++s) {
          if ((abl & 1) && s > 0) break;
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], bw[ct][s], acc, 0, 0, 0);
        }
        if (abl & 2) { if (acc[0] == 123.456f) outA[0] = acc[3]; continue; }
        // D: column = lane & 31, rows 8i + 4 (lane >> 5) + jj  ->  out[(wave, co)][tau - g*J]
        float* o = outA + ob[ct] + 32 * kti;
        constexpr auto rounds = RsRounds<G, J>::make();
#pragma unroll
        for (int c = 0; c < rounds.n; ++c) {
          // the reads of this round must be ISSUED after the writes of the previous one (other lanes' addresses):
          // a compiler-level fence; the LDS itself executes a wave's operations in order
          asm volatile("" ::: "memory");
          float old[16];
#pragma unroll
          for (int q = 0; q < 16; ++q)
            if (rounds.color[q] == c) old[q] = o[8 * (q / 4) + (q % 4)];
#pragma unroll
          for (int q = 0; q < 16; ++q)
            if (rounds.color[q] == c) o[8 * (q / 4) + (q % 4)] = old[q] + acc[q];
        }
        asm volatile("" ::: "memory");
      }
    }
    __syncthreads();

    // ---- epilogue: out tile -> y[b][t0 + u][4 hb + hh][c], whole (4 C)-float frames as float4; every column read is
    // written back as zero, so the next tile finds the overlap-add buffer clear (no separate zero-fill pass)
    int tc = p.Tout - t0;
    if (tc > 32 * kt - HALO) tc = 32 * kt - HALO;
    if (abl & 4) tc = 0;
    const int ue = 32 * kt - HALO;   // columns [HALO, 32 kt) hold outputs (the last time block stores only tc of them)
    const size_t gBase = (((size_t)b * p.Tout + t0) * p.H + hb * 4) * C + q4;
    float* s4 = outA + q4 * OT + HALO + f0;
#pragma unroll
    for (int it = 0; it < EV; ++it) {
      const int u = f0 + FSTEP * it;
      if (!act || u >= ue) break;
      float4 v = make_float4(s4[FSTEP * it], s4[OT + FSTEP * it], s4[2 * OT + FSTEP * it], s4[3 * OT + FSTEP * it]);
      s4[FSTEP * it] = 0.f; s4[OT + FSTEP * it] = 0.f; s4[2 * OT + FSTEP * it] = 0.f; s4[3 * OT + FSTEP * it] = 0.f;
      if (u < tc) {
        v.x += bias4.x; v.y += bias4.y; v.z += bias4.z; v.w += bias4.w;
        if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        const size_t gq = gBase + (size_t)u * HC;
        if (p.add) { const float4 a4 = *(const float4*)(p.add + gq); v.x += a4.x; v.y += a4.y; v.z += a4.z; v.w += a4.w; }
        if (p.accum) { const float4 y0 = *(const float4*)(p.y + gq); v.x += y0.x; v.y += y0.y; v.z += y0.z; v.w += y0.w; }
        *(float4*)(p.y + gq) = v;
      }
    }
    // the halo columns [0, HALO) and [32 kt, 32 kt + HALO) took overlap-add partial sums that belong to the neighbouring
    // time blocks (which recompute them): clear
    for (int e = tid; e < ROWS * 2 * HALO; e += 256) {
      const int row = e / (2 * HALO), c = e - row * (2 * HALO);
      outA[row * OT + (c < HALO ? c : 32 * kt + c - HALO)] = 0.f;
    }
  }
}

#ifdef W2L_PROBE  // kept for A/B work only: measured SLOWER than the cooperative kernel above
// ------------------------------------------------------------------------------------------------ v2: autonomous waves
// Measured on the cooperative kernel above (profiles/r02_run3_conv_rs_rmw_ablation.log, C = 10): 142 us, of which the
// MFMAs alone 75 -- staging the slab through registers with a transpose (31 us), the zero fill, the epilogue (26 us) and
// three workgroup barriers per tile all ran IN SERIES with the matrix pipe, and two co-resident workgroups start in
// lockstep.  v2 removes every barrier: a workgroup is ONE wave that owns (utterance, mel row, time block) tiles:
//   * staging = LDS-DMA: `buffer_load_dword ... lds` writes lane l's dword to M0 + 4 l, so with lane <-> FRAME one
//     instruction drops 64 consecutive frames of one input channel straight into the time-fastest slab row -- the
//     transpose costs nothing, no VGPRs, no ds_write; frames before / after the utterance are out of the buffer
//     resource's range and arrive as zeros (the padding);
//   * the next tile's DMA is issued right after the MFMA phase has consumed the slab, under the epilogue;
//   * 8 such waves per CU in different phases keep the matrix pipe fed while others stage or store.
// A launch block is FOUR such waves on the four adjacent mel rows of one (utterance, time block): they never
// synchronise, but they run on one CU at about the same pace, so the 128-byte lines a 40-byte (C floats) piece of a
// frame sits in are fetched into that CU's L1 once and serve all four (with one-wave blocks, adjacent rows went to
// different XCDs and every XCD's L2 pulled every line: 190 us instead of 142, profiles/r02_run6_conv_rs2_first.log).
template <int C, int G, int J, int KTMAX>
struct Rs2Cfg {
  static constexpr int NCT = (G * C + 31) / 32;
  static constexpr int NK = J * C / 2;
  static constexpr int HALO = (G - 1) * J;
  static constexpr int NFMAX = 32 * KTMAX + J - 1;
  static constexpr int FT = NFMAX | 1;
  static constexpr int OT = (32 * KTMAX + HALO) | 1;
  static constexpr int LDSF = C * FT + (C + 1) * OT + 32;   // slab | out (+ the row that absorbs padding columns) | bias
  static constexpr size_t LDS = (size_t)LDSF * sizeof(float);
};

template <int C, int G, int J, int KTMAX>
__global__ __launch_bounds__(256) void tds_conv_rs2_k(TdsRsP p, int nTiles) {
  using Cfg = Rs2Cfg<C, G, J, KTMAX>;
  constexpr int NCT = Cfg::NCT, NK = Cfg::NK, HALO = Cfg::HALO, FT = Cfg::FT, OT = Cfg::OT;
  extern __shared__ __attribute__((aligned(16))) float ldsAll[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, hf = lane >> 5;
  float* lds = ldsAll + wave * Cfg::LDSF;   // this wave's private slab / out tile / bias
  float* slab = lds;                 // [C][FT]      slab[ci][frame]
  float* outA = lds + C * FT;        // [C + 1][OT]  out[co][HALO + u]
  float* biasS = outA + (C + 1) * OT;

  float bw[NCT][NK];
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) {
    const int n = 32 * ct + r, g = n / C, co = n - g * C;
#pragma unroll
    for (int s = 0; s < NK; ++s) {
      const int j = s / (C / 2), ci = 2 * (s % (C / 2)) + hf;
      const int tap = g * J + j;
      const bool ok = g < G && tap < p.kw;
      const size_t src = !p.flip ? ((size_t)tap * C + ci) * C + co : ((size_t)(p.kw - 1 - tap) * C + co) * C + ci;
      const float t = p.w[ok ? src : 0];
      bw[ct][s] = ok ? t : 0.f;
    }
  }
  int ob[NCT];
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) {
    const int n = 32 * ct + r, g = n / C, co = n - g * C;
    ob[ct] = (g < G ? co * OT + HALO - g * J : C * OT) + 4 * hf;
  }
  if (lane < C) biasS[lane] = p.bias ? p.bias[lane] : 0.f;
  for (int e = lane; e < (C + 1) * OT; e += 64) outA[e] = 0.f;
  const float* ab = slab + hf * FT + r;
  const int HC = p.H * C;
  const int hq = p.H >> 2, tilesPerB = p.nTb * hq;   // a block tile = (b, time block, 4 mel rows); this wave: row 4 hb + wave

  // LDS-DMA of one tile's slab: channel ci, frames [64 c, 64 c + 64) -> slab[ci][64 c + lane]
  auto stage = [&](int tile) {
    const int b = tile / tilesPerB, rem = tile - b * tilesPerB, tb = rem / hq, h = 4 * (rem - tb * hq) + wave;
    const int nf = 32 * p.kt[tb] + J - 1;
    const int tIn0 = p.tStart[tb] - p.padl;
    const float* base = p.x + ((size_t)b * p.Tin * p.H + h) * C;   // frame 0, mel row h of utterance b
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)(((size_t)(p.Tin - 1) * HC + C) * sizeof(float)), 0x00020000);
    const int nch = (nf + 63) >> 6;
    for (int c = 0; c < nch; ++c) {
      const int f = 64 * c + lane;
      const int voff = (tIn0 + f) * HC * (int)sizeof(float);        // negative / past the end: out of range -> zeros
      if (f < nf) {
#pragma unroll
        for (int ci = 0; ci < C; ++ci)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(slab + ci * FT + 64 * c), 4, voff, ci * (int)sizeof(float), 0, 0);
      }
    }
  };

#ifdef W2L_PROBE
  const int abl = p.abl;   // 1: no MFMAs, 2: no overlap-add, 4: no epilogue, 8: no staging, 32: no zero fill
#else
  constexpr int abl = 0;
#endif
  int tile = blockIdx.x;
  if (tile < nTiles && !(abl & 8)) stage(tile);
  for (; tile < nTiles; tile += gridDim.x) {
    const int b = tile / tilesPerB, rem = tile - b * tilesPerB, tb = rem / hq, h = 4 * (rem - tb * hq) + wave;
    const int t0 = p.tStart[tb], kt = p.kt[tb];
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this tile's slab has landed (and the previous tile's stores have left)

    for (int kti = 0; kti < kt; ++kti) {
      const float* at = ab + 32 * kti;
      float a[NK];
#pragma unroll
      for (int s = 0; s < NK; ++s) a[s] = at[(2 * (s % (C / 2))) * FT + s / (C / 2)];
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {
        f32x16 acc;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
        for (int s = 0; s < NK; ++s) {
          if ((abl & 1) && s > 0) break;
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], bw[ct][s], acc, 0, 0, 0);
        }
        if (abl & 2) { if (acc[0] == 123.456f) outA[0] = acc[3]; continue; }
        float* o = outA + ob[ct] + 32 * kti;
        constexpr auto rounds = RsRounds<G, J>::make();
#pragma unroll
        for (int c = 0; c < rounds.n; ++c) {
          asm volatile("" ::: "memory");
          float old[16];
#pragma unroll
          for (int q = 0; q < 16; ++q)
            if (rounds.color[q] == c) old[q] = o[8 * (q / 4) + (q % 4)];
#pragma unroll
          for (int q = 0; q < 16; ++q)
            if (rounds.color[q] == c) o[8 * (q / 4) + (q % 4)] = old[q] + acc[q];
        }
        asm volatile("" ::: "memory");
      }
    }
    // the slab is consumed (all fragment reads have returned: their values fed the MFMAs above): next tile's DMA now
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
    {
      const int nxt = tile + gridDim.x;
      if (nxt < nTiles && !(abl & 8)) stage(nxt);
    }
    // ---- epilogue: out[co][u] -> y[b][t0 + u][h][co]; the out tile is left zeroed for the next tile
    int tc = p.Tout - t0;
    if (tc > 32 * kt - HALO) tc = 32 * kt - HALO;
    if (abl & 4) tc = 0;
    const size_t gBase = (((size_t)b * p.Tout + t0) * p.H + h) * C;
    const int total = tc * C;
    for (int e0 = 0; e0 < total; e0 += 256) {
      float v[4], ad[4];
      int gi[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int e = e0 + 64 * k + lane;
        const int u = e / C, co = e - u * C;
        gi[k] = e < total ? u * HC + co : -1;
        v[k] = e < total ? outA[co * OT + HALO + u] + biasS[co] : 0.f;
        ad[k] = 0.f;
        if (gi[k] >= 0) {
          if (p.add) ad[k] = p.add[gBase + gi[k]];
          if (p.accum) ad[k] += p.y[gBase + gi[k]];
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (gi[k] < 0) continue;
        float t = v[k];
        if (p.relu) t = fmaxf(t, 0.f);
        p.y[gBase + gi[k]] = t + ad[k];
      }
    }
    if (!(abl & 32)) {
      const int no = 32 * kt + HALO;
      for (int e = lane; e < C * no; e += 64) {
        const int row = e / no, u = e - row * no;
        outA[row * OT + u] = 0.f;
      }
    }
  }
}

template <int C, int G, int J, int KTMAX>
static int rs2_launch(TdsRsP p, hipStream_t s) {
  using Cfg = Rs2Cfg<C, G, J, KTMAX>;
  const int nTiles = p.B * p.nTb * (p.H / 4);
  const size_t shmem = 4 * Cfg::LDS;
  int bpc = (int)(160 * 1024 / shmem);      // blocks (of 4 waves) per CU
  if (bpc > 3) bpc = 3;
  if (bpc < 1) return W2L_EUNSUPPORTED;
  const int blocks = nTiles < 256 * bpc ? nTiles : 256 * bpc;
  static bool attr[64] = {};
  if (shmem > 64 * 1024 && first_on_device(attr)) {
    W2L_HIP_CHECK(hipFuncSetAttribute((const void*)tds_conv_rs2_k<C, G, J, KTMAX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  }
  hipLaunchKernelGGL((tds_conv_rs2_k<C, G, J, KTMAX>), dim3((unsigned)blocks), dim3(256), shmem, s, p, nTiles);
  return W2L_OK;
}

#endif  // W2L_PROBE

// time blocks: n blocks whose lengths + halo are whole 32-row tiles, at most KTMAX tiles each, as even as possible
static bool rs_plan(int Tout, int halo, int ktMax, TdsRsP& p) {
  for (int n = 1; n <= kRsMaxTb; ++n) {
    const int total = (Tout + n * halo + 31) / 32;
    if ((total + n - 1) / n > ktMax) continue;
    int t = 0;
    for (int i = 0; i < n; ++i) {
      const int k = total / n + (i < total % n ? 1 : 0);
      p.tStart[i] = t;
      p.kt[i] = k;
      t += 32 * k - halo;
    }
    p.nTb = n;
    return t >= Tout && p.tStart[n - 1] < Tout;
  }
  return false;
}

template <int C, int G, int J, int KTMAX>
static int rs_launch(TdsRsP p, hipStream_t s) {
  using Cfg = RsCfg<C, G, J, KTMAX>;
  p.hBlocks = p.H / Cfg::HH;
  const int nTiles = p.B * p.nTb * p.hBlocks;
  const int perCu = Cfg::WGS;
  const int blocks = nTiles < 256 * perCu ? nTiles : 256 * perCu;
  static bool attr[64] = {};
  if (Cfg::LDS > 64 * 1024 && first_on_device(attr)) {
    W2L_HIP_CHECK(hipFuncSetAttribute((const void*)tds_conv_rs_k<C, G, J, KTMAX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS));
  }
  hipLaunchKernelGGL((tds_conv_rs_k<C, G, J, KTMAX>), dim3((unsigned)blocks), dim3(256), Cfg::LDS, s, p, nTiles);
  return W2L_OK;
}

template <int C, int G, int J, int HH, int CS, int KT>
static int rs3_launch(const TdsRsP& q, hipStream_t s) {
  using Cfg = Rs3Cfg<C, G, J, HH, CS, KT>;
  if (q.H % HH) return W2L_EUNSUPPORTED;
  TdsRs3P p{};
  p.x = q.x; p.w = q.w; p.bias = q.bias; p.add = q.add; p.y = q.y;
  p.B = q.B; p.Tin = q.Tin; p.Tout = q.Tout; p.H = q.H; p.kw = q.kw; p.padl = q.padl; p.relu = q.relu; p.accum = q.accum; p.flip = q.flip;
  p.abl = q.abl;
  { const char* e = tune_env("W2L_TDS_RS3_DBG"); p.dbg = e ? (long long*)strtoull(e, nullptr, 10) : nullptr; }
  p.hBlocks = q.H / HH;
  const long long total = (long long)q.B * p.hBlocks * q.Tout;
  if (total <= 0 || total > (1ll << 30)) return W2L_EUNSUPPORTED;
  if ((long long)q.Tin * q.H * C * 4 >= (1ll << 31) || (long long)q.Tout * q.H * C * 4 >= (1ll << 31)) return W2L_EUNSUPPORTED;   // one utterance per buffer resource
  // one workgroup per CU; short inputs: at least two full tiles of outputs per workgroup
  long long wgs = total / (2 * Cfg::L);
  if (wgs < 1) wgs = 1;
  if (wgs > 256) wgs = 256;
  p.quota = (int)((total + wgs - 1) / wgs);
  const int blocks = (int)((total + p.quota - 1) / p.quota);
#ifdef W2L_PROBE
  if (p.abl) {
    bool done = false;
    auto go = [&](auto tag) {
      constexpr int M = decltype(tag)::value;
      if (p.abl != M || done || p.add) return;
      (void)hipFuncSetAttribute((const void*)tds_conv_rs3_k<C, G, J, HH, CS, KT, false, M>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS);
      hipLaunchKernelGGL((tds_conv_rs3_k<C, G, J, HH, CS, KT, false, M>), dim3((unsigned)blocks), dim3(768), Cfg::LDS, s, p);
      done = true;
    };
    go(std::integral_constant<int, 2>{}); go(std::integral_constant<int, 32>{}); go(std::integral_constant<int, 34>{});
    go(std::integral_constant<int, 92>{}); go(std::integral_constant<int, 94>{}); go(std::integral_constant<int, 126>{});
    go(std::integral_constant<int, 35>{}); go(std::integral_constant<int, 28>{});
    if (done) return W2L_OK;
  }
#endif
  static bool attr[64] = {};
  if (first_on_device(attr)) {
    W2L_HIP_CHECK(hipFuncSetAttribute((const void*)tds_conv_rs3_k<C, G, J, HH, CS, KT, false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS));
    W2L_HIP_CHECK(hipFuncSetAttribute((const void*)tds_conv_rs3_k<C, G, J, HH, CS, KT, true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS));
  }
  if (p.add) hipLaunchKernelGGL((tds_conv_rs3_k<C, G, J, HH, CS, KT, true, 0>), dim3((unsigned)blocks), dim3(768), Cfg::LDS, s, p);
  else hipLaunchKernelGGL((tds_conv_rs3_k<C, G, J, HH, CS, KT, false, 0>), dim3((unsigned)blocks), dim3(768), Cfg::LDS, s, p);
  return W2L_OK;
}


// ================================================================================================ block-Toeplitz generation
}  // namespace w2l
#include "conv_tds_tz.hpp"
namespace w2l {

template <int CI, int CO, int R, int NCT, int SIG, int KWM, int ST, bool FWD>
static int tz_launch(TdsTzP p, int abl, hipStream_t s) {
  using Cfg = TzCfg<CI, CO, R, NCT, SIG, KWM, ST>;
  { const char* e = tune_env("W2L_TDS_TZ_DBG"); p.dbg = e ? (long long*)strtoull(e, nullptr, 10) : nullptr; }
  p.hBlocks = p.H / Cfg::HB;
  p.rps = (p.Tout + Cfg::RF - 1) / Cfg::RF;
  const long long rounds = (long long)p.B * p.hBlocks * p.rps;
  if (rounds <= 0 || rounds > (1ll << 30)) return W2L_EUNSUPPORTED;
  p.nRounds = (int)rounds;
  // two workgroups per CU; equal contiguous shares of the round axis
  // 512 = two resident workgroups per CU; where the rounds do not divide by 512 but do by 768 (C = 14: 3840 rounds = 7.5 per
  // workgroup at 512, 5 at 768) the finer cut wins: 79.7 / 76.3 us against 82.6 / 79.2 (profiles/r05_run16_conv_tz.log)
  int wgMax = (p.nRounds % 512 != 0 && p.nRounds % 768 == 0) ? 768 : 512;
  { const char* e = tune_env("W2L_TDS_TZ_WGS"); if (e && atoi(e) > 0) wgMax = atoi(e); }
  const int wgs = p.nRounds < wgMax ? p.nRounds : wgMax;
  p.rpw = (p.nRounds + wgs - 1) / wgs;
  const int blocks = (p.nRounds + p.rpw - 1) / p.rpw;
  constexpr bool DEFER = true;       // (C = 18 has the registers for the second accumulator set since its waves walk their own 22-frame windows)
  constexpr int M0 = FWD ? 0 : 2;    // the two modes of the direction: plain / + ReLU, plain / + addend
#ifdef W2L_PROBE
  if (abl && FWD) {
    bool done = false;
    auto go = [&](auto tag) {
      constexpr int M = decltype(tag)::value;
      if (abl != M || done || !p.relu) return;
      (void)hipFuncSetAttribute((const void*)tds_conv_tz_k<CI, CO, R, NCT, SIG, KWM, ST, 1, DEFER, M>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS);
      hipLaunchKernelGGL((tds_conv_tz_k<CI, CO, R, NCT, SIG, KWM, ST, 1, DEFER, M>), dim3((unsigned)blocks), dim3(256), Cfg::LDS, s, p);
      done = true;
    };
    go(std::integral_constant<int, 1>{}); go(std::integral_constant<int, 2>{}); go(std::integral_constant<int, 4>{});
    go(std::integral_constant<int, 8>{}); go(std::integral_constant<int, 14>{}); go(std::integral_constant<int, 12>{});
    if (done) return W2L_OK;
  }
#endif
  static bool attr[64] = {};
  if (first_on_device(attr)) {
    W2L_HIP_CHECK(hipFuncSetAttribute((const void*)tds_conv_tz_k<CI, CO, R, NCT, SIG, KWM, ST, M0, DEFER, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS));
    W2L_HIP_CHECK(hipFuncSetAttribute((const void*)tds_conv_tz_k<CI, CO, R, NCT, SIG, KWM, ST, M0 + 1, DEFER, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS));
  }
  const bool second = FWD ? p.relu != 0 : p.add != nullptr;
  if (second) hipLaunchKernelGGL((tds_conv_tz_k<CI, CO, R, NCT, SIG, KWM, ST, M0 + 1, DEFER, 0>), dim3((unsigned)blocks), dim3(256), Cfg::LDS, s, p);
  else hipLaunchKernelGGL((tds_conv_tz_k<CI, CO, R, NCT, SIG, KWM, ST, M0, DEFER, 0>), dim3((unsigned)blocks), dim3(256), Cfg::LDS, s, p);
  return W2L_OK;
}

// true + *status when the block-Toeplitz generation (conv_tds_tz.hpp) runs this convolution: the TDS convolutions proper
// (C -> C, stride 1, both directions), the strided sub-sampling layers between the stages (10 -> 14, 14 -> 18, stride 2)
// forward, and the phases of their backward-data pass (every second tap, every second output frame)
bool tds_tz_try(const float* x, const float* w, const float* bias, const float* add, float* y, int B, int Tin, int Tout, int H,
                int Cin, int Cout, int kw, int stride, int padl, int relu, int accum, int flip, int tapOff, int tapStep, int oOff,
                int oStep, int ToutFull, int profKind, hipStream_t s, int* status) {
  if (tune_env("W2L_TDS_TZ_OFF") || H % 16 || accum || kw < 1) return false;
  if (flip ? (bias || relu || stride != 1) : (add || tapOff || tapStep != 1 || oOff || oStep != 1)) return false;
  if ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)add | (uintptr_t)w) & 15) != 0) return false;
  if ((long long)Tin * H * Cin * 4 >= (1ll << 31) || (long long)ToutFull * H * Cout * 4 >= (1ll << 31)) return false;   // one utterance per buffer resource
  TdsTzP p{};
  p.x = x; p.w = w; p.bias = bias; p.add = add; p.y = y;
  p.B = B; p.Tin = Tin; p.Tout = Tout; p.H = H; p.kw = kw; p.padl = padl; p.relu = relu; p.flip = flip;
  p.kwFull = flip ? tapOff + tapStep * (kw - 1) + 1 : kw;
  p.tapOff = tapOff; p.oOff = oOff; p.oStep = oStep; p.ToutFull = ToutFull;
  int abl = 0;
  { const char* e = tune_env("W2L_TDS_RS_ABL"); abl = e ? atoi(e) : 0; }
  int st = W2L_EUNSUPPORTED;
  const bool same = Cin == Cout && stride == 1 && tapStep == 1 && kw <= 21;
  const bool sub = !flip && stride == 2 && kw <= 21 && !tune_env("W2L_TDS_TZ_C2_OFF");
  const bool phase = flip && tapStep == 2 && kw <= 11 && !tune_env("W2L_TDS_TZ_C2_OFF");
  if (!(same && (Cin == 10 || Cin == 14 || Cin == 18)) && !(sub && ((Cin == 10 && Cout == 14) || (Cin == 14 && Cout == 18))) &&
      !(phase && ((Cin == 14 && Cout == 10) || (Cin == 18 && Cout == 14))))
    return false;
  prof_begin(s, 2.0 * B * Tout * (double)H * kw * Cin * Cout, profKind);
  if (same) {
    if (!flip) st = Cin == 10 ? tz_launch<10, 10, 3, 1, 1, 21, 1, true>(p, abl, s) : Cin == 14 ? tz_launch<14, 14, 2, 1, 1, 21, 1, true>(p, abl, s)
                                                                                                  : tz_launch<18, 18, 3, 2, 1, 21, 1, true>(p, abl, s);
    else st = Cin == 10 ? tz_launch<10, 10, 3, 1, 1, 21, 1, false>(p, abl, s) : Cin == 14 ? tz_launch<14, 14, 2, 1, 1, 21, 1, false>(p, abl, s)
                                                                                              : tz_launch<18, 18, 3, 2, 1, 21, 1, false>(p, abl, s);
  } else if (sub) {
    st = Cin == 10 ? tz_launch<10, 14, 2, 1, 2, 21, 1, true>(p, 0, s) : tz_launch<14, 18, 3, 2, 2, 21, 1, true>(p, 0, s);
  } else {
    st = Cin == 14 ? tz_launch<14, 10, 3, 1, 1, 11, 2, false>(p, 0, s) : tz_launch<18, 14, 2, 1, 1, 11, 2, false>(p, 0, s);
  }
  prof_end(s);
  if (st == W2L_EUNSUPPORTED) return false;   // tz_launch refused (round count out of range): the older generations take it
  if (st == W2L_OK && hipGetLastError() != hipSuccess) st = W2L_EHIP;
  *status = st;
  return true;
}

// true + *status when this geometry runs on the role-swapped kernel
bool tds_rs_try(const float* x, const float* w, const float* bias, const float* add, float* y, int B, int Tin, int Tout, int H,
                int C, int kw, int padl, int relu, int accum, int flip, int profKind, hipStream_t s, int* status) {
  if (tune_env("W2L_TDS_RS_OFF")) return false;
  if (!(C == 10 || C == 14 || C == 18) || kw > 21 || kw < 1 || H % 4) return false;
  if ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)add) & 15) != 0) return false;
  TdsRsP p{};
  p.x = x; p.w = w; p.bias = bias; p.add = add; p.y = y;
  p.B = B; p.Tin = Tin; p.Tout = Tout; p.H = H; p.kw = kw; p.padl = padl; p.relu = relu; p.accum = accum; p.flip = flip;
  { const char* e = tune_env("W2L_TDS_RS_ABL"); p.abl = e ? atoi(e) : 0; }
  { const char* e = tune_env("W2L_TDS_RS_STAGGER"); p.stagger = e ? atoi(e) : (C == 10 ? 4 : 0); }
  // third generation (conv_tds_rs3.hpp): wave-specialised, streamed time axis.  One utterance per 2 GiB buffer resource.
  const bool rs3 = !tune_env("W2L_TDS_RS3_OFF") && H % 8 == 0 && !accum && (long long)B * (H / 4) * Tout <= (1ll << 30) &&
                   (long long)Tin * H * C * 4 < (1ll << 31) && (long long)Tout * H * C * 4 < (1ll << 31);
  if (rs3) {
    prof_begin(s, 2.0 * B * Tout * (double)H * kw * C * C, profKind);
    int st = C == 10 ? rs3_launch<10, 3, 7, 8, 1, 2>(p, s) : C == 14 ? rs3_launch<14, 2, 11, 8, 1, 1>(p, s) : rs3_launch<18, 7, 3, 4, 2, 2>(p, s);
    prof_end(s);
    if (st == W2L_OK && hipGetLastError() != hipSuccess) st = W2L_EHIP;
    *status = st;
    return true;
  }
  if (C == 14 && !tune_env("W2L_TDS_RS_C14")) return false;   // C = 14: conv_tds.hip's 16-wide tiles (87.5 % of the lanes) stay ahead
  const bool v2 = tune_env("W2L_TDS_RS_V2") != nullptr;       // probe build: the autonomous-wave LDS-DMA variant (measured slower)
  const bool planned = C == 10 ? rs_plan(Tout, RsCfg<10, 3, 7, 4>::HALO, v2 ? 6 : 4, p)
                     : C == 14 ? rs_plan(Tout, RsCfg<14, 2, 11, 5>::HALO, 5, p) : rs_plan(Tout, RsCfg<18, 7, 3, 4>::HALO, 4, p);
  if (!planned) return false;
  prof_begin(s, 2.0 * B * Tout * (double)H * kw * C * C, profKind);
  int st;
  if (!v2) {
    if (C == 10) st = rs_launch<10, 3, 7, 4>(p, s);
    else if (C == 14) st = rs_launch<14, 2, 11, 5>(p, s);
    else st = rs_launch<18, 7, 3, 4>(p, s);
  } else {
#ifdef W2L_PROBE
    if (C == 10) st = rs2_launch<10, 3, 7, 6>(p, s);
    else if (C == 14) st = rs2_launch<14, 2, 11, 5>(p, s);
    else st = rs2_launch<18, 7, 3, 4>(p, s);
#else
    st = W2L_EUNSUPPORTED;
#endif
  }
  prof_end(s);
  if (st == W2L_OK && hipGetLastError() != hipSuccess) st = W2L_EHIP;
  *status = st;
  return true;
}

// ================================================================================================ backward-filter
// dW[tap][ci][co] = sum_{b,t,h} X[t + tap - padl][h][ci] * dY[t][h][co],   dbias[co] = sum dY[t][h][co]
// is a GEMM with a huge K = (b, t, h) and a 21*C x C result: with the result on 16-wide tiles of 16x16x4 MFMAs
// (conv_tds.hip) C = 10 / 14 / 18 columns fill 62.5 / 87.5 / 56 % of the lanes and the loop was issue-bound at ~30-47 %.
// Role swap for the filter gradient: split tap = ga*GB + gb and shift BOTH operands along time,
//     D[(ga, ci)][(gb, co)] = sum_{t'} X[t' + ga*GB - padl][ci] * dY[t' - gb][co]          (t' = t + gb)
// rows (ga, ci) = GA*C (+ one all-ones row: its gb = 0 columns are the bias gradient), columns (gb, co) = GB*C, K = time:
//     C = 10: GA = 3, GB = 7 : 1 x 3 tiles of 32x32, 21 taps     68 % of the MFMA lanes carry a wanted product
//     C = 14: GA = 2, GB = 11: 1 x 5 tiles, 22 taps (21 wanted)  80 %
//     C = 18: GA = 7, GB = 3 : 4 x 2 tiles, 21 taps              83 %
// Both operands are read straight out of time-fastest slabs (x[(h, ci)][frame], dy[(h, co)][frame]): a fragment read is
// 32 consecutive floats of one row, the K step is an immediate offset, NRT + NCT ds_read_b32 feed NRT*NCT 64-cycle MFMAs.
// A wave owns one mel row of the workgroup's tile and keeps its NRT x NCT accumulators in registers over ALL its tiles;
// at the end the four waves of a workgroup are added in wave order through LDS and one partial per workgroup goes to
// the stream scratch; tds_rsf_reduce_k adds the partials in workgroup order (deterministic) and scatters into dW / dbias.
struct TdsRsfP {
  const float* x;   // [B][Tin][H][C]
  const float* dy;  // [B][Tout][H][C]
  int B, Tin, Tout, H, kw, padl;
  int nStrips, hBlocks;
};

template <int C, int GA, int GB, int TS>
struct RsfCfg {
  static constexpr int HH = 4;
  static constexpr int ROWS = HH * C;
  static constexpr int Q = ROWS / 4;
  static constexpr int NRT = (GA * C + 1 + 31) / 32;
  static constexpr int NCT = (GB * C + 31) / 32;
  static constexpr int XF = TS + (GA - 1) * GB;   // x slab frames
  static constexpr int DF = TS + GB - 1;          // dy slab frames
  static constexpr int FT = XF | 1;
  static constexpr int DT = DF | 1;
  static constexpr int ZT = (TS + 2) | 1;         // zero / ones rows
  static constexpr int XOFF = 0, DOFF = ROWS * FT, ZOFF = DOFF + ROWS * DT, OOFF = ZOFF + ZT, END = OOFF + ZT;
  static constexpr int ACCF = NRT * NCT * 16 * 64;                      // floats of one wave's accumulators
  static constexpr int LDSF = END > ACCF ? END : ACCF;
  static constexpr size_t LDS = (size_t)LDSF * sizeof(float);
  static constexpr int XV = (XF * Q + 255) / 256, DV = (DF * Q + 255) / 256;
  static_assert(TS % 2 == 0 && C % 2 == 0, "strip length / channel count");
};

template <int C, int GA, int GB, int TS>
__global__ __launch_bounds__(256, (160 * 1024 / RsfCfg<C, GA, GB, TS>::LDS) >= 3 ? 3 : 2) void tds_conv_rsf_k(TdsRsfP p, float* __restrict__ partial, int nTiles) {
  using Cfg = RsfCfg<C, GA, GB, TS>;
  constexpr int NRT = Cfg::NRT, NCT = Cfg::NCT, FT = Cfg::FT, DT = Cfg::DT, Q = Cfg::Q, XF = Cfg::XF, DF = Cfg::DF, XV = Cfg::XV,
                DV = Cfg::DV;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hf = lane >> 5;

  int ab[NRT], bb[NCT];
#pragma unroll
  for (int rt = 0; rt < NRT; ++rt) {
    const int m = 32 * rt + r, ga = m / C, ci = m - ga * C;
    ab[rt] = (m < GA * C ? Cfg::XOFF + (wave * C + ci) * FT + ga * GB : m == GA * C ? Cfg::OOFF : Cfg::ZOFF) + hf;
  }
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) {
    const int n = 32 * ct + r, gb = n / C, co = n - gb * C;
    bb[ct] = (gb < GB ? Cfg::DOFF + (wave * C + co) * DT + (GB - 1 - gb) : Cfg::ZOFF) + hf;
  }
  for (int e = tid; e < Cfg::ZT; e += 256) { lds[Cfg::ZOFF + e] = 0.f; lds[Cfg::OOFF + e] = 1.f; }

  f32x16 acc[NRT][NCT];
#pragma unroll
  for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[rt][ct][q] = 0.f;

  const int HC = p.H * C;
  for (int tile = blockIdx.x; tile < nTiles; tile += gridDim.x) {
    const int hb = tile % p.hBlocks, st = (tile / p.hBlocks) % p.nStrips, b = tile / (p.hBlocks * p.nStrips);
    const int s0 = st * TS;                       // first t' of the strip
    __syncthreads();                              // the previous tile's fragment reads are done
    {  // x slab: frame f <-> input frame s0 - padl + f
      const float* xb = p.x + ((size_t)b * p.Tin * p.H + hb * 4) * C;
      float4 v[XV];
#pragma unroll
      for (int k = 0; k < XV; ++k) {
        const int e = tid + 256 * k, f = e / Q, q = e - f * Q;
        const int ti = s0 - p.padl + f;
        const bool ok = f < XF && ti >= 0 && ti < p.Tin;
        const float4 t4 = *(const float4*)(xb + (size_t)(ok ? ti : 0) * HC + 4 * q);
        v[k] = ok ? t4 : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int k = 0; k < XV; ++k) {
        const int e = tid + 256 * k, f = e / Q, q = e - f * Q;
        if (f < XF) {
          float* d = lds + Cfg::XOFF + (4 * q) * FT + f;
          d[0] = v[k].x; d[FT] = v[k].y; d[2 * FT] = v[k].z; d[3 * FT] = v[k].w;
        }
      }
    }
    {  // dy slab: frame f <-> output frame s0 - (GB - 1) + f
      const float* db = p.dy + ((size_t)b * p.Tout * p.H + hb * 4) * C;
      float4 v[DV];
#pragma unroll
      for (int k = 0; k < DV; ++k) {
        const int e = tid + 256 * k, f = e / Q, q = e - f * Q;
        const int ti = s0 - (GB - 1) + f;
        const bool ok = f < DF && ti >= 0 && ti < p.Tout;
        const float4 t4 = *(const float4*)(db + (size_t)(ok ? ti : 0) * HC + 4 * q);
        v[k] = ok ? t4 : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int k = 0; k < DV; ++k) {
        const int e = tid + 256 * k, f = e / Q, q = e - f * Q;
        if (f < DF) {
          float* d = lds + Cfg::DOFF + (4 * q) * DT + f;
          d[0] = v[k].x; d[DT] = v[k].y; d[2 * DT] = v[k].z; d[3 * DT] = v[k].w;
        }
      }
    }
    __syncthreads();
    // K loop over the strip: step s covers t' = s0 + 2s (lanes 0-31) and s0 + 2s + 1 (lanes 32-63)
#pragma unroll 8
    for (int s = 0; s < TS / 2; ++s) {
      float a[NRT], bv[NCT];
#pragma unroll
      for (int rt = 0; rt < NRT; ++rt) a[rt] = lds[ab[rt] + 2 * s];
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) bv[ct] = lds[bb[ct] + 2 * s];
#pragma unroll
      for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[rt], bv[ct], acc[rt][ct], 0, 0, 0);
    }
  }
  // ---- the four waves of the workgroup, added in wave order through LDS (register layout kept: [tile][q][lane])
  for (int w = 0; w < 4; ++w) {
    __syncthreads();
    if (wave == w) {
#pragma unroll
      for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            float* d = lds + ((rt * NCT + ct) * 16 + q) * 64 + lane;
            *d = (w == 0 ? 0.f : *d) + acc[rt][ct][q];
          }
    }
  }
  __syncthreads();
  float* dst = partial + (size_t)blockIdx.x * Cfg::ACCF;
  for (int e = tid; e < Cfg::ACCF; e += 256) dst[e] = lds[e];
}

// dw / dbias = sum over the workgroups' partials, in workgroup order.  One thread per (element, slice of the partials);
// 64 elements x 16 slices per block (the partials of a slice are independent loads: 16 in flight per thread with 256
// workgroups), slices combined in fixed order through LDS.  (With 4 slices and one load stream per thread this launch
// took 17-20 us, a fifth of the whole filter gradient: profiles/r02_run15_*.)
template <int C, int GA, int GB, int NRT, int NCT>
__global__ __launch_bounds__(1024) void tds_rsf_reduce_k(const float* __restrict__ partial, int nParts, int kw, float* __restrict__ dw,
                                                        float* __restrict__ dbias) {
  constexpr int ACCF = NRT * NCT * 16 * 64, NS = 16;
  __shared__ float red[NS][64];
  const int el = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + el;   // index in register order: ((rt*NCT + ct)*16 + q)*64 + lane
  float s = 0.f;
  const int per = (nParts + NS - 1) / NS;
  const int g0 = sl * per, g1 = g0 + per < nParts ? g0 + per : nParts;
  if (e < ACCF) {
    int g = g0;
    for (; g + 4 <= g1; g += 4) {
      const float a0 = partial[(size_t)g * ACCF + e], a1 = partial[(size_t)(g + 1) * ACCF + e], a2 = partial[(size_t)(g + 2) * ACCF + e],
                  a3 = partial[(size_t)(g + 3) * ACCF + e];
      s = (((s + a0) + a1) + a2) + a3;
    }
    for (; g < g1; ++g) s += partial[(size_t)g * ACCF + e];
  }
  red[sl][el] = s;
  __syncthreads();
  if (sl == 0 && e < ACCF) {
    float t = red[0][el];
#pragma unroll
    for (int k = 1; k < NS; ++k) t += red[k][el];
    const int lane = e & 63, q = (e >> 6) & 15, tl = e >> 10, rt = tl / NCT, ct = tl - rt * NCT;
    const int m = 32 * rt + 8 * (q >> 2) + 4 * (lane >> 5) + (q & 3), n = 32 * ct + (lane & 31);
    const int gb = n / C, co = n - gb * C;
    if (gb < GB) {
      if (m < GA * C) {
        const int ga = m / C, ci = m - ga * C, tap = ga * GB + gb;
        if (tap < kw) dw[((size_t)tap * C + ci) * C + co] = t;
      } else if (m == GA * C && gb == 0 && dbias) {
        dbias[co] = t;
      }
    }
  }
}

float* sk_scratch(hipStream_t s, size_t bytes);

}  // namespace w2l
#include "conv_tds_rsf3.hpp"
#include "conv_tds_tzf.hpp"
#include "conv_tds_c1.hpp"
namespace w2l {

template <int C, int GA, int GB, int HH, int TS>
static int rsf3_launch(const TdsRsfP& q, float* dw, float* dbias, hipStream_t s) {
  using Cfg = Rsf3Cfg<C, GA, GB, HH, TS>;
  if (q.H % HH) return W2L_EUNSUPPORTED;
  if ((long long)q.Tin * q.H * C * 4 >= (1ll << 31) || (long long)q.Tout * q.H * C * 4 >= (1ll << 31)) return W2L_EUNSUPPORTED;   // one utterance per buffer resource
  TdsRsf3P p{};
  p.x = q.x; p.dy = q.dy; p.B = q.B; p.Tin = q.Tin; p.Tout = q.Tout; p.H = q.H; p.kw = q.kw; p.padl = q.padl;
  p.hBlocks = q.H / HH;
  p.nStrips = (q.Tout + GB - 1 + TS - 1) / TS;
  const long long tiles = (long long)q.B * p.nStrips * p.hBlocks;
  if (tiles > (1ll << 30)) return W2L_EUNSUPPORTED;
  p.nTiles = (int)tiles;
  const int blocks = p.nTiles < 256 ? p.nTiles : 256;
  float* partial = sk_scratch(s, kSkScratchBytes);
  if (!partial || (size_t)blocks * Cfg::ACCF * sizeof(float) > kSkScratchBytes) return W2L_EUNSUPPORTED;
  static bool attr[64] = {};
  if (first_on_device(attr)) {
    W2L_HIP_CHECK(hipFuncSetAttribute((const void*)tds_conv_rsf3_k<C, GA, GB, HH, TS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS));
  }
  hipLaunchKernelGGL((tds_conv_rsf3_k<C, GA, GB, HH, TS>), dim3((unsigned)blocks), dim3(Cfg::WAVES * 64), Cfg::LDS, s, p, partial);
  hipLaunchKernelGGL((tds_rsf_reduce_k<C, GA, GB, Cfg::NRT, Cfg::NCT>), dim3((unsigned)((Cfg::ACCF + 63) / 64)), dim3(1024), 0, s, partial,
                     blocks, q.kw, dw, dbias);
  return W2L_OK;
}

// block-Toeplitz filter gradient (conv_tds_tzf.hpp)
template <int CI, int CO, int R, int GR, int SIG>
static int tzf_launch(const TdsRsfP& q, float* dw, float* dbias, hipStream_t s) {
  using Cfg = TzfCfg<CI, CO, R, GR, SIG>;
  if (q.H % Cfg::HB) return W2L_EUNSUPPORTED;
  if ((long long)q.Tin * q.H * CI * 4 >= (1ll << 31) || (long long)q.Tout * q.H * CO * 4 >= (1ll << 31)) return W2L_EUNSUPPORTED;   // one utterance per buffer resource
  TdsTzfP p{};
  p.x = q.x; p.dy = q.dy; p.B = q.B; p.Tin = q.Tin; p.Tout = q.Tout; p.H = q.H; p.kw = q.kw; p.padl = q.padl;
  p.hBlocks = q.H / Cfg::HB;
  p.rps = (q.Tout + Cfg::RF - 1) / Cfg::RF;
  const long long rounds = (long long)q.B * p.hBlocks * p.rps;
  if (rounds <= 0 || rounds > (1ll << 30)) return W2L_EUNSUPPORTED;
  p.nRounds = (int)rounds;
  const int wgs = p.nRounds < 256 ? p.nRounds : 256;   // one workgroup per CU
  p.rpw = (p.nRounds + wgs - 1) / wgs;
  const int blocks = (p.nRounds + p.rpw - 1) / p.rpw;
  float* partial = sk_scratch(s, kSkScratchBytes);
  if (!partial || (size_t)blocks * Cfg::ACCF * sizeof(float) > kSkScratchBytes) return W2L_EUNSUPPORTED;
  static bool attr[64] = {};
  if (first_on_device(attr)) {
    W2L_HIP_CHECK(hipFuncSetAttribute((const void*)tds_conv_tzf_k<CI, CO, R, GR, SIG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS));
  }
  hipLaunchKernelGGL((tds_conv_tzf_k<CI, CO, R, GR, SIG>), dim3((unsigned)blocks), dim3(512), Cfg::LDS, s, p, partial);
  hipLaunchKernelGGL((tds_tzf_reduce_k<CI, CO, R, GR, SIG>), dim3((unsigned)((q.kw * CI * CO + CO + 15) / 16)), dim3(1024), 0, s, partial, blocks, q.kw, dw, dbias);
  return W2L_OK;
}

template <int C, int GA, int GB, int TS>
static int rsf_launch(TdsRsfP p, float* dw, float* dbias, hipStream_t s) {
  using Cfg = RsfCfg<C, GA, GB, TS>;
  p.hBlocks = p.H / Cfg::HH;
  p.nStrips = (p.Tout + GB - 1 + TS - 1) / TS;
  const int nTiles = p.B * p.nStrips * p.hBlocks;
  const int perCu = (int)(160 * 1024 / Cfg::LDS) >= 3 ? 3 : 2;
  const int blocks = nTiles < 256 * perCu ? nTiles : 256 * perCu;
  float* partial = sk_scratch(s, kSkScratchBytes);
  if (!partial || (size_t)blocks * Cfg::ACCF * sizeof(float) > kSkScratchBytes) return W2L_EUNSUPPORTED;
  static bool attr[64] = {};
  if (Cfg::LDS > 64 * 1024 && first_on_device(attr)) {
    W2L_HIP_CHECK(hipFuncSetAttribute((const void*)tds_conv_rsf_k<C, GA, GB, TS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS));
  }
  hipLaunchKernelGGL((tds_conv_rsf_k<C, GA, GB, TS>), dim3((unsigned)blocks), dim3(256), Cfg::LDS, s, p, partial, nTiles);
  hipLaunchKernelGGL((tds_rsf_reduce_k<C, GA, GB, Cfg::NRT, Cfg::NCT>), dim3((unsigned)((Cfg::ACCF + 63) / 64)), dim3(1024), 0, s, partial,
                     blocks, p.kw, dw, dbias);
  return W2L_OK;
}

// the one-input-channel first layer of the TDS recipes (conv_tds_c1.hpp): forward, and filter + bias gradient
bool tds_c1_fwd_try(const float* x, const float* w, const float* bias, float* y, int B, int Tin, int Tout, int H, int Cout, int kw, int stride,
                    int padl, int relu, int profKind, hipStream_t s, int* status) {
  if (tune_env("W2L_TDS_C1_OFF") || Cout != 10 || kw > 21 || kw < 1 || (((uintptr_t)y) & 15) != 0) return false;
  TdsC1P p{};
  p.x = x; p.w = w; p.bias = bias; p.y = y; p.B = B; p.Tin = Tin; p.Tout = Tout; p.H = H; p.kw = kw; p.stride = stride; p.padl = padl; p.relu = relu;
  const long long total = (long long)B * Tout * H;
  if (total <= 0 || total > (1ll << 31) - 512) return false;
  prof_begin(s, 2.0 * B * Tout * (double)H * kw * Cout, profKind);
  hipLaunchKernelGGL((tds_c1_fwd_k<10, 21>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, p);
  prof_end(s);
  *status = hipGetLastError() == hipSuccess ? W2L_OK : W2L_EHIP;
  return true;
}

bool tds_c1_filter_try(const float* x, const float* dy, float* dw, float* dbias, int B, int Tin, int Tout, int H, int Cout, int kw, int stride,
                       int padl, hipStream_t s, int* status) {
  if (tune_env("W2L_TDS_C1_OFF") || Cout != 10 || kw > 21 || kw < 1 || stride != 2) return false;   // (the kernel's sliding window is written for stride 2)
  TdsC1P p{};
  p.x = x; p.dy = dy; p.B = B; p.Tin = Tin; p.Tout = Tout; p.H = H; p.kw = kw; p.stride = stride; p.padl = padl;
  const long long total = (long long)B * ((Tout + 3) / 4) * H;   // runs of four output frames
  if (total <= 0 || (long long)B * Tout * H > (1ll << 31) - 65536 * 128) return false;
  long long blocks = (total + 127) / 128;
  if (blocks > 512) blocks = 512;
  constexpr int ROW = 21 * 10 + 10;
  float* partial = sk_scratch(s, kSkScratchBytes);
  if (!partial || (size_t)blocks * ROW * sizeof(float) > kSkScratchBytes) return false;
  prof_begin(s, 2.0 * B * Tout * (double)H * kw * Cout, PROF_TDS_BWD_FILTER);
  hipLaunchKernelGGL((tds_c1_filter_k<10, 21>), dim3((unsigned)blocks), dim3(256), 0, s, p, partial);
  hipLaunchKernelGGL((tds_c1_filter_reduce_k<10, 21>), dim3((ROW + 31) / 32), dim3(1024), 0, s, partial, (int)blocks, kw, dw, dbias);
  prof_end(s);
  *status = hipGetLastError() == hipSuccess ? W2L_OK : W2L_EHIP;
  return true;
}

// true + *status when the block-Toeplitz filter gradient runs a strided sub-sampling layer (10 -> 14, 14 -> 18, stride 2)
bool tds_tzf_strided_try(const float* x, const float* dy, float* dw, float* dbias, int B, int Tin, int Tout, int H, int Cin, int Cout, int kw,
                         int stride, int padl, hipStream_t s, int* status) {
  if (tune_env("W2L_TDS_TZF_OFF") || tune_env("W2L_TDS_TZ_C2_OFF") || stride != 2 || kw > 21 || kw < 1 || H % 16) return false;
  if (!((Cin == 10 && Cout == 14) || (Cin == 14 && Cout == 18))) return false;
  if ((((uintptr_t)x | (uintptr_t)dy) & 15) != 0) return false;
  TdsRsfP p{};
  p.x = x; p.dy = dy; p.B = B; p.Tin = Tin; p.Tout = Tout; p.H = H; p.kw = kw; p.padl = padl;
  prof_begin(s, 2.0 * B * Tout * (double)H * kw * Cin * Cout, PROF_TDS_BWD_FILTER);
  int st = Cin == 10 ? tzf_launch<10, 14, 2, 12, 2>(p, dw, dbias, s) : tzf_launch<14, 18, 1, 12, 2>(p, dw, dbias, s);
  prof_end(s);
  if (st == W2L_EUNSUPPORTED) return false;
  if (st == W2L_OK && hipGetLastError() != hipSuccess) st = W2L_EHIP;
  *status = st;
  return true;
}

// true + *status when this geometry runs on the role-swapped filter-gradient kernel
bool tds_rsf_try(const float* x, const float* dy, float* dw, float* dbias, int B, int Tin, int Tout, int H, int C, int kw, int padl,
                 hipStream_t s, int* status) {
  if (tune_env("W2L_TDS_RS_OFF") || tune_env("W2L_TDS_RSF_OFF")) return false;
  if (!(C == 10 || C == 14 || C == 18) || kw > 21 || kw < 1 || H % 4) return false;
  if ((((uintptr_t)x | (uintptr_t)dy) & 15) != 0) return false;
  const bool tzf = !tune_env("W2L_TDS_TZF_OFF") && H % 16 == 0 && (C == 10 || C == 14);   // block-Toeplitz generation (conv_tds_tzf.hpp)
  if (C == 14 && !tzf && !tune_env("W2L_TDS_RS_C14")) return false;   // measured: 152 us against 103 us of conv_tds.hip's kernel
  TdsRsfP p{};
  p.x = x; p.dy = dy; p.B = B; p.Tin = Tin; p.Tout = Tout; p.H = H; p.kw = kw; p.padl = padl;
  prof_begin(s, 2.0 * B * Tout * (double)H * kw * C * C, PROF_TDS_BWD_FILTER);
  int st = W2L_EUNSUPPORTED;
  if (tzf) st = C == 10 ? tzf_launch<10, 10, 3, 16, 1>(p, dw, dbias, s) : tzf_launch<14, 14, 2, 12, 1>(p, dw, dbias, s);
  if (st != W2L_EUNSUPPORTED) {
  } else
  if (!tune_env("W2L_TDS_RSF3_OFF") && H % 8 == 0 && C != 14)   // wave-specialised generation (conv_tds_rsf3.hpp)
    st = C == 10 ? rsf3_launch<10, 3, 7, 8, 96>(p, dw, dbias, s) : rsf3_launch<18, 7, 3, 4, 96>(p, dw, dbias, s);
  if (st != W2L_EUNSUPPORTED) {
  } else
  if (C == 10) st = rsf_launch<10, 3, 7, 128>(p, dw, dbias, s);
  else if (C == 14) st = rsf_launch<14, 2, 11, 128>(p, dw, dbias, s);
  else st = rsf_launch<18, 7, 3, 96>(p, dw, dbias, s);
  prof_end(s);
  if (st == W2L_EUNSUPPORTED) return false;
  if (st == W2L_OK && hipGetLastError() != hipSuccess) st = W2L_EHIP;
  *status = st;
  return true;
}

}  // namespace w2l
