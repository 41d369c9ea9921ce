// gemm.hpp -- operand loaders and launch interface of the fp32 MFMA GEMM engine.
#pragma once
#include <vector>

#include "common.hpp"

#ifndef W2L_MIDSTORE
#define W2L_MIDSTORE 0
#endif
#ifndef W2L_BRANCHFREE
#define W2L_BRANCHFREE 0
#endif

namespace w2l {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { EPI_BIAS = 1, EPI_RELU = 2, EPI_ACCUM = 4, EPI_ATOMIC = 8, EPI_MASK = 16, EPI_DROPOUT = 32 };

struct GemmOut {
  float* C;
  const float* bias;
  int M, N, K, ldc;
  int epi;  // EPI_* flags (runtime: the epilogue is outside the K loop)
  const float* mask = nullptr;  // EPI_MASK: v = mask[m][n] > 0 ? v * maskScale : 0 (same ld as C)
  float maskScale = 1.f;
  const float* addend = nullptr;  // EPI_ACCUM: v += addend[m][n] (same ld as C); null = accumulate into C itself
  // EPI_DROPOUT (after bias / ReLU): v = keep_elem(m * ldc + n, seed, stream, thr) ? v * dropScale : 0 -- the same
  // stateless hash, on the same flat index, as dropout_k over the dense [M][ldc] output
  uint32_t dropThr = 0, dropSeed = 0, dropStream = 0;
  float dropScale = 1.f;
  // Row remap of the epilogue (overlapping-row convolution GEMMs, conv.hip): GEMM row m = b * rowPin + t' is stored as
  // output row b * rowPout + (t' - rowOff) if 0 <= t' - rowOff < rowPout and dropped otherwise.  rowPin = 0: identity.
  int rowPin = 0, rowPout = 0, rowOff = 0;
  uint32_t rowMul = 0, rowShr = 0;  // fast division by rowPin
  // Column sums of the B operand over the whole reduction, colsum[n] = sum_k B[k][n] (gemm160_kernel<false, false, *, true> only):
  // the bias gradient of fl::Linear riding on its weight-gradient product x^T dy -- the tiles of the first tile row add up the dy
  // fragments they already hold in registers.  csPart: per-slab partial sums of the stream-K ranges (behind the slabs in the
  // stream scratch), added in range order by the tile's last arriver: deterministic like the product itself.
  float* colsum = nullptr;
  float* csPart = nullptr;
  // bf16 images of the RESULT written by the epilogue itself (gemm128g_epilogue_wide<true>: the bf16-operand kernels), instead of or
  // beside the fp32 C (C may be null then): imgRows [M][ldImgRows] row-major, imgTrans [N][ldImgTrans] transposed -- what
  // w2l_bf16_convert would make of C, bit for bit (round to nearest even), only elements of the matrix are written (the zero
  // padding is the caller's).  maskH: the EPI_MASK operand as a bf16 row-major image (> 0 test on the bf16 value).
  uint16_t* imgRows = nullptr;
  uint16_t* imgTrans = nullptr;
  const uint16_t* maskH = nullptr;
  int ldImgRows = 0, ldImgTrans = 0, ldMaskH = 0;
  int ntStore = 0;   // probe (W2L_GEMM_NTSTORE): the wide epilogues store C nontemporal
};

inline void gemm_set_row_remap(GemmOut& o, int pin, int pout, int off) {
  o.rowPin = pin; o.rowPout = pout; o.rowOff = off;
  uint32_t sh = 0;
  while ((1ull << sh) < (uint32_t)pin) ++sh;
  o.rowShr = sh;
  o.rowMul = (uint32_t)((((1ull << sh) - (uint32_t)pin) << 32) / (uint32_t)pin + 1);
}

// GEMM row -> output row (false: the row is not stored)
__device__ __forceinline__ bool gemm_out_row(const GemmOut& o, int m, int& mo) {
  if (m >= o.M) return false;
  mo = m;
  if (!o.rowPin) return true;
  const uint32_t b = (uint32_t)(((uint64_t)__umulhi((uint32_t)m, o.rowMul) + (uint32_t)m) >> o.rowShr);
  const int t = m - (int)b * o.rowPin - o.rowOff;
  if (t < 0 || t >= o.rowPout) return false;
  mo = (int)b * o.rowPout + t;
  return true;
}

// A GEMM operand viewed as op(k, i): i = row of A (m) or column of B (n).
//   KCONTIG = true : element (k,i) at p[i*ld + k]   (reduction index contiguous)
//   KCONTIG = false: element (k,i) at p[k*ld + i]   ("k-rows")
// V = global-load vector width in floats (4/2/1), chosen by the host on alignment.
// load()/store() stage a [32][128] tile (k x i) through 16 registers per thread into
// the K-major LDS image lds[k*ldS + i]; *_bk<> are the [BK][BM] forms of the skinny kernel.
template <bool KCONTIG, int V>
struct PlainOp {
  const float* p;
  int ld;
  int extent;  // number of valid i
  int K;
  static constexpr int kPad = KCONTIG ? 1 : 4;
  static constexpr int kPadSkinny = KCONTIG ? 2 : 4;

  // Fast path of the 128x128x32 main loop: per-thread source pointers are computed ONCE per tile with
  // out-of-range rows / columns CLAMPED to valid ones (their products only reach accumulator rows /
  // columns that the epilogue never stores), so full K tiles load unconditionally -- no per-load
  // branches, selects or 64-bit multiplies between the MFMAs.  Only a ragged last K tile zero-fills.
  static constexpr bool kFast = true;
  static constexpr int kNP = 16 / V;
  struct Ptrs { const float* q[kNP]; };
  __device__ __forceinline__ void init(Ptrs& P, int i0, int tid) const {
    if (KCONTIG) {
      constexpr int TPR = 32 / V, RPP = 256 / TPR;
      const int kc = (tid % TPR) * V, rr = tid / TPR;
#pragma unroll
      for (int j = 0; j < kNP; ++j) {
        int gi = i0 + rr + RPP * j;
        if (gi > extent - 1) gi = extent - 1;
        P.q[j] = p + (size_t)gi * ld + kc;
      }
    } else {
      constexpr int TPR = 128 / V, RPP = 256 / TPR;
      const int kr = tid / TPR;
      int gi = i0 + (tid % TPR) * V;
      if (gi > extent - V) gi = extent - V;  // V > 1 implies extent % V == 0 (pick_vec): stays aligned
      if (gi < 0) gi = 0;
#pragma unroll
      for (int j = 0; j < kNP; ++j) P.q[j] = p + (size_t)(kr + RPP * j) * ld + gi;
    }
  }
  __device__ __forceinline__ void load_fast(float (&r)[16], const Ptrs& P, int k0) const {
    const size_t off = KCONTIG ? (size_t)k0 : (size_t)k0 * ld;
#pragma unroll
    for (int j = 0; j < kNP; ++j) {
      const float* src = P.q[j] + off;
      if (V == 4) {
        float4 v = *(const float4*)src;
        r[4 * j] = v.x; r[4 * j + 1] = v.y; r[4 * j + 2] = v.z; r[4 * j + 3] = v.w;
      } else if (V == 2) {
        float2 v = *(const float2*)src;
        r[2 * j] = v.x; r[2 * j + 1] = v.y;
      } else {
        r[j] = *src;
      }
    }
  }

  template <int BI, int BK>
  __device__ __forceinline__ void load_t(float (&r)[16], int i0, int k0, int tid) const {
    static_assert(BI * BK == 4096, "16 floats per thread");
    if (KCONTIG) {
      constexpr int TPR = BK / V;          // threads per row
      constexpr int RPP = 256 / TPR;       // rows per pass
      constexpr int NP = BI / RPP;         // passes
      const int kc = (tid % TPR) * V;
      const int rr = tid / TPR;
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        const int gi = i0 + rr + RPP * j, gk = k0 + kc;
        const bool ok = gi < extent && gk < K;
        // branch-free guard: load from a clamped (always valid) address, then select zero
        const float* src = p + (size_t)((ok || !W2L_BRANCHFREE) ? gi : 0) * ld + ((ok || !W2L_BRANCHFREE) ? gk : 0);
        if (V == 4) {
          float4 v = (W2L_BRANCHFREE || ok) ? *(const float4*)src : make_float4(0.f, 0.f, 0.f, 0.f);
          r[4 * j] = ok ? v.x : 0.f; r[4 * j + 1] = ok ? v.y : 0.f; r[4 * j + 2] = ok ? v.z : 0.f; r[4 * j + 3] = ok ? v.w : 0.f;
        } else if (V == 2) {
          float2 v = (W2L_BRANCHFREE || ok) ? *(const float2*)src : make_float2(0.f, 0.f);
          r[2 * j] = ok ? v.x : 0.f; r[2 * j + 1] = ok ? v.y : 0.f;
        } else {
          const float v = (W2L_BRANCHFREE || ok) ? *src : 0.f;
          r[j] = ok ? v : 0.f;
        }
      }
    } else {
      constexpr int TPR = BI / V;          // threads per k-row
      constexpr int RPP = 256 / TPR;       // k-rows per pass
      constexpr int NP = BK / RPP;
      const int ic = (tid % TPR) * V;
      const int kr = tid / TPR;
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        const int gi = i0 + ic, gk = k0 + kr + RPP * j;
        const bool ok = gi < extent && gk < K;
        const float* src = p + (size_t)((ok || !W2L_BRANCHFREE) ? gk : 0) * ld + ((ok || !W2L_BRANCHFREE) ? gi : 0);
        if (V == 4) {
          float4 v = (W2L_BRANCHFREE || ok) ? *(const float4*)src : make_float4(0.f, 0.f, 0.f, 0.f);
          r[4 * j] = ok ? v.x : 0.f; r[4 * j + 1] = ok ? v.y : 0.f; r[4 * j + 2] = ok ? v.z : 0.f; r[4 * j + 3] = ok ? v.w : 0.f;
        } else if (V == 2) {
          float2 v = (W2L_BRANCHFREE || ok) ? *(const float2*)src : make_float2(0.f, 0.f);
          r[2 * j] = ok ? v.x : 0.f; r[2 * j + 1] = ok ? v.y : 0.f;
        } else {
          const float v = (W2L_BRANCHFREE || ok) ? *src : 0.f;
          r[j] = ok ? v : 0.f;
        }
      }
    }
  }

  template <int BI, int BK>
  __device__ __forceinline__ void store_t(float* lds, int ldS, const float (&r)[16], int tid) const {
    if (KCONTIG) {
      constexpr int TPR = BK / V;
      constexpr int RPP = 256 / TPR;
      constexpr int NP = BI / RPP;
      const int kc = (tid % TPR) * V;
      const int rr = tid / TPR;
#pragma unroll
      for (int j = 0; j < NP; ++j)
#pragma unroll
        for (int e = 0; e < V; ++e) lds[(kc + e) * ldS + rr + RPP * j] = r[V * j + e];
    } else {
      constexpr int TPR = BI / V;
      constexpr int RPP = 256 / TPR;
      constexpr int NP = BK / RPP;
      const int ic = (tid % TPR) * V;
      const int kr = tid / TPR;
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        float* dst = lds + (kr + RPP * j) * ldS + ic;
        if (V == 4) *(float4*)dst = make_float4(r[4 * j], r[4 * j + 1], r[4 * j + 2], r[4 * j + 3]);
        else if (V == 2) *(float2*)dst = make_float2(r[2 * j], r[2 * j + 1]);
        else *dst = r[j];
      }
    }
  }

  __device__ __forceinline__ void load(float (&r)[16], int i0, int k0, int tid) const { load_t<128, 32>(r, i0, k0, tid); }
  __device__ __forceinline__ void store(float* lds, int ldS, const float (&r)[16], int tid) const { store_t<128, 32>(lds, ldS, r, tid); }
  template <int BM, int BK>
  __device__ __forceinline__ void load_bk(float (&r)[16], int i0, int k0, int tid) const { load_t<BM, BK>(r, i0, k0, tid); }
  template <int BM, int BK>
  __device__ __forceinline__ void store_bk(float* lds, int ldS, const float (&r)[16], int tid) const { store_t<BM, BK>(lds, ldS, r, tid); }

  // small B tiles of the skinny kernel: [BK][BN] with BN*BK/256 scalars per thread
  template <int BN, int BK>
  __device__ __forceinline__ void load_small(float (&r)[(BK * BN + 255) / 256], int n0, int k0, int tid) const {
    constexpr int NE = (BK * BN + 255) / 256;
#pragma unroll
    for (int j = 0; j < NE; ++j) {
      const int e = tid + 256 * j;
      int k, n;
      if (KCONTIG) { k = e % BK; n = e / BK; } else { n = e % BN; k = e / BN; }
      const int gn = n0 + n, gk = k0 + k;
      const bool ok = e < BK * BN && gn < extent && gk < K;
      r[j] = ok ? (KCONTIG ? p[(size_t)gn * ld + gk] : p[(size_t)gk * ld + gn]) : 0.f;
    }
  }
  template <int BN, int BK>
  __device__ __forceinline__ void store_small(float* lds, int ldS, const float (&r)[(BK * BN + 255) / 256], int tid) const {
    constexpr int NE = (BK * BN + 255) / 256;
#pragma unroll
    for (int j = 0; j < NE; ++j) {
      const int e = tid + 256 * j;
      int k, n;
      if (KCONTIG) { k = e % BK; n = e / BK; } else { n = e % BN; k = e / BN; }
      if (e < BK * BN) lds[k * ldS + n] = r[j];
    }
  }
};

// ---------------------------------------------------------------- launch profiling
// bench.py's roofline leg: when enabled, every launch of a roofline-relevant kernel is bracketed
// by a pair of hipEvents on ITS stream; w2l_profile_report(kind) sums durations and the
// algorithmic work (FLOPs for the MFMA kinds, bytes for the HBM-streaming kind).
enum { PROF_GEMM128 = 0, PROF_SKINNY = 1, PROF_TDSCONV = 2 /* forward */, PROF_FCC_STREAM = 3, PROF_TDS_BWD_DATA = 4,
       PROF_TDS_BWD_FILTER = 5, PROF_GEMM_BF16 = 6, PROF_KINDS = 7 };
struct GemmProf {
  bool on = false;
  std::vector<hipEvent_t> ev;      // pairs
  std::vector<double> work;
  std::vector<int> kind;
  std::vector<int> dims;           // 4 per launch: M, N, K, kernel tag (0 = not a GEMM / unknown): the per-shape table of w2l_profile_launches
  size_t used = 0;
};
GemmProf& gemm_prof();
// kernel tags of the per-launch table: 1 = gemm128_kernel (generic operands), 2 = gemm128g_kernel (LDS-DMA 128 x 128), 3 = gemm160_kernel
// 128 x 160, 4 = gemm160_kernel 160 x 128 ("tall")
inline void prof_begin(hipStream_t s, double work, int kind = PROF_GEMM128, int M = 0, int N = 0, int K = 0, int tag = 0) {
  GemmProf& p = gemm_prof();
  if (!p.on) return;
  if (p.used + 2 > p.ev.size()) {
    for (int i = 0; i < 2; ++i) { hipEvent_t e; (void)hipEventCreate(&e); p.ev.push_back(e); }
  }
  p.work.push_back(work);
  p.kind.push_back(kind);
  p.dims.push_back(M); p.dims.push_back(N); p.dims.push_back(K); p.dims.push_back(tag);
  (void)hipEventRecord(p.ev[p.used], s);
}
inline void prof_end(hipStream_t s) {
  GemmProf& p = gemm_prof();
  if (!p.on) return;
  (void)hipEventRecord(p.ev[p.used + 1], s);
  p.used += 2;
}

// ---------------------------------------------------------------- stream-K schedule
// A data-parallel grid of 128x128 tiles quantises badly on the TDS shapes (M = 6016, N = 1440:
// 564 tiles on 512 resident workgroups = 2 rounds for 1.1 rounds of work).  Hybrid schedule:
// the first `dpTiles` tiles (whole rounds) run one tile per workgroup; the K iterations of the
// remaining `skTiles` tiles are cut into `skBlocks` EQUAL contiguous ranges, one per workgroup
// (each range touches at most two tiles).  A range that covers a whole tile runs the normal
// epilogue; otherwise the accumulators go to a per-workgroup slab (in MFMA register order,
// coalesced) and gemm128_fixup adds the slabs of a tile IN RANGE ORDER and runs the epilogue:
// no atomics, run-to-run deterministic, no pre-zeroed output.
struct SkPlan {
  int tilesM, tilesN, kTiles;
  int dpTiles, skTiles, skBlocks;
  float* slabs;  // [skBlocks][2][128*128]
  int grouped;   // tile rasterisation: 0 = M-fastest, 1 = groups of 8 tile-columns, N-fastest inside a group
  unsigned* counters;  // per stream-K tile arrival tickets for the in-kernel slab reduction (null: separate fix-up launch)
  // aligned K split (gemm160 only; 0 = off): the K axis is cut into `ksplit` chunks of kChunk K tiles, the SAME cut for
  // every tile, and the units (chunk, tile) are dealt chunk-major: unit u = worker + round * workers.  The 64 workers of
  // an XCD then multiply 64 neighbouring tiles over the same K range at the same time and share their operand panels
  // in that XCD's L2 -- the classic stream-K ranges start at a different k in every tile and share nothing (weight
  // gradients: 10 % L2 hits, profiles/r02_run17_*).  One partial slab per unit (index u), ksplit arrivals per tile.
  int ksplit = 0, kChunk = 0;
  // probe library only (W2L_GEMM_DBG = device address): per workgroup {start, end (100 MHz ticks), hardware id, XCC id}
  long long* dbg = nullptr;
  // wave priority of the two workgroups that share a CU (g_tile_prio, gemm_glds.hpp): 0 = leave it, 1 = alternate per segment
  int prio = 0;
};
constexpr int kSkSlots = 512;           // resident 256-thread workgroups (2 per CU)
constexpr int kSlabFloats = 128 * 128;
// one stream scratch serves every user (stream-K slabs of the 128x128 and 128x160 kernels, conv filter partials,
// column-sum partials): all of them request THIS size so the buffer is allocated once per stream
constexpr size_t kSkSlabBytes = (size_t)kSkSlots * 2 * (128 * 160) * sizeof(float);
constexpr size_t kSkScratchBytes = kSkSlabBytes + (size_t)kSkSlots * 2 * 160 * sizeof(float);   // + column-sum partials (GemmOut::csPart)

__host__ __device__ inline long long sk_begin(const SkPlan& p, int s) {
  return (long long)p.skTiles * p.kTiles * s / p.skBlocks;
}

// linear tile index -> tile coordinates (bijective for either rasterisation)
__host__ __device__ inline void sk_tile_xy(const SkPlan& p, int t, int& bx, int& by) {
  if (!p.grouped) { bx = t % p.tilesM; by = t / p.tilesM; return; }
  const int perGroup = 8 * p.tilesM;
  const int grp = t / perGroup, first = grp * 8;
  const int gsize = p.tilesN - first < 8 ? p.tilesN - first : 8;
  const int tin = t - grp * perGroup;
  by = first + tin % gsize;
  bx = tin / gsize;
}

inline SkPlan make_sk_plan(int M, int N, int K, bool allowSk, int tileM = 128, int tileN = 128, int slots = kSkSlots) {
  SkPlan p;
  p.tilesM = (M + tileM - 1) / tileM;
  p.tilesN = (N + tileN - 1) / tileN;
  p.kTiles = (K + 31) / 32;
  const int tiles = p.tilesM * p.tilesN;
  p.dpTiles = tiles; p.skTiles = 0; p.skBlocks = 0; p.slabs = nullptr; p.grouped = 0; p.counters = nullptr;
  p.ksplit = 0; p.kChunk = 0; p.dbg = nullptr; p.prio = 0;
  if (!allowSk || p.kTiles < 8) return p;
  const int rounds = (tiles + slots - 1) / slots;
  const double eff = (double)tiles / ((double)rounds * slots);
  if (eff >= 0.93) return p;
  // A grid that fits one workgroup per CU with a SHORT reduction is better left data-parallel: splitting it 512 ways costs
  // a partial-tile round trip per worker (~43 us at 192 tiles) that a K of a few tiles cannot repay.  Fitted on MI355X
  // (profiles/r02_run22_gemm_c5_shapes.log; M = 3008, N = 1024: K = 1024 -> 74.5 us data-parallel against 86.1 stream-K,
  // K = 4096 -> 258 against 213):  t_dp = 2.1 kTiles + 7,  t_sk = 3.6 tiles kTiles / 512 + 43  [us]
  // (fitted between 128 and 256 tiles; a grid of a few tiles still gains from being split, its partial traffic is small)
  if (slots == kSkSlots && tiles > kSkSlots / 4 && tiles <= kSkSlots / 2 && (double)p.kTiles * (2.1 - 3.6 * tiles / kSkSlots) < 36.0) return p;
  const int full = tiles / slots;
  p.dpTiles = full * slots;
  p.skTiles = tiles - p.dpTiles;
  long long iters = (long long)p.skTiles * p.kTiles;
  long long blocks = iters / 4;  // >= 4 K iterations per workgroup
  if (blocks > slots) blocks = slots;
  if (blocks < 1) blocks = 1;
  p.skBlocks = (int)blocks;
  // every range must stay within two tiles
  if (iters / p.skBlocks + 1 > p.kTiles) { p.dpTiles = tiles; p.skTiles = 0; p.skBlocks = 0; }
  return p;
}

float* sk_scratch(hipStream_t s, size_t bytes);  // library-owned, one buffer per stream (gemm.hip)
bool matmul_bf16_mode();                         // w2l_set_matmul_precision(1) is in force (the mixed-precision network pass)
int gemm_bf16_images(const uint16_t* A, int lda, unsigned long long aView, const uint16_t* B, int ldb, unsigned long long bView,
                     const GemmOut& o, int epi, hipStream_t s);   // gemm_bf16g.hpp's launch128h (gemm.hip)
unsigned* sk_counters(hipStream_t s);            // 1024 zero-initialised arrival tickets per stream (self-resetting)

// first / last stream-K range that overlaps stream-K tile t
__host__ __device__ inline void sk_tile_ranges(const SkPlan& p, int t, int& sFirst, int& sLast) {
  const long long tb = (long long)t * p.kTiles, te = tb + p.kTiles;
  const long long I = (long long)p.skTiles * p.kTiles;
  int s = (int)(tb * p.skBlocks / I);
  while (s + 1 < p.skBlocks && sk_begin(p, s + 1) <= tb) ++s;
  while (s > 0 && sk_begin(p, s) > tb) --s;
  sFirst = s;
  while (s + 1 < p.skBlocks && sk_begin(p, s + 1) < te) ++s;
  sLast = s;
}
bool sk_enabled();                               // W2L_GEMM_SK=0 turns the schedule off (A/B runs)

// ---------------------------------------------------------------- kernels
template <class AOp, class BOp>
__device__ __forceinline__ void gemm128_mainloop(const AOp& aop, const BOp& bop, int m0, int n0, int ktBegin, int ktEnd,
                                                 int K, float* smem, f32x16 (&acc)[2][2]) {
  constexpr int BM = 128, BN = 128, BK = 32;
  constexpr int LDA_S = BM + AOp::kPad, LDB_S = BN + BOp::kPad;
  float* As0 = smem;
  float* Bs0 = As0 + BK * LDA_S;
  float* As1 = Bs0 + BK * LDB_S;
  float* Bs1 = As1 + BK * LDA_S;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  const int li = lane & 31, lh = lane >> 5;

#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float ra[16], rb[16];
  typename AOp::Ptrs pa;
  typename BOp::Ptrs pb;
  // hoisted pointers pay off on long reductions (dW: +3 %, 4096^3: +2..6 %); on short K (fc1: K = 800)
  // their per-tile set-up costs more than it saves (-3 %, within-run A/B on MI355X)
  const bool fast = K >= 2048;
  if (AOp::kFast && fast) aop.init(pa, m0, tid);
  if (BOp::kFast && fast) bop.init(pb, n0, tid);
  const int kFull = fast ? K / BK : 0;  // K tiles below this index load unconditionally
  auto loadA = [&](int kt) {
    if (AOp::kFast && kt < kFull) aop.load_fast(ra, pa, kt * BK);
    else aop.load(ra, m0, kt * BK, tid);
  };
  auto loadB = [&](int kt) {
    if (BOp::kFast && kt < kFull) bop.load_fast(rb, pb, kt * BK);
    else bop.load(rb, n0, kt * BK, tid);
  };
  __syncthreads();  // a previous segment of this workgroup may still be reading the LDS tiles
  if (ktBegin < ktEnd) {
    loadA(ktBegin);
    loadB(ktBegin);
    aop.store(As0, LDA_S, ra, tid);
    bop.store(Bs0, LDB_S, rb, tid);
  }
  __syncthreads();

  for (int kt = ktBegin; kt < ktEnd; ++kt) {
    const bool more = kt + 1 < ktEnd;
    const int cur = (kt - ktBegin) & 1;
    const float* As = cur ? As1 : As0;
    const float* Bs = cur ? Bs1 : Bs0;
    if (more) {
      loadA(kt + 1);
      loadB(kt + 1);
    }
    // Fragments of k-pair kp+1 are read from LDS BEFORE the four MFMAs of k-pair kp are issued (two
    // named register sets, order pinned with sched_barrier): left alone hipcc emits
    // "ds_read; s_waitcnt lgkmcnt(0); 4 x mfma" per k-pair and the LDS latency is exposed 16 times
    // per K tile (MFMA pipe 75 % busy at 4096^3 by rocprof PMC).
    const float* ar = As + lh * LDA_S + wm + li;
    const float* br = Bs + lh * LDB_S + wn + li;
    float a0 = ar[0], a1 = ar[32], b0 = br[0], b1 = br[32];
    float c0, c1, d0, d1;
#pragma unroll
    for (int kp = 0; kp < BK / 2; kp += 2) {
      if (W2L_MIDSTORE && kp == BK / 4 && more) {
        // the next tile's registers go to the other LDS buffer in the MIDDLE of the MFMA phase (their
        // global loads were issued ~2000 cycles ago), so no separate write phase precedes the barrier
        aop.store(cur ? As0 : As1, LDA_S, ra, tid);
        bop.store(cur ? Bs0 : Bs1, LDB_S, rb, tid);
        __builtin_amdgcn_sched_barrier(0);
      }
      c0 = ar[(2 * kp + 2) * LDA_S]; c1 = ar[(2 * kp + 2) * LDA_S + 32];
      d0 = br[(2 * kp + 2) * LDB_S]; d1 = br[(2 * kp + 2) * LDB_S + 32];
      __builtin_amdgcn_sched_barrier(0);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (kp + 2 < BK / 2) {
        a0 = ar[(2 * kp + 4) * LDA_S]; a1 = ar[(2 * kp + 4) * LDA_S + 32];
        b0 = br[(2 * kp + 4) * LDB_S]; b1 = br[(2 * kp + 4) * LDB_S + 32];
      }
      __builtin_amdgcn_sched_barrier(0);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(c0, d0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(c0, d1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(c1, d0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(c1, d1, acc[1][1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (!W2L_MIDSTORE && more) {
      aop.store(cur ? As0 : As1, LDA_S, ra, tid);
      bop.store(cur ? Bs0 : Bs1, LDB_S, rb, tid);
    }
    __syncthreads();
  }
}

// epilogue. C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
__device__ __forceinline__ void gemm128_epilogue(const GemmOut& out, int m0, int n0, const f32x16 (&acc)[2][2],
                                                 int waveOverride = -1) {
  const int EPI = out.epi;
  const int lane = threadIdx.x & 63, wave = waveOverride >= 0 ? waveOverride : (int)(threadIdx.x >> 6);
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  const int li = lane & 31, lh = lane >> 5;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wn + j * 32 + li;
      if (n >= out.N) continue;
      float bv = 0.f;
      if (EPI & EPI_BIAS) bv = out.bias[n];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int m;
        if (!gemm_out_row(out, m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, m)) continue;
        float v = acc[i][j][r] + bv;
        float* dst = out.C + (size_t)m * out.ldc + n;
        if (EPI & EPI_RELU) v = fmaxf(v, 0.f);
        if (EPI & EPI_DROPOUT) v = keep_elem((uint64_t)m * out.ldc + n, out.dropSeed, out.dropStream, out.dropThr) ? v * out.dropScale : 0.f;
        if (EPI & EPI_MASK) v = out.mask[(size_t)m * out.ldc + n] > 0.f ? v * out.maskScale : 0.f;
        if (EPI & EPI_ACCUM) v += out.addend ? out.addend[(size_t)m * out.ldc + n] : *dst;
        *dst = v;
      }
    }
}

// slab float4 of (acc index a = 2i+j, register quad q) of this thread: ((wave*4 + a)*4 + q)*64 + lane
// (16-byte stores: 16 per lane instead of 64 dword stores)
__device__ __forceinline__ void gemm128_store_partial(float* slab, const f32x16 (&acc)[2][2]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
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

template <class AOp, class BOp>
__global__ __launch_bounds__(256, 2) void gemm128_kernel(AOp aop, BOp bop, GemmOut out, SkPlan plan) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  f32x16 acc[2][2];
  const int bid = blockIdx.x;
  // tile coordinates: the tile index walks M fastest so that neighbouring workgroups on one
  // XCD share the same B (weight) panel in L2
  if (bid < plan.dpTiles) {
    int bx, by;
    sk_tile_xy(plan, bid, bx, by);
    gemm128_mainloop(aop, bop, bx * 128, by * 128, 0, plan.kTiles, out.K, smem, acc);
    gemm128_epilogue(out, bx * 128, by * 128, acc);
    return;
  }
  const int s = bid - plan.dpTiles;
  long long it = sk_begin(plan, s);
  const long long itEnd = sk_begin(plan, s + 1);
  int seg = 0;
  while (it < itEnd) {
    const int tile = plan.dpTiles + (int)(it / plan.kTiles);
    const int kb = (int)(it % plan.kTiles);
    int ke = plan.kTiles;
    if (itEnd - it < (long long)(ke - kb)) ke = kb + (int)(itEnd - it);
    int bx, by;
    sk_tile_xy(plan, tile, bx, by);
    gemm128_mainloop(aop, bop, bx * 128, by * 128, kb, ke, out.K, smem, acc);
    if (kb == 0 && ke == plan.kTiles) gemm128_epilogue(out, bx * 128, by * 128, acc);
    else gemm128_store_partial(plan.slabs + ((size_t)s * 2 + seg) * kSlabFloats, acc);
    it += ke - kb;
    ++seg;
  }
}

// one WAVEFRONT per quadrant of a stream-K tile (grid = 4 x skTiles workgroups of 64 threads: four times the
// waves of the one-workgroup-per-tile version, the kernel is latency-bound on the slab reads): add the
// partial slabs in range order, then the epilogue
template <int kUnused>
__global__ __launch_bounds__(64) void gemm128_fixup(GemmOut out, SkPlan plan) {
  const int t = blockIdx.x >> 2;  // index inside the stream-K region
  const int wave = blockIdx.x & 3, lane = threadIdx.x;
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
    const f32x4* s4 = (const f32x4*)(plan.slabs + ((size_t)s * 2 + segIdx) * kSlabFloats);
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
  const int tile = plan.dpTiles + t;
  int bx, by;
  sk_tile_xy(plan, tile, bx, by);
  gemm128_epilogue(out, bx * 128, by * 128, acc, wave);
}

// Skinny-N variant for small output widths (TDS time convolutions, C = 10..18):
// 16x16x4 MFMA, block tile 256 x BN (BN = 16 or 32), 4 waves stacked along M, each
// wave 64 x BN = 4 x (BN/16) MFMA tiles (>= 4 independent accumulators cover the
// 40-cycle dependent latency of v_mfma_f32_16x16x4_f32).
template <class AOp, class BOp, int BN>
__global__ __launch_bounds__(256, 2) void gemm_skinny_kernel(AOp aop, BOp bop, GemmOut out) {
  constexpr int BM = 256, BK = 16;
  constexpr int NT = BN / 16;
  constexpr int LDA_S = BM + AOp::kPadSkinny, LDB_S = BN + 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As0 = smem;
  float* Bs0 = As0 + BK * LDA_S;
  float* As1 = Bs0 + BK * LDB_S;
  float* Bs1 = As1 + BK * LDA_S;

  const int EPI = out.epi;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave * 64;
  const int li = lane & 15, lq = lane >> 4;

  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int kTilesTotal = (out.K + BK - 1) / BK;
  const int perSplit = (kTilesTotal + gridDim.z - 1) / gridDim.z;
  const int ktBegin = blockIdx.z * perSplit;
  int ktEnd = ktBegin + perSplit;
  if (ktEnd > kTilesTotal) ktEnd = kTilesTotal;

  f32x4 acc[4][NT];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

  float ra[16];
  float rb[(BK * BN + 255) / 256];
  if (ktBegin < ktEnd) {
    aop.template load_bk<BM, BK>(ra, m0, ktBegin * BK, tid);
    bop.template load_small<BN, BK>(rb, n0, ktBegin * BK, tid);
    aop.template store_bk<BM, BK>(As0, LDA_S, ra, tid);
    bop.template store_small<BN, BK>(Bs0, LDB_S, rb, tid);
  }
  __syncthreads();
  for (int kt = ktBegin; kt < ktEnd; ++kt) {
    const bool more = kt + 1 < ktEnd;
    const int cur = (kt - ktBegin) & 1;
    const float* As = cur ? As1 : As0;
    const float* Bs = cur ? Bs1 : Bs0;
    if (more) {
      aop.template load_bk<BM, BK>(ra, m0, (kt + 1) * BK, tid);
      bop.template load_small<BN, BK>(rb, n0, (kt + 1) * BK, tid);
    }
#pragma unroll
    for (int kq = 0; kq < BK / 4; ++kq) {
      const float* ar = As + (4 * kq + lq) * LDA_S + wm + li;
      const float* br = Bs + (4 * kq + lq) * LDB_S + li;
      float a[4], b[NT];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = ar[16 * i];
#pragma unroll
      for (int j = 0; j < NT; ++j) b[j] = br[16 * j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (more) {
      aop.template store_bk<BM, BK>(cur ? As0 : As1, LDA_S, ra, tid);
      bop.template store_small<BN, BK>(cur ? Bs0 : Bs1, LDB_S, rb, tid);
    }
    __syncthreads();
  }
  // C/D layout of 16x16 MFMA: col = lane&15, row = 4*(lane>>4) + r
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = n0 + j * 16 + li;
      if (n >= out.N) continue;
      float bv = 0.f;
      if ((EPI & EPI_BIAS) && blockIdx.z == 0) bv = out.bias[n];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm + i * 16 + 4 * lq + r;
        if (m >= out.M) continue;
        float v = acc[i][j][r] + bv;
        float* dst = out.C + (size_t)m * out.ldc + n;
        if (EPI & EPI_ATOMIC) {
          atomicAdd(dst, v);
        } else {
          if (EPI & EPI_RELU) v = fmaxf(v, 0.f);
          if (EPI & EPI_DROPOUT) v = keep_elem((uint64_t)m * out.ldc + n, out.dropSeed, out.dropStream, out.dropThr) ? v * out.dropScale : 0.f;
          if (EPI & EPI_MASK) v = out.mask[(size_t)m * out.ldc + n] > 0.f ? v * out.maskScale : 0.f;
          if (EPI & EPI_ACCUM) v += out.addend ? out.addend[(size_t)m * out.ldc + n] : *dst;
          *dst = v;
        }
      }
    }
}

// ---------------------------------------------------------------------------
template <class AOp, class BOp>
inline int launch128(const AOp& a, const BOp& b, GemmOut o, int epi, int splitk, hipStream_t s) {
  constexpr int BK = 32;
  (void)splitk;          // K is split by the stream-K schedule, deterministically
  epi &= ~EPI_ATOMIC;    // (callers of the old atomic split-K path need no pre-zeroed C any more)
  const size_t shmem = 2 * (size_t)BK * ((128 + AOp::kPad) + (128 + BOp::kPad)) * sizeof(float);
  SkPlan plan = make_sk_plan(o.M, o.N, o.K, sk_enabled());
  if (plan.skBlocks > 0) {
    plan.slabs = sk_scratch(s, kSkScratchBytes);
    if (!plan.slabs) plan = make_sk_plan(o.M, o.N, o.K, false);
  }
  dim3 grid((unsigned)(plan.dpTiles + plan.skBlocks)), block(256);
  o.epi = epi;
  prof_begin(s, 2.0 * o.M * (double)o.N * o.K, PROF_GEMM128, o.M, o.N, o.K, 1);
  hipLaunchKernelGGL((gemm128_kernel<AOp, BOp>), grid, block, shmem, s, a, b, o, plan);
  if (plan.skBlocks > 0) hipLaunchKernelGGL(gemm128_fixup<0>, dim3((unsigned)plan.skTiles * 4), dim3(64), 0, s, o, plan);
  prof_end(s);
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}

template <class AOp, class BOp, int BN>
inline int launch_skinny(const AOp& a, const BOp& b, GemmOut o, int epi, int splitk, hipStream_t s) {
  constexpr int BK = 16;
  const size_t shmem = 2 * (size_t)BK * ((256 + AOp::kPadSkinny) + (BN + 4)) * sizeof(float);
  dim3 grid((unsigned)((o.M + 255) / 256), (unsigned)((o.N + BN - 1) / BN), (unsigned)splitk), block(256);
  o.epi = epi;
  prof_begin(s, 2.0 * o.M * (double)o.N * o.K, PROF_SKINNY);
  hipLaunchKernelGGL((gemm_skinny_kernel<AOp, BOp, BN>), grid, block, shmem, s, a, b, o);
  prof_end(s);
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}

// C[M][N] = op(A)[M][K] . op(B)[K][N] (+bias[n]) (relu) ; a_kcontig: A is [M][K] row-major;
// b_kcontig: B is stored [N][K] row-major.  epi = EPI_* flags; splitk > 1 needs EPI_ATOMIC
// and a pre-zeroed C.
struct GemmExtra {  // optional epilogue operands of gemm_f32
  const float* addend = nullptr;
  uint32_t dropThr = 0, dropSeed = 0, dropStream = 0;
  float dropScale = 1.f;
  float* colsum = nullptr;  // also produce colsum[n] = sum_k B[k][n] (GemmOut::colsum) where the kernel that runs can ...
  mutable bool colsumDone = false;   // ... and say so: false -> the caller runs its own column-sum launch
};
int colsum(const float* x, float* out, size_t M, int N, hipStream_t s);   // conv.hip: out[n] = sum_m x[m][n]
int gemm_f32(const float* A, int lda, int a_kcontig, const float* B, int ldb, int b_kcontig, float* C,
             int ldc, int M, int N, int K, const float* bias, int epi, int splitk, hipStream_t s,
             const float* mask = nullptr, float maskScale = 1.f, const GemmExtra* extra = nullptr);
// LDS-DMA kernels on caller-prepared operand views (see gemm.hip); o carries M, N, K, C, ldc, bias and the row remap
int gemm_glds_raw(const float* A, int lda, bool akc, size_t aBytes, const float* B, int ldb, bool bkc, size_t bBytes,
                  GemmOut o, int epi, hipStream_t s);

}  // namespace w2l
