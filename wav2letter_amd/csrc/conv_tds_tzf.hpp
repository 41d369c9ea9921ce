// conv_tds_tzf.hpp -- the filter gradient of the TDS time convolution (C = 10 / 14) in the BLOCK-TOEPLITZ form of
// conv_tds_tz.hpp:   dW[j][ci][co] = sum_{b, h, t} x[t + j - padl][h][ci] dy[t][h][co]   (+ the bias gradient sum dy).
// With t = R i + r the product of the two activations is ONE outer-product accumulation
//     D[(s, ci)][(r, co)] = sum_{(i, h)}  x[R i + s - padl][h][ci] * dy[R i + r][h][co],        s < S = R + 20
//     dW[j][ci][co]       = sum_{r < R}   D[(j + r, ci)][(r, co)]        (strided layers: s < SIG (R - 1) + 21, x[SIG R i + s], D[(j + SIG r, ci)])
// rows (s, ci) = S C of the MFMA's 32-row tiles, columns (r, co) = R C <= 32, the reduction runs over (group i, mel row h):
//     C = 10: R = 3, 115 row PAIRS (s, ci / 2) in 4 pair tiles, 30 of 32 columns      useful / issued 0.77
//     C = 14: R = 2, 154 row pairs in 5 pair tiles, 28 of 32 columns                  0.80
// (the role-swapped generation, conv_tds_rsf3.hpp: 0.67 at C = 10; C = 14 ran on the round-1 kernel of conv_tds.hip.)
// Both slabs keep the GLOBAL layout [frame][h][c] (buffer_load_dwordx4 ... lds of whole frames, zero padding by the
// range check, no transposition, no mover waves).  One ds_read_b64 of x gives a lane the channel pair (2 cp, 2 cp + 1) of
// its row pair: the A operands of TWO accumulators (even / odd channel); the dy fragment is one ds_read_b32 shared by all
// 2 NPT accumulators of the step: NPT + 1 DS instructions per 2 NPT MFMAs (0.6), all MFMAs of a step independent.
// Machine shape: ONE workgroup of eight waves per CU; wave w owns the mel rows 2 w, 2 w + 1 of the strip (the two lane
// halves of the MFMA's K pair) and walks the groups i of the round; 2 NPT accumulators (128 / 160 registers) live in
// registers over ALL rounds of the workgroup; the staging of the next round rides between the MFMA groups; one
// LDS/DMA barrier per round.  The bias gradient is a spare row pair whose lanes read 1.0 out of the x slab's frame
// padding (never touched by the staging).  At the end the eight waves are added through LDS in wave order, one partial
// image per workgroup goes to the stream scratch, and tds_tzf_reduce_k adds the workgroups in order and folds the R
// diagonals of D into dW: deterministic.
#pragma once

namespace w2l {

struct TdsTzfP {
  const float* x;   // [B][Tin][H][C]
  const float* dy;  // [B][Tout][H][C]
  int B, Tin, Tout, H, kw, padl;
  int hBlocks, rps, nRounds, rpw;
};

template <int CI, int CO, int R, int GR, int SIG>
struct TzfCfg {
  static constexpr int KW = 21, HB = 16;
  static constexpr int S = SIG * (R - 1) + KW;
  static constexpr int C2 = CI / 2;
  static constexpr int NPR = S * C2;                   // row pairs (s, cp); pair NPR is the all-ones row of the bias gradient
  static constexpr int NPT = (NPR + 1 + 31) / 32;      // pair tiles
  static constexpr int RF = GR * R;                    // output frames of a round
  static constexpr int NFX = SIG * R * (GR - 1) + S;   // x slab frames (groups SIG R input frames apart)
  static constexpr int NFD = GR * R;                   // dy slab frames
  // x frame pitch: > HB C (the padding holds the 1.0 of the bias row), = 8 or 16 (mod 32) dwords so that the 32 row
  // pairs of a lane half spread over the 64-bit bank pairs
  static constexpr int px_pick() {
    int p = HB * CI + 4;
    while (p % 64 != 40 && p % 64 != 48 && p % 64 != 8 && p % 64 != 16 && p % 64 != 24 && p % 64 != 56) p += 4;
    return p;
  }
  static constexpr int PX = px_pick();
  static constexpr int PD = HB * CO + ((HB * CO) % 32 == 0 ? 12 : 0);   // dy frame pitch: r PD (mod 32) apart from the 0 .. CO-1 of r = 0
  static constexpr int CPFX = HB * CI / 4, CPFD = HB * CO / 4;   // 16-byte chunks of a frame of the strip
  static constexpr int PARTSD = (CPFD + 63) / 64;      // LDS-DMA instructions per dy frame
  static_assert(CPFX <= 64, "one LDS-DMA instruction per x frame");
  static constexpr int XB = NFX * PX * 4, DB = NFD * PD * 4;
  static constexpr int BUFB = (XB + DB + 15) / 16 * 16;
  static constexpr int ACCF = NPT * 2 * 16 * 64;       // floats of one wave's accumulators = of a workgroup's partial image
  static constexpr size_t LDS = 2 * (size_t)BUFB > 8 * 2 * 16 * 64 * 4 ? 2 * (size_t)BUFB : 8 * 2 * 16 * 64 * 4;
  static constexpr int NDX = (NFX + 7) / 8, NDD = (NFD * PARTSD + 7) / 8;   // LDS-DMA instructions per wave and round
  static_assert(CI % 2 == 0 && R * CO <= 32 && 2 * GR >= NDX + NDD && NPT >= 2, "shape");
  static_assert(LDS <= 160 * 1024, "LDS");
  static_assert((GR - 1) * SIG * R * PX * 4 + HB * CI * 4 + 8 < 65536 && (GR - 1) * R * PD * 4 < 65536, "ds offset field");
};

template <int CI, int CO, int R, int GR, int SIG>
__global__ __launch_bounds__(512) void tds_conv_tzf_k(TdsTzfP p, float* __restrict__ partial) {
  using Cfg = TzfCfg<CI, CO, R, GR, SIG>;
  constexpr int HB = Cfg::HB, C2 = Cfg::C2, NPR = Cfg::NPR, NPT = Cfg::NPT, RF = Cfg::RF, NFX = Cfg::NFX, NFD = Cfg::NFD,
                PX = Cfg::PX, PD = Cfg::PD, CPFX = Cfg::CPFX, CPFD = Cfg::CPFD, PARTSD = Cfg::PARTSD, XB = Cfg::XB, BUFB = Cfg::BUFB,
                NDX = Cfg::NDX, NDD = Cfg::NDD;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  char* const ldsb = (char*)lds;
  typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef __attribute__((address_space(3))) const char* lcptr_t;
  typedef __attribute__((address_space(3))) const f32x2* lfrag_t;
  typedef __attribute__((address_space(3))) const float* lf32_t;
  const int tid = threadIdx.x, lane = tid & 63, n = lane & 31, hf = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int HCI = p.H * CI, HCO = p.H * CO;
  const unsigned ldsBase = (unsigned)(size_t)(lcptr_t)ldsb;

  int rd = blockIdx.x * p.rpw;
  int rdEnd = rd + p.rpw;
  if (rdEnd > p.nRounds) rdEnd = p.nRounds;

  struct Pos { int b, hb, k; };
  Pos nx;
  {
    const int per = p.hBlocks * p.rps;
    const int q = rd < p.nRounds ? rd : 0;
    nx.b = q / per;
    const int rem = q - nx.b * per;
    nx.hb = rem / p.rps;
    nx.k = rem - nx.hb * p.rps;
  }
  auto advance = [&](Pos& q) {
    if (++q.k == p.rps) { q.k = 0; if (++q.hb == p.hBlocks) { q.hb = 0; ++q.b; } }
  };
  auto vsharp = [](const void* base, unsigned bytes) -> u32x4v {
    const unsigned long long a = (unsigned long long)base;
    return u32x4v{(unsigned)a, (unsigned)(a >> 32) & 0xffffu, bytes, 0x00020000u};
  };
  // one LDS-DMA instruction under the lane mask `mask` (see conv_tds_tz.hpp): 16 bytes per lane, zeros when out of range
  auto dma = [&](const u32x4v& rs, unsigned ldsAddr, int voff, unsigned long long mask) {
    asm volatile("s_mov_b32 m0, %0\n\ts_mov_b64 exec, %3\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b64 exec, -1"
                 ::"s"(ldsAddr), "v"(voff), "s"(rs), "s"(mask) : "memory", "m0");
  };
  struct Stage { u32x4v rx, rd; int bx, bd; };
  auto stage_of = [&](const Pos& q, bool live) -> Stage {
    Stage st;
    st.rx = vsharp(p.x + (size_t)q.b * p.Tin * HCI, live ? (unsigned)(p.Tin * HCI * 4) : 0u);
    st.rd = vsharp(p.dy + (size_t)q.b * p.Tout * HCO, live ? (unsigned)(p.Tout * HCO * 4) : 0u);
    st.bx = ((q.k * RF * SIG - p.padl) * HCI + q.hb * HB * CI) * 4;   // x slab frame f <-> input frame SIG t0 - padl + f
    st.bd = (q.k * RF * HCO + q.hb * HB * CO) * 4;                    // dy slab frame f <-> output frame t0 + f
    return st;
  };
  const int dmaLane = lane * 16;
  auto lanes = [](int nl) -> unsigned long long { return nl >= 64 ? ~0ull : nl <= 0 ? 0ull : (1ull << nl) - 1; };
  // instruction j of a wave's NDX + NDD per round: wave w stages the x frames w, w + 8, ... and the dy frame parts likewise
  auto stage_issue = [&](const Stage& st, int buf, int j) {
    if (j < NDX) {
      const int f = wave + 8 * j;
      const unsigned long long m = (NFX % 8 == 0 || j + 1 < NDX || f < NFX) ? lanes(CPFX) : 0ull;
      dma(st.rx, ldsBase + buf * BUFB + f * (PX * 4), st.bx + f * HCI * 4 + dmaLane, m);
    } else {
      const int q = wave + 8 * (j - NDX);           // (frame, part) = (q / PARTSD, q % PARTSD)
      const int f = PARTSD == 1 ? q : q / PARTSD, part = PARTSD == 1 ? 0 : q - f * PARTSD;
      unsigned long long m = PARTSD == 1 ? lanes(CPFD) : (part == 0 ? lanes(64) : lanes(CPFD - 64));
      if (!((NFD * PARTSD) % 8 == 0 || j + 1 < NDX + NDD || q < NFD * PARTSD)) m = 0ull;
      dma(st.rd, ldsBase + buf * BUFB + XB + f * (PD * 4) + part * 1024, st.bd + f * HCO * 4 + part * 1024 + dmaLane, m);
    }
  };

  // ---- prologue: ones into the x slabs' frame padding, first slabs
  for (int e = tid; e < 2 * NFX * (PX - HB * CI); e += 512) {
    const int buf = e / (NFX * (PX - HB * CI)), r2 = e - buf * (NFX * (PX - HB * CI)), f = r2 / (PX - HB * CI), c = r2 - f * (PX - HB * CI);
    *(float*)(ldsb + buf * BUFB + (f * PX + HB * CI + c) * 4) = 1.f;
  }
  const bool any = rd < rdEnd;
  {
    const Stage s0 = stage_of(nx, any);
#pragma unroll
    for (int j = 0; j < NDX + NDD; ++j) stage_issue(s0, 0, j);
  }
  advance(nx);

  // ---- per-lane addresses.  Pair tile pt, lane n: row pair 32 pt + n = (s, cp); the wave's mel rows 2 w + hf.
  lcptr_t aBase[NPT];
#pragma unroll
  for (int pt = 0; pt < NPT; ++pt) {
    const int pr = 32 * pt + n;
    const int s = pr / C2, cp = pr - s * C2;
    // pairs >= NPR: the ones of frame 0's padding (pair NPR is the bias row; the others are never exported)
    const int off = pr < NPR ? s * PX + (2 * wave + hf) * CI + 2 * cp : HB * CI;
    aBase[pt] = (lcptr_t)ldsb + off * 4;
  }
  const int rr = n < R * CO ? n / CO : 0, co = n < R * CO ? n - (n / CO) * CO : 0;
  const lcptr_t bBase = (lcptr_t)ldsb + XB + (rr * PD + (2 * wave + hf) * CO + co) * 4;

  f32x16 acc[NPT][2];
#pragma unroll
  for (int pt = 0; pt < NPT; ++pt)
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[pt][e][v] = 0.f;

  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

  for (int it = 0; rd < rdEnd; ++rd, ++it) {
    const int cur = it & 1;
    const Stage st = stage_of(nx, rd + 1 < rdEnd);
    advance(nx);
    f32x2 af[2][NPT];
    float bf[2];
    auto load = [&](int slot, int i) {
#pragma unroll
      for (int pt = 0; pt < NPT; ++pt) af[slot][pt] = *(lfrag_t)(aBase[pt] + cur * BUFB + i * (SIG * R * PX * 4));
      bf[slot] = *(lf32_t)(bBase + cur * BUFB + i * (R * PD * 4));
    };
    load(0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < GR; ++i) {
      const int slot = i & 1;
      if (i + 1 < GR) load(slot ^ 1, i + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int pt = 0; pt < NPT; ++pt) {
        acc[pt][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[slot][pt].x, bf[slot], acc[pt][0], 0, 0, 0);
        acc[pt][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[slot][pt].y, bf[slot], acc[pt][1], 0, 0, 0);
        // two staging instructions per group: all of the next round is under way by the middle of this one
        if (pt < 2 && 2 * i + pt < NDX + NDD) stage_issue(st, cur ^ 1, 2 * i + pt);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }

  // ---- the eight waves' accumulators, added in wave order: one pair tile per pass through LDS ([wave][e][v][lane])
  float* dst = partial + (size_t)blockIdx.x * Cfg::ACCF;
#pragma unroll
  for (int pt = 0; pt < NPT; ++pt) {
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int v = 0; v < 16; ++v) lds[((wave * 2 + e) * 16 + v) * 64 + lane] = acc[pt][e][v];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int el = tid + 512 * q;     // (e, v, lane) of the pair tile: 2048 elements
      float t = lds[el];
#pragma unroll
      for (int w = 1; w < 8; ++w) t += lds[w * 2048 + el];
      dst[pt * 2048 + el] = t;
    }
    __syncthreads();
  }
}

// dW[j][ci][co] = sum_{r} D[(j + r, ci)][(r, co)] over the workgroups' partial images (register layout
// [pair tile][channel parity][accumulator register][lane]), dbias[co] = sum_r D[ones row][(r, co)]: 16 outputs per
// workgroup x 64 slices of the partials (every thread's handful of loads in flight together: ONE round trip to L2 / HBM),
// slices added in order.  (16 slices of 16 workgroups, eight loads at a time: 9.6 us, a tenth of the gradient.)
template <int CI, int CO, int R, int GR, int SIG>
__global__ __launch_bounds__(1024) void tds_tzf_reduce_k(const float* __restrict__ partial, int nParts, int kw, float* __restrict__ dw,
                                                        float* __restrict__ dbias) {
  using Cfg = TzfCfg<CI, CO, R, GR, SIG>;
  constexpr int NS = 64, NO = 16, PER = 4, C2 = Cfg::C2, ACCF = Cfg::ACCF;   // PER * NS >= 256 workgroups
  __shared__ float red[NS][NO];
  const int el = threadIdx.x & (NO - 1), sl = threadIdx.x / NO;
  const int o = blockIdx.x * NO + el;       // dW element ((j CI + ci) CO + co), then the CO bias elements
  const int nW = kw * CI * CO;
  float s = 0.f;
  if (o < nW + CO) {
    int idx[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      int pr, e, co;
      if (o < nW) {
        const int j = o / (CI * CO), rem = o - j * CI * CO, ci = rem / CO;
        co = rem - ci * CO;
        pr = (j + SIG * r) * C2 + (ci >> 1);
        e = ci & 1;
      } else {
        co = o - nW;
        pr = Cfg::NPR;
        e = 0;
      }
      const int pt = pr >> 5, row = pr & 31, col = r * CO + co;
      idx[r] = ((pt * 2 + e) * 16 + 4 * (row >> 3) + (row & 3)) * 64 + 32 * ((row >> 2) & 1) + col;
    }
    for (int g0 = sl * PER; g0 < nParts; g0 += NS * PER) {   // (one trip for up to 256 workgroups)
      float t[PER][R];
#pragma unroll
      for (int u = 0; u < PER; ++u)
#pragma unroll
        for (int r = 0; r < R; ++r) t[u][r] = g0 + u < nParts ? partial[(size_t)(g0 + u) * ACCF + idx[r]] : 0.f;
#pragma unroll
      for (int u = 0; u < PER; ++u)
#pragma unroll
        for (int r = 0; r < R; ++r) s += t[u][r];
    }
  }
  red[sl][el] = s;
  __syncthreads();
  if (sl == 0 && o < nW + CO) {
    float t = red[0][el];
#pragma unroll 8
    for (int k = 1; k < NS; ++k) t += red[k][el];
    if (o < nW) dw[o] = t;
    else if (dbias) dbias[o - nW] = t;
  }
}

}  // namespace w2l
