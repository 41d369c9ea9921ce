// conv_tds_tz.hpp -- fourth generation of the TDS time convolution (fl::TDSBlock's Conv2D kw x 1, C -> C channels,
// C = 10 / 14 / 18, kw <= 21, stride 1; recipes/sota/2019/am_arch/am_tds_ctc.arch:7-37, data flow
// recipes/streaming_convnets/inference/inference/module/nn/TDSBlock.cpp:58-70), forward and backward-data:
// the BLOCK-TOEPLITZ form.
//
// What the role-swapped generations (conv_tds_rs.hip, conv_tds_rs3.hpp) pay for: tap GROUPS on the MFMA columns need an
// overlap-add of the partial sums through LDS (1.2-1.6 DS and ~0.9 VALU instructions per 64-cycle MFMA), a time-fastest
// slab that somebody has to transpose into LDS (the mover waves), and a ring / epilogue pipeline around both.  By the
// issue model of tools/micro/mfma_rate.hip every DS instruction costs ~10 and every VALU ~3-5 cycles of matrix-pipe
// time, so those kernels top out at 0.58-0.66 of the fp32 peak on paper and 0.45-0.50 measured.
//
// Here R consecutive OUTPUT FRAMES join the output channel on the MFMA columns and the weights are expanded, once per
// wave, into a block-Toeplitz B operand that lives in registers:
//     D[(i, h)][(r, co)] = sum_{s < S, ci}  X[R i + s - padl][h][ci] * Wt[(s, ci)][(r, co)],   S = R + 20
//     Wt[(s, ci)][(r, co)] = W[s - r][ci][co]   if 0 <= s - r < kw, else 0
//     out[R i + r][h][co] = D[(i, h)][(r, co)]
// rows = 32 (frame group i, mel row h) pairs, K = (s, ci), columns = (r, co):
//     C = 10: R = 3, 30 of 32 columns, K = 230 (116 MFMAs per 32x32 tile: 115 + one padding step)   useful / issued 0.85
//     C = 14: R = 2, 28 of 32 columns, K = 308 (154 MFMAs)                                          0.835
//     C = 18: R = 3, 54 of 64 columns (two column tiles, one per wave of a pair), K = 414 (208)     0.765
// The price is the (R + 20) / 21 longer reduction.  What it buys:
//   * NO overlap-add, no halo rows, no second LDS structure: a tile's 16 accumulator registers ARE 96 / 64 finished
//     outputs (bias is the accumulators' initial value, the backward pass's residual addend likewise);
//   * the slab keeps the GLOBAL layout x[frame][h][c]: staging is a plain buffer_load_dwordx4 ... lds copy of whole
//     frames (one instruction per frame and wave, the buffer range check supplies the zero padding) -- no mover
//     waves, no transposition, no staging registers;
//   * the A operand of row (i, h) is CONTIGUOUS in ci: one ds_read_b64 feeds TWO MFMAs (the two lane halves read the
//     frames s and s + 1, so one instruction covers four k values): 0.5 DS instructions per MFMA, every (s, ci) an
//     immediate offset of one address register; mel rows 10 / 14 / 18 dwords apart and groups R frames apart put the 32
//     lanes of a half on 32 different bank pairs when R * PITCH = 32 (mod 64) dwords (C = 14: frame pitch 240, not 224).
// Machine shape: workgroups of FOUR waves, two per CU (one wave of each on every SIMD: the two run out of phase, so one
// wave's barrier / epilogue / first-fragment latency is the other one's matrix-pipe time); a round = 8 (C = 18: 4) frame
// groups of one 16-mel-row strip; the slab is double-buffered and every round loads its own S - 1 halo frames again
// (from L2: the previous round of the same workgroup touched them), which makes rounds INDEPENDENT work items: the
// flattened (utterance, strip, round) axis is cut into equal contiguous shares, one per workgroup.  One LDS/DMA barrier
// per round.  The sum order of every output is program order over K: deterministic, independent of the grid.
//
// The same kernel runs the strided sub-sampling layers between the TDS stages (`C2 cin cout 21 1 2 1 -1 -1`,
// am_tds_ctc.arch:3, :12, :22): stride SIG in the forward pass is a group step of SIG R input frames and the Toeplitz index
// s - SIG r (S = SIG (R - 1) + kw); their backward-data pass is, per phase of the stride, a stride-1 correlation of dy with
// every SIG-th tap (conv_tds.hip, tds_conv_backward_data) whose outputs are every SIG-th frame of dx: tap step ST, output
// frame step / offset in the store addresses.  CI != CO: the slab / K side counts CI channels, the columns / outputs CO.
#pragma once

namespace w2l {

struct TdsTzP {
  const float* x;     // [B][Tin][H][CI]
  const float* w;     // [kwFull][CI][CO] (forward);  backward-data reads it as w[tapOff + tapStep (kw - 1 - tap)][co][ci]
  const float* bias;  // [CO] or null
  const float* add;   // optional addend with the layout of y (no ReLU then), or null
  float* y;           // [B][ToutFull][H][CO]; output frame u of this launch is frame oOff + oStep u
  int B, Tin, Tout, H, kw, padl;
  int kwFull, tapOff, oOff, oStep, ToutFull;
  int relu, flip;
  int hBlocks;        // H / 16
  int rps;            // rounds per (utterance, strip)
  int nRounds;        // B * hBlocks * rps
  int rpw;            // rounds per workgroup
  long long* dbg;     // probe build: per workgroup 16 slots of wave 0: cycles in (stage issue + set-up, chain, DMA wait, epilogue, barrier),
                      // rounds, HW_ID | XCC_ID << 32, 100 MHz wall clock at (entry, first round, exit)
};

template <int CI, int CO, int R, int NCT, int SIG, int KWM, int ST>
struct TzCfg {
  static constexpr int KW = KWM;                     // taps the instance is built for (a launch may have fewer)
  static constexpr int HB = 16;                      // mel rows of a strip
  static constexpr int S = SIG * (R - 1) + KW;       // input frames one output group reaches
  static constexpr int C2 = CI / 2;                  // channel pairs
  // Two column tiles of R = 3 output frames (C = 18: 54 columns): tile ct holds the frames r = ct, ct + 1 only, so its
  // Toeplitz rows are zero outside the window s in [SIG ct, SIG ct + SW), SW = S - SIG: each wave of the pair walks ITS
  // window (198 MFMAs instead of 208 at C = 18, an even window: no split tail, ten weight registers less)
  static constexpr int SSPLIT = (NCT == 2 && R == 3) ? 1 : 0;
  static constexpr int SW = S - SIG * SSPLIT;        // frames a wave's chain walks
  static constexpr int SP = SW / 2;                  // frame pairs (s, s + 1): lane half hf reads frame s0 + 2 sp + hf
  static constexpr int TAIL = SW % 2;                // SW odd: the last frame's channel pairs are split between the halves
  static constexpr int TR = TAIL ? (C2 + 1) / 2 : 0; // ... TR reads: half 0 pairs [0, TR), half 1 pairs [TR, 2 TR) (the last one may be padding)
  static constexpr int NRD = SP * C2 + TR;           // ds_read_b64 per chain
  static constexpr int NK = 2 * NRD;                 // MFMAs per chain
  static constexpr int RT = 4 / NCT;                 // row tiles of a round (one per wave, or per wave pair)
  static constexpr int GR = 2 * RT;                  // frame groups of a round (a row tile = 2 groups x 16 mel rows)
  static constexpr int RF = GR * R;                  // output frames of a round
  static constexpr int GSTEP = SIG * R;              // input frames between groups
  static constexpr int NF = (GSTEP * (GR - 1) + S + 3) / 4 * 4;   // slab frames of a round (every wave stages NF / 4 of them)
  // frame pitch: GSTEP PITCH = 32 (mod 64) dwords puts the two groups of a lane half on disjoint bank pairs; where two
  // workgroups' double slabs do not fit the CU with that padding, the plain pitch (2-way conflicts on the fragment reads)
  static constexpr int pitch_pick() {
    int p = HB * CI;
    while ((GSTEP * p) % 64 != 32) p += 4;
    return 4 * ((NF * p * 4 + 64 + 1023) / 1024 * 1024) <= 160 * 1024 ? p : HB * CI;
  }
  static constexpr int PITCH = pitch_pick();         // dwords between slab frames
  static constexpr int CPF = HB * CI / 4;            // 16-byte chunks of one frame of the strip
  static constexpr int PARTS = (CPF + 63) / 64;      // LDS-DMA instructions per frame
  static constexpr int BUFB = (NF * PITCH * 4 + 64 + 1023) / 1024 * 1024;   // (+ 64: the padding pair of the last frame's last row)
  static constexpr size_t LDS = 2 * (size_t)BUFB;
  // taps of the zero-padded weight copy in LDS (second slab, prologue only)
  static constexpr int NQF = S + SIG * (R - 1);      // forward
  static constexpr int NQB = ST * (S + R - 1);       // backward-data
  // ... and its tap pitch in dwords: a lane gathers at (frame shift of its column) * pitch + co + immediates, the shift of
  // output frame rr being -SIG rr taps (forward) / +ST rr taps (backward-data); with SIG * pitch = -CO, ST * pitch = +CO
  // (mod 32) the 32 lanes of a half read banks nn mod 32: conflict-free (at the plain pitch CI CO the three output frames'
  // lanes overlapped: 2-3 passes per ds_read_b32, 116-208 of them per lane)
  static constexpr int wp_pick(int mult, int target) {
    int q = CI * CO;
    while ((mult * q) % 32 != target) ++q;
    return q;
  }
  static constexpr int WPF = wp_pick(SIG, (32 - CO % 32) % 32);
  static constexpr int WPB = wp_pick(ST, CO % 32);
  static_assert((SIG % 2 == 1 && ST % 2 == 1) || CO % 2 == 0, "an even tap step needs an even CO for the conflict-free pitch");
  static_assert(CI % 2 == 0 && (HB * CI) % 4 == 0 && (CI * CO) % 4 == 0, "channel pairs, 16-byte chunks");
  static_assert(R * CO <= 32 * NCT, "columns");
  static_assert(2 * LDS <= 160 * 1024, "two workgroups per CU");
  static_assert((2 * (SP - 1) + 1) * PITCH * 4 + CI * 4 < 65536 && (SW - 1) * PITCH * 4 + 2 * TR * 8 < 65536, "ds offset field");
};

// MODE: 0 forward, 1 forward + ReLU, 2 backward-data (tap-flipped transposed weights), 3 backward-data + residual addend.
// DEFER: the epilogue of tile n-1 (bias, ReLU, 16 stores) rides between the MFMAs of tile n's chain (a second accumulator
//   set; C = 18 has no registers for it and needs it least: its chain is 208 MFMAs long).
// ABL (probe library only; results are garbage): 1 one MFMA per chain, 2 no fragment reads, 4 no stores / addend loads, 8 no DMA
template <int CI, int CO, int R, int NCT, int SIG, int KWM, int ST, int MODE, bool DEFER, int ABL>
__global__ __launch_bounds__(256, 2) void tds_conv_tz_k(TdsTzP p) {
  using Cfg = TzCfg<CI, CO, R, NCT, SIG, KWM, ST>;
  constexpr bool FLIP = MODE >= 2, ADD = MODE == 3, RELU = MODE == 1;
  constexpr int HB = Cfg::HB, S = Cfg::S, C2 = Cfg::C2, SP = Cfg::SP, TR = Cfg::TR, NRD = Cfg::NRD, NK = Cfg::NK, RF = Cfg::RF,
                NF = Cfg::NF, PITCH = Cfg::PITCH, CPF = Cfg::CPF, PARTS = Cfg::PARTS, BUFB = Cfg::BUFB, GSTEP = Cfg::GSTEP, SW = Cfg::SW;
  static_assert(!FLIP || SIG == 1, "a backward-data phase is a stride-1 correlation");
  static_assert(NF % 4 == 0, "every wave stages the same number of frames");
  constexpr int NDMA = NF / 4 * PARTS;                 // LDS-DMA instructions per wave and round
  static_assert(2 * NDMA + 2 <= NRD && 34 < NRD, "the chain carries the staging and the deferred epilogue");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  char* const ldsb = (char*)lds;
  typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) const char* lcptr_t;
  typedef __attribute__((address_space(3))) const float* lfptr_t;
  const int tid = threadIdx.x, lane = tid & 63, n = lane & 31, hf = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rt = NCT == 1 ? wave : wave >> 1, ct = NCT == 1 ? 0 : wave & 1;
  const int HCI = p.H * CI, HCO = p.H * CO;
  const unsigned ldsBase = (unsigned)(size_t)(lcptr_t)ldsb;

#ifdef W2L_PROBE
  const long long wEntry = wall_clock64();
#endif
  int rd = blockIdx.x * p.rpw;
  int rdEnd = rd + p.rpw;
  if (rdEnd > p.nRounds) rdEnd = p.nRounds;
  if (rd >= rdEnd) return;

  // where a round lives: utterance, strip, round of the strip (advanced incrementally: no division per round)
  struct Pos { int b, hb, k; };
  Pos nx;
  {
    const int per = p.hBlocks * p.rps;
    nx.b = rd / per;
    const int rem = rd - nx.b * per;
    nx.hb = rem / p.rps;
    nx.k = rem - nx.hb * p.rps;
  }
  auto advance = [&](Pos& q) {
    if (++q.k == p.rps) { q.k = 0; if (++q.hb == p.hBlocks) { q.hb = 0; ++q.b; } }
  };
  // buffer descriptor (raw, 32-bit offsets, range-checked) of `bytes` bytes at `base`, as four scalars for inline asm
  auto vsharp = [](const void* base, unsigned bytes) -> u32x4v {
    const unsigned long long a = (unsigned long long)base;
    return u32x4v{(unsigned)a, (unsigned)(a >> 32) & 0xffffu, bytes, 0x00020000u};
  };
  // One LDS-DMA instruction: lanes [0, nLanes) copy 16 bytes each from rs[voff] to LDS ldsAddr + 16 lane.  As asm so that
  // it can sit BETWEEN the MFMAs of a chain without a branch (the lane mask is an exec write, not a divergent `if`); an
  // out-of-range source arrives as zeros.  (s_mov m0 needs one wait state before the LDS-DMA reads it.)  m0 is declared clobbered;
  // exec is written and restored to ALL LANES, which is only right in wave-uniform control flow: every call site is -- the round
  // loop, its prologue and epilogue branch on kernel arguments and round counters only (never on a lane index).
  auto dma = [&](const u32x4v& rs, unsigned ldsAddr, int voff, int nLanes) {
    if (ABL & 8) return;
    if (nLanes >= 64) {
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(ldsAddr), "v"(voff), "s"(rs) : "memory", "m0");
    } else {
      const unsigned long long mask = (1ull << nLanes) - 1;
      asm volatile("s_mov_b32 m0, %0\n\ts_mov_b64 exec, %3\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b64 exec, -1"
                   ::"s"(ldsAddr), "v"(voff), "s"(rs), "s"(mask) : "memory", "m0");
    }
  };
  // staging of round q into slab `buf`: NDMA instructions per wave (wave w: frames w, w + 4, ...); frames outside the
  // utterance are outside the descriptor's range and arrive as zeros.  `live` false: a zero-length descriptor (nothing read)
  struct Stage { u32x4v rs; int base; };
  auto stage_of = [&](const Pos& q, bool live) -> Stage {
    Stage st;
    st.rs = vsharp(p.x + (size_t)q.b * p.Tin * HCI, live ? (unsigned)(p.Tin * HCI * 4) : 0u);
    st.base = ((q.k * RF * SIG - p.padl) * HCI + q.hb * HB * CI) * 4;
    return st;
  };
  const int dmaLane = lane * 16;
  auto stage_issue = [&](const Stage& st, int buf, int j) {   // instruction j of NDMA
    const int f = wave + 4 * (j / PARTS), part = j % PARTS;
    dma(st.rs, ldsBase + buf * BUFB + f * (PITCH * 4) + part * 1024, st.base + f * HCI * 4 + part * 1024 + dmaLane, CPF - 64 * part);
  };

  // ---- prologue: the weights -> registers -> buffer 1 (padded, transposed), first slab -> buffer 0
  // The weights, [tap][.][.] as they lie in HBM, with P zero taps in front and zeros behind (NQ taps in all: the range check of
  // the descriptor IS the zero padding) are loaded FIRST, 16 bytes per lane and instruction: they come back from L2 while the
  // slab's frames, issued behind them, come from HBM, and the gather below runs under that latency.  (Loads as asm with a
  // counted wait: the compiler does not see the LDS-DMA instructions behind them and would wait for everything.)
  constexpr int NQ = FLIP ? Cfg::NQB : Cfg::NQF;
  constexpr int WP = FLIP ? Cfg::WPB : Cfg::WPF;
  constexpr int CC = CI * CO;
  constexpr int WJ = (NQ * CC / 4 + 255) / 256;        // 16-byte loads per lane
  static_assert(NQ * WP * 4 <= NF * PITCH * 4, "weights fit the frames of the second slab");
  static_assert(NDMA <= 48, "counted wait");
  const int P = FLIP ? ST * (S - p.kw) : SIG * (R - 1);
  typedef float f32x4v __attribute__((ext_vector_type(4)));
  f32x4v wv[WJ];
  {
    const u32x4v rw = vsharp(p.w, (unsigned)(p.kwFull * CC * 4));
#pragma unroll
    for (int j = 0; j < WJ; ++j) {
      const int voff = ((j * 256 + tid) * 4 - P * CC) * 4;
      asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(wv[j]) : "v"(voff), "s"(rw) : "memory");
    }
  }
  {
    const Stage s0 = stage_of(nx, true);
#pragma unroll
    for (int j = 0; j < NDMA; ++j) stage_issue(s0, 0, j);
  }
  Pos here = nx;
  advance(nx);
  // the padding pair of the last frame's last mel row lies behind the slab: its weights are zero, the bytes must not be NaN
  if (tid < 32) *(float*)(ldsb + (tid >> 4) * BUFB + NF * PITCH * 4 + (tid & 15) * 4) = 0.f;
  // ... and with a padded frame pitch it lies in the frame's padding, which the staging never writes (the second slab
  // holds the weights for now: its padding is cleared once they are in registers)
  constexpr int PADW = PITCH - HB * CI;
  auto clear_padding = [&](int buf) {
    if (TR > 0 && PADW > 0)
      for (int e = tid; e < NF * PADW; e += 256) {
        const int f = e / (PADW > 0 ? PADW : 1), c = e - f * PADW;
        *(float*)(ldsb + buf * BUFB + (f * PITCH + HB * CI + c) * 4) = 0.f;
      }
  };
  clear_padding(0);

  // The copy in LDS is [tap][ci][co] at the tap pitch WP for both passes (the backward pass's w[tap][co][ci] is transposed by
  // the scatter); every lane then gathers the block-Toeplitz column it owns -- step 2 (sp C2 + cp) + e is k = (s = 2 sp + hf,
  // ci = 2 cp + e), the tail steps are (s = s0 + SW - 1, ci = 2 (q + hf TR) + e) -- with ds_read_b32 at immediate offsets
  // of ONE address.  (First version: one global gather per register: 116-208 divergent loads per wave at ~32 TCP cycles each
  // = 14-22 us before the first MFMA; profiles/r05_run3_conv_tz_independent_weight_loads.log.)
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");   // the weights have landed (the slab's DMA may still fly)
#pragma unroll
  for (int j = 0; j < WJ; ++j) asm volatile("" : "+v"(wv[j]));   // (uses stay behind the wait)
  {
    float* const wl = (float*)(ldsb + BUFB);
#pragma unroll
    for (int j = 0; j < WJ; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int e = (j * 256 + tid) * 4 + q;             // element of the unpadded [NQ][CC] copy
        if (e < NQ * CC) {
          const int tap = e / CC, r = e - tap * CC;
          int dst = tap * WP + r;
          if (FLIP) { const int c0 = r / CI, c1 = r - c0 * CI; dst = tap * WP + c1 * CO + c0; }
          wl[dst] = wv[j][q];
        }
      }
  }
  const int nn = 32 * ct + n;
  const bool colOk = nn < R * CO;
  const int s0 = Cfg::SSPLIT ? SIG * ct : 0;         // first frame of this wave's window
  // (the padding columns compute something finite nobody stores: any output frame of the tile)
  const int rr = colOk ? nn / CO : (Cfg::SSPLIT ? ct : 0), co = colOk ? nn - (nn / CO) * CO : 0;
  float biasv = 0.f;
  if (p.bias && colOk) biasv = p.bias[co];
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  float bw[NK];
  {
    const lfptr_t wl = (lfptr_t)(ldsb + BUFB);
    // forward: tap (2 sp + hf) - SIG rr is LDS tap (hf - SIG rr + P) + 2 sp;  backward-data: w'[tap] = w[tapOff + ST (kw - 1 - tap)],
    // tap 2 sp + hf - rr is LDS tap tapOff + ST (kw - 1 - hf + rr) + P - 2 ST sp
    const lfptr_t bm = FLIP ? wl + (p.tapOff + ST * (p.kw - 1 - hf - s0 + rr) + P - 2 * ST * (SP - 1)) * WP + co
                            : wl + (s0 + hf - SIG * rr + P) * WP + co;
#pragma unroll
    for (int sp = 0; sp < SP; ++sp)
#pragma unroll
      for (int u = 0; u < CI; ++u) bw[2 * sp * C2 + u] = bm[(FLIP ? 2 * ST * (SP - 1 - sp) : 2 * sp) * WP + u * CO];
    if (TR > 0) {
      const lfptr_t bt = (FLIP ? wl + (p.tapOff + ST * (p.kw - s0 - SW + rr) + P) * WP : wl + (s0 + SW - 1 - SIG * rr + P) * WP) + 2 * hf * TR * CO + co;
#pragma unroll
      for (int u = 0; u < 2 * TR; ++u) {
        float t = bt[u * CO];
        if (2 * TR + u >= CI) t = hf ? 0.f : t;       // ci = 2 hf TR + u >= CI: the padding pair
        bw[2 * SP * C2 + u] = t;
      }
    }
  }

  // ---- per-lane addresses.  Row n of the tile = (group n >> 4, mel row n & 15) of the wave's two groups.
  const int rowOff = (GSTEP * (n >> 4) + 2 * GSTEP * rt) * PITCH + (n & 15) * CI;              // dwords into the slab
  const lcptr_t aMain = (lcptr_t)ldsb + (rowOff + (s0 + hf) * PITCH) * 4;   // half hf reads frame s0 + 2 sp + hf
  const lcptr_t aTail = (lcptr_t)ldsb + (rowOff + s0 * PITCH) * 4 + hf * TR * 8;   // ... and the channel pairs [hf TR, hf TR + TR) of the last frame
  // accumulator v of this lane: row 8 (v >> 2) + 4 hf + (v & 3) -> group v >> 3, mel row 8 ((v >> 2) & 1) + 4 hf + (v & 3);
  // column (rr, co) -> output frame t0 + R (2 rt + group) + rr
  // (output frame u of the launch is frame oOff + oStep u of the tensor)
  const int yLane = colOk ? ((rr + 2 * R * rt) * p.oStep * HCO + 4 * hf * CO + co) * 4 : (int)0x80000000;   // invalid columns: out of every range
  auto vOff = [](int v) { return (8 * ((v >> 2) & 1) + (v & 3)) * CO * 4; };
  const int yBytes = p.ToutFull * HCO * 4;
  // output offsets of the two groups of a round as OPAQUE values: the per-accumulator constants then fold into the
  // instructions' immediate fields (left alone, hipcc re-associates them onto the loop-invariant lane part: 16 registers)
  auto y_offsets = [&](const Pos& q, int (&g)[2]) {
    g[0] = yLane + ((q.k * RF * p.oStep + p.oOff) * HCO + q.hb * HB * CO) * 4;
    g[1] = g[0] + R * p.oStep * HCO * 4;
    asm volatile("" : "+v"(g[0]), "+v"(g[1]));
  };
  auto finish = [&](float v) -> float {               // bias, ReLU (as asm: fmaxf() puts a canonicalising v_max in front;
    if (!FLIP) asm("v_add_f32 %0, %0, %1" : "+v"(v) : "v"(biasv));   //  a vector add keeps a 16-register splat of the bias alive)
    if (RELU) asm("v_max_f32 %0, %0, 0" : "+v"(v));
    return v;
  };

#ifdef W2L_PROBE
  long long cStage = 0, cChain = 0, cWait = 0, cEpi = 0, cBar = 0;
  const long long wLoop = wall_clock64();
#endif
  // every lane has its weights: the second slab may be overwritten; the first slab has landed
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  clear_padding(1);   // (read first in round 1, a barrier away)

  f32x16 accPrev;                 // DEFER: the finished tile of the previous round
  int yPrev[2] = {(int)0x80000000, (int)0x80000000};
  int bPrev = 0;
  // the residual addend of the NEXT round's tile is fetched under the chain where there are registers for a third
  // accumulator set (one column tile); with two column tiles (C = 18) it is loaded at the start of its own chain
  constexpr bool PFA = DEFER && ADD && NCT == 1;
  f32x16 addNext;                 // PFA: the residual addend of the next round's tile
#pragma unroll
  for (int v = 0; v < 16; ++v) { accPrev[v] = 0.f; addNext[v] = 0.f; }
  if (PFA && !(ABL & 4)) {
    int g[2];
    y_offsets(here, g);
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)(p.add + (size_t)here.b * p.ToutFull * HCO), 0, yBytes, 0x00020000);
#pragma unroll
    for (int v = 0; v < 16; ++v) addNext[v] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra, g[v >> 3] + vOff(v), 0, 0));
  }

  for (int it = 0; rd < rdEnd; ++rd, ++it) {
#ifdef W2L_PROBE
    const long long c0 = clock64();
#endif
    const int cur = it & 1;
    const bool more = rd + 1 < rdEnd;
    const Stage st = stage_of(nx, more);
    int yOffG[2];
    y_offsets(here, yOffG);
    const int b = here.b;
    // DEFER: the stores of the previous round's tile go to ITS utterance (a zero-length descriptor in the first round)
    const __amdgpu_buffer_rsrc_t ryP = __builtin_amdgcn_make_buffer_rsrc((void*)(p.y + (size_t)bPrev * p.ToutFull * HCO), 0, it > 0 ? yBytes : 0, 0x00020000);
    // DEFER && ADD: the addend of the NEXT round's tile is fetched under this chain
    int yNext[2] = {0, 0};
    if (PFA) y_offsets(nx, yNext);
    const __amdgpu_buffer_rsrc_t raN = __builtin_amdgcn_make_buffer_rsrc((void*)(p.add + (size_t)nx.b * p.ToutFull * HCO), 0, more ? yBytes : 0, 0x00020000);
    here = nx;
    advance(nx);
    // the accumulators start from the residual addend (backward-data) or from zero (an inline constant of the first MFMA:
    // no registers); the bias joins in the epilogue
    f32x16 acc;
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[v] = 0.f;
    if (ADD && !(ABL & 4)) {
      if (PFA) {
        acc = addNext;
      } else {
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)(p.add + (size_t)b * p.ToutFull * HCO), 0, yBytes, 0x00020000);
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[v] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra, yOffG[v >> 3] + vOff(v), 0, 0));
      }
    }
#ifdef W2L_PROBE
    const long long c1 = clock64();
#endif
    // ---- the chain: NK dependent MFMAs; fragment read d + D is issued in the slots of read d.  The chain's FIRST HALF also
    // carries, one instruction per MFMA slot, the staging of the next round (even d) and the deferred epilogue of the
    // previous tile (odd d): bias, ReLU, store -- and the next tile's addend load.  All vector memory traffic of a round is
    // issued by d = 36, so the s_waitcnt vmcnt(0) behind the chain finds it complete.
    {
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      constexpr int D = 3, RING = 4;
      typedef __attribute__((address_space(3))) const f32x2* lfrag_t;
      const lcptr_t am = aMain + cur * BUFB, at = aTail + cur * BUFB;
      f32x2 ring[RING];
      auto rdfrag = [&](int d) -> f32x2 {
        if (ABL & 2) return f32x2{bw[0], bw[1]};
        if (d < SP * C2) {
          const int sp = d / C2, cp = d - sp * C2;
          return *(lfrag_t)(am + (2 * sp * PITCH + 2 * cp) * 4);
        }
        return *(lfrag_t)(at + ((SW - 1) * PITCH + 2 * (d - SP * C2)) * 4);
      };
#pragma unroll
      for (int d = 0; d < D; ++d) ring[d % RING] = rdfrag(d);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int d = 0; d < NRD; ++d) {
        if (d + D < NRD) ring[(d + D) % RING] = rdfrag(d + D);
        if (!((ABL & 1) && d > 0)) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ring[d % RING].x, bw[2 * d], acc, 0, 0, 0);
        if (d % 2 == 0 && d / 2 < NDMA) stage_issue(st, cur ^ 1, d / 2);
        if (DEFER && d % 2 == 1 && d / 2 < 16 && !(ABL & 4)) {
          const int v = d / 2;
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, finish(accPrev[v])), ryP, yPrev[v >> 3] + vOff(v), 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (DEFER && d == NRD - 1) {
          // the chain's last MFMA writes the finished tile where the next round's epilogue expects it (no 16-register copy)
          accPrev = __builtin_amdgcn_mfma_f32_32x32x2f32(ring[d % RING].y, bw[2 * d + 1], acc, 0, 0, 0);
        } else if (!(ABL & 1)) {
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ring[d % RING].y, bw[2 * d + 1], acc, 0, 0, 0);
        }
        if (PFA && d % 2 == 1 && d / 2 < 16 && !(ABL & 4)) {
          const int v = d / 2;
          addNext[v] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(raN, yNext[v >> 3] + vOff(v), 0, 0));
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#ifdef W2L_PROBE
    const long long c2 = clock64();
#endif
    // the next round's slab has landed; nothing else of this wave is in flight
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef W2L_PROBE
    const long long c3 = clock64();
#endif
    if (DEFER) {
      yPrev[0] = yOffG[0]; yPrev[1] = yOffG[1];
      bPrev = b;
    } else if (!(ABL & 4)) {
      const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)(p.y + (size_t)b * p.ToutFull * HCO), 0, yBytes, 0x00020000);
#pragma unroll
      for (int v = 0; v < 16; ++v)
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, finish(acc[v])), ry, yOffG[v >> 3] + vOff(v), 0, 0);
    }
    if ((ABL & 4) && (DEFER ? accPrev[0] : acc[0]) == 123.456f) p.y[0] = DEFER ? accPrev[5] : acc[5];
#ifdef W2L_PROBE
    const long long c4 = clock64();
#endif
    // every wave is done with slab `cur` (it becomes the round-after-next's target) and the other slab is complete
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#ifdef W2L_PROBE
    const long long c5 = clock64();
    cStage += c1 - c0; cChain += c2 - c1; cWait += c3 - c2; cEpi += c4 - c3; cBar += c5 - c4;
#endif
  }
  if (DEFER && !(ABL & 4)) {   // the last tile
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)(p.y + (size_t)bPrev * p.ToutFull * HCO), 0, yBytes, 0x00020000);
#pragma unroll
    for (int v = 0; v < 16; ++v)
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, finish(accPrev[v])), ry, yPrev[v >> 3] + vOff(v), 0, 0);
  }
#ifdef W2L_PROBE
  if (p.dbg && tid == 0) {
    long long* d = p.dbg + 16 * blockIdx.x;
    d[0] = cStage; d[1] = cChain; d[2] = cWait; d[3] = cEpi; d[4] = cBar; d[5] = rdEnd - blockIdx.x * p.rpw;
    d[6] = (long long)(unsigned)__builtin_amdgcn_s_getreg(63492) | ((long long)(unsigned)__builtin_amdgcn_s_getreg(63508) << 32);
    d[7] = wEntry; d[8] = wLoop; d[9] = wall_clock64();
  }
#endif
}

}  // namespace w2l
