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
#pragma once

namespace w2l {

struct TdsTzP {
  const float* x;     // [B][Tin][H][C]
  const float* w;     // [kw][C][C]
  const float* bias;  // [C] or null
  const float* add;   // optional addend with the layout of y (no ReLU then), or null
  float* y;           // [B][Tout][H][C]
  int B, Tin, Tout, H, kw, padl;
  int relu, flip;
  int hBlocks;        // H / 16
  int rps;            // rounds per (utterance, strip)
  int nRounds;        // B * hBlocks * rps
  int rpw;            // rounds per workgroup
  long long* dbg;     // probe build: per workgroup 16 slots of wave 0: cycles in (stage issue + set-up, chain, DMA wait, epilogue, barrier),
                      // rounds, HW_ID | XCC_ID << 32, 100 MHz wall clock at (entry, first round, exit)
};

template <int C, int R, int NCT>
struct TzCfg {
  static constexpr int KW = 21;
  static constexpr int HB = 16;                      // mel rows of a strip
  static constexpr int S = R + KW - 1;               // frames one output group reaches
  static constexpr int C2 = C / 2;                   // channel pairs
  static constexpr int SP = S / 2;                   // frame pairs (s, s + 1): lane half hf reads frame 2 sp + hf
  static constexpr int TAIL = S % 2;                 // S odd: the last frame's channel pairs are split between the halves
  static constexpr int TR = TAIL ? (C2 + 1) / 2 : 0; // ... TR reads: half 0 pairs [0, TR), half 1 pairs [TR, 2 TR) (the last one may be padding)
  static constexpr int NRD = SP * C2 + TR;           // ds_read_b64 per chain
  static constexpr int NK = 2 * NRD;                 // MFMAs per chain
  static constexpr int RT = 4 / NCT;                 // row tiles of a round (one per wave, or per wave pair)
  static constexpr int GR = 2 * RT;                  // frame groups of a round (a row tile = 2 groups x 16 mel rows)
  static constexpr int RF = GR * R;                  // output frames of a round
  static constexpr int NF = (GR - 1) * R + S;        // slab frames of a round
  static constexpr int pitch_pick() {
    int p = HB * C;
    while ((R * p) % 64 != 32) p += 4;
    return p;
  }
  static constexpr int PITCH = pitch_pick();         // dwords between slab frames
  static constexpr int CPF = HB * C / 4;             // 16-byte chunks of one frame of the strip
  static constexpr int PARTS = (CPF + 63) / 64;      // LDS-DMA instructions per frame
  static constexpr int BUFB = (NF * PITCH * 4 + 64 + 1023) / 1024 * 1024;   // (+ 64: the padding pair of the last frame's last row)
  static constexpr size_t LDS = 2 * (size_t)BUFB;
  static_assert(C % 2 == 0 && (HB * C) % 4 == 0, "channel pairs, 16-byte chunks");
  static_assert(R * C <= 32 * NCT, "columns");
  static_assert(2 * LDS <= 160 * 1024, "two workgroups per CU");
  static_assert((2 * (SP - 1) + 1) * PITCH * 4 + C * 4 < 65536 && (S - 1) * PITCH * 4 + 2 * TR * 8 < 65536, "ds offset field");
};

// ABL (probe library only; results are garbage): 1 one MFMA per chain, 2 no fragment reads, 4 no stores / addend loads, 8 no DMA
template <int C, int R, int NCT, bool ADD, bool FLIP, int ABL>
__global__ __launch_bounds__(256, 2) void tds_conv_tz_k(TdsTzP p) {
  using Cfg = TzCfg<C, R, NCT>;
  constexpr int HB = Cfg::HB, S = Cfg::S, C2 = Cfg::C2, SP = Cfg::SP, TR = Cfg::TR, NRD = Cfg::NRD, NK = Cfg::NK, RF = Cfg::RF,
                NF = Cfg::NF, PITCH = Cfg::PITCH, CPF = Cfg::CPF, PARTS = Cfg::PARTS, BUFB = Cfg::BUFB;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  char* const ldsb = (char*)lds;
  const int tid = threadIdx.x, lane = tid & 63, n = lane & 31, hf = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rt = NCT == 1 ? wave : wave >> 1, ct = NCT == 1 ? 0 : wave & 1;
  const int HC = p.H * C;

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
  // stage the NF frames [t0 - padl, ...) of strip (b, hb) into slab `buf`: wave w issues the frames w, w + 4, ...; frames
  // outside the utterance are outside the buffer's range and arrive as zeros
  auto stage = [&](const Pos& q, char* buf) {
    if (ABL & 8) return;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (size_t)q.b * p.Tin * HC), 0, p.Tin * HC * 4, 0x00020000);
    const int base = ((q.k * RF - p.padl) * HC + q.hb * HB * C) * 4;
#pragma unroll
    for (int j = 0; j < (NF + 3) / 4; ++j) {
      const int f = wave + 4 * j;
      if (f < NF) {
#pragma unroll
        for (int part = 0; part < PARTS; ++part)
          if (lane + 64 * part < CPF)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)(buf + f * (PITCH * 4) + part * 1024), 16,
                                                     base + f * HC * 4 + (lane + 64 * part) * 16, 0, 0, 0);
      }
    }
  };

  stage(nx, ldsb);
  Pos here = nx;
  advance(nx);
  // the padding pair of the last frame's last mel row lies behind the slab: its weights are zero, the bytes must not be NaN
  if (tid < 32) *(float*)(ldsb + (tid >> 4) * BUFB + NF * PITCH * 4 + (tid & 15) * 4) = 0.f;

  // ---- the block-Toeplitz weights of this lane's column (r, co) in MFMA B-operand order: step 2 (sp C2 + cp) + e is
  // k = (s = 2 sp + hf, ci = 2 cp + e); the tail steps are (s = S - 1, ci = 2 (q + hf TR) + e)
  const int nn = 32 * ct + n;
  const int rr = nn / C, co = nn - rr * C;
  const bool colOk = nn < R * C;
  // One buffer load per register, ALL in flight together: an invalid (tap, column) is an out-of-range offset and loads 0,
  // so there is neither a select nor a clamp -- and no wait between the groups (the first version waited for each frame
  // pair's ten loads in turn: twelve round trips to L2 with 2048 waves asking for the same 66 cache lines = 20-25 us
  // before the first MFMA of an 80 us kernel, profiles/r05_run2_conv_tz.log).
  float bw[NK];
  {
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.kw * C * C * 4, 0x00020000);
    constexpr int CISTEP = FLIP ? 1 : C;            // distance of consecutive ci: w[tap][ci][co], flipped w[kw-1-tap][co][ci]
    const int colPart = FLIP ? co * C : co;
    constexpr int OOB = (int)0x80000000;
#pragma unroll
    for (int sp = 0; sp < SP; ++sp) {
      const int tap = 2 * sp + hf - rr;
      const bool ok = colOk && tap >= 0 && tap < p.kw;
      const int off = ok ? ((FLIP ? p.kw - 1 - tap : tap) * C * C + colPart) * 4 : OOB;
#pragma unroll
      for (int u = 0; u < C; ++u)
        bw[2 * sp * C2 + u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rw, off + u * CISTEP * 4, 0, 0));
    }
    if (TR > 0) {
      const int tap = S - 1 - rr;
      const bool ok = colOk && tap >= 0 && tap < p.kw;
      const int base = ((FLIP ? p.kw - 1 - tap : tap) * C * C + colPart) * 4;
#pragma unroll
      for (int u = 0; u < 2 * TR; ++u) {
        const int ci = 2 * hf * TR + u;
        const int off = ok && ci < C ? base + ci * CISTEP * 4 : OOB;
        bw[2 * SP * C2 + u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rw, off, 0, 0));
      }
    }
  }
  float biasv = 0.f;
  if (p.bias && colOk) biasv = p.bias[co];

  // ---- per-lane addresses.  Row n of the tile = (group n >> 4, mel row n & 15) of the wave's two groups.
  const int rowOff = (R * (n >> 4) + 2 * R * rt) * PITCH + (n & 15) * C;                       // dwords into the slab
  typedef __attribute__((address_space(3))) const char* lcptr_t;
  const lcptr_t aMain = (lcptr_t)ldsb + (rowOff + hf * PITCH) * 4;   // half hf reads frame 2 sp + hf
  const lcptr_t aTail = (lcptr_t)ldsb + rowOff * 4 + hf * TR * 8;      // ... and the channel pairs [hf TR, hf TR + TR) of the last frame
  // accumulator v of this lane: row 8 (v >> 2) + 4 hf + (v & 3) -> group v >> 3, mel row 8 ((v >> 2) & 1) + 4 hf + (v & 3);
  // column (rr, co) -> output frame t0 + R (2 rt + group) + rr
  const int yLane = colOk ? ((rr + 2 * R * rt) * HC + 4 * hf * C + co) * 4 : (int)0x80000000;       // invalid columns: out of every range

#ifdef W2L_PROBE
  long long cStage = 0, cChain = 0, cWait = 0, cEpi = 0, cBar = 0;
  const long long wLoop = wall_clock64();
#endif
  // the first slab has to be there
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");

  for (int it = 0; rd < rdEnd; ++rd, ++it) {
#ifdef W2L_PROBE
    const long long c0 = clock64();
#endif
    const int cur = it & 1;
    if (rd + 1 < rdEnd) stage(nx, ldsb + (cur ^ 1) * BUFB);
    const int b = here.b, hb = here.hb, t0 = here.k * RF;
    here = nx;
    advance(nx);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)(p.y + (size_t)b * p.Tout * HC), 0, p.Tout * HC * 4, 0x00020000);
    // the two groups' offsets as opaque per-round values: the per-accumulator constants then fold into the instructions'
    // immediate fields (left alone, hipcc re-associates them onto the loop-invariant lane part: 16 address registers)
    int yOffG[2];
    yOffG[0] = yLane + (t0 * HC + hb * HB * C) * 4;
    yOffG[1] = yOffG[0] + R * HC * 4;
    asm volatile("" : "+v"(yOffG[0]), "+v"(yOffG[1]));
    // the accumulators start from the residual addend (backward-data) or from zero (an inline constant of the first MFMA:
    // no registers); the bias joins in the epilogue
    f32x16 acc;
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[v] = 0.f;
    if (ADD && !(ABL & 4)) {
      const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)(p.add + (size_t)b * p.Tout * HC), 0, p.Tout * HC * 4, 0x00020000);
#pragma unroll
      for (int v = 0; v < 16; ++v)
        acc[v] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra, yOffG[v >> 3] + (8 * ((v >> 2) & 1) + (v & 3)) * C * 4, 0, 0));
    }
#ifdef W2L_PROBE
    const long long c1 = clock64();
#endif
    // ---- the chain: NK dependent MFMAs, fragment read d + D issued in the slots of read d
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
        return *(lfrag_t)(at + ((S - 1) * PITCH + 2 * (d - SP * C2)) * 4);
      };
#pragma unroll
      for (int d = 0; d < D; ++d) ring[d % RING] = rdfrag(d);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int d = 0; d < NRD; ++d) {
        if (d + D < NRD) ring[(d + D) % RING] = rdfrag(d + D);
        if (!((ABL & 1) && d > 0)) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ring[d % RING].x, bw[2 * d], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (!(ABL & 1)) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ring[d % RING].y, bw[2 * d + 1], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#ifdef W2L_PROBE
    const long long c2 = clock64();
#endif
    // the next round's slab (issued a whole chain ago) has landed; nothing else of this wave is in flight
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef W2L_PROBE
    const long long c3 = clock64();
#endif
    if (!(ABL & 4)) {
      if (p.bias) {   // (as asm: hipcc otherwise keeps a 16-register splat of the bias alive through the chain)
#pragma unroll
        for (int v = 0; v < 16; ++v) asm("v_add_f32 %0, %0, %1" : "+v"(acc[v]) : "v"(biasv));
      }
      if (p.relu) {
#pragma unroll
        for (int v = 0; v < 16; ++v) asm("v_max_f32 %0, %0, 0" : "+v"(acc[v]));   // (fmaxf() puts a canonicalising v_max in front)
      }
#pragma unroll
      for (int v = 0; v < 16; ++v)
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (float)acc[v]), ry, yOffG[v >> 3] + (8 * ((v >> 2) & 1) + (v & 3)) * C * 4, 0, 0);
    } else if (acc[0] == 123.456f) {
      p.y[0] = acc[5];
    }
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
