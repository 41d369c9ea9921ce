// conv_tds_rs3.hpp -- third generation of the role-swapped TDS time convolution (see conv_tds_rs.hip for the
// formulation: tap groups on the MFMA column axis, overlap-add through LDS).  Same math, different MACHINE SHAPE.
//
// What was measured on the cooperative kernel (tds_conv_rs_k; profiles/r02_run8_conv_rs_diet_ablation.log): time =
// MFMA time + everything else.  A workgroup's phases (stage the slab, MFMA + overlap-add, epilogue) run in series
// between workgroup barriers, and the two or three co-resident workgroups of a CU, doing identical work, stay in
// lockstep -- their MFMA phases collide on the matrix pipe, their memory phases collide on the LDS, nothing overlaps.
// On top of that every time block recomputes a halo of (G-1)*J rows (T = 750 in blocks of 114 + 14: 84 % useful).
//
// Here ONE workgroup of twelve waves owns a CU and the roles are split by wave, so the overlap is by construction:
//   * waves 0-7  CONSUMERS (two per SIMD): each owns (mel row, output-channel slice) and does nothing but fragment
//     reads, v_mfma_f32_32x32x2_f32 and the ordered read-modify-write overlap-add; while one of a SIMD's two
//     consumers waits for its LDS round trips the other one owns the matrix pipe;
//   * waves 8-11 MOVERS (one per SIMD): global -> registers -> time-fastest slab for tile r+1, the global fetch of
//     tile r+2, and the epilogue (bias / ReLU / addend, float4 stores) of tile r-2, all while the consumers are on
//     tile r-1.  The slab is double-buffered, the overlap-add buffer is a ring of four;
//   * one LDS-only barrier per round (s_waitcnt lgkmcnt(0); s_barrier): global loads and stores stay in flight
//     across it (the compiler's own vmcnt bookkeeping guards the registers).
// And the time axis is STREAMED: a workgroup walks consecutive tiles of one (utterance, mel-row block) strip and the
// partial sums of a tile's last (G-1)*J positions are completed by the next tile (the epilogue of tile k adds the
// tail of tile k-1's buffer), so the halo is paid once per SEGMENT, not once per tile.  The flattened
// (utterance, mel-row block, output frame) axis is cut into equal quotas, one per workgroup; a quota is one to
// three segments of one strip each.
//     C = 10: G = 3, J = 7,  8 mel rows x 1 channel slice,  30 of 32 columns
//     C = 18: G = 7, J = 3,  4 mel rows x 2 slices of 9 output channels, 63 of 64 columns per wave
//     C = 14: G = 2, J = 11, 8 mel rows x 1 slice, 28 of 32 columns, 21 of 22 taps
// LDS layouts: the slab is time-fastest (slab[(mel row, ci)][frame]: a fragment read is 32 consecutive floats, the K
// step an immediate offset); the overlap-add ring is POSITION-major (ring[slot][position][RS], RS = rows padded so
// that the G tap groups of a lane group land in different banks): the epilogue reads and clears whole float4s of one
// frame (one ds_read_b128 + one ds_write_b128 per piece), exactly the shape of the global store.
// The sum order of every output element is fixed by program order (tile order, K order, overlap-add round order):
// run-to-run deterministic, independent of the workgroup count only up to the position of the segment cuts.
#pragma once

namespace w2l {

struct TdsRs3P {
  const float* x;     // [B][Tin][H][C]
  const float* w;     // [kw][C][C]
  const float* bias;  // [C] or null
  const float* add;   // optional addend with the layout of y, or null
  float* y;           // [B][Tout][H][C]
  int B, Tin, Tout, H, kw, padl;
  int relu, accum, flip;
  int hBlocks;        // H / HH
  int quota;          // output frames of the flattened (utterance, mel-row block, frame) axis per workgroup
  int abl;            // probe build only: timing ablations (results are garbage)
  long long* dbg;     // probe build: per workgroup 8 cycle counters (consumer work / wait, mover stage / fetch / epilogue / wait)
};

template <int C, int G, int J, int HH, int CS, int KT>
struct Rs3Cfg {
  static constexpr int CW = C / CS;                  // output channels of one consumer wave
  static constexpr int NCT = (G * CW + 31) / 32;     // column tiles of one consumer wave
  static constexpr int NK = J * C / 2;               // MFMA steps per column tile
  static constexpr int HALO = (G - 1) * J;
  static constexpr int L = 32 * KT;                  // Y rows of a full tile
  static constexpr int NF = L + J - 1;               // slab frames
  static constexpr int FT = NF | 1;                  // slab row stride (floats), odd
  static constexpr int ROWS = HH * C;
  // ring row stride: a multiple of 4 (float4 epilogue) > ROWS (one spare column absorbs the padding MFMA columns), with
  // J*RS mod 32 at least 8 away from 0: the tap groups g, g+1 of one ds instruction are J positions = J*RS floats apart
  static constexpr int rs_pick() {
    int rs = (ROWS + 1 + 3) / 4 * 4;
    while (true) {
      const int m = (J * rs) % 32;
      if (m >= 8 && m <= 24) return rs;
      rs += 4;
    }
  }
  static constexpr int RS = rs_pick();
  static constexpr int Q = ROWS / 4;                 // float4 pieces per frame
  static constexpr int FSTEP = 256 / Q;              // mover thread = (frame chunk f0 < FSTEP, piece q < Q)
  static constexpr int XV = (NF + FSTEP - 1) / FSTEP;
  static constexpr int EV = (L + FSTEP - 1) / FSTEP;
  static constexpr int SLABF = ROWS * FT;
  static constexpr int OUTF = (L + HALO) * RS;
  static constexpr size_t LDS = (size_t)(2 * SLABF + 4 * OUTF) * sizeof(float);
  static_assert(HH * CS == 8, "eight consumer waves");
  static_assert(C % CS == 0 && C % 2 == 0 && ROWS % 4 == 0, "channel split");
  static_assert(HALO < L, "a tile's tail is completed by ONE following tile");
  static_assert(LDS <= 160 * 1024, "LDS");
};

struct Rs3Tile {
  int b, hb;      // strip
  int tau0;       // first Y row of the tile = output frame of overlap-add position HALO
  int uA, uB;     // the segment's output frames [uA, uB)
  int fl;         // kt (32-row tiles) | first << 4 | last << 5 | valid << 6
  __device__ __forceinline__ int kt() const { return fl & 15; }
  __device__ __forceinline__ bool first() const { return fl & 16; }
  __device__ __forceinline__ bool last() const { return fl & 32; }
  __device__ __forceinline__ bool valid() const { return fl & 64; }
};

// walks a workgroup's quota [p, pe) of the flattened output axis: segments (one strip each), tiles of <= KT row tiles
template <int L, int HALO, int KT>
struct Rs3Iter {
  int left, T, hBlocks;     // output frames still to hand out
  int b, hb, u;             // where the next segment starts
  int uA, uB, j, inSeg;
  __device__ __forceinline__ void init(int p0, int pe, int T_, int hBlocks_) {
    T = T_; hBlocks = hBlocks_; left = pe - p0; inSeg = 0; j = 0; uA = uB = 0;
    const int s = p0 / T;
    u = p0 - s * T;
    b = s / hBlocks;
    hb = s - b * hBlocks;
  }
  __device__ __forceinline__ Rs3Tile next() {
    Rs3Tile t{};
    if (!inSeg) {
      if (left <= 0) return t;
      uA = u;
      const int len = (T - uA) < left ? (T - uA) : left;
      uB = uA + len;
      j = 0;
      inSeg = 1;
    }
    t.tau0 = uA + j * L;
    const int rows = uB + HALO - t.tau0;   // Y rows still to do: outputs < uB need rows < uB + HALO
    int kt = (rows + 31) >> 5;
    const bool last = kt <= KT;
    if (!last) kt = KT;
    t.b = b; t.hb = hb; t.uA = uA; t.uB = uB;
    t.fl = kt | (j == 0 ? 16 : 0) | (last ? 32 : 0) | 64;
    if (last) {
      inSeg = 0;
      left -= uB - uA;
      u = uB;
      if (u == T) { u = 0; if (++hb == hBlocks) { hb = 0; ++b; } }
    } else {
      ++j;
    }
    return t;
  }
};

// pairs of k-steps (s = j * C/2 + pp <-> slab row 2 pp + hf, frame offset j) for the fragment prefetch: both reads of a
// pair come out of ONE ds_read2_b32 when their offsets differ by < 256 dwords: channel pairs (2i, 2i+1) of one frame
// offset (2 rows apart), and with C/2 odd the last channel pair of two consecutive frame offsets (1 dword apart)
template <int C, int J>
struct Rs3Prefetch {
  static constexpr int H2 = C / 2, NK = J * H2;
  struct Table { int a[(NK + 1) / 2]; int b[(NK + 1) / 2]; };
  static constexpr Table make() {
    Table t{};
    int n = 0;
    for (int j = 0; j < J; ++j)
      for (int pp = 0; pp + 1 < H2; pp += 2) { t.a[n] = j * H2 + pp; t.b[n] = j * H2 + pp + 1; ++n; }
    if (H2 % 2)
      for (int j = 0; j < J; j += 2) { t.a[n] = j * H2 + H2 - 1; t.b[n] = j + 1 < J ? (j + 1) * H2 + H2 - 1 : -1; ++n; }
    return t;
  }
};

__device__ __forceinline__ void rs3_barrier() {
  // LDS-only barrier: the slab / overlap-add hand-offs are LDS traffic; global loads (next tile's pieces) and stores
  // (epilogue) stay in flight across it
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// ABL: timing ablations as a COMPILE-TIME mask (probe library only; results are garbage): a run-time mask would put a
// branch around every MFMA and measure that instead
template <int C, int G, int J, int HH, int CS, int KT, bool ADD, int ABL>
__global__ __launch_bounds__(768) void tds_conv_rs3_k(TdsRs3P p) {
  using Cfg = Rs3Cfg<C, G, J, HH, CS, KT>;
  constexpr int CW = Cfg::CW, NCT = Cfg::NCT, NK = Cfg::NK, HALO = Cfg::HALO, L = Cfg::L, FT = Cfg::FT, RS = Cfg::RS, ROWS = Cfg::ROWS,
                Q = Cfg::Q, FSTEP = Cfg::FSTEP, XV = Cfg::XV, EV = Cfg::EV, SLABF = Cfg::SLABF, OUTF = Cfg::OUTF;
  using Iter = Rs3Iter<L, HALO, KT>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* const slab0 = lds;               // [2][ROWS][FT]        slab[(hh*C + ci)][frame]
  float* const out0 = lds + 2 * SLABF;    // [4][L + HALO][RS]    ring[position][(hh*C + co)]; position q <-> output tau0 - HALO + q
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, hf = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // 1: one MFMA per column tile, 2: no overlap-add, 4: no epilogue stores, 8: no slab writes, 16: no fetch, 32: no fragment reads, 64: no epilogue
  constexpr int abl = ABL;

  const int total = p.B * p.hBlocks * p.Tout;
  int p0 = blockIdx.x * p.quota;
  if (p0 > total) p0 = total;
  int pe = p0 + p.quota;
  if (pe > total) pe = total;

  if (wave < 8) {
    // ================================================================================================ consumers
    // weights of this wave's column tiles in MFMA B-operand order (global loads first: their latency hides behind the set-up)
    const int hh = CS == 1 ? wave : wave / CS, cs = CS == 1 ? 0 : wave % CS;
    float bw[NCT][NK];
    int ob[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      const int nn = 32 * ct + r, g = nn / CW, co = cs * CW + (nn - g * CW);
#pragma unroll
      for (int s = 0; s < NK; ++s) {
        const int j = s / (C / 2), ci = 2 * (s % (C / 2)) + hf;
        const int tap = g * J + j;
        const bool ok = g < G && tap < p.kw;
        const size_t src = !p.flip ? ((size_t)tap * C + ci) * C + co : ((size_t)(p.kw - 1 - tap) * C + co) * C + ci;
        const float t = p.w[ok ? src : 0];
        bw[ct][s] = ok ? t : 0.f;
      }
      // overlap-add base of this lane's column: ring position HALO - g*J (+ 4 for the upper lane half), column (hh, co);
      // the padding columns go to the spare column ROWS
      ob[ct] = (g < G ? (HALO - g * J) * RS + hh * C + co : ROWS) + 4 * hf * RS;
    }
    for (int e = tid; e < OUTF; e += 512) *(float4*)(out0 + 4 * e) = make_float4(0.f, 0.f, 0.f, 0.f);   // the whole ring: 4 OUTF floats
    int n = 0;
    {
      Iter it;
      it.init(p0, pe, p.Tout, p.hBlocks);
      while (it.next().valid()) ++n;
    }
    const int aoff = (hh * C + hf) * FT + r;

    // A column tile's chain of NK dependent MFMAs keeps the matrix pipe busy for 64 NK cycles, but the wave issues in
    // order: whatever is to run BESIDE the chain has to sit between its MFMAs in program order.  So the overlap-add of a
    // unit (row tile, column tile) is deferred: its accumulators stay in registers (accP) and its read-modify-write
    // rounds are slotted between the MFMAs of the NEXT unit's chain -- round c reads after MFMA c PER, adds and writes
    // PER - 1 MFMAs (>= 128 cycles) later -- together with the fragment reads of the next row tile (two per second MFMA).
    // The last unit of a tile is overlapped with the first chain of the next tile (other ring slot; the epilogue runs
    // three rounds behind), the very last one is drained in the round after the last tile.
    // (tools/micro/mfma_rate.hip: bare chains run at 97-98 % of the MFMA peak with two waves per SIMD; every DS instruction
    // slotted between them costs the pipe ~10 cycles -- the instruction COUNT is what matters, hence ds_read2 / ds_write2.)
    // Register pairs (2m, 2m+1) = two adjacent rows of the tile: where no two rows of a pair can meet at one output
    // (J = 7, 11) the rounds are coloured over PAIRS and a round is ds_read2_b32 / v_pk_add_f32 / ds_write2_b32 per pair.
    constexpr bool PAIR = RsPairRounds<G, J>::ok();
    constexpr auto rounds = RsRounds<G, J>::make();
    constexpr auto prounds = RsPairRounds<G, J>::make();
    constexpr int NR = PAIR ? prounds.n : rounds.n;
    constexpr int PER = NK / NR;
    static_assert(PER >= 2, "overlap-add rounds do not fit between the MFMAs of one chain");
    constexpr int NPF = (NK + 1) / 2;   // fragment prefetch: pairs of k-steps whose slab offsets are < 256 dwords apart (one ds_read2_b32)
    constexpr int U = KT * NCT;   // units of a full tile; unit u accumulates in set u & 1 while the set of unit u-1 is added out
    // (U odd -- one unit per tile at C = 14 -- costs one 16-register copy per tile: see the end of consume())
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x16 accS[2];
#pragma unroll
    for (int q = 0; q < 16; ++q) { accS[0][q] = 0.f; accS[1][q] = 0.f; }
    float* oP = out0 + ROWS + 4 * hf * RS;   // nothing pending yet: zeros into the spare column
    auto aidx = [&](int s) { return (2 * (s % (C / 2))) * FT + s / (C / 2); };
    auto qoff = [&](int q) { return (8 * (q / 4) + (q % 4)) * RS; };
    // one overlap-add round of the pending set P: the reads (phase 0) or the adds + writes (phase 1)
    f32x2 old2[8];
    float old1[16];
    auto rmw = [&](const f32x16& accP, int c, int phase) {
      if (PAIR) {
#pragma unroll
        for (int m = 0; m < 8; ++m)
          if (prounds.color[m] == c) {
            if (phase == 0) {
              old2[m] = f32x2{oP[qoff(2 * m)], oP[qoff(2 * m + 1)]};
            } else {
              f32x2 nv;
              const f32x2 ap = f32x2{accP[2 * m], accP[2 * m + 1]};
              asm("v_pk_add_f32 %0, %1, %2" : "=v"(nv) : "v"(old2[m]), "v"(ap));   // (hipcc scalarises this add otherwise)
              oP[qoff(2 * m)] = nv.x;
              oP[qoff(2 * m + 1)] = nv.y;
            }
          }
      } else {
#pragma unroll
        for (int q = 0; q < 16; ++q)
          if (rounds.color[q] == c) {
            if (phase == 0) old1[q] = oP[qoff(q)];
            else oP[qoff(q)] = old1[q] + accP[q];
          }
      }
    };
    auto consume = [&](const Rs3Tile& d, const float* slab, float* outB) {
      const float* ab = slab + aoff;
      // fragments: with KT > 1 the next row tile's NK values are read during this row tile's last chain (a[kti + 1]);
      // with ONE row tile per round (C = 14: 77 weight registers) they are STREAMED through a ring of RING registers,
      // each value read RING MFMAs before its use
      constexpr bool STREAM = KT == 1;
      constexpr int RING = 16;
      float a[STREAM ? 1 : KT][STREAM ? RING : NK];
#pragma unroll
      for (int s = 0; s < (STREAM ? RING : NK); ++s) a[0][s] = (abl & 32) ? bw[0][s] : ab[aidx(s)];
#pragma unroll
      for (int kti = 0; kti < KT; ++kti) {
        if (kti < d.kt()) {
#pragma unroll
          for (int ct = 0; ct < NCT; ++ct) {
            constexpr int dummy = 0;
            (void)dummy;
            const int u = kti * NCT + ct;
            f32x16& acc = accS[u & 1];
            const f32x16& accP = accS[(u + 1) & 1];
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
            for (int s = 0; s < NK; ++s) {
              if (!((abl & 1) && s > 0)) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[STREAM ? 0 : kti][STREAM ? s % RING : s], bw[ct][s], acc, 0, 0, 0);
              if (!(abl & 2)) {
#pragma unroll
                for (int c = 0; c < NR; ++c) {
                  if (s == c * PER) {
                    asm volatile("" ::: "memory");   // after round c-1's writes (LDS executes a wave's operations in order)
                    rmw(accP, c, 0);
                  }
                  if (s == c * PER + PER - 1) {
                    rmw(accP, c, 1);
                    asm volatile("" ::: "memory");
                  }
                }
              }
              if (STREAM && !(abl & 32) && s % 2 == 1) {   // the two ring slots just used take the values of steps s - 1 + RING, s + RING
                if (s - 1 + RING < NK) a[0][(s - 1) % RING] = ab[aidx(s - 1 + RING)];
                if (s + RING < NK) a[0][s % RING] = ab[aidx(s + RING)];
              }
              if (!STREAM && ct == NCT - 1 && kti + 1 < KT && !(abl & 32) && s % 2 == 0 && s / 2 < NPF) {   // (beyond the tile when kt < KT: never used)
                constexpr auto pf = Rs3Prefetch<C, J>::make();
                const int s1 = pf.a[s / 2], s2 = pf.b[s / 2];
                a[kti + 1 < KT ? kti + 1 : kti][s1] = ab[32 * (kti + 1) + aidx(s1)];
                if (s2 >= 0) a[kti + 1 < KT ? kti + 1 : kti][s2] = ab[32 * (kti + 1) + aidx(s2)];
              }
              __builtin_amdgcn_sched_barrier(0);
            }
            if ((abl & 2) && acc[0] == 123.456f) outB[0] = acc[3];   // (ablation builds: keeps the chain alive)
            oP = outB + ob[ct] + 32 * kti * RS;
          }
        }
      }
      if ((d.kt() * NCT) & 1) accS[1] = accS[0];   // a short tile with an odd number of units: the pending set is expected in set 1
    };
    auto drain = [&]() {
      if (abl & 2) return;
#pragma unroll
      for (int c = 0; c < NR; ++c) {
        asm volatile("" ::: "memory");
        rmw(accS[1], c, 0);
        rmw(accS[1], c, 1);
      }
      asm volatile("" ::: "memory");
    };

    // ---- the round pipeline, consumer side.  Round r: tile r-1 (slab r-1 & 1, ring slot (r-1) % 4) and the deferred
    // overlap-add of tile r-2's last unit (slot (r-2) % 4).  The two roles run the same number of barriers.
    Iter itC;
    itC.init(p0, pe, p.Tout, p.hBlocks);
    rs3_barrier();   // the ring is zero
#ifdef W2L_PROBE
    long long cWork = 0, cWait = 0;
#endif
    for (int rd = 0; rd < n + 3; ++rd) {
#ifdef W2L_PROBE
      const long long t0 = clock64();
#endif
      if (rd >= 1 && rd <= n) {
        const Rs3Tile d = itC.next();
        const int i = rd - 1;
        consume(d, slab0 + (i & 1) * SLABF, out0 + (i & 3) * OUTF);
      } else if (rd == n + 1) {
        drain();
      }
#ifdef W2L_PROBE
      const long long t1 = clock64();
#endif
      rs3_barrier();
#ifdef W2L_PROBE
      const long long t2 = clock64();
      if (rd >= 1 && rd <= n) { cWork += t1 - t0; cWait += t2 - t1; }
#endif
    }
#ifdef W2L_PROBE
    if (p.dbg && tid == 0) { p.dbg[8 * blockIdx.x] = cWork; p.dbg[8 * blockIdx.x + 1] = cWait; p.dbg[8 * blockIdx.x + 6] = n; }
#endif
  } else {
    // ================================================================================================ movers
    // A wave that issues MFMAs back to back keeps the SIMD's issue port: a co-resident wave WITHOUT MFMAs gets one
    // instruction through per MFMA slot at best (tools/micro/mfma_valu.hip: a 36-instruction pass takes 425 cycles alone,
    // 14 800 beside two MFMA waves).  So the movers run at raised priority and their instruction COUNT is what is
    // budgeted: raw buffer loads / stores whose range check does the padding and the clipping (frames outside the
    // utterance load zeros, outputs outside the segment are dropped: no selects, no branches), float4 LDS traffic in
    // the epilogue, packed adds.
    __builtin_amdgcn_s_setprio(3);
    const int mt = tid - 512;
    const int f0 = mt / Q, q4 = 4 * (mt - f0 * Q);
    const bool act = f0 < FSTEP;
    const int fc = act ? f0 : FSTEP - 1;   // the spare threads of the last wave repeat chunk FSTEP-1 (same data, same slab addresses)
    const int HC = p.H * C;
    const int vbase = (fc * HC + q4) * 4;   // byte offset of this thread's piece in frame fc of the workgroup's first mel row
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    auto pk_add = [](f32x2 x, f32x2 y) { f32x2 r; asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; };   // (hipcc scalarises these adds)
    auto vmax = [](float x, float y) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; };
    float4 xr[XV];
#pragma unroll
    for (int v = 0; v < XV; ++v) xr[v] = make_float4(0.f, 0.f, 0.f, 0.f);

    auto fetch = [&](const Rs3Tile& d) {
      // buffer = utterance d.b of x; frames before 0 wrap to offsets >= 2^31, frames >= Tin to offsets >= the size: zeros
      const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (size_t)d.b * p.Tin * HC), 0, p.Tin * HC * 4, 0x00020000);
      const int off = vbase + ((d.tau0 - p.padl) * HC + d.hb * HH * C) * 4;
#pragma unroll
      for (int v = 0; v < XV; ++v)
        xr[v] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rx, off + v * (FSTEP * HC * 4), 0, 0));
    };
    Iter itM;
    itM.init(p0, pe, p.Tout, p.hBlocks);
    Rs3Tile dW = itM.next(), dF = itM.next(), d1{}, d2{}, dE{};   // tiles r, r+1, r-1, r-2, r-3
    if (dW.valid() && !(abl & 16)) fetch(dW);    // first thing: the consumers cannot start before this tile is in the slab
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);   // the four channels (4q + k) mod C of this thread's pieces
    if (p.bias) bias4 = make_float4(p.bias[q4 % C], p.bias[(q4 + 1) % C], p.bias[(q4 + 2) % C], p.bias[(q4 + 3) % C]);
    const float lo = p.relu ? 0.f : -__builtin_inff();
    int n = 0;
    {
      Iter it;
      it.init(p0, pe, p.Tout, p.hBlocks);
      while (it.next().valid()) ++n;
    }

    auto stage = [&](float* slab) {
      float* dst = slab + q4 * FT + fc;
#pragma unroll
      for (int v = 0; v < XV; ++v)
        if (FSTEP * (v + 1) <= FT || fc + FSTEP * v < FT) {   // (frames beyond the tile's are written too: never read)
          dst[FSTEP * v] = xr[v].x; dst[FT + FSTEP * v] = xr[v].y; dst[2 * FT + FSTEP * v] = xr[v].z; dst[3 * FT + FSTEP * v] = xr[v].w;
        }
    };
    auto epilogue = [&](const Rs3Tile& d, float* cur, float* prev) {
      if (!act) return;
      // buffer = the segment's output frames [uA, uB) of utterance d.b: positions that belong to other segments (or to
      // nobody: the first HALO of a segment, the rows past its end) fall outside and the store is dropped
      const size_t seg = ((size_t)d.b * p.Tout + d.uA) * HC;
      const int bytes = (d.uB - d.uA) * HC * 4;
      const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)(p.y + seg), 0, bytes, 0x00020000);
      const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)((ADD ? p.add : p.y) + seg), 0, bytes, 0x00020000);
      const int off = vbase + ((d.tau0 - HALO - d.uA) * HC + d.hb * HH * C) * 4;
      float* s4 = cur + fc * RS + q4;
      float* t4 = prev + (L + fc) * RS + q4;
      const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
      // all the LDS reads first, then the clears, then arithmetic and stores: one LDS round trip per epilogue
      float4 v[EV], w[EV], ad[EV];
#pragma unroll
      for (int it = 0; it < EV; ++it) {
        v[it] = z4; w[it] = z4; ad[it] = z4;
        if (FSTEP * (it + 1) <= L || fc + FSTEP * it < L) v[it] = *(float4*)(s4 + FSTEP * it * RS);
        if (FSTEP * it < HALO && !d.first() && (FSTEP * (it + 1) <= HALO || fc + FSTEP * it < HALO)) w[it] = *(float4*)(t4 + FSTEP * it * RS);   // the previous tile's tail
        if (ADD) ad[it] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ra, off + it * (FSTEP * HC * 4), 0, 0));
      }
#pragma unroll
      for (int it = 0; it < EV; ++it) {
        if (FSTEP * (it + 1) <= L || fc + FSTEP * it < L) *(float4*)(s4 + FSTEP * it * RS) = z4;
        if (FSTEP * it < HALO && !d.first() && (FSTEP * (it + 1) <= HALO || fc + FSTEP * it < HALO)) *(float4*)(t4 + FSTEP * it * RS) = z4;
      }
      if (!(abl & 4)) {
#pragma unroll
        for (int it = 0; it < EV; ++it) {
          f32x2 lo2 = f32x2{v[it].x, v[it].y}, hi2 = f32x2{v[it].z, v[it].w};
          if (FSTEP * it < HALO) { lo2 = pk_add(lo2, f32x2{w[it].x, w[it].y}); hi2 = pk_add(hi2, f32x2{w[it].z, w[it].w}); }
          lo2 = pk_add(lo2, f32x2{bias4.x, bias4.y}); hi2 = pk_add(hi2, f32x2{bias4.z, bias4.w});
          // ReLU (or nothing: lo = -inf); as asm because fmaxf() puts a canonicalising v_max in front of every max
          float4 r4 = make_float4(vmax(lo2.x, lo), vmax(lo2.y, lo), vmax(hi2.x, lo), vmax(hi2.y, lo));
          if (ADD) {
            lo2 = pk_add(f32x2{r4.x, r4.y}, f32x2{ad[it].x, ad[it].y}); hi2 = pk_add(f32x2{r4.z, r4.w}, f32x2{ad[it].z, ad[it].w});
            r4 = make_float4(lo2.x, lo2.y, hi2.x, hi2.y);
          }
          int o = off + it * (FSTEP * HC * 4);
          if (FSTEP * (it + 1) > L) o = fc + FSTEP * it < L ? o : (int)0x80000000;   // positions >= L belong to the next tile: out of range, dropped
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, r4), ry, o, 0, 0);
        }
      }
      if (d.last()) {   // nobody completes the tail of a segment's last tile: clear positions [32 kt, 32 kt + HALO)
        float* c4 = cur + (32 * d.kt() + fc) * RS + q4;
#pragma unroll
        for (int it = 0; it * FSTEP < HALO; ++it)
          if (fc + FSTEP * it < HALO) *(float4*)(c4 + FSTEP * it * RS) = z4;
      }
    };

    // ---- the round pipeline, mover side.  Round r: write the slab of tile r (slab r & 1: tile r-2's readers passed the
    // last barrier), fetch tile r+1, epilogue of tile r-3 (slot (r-3) % 4, plus the tail of slot (r-4) % 4 = r % 4)
    rs3_barrier();   // the ring is zero
#ifdef W2L_PROBE
    long long mS = 0, mF = 0, mE = 0, mWait = 0, mV = 0;
#endif
    for (int rd = 0; rd < n + 3; ++rd) {
#ifdef W2L_PROBE
      const long long tv = clock64();
      if (p.dbg) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const long long t0 = clock64();
      if (rd >= 1 && rd <= n) mV += t0 - tv;
#endif
      if (dW.valid() && !(abl & 8)) stage(slab0 + (rd & 1) * SLABF);
#ifdef W2L_PROBE
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const long long t1 = clock64();
#endif
      if (dF.valid() && !(abl & 16)) fetch(dF);
#ifdef W2L_PROBE
      const long long t2 = clock64();
#endif
      if (dE.valid() && !(abl & 64)) epilogue(dE, out0 + ((rd + 1) & 3) * OUTF, out0 + (rd & 3) * OUTF);
      dE = d2; d2 = d1; d1 = dW; dW = dF; dF = itM.next();
#ifdef W2L_PROBE
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const long long t3 = clock64();
#endif
      rs3_barrier();
#ifdef W2L_PROBE
      const long long t4 = clock64();
      if (rd >= 1 && rd <= n) { mS += t1 - t0; mF += t2 - t1; mE += t3 - t2; mWait += t4 - t3; }
#endif
    }
#ifdef W2L_PROBE
    if (p.dbg && mt == 0) { p.dbg[8 * blockIdx.x + 2] = mS; p.dbg[8 * blockIdx.x + 3] = mF; p.dbg[8 * blockIdx.x + 4] = mE; p.dbg[8 * blockIdx.x + 5] = mWait; p.dbg[8 * blockIdx.x + 7] = mV; }
#endif
  }
}

}  // namespace w2l
