// criterion_fcc_big.hip -- FullConnectionCriterion for LARGE label sets (N > 64; the
// north-star stress shape is T = 1500, N = 9998 word pieces, B = 32), gfx950.
//
// Replaces fl::lib::cuda::FullConnectionCriterion<float> (un-vendored Flashlight; call
// sites recipes/slimIPL/src/Train.cpp:410, :1675).  Math: SURVEY.md App. B.2; CPU
// restatement oracle/criterion_oracle.c.  Same results as the small-N kernel of
// criterion_fcc.hip, different machine mapping:
//
// The transition matrix (4 N^2 = 400 MB at N = 9998) is larger than the 256 MiB Infinity
// Cache and EVERY one of the T dependent steps needs all of it, so the recursion is
// HBM-bound on re-streaming A (T (4 N^2 + 8 B N) bytes).  One step
//     alpha_t[b][i] = x_t[b][i] + LSE_j(alpha_{t-1}[b][j] + A[i][j])
// is evaluated in the exp domain as a skinny GEMM  S[b][i] = sum_j E[b][j] EA[i][j]  with
//     EA[i][j] = exp(A[i][j] - rowmax_i)  in (0, 1]      (packed once per call)
//     E[b][j]  = exp(ahat_{t-1}[b][j]),   ahat = alpha - running max   in (0, 1]
// on v_mfma_f32_32x32x2_f32 (exact fp32 fma chains): the 16 flop/byte of this product sit
// just under the fp32 ridge (157 TF / 6.3 TB/s), i.e. the matrix cores keep up with the
// stream and leave the VALU idle.  EA is stored PRE-PACKED in MFMA operand order
//     EAp[tile of 32 rows][chunk of 8 k][lane][4]  (lane = row%32 + 32*h holds k = 8c+4h+q)
// so a wavefront's global_load_dwordx4 is one fully coalesced 1 KiB burst that lands in
// registers already in fragment layout -- no LDS staging, no transposition, no im2col.
// Per step: kernel 1 (`fcc_big_gemm`: 2 persistent workgroups per CU, each streaming an equal
// contiguous share of the (64-row group x 32-k) stage-units; two register sets in ping-pong so
// a stage's loads stay in flight behind the previous stage's MFMAs, 4-wave LDS reduction,
// deterministic partial slabs per row group) and kernel 2 (`fcc_big_step`, 16 workgroups per
// utterance: adds a row group's slabs in worker order, takes log, adds x_t and rowmax, stores the un-normalised
// a_t = alpha_t - C_{t-1} plain (for backward) and packed (for the next step) plus the
// workgroup's maximum).  The exact per-utterance maximum c_t = max of the 8 partial maxima is
// applied by the NEXT kernel 1 while it builds its operand fragments (E = exp(a_t - c_t): one
// v_exp per MFMA, free beside the matrix pipe), so no step ever waits on a 32-workgroup
// reduction.  A kernel boundary (~1.5-2 us) is cheaper than a grid barrier (4-7 us) here.
//
// Backward streams the transposed pack EATp the same way:
//     dalpha_{t-1}[b][j] = e_{t-1}[b][j] * sum_i r_t[b][i] EA[i][j],  r_t = dalpha_t / s_t
// and the transition gradient is ONE fp32 MFMA GEMM over (t, b) at the end:
//     dA[i][j] = EA[i][j] * sum_{t,b} g_b r_t[b][i] e_{t-1}[b][j]      (2 N^2 T B flop).
#include <cstdlib>

#include "gemm.hpp"

namespace w2l {

constexpr int kBigU = 4;          // chunks (of 8 k) per pipeline stage -> K padded to 32
constexpr int kBigMaxNB = 4;      // utterance tiles of 32 -> B <= 128
constexpr int kBigStepThreads = 1024;  // per-utterance kernels that run once per call
constexpr int kBigParts = 16;          // workgroups (partial maxima) per utterance and step
constexpr int kBigMaxSW = 16;          // bound on partial slabs per row group
constexpr int kBigCachePermille = 500; // product default of BigDims::cache (see there): 84.5 -> 79.4 us per step at T = 300
                                       // (0.595 -> 0.633 of the HBM peak; profiles/r03_run1_fcc_cache_policy.log)

struct BigDims {
  int B, T, N;
  int RT;   // 32-row MFMA tiles per wave (row group = 32 RT rows)
  int Np;   // rows padded to a whole number of row groups
  int Kp;   // reduction length padded to 32
  int NC;   // Kp / 8 chunks
  int NB;   // ceil(B / 32) rounded to 1, 2, 4
  int Bp;   // 32 * NB
  int G;    // Np / (32 RT) row groups
  int nS;   // pipeline stages (of kBigU chunks) per row group
  int U;    // G * nS stage-units of one step
  int W;    // persistent workgroups of the streaming kernel: worker w owns units [U w / W, U (w+1) / W)
  int P;    // bound on the workers (= partial slabs) that share one row group
  int ring; // 1: three operand register sets in a ring, 0: two in ping-pong (W2L_FCC_RING, A/B runs)
  int asmv; // 1: hand-counted asm-load kernel (W2L_FCC_ASM)
  int dma;  // W2L_FCC_DMA: 0 = register ping-pong kernel; 1..5 = LDS-DMA ring (chunks per stage x depth, nontemporal): 1 = 2x3, 2 = 1x6, 3 = 2x3 nt, 4 = 1x6 nt
  int fold; // W2L_FCC_FOLD=1 (probe library): the forward step's log / add-x / rescale epilogue runs inside the streaming kernel
            // (last arriver of each row group) instead of the separate fcc_big_step launch.  Bit-identical, MEASURED SLOWER:
            // 85.3 us per step against 82.4 (profiles/r02_run21_fcc_fold_negative.log), so the product keeps two launches
  int cache; // per mille of every worker's share of the transition stream that is loaded with the DEFAULT cache policy (the
            // rest nontemporal).  A worker reads the same slice at every one of the T steps: the default-policy part can stay
            // resident in the 256 MiB Infinity Cache between steps while the nontemporal rest streams past it (a 400 MB
            // stream loaded entirely with the default policy evicts itself before it is reused).  W2L_FCC_CACHE (probe).
  int abl;  // W2L_FCC_ABL: timing-only ablations 1 = no MFMA, 2 = no E-operand traffic; 4 = nontemporal loads of the
            // transition stream (results stay correct)
  float minSum;  // range check of the scaled-exp recursion (see fcc_big_step): a sum below it flags the utterance for the exact path
};

// persistent workgroups per CU of the streaming kernel: register-limited (three operand stages of
// 48 NB VGPRs + 32 NB accumulators).  W2L_FCC_WPC overrides the NB = 1 choice (1 or 2) for A/B runs.
inline int big_workers_per_cu(int NB) {
  if (NB >= 2) return 1;
  const char* e = tune_env("W2L_FCC_WPC");
  if (e && e[0] >= '1' && e[0] <= '2') return e[0] - '0';
  return 2;
}

// 32-row tiles per wave.  One E fragment (the [utterance][k] operand, re-read by EVERY row group out of
// L2) feeds RT MFMAs: at RT = 2 a third of the kernel's vector-memory instructions were E re-reads and
// the stream sat at the per-CU load-path rate (~11 B/clk/CU) rather than at HBM's; RT = 4 makes it a
// fifth.  MEASURED (MI355X, B=32, N=9998): RT = 4 is SLOWER (109.5 us per step against 90.9 us at RT = 2),
// so the load path is not what bounds the stream; RT = 2 stays the default, W2L_FCC_RT=4 selects the
// wide shape for A/B runs.  NB >= 2 keeps RT = 2 (registers).
inline int big_row_tiles(int NB) {
  if (NB >= 2) return 2;
  const char* e = tune_env("W2L_FCC_RT");
  if (e && e[0] == '4') return 4;
  return 2;
}

inline BigDims big_dims(int B, int T, int N) {
  BigDims d;
  d.B = B; d.T = T; d.N = N;
  d.Kp = (N + 31) / 32 * 32;
  d.NC = d.Kp / 8;
  d.minSum = fmaxf(1e-28f, 1e-30f * (float)N);
  int nb = (B + 31) / 32;
  d.NB = nb <= 1 ? 1 : (nb <= 2 ? 2 : 4);
  d.Bp = 32 * d.NB;
  d.RT = big_row_tiles(d.NB);
  { const char* e = tune_env("W2L_FCC_RING"); d.ring = e ? atoi(e) : 0; }
  { const char* e = tune_env("W2L_FCC_ABL"); d.abl = e ? atoi(e) : 0; }
  { const char* e = tune_env("W2L_FCC_ASM"); d.asmv = e ? atoi(e) : 0; }
  // LDS-DMA ring with a nontemporal transition stream is the default (0 = register ping-pong): measured on MI355X
  // (profiles/r01_run16_fcc_dma_ring_variants.log) 2x3 85.5 us, 1x6 85.8, 2x3 nt 77.6, 1x6 nt 76.0, ping-pong 94.0
  { const char* e = tune_env("W2L_FCC_DMA"); d.dma = e ? atoi(e) : 4; }
  { const char* e = tune_env("W2L_FCC_FOLD"); d.fold = e ? atoi(e) : 0; }
  { const char* e = tune_env("W2L_FCC_CACHE"); d.cache = e ? atoi(e) : kBigCachePermille; if (d.cache < 0) d.cache = 0; if (d.cache > 1000) d.cache = 1000; }
  d.Np = (N + 32 * d.RT - 1) / (32 * d.RT) * (32 * d.RT);
  d.G = d.Np / (32 * d.RT);
  // The step is cut into U = G * nS stage-units (64 rows x 32 k) in row-group-major order and dealt
  // to W PERSISTENT workgroups in equal contiguous ranges: 2 workgroups per CU at B <= 32 (register-
  // limited, big_workers_per_cu) all resident at once, so the stream has no partially filled last
  // round (the old (row group x K split) grid of 1256 workgroups ran 2.45 rounds on 512 slots).  A
  // range touches at most two row groups when U / W <= nS; every row group is then shared by at
  // most P workers, each of which leaves one partial slab for the step kernel to add in worker order.
  d.nS = d.NC / kBigU;
  d.U = d.G * d.nS;
  int w = 256 * big_workers_per_cu(d.NB);
  if (w > d.U / 16) w = d.U / 16;                     // >= four stages per wave
  const int perMin = (d.nS + 13) / 14;                // keeps P <= 16
  if (w > d.U / perMin) w = d.U / perMin;
  if (w < 1) w = 1;
  d.W = w;
  const int per = d.U / d.W;                          // floor: the shortest range
  d.P = (d.nS + per - 1) / per + 1;
  if (d.P > kBigMaxSW) d.P = kBigMaxSW;
  return d;
}

__host__ __device__ inline int big_unit_begin(const BigDims& d, int w) { return (int)((long long)d.U * w / d.W); }
// the worker whose range holds stage-unit u
__host__ __device__ inline int big_worker_of(const BigDims& d, int u) {
  int w = (int)((long long)u * d.W / d.U);
  while (w + 1 < d.W && big_unit_begin(d, w + 1) <= u) ++w;
  while (w > 0 && big_unit_begin(d, w) > u) --w;
  return w;
}
// number of workers (partial slabs) of row group g
__host__ __device__ inline int big_pieces(const BigDims& d, int g) {
  return big_worker_of(d, (g + 1) * d.nS - 1) - big_worker_of(d, g * d.nS) + 1;
}

struct BigWs {
  float* rm;      // [Np] row maxima of A
  float* pack;    // [Np * Kp] EAp (forward) / EATp (backward)
  float* ep[2];   // [Bp * Kp] packed E / R operand, double-buffered over t
  float* part;    // [P][Bp][Np] partial sums of one step (slab = ordinal of the worker inside its row group)
  float* e;       // [T][B][N]  a_t = alpha_t - C_{t-1} (forward); overwritten by e_t = exp(a_t - c_t) in backward
  float* pmax;    // [T][B][kBigParts] partial maxima of a_t
  float* invs;    // [T][B][N]  1 / s_t
  float* rg;      // [T][B][N]  g_b * r_t (backward)
  double* cacc;   // [B] running sum of the per-step maxima
  float* scale;   // [B]
  float* gb;      // [B] scale * upstream grad
  unsigned* cnt;  // [G] arrival tickets of the folded step (self-resetting; zeroed once per call)
  float* cfin;    // [T][B] c_t = the maximum of a_t (of its kBigParts partial maxima), written by fcc_big_loss for the backward pass
  int* redo;      // [B + 1] range check: [b] = 1: some sum of utterance b fell under BigDims::minSum -> the exact log-domain kernels
                  //         (fcc_big_exact_*) recompute it; [B] = 1: at least one utterance is flagged
  size_t bytes;
};

__host__ __device__ inline BigWs big_ws(void* ws, const BigDims& d) {
  BigWs w;
  char* p = (char*)ws;
  auto take = [&](size_t bytes) { char* q = p; p += align_up(bytes, 256); return q; };
  const size_t btn = (size_t)d.T * d.B * d.N * sizeof(float);
  w.rm = (float*)take((size_t)d.Np * sizeof(float));
  w.pack = (float*)take((size_t)d.Np * d.Kp * sizeof(float));
  w.ep[0] = (float*)take((size_t)d.Bp * d.Kp * sizeof(float));
  w.ep[1] = (float*)take((size_t)d.Bp * d.Kp * sizeof(float));
  w.part = (float*)take((size_t)d.P * d.Bp * d.Np * sizeof(float));
  w.e = (float*)take(btn);
  w.pmax = (float*)take((size_t)d.T * d.B * kBigParts * sizeof(float));
  w.invs = (float*)take(btn);
  w.rg = (float*)take(btn);
  w.cacc = (double*)take((size_t)d.B * sizeof(double));
  w.scale = (float*)take((size_t)d.B * sizeof(float));
  w.gb = (float*)take((size_t)d.B * sizeof(float));
  w.cnt = (unsigned*)take((size_t)d.G * sizeof(unsigned));
  w.cfin = (float*)take((size_t)d.T * d.B * sizeof(float));
  w.redo = (int*)take((size_t)(d.B + 1) * sizeof(int));
  w.bytes = (size_t)(p - (char*)ws);
  return w;
}

// ------------------------------------------------------------------ packing (once per call)
__global__ __launch_bounds__(256) void big_rowmax_k(int N, int Np, const float* __restrict__ A, float* __restrict__ rm) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= Np) return;
  float m = -INFINITY;
  if (row < N)
    for (int j = lane; j < N; j += 64) m = fmaxf(m, A[(size_t)row * N + j]);
  m = wave_max(m);
  if (lane == 0) rm[row] = row < N ? m : 0.f;
}

// TRANSPOSED = false: pack[tile][chunk][lane][q] = EA[32 tile + r][8 chunk + 4 h + q]
// TRANSPOSED = true : pack[tile][chunk][lane][q] = EA[8 chunk + 4 h + q][32 tile + r]
// zero outside N x N.  One float4 per thread.
template <bool TRANSPOSED>
__global__ __launch_bounds__(256) void big_pack_k(int N, int NC, size_t total4, const float* __restrict__ A,
                                                  const float* __restrict__ rm, float4* __restrict__ pack) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total4) return;
  const int lane = (int)(idx & 63);
  const size_t tc = idx >> 6;
  const int chunk = (int)(tc % NC);
  const int tile = (int)(tc / NC);
  const int r = 32 * tile + (lane & 31);
  const int k0 = 8 * chunk + 4 * (lane >> 5);
  float v[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int k = k0 + q;
    const int i = TRANSPOSED ? k : r, j = TRANSPOSED ? r : k;
    v[q] = (i < N && j < N) ? __expf(A[(size_t)i * N + j] - rm[i]) : 0.f;
  }
  pack[idx] = make_float4(v[0], v[1], v[2], v[3]);
}

// ------------------------------------------------------------------ kernel 1: the streaming GEMM
// part[sw][b][r] = sum_{k in split} f(op[b][k]) * pack[r][k]   for the 64 rows of group g,
// f = exp(. - c_b) in the forward recursion (EXPOP), identity in the backward one.
template <int NB, int RT>
struct BigStage {
  float4 a[RT][kBigU], e[NB][kBigU];
};

template <int NB, int RT, int ABL = 0>
__device__ __forceinline__ void big_load_stage(BigStage<NB, RT>& st, const float4* __restrict__ pa, size_t tileStride,
                                               const float4* __restrict__ pe, int NC, int s) {
#pragma unroll
  for (int u = 0; u < kBigU; ++u) {
    const size_t c = (size_t)s * kBigU + u;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      if (ABL & 4) {
        const f32x4 t = __builtin_nontemporal_load((const f32x4*)(pa + rt * tileStride + c * 64));
        st.a[rt][u] = make_float4(t[0], t[1], t[2], t[3]);
      } else {
        st.a[rt][u] = pa[rt * tileStride + c * 64];
      }
    }
#pragma unroll
    for (int bt = 0; bt < NB; ++bt) st.e[bt][u] = pe[((size_t)bt * NC + ((ABL & 2) ? 0 : c)) * 64];
  }
}

template <int NB, int RT, bool EXPOP, int ABL = 0>
__device__ __forceinline__ void big_compute_stage(const BigStage<NB, RT>& st, const float (&cb)[NB], f32x16 (&acc)[NB][RT]) {
#pragma unroll
  for (int u = 0; u < kBigU; ++u) {
#pragma unroll
    for (int bt = 0; bt < NB; ++bt) {
      float ev[4] = {st.e[bt][u].x, st.e[bt][u].y, st.e[bt][u].z, st.e[bt][u].w};
      if (EXPOP) {
#pragma unroll
        for (int q = 0; q < 4; ++q) ev[q] = __expf(ev[q] - cb[bt]);  // padding holds -inf -> 0
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const float av = q == 0 ? st.a[rt][u].x : (q == 1 ? st.a[rt][u].y : (q == 2 ? st.a[rt][u].z : st.a[rt][u].w));
          if (ABL & 1) acc[bt][rt][q] += ev[q] * av;  // timing ablation: no matrix pipe
          else acc[bt][rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(ev[q], av, acc[bt][rt], 0, 0, 0);
        }
    }
  }
}

template <int NB, int RT, bool EXPOP, int ABL = 0>
__global__ __launch_bounds__(256, NB <= 1 ? 2 : 1) void fcc_big_gemm(const float4* __restrict__ pack, const float4* __restrict__ op,
                                                       const float* __restrict__ pmax, float* __restrict__ part, BigDims d) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // [4 waves][NB*RT*16 regs][64 lanes]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int NC = d.NC, Np = d.Np, Bp = d.Bp, B = d.B;
  const int w = blockIdx.x;
  int u0 = big_unit_begin(d, w);
  const int u1 = big_unit_begin(d, w + 1);

  // this lane's utterance(s): b = 32 bt + (lane & 31); c_b = exact maximum of a_{t-1}[b][:]
  float cb[NB];
#pragma unroll
  for (int bt = 0; bt < NB; ++bt) {
    cb[bt] = 0.f;
    if (EXPOP) {
      const int b = 32 * bt + (lane & 31);
      float m = -INFINITY;
      if (b < B) {
#pragma unroll
        for (int p = 0; p < kBigParts; ++p) m = fmaxf(m, pmax[(size_t)b * kBigParts + p]);
      } else {
        m = 0.f;
      }
      cb[bt] = m;
    }
  }
  const float4* pe = op + lane;

  while (u0 < u1) {
    // segment: the part of this worker's range inside row group g, split over the 4 waves
    const int g = u0 / d.nS, sb = u0 - g * d.nS;
    int len = d.nS - sb;
    if (len > u1 - u0) len = u1 - u0;
    const int s0 = sb + (int)((long long)len * wave / 4), s1 = sb + (int)((long long)len * (wave + 1) / 4);
    const int piece = w - big_worker_of(d, g * d.nS);

    f32x16 acc[NB][RT];
#pragma unroll
    for (int bt = 0; bt < NB; ++bt)
#pragma unroll
      for (int h = 0; h < RT; ++h)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[bt][h][r] = 0.f;

    const float4* pa = pack + ((size_t)(RT * g) * NC) * 64 + lane;  // row tile rt of the group: + rt * tileStride
    const size_t tileStride = (size_t)NC * 64;

    // Two operand register sets in ping-pong: the loads of stage s+1 are in flight behind the MFMAs of stage s.
    // Variants measured on MI355X (B=32, N=9998; profiles/r01_run9_fcc_stream_ab.log, r01_run10_fcc_stream_variants.log):
    //   this loop (loads under `if`)            90.9 us per step  (4.43 TB/s)
    //   clamped unconditional loads             104.5 us (two redundant tail stages per wave = +8 % traffic)
    //   three-deep ring, unconditional loads     94.1 us;  RT = 4 row tiles per wave 109.5 us
    //   peeled tails (no branch, no redundancy) 138 / 195 us -- hipcc's s_waitcnt placement got worse, not better
    //   loads only (no MFMA, no E traffic)       80.7 us = 5.0 TB/s: the ceiling of this load structure
    if (d.ring == 0 || NB >= 4 || RT >= 4) {
      BigStage<NB, RT> sa, sb2;
      if (s0 < s1) big_load_stage<NB, RT, ABL>(sa, pa, tileStride, pe, NC, s0);
      for (int s = s0; s < s1; s += 2) {
        if (s + 1 < s1) big_load_stage<NB, RT, ABL>(sb2, pa, tileStride, pe, NC, s + 1);
        big_compute_stage<NB, RT, EXPOP, ABL>(sa, cb, acc);
        if (s + 2 < s1) big_load_stage<NB, RT, ABL>(sa, pa, tileStride, pe, NC, s + 2);
        if (s + 1 < s1) big_compute_stage<NB, RT, EXPOP, ABL>(sb2, cb, acc);
      }
    } else if constexpr (NB < 4 && RT < 4) {
      // W2L_FCC_RING=1 / 2: three / two register sets, loads issued UNCONDITIONALLY so that hipcc can count
      // them (vmcnt(N) > 0 in the loop); the stage index of a load past the wave's range falls back to the
      // wave's FIRST stage (just read: an L2 hit, no extra HBM traffic; its data is never consumed).
      if (s0 < s1) {
        auto sx = [&](int t) { return t < s1 ? t : s0; };
        if (d.ring == 1) {
          BigStage<NB, RT> r0, r1, r2;
          big_load_stage<NB, RT, ABL>(r0, pa, tileStride, pe, NC, s0);
          big_load_stage<NB, RT, ABL>(r1, pa, tileStride, pe, NC, sx(s0 + 1));
          for (int s = s0; s < s1; s += 3) {
            big_load_stage<NB, RT, ABL>(r2, pa, tileStride, pe, NC, sx(s + 2));
            big_compute_stage<NB, RT, EXPOP, ABL>(r0, cb, acc);
            big_load_stage<NB, RT, ABL>(r0, pa, tileStride, pe, NC, sx(s + 3));
            if (s + 1 < s1) big_compute_stage<NB, RT, EXPOP, ABL>(r1, cb, acc);
            big_load_stage<NB, RT, ABL>(r1, pa, tileStride, pe, NC, sx(s + 4));
            if (s + 2 < s1) big_compute_stage<NB, RT, EXPOP, ABL>(r2, cb, acc);
          }
        } else {
          BigStage<NB, RT> sa, sb2;
          big_load_stage<NB, RT, ABL>(sa, pa, tileStride, pe, NC, s0);
          for (int s = s0; s < s1; s += 2) {
            big_load_stage<NB, RT, ABL>(sb2, pa, tileStride, pe, NC, sx(s + 1));
            big_compute_stage<NB, RT, EXPOP, ABL>(sa, cb, acc);
            big_load_stage<NB, RT, ABL>(sa, pa, tileStride, pe, NC, sx(s + 2));
            if (s + 1 < s1) big_compute_stage<NB, RT, EXPOP, ABL>(sb2, cb, acc);
          }
        }
      }
    }

    // 4-wave reduction through LDS, fixed order (deterministic)
    constexpr int NR = NB * RT * 16;
#pragma unroll
    for (int bt = 0; bt < NB; ++bt)
#pragma unroll
      for (int h = 0; h < RT; ++h)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((wave * NR) + (bt * RT + h) * 16 + r) * 64 + lane] = acc[bt][h][r];
    __syncthreads();
    // D layout of the 32x32 MFMA: column (row of EA) = lane & 31, row (utterance) = (r&3) + 8 (r>>2) + 4 (lane>>5)
    float* dst = part + (size_t)piece * Bp * Np;
    for (int o = threadIdx.x; o < NR * 64; o += 256) {
      const float v = (red[o] + red[NR * 64 + o]) + (red[2 * NR * 64 + o] + red[3 * NR * 64 + o]);
      const int l = o & 63, rr = o >> 6;
      const int r = rr & 15, h = (rr >> 4) % RT, bt = (rr >> 4) / RT;
      const int b = 32 * bt + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
      const int i = 32 * RT * g + 32 * h + (l & 31);
      dst[(size_t)b * Np + i] = v;
    }
    u0 += len;
    if (u0 < u1) __syncthreads();  // `red` is reused by the next segment
  }
}


// ------------------------------------------------------------------ kernel 1, hand-counted variant (B <= 32)
// Same partition, fragments and arithmetic as fcc_big_gemm<1, 2>, but the operand stream is issued by
// inline-asm loads that hipcc does not count, in a THREE-deep register ring with hand-placed
// s_waitcnt vmcnt(N): the compiler's own placement drains the queue (vmcnt(0)) somewhere in every stage
// whatever the loop shape (see the variant table in fcc_big_gemm), so a wave never had more than one
// stage in flight behind its MFMAs.  Discipline (cdna_hip_programming.md 5.7, form (ii)): every load is an
// "=v" asm output; before a chunk's first consumer a wait statement names that chunk's three destinations
// "+v"; loads retire in issue order, so vmcnt(N) with N = loads issued after the chunk's is exact.
struct AsmStage {
  f32x4 a0[kBigU], a1[kBigU], e[kBigU];
};

#define W2L_GLOAD(dst, ptr) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(ptr) : "memory")

__device__ __forceinline__ void asm_load_stage(AsmStage& st, const float4* pa0, const float4* pa1, const float4* pe, int s) {
#pragma unroll
  for (int u = 0; u < kBigU; ++u) {
    const size_t c = ((size_t)s * kBigU + u) * 64;
    W2L_GLOAD(st.a0[u], pa0 + c);
    W2L_GLOAD(st.a1[u], pa1 + c);
    W2L_GLOAD(st.e[u], pe + c);
  }
}

// AFTER = loads issued after this stage's 12 (0, 12 or 24); chunk u may be consumed once at most
// AFTER + 3 (3 - u) loads are still outstanding
template <bool EXPOP, int AFTER>
__device__ __forceinline__ void asm_compute_stage(AsmStage& st, float cb, f32x16 (&acc)[2]) {
#pragma unroll
  for (int u = 0; u < kBigU; ++u) {
    asm volatile("s_waitcnt vmcnt(%3)" : "+v"(st.a0[u]), "+v"(st.a1[u]), "+v"(st.e[u]) : "n"(AFTER + 3 * (kBigU - 1 - u)));
    float ev[4] = {st.e[u][0], st.e[u][1], st.e[u][2], st.e[u][3]};
    if (EXPOP) {
#pragma unroll
      for (int q = 0; q < 4; ++q) ev[q] = __expf(ev[q] - cb);  // padding holds -inf -> 0
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ev[q], st.a0[u][q], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ev[q], st.a1[u][q], acc[1], 0, 0, 0);
    }
  }
}

template <bool EXPOP>
__global__ __launch_bounds__(256, 2) void fcc_big_gemm_asm(const float4* __restrict__ pack, const float4* __restrict__ op,
                                                           const float* __restrict__ pmax, float* __restrict__ part, BigDims d) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // [4 waves][32 regs][64 lanes]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int NC = d.NC, Np = d.Np, Bp = d.Bp, B = d.B;
  const int w = blockIdx.x;
  int u0 = big_unit_begin(d, w);
  const int u1 = big_unit_begin(d, w + 1);

  float cb = 0.f;
  if (EXPOP) {
    const int b = lane & 31;
    float m = 0.f;
    if (b < B) {
      m = -INFINITY;
#pragma unroll
      for (int p = 0; p < kBigParts; ++p) m = fmaxf(m, pmax[(size_t)b * kBigParts + p]);
    }
    cb = m;
  }
  const float4* pe = op + lane;

  while (u0 < u1) {
    const int g = u0 / d.nS, sb = u0 - g * d.nS;
    int len = d.nS - sb;
    if (len > u1 - u0) len = u1 - u0;
    const int s0 = sb + (int)((long long)len * wave / 4), s1 = sb + (int)((long long)len * (wave + 1) / 4);
    const int piece = w - big_worker_of(d, g * d.nS);

    f32x16 acc[2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[h][r] = 0.f;
    const float4* pa0 = pack + ((size_t)(2 * g) * NC) * 64 + lane;
    const float4* pa1 = pack + ((size_t)(2 * g + 1) * NC) * 64 + lane;

    AsmStage r0, r1, r2;
    int s = s0;
    const int n = s1 - s0;
    if (n == 1) {
      asm_load_stage(r0, pa0, pa1, pe, s);
      asm_compute_stage<EXPOP, 0>(r0, cb, acc);
    } else if (n >= 2) {
      asm_load_stage(r0, pa0, pa1, pe, s);
      asm_load_stage(r1, pa0, pa1, pe, s + 1);
      // invariant: r0 = stage s, r1 = stage s + 1, both issued, nothing else outstanding
      for (; s + 4 < s1; s += 3) {
        asm_load_stage(r2, pa0, pa1, pe, s + 2);
        asm_compute_stage<EXPOP, 24>(r0, cb, acc);
        asm_load_stage(r0, pa0, pa1, pe, s + 3);
        asm_compute_stage<EXPOP, 24>(r1, cb, acc);
        asm_load_stage(r1, pa0, pa1, pe, s + 4);
        asm_compute_stage<EXPOP, 24>(r2, cb, acc);
      }
      const int left = s1 - s;  // 2, 3 or 4
      if (left == 2) {
        asm_compute_stage<EXPOP, 12>(r0, cb, acc);
        asm_compute_stage<EXPOP, 0>(r1, cb, acc);
      } else if (left == 3) {
        asm_load_stage(r2, pa0, pa1, pe, s + 2);
        asm_compute_stage<EXPOP, 24>(r0, cb, acc);
        asm_compute_stage<EXPOP, 12>(r1, cb, acc);
        asm_compute_stage<EXPOP, 0>(r2, cb, acc);
      } else {
        asm_load_stage(r2, pa0, pa1, pe, s + 2);
        asm_compute_stage<EXPOP, 24>(r0, cb, acc);
        asm_load_stage(r0, pa0, pa1, pe, s + 3);
        asm_compute_stage<EXPOP, 24>(r1, cb, acc);
        asm_compute_stage<EXPOP, 12>(r2, cb, acc);
        asm_compute_stage<EXPOP, 0>(r0, cb, acc);
      }
    }

    // 4-wave reduction through LDS, fixed order (deterministic)
    constexpr int NR = 32;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int r = 0; r < 16; ++r) red[((wave * NR) + h * 16 + r) * 64 + lane] = acc[h][r];
    __syncthreads();
    float* dst = part + (size_t)piece * Bp * Np;
    for (int o = threadIdx.x; o < NR * 64; o += 256) {
      const float v = (red[o] + red[NR * 64 + o]) + (red[2 * NR * 64 + o] + red[3 * NR * 64 + o]);
      const int l = o & 63, rr = o >> 6;
      const int r = rr & 15, h = rr >> 4;
      const int b = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
      const int i = 64 * g + 32 * h + (l & 31);
      dst[(size_t)b * Np + i] = v;
    }
    u0 += len;
    if (u0 < u1) __syncthreads();  // `red` is reused by the next segment
  }
}
#undef W2L_GLOAD


// ------------------------------------------------------------------ kernel 1, LDS-DMA ring variant (B <= 32)
// Same partition, packing and arithmetic as fcc_big_gemm<1, 2>; the operand stream goes global -> LDS by
// buffer_load_dwordx4 ... lds (no VGPR destination, so nothing for hipcc's s_waitcnt placement to drain) into
// a WAVE-PRIVATE ring of three stages of 2 chunks (6 KiB per stage: a0, a1, e), read back with lane-linear
// ds_read_b128 (the pack order IS the fragment order).  The wave issues stage s+2, waits with a counted
// vmcnt(12) until stage s has landed (its own covering vmcnt is all that orders a ds_read behind its own
// LDS-DMA), multiplies stage s.  No barrier in the loop; stages past the wave's range re-read its first
// stage (an L2 hit) so that the count stays uniform.
constexpr int kDmaWaveFloats = 3 * 2 * 256 * 3;  // ring of one wave: U x D = 6 chunks of (a0, a1, e) pieces = 18 KiB

// The forward step's second kernel folded into the stream (FOLD): the LAST worker to deliver a row group's partial slab
// (one arrival ticket per row group) adds the slabs in worker order, takes the log, adds x_t and the row maximum, and
// stores a_t plain and packed, 1 / s_t and -- by an atomic max into one of the kBigParts cells of each utterance -- the
// running maximum the next step rescales by.  Nobody waits: a worker that is not last moves on to its next segment.
// Slabs cross XCDs inside one kernel, so the hand-over is a device-scope release (L2 write-back) by the deliverer
// and a device-scope acquire (L1 / L2 invalidate) by the last arriver; everything the fold writes is consumed by the
// NEXT launch.  Removes the fcc_big_step launch at every one of the T steps -- and is slower than it: with device-scope
// fences the hand-over costs 58 us per step (whole-L2 write-back / invalidate under the stream); with per-access device-scope
// stores and loads (below) it is a chain of three memory round trips (write-through acknowledge, ticket, coherent loads)
// at the tail of every launch, 3 us longer than the kernel boundary + fcc_big_step it replaces.  Probe switch only.
__device__ __forceinline__ size_t packed_op_index(int b, int i0, int NC);

struct BigFold {
  const float* x;     // x_t row base: x + t * N, utterance stride T * N
  const float* rm;
  float* a;           // ws.e + t * B * N
  float* invs;        // ws.invs + t * B * N
  float* apk;         // ws.ep[t & 1]
  float* pmax;        // ws.pmax + t * B * kBigParts, pre-filled with -inf
  unsigned* cnt;
  size_t xStride;     // T * N
  int* redo;          // ws.redo (range check, as in fcc_big_step)
};

// max into a float cell holding -inf or a previous maximum: order-preserving integer views of IEEE floats
__device__ __forceinline__ void atomic_max_f32(float* addr, float v) {
  if (v >= 0.f) atomicMax((int*)addr, __float_as_int(v));
  else atomicMin((unsigned*)addr, __float_as_uint(v));
}

// U = chunks per stage, D = ring depth (U * D == 6), NT = nontemporal loads of the transition stream
template <bool EXPOP, int kDmaU, int kDmaD, bool NT, bool FOLD = false>
__global__ __launch_bounds__(256, 2) void fcc_big_gemm_dma(const float4* __restrict__ pack, const float4* __restrict__ op,
                                                           const float* __restrict__ pmax, float* __restrict__ part, BigDims d,
                                                           BigFold f) {
  extern __shared__ __attribute__((aligned(16))) float lds[];  // [4 waves][3 stages][6 pieces][256 floats]; reused as `red`
  __shared__ int foldLast[2];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int NC = d.NC, Np = d.Np, Bp = d.Bp, B = d.B;
  const int w = blockIdx.x;
  int u0 = big_unit_begin(d, w);
  const int u1 = big_unit_begin(d, w + 1);
  constexpr int kDmaStageFloats = 3 * kDmaU * 256;  // a0, a1, e pieces of 1 KiB
  static_assert(kDmaU * kDmaD == 6 && kBigU % kDmaU == 0, "ring is 18 KiB per wave; DMA stages tile the partition stages");
  const int ratio = kBigU / kDmaU;               // DMA stages per partition stage

  float cb = 0.f;
  if (EXPOP) {
    const int b = lane & 31;
    float m = 0.f;
    if (b < B) {
      m = -INFINITY;
#pragma unroll
      for (int p = 0; p < kBigParts; ++p) m = fmaxf(m, pmax[(size_t)b * kBigParts + p]);
    }
    cb = m;
  }
  typedef __attribute__((address_space(3))) void* lp_t;
  const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void*)pack, 0, (int)((size_t)Np * d.Kp * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t re = __builtin_amdgcn_make_buffer_rsrc((void*)op, 0, (int)((size_t)Bp * d.Kp * 4), 0x00020000);
  float* ring = lds + wave * kDmaWaveFloats;
  const int voff = lane * 16;
  // DMA stages (global index) below this one belong to the cached part of this worker's share (BigDims::cache)
  const long long sCache = NT ? ((long long)u0 + (long long)(u1 - u0) * d.cache / 1000) * (kBigU / kDmaU) : 0;

  while (u0 < u1) {
    const int g = u0 / d.nS, sb = u0 - g * d.nS;
    int len = d.nS - sb;
    if (len > u1 - u0) len = u1 - u0;
    // this wave's range in DMA stages
    const int s0 = (sb + (int)((long long)len * wave / 4)) * ratio, s1 = (sb + (int)((long long)len * (wave + 1) / 4)) * ratio;
    const int piece = w - big_worker_of(d, g * d.nS);
    const int npFold = FOLD ? big_pieces(d, g) : 0;   // off the critical path of the hand-over
    f32x16 acc[2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[h][r] = 0.f;
    const uint32_t t0 = (uint32_t)(2 * g) * (uint32_t)NC, t1 = t0 + (uint32_t)NC;  // chunk index bases of the two row tiles

    const long long sBase = (long long)g * d.nS * ratio;   // global index of DMA stage 0 of this row group
    auto issue = [&](int st, int slot) {  // 6 pieces of DMA stage st into ring slot
      float* base = ring + slot * kDmaStageFloats;
      const bool nt = NT && sBase + st >= sCache;   // wave-uniform
#pragma unroll
      for (int u = 0; u < kDmaU; ++u) {
        const uint32_t c = (uint32_t)st * kDmaU + u;
        if (nt) {
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rp, (lp_t)(base + (0 * kDmaU + u) * 256), 16, voff, (int)((t0 + c) * 1024u), 0, 2);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rp, (lp_t)(base + (1 * kDmaU + u) * 256), 16, voff, (int)((t1 + c) * 1024u), 0, 2);
        } else {
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rp, (lp_t)(base + (0 * kDmaU + u) * 256), 16, voff, (int)((t0 + c) * 1024u), 0, 0);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rp, (lp_t)(base + (1 * kDmaU + u) * 256), 16, voff, (int)((t1 + c) * 1024u), 0, 0);
        }
        __builtin_amdgcn_raw_ptr_buffer_load_lds(re, (lp_t)(base + (2 * kDmaU + u) * 256), 16, voff, (int)(c * 1024u), 0, 0);
      }
    };
    if (s0 < s1) {
#pragma unroll
      for (int k = 0; k < kDmaD - 1; ++k) issue(s0 + k < s1 ? s0 + k : s0, k);
      int slot = 0;
      for (int st = s0; st < s1; ++st) {
        const int sn = st + kDmaD - 1 < s1 ? st + kDmaD - 1 : s0;   // past the range: harmless re-read of the first stage
        const int slotN = slot >= 1 ? slot - 1 : kDmaD - 1;          // (slot + D - 1) % D
        issue(sn, slotN);
        // stages st+1 .. st+D-1 (3 U (D-1) pieces) may be outstanding: stage st has landed
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * kDmaU * (kDmaD - 1)) : "memory");
        const float* base = ring + slot * kDmaStageFloats + lane * 4;
#pragma unroll
        for (int u = 0; u < kDmaU; ++u) {
          const f32x4 a0 = *(const f32x4*)(base + (0 * kDmaU + u) * 256);
          const f32x4 a1 = *(const f32x4*)(base + (1 * kDmaU + u) * 256);
          const f32x4 e4 = *(const f32x4*)(base + (2 * kDmaU + u) * 256);
          float ev[4] = {e4[0], e4[1], e4[2], e4[3]};
          if (EXPOP) {
#pragma unroll
            for (int q = 0; q < 4; ++q) ev[q] = __expf(ev[q] - cb);  // padding holds -inf -> 0
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ev[q], a0[q], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ev[q], a1[q], acc[1], 0, 0, 0);
          }
        }
        slot = slot == kDmaD - 1 ? 0 : slot + 1;
      }
    }
    // the ring becomes the reduction buffer: every wave's LDS-DMA must have landed and its reads be done
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float* red = lds;
    constexpr int NR = 32;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int r = 0; r < 16; ++r) red[((wave * NR) + h * 16 + r) * 64 + lane] = acc[h][r];
    __syncthreads();
    float* dst = part + (size_t)piece * Bp * Np;
    for (int o = threadIdx.x; o < NR * 64; o += 256) {
      const float v = (red[o] + red[NR * 64 + o]) + (red[2 * NR * 64 + o] + red[3 * NR * 64 + o]);
      const int l = o & 63, rr = o >> 6;
      const int r = rr & 15, h = rr >> 4;
      const int b = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
      const int i = 64 * g + 32 * h + (l & 31);
      // FOLD: device-scope write-through (sc1) instead of a plain store: the slab crosses XCDs inside this kernel
      if (FOLD) __hip_atomic_store(&dst[(size_t)b * Np + i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else dst[(size_t)b * Np + i] = v;
    }
    if (FOLD) {
      // no fence: a device-scope release / acquire pair writes back and invalidates the whole L2 under the transition
      // stream (measured: 140 us per step against 82 unfolded).  The slab went out with device-scope stores; they are
      // acknowledged (vmcnt) before the barrier, the ticket is taken after it, and the last arriver reads the slabs
      // with device-scope loads, which are served at the coherence point and not by a stale line of its own L2.
      // x_t and the row maxima of this thread's eight labels do not depend on the slabs: in flight across the hand-over
      float xv[2][4], rv[2][4];
      {
        const int b = threadIdx.x >> 3, qq = threadIdx.x & 7;
        const float* xr = f.x + (size_t)(b < B ? b : 0) * f.xStride;
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int i = 64 * g + 4 * (qq + 8 * it) + u;
            const int ic = i < d.N ? i : d.N - 1;
            xv[it][u] = xr[ic];
            rv[it][u] = f.rm[ic];
          }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (threadIdx.x == 0) {
        const int np = npFold;
        const unsigned tk = __hip_atomic_fetch_add(&f.cnt[g], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = tk == (unsigned)np - 1u;
        if (last) __hip_atomic_store(&f.cnt[g], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // the next launch counts from zero
        foldLast[0] = last;
        foldLast[1] = np;
      }
      __syncthreads();
      if (foldLast[0]) {
        const int np = foldLast[1], N = d.N;
        const int b = threadIdx.x >> 3, qq = threadIdx.x & 7;
        float m = -INFINITY;
        if (b < B) {
          float* ar = f.a + (size_t)b * N;
          float* ir = f.invs + (size_t)b * N;
#pragma unroll
          for (int it = 0; it < 2; ++it) {
            const int i0 = 64 * g + 4 * (qq + 8 * it);
            if (i0 < N) {
              float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
              for (int sl = 0; sl < np; ++sl) {   // worker order: the same sum as fcc_big_step
                const float* pp = part + ((size_t)sl * Bp + b) * Np + i0;
                s4.x += __hip_atomic_load(pp + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s4.y += __hip_atomic_load(pp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s4.z += __hip_atomic_load(pp + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s4.w += __hip_atomic_load(pp + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              }
              const float ssum[4] = {s4.x, s4.y, s4.z, s4.w};
              float a4[4];
#pragma unroll
              for (int u = 0; u < 4; ++u) {
                const int i = i0 + u;
                float v = -INFINITY;
                if (i < N) {
                  if (!(ssum[u] >= d.minSum)) { f.redo[b] = 1; f.redo[B] = 1; }   // range check (see fcc_big_step)
                  const float sc = fmaxf(ssum[u], 1e-37f);
                  v = xv[it][u] + (rv[it][u] + __logf(sc));
                  ir[i] = 1.f / sc;
                  ar[i] = v;
                }
                a4[u] = v;
                m = fmaxf(m, v);
              }
              *(float4*)(f.apk + packed_op_index(b, i0, NC)) = make_float4(a4[0], a4[1], a4[2], a4[3]);
            }
          }
        }
        m = fmaxf(m, __shfl_xor(m, 1));
        m = fmaxf(m, __shfl_xor(m, 2));
        m = fmaxf(m, __shfl_xor(m, 4));
        if (qq == 0 && b < B && m > -INFINITY) atomic_max_f32(f.pmax + (size_t)b * kBigParts + (g & (kBigParts - 1)), m);
      }
    }
    u0 += len;
    __syncthreads();  // `red` (= the rings) is reused by the next segment
  }
}

// ------------------------------------------------------------------ block reductions
template <int THREADS>
__device__ __forceinline__ float block_max(float v, float* sm) {
  v = wave_max(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[w] = v;
  __syncthreads();
  float m = sm[0];
#pragma unroll
  for (int i = 1; i < THREADS / 64; ++i) m = fmaxf(m, sm[i]);
  return m;
}
template <int THREADS>
__device__ __forceinline__ float block_sum(float v, float* sm) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[w] = v;
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < THREADS / 64; ++i) s += sm[i];
  return s;
}

__device__ __forceinline__ size_t packed_op_index(int b, int i0, int NC) {
  // float index of the float4 holding labels i0..i0+3 (i0 % 4 == 0) of utterance b
  return ((((size_t)(b >> 5) * NC + (i0 >> 3)) * 64) + (b & 31) + 32 * ((i0 >> 2) & 1)) * 4;
}

__device__ __forceinline__ float big_cmax(const float* __restrict__ pmax, int B, int t, int b) {
  const float* p = pmax + ((size_t)t * B + b) * kBigParts;
  float m = p[0];
#pragma unroll
  for (int q = 1; q < kBigParts; ++q) m = fmaxf(m, p[q]);
  return m;
}

// Sum of the partial slabs of a quad, in worker order: ALL P slabs of the row group, the ones its workers never write being
// zeroed once per call (adding 0.f is exact).  The loads are unconditional and issued together.  (Earlier forms: np[g] slabs
// behind a per-lane predicate -- the piece table was a dependent round trip in front of them; behind a uniform `s < np`
// branch hipcc drained the queue at every join.)  P <= 8 at the shapes of the recipes (5 at N = 9998).
__device__ __forceinline__ float4 big_slab_sum(const BigDims& d, const BigWs& ws, int b, int i0) {
  float4 p4[kBigMaxSW];
  const float* base = ws.part + (size_t)b * d.Np + i0;
  const size_t slab = (size_t)d.Bp * d.Np;
#pragma unroll
  for (int s = 0; s < 8; ++s) p4[s] = *(const float4*)(base + (size_t)min(s, d.P - 1) * slab);
#pragma unroll
  for (int s = 8; s < kBigMaxSW; ++s) p4[s] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (d.P > 8) {
    asm volatile("" ::: "memory");   // (a real branch: not to be speculated into eight more loads per thread)
#pragma unroll
    for (int s = 8; s < kBigMaxSW; ++s) p4[s] = *(const float4*)(base + (size_t)min(s, d.P - 1) * slab);
  }
  float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int s = 0; s < kBigMaxSW; ++s) {
    const bool on = s < d.P;
    s4.x += on ? p4[s].x : 0.f; s4.y += on ? p4[s].y : 0.f; s4.z += on ? p4[s].z : 0.f; s4.w += on ? p4[s].w : 0.f;
  }
  return s4;
}

// ------------------------------------------------------------------ kernel 2 (forward)
// grid (kBigParts, B): a_t[b][i] = x_t[b][i] + rowmax_i + log(sum_s part[s][b][i])   (t = 0: x_0)
__global__ __launch_bounds__(256) void fcc_big_step(BigDims d, int t, const float* __restrict__ x, BigWs ws) {
  // Every load of a quad is issued before the first use and none sits behind a lane predicate (labels past N read label
  // N - 1 and are dropped by the stores' predicates): written with `if (i < N) { load; use; }` per element, hipcc waited
  // for each element's loads in turn -- four dependent HBM round trips behind the slab loads in a kernel that is nothing
  // but latency (6.7 us per frame; ISA in profiles/r05_fcc_step_isa_before.txt).
  __shared__ float sm[4];
  const int b = blockIdx.y, N = d.N;
  const int quads = (N + 3) >> 2;
  const int qPer = (quads + kBigParts - 1) / kBigParts;
  const int q0 = blockIdx.x * qPer, q1 = min(quads, q0 + qPer);
  const float* xr = x + ((size_t)b * d.T + t) * N;
  float* ar = ws.e + ((size_t)t * d.B + b) * N;
  float* ir = ws.invs + ((size_t)t * d.B + b) * N;
  float* apk = ws.ep[t & 1];
  float m = -INFINITY;
  // Range check.  s[i] = sum_j exp(A[i][j] - rowmax_i) exp(ahat[j]) is a sum of fp32 products of two factors in (0, 1]; a factor
  // below 1e-38 is flushed, so the sum can have lost at most N * 1.2e-38.  While every sum stays above minSum = max(1e-28, 1e-30 N)
  // that is below 1e-8 relative, and every weight exp(A) e / s that an underflow zeroes in the backward recursion is below 1e-10:
  // the recursion is exact to fp32.  A smaller sum (emissions AND transitions tens of nats wide: a state whose every term lies
  // ~65 nats under the factors' maxima) means the 1e-37 floor below may have moved the posterior: the utterance is flagged and
  // recomputed in the log domain, as the reference computes every utterance, by fcc_big_exact_fwd / _bwd.
  bool low = false;
  for (int q = q0 + threadIdx.x; q < q1; q += 256) {
    const int i0 = 4 * q;
    float xv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) xv[u] = xr[min(i0 + u, N - 1)];
    float4 s4 = make_float4(1.f, 1.f, 1.f, 1.f);
    float4 rm4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t > 0) {
      rm4 = *(const float4*)(ws.rm + i0);   // [Np], Np >= i0 + 4
      s4 = big_slab_sum(d, ws, b, i0);
    }
    const float ssum[4] = {s4.x, s4.y, s4.z, s4.w};
    const float rmv[4] = {rm4.x, rm4.y, rm4.z, rm4.w};
    float a4[4], inv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      low = low || (t > 0 && i0 + u < N && !(ssum[u] >= d.minSum));   // (NaN fails the comparison too)
      const float sc = fmaxf(ssum[u], 1e-37f);
      inv[u] = 1.f / sc;
      const float v = t > 0 ? xv[u] + (rmv[u] + __logf(sc)) : xv[u];
      a4[u] = i0 + u < N ? v : -INFINITY;   // padding labels: exp(-inf - c) = 0 in the next operand
      m = fmaxf(m, a4[u]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u;
      if (i < N) {
        // (nontemporal: no later frame of the pass reads them, and they should not displace the cached half of the
        //  transition stream in the Infinity Cache)
        if (t > 0) __builtin_nontemporal_store(inv[u], &ir[i]);
        __builtin_nontemporal_store(a4[u], &ar[i]);
      }
    }
    *(float4*)(apk + packed_op_index(b, i0, d.NC)) = make_float4(a4[0], a4[1], a4[2], a4[3]);
  }
  if (low) { ws.redo[b] = 1; ws.redo[d.B] = 1; }   // (every writer writes 1: no ordering needed)
  m = block_max<256>(m, sm);
  if (threadIdx.x == 0) ws.pmax[((size_t)t * d.B + b) * kBigParts + blockIdx.x] = m;
}

// once per call: loss[b] = scale * (sum_t c_t + log sum_i exp(a_{T-1}[i] - c_{T-1}))
__global__ __launch_bounds__(kBigStepThreads) void fcc_big_loss(BigDims d, int scaleMode, const int* __restrict__ targetSize,
                                                               float* __restrict__ loss, BigWs ws) {
  __shared__ float sm[kBigStepThreads / 64];
  __shared__ double smd[kBigStepThreads / 64];
  const int b = blockIdx.x, N = d.N, T = d.T;
  double C = 0.0;
  for (int t = threadIdx.x; t < T; t += kBigStepThreads) {
    const float m = big_cmax(ws.pmax, d.B, t, b);
    ws.cfin[(size_t)t * d.B + b] = m;
    C += (double)m;
  }
  C = wave_sum_f64(C);
  if ((threadIdx.x & 63) == 0) smd[threadIdx.x >> 6] = C;
  __syncthreads();
  C = 0.0;
  for (int i = 0; i < kBigStepThreads / 64; ++i) C += smd[i];
  const float c = big_cmax(ws.pmax, d.B, T - 1, b);
  const float* ar = ws.e + ((size_t)(T - 1) * d.B + b) * N;
  float tot = 0.f;
  for (int i = threadIdx.x; i < N; i += kBigStepThreads) tot += __expf(ar[i] - c);
  tot = block_sum<kBigStepThreads>(tot, sm);
  if (threadIdx.x == 0) {
    const float sc = scale_of(scaleMode, T, targetSize[b]);
    loss[b] = (float)((double)sc * (C + (double)__logf(tot)));
    ws.scale[b] = sc;
  }
}

// ------------------------------------------------------------------ backward kernels
// t = T-1: dalpha = softmax(a_{T-1}); converts a -> e in place; emits dx, g*r (plain) and r (packed)
__global__ __launch_bounds__(kBigStepThreads) void fcc_big_bwd_init(BigDims d, const float* __restrict__ grad,
                                                                   float* __restrict__ dx, BigWs ws) {
  __shared__ float sm[kBigStepThreads / 64];
  const int b = blockIdx.x, N = d.N, t = d.T - 1;
  float* er = ws.e + ((size_t)t * d.B + b) * N;
  const float* ir = ws.invs + ((size_t)t * d.B + b) * N;
  float* dxr = dx + ((size_t)b * d.T + t) * N;
  float* rgr = ws.rg + ((size_t)t * d.B + b) * N;
  float* rpk = ws.ep[t & 1];
  const float g = ws.scale[b] * grad[b];
  if (threadIdx.x == 0) ws.gb[b] = g;
  if (ws.redo[b]) return;   // flagged by the forward range check: fcc_big_exact_bwd differentiates this utterance (its packed
                            // operand stays zero: it rides through the streaming kernel as a zero column)
  const float c = big_cmax(ws.pmax, d.B, t, b);
  float tot = 0.f;
  for (int i = threadIdx.x; i < N; i += kBigStepThreads) {
    const float e = __expf(er[i] - c);
    er[i] = e;  // same thread re-reads it below
    tot += e;
  }
  tot = block_sum<kBigStepThreads>(tot, sm);
  const float inv = 1.f / tot;
  for (int i0 = 4 * threadIdx.x; i0 < N; i0 += 4 * kBigStepThreads) {
    float r4[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u;
      r4[u] = 0.f;
      if (i < N) {
        const float da = er[i] * inv;
        dxr[i] = g * da;
        if (t > 0) {
          r4[u] = da * ir[i];
          rgr[i] = g * r4[u];
        }
      }
    }
    if (t > 0) *(float4*)(rpk + packed_op_index(b, i0, d.NC)) = make_float4(r4[0], r4[1], r4[2], r4[3]);
  }
}

// step t -> t-1 (tm = t-1): e_tm = exp(a_tm - c_tm) (stored in place);
// dalpha_tm[b][j] = e_tm[b][j] * sum_s part[s][b][j]
__global__ __launch_bounds__(256) void fcc_big_bwd_step(BigDims d, int tm, float* __restrict__ dx, BigWs ws) {
  // (loads first and unpredicated, as in fcc_big_step: the per-element `if (i < N)` form made eight dependent HBM round trips
  //  of the a / 1 / s loads: 8.7 us per frame)
  const int b = blockIdx.y, N = d.N;
  const int i0 = 4 * (blockIdx.x * 256 + threadIdx.x);
  if (i0 >= N) return;
  if (ws.redo[b]) return;   // see fcc_big_bwd_init
  float* er = ws.e + ((size_t)tm * d.B + b) * N;
  const float* ir = ws.invs + ((size_t)tm * d.B + b) * N;
  float* dxr = dx + ((size_t)b * d.T + tm) * N;
  float* rgr = ws.rg + ((size_t)tm * d.B + b) * N;
  float av[4], iv[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int ic = min(i0 + u, N - 1);
    av[u] = er[ic];
    iv[u] = ir[ic];
  }
  // (g and c are uniform: read through a lane-opaque zero they are VECTOR loads, in flight with the rest, instead of a chain of
  //  scalar loads each waited for on its own)
  int vz;
  asm volatile("v_mov_b32 %0, 0" : "=v"(vz));
  const float g = ws.gb[b + vz];
  const float c = ws.cfin[(size_t)tm * d.B + b + vz];   // the frame's maximum, reduced once by fcc_big_loss (was: 16 partial maxima per thread)
  const float4 s4 = big_slab_sum(d, ws, b, i0);
  const float D[4] = {s4.x, s4.y, s4.z, s4.w};
  float e4[4], da4[4], r4[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    e4[u] = __expf(av[u] - c);
    da4[u] = e4[u] * D[u];
    r4[u] = i0 + u < N && tm > 0 ? da4[u] * iv[u] : 0.f;
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int i = i0 + u;
    if (i < N) {
      __builtin_nontemporal_store(e4[u], &er[i]);
      __builtin_nontemporal_store(g * da4[u], &dxr[i]);
      if (tm > 0) __builtin_nontemporal_store(g * r4[u], &rgr[i]);
    }
  }
  if (tm > 0) *(float4*)(ws.ep[tm & 1] + packed_op_index(b, i0, d.NC)) = make_float4(r4[0], r4[1], r4[2], r4[3]);
}

// dA[i][j] *= exp(A[i][j] - rowmax_i);  + the flagged utterances' share (fcc_big_exact_dtrans), if there is one
__global__ __launch_bounds__(256) void fcc_big_scale_dtrans(int N, const float* __restrict__ A, const float* __restrict__ rm,
                                                           float* __restrict__ dA, const float* __restrict__ exact,
                                                           const int* __restrict__ any) {
  const size_t n = (size_t)N * N;
  const bool add = *any != 0;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
    const int i = (int)(e / N);
    float v = dA[e] * __expf(A[e] - rm[i]);
    if (add) v += exact[e];
    dA[e] = v;
  }
}

// ------------------------------------------------------------------ the exact path behind the range check (round 6)
// The utterances fcc_big_step flags are recomputed the way the reference computes every utterance (un-vendored
// fl::lib::cpu::FullConnectionCriterion, restated in oracle/criterion_oracle.c): every term ONE exponential of a sum that is <= 0,
//   L_t[i] = LSE_j(ahat_{t-1}[j] + A[i][j]),  a_t[i] = x_t[i] + L_t[i],  ahat = a - max a,
//   w_t[i][j] = exp(ahat_{t-1}[j] + A[i][j] - L_t[i]) in (0, 1],  dalpha_{t-1}[j] = sum_i dalpha_t[i] w_t[i][j],
//   dA[i][j] += g sum_t dalpha_t[i] w_t[i][j].
// N^2 exponentials per frame and utterance on the VALU instead of a pass of the matrix pipe over the packed stream: one workgroup
// per flagged utterance scans T (a frame needs the whole previous row), ~10 s at N = 9998, T = 1500 -- on a path no recipe takes
// (logits of +-10 and transitions of a few nats are two orders of magnitude inside the range check); the kernels return at once
// when nothing is flagged.  Workspace rows of a flagged utterance: e[t] = a_t (as for the others), invs[t] = L_t, rg[t] = g dalpha_t.
constexpr int kExactThreads = 1024;

__global__ __launch_bounds__(kExactThreads) void fcc_big_exact_fwd(BigDims d, const float* __restrict__ x, const float* __restrict__ A,
                                                                    BigWs ws) {
  const int b = blockIdx.x;
  if (!ws.redo[b]) return;
  __shared__ float sm[kExactThreads / 64];
  constexpr int NW = kExactThreads / 64;
  const int N = d.N, T = d.T, B = d.B;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float cPrev;
  {   // a_0 = x_0
    const float* xr = x + ((size_t)b * T) * N;
    float* ar = ws.e + (size_t)b * N;
    float m = -INFINITY;
    for (int i = threadIdx.x; i < N; i += kExactThreads) {
      const float v = xr[i];
      ar[i] = v;
      m = fmaxf(m, v);
    }
    cPrev = block_max<kExactThreads>(m, sm);
    if (threadIdx.x < kBigParts) ws.pmax[(size_t)b * kBigParts + threadIdx.x] = cPrev;
  }
  __syncthreads();
  for (int t = 1; t < T; ++t) {
    const float* ap = ws.e + ((size_t)(t - 1) * B + b) * N;
    float* ar = ws.e + ((size_t)t * B + b) * N;
    float* Lr = ws.invs + ((size_t)t * B + b) * N;
    const float* xr = x + ((size_t)b * T + t) * N;
    float mloc = -INFINITY;
    for (int i = wave; i < N; i += NW) {   // a row per wave: lanes over j, the row of A read coalesced (twice: maximum, then sum)
      const float* Ar = A + (size_t)i * N;
      float m = -INFINITY;
      for (int j = lane; j < N; j += 64) m = fmaxf(m, (ap[j] - cPrev) + Ar[j]);
      m = wave_max(m);
      float L = -INFINITY;
      if (m > -INFINITY) {   // wave-uniform
        float s = 0.f;
        for (int j = lane; j < N; j += 64) s += fast_expf(((ap[j] - cPrev) + Ar[j]) - m);   // every term <= 1, the maximum's is 1
        L = m + fast_logf(wave_sum(s));
      }
      const float a = L > -INFINITY ? xr[i] + L : -INFINITY;
      if (lane == 0) { ar[i] = a; Lr[i] = L; }
      mloc = fmaxf(mloc, a);
    }
    cPrev = block_max<kExactThreads>(mloc, sm);   // (its barriers also publish a_t to the whole workgroup: same CU, same L1)
    if (threadIdx.x < kBigParts) ws.pmax[((size_t)t * B + b) * kBigParts + threadIdx.x] = cPrev;
    __syncthreads();
  }
}

// the backward recursion of a flagged utterance (after fcc_big_loss: cfin[t] = max a_t)
__global__ __launch_bounds__(kExactThreads) void fcc_big_exact_bwd(BigDims d, const float* __restrict__ A, float* __restrict__ dx, BigWs ws) {
  const int b = blockIdx.x;
  if (!ws.redo[b]) return;
  __shared__ float sm[kExactThreads / 64];
  const int N = d.N, T = d.T, B = d.B;
  const float g = ws.gb[b];
  {   // g dalpha_{T-1} = g softmax(a_{T-1})
    const int t = T - 1;
    const float* ar = ws.e + ((size_t)t * B + b) * N;
    float* gr = ws.rg + ((size_t)t * B + b) * N;
    float* dxr = dx + ((size_t)b * T + t) * N;
    const float c = ws.cfin[(size_t)t * B + b];
    float tot = 0.f;
    for (int i = threadIdx.x; i < N; i += kExactThreads) tot += fast_expf(ar[i] - c);
    tot = block_sum<kExactThreads>(tot, sm);
    const float inv = g / tot;
    for (int i = threadIdx.x; i < N; i += kExactThreads) {
      const float v = inv * fast_expf(ar[i] - c);
      gr[i] = v;
      dxr[i] = v;
    }
  }
  __syncthreads();
  constexpr int JR = 4;
  for (int t = T - 1; t >= 1; --t) {
    const float* gd = ws.rg + ((size_t)t * B + b) * N;
    const float* Lr = ws.invs + ((size_t)t * B + b) * N;
    const float* ap = ws.e + ((size_t)(t - 1) * B + b) * N;
    const float cp = ws.cfin[(size_t)(t - 1) * B + b];
    float* go = ws.rg + ((size_t)(t - 1) * B + b) * N;
    float* dxr = dx + ((size_t)b * T + (t - 1)) * N;
    for (int j0 = 0; j0 < N; j0 += kExactThreads * JR) {   // a thread owns JR states j; the rows of A pass by coalesced
      int js[JR];
      float ah[JR], acc[JR];
#pragma unroll
      for (int r = 0; r < JR; ++r) {
        js[r] = j0 + r * kExactThreads + (int)threadIdx.x;
        ah[r] = js[r] < N ? ap[js[r]] - cp : -INFINITY;
        acc[r] = 0.f;
        if (js[r] >= N) js[r] = N - 1;
      }
      for (int i = 0; i < N; ++i) {
        const float gi = gd[i];      // uniform
        if (!(gi != 0.f)) continue;  // a state without posterior mass hands nothing on (and its L may be -inf)
        const float Li = Lr[i];
        const float* Ar = A + (size_t)i * N;
#pragma unroll
        for (int r = 0; r < JR; ++r) acc[r] += gi * fast_expf((ah[r] + Ar[js[r]]) - Li);   // exp(-inf) = 0: dead states, padding
      }
#pragma unroll
      for (int r = 0; r < JR; ++r) {
        const int j = j0 + r * kExactThreads + (int)threadIdx.x;
        if (j < N) { go[j] = acc[r]; dxr[j] = acc[r]; }
      }
    }
    __syncthreads();
  }
}

// out[i][j] = sum over flagged utterances b, frames t >= 1 of (g dalpha_t)[b][i] exp(ahat_{t-1}[b][j] + A[i][j] - L_t[b][i]):
// fully parallel over (i, j); a wave owns 8 rows x 64 columns.  Returns at once when no utterance is flagged.
__global__ __launch_bounds__(256) void fcc_big_exact_dtrans(BigDims d, const float* __restrict__ A, float* __restrict__ out, BigWs ws) {
  const int N = d.N, T = d.T, B = d.B;
  if (!ws.redo[B]) return;
  const int j = blockIdx.x * 64 + (threadIdx.x & 63);
  const int i0 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * 8;
  if (i0 >= N) return;   // wave-uniform
  const int jc = min(j, N - 1);
  float a[8], acc[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    a[r] = A[(size_t)min(i0 + r, N - 1) * N + jc];
    acc[r] = 0.f;
  }
  for (int b = 0; b < B; ++b) {
    if (!ws.redo[b]) continue;
    for (int t = 1; t < T; ++t) {
      const float ah = ws.e[((size_t)(t - 1) * B + b) * N + jc] - ws.cfin[(size_t)(t - 1) * B + b];
      const float* gd = ws.rg + ((size_t)t * B + b) * N;
      const float* Lr = ws.invs + ((size_t)t * B + b) * N;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int i = min(i0 + r, N - 1);
        const float gi = gd[i];   // wave-uniform
        if (gi != 0.f) acc[r] += gi * fast_expf((ah + a[r]) - Lr[i]);
      }
    }
  }
  if (j < N) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
      if (i0 + r < N) out[(size_t)(i0 + r) * N + j] = acc[r];
  }
}

// the rows of a flagged utterance leave the transition-gradient product (its share came from fcc_big_exact_dtrans): zero both
// operands' rows -- a_t may hold -inf, and 0 x inf would poison the product
__global__ __launch_bounds__(256) void fcc_big_exact_clear(BigDims d, BigWs ws) {
  const int b = blockIdx.y;
  if (!ws.redo[b]) return;
  for (int t = blockIdx.x; t < d.T; t += gridDim.x) {
    float* er = ws.e + ((size_t)t * d.B + b) * d.N;
    float* gr = ws.rg + ((size_t)t * d.B + b) * d.N;
    for (int i = threadIdx.x; i < d.N; i += 256) { er[i] = 0.f; gr[i] = 0.f; }
  }
}

template <int NB, int RT, bool EXPOP>
static int launch_big_gemm(const BigDims& d, const float* pack, const float* op, const float* pmax, float* part,
                           hipStream_t s) {
  const size_t shmem = (size_t)4 * NB * RT * 16 * 64 * sizeof(float);
  if (shmem > 64 * 1024)
    W2L_HIP_CHECK(hipFuncSetAttribute((const void*)fcc_big_gemm<NB, RT, EXPOP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  // algorithmic bytes of one step: the transition matrix once + read/write of one [B][N] row pair
  prof_begin(s, 4.0 * d.N * (double)d.N + 8.0 * d.B * (double)d.N, PROF_FCC_STREAM);
  hipLaunchKernelGGL((fcc_big_gemm<NB, RT, EXPOP>), dim3((unsigned)d.W), dim3(256), shmem, s, (const float4*)pack,
                     (const float4*)op, pmax, part, d);
  prof_end(s);
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}
// the streaming kernel that can carry the folded forward step (big_gemm's first branch)
static inline bool big_fold_ok(const BigDims& d) {
  return d.fold && d.NB == 1 && d.RT == 2 && d.dma && !d.abl && (size_t)d.Np * d.Kp * 4 < 0x7fffffffull;
}

template <bool EXPOP, bool FOLD = false>
static int big_gemm(const BigDims& d, const float* pack, const float* op, const float* pmax, float* part, hipStream_t s,
                    const BigFold& fold = BigFold{}) {
  if (d.NB == 1 && d.RT == 2 && d.dma && !d.abl && (size_t)d.Np * d.Kp * 4 < 0x7fffffffull) {
    const size_t shmem = (size_t)4 * kDmaWaveFloats * sizeof(float);  // 72 KiB (>= the 32 KiB reduction buffer)
    prof_begin(s, 4.0 * d.N * (double)d.N + 8.0 * d.B * (double)d.N, PROF_FCC_STREAM);
#define W2L_DMA_LAUNCH(U, D, NTF)                                                                                              \
  do {                                                                                                                         \
    static bool attr[64] = {};                                                                                                  \
    if (first_on_device(attr)) {                                                                                                               \
      W2L_HIP_CHECK(hipFuncSetAttribute((const void*)fcc_big_gemm_dma<EXPOP, U, D, NTF, FOLD>,                                 \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));                              \
    }                                                                                                                          \
    hipLaunchKernelGGL((fcc_big_gemm_dma<EXPOP, U, D, NTF, FOLD>), dim3((unsigned)d.W), dim3(256), shmem, s,                   \
                       (const float4*)pack, (const float4*)op, pmax, part, d, fold);                                           \
  } while (0)
    switch (d.dma) {
      case 2: W2L_DMA_LAUNCH(1, 6, false); break;
      case 3: W2L_DMA_LAUNCH(2, 3, true); break;
      case 4: W2L_DMA_LAUNCH(1, 6, true); break;
      default: W2L_DMA_LAUNCH(2, 3, false); break;
    }
#undef W2L_DMA_LAUNCH
    prof_end(s);
    W2L_LAUNCH_CHECK();
    return W2L_OK;
  }
  if (d.NB == 1 && d.RT == 2 && d.asmv && !d.abl) {
    const size_t shmem = (size_t)4 * 32 * 64 * sizeof(float);
    prof_begin(s, 4.0 * d.N * (double)d.N + 8.0 * d.B * (double)d.N, PROF_FCC_STREAM);
    hipLaunchKernelGGL((fcc_big_gemm_asm<EXPOP>), dim3((unsigned)d.W), dim3(256), shmem, s, (const float4*)pack, (const float4*)op, pmax, part, d);
    prof_end(s);
    W2L_LAUNCH_CHECK();
    return W2L_OK;
  }
  if (d.NB == 1 && d.RT == 2 && d.abl) {  // timing ablations of the probe tool (results are garbage)
    const size_t shmem = (size_t)4 * 1 * 2 * 16 * 64 * sizeof(float);
    prof_begin(s, 4.0 * d.N * (double)d.N + 8.0 * d.B * (double)d.N, PROF_FCC_STREAM);
    if (d.abl == 4) hipLaunchKernelGGL((fcc_big_gemm<1, 2, EXPOP, 4>), dim3((unsigned)d.W), dim3(256), shmem, s, (const float4*)pack, (const float4*)op, pmax, part, d);
    else if (d.abl == 7) hipLaunchKernelGGL((fcc_big_gemm<1, 2, EXPOP, 7>), dim3((unsigned)d.W), dim3(256), shmem, s, (const float4*)pack, (const float4*)op, pmax, part, d);
    else if (d.abl == 1) hipLaunchKernelGGL((fcc_big_gemm<1, 2, EXPOP, 1>), dim3((unsigned)d.W), dim3(256), shmem, s, (const float4*)pack, (const float4*)op, pmax, part, d);
    else if (d.abl == 2) hipLaunchKernelGGL((fcc_big_gemm<1, 2, EXPOP, 2>), dim3((unsigned)d.W), dim3(256), shmem, s, (const float4*)pack, (const float4*)op, pmax, part, d);
    else hipLaunchKernelGGL((fcc_big_gemm<1, 2, EXPOP, 3>), dim3((unsigned)d.W), dim3(256), shmem, s, (const float4*)pack, (const float4*)op, pmax, part, d);
    prof_end(s);
    W2L_LAUNCH_CHECK();
    return W2L_OK;
  }
  if (d.NB == 1) return d.RT == 4 ? launch_big_gemm<1, 4, EXPOP>(d, pack, op, pmax, part, s)
                                  : launch_big_gemm<1, 2, EXPOP>(d, pack, op, pmax, part, s);
  if (d.NB == 2) return launch_big_gemm<2, 2, EXPOP>(d, pack, op, pmax, part, s);
  return launch_big_gemm<4, 2, EXPOP>(d, pack, op, pmax, part, s);
}

bool fcc_big_supported(int B, int T, int N) {
  (void)T;
  return B <= 32 * kBigMaxNB && N >= 64 && N <= (1 << 20);
}

size_t fcc_big_workspace_size(int B, int T, int N) {
  BigDims d = big_dims(B, T, N);
  return big_ws(nullptr, d).bytes;
}

const int* fcc_big_range_flags(int B, int T, int N, const void* workspace) {
  return big_ws((void*)workspace, big_dims(B, T, N)).redo;
}

static int big_pack(const BigDims& d, const BigWs& ws, const float* trans, bool transposed, hipStream_t s) {
  const size_t total4 = (size_t)d.Np * d.Kp / 4;
  const unsigned blocks = (unsigned)((total4 + 255) / 256);
  if (transposed)
    hipLaunchKernelGGL(big_pack_k<true>, dim3(blocks), dim3(256), 0, s, d.N, d.NC, total4, trans, ws.rm, (float4*)ws.pack);
  else
    hipLaunchKernelGGL(big_pack_k<false>, dim3(blocks), dim3(256), 0, s, d.N, d.NC, total4, trans, ws.rm, (float4*)ws.pack);
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}

// packed operand buffers: forward operand a (padding must read as -inf -> exp = 0), backward r (padding 0)
__global__ __launch_bounds__(256) void big_fill_k(float* __restrict__ p, size_t n, float v) {
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) p[e] = v;
}

int fcc_big_forward(int B, int T, int N, int scaleMode, const float* input, const int* targetSize,
                    const float* trans, float* loss, void* workspace, hipStream_t s) {
  const BigDims d = big_dims(B, T, N);
  const BigWs ws = big_ws(workspace, d);
  hipLaunchKernelGGL(big_rowmax_k, dim3((unsigned)((d.Np + 3) / 4)), dim3(256), 0, s, N, d.Np, trans, ws.rm);
  W2L_LAUNCH_CHECK();
  int st = big_pack(d, ws, trans, false, s);
  if (st) return st;
  const size_t opFloats = 2 * align_up((size_t)d.Bp * d.Kp * sizeof(float), 256) / sizeof(float);
  hipLaunchKernelGGL(big_fill_k, dim3(512), dim3(256), 0, s, ws.ep[0], opFloats, -INFINITY);
  W2L_LAUNCH_CHECK();
  // the step kernels add all P slabs of a row group: the ones its workers never write read as zero
  W2L_HIP_CHECK(hipMemsetAsync(ws.part, 0, (size_t)d.P * d.Bp * d.Np * sizeof(float), s));
  W2L_HIP_CHECK(hipMemsetAsync(ws.redo, 0, (size_t)(B + 1) * sizeof(int), s));   // range-check flags (fcc_big_step raises them)
  const dim3 sgrid(kBigParts, (unsigned)B);
  const bool fold = big_fold_ok(d) && T > 1;
  if (fold) {
    // partial maxima of steps 1 .. T-1 arrive by atomic max; arrival tickets start at zero (and reset themselves)
    hipLaunchKernelGGL(big_fill_k, dim3(512), dim3(256), 0, s, ws.pmax, (size_t)T * B * kBigParts, -INFINITY);
    W2L_LAUNCH_CHECK();
    W2L_HIP_CHECK(hipMemsetAsync(ws.cnt, 0, (size_t)d.G * sizeof(unsigned), s));
  }
  for (int t = 0; t < T; ++t) {
    if (t > 0) {
      if (fold) {
        BigFold f;
        f.x = input + (size_t)t * N; f.xStride = (size_t)T * N; f.rm = ws.rm;
        f.a = ws.e + (size_t)t * B * N; f.invs = ws.invs + (size_t)t * B * N; f.apk = ws.ep[t & 1];
        f.pmax = ws.pmax + (size_t)t * B * kBigParts; f.cnt = ws.cnt; f.redo = ws.redo;
        st = big_gemm<true, true>(d, ws.pack, ws.ep[(t - 1) & 1], ws.pmax + (size_t)(t - 1) * B * kBigParts, ws.part, s, f);
        if (st) return st;
        continue;
      }
      st = big_gemm<true>(d, ws.pack, ws.ep[(t - 1) & 1], ws.pmax + (size_t)(t - 1) * B * kBigParts, ws.part, s);
      if (st) return st;
    }
    hipLaunchKernelGGL(fcc_big_step, sgrid, dim3(256), 0, s, d, t, input, ws);
    W2L_LAUNCH_CHECK();
  }
  // the utterances the range check flagged: the whole recursion again in the log domain (returns at once for the others)
  hipLaunchKernelGGL(fcc_big_exact_fwd, dim3((unsigned)B), dim3(kExactThreads), 0, s, d, input, trans, ws);
  W2L_LAUNCH_CHECK();
  hipLaunchKernelGGL(fcc_big_loss, dim3((unsigned)B), dim3(kBigStepThreads), 0, s, d, scaleMode, targetSize, loss, ws);
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}

int fcc_big_backward(int B, int T, int N, const float* trans, const float* grad, float* inputGrad,
                     float* transGrad, void* workspace, hipStream_t s) {
  const BigDims d = big_dims(B, T, N);
  const BigWs ws = big_ws(workspace, d);
  int st = big_pack(d, ws, trans, true, s);  // rm and cfin are still valid from forward
  if (st) return st;
  W2L_HIP_CHECK(hipMemsetAsync(ws.ep[0], 0, 2 * align_up((size_t)d.Bp * d.Kp * sizeof(float), 256), s));
  W2L_HIP_CHECK(hipMemsetAsync(ws.part, 0, (size_t)d.P * d.Bp * d.Np * sizeof(float), s));
  hipLaunchKernelGGL(fcc_big_bwd_init, dim3((unsigned)B), dim3(kBigStepThreads), 0, s, d, grad, inputGrad, ws);
  W2L_LAUNCH_CHECK();
  const dim3 sgrid((unsigned)((N + 1023) / 1024), (unsigned)B);
  for (int t = T - 1; t >= 1; --t) {
    st = big_gemm<false>(d, ws.pack, ws.ep[t & 1], nullptr, ws.part, s);
    if (st) return st;
    hipLaunchKernelGGL(fcc_big_bwd_step, sgrid, dim3(256), 0, s, d, t - 1, inputGrad, ws);
    W2L_LAUNCH_CHECK();
  }
  // flagged utterances (skipped by the kernels above): exact recursion, their share of the transition gradient into the pack
  // buffer (>= N x N floats, free now), then their rows leave the product below.  Three launches that return at once otherwise.
  hipLaunchKernelGGL(fcc_big_exact_bwd, dim3((unsigned)B), dim3(kExactThreads), 0, s, d, trans, inputGrad, ws);
  W2L_LAUNCH_CHECK();
  if (T == 1) {
    W2L_HIP_CHECK(hipMemsetAsync(transGrad, 0, (size_t)N * N * sizeof(float), s));
    return W2L_OK;
  }
  hipLaunchKernelGGL(fcc_big_exact_dtrans, dim3((unsigned)((N + 63) / 64), (unsigned)((N + 31) / 32)), dim3(256), 0, s, d, trans, ws.pack, ws);
  W2L_LAUNCH_CHECK();
  hipLaunchKernelGGL(fcc_big_exact_clear, dim3(64, (unsigned)B), dim3(256), 0, s, d, ws);
  W2L_LAUNCH_CHECK();
  // dA_raw[i][j] = sum_{t>=1,b} (g r_t)[b][i] * e_{t-1}[b][j] : both operands "k-rows", reduction over (t,b)
  st = gemm_f32(ws.rg + (size_t)B * N, N, 0, ws.e, N, 0, transGrad, N, N, N, (T - 1) * B, nullptr, 0, 1, s);
  if (st) return st;
  hipLaunchKernelGGL(fcc_big_scale_dtrans, dim3(2048), dim3(256), 0, s, N, trans, ws.rm, transGrad, ws.pack, ws.redo + B);
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}

}  // namespace w2l

// =====================================================================================
// ViterbiPath for large N (ASGLoss::viterbiPath, recipes/slimIPL/src/Train.cpp:838, :1375).
// BIT-EXACT with the CPU recursion (oracle/criterion_oracle.c: v = delta[j] + A[i][j] in fp32,
// strict '>' scanning j upward, then + x[t][i]; first maximum wins everywhere).
//
// Max-plus has no matrix-core form: N^2 B (add, compare, 2 selects) per step on the VALU.
// Mapping: lane <-> destination state i over a TRANSPOSED copy AT[j][i] of the transitions
// (coalesced 256 B row segments), delta_{t-1}[b][j] is wave-uniform and comes through the
// scalar cache (stored [j][b]), so the inner loop is 4 VALU ops per (b, j) with no cross-lane
// traffic and exactly the oracle's j order.  The j range is split over gridDim.y waves per
// 64-state group (partials combined in split order => still first-max-wins).
// =====================================================================================
namespace w2l {

struct VitBigWs {
  float* AT;             // [N][Np]   AT[j][i] = A[i][j]
  float* dT[2];          // [N][Bp]   delta^T, double-buffered over t
  float* pbest;          // [S][Bp][Np]
  int* parg;             // [S][Bp][Np]
  unsigned short* psi;   // [B][T][N]
  int S, Np, Bp;
  size_t bytes;
};

__host__ inline VitBigWs vit_big_ws(void* ws, int B, int T, int N) {
  VitBigWs w;
  char* p = (char*)ws;
  auto take = [&](size_t bytes) { char* q = p; p += align_up(bytes, 256); return q; };
  w.Np = (N + 63) / 64 * 64;
  w.Bp = B <= 1 ? 1 : (B <= 8 ? 8 : (B + 31) / 32 * 32);
  const int groups = w.Np / 64;
  int S = (2560 + groups - 1) / groups;
  if (S > N / 64) S = N / 64;
  if (S < 1) S = 1;
  w.S = S;
  w.AT = (float*)take((size_t)N * w.Np * sizeof(float));
  w.dT[0] = (float*)take((size_t)N * w.Bp * sizeof(float));
  w.dT[1] = (float*)take((size_t)N * w.Bp * sizeof(float));
  w.pbest = (float*)take((size_t)S * w.Bp * w.Np * sizeof(float));
  w.parg = (int*)take((size_t)S * w.Bp * w.Np * sizeof(int));
  w.psi = (unsigned short*)take((size_t)B * T * N * sizeof(unsigned short));
  w.bytes = (size_t)(p - (char*)ws);
  return w;
}

// AT[j][i] = A[i][j], rows padded to Np with 0 (padded states are never read back)
__global__ __launch_bounds__(256) void vit_big_transpose(int N, int Np, const float* __restrict__ A, float* __restrict__ AT) {
  __shared__ float tile[32][33];
  const int i0 = blockIdx.x * 32, j0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int k = ty; k < 32; k += 8) {
    const int i = i0 + k, j = j0 + tx;
    tile[k][tx] = (i < N && j < N) ? A[(size_t)i * N + j] : 0.f;
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    const int j = j0 + k, i = i0 + tx;
    if (j < N && i < Np) AT[(size_t)j * Np + i] = tile[tx][k];
  }
}

// delta_0 = x_0, stored transposed [j][Bp]
__global__ __launch_bounds__(256) void vit_big_init(int B, int T, int N, int Bp, const float* __restrict__ x, float* __restrict__ dT) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= N) return;
  for (int b = 0; b < Bp; ++b) dT[(size_t)j * Bp + b] = b < B ? x[((size_t)b * T) * N + j] : 0.f;
}

template <int BT>
__global__ __launch_bounds__(64) void vit_big_scan(int N, int Np, int Bp, int S, const float* __restrict__ AT,
                                                   const float* __restrict__ dT, float* __restrict__ pbest,
                                                   int* __restrict__ parg) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  const int s = blockIdx.y, b0 = blockIdx.z * BT;
  const int j0 = (int)((long long)N * s / S), j1 = (int)((long long)N * (s + 1) / S);
  float best[BT];
  int arg[BT];
  {
    const float a = AT[(size_t)j0 * Np + i];
    const float* d = dT + (size_t)j0 * Bp + b0;
#pragma unroll
    for (int b = 0; b < BT; ++b) { best[b] = d[b] + a; arg[b] = j0; }
  }
  int j = j0 + 1;
  for (; j + 4 <= j1; j += 4) {
    float a[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) a[u] = AT[(size_t)(j + u) * Np + i];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float* d = dT + (size_t)(j + u) * Bp + b0;
#pragma unroll
      for (int b = 0; b < BT; ++b) {
        const float v = d[b] + a[u];
        const bool gt = v > best[b];
        best[b] = gt ? v : best[b];
        arg[b] = gt ? j + u : arg[b];
      }
    }
  }
  for (; j < j1; ++j) {
    const float a = AT[(size_t)j * Np + i];
    const float* d = dT + (size_t)j * Bp + b0;
#pragma unroll
    for (int b = 0; b < BT; ++b) {
      const float v = d[b] + a;
      const bool gt = v > best[b];
      best[b] = gt ? v : best[b];
      arg[b] = gt ? j : arg[b];
    }
  }
#pragma unroll
  for (int b = 0; b < BT; ++b) {
    pbest[((size_t)s * Bp + b0 + b) * Np + i] = best[b];
    parg[((size_t)s * Bp + b0 + b) * Np + i] = arg[b];
  }
}

// combine the S split partials in j order, add x_t, emit delta_t^T and psi_t
__global__ __launch_bounds__(256) void vit_big_combine(int B, int T, int N, int Np, int Bp, int S, int t,
                                                      const float* __restrict__ x, const float* __restrict__ pbest,
                                                      const int* __restrict__ parg, float* __restrict__ dTout,
                                                      unsigned short* __restrict__ psi) {
  const int i = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (i >= N) return;
  float best = pbest[((size_t)b) * Np + i];
  int arg = parg[((size_t)b) * Np + i];
  for (int s = 1; s < S; ++s) {
    const float v = pbest[((size_t)s * Bp + b) * Np + i];
    if (v > best) { best = v; arg = parg[((size_t)s * Bp + b) * Np + i]; }
  }
  dTout[(size_t)i * Bp + b] = best + x[((size_t)b * T + t) * N + i];
  psi[((size_t)b * T + t) * N + i] = (unsigned short)arg;
}

// final state (first argmax) + backtrace: one wavefront per utterance
__global__ __launch_bounds__(64) void vit_big_backtrace(int T, int N, int Bp, const float* __restrict__ dT,
                                                        const unsigned short* __restrict__ psi, int* __restrict__ path) {
  const int b = blockIdx.x, lane = threadIdx.x;
  float best = -INFINITY;
  int arg = 0x7fffffff;
  for (int i = lane; i < N; i += 64) {
    const float v = dT[(size_t)i * Bp + b];
    if (v > best || (v == best && i < arg)) { best = v; arg = i; }
  }
  // lexicographic (max value, min index) across lanes
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float ov = __shfl_xor(best, off);
    const int oa = __shfl_xor(arg, off);
    if (ov > best || (ov == best && oa < arg)) { best = ov; arg = oa; }
  }
  if (lane == 0) {
    int cur = arg == 0x7fffffff ? 0 : arg;
    int* pb = path + (size_t)b * T;
    pb[T - 1] = cur;
    for (int t = T - 1; t >= 1; --t) {
      cur = psi[((size_t)b * T + t) * N + cur];
      pb[t - 1] = cur;
    }
  }
}

bool viterbi_big_supported(int B, int T, int N) {
  (void)T;
  return N >= 64 && N <= 65535 && B >= 1;
}
size_t viterbi_big_workspace_size(int B, int T, int N) { return vit_big_ws(nullptr, B, T, N).bytes; }

int viterbi_big_compute(int B, int T, int N, const float* input, const float* trans, int* path, void* workspace,
                        hipStream_t s) {
  const VitBigWs w = vit_big_ws(workspace, B, T, N);
  hipLaunchKernelGGL(vit_big_transpose, dim3((unsigned)(w.Np / 32), (unsigned)((N + 31) / 32)), dim3(256), 0, s, N, w.Np,
                     trans, w.AT);
  W2L_LAUNCH_CHECK();
  hipLaunchKernelGGL(vit_big_init, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, B, T, N, w.Bp, input, w.dT[0]);
  W2L_LAUNCH_CHECK();
  const int BT = w.Bp >= 32 ? 32 : w.Bp;
  const dim3 sgrid((unsigned)(w.Np / 64), (unsigned)w.S, (unsigned)(w.Bp / BT));
  const dim3 cgrid((unsigned)((N + 255) / 256), (unsigned)B);
  for (int t = 1; t < T; ++t) {
    const float* din = w.dT[(t - 1) & 1];
    if (BT == 32)
      hipLaunchKernelGGL(vit_big_scan<32>, sgrid, dim3(64), 0, s, N, w.Np, w.Bp, w.S, w.AT, din, w.pbest, w.parg);
    else if (BT == 8)
      hipLaunchKernelGGL(vit_big_scan<8>, sgrid, dim3(64), 0, s, N, w.Np, w.Bp, w.S, w.AT, din, w.pbest, w.parg);
    else
      hipLaunchKernelGGL(vit_big_scan<1>, sgrid, dim3(64), 0, s, N, w.Np, w.Bp, w.S, w.AT, din, w.pbest, w.parg);
    W2L_LAUNCH_CHECK();
    hipLaunchKernelGGL(vit_big_combine, cgrid, dim3(256), 0, s, B, T, N, w.Np, w.Bp, w.S, t, input, w.pbest, w.parg,
                       w.dT[t & 1], w.psi);
    W2L_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(vit_big_backtrace, dim3((unsigned)B), dim3(64), 0, s, T, N, w.Bp, w.dT[(T - 1) & 1], w.psi, path);
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}

}  // namespace w2l
