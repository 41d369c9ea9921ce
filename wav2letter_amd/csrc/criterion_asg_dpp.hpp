// criterion_asg_dpp.hpp -- FullConnectionCriterion and ViterbiPath for N <= 31 states (the ASG letter sets: N = 30 for
// LibriSpeech, recipes/conv_glu/librispeech/train.cfg), one wavefront per utterance, included by criterion_fcc.hip.
//
// Replaces Flashlight's fl::lib::{cpu,cuda}::FullConnectionCriterion<float> / ViterbiPath<float> (un-vendored; call sites
// recipes/slimIPL/src/Train.cpp:408-410, :1675, :838; math SURVEY.md App. B.2 / B.3; CPU restatement oracle/criterion_oracle.c;
// the arithmetic of THIS file is modelled op for op in oracle/asg_linear_domain.py::fcc_kernel_model*).
//
// A frame of these scans is a 32 x 32 matrix-vector product whose result feeds the next frame: T dependent steps, so what counts
// is the number of instructions ONE wave must issue per frame.  The previous generation (fcc_fwd_small) broadcast every state
// through v_readlane into an SGPR operand -- 32 + 16 instructions -- and carried exp / log / a wave maximum on the chain
// (~590 cycles per frame).  Here:
//   * the vector lives in a SCALED LINEAR domain (no exp / log on the chain): u_t = (E u_{t-1}) * q_t with
//     E = exp(A - rowmax), q_t = 2^(x_t log2 e + rowmax log2 e - max) * 2^-k_t;  k_t is a power-of-two scale that follows the
//     total mass with a LAG -- k_{t+1} = exponent(sum_j u_{t-1}[j]) - k_t -- so neither a maximum nor a normalisation sits on
//     the chain, and the magnitude of u_t is bounded by the growth of two frames.  The sum arrives for free as row 31 of the
//     product (E[31][j] = 1; that is why N <= 31);
//   * the product runs on DPP row rotations: the 64 lanes are 4 rows of 16; every row holds one 16-state half of the vector
//     and computes, with 16 `v_fmac_f32_dpp row_ror:n` (rotation and multiply-add in ONE instruction, no broadcast at all),
//     the partial sums of 16 output states over the 16 inputs it holds: 4 rows = the 4 blocks of the 32 x 32 matrix.  The two
//     partials of an output state are added across rows by ONE v_permlane32_swap (rows 0+2, 1+3) or v_permlane16_swap
//     (rows 0+1, 2+3) -- gfx950 instructions.  A combine leaves each half of the result in TWO rows, but not the two rows the
//     same step needs as input, so the steps ALTERNATE between two arrangements and two register sets of the matrix:
//       arrangement H: rows hold halves [0,0,1,1]  --step A: rows produce halves [0,1,0,1], permlane32 combine-->  arrangement G
//       arrangement G: rows hold halves [0,1,0,1]  --step B: rows produce halves [0,0,1,1], permlane16 combine-->  arrangement H
//     ~21 chain instructions per frame instead of ~60.  The lane <-> source-lane map of `row_ror:n` is CALIBRATED at kernel
//     start (a rotation of the lane index), and the two swap instructions are checked on known values: a semantic surprise
//     poisons the loss (NaN) instead of producing wrong numbers.
// Workspace (FccWs): `ahat` holds u_t, `logs` holds q_t, `r` holds r_t = b_t q_t of the backward scan ([B][T][N] each).
// Backward: b_{t-1} = E^T (b_t q_t) in the same scaled domain (the beta recursion), d loss / d x_t = u_t b_t, and the transition
// gradient E .* sum_t r_t u_{t-1}^T by the parallel kernel fcc_dtrans_small<.., true>.
// Viterbi: the same rotations with (+, max): 16 v_add_f32_dpp + 8 v_max3_f32 per frame.  Only delta is on the chain; the
// back-pointers psi_t[i] = first argmax_j(delta_{t-1}[j] + A[i][j]) -- same fp32 sums, first maximum wins, as the oracle's
// strict '>' scan -- are recomputed from the stored delta rows by the parallel kernel vit_psi_k and walked by vit_walk_k.
#pragma once
#include "common.hpp"
#include <type_traits>

namespace w2l {

constexpr int kDppChunk = 16;   // frames per prefetch chunk (even: a frame's arrangement depends on its parity only)

template <int N> __device__ __forceinline__ int dpp_ror_i(int v) {
  return __builtin_amdgcn_update_dpp(0, v, 0x120 + N, 0xf, 0xf, true);
}

// sum_n E[n] * u[source lane of row_ror:n]  (n = 0: the lane itself); two accumulators
__device__ __forceinline__ float dpp_dot16(float u, const float (&E)[16]) {
  float a0, a1;
  asm("s_nop 1\n\t"
      "v_mul_f32_e32 %0, %2, %3\n\t"
      "v_mul_f32_dpp %1, %2, %4 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %0, %2, %5 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %1, %2, %6 row_ror:3 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %0, %2, %7 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %1, %2, %8 row_ror:5 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %0, %2, %9 row_ror:6 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %1, %2, %10 row_ror:7 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %0, %2, %11 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %1, %2, %12 row_ror:9 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %0, %2, %13 row_ror:10 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %1, %2, %14 row_ror:11 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %0, %2, %15 row_ror:12 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %1, %2, %16 row_ror:13 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %0, %2, %17 row_ror:14 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %1, %2, %18 row_ror:15 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0"
      : "=&v"(a0), "=&v"(a1)
      : "v"(u), "v"(E[0]), "v"(E[1]), "v"(E[2]), "v"(E[3]), "v"(E[4]), "v"(E[5]), "v"(E[6]), "v"(E[7]), "v"(E[8]), "v"(E[9]),
        "v"(E[10]), "v"(E[11]), "v"(E[12]), "v"(E[13]), "v"(E[14]), "v"(E[15]));
  return a0 + a1;
}

// max_n (A[n] + d[source lane of row_ror:n])  -- the (max, +) form of dpp_dot16: 16 adds with the rotation folded in, 8 v_max3
__device__ __forceinline__ float dpp_maxplus16(float d, const float (&A)[16]) {
  float m, t0, t1;
  asm("s_nop 1\n\t"
      "v_add_f32_e32 %0, %3, %4\n\t"
      "v_add_f32_dpp %1, %3, %5 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %3, %6 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
      "v_max3_f32 %0, %0, %1, %2\n\t"
      "v_add_f32_dpp %1, %3, %7 row_ror:3 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %3, %8 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_max3_f32 %0, %0, %1, %2\n\t"
      "v_add_f32_dpp %1, %3, %9 row_ror:5 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %3, %10 row_ror:6 row_mask:0xf bank_mask:0xf\n\t"
      "v_max3_f32 %0, %0, %1, %2\n\t"
      "v_add_f32_dpp %1, %3, %11 row_ror:7 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %3, %12 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_max3_f32 %0, %0, %1, %2\n\t"
      "v_add_f32_dpp %1, %3, %13 row_ror:9 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %3, %14 row_ror:10 row_mask:0xf bank_mask:0xf\n\t"
      "v_max3_f32 %0, %0, %1, %2\n\t"
      "v_add_f32_dpp %1, %3, %15 row_ror:11 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %3, %16 row_ror:12 row_mask:0xf bank_mask:0xf\n\t"
      "v_max3_f32 %0, %0, %1, %2\n\t"
      "v_add_f32_dpp %1, %3, %17 row_ror:13 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %3, %18 row_ror:14 row_mask:0xf bank_mask:0xf\n\t"
      "v_max3_f32 %0, %0, %1, %2\n\t"
      "v_add_f32_dpp %1, %3, %19 row_ror:15 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_max_f32_e32 %0, %0, %1"
      : "=&v"(m), "=&v"(t0), "=&v"(t1)
      : "v"(d), "v"(A[0]), "v"(A[1]), "v"(A[2]), "v"(A[3]), "v"(A[4]), "v"(A[5]), "v"(A[6]), "v"(A[7]), "v"(A[8]), "v"(A[9]),
        "v"(A[10]), "v"(A[11]), "v"(A[12]), "v"(A[13]), "v"(A[14]), "v"(A[15]));
  return m;
}

// lanes l and l ^ 32 (rows 0+2, 1+3) / l and l ^ 16 (rows 0+1, 2+3): both lanes receive the combination of the two values
__device__ __forceinline__ void swap32(float p, float& a, float& b) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_int(p), __float_as_int(p), false, false);
  a = __int_as_float(r[0]); b = __int_as_float(r[1]);
}
__device__ __forceinline__ void swap16(float p, float& a, float& b) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_int(p), __float_as_int(p), false, false);
  a = __int_as_float(r[0]); b = __int_as_float(r[1]);
}
__device__ __forceinline__ float comb_add32(float p) { float a, b; swap32(p, a, b); return a + b; }
__device__ __forceinline__ float comb_add16(float p) { float a, b; swap16(p, a, b); return a + b; }
__device__ __forceinline__ float comb_max32(float p) { float a, b; swap32(p, a, b); return fmaxf(a, b); }
__device__ __forceinline__ float comb_max16(float p) { float a, b; swap16(p, a, b); return fmaxf(a, b); }

// the two arrangements of a 32-vector over the 4 rows of a wave
struct DppGeom {
  int sH, sG;        // the state this lane holds in arrangement H (rows hold halves [0,0,1,1]) / G ([0,1,0,1])
  bool primH, primG; // this lane is the copy that loads / stores the state (every state lives in two lanes)
  bool ok;           // the swap instructions behave as the schedule assumes
};
__device__ __forceinline__ DppGeom dpp_geom(int lane) {
  DppGeom g;
  const int row = lane >> 4, c = lane & 15;
  g.sH = 16 * (row >> 1) + c;
  g.sG = 16 * (row & 1) + c;
  g.primH = (row & 1) == 0;
  g.primG = row < 2;
  const float v = (float)(1 << row);
  const float want32 = (row & 1) ? 10.f : 5.f, want16 = (row >> 1) ? 12.f : 3.f;
  g.ok = __all(comb_add32(v) == want32 && comb_add16(v) == want16);
  return g;
}

// the matrix registers of the two steps.  f(i, j): entry "to i from j" of the operator that is applied (E, E^T, A as the caller
// defines it; i, j in 0..31).  Step A: the lane produces state sG from the inputs of arrangement H; step B: sH from G.
template <class F>
__device__ __forceinline__ void dpp_tables(int lane, const DppGeom& g, F f, float (&TA)[16], float (&TB)[16]) {
  int src[16];
  src[0] = lane;
  src[1] = dpp_ror_i<1>(lane); src[2] = dpp_ror_i<2>(lane); src[3] = dpp_ror_i<3>(lane); src[4] = dpp_ror_i<4>(lane);
  src[5] = dpp_ror_i<5>(lane); src[6] = dpp_ror_i<6>(lane); src[7] = dpp_ror_i<7>(lane); src[8] = dpp_ror_i<8>(lane);
  src[9] = dpp_ror_i<9>(lane); src[10] = dpp_ror_i<10>(lane); src[11] = dpp_ror_i<11>(lane); src[12] = dpp_ror_i<12>(lane);
  src[13] = dpp_ror_i<13>(lane); src[14] = dpp_ror_i<14>(lane); src[15] = dpp_ror_i<15>(lane);
#pragma unroll
  for (int n = 0; n < 16; ++n) {
    const int srow = src[n] >> 4, sc = src[n] & 15;
    TA[n] = f(g.sG, 16 * (srow >> 1) + sc);   // the source lane holds (arrangement H) state 16 (srow >> 1) + sc
    TB[n] = f(g.sH, 16 * (srow & 1) + sc);    // ... (arrangement G) state 16 (srow & 1) + sc
  }
}

// maximum over the lanes that hold the states once: arrangement G -> rows 0, 1; arrangement H -> rows 0, 2 (uniform result)
template <bool ARR_G>
__device__ __forceinline__ float dpp_state_max(float v) {
  asm volatile(
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0"
      : "+v"(v));
  return fmaxf(readlane(v, 0), readlane(v, ARR_G ? 16 : 32));
}

constexpr float kLog2e = 1.44269504088896341f;
constexpr int kFccKClamp = 64;
// the mass of u_t is bounded below by the growth of two frames, each at least exp(-spread of a transition row): rows spread
// over more than this many nats could carry the fp32 vector into the denormals -> such a call runs on the log-domain kernels
constexpr float kFccSafeSpread = 30.f;

__device__ __forceinline__ float ldexp_f32(float v, int e) { return __builtin_amdgcn_ldexpf(v, e); }

// ------------------------------------------------------------------------------------------------ FCC forward
__global__ __launch_bounds__(64) void fcc_fwd_dpp(int T, int N, int scaleMode, const float* __restrict__ x,
                                                  const int* __restrict__ targetSize, const float* __restrict__ trans,
                                                  float* __restrict__ loss, FccWs ws) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const DppGeom g = dpp_geom(lane);
  const float NEG = -INFINITY;

  // rowmax of the two rows of A this lane produces (sG in step A, sH in step B); the spread of row sG for the range check
  float rmG = NEG, rmH = NEG, rnG = INFINITY;
  bool nanRow = false;
#pragma unroll
  for (int j = 0; j < 32; ++j) {
    const float aG = (g.sG < N && j < N) ? trans[(size_t)g.sG * N + j] : NEG;
    const float aH = (g.sH < N && j < N) ? trans[(size_t)g.sH * N + j] : NEG;
    rmG = fmaxf(rmG, aG);
    rmH = fmaxf(rmH, aH);
    if (g.sG < N && j < N) { rnG = fminf(rnG, aG); nanRow = nanRow || aG != aG; }
  }
  {
    const float sp = wave_max(g.sG < N ? rmG - rnG : 0.f);
    const bool risky = __any(nanRow) || !(sp <= kFccSafeSpread);
    if (lane == 0) ws.redo[b] = risky ? 1 : 0;
    if (risky) return;   // fcc_fwd_small, launched behind this kernel, computes the utterance
  }
  float EA[16], EB[16];
  dpp_tables(lane, g, [&](int i, int j) -> float {
    if (j >= N) return 0.f;
    if (i == 31) return 1.f;                       // row 31: the total mass sum_j u[j]
    if (i >= N) return 0.f;
    const float rm = (i == g.sG) ? rmG : rmH;     // i is one of the lane's own two rows
    return __expf(trans[(size_t)i * N + j] - rm);
  }, EA, EB);
  const bool actG = g.sG < N, actH = g.sH < N;
  const float rmlG = actG ? rmG * kLog2e : 0.f, rmlH = actH ? rmH * kLog2e : 0.f;

  const float* xb = x + (size_t)b * T * N;
  float* ub = ws.ahat + (size_t)b * T * N;
  float* qb = ws.logs + (size_t)b * T * N;

  // frame t is held in arrangement G when t is odd, H when t is even (frame 0: H)
  float xc[kDppChunk], xn[kDppChunk];
#pragma unroll
  for (int s = 0; s < kDppChunk; ++s) {
    const bool odd = s & 1;
    const bool act = odd ? actG : actH;
    const int st = odd ? g.sG : g.sH;
    xc[s] = (act && s < T) ? xb[(size_t)s * N + st] : 0.f;
  }
  float u = 0.f;
  double C2 = 0.0;   // sum_t max_t, base-2 units
  int ksum = 0;      // sum_t k_t (scalar unit)
  int k = 0;         // k_t of the frame being computed
  for (int t0 = 0; t0 < T; t0 += kDppChunk) {
#pragma unroll
    for (int s = 0; s < kDppChunk; ++s) {
      const int tn = t0 + kDppChunk + s;
      const bool odd = s & 1;
      const bool act = odd ? actG : actH;
      const int st = odd ? g.sG : g.sH;
      xn[s] = (act && tn < T) ? xb[(size_t)tn * N + st] : 0.f;
    }
    // off the chain, for the whole chunk: P_t = 2^(zz_t - max zz_t), the maxima summed into C2
    float Pc[kDppChunk];
#pragma unroll
    for (int s = 0; s < kDppChunk; ++s) {
      const int t = t0 + s;
      const bool odd = s & 1;
      const bool act = odd ? actG : actH;
      const float rml = t == 0 ? 0.f : (odd ? rmlG : rmlH);
      const float zz = act ? fmaf(xc[s], kLog2e, rml) : NEG;
      const float mz = odd ? dpp_state_max<true>(zz) : dpp_state_max<false>(zz);
      Pc[s] = act ? __builtin_amdgcn_exp2f(zz - mz) : 0.f;
      if (t < T) C2 += (double)mz;
    }
    float us[kDppChunk], qs[kDppChunk];
#pragma unroll
    for (int s = 0; s < kDppChunk; ++s) {
      const int t = t0 + s;
      us[s] = 0.f; qs[s] = 0.f;
      if (t < T) {   // wave-uniform
        if (t == 0) {
          u = Pc[s];
        } else {
          const bool odd = s & 1;
          const float q = ldexp_f32(Pc[s], -k);      // off the chain: k was fixed a frame ago
          float sv;
          if (odd) sv = comb_add32(dpp_dot16(u, EA));   // H -> G
          else sv = comb_add16(dpp_dot16(u, EB));       // G -> H
          u = sv * q;
          qs[s] = q;
          // row 31 of the product = sum_j u_{t-1}[j]: its exponent sets the scale of the NEXT frame (scalar unit)
          const float mass = readlane(sv, odd ? 31 : 47);
          const int e = (int)((__float_as_uint(mass) >> 23) & 0xffu) - 127;
          ksum += k;
          int kn = e - k;
          kn = kn < -kFccKClamp ? -kFccKClamp : (kn > kFccKClamp ? kFccKClamp : kn);
          k = kn;
        }
        us[s] = u;
      }
    }
#pragma unroll
    for (int s = 0; s < kDppChunk; ++s) {
      const int t = t0 + s;
      const bool odd = s & 1;
      const bool st = odd ? (g.primG && actG) : (g.primH && actH);
      const int sx = odd ? g.sG : g.sH;
      if (st && t < T) {
        ub[(size_t)t * N + sx] = us[s];
        qb[(size_t)t * N + sx] = qs[s];
      }
    }
#pragma unroll
    for (int s = 0; s < kDppChunk; ++s) asm volatile("" : "+v"(xn[s]));   // one vmcnt drain per chunk (loads and the frames' stores)
#pragma unroll
    for (int s = 0; s < kDppChunk; ++s) xc[s] = xn[s];
  }
  const bool lastOdd = (T - 1) & 1;
  const bool prim = lastOdd ? (g.primG && actG) : (g.primH && actH);
  const float tot = wave_sum(prim ? u : 0.f);
  const float sc = scale_of(scaleMode, T, targetSize[b]);
  if (lane == 0) {
    const double l = (double)sc * ((C2 + (double)ksum) * 0.69314718055994530942 + (double)__logf(tot));
    loss[b] = g.ok ? (float)l : __builtin_nanf("");
    ws.scale[b] = sc;
  }
}

// ------------------------------------------------------------------------------------------------ FCC backward
__global__ __launch_bounds__(64) void fcc_bwd_dpp(int T, int N, const float* __restrict__ trans, const float* __restrict__ grad,
                                                  float* __restrict__ inputGrad, FccWs ws) {
  const int b = blockIdx.x, lane = threadIdx.x;
  if (ws.redo[b]) return;   // this utterance ran (and will be differentiated) on the log-domain kernels
  const DppGeom g = dpp_geom(lane);
  const float NEG = -INFINITY;
  // E^T: entry "to j from i" = exp(A[i][j] - rowmax_i): the lane needs rowmax of every SOURCE row -> all 32 in LDS
  __shared__ float sRm[32];
  if (lane < 32) {
    float rm = NEG;
#pragma unroll
    for (int j = 0; j < 32; ++j) rm = fmaxf(rm, (lane < N && j < N) ? trans[(size_t)lane * N + j] : NEG);
    sRm[lane] = rm;
  }
  __syncthreads();
  float EA[16], EB[16];
  dpp_tables(lane, g, [&](int j, int i) -> float {   // produces b[j] from r[i]
    if (i >= N || j >= N) return 0.f;
    return __expf(trans[(size_t)i * N + j] - sRm[i]);
  }, EA, EB);
  const bool actG = g.sG < N, actH = g.sH < N;

  const float* __restrict__ ub = ws.ahat + (size_t)b * T * N;
  const float* __restrict__ qb = ws.logs + (size_t)b * T * N;
  float* __restrict__ rb = ws.r + (size_t)b * T * N;
  float* __restrict__ dxb = inputGrad + (size_t)b * T * N;
  const float gsc = ws.scale[b] * grad[b];

  // frame t = thi - s; arrangement of frame t: G when t is odd.  thi stays congruent to T - 1 (mod 2) along the chunks.
  const int par = (T - 1) & 1;
  float uc[kDppChunk], qc[kDppChunk], un[kDppChunk], qn[kDppChunk];
#pragma unroll
  for (int s = 0; s < kDppChunk; ++s) {
    const int t = T - 1 - s;
    const bool odd = (s & 1) != par;
    const bool act = odd ? actG : actH;
    const int st = odd ? g.sG : g.sH;
    uc[s] = (act && t >= 0) ? ub[(size_t)t * N + st] : 0.f;
    qc[s] = (act && t >= 1) ? qb[(size_t)t * N + st] : 0.f;
  }
  // b_{T-1}[i] = 1 / sum_j u_{T-1}[j]
  float bv;
  {
    const bool odd = par;
    const bool prim = odd ? (g.primG && actG) : (g.primH && actH);
    const float tot = wave_sum(prim ? uc[0] : 0.f);
    const bool act = odd ? actG : actH;
    bv = act ? 1.f / tot : 0.f;
  }
  for (int thi = T - 1; thi >= 0; thi -= kDppChunk) {
#pragma unroll
    for (int s = 0; s < kDppChunk; ++s) {
      const int t = thi - kDppChunk - s;
      const bool odd = (s & 1) != par;
      const bool act = odd ? actG : actH;
      const int st = odd ? g.sG : g.sH;
      un[s] = (act && t >= 0) ? ub[(size_t)t * N + st] : 0.f;
      qn[s] = (act && t >= 1) ? qb[(size_t)t * N + st] : 0.f;
    }
    float dxs[kDppChunk], rs[kDppChunk];
#pragma unroll
    for (int s = 0; s < kDppChunk; ++s) {
      const int t = thi - s;
      dxs[s] = 0.f; rs[s] = 0.f;
      if (t >= 0) {   // wave-uniform
        dxs[s] = gsc * (uc[s] * bv);
        if (t >= 1) {
          const bool odd = (s & 1) != par;
          const float r = bv * qc[s];
          rs[s] = r;
          if (odd) bv = comb_add16(dpp_dot16(r, EB));   // frame t in G -> frame t-1 in H
          else bv = comb_add32(dpp_dot16(r, EA));       // H -> G
        }
      }
    }
#pragma unroll
    for (int s = 0; s < kDppChunk; ++s) {
      const int t = thi - s;
      const bool odd = (s & 1) != par;
      const bool st = odd ? (g.primG && actG) : (g.primH && actH);
      const int sx = odd ? g.sG : g.sH;
      if (st && t >= 0) {
        dxb[(size_t)t * N + sx] = g.ok ? dxs[s] : __builtin_nanf("");
        if (t >= 1) rb[(size_t)t * N + sx] = rs[s];
      }
    }
#pragma unroll
    for (int s = 0; s < kDppChunk; ++s) { asm volatile("" : "+v"(un[s]), "+v"(qn[s])); }
#pragma unroll
    for (int s = 0; s < kDppChunk; ++s) { uc[s] = un[s]; qc[s] = qn[s]; }
  }
}

// ------------------------------------------------------------------------------------------------ FCC, two waves per utterance
// tools/micro/clock_probe.hip (profiles/r04_run3_clock_probe.log): a wave alone on its SIMD issues one VALU instruction every
// ~6.5 cycles, dependent or not -- a frame of a one-wave scan costs (instructions it issues) x 6.5 cycles, and fcc_fwd_dpp above
// issues ~67 per frame of which only ~25 are the dependency chain (0.46 ms at T = 2000).  So the utterance gets a SECOND wave on
// another SIMD of the same CU for everything that is not the chain:
//   wave 0 (chain):  q = ldexp(P_t, -k_t) ; u = combine(dpp_dot16(u, E)) * q ; scale bookkeeping ; u, q -> LDS
//   wave 1 (helper): P_t = 2^(zz_t - max zz_t) of the NEXT chunk of 16 frames -> LDS (emission loads, the wave maxima, exp2,
//                    the fp64 sum of the maxima) and the PREVIOUS chunk's u, q from LDS -> workspace
// one s_barrier per chunk of 16 frames (double-buffered LDS rings).  The backward scan splits the same way.
struct FccPair { float u, q; };

__global__ __launch_bounds__(128) void fcc_fwd_dpp2(int T, int N, int scaleMode, const float* __restrict__ x,
                                                    const int* __restrict__ targetSize, const float* __restrict__ trans,
                                                    float* __restrict__ loss, FccWs ws) {
  __shared__ float sP[2][kDppChunk][64];
  __shared__ float sU[2][kDppChunk][2][64];   // [.][.][0] = u_t, [1] = q_t (one ds_write2_b32)
  __shared__ double sC2;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const bool chain = __builtin_amdgcn_readfirstlane(tid >> 6) == 0;
  const DppGeom g = dpp_geom(lane);
  const float NEG = -INFINITY;
  const bool actG = g.sG < N, actH = g.sH < N;

  // rowmax of the two rows of A this lane works with; the spread of row sG for the range check (both waves compute it: uniform)
  float rmG = NEG, rmH = NEG, rnG = INFINITY;
  bool nanRow = false;
#pragma unroll
  for (int j = 0; j < 32; ++j) {
    const float aG = (g.sG < N && j < N) ? trans[(size_t)g.sG * N + j] : NEG;
    const float aH = (g.sH < N && j < N) ? trans[(size_t)g.sH * N + j] : NEG;
    rmG = fmaxf(rmG, aG);
    rmH = fmaxf(rmH, aH);
    if (g.sG < N && j < N) { rnG = fminf(rnG, aG); nanRow = nanRow || aG != aG; }
  }
  {
    const float sp = wave_max(g.sG < N ? rmG - rnG : 0.f);
    const bool risky = __any(nanRow) || !(sp <= kFccSafeSpread);
    if (tid == 0) ws.redo[b] = risky ? 1 : 0;
    if (risky) return;   // fcc_fwd_small, launched behind this kernel, computes the utterance (both waves leave: no barrier yet)
  }
  const float* xb = x + (size_t)b * T * N;

  if (!chain) {
    // ---------------------------------------------------------------- helper wave
    const float rmlG = actG ? rmG * kLog2e : 0.f, rmlH = actH ? rmH * kLog2e : 0.f;
    float* ub = ws.ahat + (size_t)b * T * N;
    float* qb = ws.logs + (size_t)b * T * N;
    float xc[kDppChunk], xn[kDppChunk];
#pragma unroll
    for (int s = 0; s < kDppChunk; ++s) {
      const bool odd = s & 1;
      xc[s] = ((odd ? actG : actH) && s < T) ? xb[(size_t)s * N + (odd ? g.sG : g.sH)] : 0.f;
    }
    double C2 = 0.0;
    auto produce = [&](const float (&xv)[kDppChunk], int t0, int buf) {   // P of frames t0 .. t0 + 15 -> sP[buf]
#pragma unroll
      for (int s = 0; s < kDppChunk; ++s) {
        const int t = t0 + s;
        const bool odd = s & 1;
        const bool act = odd ? actG : actH;
        const float rml = t == 0 ? 0.f : (odd ? rmlG : rmlH);
        const float zz = act ? fmaf(xv[s], kLog2e, rml) : NEG;
        const float mz = odd ? dpp_state_max<true>(zz) : dpp_state_max<false>(zz);
        sP[buf][s][lane] = act ? __builtin_amdgcn_exp2f(zz - mz) : 0.f;
        if (t < T) C2 += (double)mz;
      }
    };
    auto flush = [&](int t0, int buf) {   // u, q of frames t0 .. t0 + 15: LDS -> workspace
#pragma unroll
      for (int s = 0; s < kDppChunk; ++s) {
        const int t = t0 + s;
        const bool odd = s & 1;
        const bool st = odd ? (g.primG && actG) : (g.primH && actH);
        const int sx = odd ? g.sG : g.sH;
        const float uv = sU[buf][s][0][lane], qv = sU[buf][s][1][lane];
        if (st && t < T) {
          ub[(size_t)t * N + sx] = uv;
          qb[(size_t)t * N + sx] = qv;
        }
      }
    };
    produce(xc, 0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    int c = 0;
    for (int t0 = 0; t0 < T; t0 += kDppChunk, ++c) {
#pragma unroll
      for (int s = 0; s < kDppChunk; ++s) {
        const int tn = t0 + 2 * kDppChunk + s;
        const bool odd = s & 1;
        xn[s] = ((odd ? actG : actH) && tn < T) ? xb[(size_t)tn * N + (odd ? g.sG : g.sH)] : 0.f;
      }
      // (xc holds chunk c + 1 from the second iteration on: shift below)
      if (c == 0) {
#pragma unroll
        for (int s = 0; s < kDppChunk; ++s) {
          const int tn = kDppChunk + s;
          const bool odd = s & 1;
          xc[s] = ((odd ? actG : actH) && tn < T) ? xb[(size_t)tn * N + (odd ? g.sG : g.sH)] : 0.f;
        }
      }
      if (t0 + kDppChunk < T) produce(xc, t0 + kDppChunk, (c + 1) & 1);
      if (c >= 1) flush(t0 - kDppChunk, (c - 1) & 1);
#pragma unroll
      for (int s = 0; s < kDppChunk; ++s) asm volatile("" : "+v"(xn[s]));
#pragma unroll
      for (int s = 0; s < kDppChunk; ++s) xc[s] = xn[s];
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    flush((c - 1) * kDppChunk, (c - 1) & 1);
    if (lane == 0) sC2 = C2;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    return;
  }

  // ------------------------------------------------------------------ chain wave
  float EA[16], EB[16];
  dpp_tables(lane, g, [&](int i, int j) -> float {
    if (j >= N) return 0.f;
    if (i == 31) return 1.f;                       // row 31: the total mass sum_j u[j]
    if (i >= N) return 0.f;
    const float rm = (i == g.sG) ? rmG : rmH;
    return __expf(trans[(size_t)i * N + j] - rm);
  }, EA, EB);
  float u = 0.f;
  int ksum = 0, k = 0;
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // P of chunk 0 is in sP[0]
  int c = 0;
  for (int t0 = 0; t0 < T; t0 += kDppChunk, ++c) {
    const int buf = c & 1;
    float Pc[kDppChunk];
#pragma unroll
    for (int s = 0; s < kDppChunk; ++s) Pc[s] = sP[buf][s][lane];
    auto frames = [&](auto full) {   // full: every frame of the chunk exists -- no per-frame bound check in the instruction stream
#pragma unroll
      for (int s = 0; s < kDppChunk; ++s) {
        const int t = t0 + s;
        if (decltype(full)::value || t < T) {   // wave-uniform
          float q = 0.f;
          if (t == 0) {
            u = Pc[s];
          } else {
            const bool odd = s & 1;
            q = ldexp_f32(Pc[s], -k);
            float sv;
            if (odd) sv = comb_add32(dpp_dot16(u, EA));   // H -> G
            else sv = comb_add16(dpp_dot16(u, EB));       // G -> H
            u = sv * q;
            const float mass = readlane(sv, 63);          // row 31 of the product (lane 63 holds state 31 in both arrangements)
            const int e = (int)((__float_as_uint(mass) >> 23) & 0xffu) - 127;
            ksum += k;
            int kn = e - k;
            kn = kn < -kFccKClamp ? -kFccKClamp : (kn > kFccKClamp ? kFccKClamp : kn);
            k = __builtin_amdgcn_readfirstlane(kn);       // (uniform: keep the bookkeeping on the scalar unit)
          }
          sU[buf][s][0][lane] = u;
          sU[buf][s][1][lane] = q;
        }
      }
    };
    if (t0 + kDppChunk <= T) frames(std::true_type{});
    else frames(std::false_type{});
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  const bool lastOdd = (T - 1) & 1;
  const bool prim = lastOdd ? (g.primG && actG) : (g.primH && actH);
  const float tot = wave_sum(prim ? u : 0.f);
  const float sc = scale_of(scaleMode, T, targetSize[b]);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // the helper's sum of the frame maxima is in sC2
  if (lane == 0) {
    const double l = (double)sc * ((sC2 + (double)ksum) * 0.69314718055994530942 + (double)__logf(tot));
    loss[b] = g.ok ? (float)l : __builtin_nanf("");
    ws.scale[b] = sc;
  }
}

__global__ __launch_bounds__(128) void fcc_bwd_dpp2(int T, int N, const float* __restrict__ trans, const float* __restrict__ grad,
                                                    float* __restrict__ inputGrad, FccWs ws) {
  __shared__ float sQ[2][kDppChunk][64];   // q_t of the chunk the chain wave works on (helper -> chain)
  __shared__ float sB[2][kDppChunk][64];   // b_t before the frame's step (chain -> helper)
  __shared__ float sRm[32];
  __shared__ float sTot;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  if (ws.redo[b]) return;   // this utterance ran (and will be differentiated) on the log-domain kernels
  const bool chain = __builtin_amdgcn_readfirstlane(tid >> 6) == 0;
  const DppGeom g = dpp_geom(lane);
  const float NEG = -INFINITY;
  const bool actG = g.sG < N, actH = g.sH < N;
  const float* __restrict__ ub = ws.ahat + (size_t)b * T * N;
  const float* __restrict__ qb = ws.logs + (size_t)b * T * N;
  const int par = (T - 1) & 1;   // frame t = thi - s is held in arrangement G when t is odd: odd(s) = (s & 1) != par
  if (tid < 32) {
    float rm = NEG;
#pragma unroll
    for (int j = 0; j < 32; ++j) rm = fmaxf(rm, (tid < N && j < N) ? trans[(size_t)tid * N + j] : NEG);
    sRm[tid] = rm;
  }
  if (tid == 64) sTot = 0.f;
  __syncthreads();

  if (!chain) {
    // ---------------------------------------------------------------- helper wave: loads u, q; q -> LDS; dx, r from the chain's b
    float* __restrict__ rb = ws.r + (size_t)b * T * N;
    float* __restrict__ dxb = inputGrad + (size_t)b * T * N;
    const float gsc = ws.scale[b] * grad[b];
    float uc[kDppChunk], qc[kDppChunk], un[kDppChunk], qn[kDppChunk];
    auto fetch = [&](float (&uv)[kDppChunk], float (&qv)[kDppChunk], int thi) {
#pragma unroll
      for (int s = 0; s < kDppChunk; ++s) {
        const int t = thi - s;
        const bool odd = (s & 1) != par;
        const bool act = odd ? actG : actH;
        const int st = odd ? g.sG : g.sH;
        uv[s] = (act && t >= 0) ? ub[(size_t)t * N + st] : 0.f;
        qv[s] = (act && t >= 1) ? qb[(size_t)t * N + st] : 0.f;
      }
    };
    fetch(uc, qc, T - 1);
    {   // sum_j u_{T-1}[j] for b_{T-1}
      const bool prim = par ? (g.primG && actG) : (g.primH && actH);
      const float tot = wave_sum(prim ? uc[0] : 0.f);
      if (lane == 0) sTot = tot;
    }
#pragma unroll
    for (int s = 0; s < kDppChunk; ++s) sQ[0][s][lane] = qc[s];
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    int c = 0;
    float up[kDppChunk], qp[kDppChunk];   // the chunk the chain wave has just finished (its b values are in sB[(c - 1) & 1])
    for (int thi = T - 1; thi >= 0; thi -= kDppChunk, ++c) {
      fetch(un, qn, thi - kDppChunk);
#pragma unroll
      for (int s = 0; s < kDppChunk; ++s) sQ[(c + 1) & 1][s][lane] = qn[s];
      if (c >= 1) {
        const int th = thi + kDppChunk;
#pragma unroll
        for (int s = 0; s < kDppChunk; ++s) {
          const int t = th - s;
          const bool odd = (s & 1) != par;
          const bool st = odd ? (g.primG && actG) : (g.primH && actH);
          const int sx = odd ? g.sG : g.sH;
          const float bv = sB[(c - 1) & 1][s][lane];
          if (st && t >= 0) {
            dxb[(size_t)t * N + sx] = g.ok ? gsc * (up[s] * bv) : __builtin_nanf("");
            if (t >= 1) rb[(size_t)t * N + sx] = bv * qp[s];
          }
        }
      }
#pragma unroll
      for (int s = 0; s < kDppChunk; ++s) { up[s] = uc[s]; qp[s] = qc[s]; uc[s] = un[s]; qc[s] = qn[s]; }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    {   // the last chunk
      const int th = T - 1 - (c - 1) * kDppChunk;
#pragma unroll
      for (int s = 0; s < kDppChunk; ++s) {
        const int t = th - s;
        const bool odd = (s & 1) != par;
        const bool st = odd ? (g.primG && actG) : (g.primH && actH);
        const int sx = odd ? g.sG : g.sH;
        const float bv = sB[(c - 1) & 1][s][lane];
        if (st && t >= 0) {
          dxb[(size_t)t * N + sx] = g.ok ? gsc * (up[s] * bv) : __builtin_nanf("");
          if (t >= 1) rb[(size_t)t * N + sx] = bv * qp[s];
        }
      }
    }
    return;
  }

  // ------------------------------------------------------------------ chain wave: b_{t-1} = E^T (b_t q_t)
  float EA[16], EB[16];
  dpp_tables(lane, g, [&](int j, int i) -> float {   // produces b[j] from r[i]
    if (i >= N || j >= N) return 0.f;
    return __expf(trans[(size_t)i * N + j] - sRm[i]);
  }, EA, EB);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // q of the first chunk and sTot are there
  float bv = (par ? actG : actH) ? 1.f / sTot : 0.f;
  int c = 0;
  for (int thi = T - 1; thi >= 0; thi -= kDppChunk, ++c) {
    const int buf = c & 1;
    float qv[kDppChunk];
#pragma unroll
    for (int s = 0; s < kDppChunk; ++s) qv[s] = sQ[buf][s][lane];
    auto frames = [&](auto full) {   // full: the chunk does not reach frame 0 -- no per-frame checks
#pragma unroll
      for (int s = 0; s < kDppChunk; ++s) {
        const int t = thi - s;
        if (decltype(full)::value || t >= 0) {   // wave-uniform
          sB[buf][s][lane] = bv;
          if (decltype(full)::value || t >= 1) {
            const float r = bv * qv[s];
            // frame t in G (t odd) -> frame t - 1 in H through step B; H -> G through step A.  par is uniform: two unrolled bodies
            if (((s & 1) != 0) != (par != 0)) bv = comb_add16(dpp_dot16(r, EB));
            else bv = comb_add32(dpp_dot16(r, EA));
          }
        }
      }
    };
    if (thi - kDppChunk >= 0) frames(std::true_type{});
    else frames(std::false_type{});
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
}

// ------------------------------------------------------------------------------------------------ Viterbi
struct VitDppWs {
  float* delta;         // [B][T][N]
  unsigned char* psi;   // (unused by the dpp path; kept so that the size covers the old layout)
};

__global__ __launch_bounds__(64) void vit_fwd_dpp(int T, int N, const float* __restrict__ x, const float* __restrict__ trans,
                                                  float* __restrict__ deltaAll) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const DppGeom g = dpp_geom(lane);
  const float NEG = -INFINITY;
  float AA[16], AB[16];
  dpp_tables(lane, g, [&](int i, int j) -> float { return (i < N && j < N) ? trans[(size_t)i * N + j] : NEG; }, AA, AB);
  const bool actG = g.sG < N, actH = g.sH < N;
  const float* xb = x + (size_t)b * T * N;
  float* db = deltaAll + (size_t)b * T * N;

  float xc[kDppChunk], xn[kDppChunk];
#pragma unroll
  for (int s = 0; s < kDppChunk; ++s) {
    const bool odd = s & 1;
    const bool act = odd ? actG : actH;
    const int st = odd ? g.sG : g.sH;
    xc[s] = (act && s < T) ? xb[(size_t)s * N + st] : 0.f;
  }
#pragma unroll
  for (int s = 0; s < kDppChunk; ++s) asm volatile("" : "+v"(xc[s]));   // landed before the loop: no pending load at its head
  float d = NEG;
  for (int t0 = 0; t0 < T; t0 += kDppChunk) {
#pragma unroll
    for (int s = 0; s < kDppChunk; ++s) {
      const int tn = t0 + kDppChunk + s;
      const bool odd = s & 1;
      const bool act = odd ? actG : actH;
      const int st = odd ? g.sG : g.sH;
      xn[s] = (act && tn < T) ? xb[(size_t)tn * N + st] : 0.f;
    }
    float ds[kDppChunk];
    auto frames = [&](auto full) {
#pragma unroll
      for (int s = 0; s < kDppChunk; ++s) {
        const int t = t0 + s;
        ds[s] = NEG;
        if (decltype(full)::value || t < T) {
          const bool odd = s & 1;
          const bool act = odd ? actG : actH;
          if (t == 0) {
            d = act ? xc[s] : NEG;
          } else {
            float best;
            if (odd) best = comb_max32(dpp_maxplus16(d, AA));
            else best = comb_max16(dpp_maxplus16(d, AB));
            d = act ? best + xc[s] : NEG;
          }
          ds[s] = d;
        }
      }
    };
    if (t0 + kDppChunk <= T) frames(std::true_type{});
    else frames(std::false_type{});
    // the prefetched chunk is consumed BEFORE the frames' stores are issued: hipcc waits vmcnt(0) at the first use of a loaded
    // register when stores may have been issued behind the load (the counter is shared and in order), i.e. a store round trip
    // per chunk if the stores come first -- this way the loads have had the whole chunk to land and the stores drain meanwhile
#pragma unroll
    for (int s = 0; s < kDppChunk; ++s) asm volatile("" : "+v"(xn[s]));
#pragma unroll
    for (int s = 0; s < kDppChunk; ++s) {
      const int t = t0 + s;
      const bool odd = s & 1;
      const bool st = odd ? (g.primG && actG) : (g.primH && actH);
      const int sx = odd ? g.sG : g.sH;
      if (st && t < T) db[(size_t)t * N + sx] = g.ok ? ds[s] : __builtin_nanf("");
    }
#pragma unroll
    for (int s = 0; s < kDppChunk; ++s) xc[s] = xn[s];
  }
}

// back-pointers from the stored delta rows + the walk, in two parallel kernels (a frame of a serial walk is a dependent LDS
// read: 2000 of them cost more than the whole scan -- vit_bt_k of the first version, 520 us at T = 2000):
//   vit_psi_k   grid (chunks of kVpChunk frames, B): psi_t[i] = first j maximising delta_{t-1}[j] + A[i][j] (fp32 sums, strict '>'
//               upwards in j: the oracle's scan) for the chunk's frames -> psi bytes [B][T][32]; then the chunk's back-pointers
//               are COMPOSED: comp_c[i] = the state at the chunk's first frame - 1 when the path is in state i at its last frame
//               (kVpChunk dependent LDS reads, every chunk of every utterance in parallel);
//   vit_walk_k  one wave per utterance: the end state of every chunk by walking the composed maps (T / kVpChunk dependent
//               steps), then every lane walks ONE chunk from its end state through the psi bytes staged in LDS.
// Serial depth 2 * kVpChunk + T / kVpChunk steps instead of T.
constexpr int kVpChunk = 32;
struct VitBtWs {
  unsigned char* psi;    // [B][T][32]
  unsigned char* comp;   // [B][nChunks][32]
};
__host__ __device__ inline int vit_chunks(int T) { return (T - 1 + kVpChunk - 1) / kVpChunk; }   // frames 1 .. T-1 in chunks
__host__ __device__ inline VitBtWs vit_bt_ws(void* base, int B, int T, int N) {
  VitBtWs w;
  char* p = (char*)base + align_up((size_t)B * T * N * sizeof(float), 256);   // behind the delta rows
  w.psi = (unsigned char*)p; p += align_up((size_t)B * T * 32, 256);
  w.comp = (unsigned char*)p;
  return w;
}
__host__ __device__ inline size_t vit_dpp_ws_bytes(int B, int T, int N) {
  return align_up((size_t)B * T * N * sizeof(float), 256) + align_up((size_t)B * T * 32, 256) +
         align_up((size_t)B * (size_t)(vit_chunks(T) + 1) * 32, 256);
}

// chunk c covers frames t = 1 + c * kVpChunk .. min(T - 1, (c + 1) * kVpChunk)
__global__ __launch_bounds__(64) void vit_psi_k(int T, int N, const float* __restrict__ trans, const float* __restrict__ deltaAll,
                                                VitBtWs ws) {
  __shared__ float sA[32 * 33];
  __shared__ float sD[kVpChunk * 32];          // delta rows t - 1 of the chunk's frames
  __shared__ unsigned char sPsi[kVpChunk * 32];
  const int c = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
  const int tlo = 1 + c * kVpChunk;
  int thi = tlo + kVpChunk - 1;
  if (thi > T - 1) thi = T - 1;
  const int nf = thi - tlo + 1;
  const float* db = deltaAll + (size_t)b * T * N;
  for (int e = lane; e < N * N; e += 64) sA[(e / N) * 33 + (e % N)] = trans[e];
  for (int e = lane; e < nf * N; e += 64) sD[(e / N) * 32 + (e % N)] = db[(size_t)(tlo - 1) * N + e];
  __syncthreads();
  // two frames at a time: lanes 0..31 the even, 32..63 the odd frame of a pair; lane & 31 = state i
  const int i = lane & 31, half = lane >> 5;
  for (int f = half; f < nf; f += 2) {
    if (i < N) {
      const float* dr = sD + f * 32;
      const float* ar = sA + i * 33;
      float best = dr[0] + ar[0];
      int arg = 0;
      for (int j = 1; j < N; ++j) {
        const float v = dr[j] + ar[j];
        if (v > best) { best = v; arg = j; }
      }
      sPsi[f * 32 + i] = (unsigned char)arg;
    }
  }
  __syncthreads();
  unsigned char* pg = ws.psi + ((size_t)b * T + tlo) * 32;
  for (int e = lane; e < nf * 32; e += 64) pg[e] = sPsi[e];
  if (lane < 32) {   // compose from the chunk's last frame down: state at frame tlo - 1 given state `lane` at frame thi
    int cur = lane < N ? lane : 0;
    for (int f = nf - 1; f >= 0; --f) cur = sPsi[f * 32 + cur];
    ws.comp[((size_t)b * (vit_chunks(T) + 1) + c) * 32 + lane] = (unsigned char)cur;
  }
}

// stage = 1: the utterance's psi bytes fit in LDS (T <= ~4800) and are walked there; 0: walked in global memory
__global__ __launch_bounds__(64) void vit_walk_k(int T, int N, int stage, const float* __restrict__ deltaAll, VitBtWs ws,
                                                 int* __restrict__ path) {
  extern __shared__ unsigned char sm[];   // [psi bytes of the utterance [T][32],] comp [nChunks][32], end states [nChunks + 1] ints
  const int b = blockIdx.x, lane = threadIdx.x;
  const int nC = vit_chunks(T);
  const size_t psiBytes = stage ? (size_t)T * 32 : 0;
  unsigned char* sComp = sm + psiBytes;
  int* sEnd = (int*)(sm + ((psiBytes + (size_t)nC * 32 + 15) & ~(size_t)15));
  const unsigned char* sPsi = stage ? sm : ws.psi + (size_t)b * T * 32;
  {   // stage (16 B per lane and load)
    const uint4* src = (const uint4*)(ws.psi + (size_t)b * T * 32);
    uint4* dst = (uint4*)sm;
    if (stage)
      for (int e = lane; e < T * 2; e += 64) dst[e] = src[e];
    const uint4* cs = (const uint4*)(ws.comp + (size_t)b * (nC + 1) * 32);
    uint4* cd = (uint4*)sComp;
    for (int e = lane; e < nC * 2; e += 64) cd[e] = cs[e];
  }
  // final state: first argmax_i delta_{T-1}[i]
  const float* db = deltaAll + (size_t)b * T * N;
  const float v = lane < N ? db[(size_t)(T - 1) * N + lane] : -INFINITY;
  const float m = wave_max(v);
  const unsigned long long eq = __ballot(lane < N && v == m);
  int cur = eq ? __ffsll((long long)eq) - 1 : 0;
  __syncthreads();
  if (lane == 0) {   // end state of chunk c = state at frame min(T - 1, (c + 1) kVpChunk); sEnd[c] for c = nC - 1 .. 0, sEnd[-1] -> frame 0
    for (int c = nC - 1; c >= 0; --c) {
      sEnd[c + 1] = cur;
      cur = sComp[c * 32 + cur];
    }
    sEnd[0] = cur;   // the state at frame 0
  }
  __syncthreads();
  int* pb = path + (size_t)b * T;
  if (lane == 0) pb[0] = sEnd[0];
  for (int c = lane; c < nC; c += 64) {
    const int tlo = 1 + c * kVpChunk;
    int thi = tlo + kVpChunk - 1;
    if (thi > T - 1) thi = T - 1;
    int s = sEnd[c + 1];
    for (int t = thi; t >= tlo; --t) {
      pb[t] = s;
      s = sPsi[(size_t)t * 32 + s];
    }
  }
}

}  // namespace w2l
