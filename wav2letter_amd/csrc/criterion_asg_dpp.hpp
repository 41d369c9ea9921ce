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
// strict '>' scan -- are recomputed from the stored delta rows by the parallel kernel vit_bt_k, which also walks the path.
#pragma once
#include "common.hpp"

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
#pragma unroll
    for (int s = 0; s < kDppChunk; ++s) {
      const int t = t0 + s;
      ds[s] = NEG;
      if (t < T) {
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
#pragma unroll
    for (int s = 0; s < kDppChunk; ++s) {
      const int t = t0 + s;
      const bool odd = s & 1;
      const bool st = odd ? (g.primG && actG) : (g.primH && actH);
      const int sx = odd ? g.sG : g.sH;
      if (st && t < T) db[(size_t)t * N + sx] = g.ok ? ds[s] : __builtin_nanf("");
    }
#pragma unroll
    for (int s = 0; s < kDppChunk; ++s) asm volatile("" : "+v"(xn[s]));
#pragma unroll
    for (int s = 0; s < kDppChunk; ++s) xc[s] = xn[s];
  }
}

// back-pointers from the stored delta rows + the walk.  One workgroup per utterance, chunks of kVbChunk frames from the end:
// psi_t[i] = first j maximising delta_{t-1}[j] + A[i][j] (fp32 sums, strict '>' upwards in j: the oracle's scan), then lane 0
// follows the path through the chunk.
constexpr int kVbChunk = 128;
__global__ __launch_bounds__(256) void vit_bt_k(int T, int N, const float* __restrict__ trans, const float* __restrict__ deltaAll,
                                                int* __restrict__ path) {
  __shared__ float sA[32 * 33];
  __shared__ float sD[(kVbChunk + 1) * 32];
  __shared__ unsigned char sPsi[kVbChunk * 32];
  __shared__ int sPath[kVbChunk];
  __shared__ int sCur;
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* db = deltaAll + (size_t)b * T * N;
  int* pb = path + (size_t)b * T;
  for (int e = tid; e < N * N; e += 256) sA[(e / N) * 33 + (e % N)] = trans[e];
  if (tid < 64) {   // final state: first argmax_i delta_{T-1}[i]
    const float v = tid < N ? db[(size_t)(T - 1) * N + tid] : -INFINITY;
    const float m = wave_max(v);
    const unsigned long long eq = __ballot(tid < N && v == m);
    if (tid == 0) sCur = eq ? __ffsll((long long)eq) - 1 : 0;
  }
  __syncthreads();
  for (int thi = T - 1; thi >= 0; thi -= kVbChunk) {
    int tlo = thi - kVbChunk + 1;
    if (tlo < 0) tlo = 0;
    const int nst = thi - tlo + 1;
    // delta rows tlo-1 .. thi-1 (row r of sD = frame tlo - 1 + r)
    const int r0 = tlo >= 1 ? 0 : 1;
    for (int e = tid; e < (nst + 1 - r0) * N; e += 256) {
      const int r = r0 + e / N, j = e % N;
      sD[r * 32 + j] = db[(size_t)(tlo - 1 + r) * N + j];
    }
    __syncthreads();
    for (int e = tid; e < nst * N; e += 256) {
      const int tt = e / N, i = e % N;   // frame tlo + tt, previous frame = row tt of sD
      int arg = 0;
      if (tlo + tt >= 1) {
        const float* dr = sD + tt * 32;
        const float* ar = sA + i * 33;
        float best = dr[0] + ar[0];
        for (int j = 1; j < N; ++j) {
          const float v = dr[j] + ar[j];
          if (v > best) { best = v; arg = j; }
        }
      }
      sPsi[tt * 32 + i] = (unsigned char)arg;
    }
    __syncthreads();
    if (tid == 0) {
      int cur = sCur;
      for (int t = thi; t >= tlo; --t) {
        sPath[t - tlo] = cur;
        if (t >= 1) cur = sPsi[(t - tlo) * 32 + cur];
      }
      sCur = cur;
    }
    __syncthreads();
    for (int e = tid; e < nst; e += 256) pb[tlo + e] = sPath[e];
    __syncthreads();
  }
}

}  // namespace w2l
