// criterion_asg_mitm.hpp -- FullConnectionCriterion for N <= 31 states, MEET IN THE MIDDLE (round 6); included by criterion_fcc.hip
// behind criterion_asg_dpp.hpp, whose machinery (scaled linear domain, DPP row-rotation products, the two arrangements of a 32-vector,
// chain wave + helper wave) it reuses.
//
// Replaces fl::lib::{cpu,cuda}::FullConnectionCriterion<float> (un-vendored; call sites recipes/slimIPL/src/Train.cpp:408-410,
// :1675; math SURVEY.md App. B.2; CPU restatement oracle/criterion_oracle.c).
//
// The scans of criterion_asg_dpp.hpp are T dependent frames per pass: at T = 2000 the forward pass (alpha, 0 -> T-1) took 310 us
// and the backward pass (beta, T-1 -> 0) 330 us on 64 + 64 waves of an otherwise idle chip.  alpha and beta are independent
// recursions, so each pass runs BOTH, from the two ends to the middle frame m = (T - 1) / 2, in two workgroups per utterance:
//   forward  (fcc_mitm_fwd):  block 0: u_t = q_t . (E u_{t-1}),   t = 0 .. m       (alpha: exactly fcc_fwd_dpp2, stopped at m)
//                             block 1: b_{t-1} = E^T (b_t . q'_t), t = T-1 .. m+1   (beta with a scale sequence of its own)
//                             loss = scale ((S_m + S'_m) ln 2 + log sum_i u_m[i] b_m[i])   (fcc_mitm_finish)
//   backward (fcc_mitm_bwd):  block 0: b_{t-1} = E^T (b_t . q_t),  t = m .. 1       (beta continued on the ALPHA half's scales)
//                             block 1: u_t = q'_t . (E u_{t-1}),   t = m+1 .. T-1   (alpha continued on the BETA half's scales)
// Continuing each recursion on the OTHER half's scale sequence keeps sum_i u_t[i] b_t[i] = G (the middle frame's value) at every
// frame, so the posterior is u_t b_t / G with no per-frame normaliser, the continuation chains carry no scale bookkeeping at all, and
//   d loss / d x_t = g u_t b_t / G,   r_t = b_t q_t / G  (the operand of the transition-gradient kernel fcc_dtrans_mfma, unchanged).
// T / 2 dependent frames per pass instead of T, 2 B workgroups instead of B.  Workspace per frame: ahat[t] = u_t, logs[t] = the scale
// q the frame's u_t was produced with (q_t for t <= m, q'_t above), r[t] = r_t; r2[t] = b_t q'_t of the forward pass (t > m).
#pragma once
#include "criterion_asg_dpp.hpp"

namespace w2l {

// the middle frame.  Not the centre: per frame the beta half of the forward pass costs 158 ns against the alpha half's 136 (its
// emission rows, scale rows and r rows run against the address order) and the continuations 135 / 148 ns
// (profiles/r06_run10_asg_mitm_halves_after_fixes.log), so alpha takes 8 / 15 of the frames
__host__ __device__ inline int fcc_mitm_mid(int T) { return (int)(((long long)(T - 1) * 8) / 15); }

// E: entry "to i from j" of the forward operator (row 31: the total mass when MASS); ET: "to j from i" of the backward one
struct MitmRows {
  float rmG, rmH;   // row maxima of the two rows of A this lane produces (sG in step A, sH in step B)
  bool risky;
};

__device__ __forceinline__ MitmRows mitm_rows(int lane, const DppGeom& g, int N, const float* __restrict__ trans) {
  const float NEG = -INFINITY;
  MitmRows r;
  r.rmG = NEG; r.rmH = NEG;
  float rnG = INFINITY;
  bool nanRow = false;
#pragma unroll
  for (int j = 0; j < 32; ++j) {
    const float aG = (g.sG < N && j < N) ? trans[(size_t)g.sG * N + j] : NEG;
    const float aH = (g.sH < N && j < N) ? trans[(size_t)g.sH * N + j] : NEG;
    r.rmG = fmaxf(r.rmG, aG);
    r.rmH = fmaxf(r.rmH, aH);
    if (g.sG < N && j < N) { rnG = fminf(rnG, aG); nanRow = nanRow || aG != aG; }
  }
  const float sp = wave_max(g.sG < N ? r.rmG - rnG : 0.f);
  r.risky = __any(nanRow) || !(sp <= kFccSafeSpread);
  return r;
}

#define W2L_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// ------------------------------------------------------------------------------------------------ forward: the two halves
// (the body is also a role of the one-launch forward pass of the ASG criterion, asg_mitm_fwd in criterion_asg.hip: `dir` = which half,
//  threads 0 .. 127 of the workgroup)
__device__ __forceinline__ void fcc_mitm_fwd_body(int T, int N, const float* __restrict__ x, const float* __restrict__ trans, const FccWs& ws, int dir) {
  __shared__ float sP[2][kDppChunk][64];
  __shared__ float sU[2][kDppChunk][2][64];   // chain -> helper: [0] = u_t (alpha) / r_t = b_t q'_t (beta), [1] = the frame's scale q
  __shared__ double sC2;
  __shared__ float sRm[32];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const bool chain = __builtin_amdgcn_readfirstlane(tid >> 6) == 0;
  const DppGeom g = dpp_geom(lane);
  const float NEG = -INFINITY;
  const bool actG = g.sG < N, actH = g.sH < N;
  const MitmRows rows = mitm_rows(lane, g, N, trans);
  if (dir == 0 && tid == 0) ws.redo[b] = rows.risky ? 1 : 0;
  if (rows.risky) return;   // fcc_fwd_log, launched behind this kernel, computes the utterance (every wave leaves: no barrier yet)
  const int m = fcc_mitm_mid(T);
  const float* xb = x + (size_t)b * T * N;
  const float rmlG = actG ? rows.rmG * kLog2e : 0.f, rmlH = actH ? rows.rmH * kLog2e : 0.f;

  if (dir == 0) {
    // ================================================================ alpha: frames 0 .. m (fcc_fwd_dpp2 with the end at m)
    const int TE = m + 1;
    if (!chain) {
      float* ub = ws.ahat + (size_t)b * T * N;
      float* qb = ws.logs + (size_t)b * T * N;
      float xc[kDppChunk], xn[kDppChunk];
#pragma unroll
      for (int s = 0; s < kDppChunk; ++s) {
        const bool odd = s & 1;
        xc[s] = ((odd ? actG : actH) && s < TE) ? xb[(size_t)s * N + (odd ? g.sG : g.sH)] : 0.f;
      }
      double C2 = 0.0;
      auto produce = [&](const float (&xv)[kDppChunk], int t0, int buf) {
#pragma unroll
        for (int s = 0; s < kDppChunk; ++s) {
          const int t = t0 + s;
          const bool odd = s & 1;
          const bool act = odd ? actG : actH;
          const float rml = t == 0 ? 0.f : (odd ? rmlG : rmlH);
          const float zz = act ? fmaf(xv[s], kLog2e, rml) : NEG;
          const float mz = odd ? dpp_state_max<true>(zz) : dpp_state_max<false>(zz);
          sP[buf][s][lane] = act ? __builtin_amdgcn_exp2f(zz - mz) : 0.f;
          if (t < TE) C2 += (double)mz;
        }
      };
      auto flush = [&](int t0, int buf) {
#pragma unroll
        for (int s = 0; s < kDppChunk; ++s) {
          const int t = t0 + s;
          const bool odd = s & 1;
          const bool st = odd ? (g.primG && actG) : (g.primH && actH);
          const int sx = odd ? g.sG : g.sH;
          const float uv = sU[buf][s][0][lane], qv = sU[buf][s][1][lane];
          if (st && t < TE) {
            ub[(size_t)t * N + sx] = uv;
            qb[(size_t)t * N + sx] = qv;
          }
        }
      };
      produce(xc, 0, 0);
      W2L_LDS_BARRIER();
      int c = 0;
      for (int t0 = 0; t0 < TE; t0 += kDppChunk, ++c) {
#pragma unroll
        for (int s = 0; s < kDppChunk; ++s) {
          const int tn = t0 + 2 * kDppChunk + s;
          const bool odd = s & 1;
          xn[s] = ((odd ? actG : actH) && tn < TE) ? xb[(size_t)tn * N + (odd ? g.sG : g.sH)] : 0.f;
        }
        if (c == 0) {
#pragma unroll
          for (int s = 0; s < kDppChunk; ++s) {
            const int tn = kDppChunk + s;
            const bool odd = s & 1;
            xc[s] = ((odd ? actG : actH) && tn < TE) ? xb[(size_t)tn * N + (odd ? g.sG : g.sH)] : 0.f;
          }
        }
        if (t0 + kDppChunk < TE) produce(xc, t0 + kDppChunk, (c + 1) & 1);
        if (c >= 1) flush(t0 - kDppChunk, (c - 1) & 1);
#pragma unroll
        for (int s = 0; s < kDppChunk; ++s) asm volatile("" : "+v"(xn[s]));
#pragma unroll
        for (int s = 0; s < kDppChunk; ++s) xc[s] = xn[s];
        W2L_LDS_BARRIER();
      }
      flush((c - 1) * kDppChunk, (c - 1) & 1);
      if (lane == 0) sC2 = C2;
      W2L_LDS_BARRIER();
      return;
    }
    float EA[16], EB[16];
    dpp_tables(lane, g, [&](int i, int j) -> float {
      if (j >= N) return 0.f;
      if (i == 31) return 1.f;
      if (i >= N) return 0.f;
      const float rm = (i == g.sG) ? rows.rmG : rows.rmH;
      return __expf(trans[(size_t)i * N + j] - rm);
    }, EA, EB);
    float u = 0.f;
    int ksum = 0, k = 0;
    W2L_LDS_BARRIER();
    int c = 0;
    for (int t0 = 0; t0 < TE; t0 += kDppChunk, ++c) {
      const int buf = c & 1;
      float Pc[kDppChunk];
#pragma unroll
      for (int s = 0; s < kDppChunk; ++s) Pc[s] = sP[buf][s][lane];
      auto frames = [&](auto full) {
#pragma unroll
        for (int s = 0; s < kDppChunk; ++s) {
          const int t = t0 + s;
          if (decltype(full)::value || t < TE) {
            float q = 0.f;
            if (t == 0) {
              u = Pc[s];
            } else {
              const bool odd = s & 1;
              q = ldexp_f32(Pc[s], -k);
              float sv;
              if (odd) sv = comb_add32(dpp_dot16(u, EA));
              else sv = comb_add16(dpp_dot16(u, EB));
              u = sv * q;
              const float mass = readlane(sv, 63);
              const int e = (int)((__float_as_uint(mass) >> 23) & 0xffu) - 127;
              ksum += k;
              int kn = e - k;
              kn = kn < -kFccKClamp ? -kFccKClamp : (kn > kFccKClamp ? kFccKClamp : kn);
              k = __builtin_amdgcn_readfirstlane(kn);
            }
            sU[buf][s][0][lane] = u;
            sU[buf][s][1][lane] = q;
          }
        }
      };
      if (t0 + kDppChunk <= TE) frames(std::true_type{});
      else frames(std::false_type{});
      W2L_LDS_BARRIER();
    }
    W2L_LDS_BARRIER();   // the helper's sum of the frame maxima is in sC2
    if (lane == 0) ws.half[2 * b] = g.ok ? sC2 + (double)ksum : (double)__builtin_nanf("");
    return;
  }

  // ================================================================== beta: frames T-1 .. m
  // The chain carries r_t = b_t q'_t, which obeys the SAME recursion as the alpha half's u_t with E^T for E and time reversed:
  //   r_{T-1} = P_{T-1},   r_t = q'_t . (E^T r_{t+1}),   q'_t = P_t 2^-k_t,   k_t fixed one frame earlier from sum_i r_{t+1}[i]
  // (carrying b_t instead -- b_{t-1} = E^T (b_t q'_t) -- puts the scale bookkeeping, a readlane and four scalar instructions, ON the
  // dependency chain: measured 228 ns per frame against 141).  b_t = E^T r_{t+1} is the product before the scaling; only b_m is kept.
  // Frames are counted j = 0 .. nF - 1, frame t = T - 1 - j (t >= m + 1); one more product gives b_m.
  // (the arrangement of a frame depends on its parity: the parity of the half's first frame is a COMPILE-TIME constant of the role
  //  bodies below, dispatched once -- with a run-time parity every frame of helper and chain carried a scalar branch and the
  //  helper twice the instructions: 228 ns per frame against the alpha half's 141)
  const int nF = T - 1 - m;
  auto betaRole = [&](auto parTag) {
    constexpr bool PAR = decltype(parTag)::value;   // frame t_j = T - 1 - j is held in arrangement G when it is odd: odd(s) = (s & 1) != PAR
  if (!chain) {
    float* rb = ws.r2 + (size_t)b * T * N;
    float* qb = ws.logs + (size_t)b * T * N;
    float xc[kDppChunk], xn[kDppChunk];
    auto fetch = [&](float (&xv)[kDppChunk], int j0) {
#pragma unroll
      for (int s = 0; s < kDppChunk; ++s) {
        const int t = T - 1 - (j0 + s);
        const bool odd = ((s & 1) != 0) != PAR;
        xv[s] = ((odd ? actG : actH) && j0 + s < nF) ? xb[(size_t)t * N + (odd ? g.sG : g.sH)] : 0.f;
      }
    };
    double C2 = 0.0;
    auto produce = [&](const float (&xv)[kDppChunk], int j0, int buf) {
#pragma unroll
      for (int s = 0; s < kDppChunk; ++s) {
        const bool odd = ((s & 1) != 0) != PAR;
        const bool act = odd ? actG : actH;
        const float zz = act ? fmaf(xv[s], kLog2e, odd ? rmlG : rmlH) : NEG;   // (every frame here is >= m + 1 >= 1)
        const float mz = odd ? dpp_state_max<true>(zz) : dpp_state_max<false>(zz);
        sP[buf][s][lane] = act ? __builtin_amdgcn_exp2f(zz - mz) : 0.f;
        if (j0 + s < nF) C2 += (double)mz;
      }
    };
    auto flush = [&](int j0, int buf) {
#pragma unroll
      for (int s = 0; s < kDppChunk; ++s) {
        const int t = T - 1 - (j0 + s);
        const bool odd = ((s & 1) != 0) != PAR;
        const bool st = odd ? (g.primG && actG) : (g.primH && actH);
        const int sx = odd ? g.sG : g.sH;
        const float rv = sU[buf][s][0][lane], qv = sU[buf][s][1][lane];
        if (st && j0 + s < nF) {
          rb[(size_t)t * N + sx] = rv;
          qb[(size_t)t * N + sx] = qv;
        }
      }
    };
    fetch(xc, 0);
    produce(xc, 0, 0);
    fetch(xc, kDppChunk);
    W2L_LDS_BARRIER();
    int c = 0;
    for (int j0 = 0; j0 < nF; j0 += kDppChunk, ++c) {
      fetch(xn, j0 + 2 * kDppChunk);
      if (j0 + kDppChunk < nF) produce(xc, j0 + kDppChunk, (c + 1) & 1);
      if (c >= 1) flush(j0 - kDppChunk, (c - 1) & 1);
#pragma unroll
      for (int s = 0; s < kDppChunk; ++s) asm volatile("" : "+v"(xn[s]));
#pragma unroll
      for (int s = 0; s < kDppChunk; ++s) xc[s] = xn[s];
      W2L_LDS_BARRIER();
    }
    if (c >= 1) flush((c - 1) * kDppChunk, (c - 1) & 1);
    if (lane == 0) sC2 = C2;
    W2L_LDS_BARRIER();
    return;
  }
  if (tid < 32) {   // (the chain wave is wave 0: lanes 0 .. 31) row maxima of every source row of E^T
    float rm = NEG;
#pragma unroll
    for (int j = 0; j < 32; ++j) rm = fmaxf(rm, (tid < N && j < N) ? trans[(size_t)tid * N + j] : NEG);
    sRm[tid] = rm;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // one wave: the LDS executes its operations in order
  float EA[16], EB[16];
  dpp_tables(lane, g, [&](int j, int i) -> float {   // produces b[j] from r[i]; row 31: the total mass sum_i r[i]
    if (i >= N) return 0.f;
    if (j == 31) return 1.f;
    if (j >= N) return 0.f;
    return __expf(trans[(size_t)i * N + j] - sRm[i]);
  }, EA, EB);
  float r = (PAR ? actG : actH) ? 1.f : 0.f;   // (T = 1: no frame above the middle; b_m = beta_{T-1} = 1)
  int ksum = 0, k = 0;
  W2L_LDS_BARRIER();
  int c = 0;
  for (int j0 = 0; j0 < nF; j0 += kDppChunk, ++c) {
    const int buf = c & 1;
    float Pc[kDppChunk];
#pragma unroll
    for (int s = 0; s < kDppChunk; ++s) Pc[s] = sP[buf][s][lane];
    auto frames = [&](auto full) {
#pragma unroll
      for (int s = 0; s < kDppChunk; ++s) {
        const int j = j0 + s;
        if (decltype(full)::value || j < nF) {
          float q;
          if (j == 0) {
            q = Pc[s];   // r_{T-1} = b_{T-1} q'_{T-1} with b = 1, k = 0
            r = q;
          } else {
            q = ldexp_f32(Pc[s], -k);
            float sv;
            // the frame produced is odd: H -> G through step A; even: G -> H through step B (as the alpha half)
            if (((s & 1) != 0) != PAR) sv = comb_add32(dpp_dot16(r, EA));
            else sv = comb_add16(dpp_dot16(r, EB));
            r = sv * q;
            const float mass = readlane(sv, 63);   // state 31 = sum_i r_{t+1}[i] (its own q is 0: it takes no part in the next product)
            const int e = (int)((__float_as_uint(mass) >> 23) & 0xffu) - 127;
            ksum += k;
            int kn = e - k;
            kn = kn < -kFccKClamp ? -kFccKClamp : (kn > kFccKClamp ? kFccKClamp : kn);
            k = __builtin_amdgcn_readfirstlane(kn);
          }
          sU[buf][s][0][lane] = r;
          sU[buf][s][1][lane] = q;
        }
      }
    };
    if (j0 + kDppChunk <= nF) frames(std::true_type{});
    else frames(std::false_type{});
    W2L_LDS_BARRIER();
  }
  W2L_LDS_BARRIER();
  {   // b_m = E^T r_{m+1} (frame m's arrangement), by state, and the half's scale sum
    const bool odd = m & 1;
    float bm = r;
    if (nF > 0) bm = odd ? comb_add32(dpp_dot16(r, EA)) : comb_add16(dpp_dot16(r, EB));
    const bool st = odd ? (g.primG && actG) : (g.primH && actH);
    if (st) ws.bm[(size_t)b * 32 + (odd ? g.sG : g.sH)] = bm;
    if (lane == 0) ws.half[2 * b + 1] = g.ok ? sC2 + (double)ksum : (double)__builtin_nanf("");
  }
  };
  if ((T - 1) & 1) betaRole(std::true_type{});
  else betaRole(std::false_type{});
}

// loss, scale and 1 / G of the utterances that stayed on the linear-domain halves (one wave per utterance); launched as the
// "not flagged" branch of fcc_fwd_log's grid so that the forward call stays at two launches
__device__ __forceinline__ void fcc_mitm_finish(int b, int lane, int T, int N, int scaleMode, const int* __restrict__ targetSize,
                                                float* __restrict__ loss, const FccWs& ws) {
  const int m = fcc_mitm_mid(T);
  const float u = lane < N ? ws.ahat[((size_t)b * T + m) * N + lane] : 0.f;
  const float bm = lane < N ? ws.bm[(size_t)b * 32 + lane] : 0.f;
  const float G = wave_sum(u * bm);
  const float sc = scale_of(scaleMode, T, targetSize[b]);
  if (lane == 0) {
    const double S = ws.half[2 * b] + ws.half[2 * b + 1];
    loss[b] = (float)((double)sc * (S * 0.69314718055994530942 + (double)__logf(G)));
    ws.scale[b] = sc;
    ws.ginv[b] = 1.f / G;
  }
}

__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(1, 2))) void fcc_mitm_fwd(int T, int N, const float* __restrict__ x, const float* __restrict__ trans, FccWs ws,
                                                    int dir0 = 0 /* probe: time one half alone (grid (B, 1)) */) {
  fcc_mitm_fwd_body(T, N, x, trans, ws, (int)blockIdx.y + dir0);
}

// ------------------------------------------------------------------------------------------------ backward: the two continuations
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(1, 2))) void fcc_mitm_bwd(int T, int N, const float* __restrict__ trans, const float* __restrict__ grad,
                                                    float* __restrict__ inputGrad, FccWs ws, int dir0 = 0) {
  __shared__ float sQ[2][kDppChunk][64];   // helper -> chain: the frame's scale
  __shared__ float sB[2][kDppChunk][64];   // chain -> helper: b_t before the frame's step (block 0) / E u_{t-1} of the frame (block 1)
  __shared__ float sRm[32];
  const int b = blockIdx.x, dir = blockIdx.y + dir0, tid = threadIdx.x, lane = tid & 63;
  if (ws.redo[b]) return;   // this utterance ran (and will be differentiated) on the log-domain kernels
  const bool chain = __builtin_amdgcn_readfirstlane(tid >> 6) == 0;
  const DppGeom g = dpp_geom(lane);
  const float NEG = -INFINITY;
  const bool actG = g.sG < N, actH = g.sH < N;
  const int m = fcc_mitm_mid(T);
  const float* __restrict__ qb = ws.logs + (size_t)b * T * N;
  float* __restrict__ dxb = inputGrad + (size_t)b * T * N;
  float* __restrict__ rb = ws.r + (size_t)b * T * N;
  const float ginv = ws.ginv[b];
  const float gsc = ws.scale[b] * grad[b] * ginv;
  if (tid < 32) {
    float rm = NEG;
#pragma unroll
    for (int j = 0; j < 32; ++j) rm = fmaxf(rm, (tid < N && j < N) ? trans[(size_t)tid * N + j] : NEG);
    sRm[tid] = rm;
  }
  __syncthreads();

  if (dir == 0) {
    // ================================================================ beta continued: frames m .. 0 on the alpha half's scales
    const float* __restrict__ ub = ws.ahat + (size_t)b * T * N;
    auto downRole = [&](auto parTag) {
      constexpr bool PAR = decltype(parTag)::value;   // frame t = thi - s is held in arrangement G when t is odd: odd(s) = (s & 1) != PAR (PAR = m & 1)
    if (!chain) {
      float uc[kDppChunk], qc[kDppChunk], un[kDppChunk], qn[kDppChunk];
      auto fetch = [&](float (&uv)[kDppChunk], float (&qv)[kDppChunk], int thi) {
#pragma unroll
        for (int s = 0; s < kDppChunk; ++s) {
          const int t = thi - s;
          const bool odd = ((s & 1) != 0) != PAR;
          const bool act = odd ? actG : actH;
          const int st = odd ? g.sG : g.sH;
          uv[s] = (act && t >= 0) ? ub[(size_t)t * N + st] : 0.f;
          qv[s] = (act && t >= 1) ? qb[(size_t)t * N + st] : 0.f;
        }
      };
      fetch(uc, qc, m);
#pragma unroll
      for (int s = 0; s < kDppChunk; ++s) sQ[0][s][lane] = qc[s];
      W2L_LDS_BARRIER();
      int c = 0;
      float up[kDppChunk], qp[kDppChunk];
      auto emit = [&](int th, int buf) {
#pragma unroll
        for (int s = 0; s < kDppChunk; ++s) {
          const int t = th - s;
          const bool odd = ((s & 1) != 0) != PAR;
          const bool st = odd ? (g.primG && actG) : (g.primH && actH);
          const int sx = odd ? g.sG : g.sH;
          const float bv = sB[buf][s][lane];
          if (st && t >= 0) {
            dxb[(size_t)t * N + sx] = g.ok ? gsc * (up[s] * bv) : __builtin_nanf("");
            if (t >= 1) rb[(size_t)t * N + sx] = ginv * (bv * qp[s]);
          }
        }
      };
      for (int thi = m; thi >= 0; thi -= kDppChunk, ++c) {
        fetch(un, qn, thi - kDppChunk);
#pragma unroll
        for (int s = 0; s < kDppChunk; ++s) sQ[(c + 1) & 1][s][lane] = qn[s];
        if (c >= 1) emit(thi + kDppChunk, (c - 1) & 1);
#pragma unroll
        for (int s = 0; s < kDppChunk; ++s) { up[s] = uc[s]; qp[s] = qc[s]; uc[s] = un[s]; qc[s] = qn[s]; }
        W2L_LDS_BARRIER();
      }
      emit(m - (c - 1) * kDppChunk, (c - 1) & 1);
      return;
    }
    float EA[16], EB[16];
    dpp_tables(lane, g, [&](int j, int i) -> float {
      if (i >= N || j >= N) return 0.f;
      return __expf(trans[(size_t)i * N + j] - sRm[i]);
    }, EA, EB);
    float bv;
    {
      const bool odd = PAR;
      const bool act = odd ? actG : actH;
      bv = act ? ws.bm[(size_t)b * 32 + (odd ? g.sG : g.sH)] : 0.f;
    }
    W2L_LDS_BARRIER();
    int c = 0;
    for (int thi = m; thi >= 0; thi -= kDppChunk, ++c) {
      const int buf = c & 1;
      float qv[kDppChunk];
#pragma unroll
      for (int s = 0; s < kDppChunk; ++s) qv[s] = sQ[buf][s][lane];
      auto frames = [&](auto full) {
#pragma unroll
        for (int s = 0; s < kDppChunk; ++s) {
          const int t = thi - s;
          if (decltype(full)::value || t >= 0) {
            sB[buf][s][lane] = bv;
            if (decltype(full)::value || t >= 1) {
              const float r = bv * qv[s];
              if (((s & 1) != 0) != PAR) bv = comb_add16(dpp_dot16(r, EB));
              else bv = comb_add32(dpp_dot16(r, EA));
            }
          }
        }
      };
      if (thi - kDppChunk >= 0) frames(std::true_type{});
      else frames(std::false_type{});
      W2L_LDS_BARRIER();
    }
    };
    if (m & 1) downRole(std::true_type{});
    else downRole(std::false_type{});
    return;
  }

  // ================================================================== alpha continued: frames m+1 .. T-1 on the beta half's scales
  float* __restrict__ uw = ws.ahat + (size_t)b * T * N;
  const float* __restrict__ r2 = ws.r2 + (size_t)b * T * N;
  const int t1 = m + 1, nF = T - 1 - m;   // frame t = t1 + k, k = 0 .. nF - 1
  auto upRole = [&](auto parTag) {
    constexpr bool PAR = decltype(parTag)::value;   // frame parity = (s + t1) & 1: odd(s) = (s & 1) != PAR (PAR = t1 & 1)
  if (!chain) {
    float qc[kDppChunk], rc[kDppChunk], qn[kDppChunk], rn[kDppChunk], qp[kDppChunk], rp[kDppChunk];
    auto fetch = [&](float (&qv)[kDppChunk], float (&rv)[kDppChunk], int k0) {
#pragma unroll
      for (int s = 0; s < kDppChunk; ++s) {
        const int t = t1 + k0 + s;
        const bool odd = ((s & 1) != 0) != PAR;
        const bool act = odd ? actG : actH;
        const int st = odd ? g.sG : g.sH;
        const bool in = act && k0 + s < nF;
        qv[s] = in ? qb[(size_t)t * N + st] : 0.f;
        rv[s] = in ? r2[(size_t)t * N + st] : 0.f;
      }
    };
    fetch(qc, rc, 0);
#pragma unroll
    for (int s = 0; s < kDppChunk; ++s) sQ[0][s][lane] = qc[s];
    W2L_LDS_BARRIER();
    auto emit = [&](int k0, int buf) {
#pragma unroll
      for (int s = 0; s < kDppChunk; ++s) {
        const int t = t1 + k0 + s;
        const bool odd = ((s & 1) != 0) != PAR;
        const bool st = odd ? (g.primG && actG) : (g.primH && actH);
        const int sx = odd ? g.sG : g.sH;
        const float sv = sB[buf][s][lane];
        if (st && k0 + s < nF) {
          uw[(size_t)t * N + sx] = sv * qp[s];                                        // u_t (the chain's own product: same rounding)
          dxb[(size_t)t * N + sx] = g.ok ? gsc * (sv * rp[s]) : __builtin_nanf("");   // u_t b_t = (E u_{t-1}) (q_t b_t)
          rb[(size_t)t * N + sx] = ginv * rp[s];
        }
      }
    };
    int c = 0;
    for (int k0 = 0; k0 < nF; k0 += kDppChunk, ++c) {
      fetch(qn, rn, k0 + kDppChunk);
#pragma unroll
      for (int s = 0; s < kDppChunk; ++s) sQ[(c + 1) & 1][s][lane] = qn[s];
      if (c >= 1) emit(k0 - kDppChunk, (c - 1) & 1);
#pragma unroll
      for (int s = 0; s < kDppChunk; ++s) { qp[s] = qc[s]; rp[s] = rc[s]; qc[s] = qn[s]; rc[s] = rn[s]; }
      W2L_LDS_BARRIER();
    }
    if (c >= 1) emit((c - 1) * kDppChunk, (c - 1) & 1);
    return;
  }
  float EA[16], EB[16];
  dpp_tables(lane, g, [&](int i, int j) -> float {
    if (i >= N || j >= N) return 0.f;
    return __expf(trans[(size_t)i * N + j] - sRm[i]);
  }, EA, EB);
  float u;
  {
    const bool odd = m & 1;
    const bool act = odd ? actG : actH;
    u = act ? uw[(size_t)m * N + (odd ? g.sG : g.sH)] : 0.f;
  }
  W2L_LDS_BARRIER();
  int c = 0;
  for (int k0 = 0; k0 < nF; k0 += kDppChunk, ++c) {
    const int buf = c & 1;
    float qv[kDppChunk];
#pragma unroll
    for (int s = 0; s < kDppChunk; ++s) qv[s] = sQ[buf][s][lane];
    auto frames = [&](auto full) {
#pragma unroll
      for (int s = 0; s < kDppChunk; ++s) {
        if (decltype(full)::value || k0 + s < nF) {
          float sv;
          // frame t odd: H -> G through step A (as the alpha half)
          if (((s & 1) != 0) != PAR) sv = comb_add32(dpp_dot16(u, EA));
          else sv = comb_add16(dpp_dot16(u, EB));
          sB[buf][s][lane] = sv;
          u = sv * qv[s];
        }
      }
    };
    if (k0 + kDppChunk <= nF) frames(std::true_type{});
    else frames(std::false_type{});
    W2L_LDS_BARRIER();
  }
  };
  if (t1 & 1) upRole(std::true_type{});
  else upRole(std::false_type{});
}

}  // namespace w2l
