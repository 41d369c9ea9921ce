// conv_tds_c1.hpp -- the FIRST convolution of the TDS recipes, `C2 1 10 21 1 2 1 -1 -1` (am_tds_ctc.arch:3): one input
// channel (the 80 mel rows are the H axis), 10 output channels, 21 taps, stride 2 over T = 1500 frames.
// 0.8 GFLOP against 92 MB of traffic (x 15 MB, y / dy 77 MB): HBM-bound by a wide margin, nothing for the matrix cores
// (K = 21).  The generic slab kernels of conv_tds.hip ran it at 75 us forward and 200 us for the filter gradient; here
// it is plain streaming VALU work:
//   forward   one thread per (utterance, output frame, mel row): 21 coalesced loads of x (lanes <-> mel rows), the 210
//             weights as scalar operands, 10 accumulators, five 8-byte stores (a thread's 10 outputs are contiguous);
//   filter    one thread per (position, half of the output channels): 21 x values x 5 dy values into 105 accumulators
//             (+ 5 for the bias gradient) over a strided walk of the positions; waves reduced by DPP, workgroups through
//             LDS, one partial row per workgroup, c1_filter_reduce_k adds the rows in order (deterministic).
#pragma once

namespace w2l {

struct TdsC1P {
  const float* x;     // [B][Tin][H] (one channel)
  const float* w;     // [kw][1][CO]
  const float* bias;  // [CO] or null
  const float* dy;    // filter: [B][Tout][H][CO]
  float* y;           // forward: [B][Tout][H][CO]
  int B, Tin, Tout, H, kw, stride, padl, relu;
};

template <int CO, int KWM>
__global__ __launch_bounds__(256) void tds_c1_fwd_k(TdsC1P p) {
  // a thread's CO outputs are contiguous, a workgroup's 256 x CO too: they leave through LDS as whole 16-byte vectors (five 8-byte
  // stores per thread, 40 bytes apart between lanes, touched every cache line of the span five times: 43 us for 77 MB)
  __shared__ float so[256 * CO];
  const long long total = (long long)p.B * p.Tout * p.H;
  const long long base = (long long)blockIdx.x * 256;
  const long long idx = base + threadIdx.x;
  if (idx < total) {
    const int h = (int)((unsigned)idx % (unsigned)p.H);
    const unsigned bt = (unsigned)idx / (unsigned)p.H;
    const int t = (int)(bt % (unsigned)p.Tout), b = (int)(bt / (unsigned)p.Tout);
    const float* xb = p.x + (size_t)b * p.Tin * p.H + h;
    const int ti0 = t * p.stride - p.padl;
    float acc[CO];
#pragma unroll
    for (int c = 0; c < CO; ++c) acc[c] = p.bias ? p.bias[c] : 0.f;
    // all 21 loads first, unconditional (a frame outside the utterance reads frame 0 and is zeroed by a select): behind a
    // predicate every load waits for the one before it
    float xv[KWM];
#pragma unroll
    for (int j = 0; j < KWM; ++j) {
      const int ti = ti0 + j;
      const bool ok = j < p.kw && ti >= 0 && ti < p.Tin;
      const float v = xb[(size_t)(ok ? ti : 0) * p.H];
      xv[j] = ok ? v : 0.f;
    }
#pragma unroll
    for (int j = 0; j < KWM; ++j) {
      const float* wj = p.w + (j < p.kw ? j : 0) * CO;      // uniform: scalar loads
#pragma unroll
      for (int c = 0; c < CO; ++c) acc[c] = fmaf(xv[j], wj[c], acc[c]);
    }
#pragma unroll
    for (int c = 0; c < CO; ++c) so[threadIdx.x * CO + c] = p.relu ? fmaxf(acc[c], 0.f) : acc[c];
  }
  __syncthreads();
  const long long left = total - base;
  const int nOut = (int)(left < 256 ? left : 256) * CO;           // floats of this workgroup (256 CO = a multiple of 4)
  float* dst = p.y + (size_t)base * CO;                            // 16-byte aligned: base CO floats = 256 CO blockIdx
  for (int e = 4 * threadIdx.x; e < nOut; e += 1024) {
    if (e + 4 <= nOut) *(float4*)(dst + e) = *(const float4*)(so + e);
    else for (int k = e; k < nOut; ++k) dst[k] = so[k];
  }
}

// partial[blockIdx][kw CO + CO]: this workgroup's share of dW[j][0][co] and dbias[co]
template <int CO, int KWM>
__global__ __launch_bounds__(256) void tds_c1_filter_k(TdsC1P p, float* __restrict__ partial) {
  constexpr int CH = CO / 2;                    // output channels per thread: the two halves of a workgroup split them
  __shared__ float red[4][2 * (KWM * CH + CH)];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = wave & 1;                    // waves 0, 2: channels [0, CH); waves 1, 3: [CH, CO)
  const int slot = (wave >> 1) * 64 + lane;     // 128 positions per workgroup and step
  float acc[KWM][CH], accb[CH];
#pragma unroll
  for (int j = 0; j < KWM; ++j)
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[j][c] = 0.f;
#pragma unroll
  for (int c = 0; c < CH; ++c) accb[c] = 0.f;
  // A thread walks RUNS of KT consecutive output frames of one mel row: the 2 KT + 19 input frames of a run are loaded once
  // and shared by its KT positions (a position alone loads 21 + 5 values for 110 FMAs -- one dependent global round trip
  // per 110 FMAs was all the first version did: 67 us for 92 MB), and the next run's loads are issued before this run's FMAs.
  constexpr int KT = 4;
  const unsigned H = (unsigned)p.H, nRunsT = ((unsigned)p.Tout + KT - 1) / KT;
  const unsigned totalRuns = (unsigned)p.B * nRunsT * H;          // run index = (b * nRunsT + r) * H + h
  const unsigned step = gridDim.x * 128u;
  constexpr int NX = 2 * (KT - 1) + KWM;                          // stride 2 only (host-checked)
  auto fetch = [&](unsigned run, float (&xv)[NX], float (&dv)[KT][CH]) {
    const bool live = run < totalRuns;
    const unsigned q = live ? run : 0u;
    const unsigned h = q % H, br = q / H;
    const unsigned r = br % nRunsT, b = br / nRunsT;
    const int t0 = (int)r * KT;
    const float* xb = p.x + (size_t)b * p.Tin * H + h;
    const int ti0 = t0 * 2 - p.padl;
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      const bool okt = live && t0 + k < p.Tout;
      const float* dp = p.dy + (((size_t)b * p.Tout + (okt ? t0 + k : 0)) * H + h) * CO + half * CH;
#pragma unroll
      for (int c = 0; c < CH; ++c) { const float v = dp[c]; dv[k][c] = okt ? v : 0.f; }
    }
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      const int ti = ti0 + j;
      const bool ok = ti >= 0 && ti < p.Tin;
      const float v = xb[(size_t)(ok ? ti : 0) * H];
      xv[j] = ok ? v : 0.f;
    }
  };
  float xc[NX], dc[KT][CH], xn[NX], dn[KT][CH];
  unsigned run = blockIdx.x * 128u + slot;
  fetch(run, xc, dc);
  for (; run < totalRuns; run += step) {
    fetch(run + step, xn, dn);
#pragma unroll
    for (int k = 0; k < KT; ++k) {
#pragma unroll
      for (int c = 0; c < CH; ++c) accb[c] += dc[k][c];
#pragma unroll
      for (int j = 0; j < KWM; ++j)
#pragma unroll
        for (int c = 0; c < CH; ++c) acc[j][c] = fmaf(xc[2 * k + j], dc[k][c], acc[j][c]);   // (taps >= kw: computed, never exported)
    }
#pragma unroll
    for (int j = 0; j < NX; ++j) xc[j] = xn[j];
#pragma unroll
    for (int k = 0; k < KT; ++k)
#pragma unroll
      for (int c = 0; c < CH; ++c) dc[k][c] = dn[k][c];
  }
  // waves by DPP, then the two waves of a half through LDS, in wave order
#pragma unroll
  for (int j = 0; j < KWM; ++j)
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const float s = wave_sum(acc[j][c]);
      if (lane == 0) red[wave][j * CH + c] = s;
    }
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const float s = wave_sum(accb[c]);
    if (lane == 0) red[wave][KWM * CH + c] = s;
  }
  __syncthreads();
  float* dst = partial + (size_t)blockIdx.x * (KWM * CO + CO);
  for (int e = tid; e < KWM * CO + CO; e += 256) {
    // element e = j CO + co (co < CO), then the bias gradient
    const int j = e < KWM * CO ? e / CO : KWM, co = e < KWM * CO ? e - j * CO : e - KWM * CO;
    const int hw = co / CH, c = co - hw * CH;
    const int k = (j < KWM ? j * CH : KWM * CH) + c;
    dst[e] = red[hw][k] + red[hw + 2][k];
  }
}

template <int CO, int KWM>
__global__ __launch_bounds__(1024) void tds_c1_filter_reduce_k(const float* __restrict__ partial, int nParts, int kw, float* __restrict__ dw,
                                                               float* __restrict__ dbias) {
  constexpr int ROW = KWM * CO + CO, NS = 32, PER = 8;
  __shared__ float red[NS][32];
  const int el = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int e = blockIdx.x * 32 + el;
  float s = 0.f;
  if (e < ROW)
    for (int g0 = sl * PER; g0 < nParts; g0 += NS * PER) {   // slice sl: rows [sl PER, sl PER + PER), then NS PER further on; all PER loads in flight
      float t[PER];
#pragma unroll
      for (int u = 0; u < PER; ++u) t[u] = g0 + u < nParts ? partial[(size_t)(g0 + u) * ROW + e] : 0.f;
#pragma unroll
      for (int u = 0; u < PER; ++u) s += t[u];
    }
  red[sl][el] = s;
  __syncthreads();
  if (sl == 0 && e < ROW) {
    float t = red[0][el];
#pragma unroll 8
    for (int k = 1; k < NS; ++k) t += red[k][el];
    if (e < KWM * CO) {
      if (e / CO < kw) dw[e] = t;
    } else if (dbias) {
      dbias[e - KWM * CO] = t;
    }
  }
}

}  // namespace w2l
