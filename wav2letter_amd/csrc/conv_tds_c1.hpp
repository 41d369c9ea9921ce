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
  const long long total = (long long)p.B * p.Tout * p.H;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int h = (int)(idx % p.H);
  const long long bt = idx / p.H;
  const int t = (int)(bt % p.Tout), b = (int)(bt / p.Tout);
  const float* xb = p.x + (size_t)b * p.Tin * p.H + h;
  const int ti0 = t * p.stride - p.padl;
  float acc[CO];
#pragma unroll
  for (int c = 0; c < CO; ++c) acc[c] = p.bias ? p.bias[c] : 0.f;
#pragma unroll
  for (int j = 0; j < KWM; ++j) {
    const int ti = ti0 + j;
    float xv = 0.f;
    if (j < p.kw && ti >= 0 && ti < p.Tin) xv = xb[(size_t)ti * p.H];
    const float* wj = p.w + (j < p.kw ? j : 0) * CO;      // uniform: scalar loads
#pragma unroll
    for (int c = 0; c < CO; ++c) acc[c] = fmaf(xv, wj[c], acc[c]);
  }
  float2* dst = (float2*)(p.y + (size_t)idx * CO);
#pragma unroll
  for (int c = 0; c < CO; c += 2) {
    float a0 = acc[c], a1 = acc[c + 1];
    if (p.relu) { a0 = fmaxf(a0, 0.f); a1 = fmaxf(a1, 0.f); }
    dst[c / 2] = make_float2(a0, a1);
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
  const long long total = (long long)p.B * p.Tout * p.H;
  float acc[KWM][CH], accb[CH];
#pragma unroll
  for (int j = 0; j < KWM; ++j)
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[j][c] = 0.f;
#pragma unroll
  for (int c = 0; c < CH; ++c) accb[c] = 0.f;
  for (long long pos = (long long)blockIdx.x * 128 + slot; pos < total; pos += (long long)gridDim.x * 128) {
    const int h = (int)(pos % p.H);
    const long long bt = pos / p.H;
    const int t = (int)(bt % p.Tout), b = (int)(bt / p.Tout);
    const float* xb = p.x + (size_t)b * p.Tin * p.H + h;
    const int ti0 = t * p.stride - p.padl;
    float dv[CH];
    const float* dp = p.dy + (size_t)pos * CO + half * CH;
#pragma unroll
    for (int c = 0; c < CH; ++c) dv[c] = dp[c];
#pragma unroll
    for (int c = 0; c < CH; ++c) accb[c] += dv[c];
#pragma unroll
    for (int j = 0; j < KWM; ++j) {
      const int ti = ti0 + j;
      float xv = 0.f;
      if (j < p.kw && ti >= 0 && ti < p.Tin) xv = xb[(size_t)ti * p.H];
#pragma unroll
      for (int c = 0; c < CH; ++c) acc[j][c] = fmaf(xv, dv[c], acc[j][c]);
    }
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
