// conv_tds_bf16.hip -- the TDS time convolution in the mixed-precision mode (BASELINE config 3: streaming_convnets
// am_500ms_future_context.arch, TDS blocks with c = 15 / 19 / 23 / 27 channels, kw = 9 / 11, 80 mel rows): forward and
// backward-data on v_mfma_f32_32x32x16_bf16.
//
// Reference: fl::TDSBlock's Conv2D (recipes/joint_training_vox_populi/cpc/SequentialBuilder.cpp:254-268; data flow
// recipes/streaming_convnets/inference/inference/module/nn/TDSBlock.cpp:58-70) under fl's AMP, which casts the operands of
// conv2d to half precision (recipes/slimIPL/src/Train.cpp:211, :1681-1760) -- here bf16 operands (x / dy and the weights
// rounded to nearest even), fp32 accumulation, fp32 bias / ReLU / residual addend and fp32 results.
//
// Why a second kernel family: the fp32 kernels of conv_tds.hip run these odd channel counts at 57 TF/s (forward /
// backward-data, one ds_read_b32 per MFMA operand element): 7.7 ms of the 50 ms config-3 step (run j2).  With bf16
// operands the arithmetic is 16x cheaper and the convolution becomes what its byte count says it is -- an HBM pass
// (4 C bytes read + 4 C written per (frame, mel row)): everything below is about staging once and reading wide.
//
//   * implicit GEMM, im2col-free: rows = 32 CONSECUTIVE FRAMES of one (utterance, mel row), columns = the C <= 32 output
//     channels, K = (tap, c_in) with c_in padded to CP = 16 / 24 / 32: k = tap * CP + c_in.  The input slab of a workgroup
//     (64 + kw - 1 frames x 8 mel rows) is staged ONCE into LDS as bf16, frame-major: slab[f][h][CP].  The 8 consecutive k a
//     lane needs for one MFMA (8 channels of frame t + tap) are 16 contiguous bytes of that slab: ONE ds_read_b128 per
//     MFMA operand, no per-element gathers, no division in the loop (every (tap, c_in) offset is a compile-time constant).
//     The frame pitch is 16 bytes x an odd number, so the 32 frames of a fragment read hit distinct bank groups;
//   * the weights are a bf16 image [32 columns][K] written once per step by tds_bf_wprep_k (forward orientation, or
//     tap-flipped and transposed for backward-data) and live in REGISTERS for the whole workgroup (4 VGPRs per k-step);
//   * epilogue from the accumulators: bias, ReLU, the residual / upstream addend of backward-data, fp32 stores of C
//     consecutive floats per (frame, mel row).
// Padded channels and taps multiply zeros of the weight image; the slab is zero-filled first so that they read finite
// values.  Geometry outside (stride 1, C <= 32, H * C % 4 == 0, the instantiated (CP, k-steps) pairs) returns
// W2L_EUNSUPPORTED and the caller stays on the fp32 kernels.
#include "gemm.hpp"

namespace w2l {

typedef __bf16 tb_bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 tb_bf16x2_t __attribute__((ext_vector_type(2)));
typedef float tb_f32x2_t __attribute__((ext_vector_type(2)));

constexpr int kTbTT = 64;   // output frames per workgroup
constexpr int kTbHB = 8;    // mel rows per workgroup (two per wave)

struct TdsBfP {
  const float* x;        // [B][T][H][C] fp32: the activations (forward) or the output gradient (backward-data)
  const uint16_t* wimg;  // [32][Kp] bf16, k = tap * CP + c
  const float* bias;     // [C] or null
  const float* add;      // layout of y, or null
  float* y;              // [B][T][H][C]
  int B, T, H, C, kw, padl, relu;
  uint32_t cMagic;       // ceil(2^32 / C): e / C for e < 2^16
};

__device__ __forceinline__ uint16_t tb_bf16(float v) {
  const tb_f32x2_t p = {v, 0.f};
  return (uint16_t)(__builtin_bit_cast(uint32_t, __builtin_convertvector(p, tb_bf16x2_t)) & 0xffffu);   // round to nearest even
}

// frame pitch in bytes: HB rows of CP bf16, rounded up to 16 bytes x an odd number
__host__ __device__ constexpr int tb_frame_pitch(int CP) { return ((kTbHB * CP * 2 / 16) | 1) * 16; }

template <int CP, int NSTEP>
__global__ __launch_bounds__(256) void tds_conv_bf_k(TdsBfP p) {
  static_assert((NSTEP * 16) % CP == 0, "whole taps");
  constexpr int KWP = NSTEP * 16 / CP;          // taps the K loop walks (>= kw; the weight image is zero beyond kw)
  constexpr int NF = kTbTT + KWP - 1;           // slab frames
  constexpr int FS = tb_frame_pitch(CP);        // bytes
  extern __shared__ __attribute__((aligned(16))) unsigned char slab[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int h0 = blockIdx.x * kTbHB, t0 = blockIdx.y * kTbTT, b = blockIdx.z;
  const int C = p.C;

  // ---- weight fragments: column li, k = 16 s + 8 lh .. + 8 of every k-step (registers for the whole workgroup)
  tb_bf16x8_t wf[NSTEP];
  {
    const uint16_t* wr = p.wimg + (size_t)li * (NSTEP * 16) + 8 * lh;
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) wf[s] = *(const tb_bf16x8_t*)(wr + 16 * s);
  }

  // ---- slab: zero fill, then the valid (frame, mel row, channel) elements as bf16
  {
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    for (int o = tid * 16; o < NF * FS; o += 256 * 16) *(uint4*)(slab + o) = z;
  }
  __syncthreads();
  {
    // (the loads of a batch are issued back to back, THEN converted and stored: one exposed memory latency per batch of 8
    // float4 instead of one per float4 -- the first build waited for every load before issuing the next: 150 - 350 us per call)
    const int runF4 = 2 * C;                     // float4 per frame: 8 mel rows x C floats (H * C % 4 == 0: host-checked)
    const float* xb = p.x + ((size_t)b * p.T * p.H + h0) * C;
    const size_t frameStride = (size_t)p.H * C;
    const int hValid = kTbHB;                    // H % 8 == 0 (host-checked)
    const int total = NF * runF4;
    constexpr int NV = 8;
    for (int base = 0; base < total; base += NV * 256) {
      float4 v4[NV];
      int fo[NV], eo[NV];
#pragma unroll
      for (int u = 0; u < NV; ++u) {
        // UNCONDITIONAL loads from clamped (always valid) addresses, zero selected afterwards: a load inside a branch makes
        // hipcc wait vmcnt(0) right behind it (seen in the second build's ISA: eight serialized round trips per batch)
        const int q = base + tid + 256 * u;
        const int qq = q < total ? q : total - 1;
        const int f = qq / runF4, j = qq - f * runF4;
        const int tin = t0 - p.padl + f;
        const int tc = tin < 0 ? 0 : (tin >= p.T ? p.T - 1 : tin);
        const bool ok = q < total && tin == tc && 4 * j < hValid * C;
        fo[u] = ok ? f : -1;
        eo[u] = 4 * j;
        v4[u] = *(const float4*)(xb + (size_t)tc * frameStride + 4 * j);
      }
#pragma unroll
      for (int u = 0; u < NV; ++u) {
        if (fo[u] < 0) continue;
        const float v[4] = {v4[u].x, v4[u].y, v4[u].z, v4[u].w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int e = eo[u] + k;
          const int h = (int)(((uint64_t)e * p.cMagic) >> 32);
          const int c = e - h * C;
          if (h < hValid) *(uint16_t*)(slab + fo[u] * FS + (h * CP + c) * 2) = tb_bf16(v[k]);
        }
      }
    }
  }
  __syncthreads();

  // ---- two mel rows per wave, two 32-frame halves each: rows = frames, K = (tap, c_in)
#pragma unroll 1
  for (int th = 0; th < 2; ++th) {
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    const unsigned char* a0 = slab + (32 * th + li) * FS + (2 * wave) * CP * 2;
    const unsigned char* a1 = a0 + CP * 2;
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
      // k = 16 s + 8 lh: tap = k / CP, c0 = k % CP -- constants of the unrolled step, selected by the lane half
      const int kLo = 16 * s, kHi = 16 * s + 8;
      const int offLo = (kLo / CP) * FS + (kLo % CP) * 2, offHi = (kHi / CP) * FS + (kHi % CP) * 2;
      const int off = lh ? offHi : offLo;
      const tb_bf16x8_t f0 = *(const tb_bf16x8_t*)(a0 + off);
      const tb_bf16x8_t f1 = *(const tb_bf16x8_t*)(a1 + off);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f0, wf[s], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f1, wf[s], acc1, 0, 0, 0);
    }
    // epilogue: C layout col = li (output channel), row = (r & 3) + 8 (r >> 2) + 4 lh (frame).  The addend of backward-data
    // is fetched for all 32 outputs of the lane FIRST, unconditionally from clamped addresses (a load inside the bounds
    // branch serialises: hipcc waits vmcnt(0) behind each, and on gfx9 that counter also holds the stores in between)
    const int lc = li < C ? li : C - 1;
    const float bv = p.bias ? p.bias[lc] : 0.f;
    const size_t tile = (((size_t)b * p.T + t0) * p.H + h0) * C;   // wave-uniform; per-output offsets stay 32-bit
    const float* addb = p.add ? p.add + tile : nullptr;
    float* yb = p.y + tile;
    const int tl0 = 32 * th + 4 * lh, tlMax = p.T - 1 - t0, hC = p.H * C;
    float av[2][16];
    auto off = [&](int hh, int r) {
      const int tl = tl0 + (r & 3) + 8 * (r >> 2);
      return (tl < tlMax ? tl : tlMax) * hC + (2 * wave + hh) * C + lc;
    };
    if (addb) {
#pragma unroll
      for (int hh = 0; hh < 2; ++hh)
#pragma unroll
        for (int r = 0; r < 16; ++r) av[hh][r] = addb[off(hh, r)];
    } else {
#pragma unroll
      for (int hh = 0; hh < 2; ++hh)
#pragma unroll
        for (int r = 0; r < 16; ++r) av[hh][r] = 0.f;
    }
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = (hh ? acc1[r] : acc0[r]) + bv;
        if (p.relu) v = fmaxf(v, 0.f);
        v += av[hh][r];
        asm volatile("" : "+v"(v));   // the sum (and with it the wait for the addend) stays outside the bounds branch
        av[hh][r] = v;
      }
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int tl = tl0 + (r & 3) + 8 * (r >> 2);
        if (li < C && tl <= tlMax) yb[off(hh, r)] = av[hh][r];
      }
  }
}

// weight image [32][Kp] bf16 of w [kw][C][C] (forward: w[tap][ci][co]):
//   flip == 0: img[co][tap * CP + ci] = w[tap][ci][co]
//   flip == 1: img[ci][j * CP + co]   = w[kw - 1 - j][ci][co]      (backward-data: a forward pass over dy)
__global__ __launch_bounds__(256) void tds_bf_wprep_k(const float* __restrict__ w, int kw, int C, int CP, int Kp, int flip,
                                                      uint16_t* __restrict__ img) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= 32 * Kp) return;
  const int n = e / Kp, k = e - n * Kp;
  const int tap = k / CP, c = k - tap * CP;
  float v = 0.f;
  if (n < C && c < C && tap < kw) v = flip ? w[((size_t)(kw - 1 - tap) * C + n) * C + c] : w[((size_t)tap * C + c) * C + n];
  img[e] = tb_bf16(v);
}


// ---------------------------------------------------------------------------------------------------------------------
// backward-filter:  dW[tap][ci][co] = sum over (utterance, frame t, mel row h) of x[t + tap - padl][h][ci] * dy[t][h][co]
// The contraction runs over POSITIONS, so both MFMA operands need 8 consecutive positions per lane: the slabs are staged
// MEL-FASTEST, xs[frame][ci][h] and ys[frame][co][h] (bf16), and the 16 mel rows of a workgroup's block are the k dimension
// of one v_mfma_f32_32x32x16_bf16 (a tap shift moves whole frames, never the 16-byte alignment of a fragment; a
// time-fastest slab would misalign every odd tap).  Rows of the product = (tap, ci) in the forward K order k = tap CP + ci,
// 32 per tile; columns = co.  The transposition happens in the STAGING: lane = (channel c = lane >> 3, mel pair = lane & 7)
// loads x[..][2 hp][c] and x[..][2 hp + 1][c] and stores ONE packed dword -- 8 channels x 8 pairs per instruction land in
// (nearly) distinct banks at a 48-byte row pitch.  512 persistent workgroups walk the (utterance, 16-frame tile, 16-mel
// block) items; a wave owns a quarter of the row tiles and keeps their accumulators in registers over every item (no
// cross-wave reduction), the per-workgroup partials are added in workgroup order by tds_bf_filter_reduce_k (deterministic).
constexpr int kTfTT = 16;      // frames per item
constexpr int kTfHB = 16;      // mel rows per item = k of one MFMA
constexpr int kTfPitch = 48;   // bytes per (frame, channel) row: 16 mel rows of bf16 + 16 bytes (16 x an odd number)
constexpr int kTfWorkers = 512;

struct TdsBfFilterP {
  const float* x;    // [B][T][H][C]
  const float* dy;   // [B][T][H][C]
  float* partial;    // [workers][NRT * 32][32]
  int B, T, H, C, kw, padl;
};

template <int CP, int NSTEP>
__global__ __launch_bounds__(256) void tds_conv_bf_filter_k(TdsBfFilterP p) {
  constexpr int KWP = NSTEP * 16 / CP, Kp = NSTEP * 16;
  constexpr int NRT = (Kp + 31) / 32;            // row tiles of 32 (tap, ci) rows
  constexpr int NRTW = (NRT + 3) / 4;            // row tiles per wave
  constexpr int NFX = kTfTT + KWP - 1;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int C = p.C;
  const int FSX = C * kTfPitch;                  // bytes per slab frame (C rows)
  unsigned char* xs = lds;
  unsigned char* ys = lds + NFX * FSX;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int tTiles = (p.T + kTfTT - 1) / kTfTT, hBlocks = p.H / kTfHB;
  const int nItems = p.B * tTiles * hBlocks;

  f32x16 acc[NRTW];
  int rowOff[NRTW];   // byte offset of this lane's (tap, ci) row inside the x slab, + the lane half's 8 mel rows
#pragma unroll
  for (int j = 0; j < NRTW; ++j) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const int rt = wave + 4 * j;
    int k = 32 * rt + li;
    if (k > Kp - 1) k = Kp - 1;                  // rows past K: any valid address (never stored)
    const int tap = k / CP;
    int ci = k - tap * CP;
    if (ci > C - 1) ci = C - 1;                  // padded channels: any valid row (never stored)
    rowOff[j] = tap * FSX + ci * kTfPitch + 16 * lh;
  }
  const int bOff = (li < C ? li : C - 1) * kTfPitch + 16 * lh;
  const int cl = lane >> 3, hp = lane & 7;       // staging role: channel cl (+ 8 per pass), mel rows 2 hp, 2 hp + 1

  for (int item = blockIdx.x; item < nItems; item += gridDim.x) {
    const int hb = item % hBlocks, tt = (item / hBlocks) % tTiles, b = item / (hBlocks * tTiles);
    const int t0 = tt * kTfTT, h0 = hb * kTfHB;
    __syncthreads();                             // the previous item's fragments have been read
    // ---- staging: frames of x (with the tap halo) and of dy, transposed to mel-fastest bf16; zeros outside the utterance
    // (two frames x up to four channel passes = 16 scalar loads in flight per lane before the first conversion)
    for (int f0 = wave; f0 < NFX + kTfTT; f0 += 8) {
      float v0[2][4], v1[2][4];
#pragma unroll
      for (int ff = 0; ff < 2; ++ff) {
        const int f = f0 + 4 * ff;
        const bool isX = f < NFX;
        const int fl = isX ? f : f - NFX;
        const int tin = isX ? t0 - p.padl + fl : t0 + fl;
        const bool in = f < NFX + kTfTT && tin >= 0 && tin < p.T;
        const float* src = (isX ? p.x : p.dy) + ((((size_t)b * p.T + (in ? tin : 0)) * p.H + h0 + 2 * hp) * C);
#pragma unroll
        for (int cp = 0; cp < 4; ++cp) {   // unconditional loads from clamped addresses, zero selected afterwards
          const int c = cl + 8 * cp;
          const int cc = c < C ? c : C - 1;
          const float a0 = src[cc], a1 = src[C + cc];
          const bool ok = in && c < C;
          v0[ff][cp] = ok ? a0 : 0.f;
          v1[ff][cp] = ok ? a1 : 0.f;
        }
      }
#pragma unroll
      for (int ff = 0; ff < 2; ++ff) {
        const int f = f0 + 4 * ff;
        if (f >= NFX + kTfTT) continue;
        const bool isX = f < NFX;
        const int fl = isX ? f : f - NFX;
        unsigned char* dst = (isX ? xs : ys) + fl * FSX + hp * 4;
#pragma unroll
        for (int cp = 0; cp < 4; ++cp) {
          const int c = cl + 8 * cp;
          if (c >= C) continue;
          const tb_f32x2_t pr = {v0[ff][cp], v1[ff][cp]};
          *(uint32_t*)(dst + c * kTfPitch) = __builtin_bit_cast(uint32_t, __builtin_convertvector(pr, tb_bf16x2_t));
        }
      }
    }
    __syncthreads();
    // ---- one MFMA per (frame, row tile): k = the 16 mel rows of the block
#pragma unroll 4
    for (int t = 0; t < kTfTT; ++t) {
      const tb_bf16x8_t bf = *(const tb_bf16x8_t*)(ys + t * FSX + bOff);
#pragma unroll
      for (int j = 0; j < NRTW; ++j) {
        if (wave + 4 * j < NRT) {
          const tb_bf16x8_t af = *(const tb_bf16x8_t*)(xs + t * FSX + rowOff[j]);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[j], 0, 0, 0);
        }
      }
    }
  }
  // ---- this workgroup's partial: [NRT * 32 rows][32 columns] fp32
  float* out = p.partial + (size_t)blockIdx.x * (NRT * 32) * 32;
#pragma unroll
  for (int j = 0; j < NRTW; ++j) {
    const int rt = wave + 4 * j;
    if (rt >= NRT) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = 32 * rt + (r & 3) + 8 * (r >> 2) + 4 * lh;
      out[(size_t)row * 32 + li] = acc[j][r];
    }
  }
}

// dw[tap][ci][co] = sum over the workgroups' partials, always in the same order: 16 outputs x 16 worker lanes per
// workgroup, a lane adds every 16th partial, the 16 lane sums are added in lane order (the first build walked all 512
// partials in ONE thread per output: 135 us per call)
__global__ __launch_bounds__(256) void tds_bf_filter_reduce_k(const float* __restrict__ partial, int workers, int rows32, int kw, int C, int CP,
                                                              float* __restrict__ dw) {
  __shared__ float sm[16][17];
  const int o = threadIdx.x & 15, wl = threadIdx.x >> 4;
  const int e = blockIdx.x * 16 + o, n = kw * C * C;
  float s = 0.f;
  if (e < n) {
    const int co = e % C, ci = (e / C) % C, tap = e / (C * C);
    const size_t at = (size_t)(tap * CP + ci) * 32 + co;
    for (int w = wl; w < workers; w += 16) s += partial[(size_t)w * rows32 * 32 + at];
  }
  sm[wl][o] = s;
  __syncthreads();
  if (wl == 0 && e < n) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += sm[k][o];
    dw[e] = t;
  }
}

template <int CP, int NSTEP>
static int tb_launch_filter(const TdsBfFilterP& p0, float* dw, hipStream_t s) {
  constexpr int KWP = NSTEP * 16 / CP, Kp = NSTEP * 16, NRT = (Kp + 31) / 32;
  TdsBfFilterP p = p0;
  const int tTiles = (p.T + kTfTT - 1) / kTfTT, nItems = p.B * tTiles * (p.H / kTfHB);
  const int workers = nItems < kTfWorkers ? nItems : kTfWorkers;
  const size_t need = (size_t)workers * NRT * 32 * 32 * sizeof(float);
  if (need > kSkScratchBytes) return W2L_EUNSUPPORTED;
  p.partial = sk_scratch(s, kSkScratchBytes);
  if (!p.partial) return W2L_EHIP;
  const size_t shmem = (size_t)(kTfTT + KWP - 1 + kTfTT) * p.C * kTfPitch;
  static bool attr = false;
  if (!attr) {
    W2L_HIP_CHECK(hipFuncSetAttribute((const void*)tds_conv_bf_filter_k<CP, NSTEP>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    attr = true;
  }
  if (shmem > 96 * 1024) return W2L_EUNSUPPORTED;
  hipLaunchKernelGGL((tds_conv_bf_filter_k<CP, NSTEP>), dim3((unsigned)workers), dim3(256), shmem, s, p);
  const int n = p.kw * p.C * p.C;
  hipLaunchKernelGGL(tds_bf_filter_reduce_k, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, s, p.partial, workers, NRT * 32, p.kw, p.C, CP, dw);
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}

struct TbGeom { int CP, NSTEP; };
static bool tb_geometry(const w2l_conv_desc* d, TbGeom& g) {
  if (!d || d->stride != 1 || d->Cin != d->Cout || d->Cin < 1 || d->Cin > 32 || d->kw < 1) return false;
  if (d->padl + d->padr != d->kw - 1 || d->H % kTbHB != 0) return false;   // whole 8-row mel blocks (16-byte aligned runs of 8 C floats)
  const int C = d->Cin;
  g.CP = C <= 16 ? 16 : C <= 24 ? 24 : 32;
  int n = (d->kw * g.CP + 15) / 16;
  while ((n * 16) % g.CP) ++n;
  g.NSTEP = n;
  // the instantiated pairs (streaming recipe: (16, 9) (24, 15) (24, 18) (32, 22); sota/2019 TDS-CTC channel counts at kw = 21: (16, 21) (24, 33))
  return (g.CP == 16 && (n == 9 || n == 21)) || (g.CP == 24 && (n == 15 || n == 18 || n == 33)) || (g.CP == 32 && n == 22);
}

template <int CP, int NSTEP>
static int tb_launch(const TdsBfP& p, hipStream_t s) {
  constexpr int KWP = NSTEP * 16 / CP;
  const size_t shmem = (size_t)(kTbTT + KWP - 1) * tb_frame_pitch(CP);
  static bool attr = false;
  if (!attr && shmem > 64 * 1024) {
    W2L_HIP_CHECK(hipFuncSetAttribute((const void*)tds_conv_bf_k<CP, NSTEP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    attr = true;
  }
  const dim3 grid((unsigned)((p.H + kTbHB - 1) / kTbHB), (unsigned)((p.T + kTbTT - 1) / kTbTT), (unsigned)p.B);
  hipLaunchKernelGGL((tds_conv_bf_k<CP, NSTEP>), grid, dim3(256), shmem, s, p);
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}

static int tb_dispatch(const TbGeom& g, const TdsBfP& p, hipStream_t s) {
  if (g.CP == 16 && g.NSTEP == 9) return tb_launch<16, 9>(p, s);
  if (g.CP == 16 && g.NSTEP == 21) return tb_launch<16, 21>(p, s);
  if (g.CP == 24 && g.NSTEP == 15) return tb_launch<24, 15>(p, s);
  if (g.CP == 24 && g.NSTEP == 18) return tb_launch<24, 18>(p, s);
  if (g.CP == 24 && g.NSTEP == 33) return tb_launch<24, 33>(p, s);
  if (g.CP == 32 && g.NSTEP == 22) return tb_launch<32, 22>(p, s);
  return W2L_EUNSUPPORTED;
}

static int tb_dispatch_filter(const TbGeom& g, const TdsBfFilterP& p, float* dw, hipStream_t s) {
  if (g.CP == 16 && g.NSTEP == 9) return tb_launch_filter<16, 9>(p, dw, s);
  if (g.CP == 16 && g.NSTEP == 21) return tb_launch_filter<16, 21>(p, dw, s);
  if (g.CP == 24 && g.NSTEP == 15) return tb_launch_filter<24, 15>(p, dw, s);
  if (g.CP == 24 && g.NSTEP == 18) return tb_launch_filter<24, 18>(p, dw, s);
  if (g.CP == 24 && g.NSTEP == 33) return tb_launch_filter<24, 33>(p, dw, s);
  if (g.CP == 32 && g.NSTEP == 22) return tb_launch_filter<32, 22>(p, dw, s);
  return W2L_EUNSUPPORTED;
}

}  // namespace w2l

using namespace w2l;

// bf16 elements of ONE weight image of this geometry (0: the geometry has no bf16 kernel -- stay on w2l_conv_*)
W2L_API size_t w2l_tds_conv_bf16_image_elems(const w2l_conv_desc* d) {
  TbGeom g;
  return tb_geometry(d, g) ? (size_t)32 * g.NSTEP * 16 : 0;
}

// once per step: the forward and the backward-data weight images of w [kw][C][C] (fp32 master weights)
W2L_API int w2l_tds_conv_bf16_prepare(const w2l_conv_desc* d, const float* w, uint16_t* imgForward, uint16_t* imgBackward,
                                      w2l_stream_t stream) {
  TbGeom g;
  if (!w || (!imgForward && !imgBackward)) return W2L_EINVAL;
  if (!tb_geometry(d, g)) return W2L_EUNSUPPORTED;
  const int Kp = g.NSTEP * 16;
  const unsigned blocks = (unsigned)((32 * Kp + 255) / 256);
  if (imgForward) hipLaunchKernelGGL(tds_bf_wprep_k, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, d->kw, d->Cin, g.CP, Kp, 0, imgForward);
  if (imgBackward) hipLaunchKernelGGL(tds_bf_wprep_k, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, d->kw, d->Cin, g.CP, Kp, 1, imgBackward);
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}

// y = (relu)(conv(bf16(x), imgForward) + bias)
W2L_API int w2l_tds_conv_bf16_forward(const w2l_conv_desc* d, const float* x, const uint16_t* imgForward, const float* bias, float* y,
                                      int relu, w2l_stream_t stream) {
  TbGeom g;
  if (!x || !imgForward || !y) return W2L_EINVAL;
  if (!tb_geometry(d, g)) return W2L_EUNSUPPORTED;
  TdsBfP p{x, imgForward, bias, nullptr, y, d->B, d->T, d->H, d->Cin, d->kw, d->padl, relu, (uint32_t)((0x100000000ull + d->Cin - 1) / d->Cin)};
  prof_begin((hipStream_t)stream, 2.0 * d->B * (double)d->T * d->H * d->Cin * (double)d->Cout * d->kw, PROF_TDSCONV);
  const int st = tb_dispatch(g, p, (hipStream_t)stream);
  prof_end((hipStream_t)stream);
  return st;
}

// dx = (add +) conv^T(bf16(dy), w): a forward pass over dy with the flipped image and the mirrored left padding
W2L_API int w2l_tds_conv_bf16_backward_data(const w2l_conv_desc* d, const float* dy, const uint16_t* imgBackward, const float* add,
                                            float* dx, w2l_stream_t stream) {
  TbGeom g;
  if (!dy || !imgBackward || !dx) return W2L_EINVAL;
  if (!tb_geometry(d, g)) return W2L_EUNSUPPORTED;
  TdsBfP p{dy, imgBackward, nullptr, add, dx, d->B, d->T, d->H, d->Cin, d->kw, d->kw - 1 - d->padl, 0, (uint32_t)((0x100000000ull + d->Cin - 1) / d->Cin)};
  prof_begin((hipStream_t)stream, 2.0 * d->B * (double)d->T * d->H * d->Cin * (double)d->Cout * d->kw, PROF_TDS_BWD_DATA);
  const int st = tb_dispatch(g, p, (hipStream_t)stream);
  prof_end((hipStream_t)stream);
  return st;
}

// dw [kw][C][C] = x (*) dy on bf16-rounded operands (fp32 accumulation); the bias gradient is an fp32 column sum of dy
// (w2l_colsum over [B T H][C]) and not part of this call.  H must be a multiple of 16.
W2L_API int w2l_tds_conv_bf16_backward_filter(const w2l_conv_desc* d, const float* x, const float* dy, float* dw, w2l_stream_t stream) {
  TbGeom g;
  if (!x || !dy || !dw) return W2L_EINVAL;
  if (!tb_geometry(d, g) || d->H % kTfHB != 0) return W2L_EUNSUPPORTED;
  TdsBfFilterP p{x, dy, nullptr, d->B, d->T, d->H, d->Cin, d->kw, d->padl};
  prof_begin((hipStream_t)stream, 2.0 * d->B * (double)d->T * d->H * d->Cin * (double)d->Cout * d->kw, PROF_TDS_BWD_FILTER);
  const int st = tb_dispatch_filter(g, p, dw, (hipStream_t)stream);
  prof_end((hipStream_t)stream);
  return st;
}
