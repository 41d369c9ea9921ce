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
// values.
//
// The same kernels serve the recipe's SUB-SAMPLING convolutions (`C2 cin cout kw 1 s 1`: 1 -> 15 -> 19 -> 23 -> 27 channels,
// stride 2 / 2 / 2 / 1): C_in != C_out (CP pads the INPUT channels, the <= 32 output channels are the MFMA columns), the
// stride is a template parameter (row t of a fragment reads slab frame STRIDE * t + tap), and a strided backward-data is one
// stride-1 launch per PHASE f = (frame + padl) mod stride over dy with the taps f, f + s, ... (tap-flipped image per phase),
// written to every s-th frame of dx -- no zero-stuffed dy, no wasted multiplies (the decomposition of conv_tds.hip).
// Geometry outside (stride 1 / 2, channels <= 32, H % 8 == 0, the instantiated (CP, k-steps, stride) triples) returns
// W2L_EUNSUPPORTED and the caller stays on the fp32 kernels.
#include "gemm.hpp"

namespace w2l {

typedef __bf16 tb_bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 tb_bf16x2_t __attribute__((ext_vector_type(2)));
typedef float tb_f32x2_t __attribute__((ext_vector_type(2)));

constexpr int kTbTT = 64;   // output frames per workgroup
constexpr int kTbHB = 8;    // mel rows per workgroup (two per wave)

struct TdsBfP {
  const float* x;        // [B][Tin][H][Cin] fp32: the activations (forward) or the output gradient (backward-data)
  const uint16_t* wimg;  // [32][Kp] bf16, k = tap * CP + c_in
  const float* bias;     // [Cout] or null
  const float* add;      // layout of y, or null
  float* y;              // [B][ToutFull][H][Cout]; row t of this launch is frame oOff + oStep * t of it
  int B, Tin, Tout, H, Cin, Cout, padl, relu;
  int oOff, oStep, ToutFull;
  uint32_t cMagic;       // ceil(2^32 / Cin): e / Cin for e < 2^16 (unused at Cin == 1)
  int abl;               // timing ablations (probe library only; 0 in the product): 1 no staging, 2 no MFMAs, 4 no stores
};

__device__ __forceinline__ uint16_t tb_bf16(float v) {
  const tb_f32x2_t p = {v, 0.f};
  return (uint16_t)(__builtin_bit_cast(uint32_t, __builtin_convertvector(p, tb_bf16x2_t)) & 0xffffu);   // round to nearest even
}

// frame pitch in bytes: HB rows of CP bf16, rounded up to 16 bytes x an odd number
__host__ __device__ constexpr int tb_frame_pitch(int CP) { return ((kTbHB * CP * 2 / 16) | 1) * 16; }

// ---- staging: frames tFirst .. tFirst + nFrames - 1 of one utterance's HB mel rows (src = its [T][H][C] block + h0 * C floats)
// -> slab[f][h][CP] as bf16.  Frames outside [0, T) and the padded channels c >= C are written as zeros: tb_stage defines every
// byte of the frames it stages (tb_stage_generic leaves the padded channels alone: its callers zero-fill the slab first).
//
// The first generation walked the fp32 block as aligned float4 and scattered the four elements one by one -- with odd channel
// counts every element needs its own (mel row, channel) split and its own 2-byte LDS store: ~130 VALU instructions per 16 bytes
// (ISA count), and the convolutions ran at the rate the SIMDs could stage, 1.7 - 2.9 TB/s.  Here a lane owns one CHUNK of the
// slab -- 4 consecutive channels of one (frame, mel row): a dword-aligned 16-byte global load (channel counts are odd: the
// chunk starts wherever the row starts), two packed conversions, ONE 8-byte LDS store; all index arithmetic divides by
// compile-time constants.  The chunk that holds the row's last channels is loaded shifted back to END at the row's end (no
// read past the tensor) and shifted down in registers, zeros filling its padded slots.  C >= 4.
typedef float tb_f32x4u_t __attribute__((ext_vector_type(4), aligned(4)));

template <int CP, int HB>
__device__ __forceinline__ void tb_stage(unsigned char* slab, int FS, const float* __restrict__ src, int C, size_t frameStride, int tFirst,
                                         int nFrames, int T, int tid) {
  // A thread keeps ONE chunk role (mel row h, channels 4 cg .. 4 cg + 3) for the whole call and walks frames: FPT frames per
  // pass over the workgroup's FPT * PER role slots (threads past them idle: 16 of 256 at CP = 24, HB = 8).  Everything but
  // the frame index is then loop-invariant -- the address arithmetic per chunk shrinks to a clamp and one multiply-add
  // (SQ counters of the first chunk version, which re-derived (frame, h, cg) from a flat index per load: 35 VALU instructions
  // per MFMA, ~60 us of VALU issue in the 122 us forward pass at C = 15, profiles/r03_run24_conv_bf16_sq_lds_pmc.csv).
  constexpr int G = CP / 4, PER = HB * G, FPT = 256 / PER, NV = 8;
  static_assert(FPT >= 1, "a frame's chunks fit one pass");
  const int gLast = (C - 1) >> 2;         // the chunk with the row's last channels
  const int over = 4 * gLast + 4 - C;     // its 0 .. 3 slots past the row
  const int role = tid % PER, fl0 = tid / PER;
  const bool active = fl0 < FPT;
  const int h = role / G, cg = role - h * G;
  const int cgc = cg < gLast ? cg : gLast;
  const int srcOff = h * C + (cgc < gLast ? 4 * cgc : C - 4);   // floats inside the frame's block
  const int dstOff = (h * CP + 4 * cg) * 2;                     // bytes inside the slab frame
  const bool last = cg == gLast, pad = cg > gLast;
  for (int k0 = 0; k0 < nFrames; k0 += FPT * NV) {
    tb_f32x4u_t w[NV];
    int fo[NV];                           // slab frame | 0x40000000: zeros (frame outside the utterance); -1: nothing to write
#pragma unroll
    for (int u = 0; u < NV; ++u) {        // unconditional loads from clamped addresses (a load inside a branch serialises)
      const int f = k0 + fl0 + FPT * u;
      const int fc = f < nFrames ? f : nFrames - 1;
      const int tin = tFirst + fc;
      const int tc = tin < 0 ? 0 : (tin >= T ? T - 1 : tin);
      w[u] = *(const tb_f32x4u_t*)(src + (size_t)tc * frameStride + srcOff);
      fo[u] = (active && f < nFrames) ? (tin != tc ? (f | 0x40000000) : f) : -1;
    }
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const tb_f32x4u_t a = w[u];
      // the last chunk was loaded `over` channels early: slot k holds channel 4 gLast + k - over
      const float s0 = over == 0 ? a.x : over == 1 ? a.y : over == 2 ? a.z : a.w;
      const float s1 = over == 0 ? a.y : over == 1 ? a.z : over == 2 ? a.w : 0.f;
      const float s2 = over == 0 ? a.z : over == 1 ? a.w : 0.f;
      const float s3 = over == 0 ? a.w : 0.f;
      const tb_f32x2_t lo = {last ? s0 : a.x, last ? s1 : a.y}, hi = {last ? s2 : a.z, last ? s3 : a.w};
      uint2 pk = make_uint2(__builtin_bit_cast(uint32_t, __builtin_convertvector(lo, tb_bf16x2_t)),
                            __builtin_bit_cast(uint32_t, __builtin_convertvector(hi, tb_bf16x2_t)));
      if (pad || (fo[u] & 0x40000000)) pk = make_uint2(0u, 0u);
      if (fo[u] >= 0) *(uint2*)(slab + (fo[u] & 0x3fffffff) * FS + dstOff) = pk;
    }
  }
}

// any C (the one-channel input of the first sub-sampling convolution): aligned float4 over the HB * C floats of a frame,
// element-wise split into (mel row, channel)
template <int CP, int HB>
__device__ __forceinline__ void tb_stage_generic(unsigned char* slab, int FS, const float* __restrict__ src, int C, uint32_t magic,
                                                 size_t frameStride, int tFirst, int nFrames, int T, int tid) {
  const int runF4 = HB * C / 4;                  // float4 per frame (HB % 4 == 0)
  const int total = nFrames * runF4;
  constexpr int NV = 8;
  for (int base = 0; base < total; base += NV * 256) {
    float4 v4[NV];
    int fo[NV], eo[NV];
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int q = base + tid + 256 * u;
      const int qq = q < total ? q : total - 1;
      const int f = qq / runF4, j = qq - f * runF4;
      const int tin = tFirst + f;
      const int tc = tin < 0 ? 0 : (tin >= T ? T - 1 : tin);
      fo[u] = q < total ? (tin == tc ? f : f | 0x40000000) : -1;
      eo[u] = 4 * j;
      v4[u] = *(const float4*)(src + (size_t)tc * frameStride + 4 * j);
    }
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      if (fo[u] < 0) continue;
      const bool zero = (fo[u] & 0x40000000) != 0;
      const int f = fo[u] & 0x3fffffff;
      const float v[4] = {v4[u].x, v4[u].y, v4[u].z, v4[u].w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int e = eo[u] + k;
        const int h = C == 1 ? e : (int)(((uint64_t)e * magic) >> 32);
        const int c = e - h * C;
        *(uint16_t*)(slab + f * FS + (h * CP + c) * 2) = zero ? (uint16_t)0 : tb_bf16(v[k]);
      }
    }
  }
}

template <int CP, int NSTEP, int STRIDE>
__global__ __launch_bounds__(256) void tds_conv_bf_k(TdsBfP p) {
  static_assert((NSTEP * 16) % CP == 0, "whole taps");
  constexpr int KWP = NSTEP * 16 / CP;          // taps the K loop walks (>= kw; the weight image is zero beyond kw)
  constexpr int NF = (kTbTT - 1) * STRIDE + KWP;   // slab frames
  constexpr int FS = tb_frame_pitch(CP);        // bytes
  extern __shared__ __attribute__((aligned(16))) unsigned char slab[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int h0 = blockIdx.x * kTbHB, t0 = blockIdx.y * kTbTT, b = blockIdx.z;
  const int C = p.Cin, Co = p.Cout;

  // ---- weight fragments: column li, k = 16 s + 8 lh .. + 8 of every k-step (registers for the whole workgroup)
  tb_bf16x8_t wf[NSTEP];
  {
    const uint16_t* wr = p.wimg + (size_t)li * (NSTEP * 16) + 8 * lh;
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) wf[s] = *(const tb_bf16x8_t*)(wr + 16 * s);
  }

  // ---- slab: zero fill (the padded channels are never written; frames outside the utterance are staged as zeros).
  // One tile per workgroup: a persistent variant (launch, weight fetch and zero fill once per workgroup -- 41 of the 135 us of
  // the C = 15 forward pass are such fixed costs, run 45) was SLOWER, 144 us: with one tile per workgroup the dispatcher
  // starts the next workgroup while this one's stores drain; a persistent workgroup serialises stage / multiply / store.
  if (C < 4) {   // (tb_stage writes the whole slab itself)
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    for (int o = tid * 16; o < NF * FS; o += 256 * 16) *(uint4*)(slab + o) = z;
    __syncthreads();
  }
  {
    const float* xb = p.x + ((size_t)b * p.Tin * p.H + h0) * C;
    const size_t frameStride = (size_t)p.H * C;
    if (p.abl & 1) {}
    else if (C >= 4) tb_stage<CP, kTbHB>(slab, FS, xb, C, frameStride, t0 * STRIDE - p.padl, NF, p.Tin, tid);
    else tb_stage_generic<CP, kTbHB>(slab, FS, xb, C, p.cMagic, frameStride, t0 * STRIDE - p.padl, NF, p.Tin, tid);
  }
  __syncthreads();

  // ---- two mel rows per wave, two 32-frame halves each: rows = frames, K = (tap, c_in)
#pragma unroll 1
  for (int th = 0; th < 2; ++th) {
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    const unsigned char* a0 = slab + (STRIDE * (32 * th + li)) * FS + (2 * wave) * CP * 2;
    const unsigned char* a1 = a0 + CP * 2;
    if (!(p.abl & 2))
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
    const int lc = li < Co ? li : Co - 1;
    const float bv = p.bias ? p.bias[lc] : 0.f;
    const size_t tile = (((size_t)b * p.ToutFull + p.oOff + (size_t)p.oStep * t0) * p.H + h0) * Co;   // wave-uniform; per-output offsets stay 32-bit
    const float* addb = p.add ? p.add + tile : nullptr;
    float* yb = p.y + tile;
    const int tl0 = 32 * th + 4 * lh, tlMax = p.Tout - 1 - t0, hC = p.oStep * p.H * Co;
    float av[2][16];
    auto off = [&](int hh, int r) {
      const int tl = tl0 + (r & 3) + 8 * (r >> 2);
      return (tl < tlMax ? tl : tlMax) * hC + (2 * wave + hh) * Co + lc;
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
        if (li < Co && tl <= tlMax && (!(p.abl & 4) || av[hh][r] == 12345.f)) yb[off(hh, r)] = av[hh][r];
      }
  }
}

// weight image [32][Kp] bf16 of w [kw][Cin][Cout] (w[tap][ci][co]); the image walks kwEff logical taps j:
//   flip == 0 (forward):        img[co][j * CP + ci] = w[j][ci][co]                                    CP pads Cin
//   flip == 1 (backward-data):  img[ci][j * CP + co] = w[tapOff + tapStep * (kwEff - 1 - j)][ci][co]   CP pads Cout
// (a forward pass over dy; tapOff / tapStep select the taps of one phase of a strided convolution)
__global__ __launch_bounds__(256) void tds_bf_wprep_k(const float* __restrict__ w, int kw, int Cin, int Cout, int CP, int Kp, int kwEff,
                                                      int tapOff, int tapStep, int flip, uint16_t* __restrict__ img) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= 32 * Kp) return;
  const int n = e / Kp, k = e - n * Kp;
  const int j = k / CP, c = k - j * CP;
  float v = 0.f;
  if (j < kwEff) {
    if (flip) {
      const int tap = tapOff + tapStep * (kwEff - 1 - j);
      if (n < Cin && c < Cout && tap < kw) v = w[((size_t)tap * Cin + n) * Cout + c];
    } else if (n < Cout && c < Cin) {
      v = w[((size_t)j * Cin + c) * Cout + n];
    }
  }
  img[e] = tb_bf16(v);
}


// ---------------------------------------------------------------------------------------------------------------------
// backward-filter:  dW[tap][ci][co] = sum over (utterance, frame t, mel row h) of x[STRIDE t + tap - padl][h][ci] * dy[t][h][co]
// The contraction runs over POSITIONS, so both MFMA operands need 8 consecutive positions per lane -- k of one
// v_mfma_f32_32x32x16_bf16 = the 16 mel rows of a workgroup's block (a tap shift moves whole frames, never the alignment of a
// fragment; a time-fastest slab would misalign every odd tap); rows of the product = (tap, ci) in the forward K order
// k = tap CP + ci, 32 per tile; columns = co.
//
// The slabs are FRAME-MAJOR like the forward kernel's, slab[frame][mel row][CP channels], and the position-contiguous fragments
// come out of the LDS TRANSPOSE READ: with ds_read_b64_tr_b16 (gfx950) the 8-byte chunks of the 16 lanes of a group form a 4 x 16
// matrix (row r = the chunks of lanes 4 r .. 4 r + 3) and lane l receives COLUMN l & 15 (tools/micro/tr16_probe.hip pins this
// with permuted addresses) -- so when lane (r, cb) supplies the address of channels 4 cb .. 4 cb + 3 at mel row r, every lane
// ends up with ONE channel at four consecutive mel rows.  Two reads = the 8 positions of an MFMA operand; four consecutive rows
// (tap, ci) of a chunk never straddle a tap because CP % 4 == 0.  Per dy frame two reads for B (shared by the row tiles), per
// MFMA two for A.  Frame pitch = 16 rows * CP * 2 + 128 bytes: at CP = 16 the two 16-row groups of a read (two taps) land in
// disjoint bank halves.  Staging = tb_stage for both x and dy.  512 persistent workgroups walk the (utterance, 16 dy frames,
// 16 mel rows) items, two per CU (one stages while the other multiplies); a wave owns a quarter of the row tiles and keeps
// their accumulators in registers over every item (no cross-wave reduction); the per-workgroup partials are added in workgroup
// order by tds_bf_filter_reduce_k (deterministic).
// (First generation, run 41: mel-fastest slabs xs[frame][ci][h] filled by scalar loads and packed pair stores -- 56 - 70 TFLOP/s
// in the config-3 step; the transposition was paid in the staging.  profiles/r03_run13_*.)
constexpr int kTfWorkers = 512;

struct TdsBfFilterP {
  const float* x;    // [B][Tin][H][Cin]
  const float* dy;   // [B][Tout][H][Cout]
  float* partial;    // [workers][NRT * 32][32], then (withBias) [workers][32]: the workers' column sums of dy
  int B, Tin, Tout, H, Cin, Cout, kw, padl;
  int abl;           // timing ablations (probe library only; 0 in the product): 1 no staging, 2 no MFMAs
  int withBias;      // also sum the (bf16-rounded) dy slab per output channel: the bias gradient, no column-sum launches
};

// dw[tap][ci][co] = sum over the workgroups' partials, always in the same order: 16 outputs x 16 worker lanes per
// workgroup, a lane adds every 16th partial, the 16 lane sums are added in lane order (the first build walked all 512
// partials in ONE thread per output: 135 us per call)
__global__ __launch_bounds__(256) void tds_bf_filter_reduce_k(const float* __restrict__ partial, int workers, int rows32, int kw, int Cin, int Cout,
                                                              int CP, float* __restrict__ dw, float* __restrict__ dbias) {
  __shared__ float sm[16][17];
  const int o = threadIdx.x & 15, wl = threadIdx.x >> 4;
  const int e = blockIdx.x * 16 + o, n = kw * Cin * Cout, nAll = n + (dbias ? Cout : 0);
  float s = 0.f;
  if (e < n) {
    const int co = e % Cout, ci = (e / Cout) % Cin, tap = e / (Cin * Cout);
    const size_t at = (size_t)(tap * CP + ci) * 32 + co;
    for (int w = wl; w < workers; w += 16) s += partial[(size_t)w * rows32 * 32 + at];
  } else if (e < nAll) {   // bias gradient: the workers' column sums of dy, stored behind the product partials
    const float* pb = partial + (size_t)workers * rows32 * 32 + (e - n);
    for (int w = wl; w < workers; w += 16) s += pb[(size_t)w * 32];
  }
  sm[wl][o] = s;
  __syncthreads();
  if (wl == 0 && e < nAll) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += sm[k][o];
    if (e < n) dw[e] = t; else dbias[e - n] = t;
  }
}

constexpr int kTgTT = 16, kTgHB = 16;
__host__ __device__ constexpr int tg_pitch(int CP) { return kTgHB * CP * 2 + 128; }

typedef short tg_s16x4_t __attribute__((ext_vector_type(4)));
typedef short tg_s16x8_t __attribute__((ext_vector_type(8)));

__device__ __forceinline__ tb_bf16x8_t tg_read8(const unsigned char* p0, int step) {   // k 0 .. 3 at p0, k 4 .. 7 at p0 + step
  typedef __attribute__((address_space(3))) tg_s16x4_t* lptr;
  const tg_s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(p0));
  const tg_s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(p0 + step));
  return __builtin_bit_cast(tb_bf16x8_t, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

template <int CP, int NSTEP, int STRIDE, int CPO>
__global__ __launch_bounds__(256) void tds_conv_bf_filter_tr_k(TdsBfFilterP p, uint32_t magicI, uint32_t magicO) {
  constexpr int KWP = NSTEP * 16 / CP, Kp = NSTEP * 16;
  constexpr int NRT = (Kp + 31) / 32, NRTW = (NRT + 3) / 4;
  constexpr int NFX = (kTgTT - 1) * STRIDE + KWP;
  constexpr int FSX = tg_pitch(CP);
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int FSY = tg_pitch(CPO);
  unsigned char* xs = lds;
  unsigned char* ys = lds + NFX * FSX;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int tTiles = (p.Tout + kTgTT - 1) / kTgTT, hBlocks = p.H / kTgHB;
  const int nItems = p.B * tTiles * hBlocks;

  {   // zero fill once: the padded channels are never written again
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    const int bytes = NFX * FSX + kTgTT * FSY;
    for (int o = tid * 16; o < bytes; o += 256 * 16) *(uint4*)(lds + o) = z;
  }
  // this lane's chunk in a transpose read: mel row kRow (+ 4 for the second read), operand rows / columns idx0 .. idx0 + 3
  const int q = lane & 15, grp = lane >> 4;
  const int kRow = 8 * (grp >> 1) + (q >> 2);
  const int idx0 = 16 * (grp & 1) + 4 * (q & 3);
  f32x16 acc[NRTW];
  int rowOff[NRTW];
#pragma unroll
  for (int j = 0; j < NRTW; ++j) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    int i = 32 * (wave + 4 * j) + idx0;
    if (i > Kp - 4) i = Kp - 4;                  // rows past K: any valid chunk (never stored)
    const int tap = i / CP, ci = i - tap * CP;
    rowOff[j] = tap * FSX + kRow * CP * 2 + ci * 2;
  }
  const int bOff = kRow * CPO * 2 + (idx0 < CPO ? idx0 : 0) * 2;
  const size_t fsIn = (size_t)p.H * p.Cin, fsOut = (size_t)p.H * p.Cout;
  // bias gradient: thread (co = tid & 31, part = tid >> 5) sums 32 of an item's 256 (frame, mel row) positions of channel co
  float bsum = 0.f;
  const int bco = (tid & 31) < CPO ? (tid & 31) : 0, bpart = tid >> 5;

  for (int item = blockIdx.x; item < nItems; item += gridDim.x) {
    const int hb = item % hBlocks, tt = (item / hBlocks) % tTiles, b = item / (hBlocks * tTiles);
    const int t0 = tt * kTgTT, h0 = hb * kTgHB;
    __syncthreads();                             // the previous item's fragments have been read (and the zero fill is done)
    if (!(p.abl & 1)) {
      const float* xb = p.x + ((size_t)b * p.Tin * p.H + h0) * p.Cin;
      const float* yb = p.dy + ((size_t)b * p.Tout * p.H + h0) * p.Cout;
      if (p.Cin >= 4) tb_stage<CP, kTgHB>(xs, FSX, xb, p.Cin, fsIn, t0 * STRIDE - p.padl, NFX, p.Tin, tid);
      else tb_stage_generic<CP, kTgHB>(xs, FSX, xb, p.Cin, magicI, fsIn, t0 * STRIDE - p.padl, NFX, p.Tin, tid);
      if (p.Cout >= 4) tb_stage<CPO, kTgHB>(ys, FSY, yb, p.Cout, fsOut, t0, kTgTT, p.Tout, tid);
      else tb_stage_generic<CPO, kTgHB>(ys, FSY, yb, p.Cout, magicO, fsOut, t0, kTgTT, p.Tout, tid);
    }
    __syncthreads();
    if (p.withBias) {
#pragma unroll 8
      for (int q2 = 0; q2 < 32; ++q2) {
        const int pos = bpart * 32 + q2;   // frame pos >> 4, mel row pos & 15
        const uint32_t h16 = *(const uint16_t*)(ys + (pos >> 4) * FSY + ((pos & 15) * CPO + bco) * 2);
        bsum += __builtin_bit_cast(float, h16 << 16);
      }
    }
    // ---- one MFMA per (dy frame, row tile): k = the 16 mel rows of the block
    if (!(p.abl & 2))
#pragma unroll 4
    for (int t = 0; t < kTgTT; ++t) {
      const tb_bf16x8_t bf = tg_read8(ys + t * FSY + bOff, 4 * CPO * 2);
      // every wave multiplies NRTW row tiles, present or not (a tile past NRT reads clamped rows and is never stored): a
      // wave-uniform `if` around the MFMA made hipcc branch per MFMA and copy the accumulators AGPR <-> VGPR around each --
      // the loop took 183 of the kernel's 295 us (run 45 ablations)
#pragma unroll
      for (int j = 0; j < NRTW; ++j) {
        const tb_bf16x8_t af = tg_read8(xs + (t * STRIDE) * FSX + rowOff[j], 4 * CP * 2);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[j], 0, 0, 0);
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
  if (p.withBias) {   // the 8 parts of a channel in part order (deterministic), one row of 32 sums per workgroup
    __syncthreads();
    float* sb = (float*)lds;
    sb[tid] = bsum;
    __syncthreads();
    if (tid < 32) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) t += sb[32 * k + tid];
      p.partial[(size_t)gridDim.x * (NRT * 32) * 32 + (size_t)blockIdx.x * 32 + tid] = tid < p.Cout ? t : 0.f;
    }
  }
}

static inline int tb_cp(int C) { return C <= 16 ? 16 : C <= 24 ? 24 : 32; }
static inline int tb_abl() { const char* e = tune_env("W2L_TBF_ABL"); return e ? atoi(e) : 0; }
static inline uint32_t tb_magic(int C) { return C == 1 ? 0u : (uint32_t)((0x100000000ull + C - 1) / C); }

template <int CP, int NSTEP, int STRIDE, int CPO>
static int tb_launch_filter(const TdsBfFilterP& p0, float* dw, float* dbias, hipStream_t s) {
  constexpr int KWP = NSTEP * 16 / CP, Kp = NSTEP * 16, NRT = (Kp + 31) / 32, NFX = (kTgTT - 1) * STRIDE + KWP;
  TdsBfFilterP p = p0;
  const int tTiles = (p.Tout + kTgTT - 1) / kTgTT, nItems = p.B * tTiles * (p.H / kTgHB);
  const int workers = nItems < kTfWorkers ? nItems : kTfWorkers;
  const size_t need = ((size_t)workers * NRT * 32 * 32 + (size_t)workers * 32) * sizeof(float);
  if (need > kSkScratchBytes) return W2L_EUNSUPPORTED;
  p.withBias = dbias ? 1 : 0;
  const size_t shmem = (size_t)NFX * tg_pitch(CP) + (size_t)kTgTT * tg_pitch(CPO);
  if (shmem > 80 * 1024) return W2L_EUNSUPPORTED;   // two workgroups per CU: one stages while the other multiplies
  p.partial = sk_scratch(s, kSkScratchBytes);
  if (!p.partial) return W2L_EHIP;
  static bool attr[64] = {};
  if (first_on_device(attr)) {
    W2L_HIP_CHECK(hipFuncSetAttribute((const void*)tds_conv_bf_filter_tr_k<CP, NSTEP, STRIDE, CPO>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
  }
  hipLaunchKernelGGL((tds_conv_bf_filter_tr_k<CP, NSTEP, STRIDE, CPO>), dim3((unsigned)workers), dim3(256), shmem, s, p, tb_magic(p.Cin), tb_magic(p.Cout));
  const int n = p.kw * p.Cin * p.Cout + (dbias ? p.Cout : 0);
  hipLaunchKernelGGL(tds_bf_filter_reduce_k, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, s, p.partial, workers, NRT * 32, p.kw, p.Cin, p.Cout, CP, dw,
                     dbias);
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// geometry of a w2l_conv_desc: padded channel counts, k-steps, the phases of its backward-data pass
static inline int tb_nstep(int kw, int CP) {
  int n = (kw * CP + 15) / 16;
  while ((n * 16) % CP) ++n;
  return n;
}
struct TbPhase { int kwf, c0, U, padl, tapOff; };
struct TbGeom {
  int stride, To;
  int CPf, NSf;        // forward and backward-filter: CP pads Cin, k-steps over kw taps
  int CPb, NSb;        // backward-data: CP pads Cout (the channels of dy), k-steps over the longest phase
  int phases;
  TbPhase ph[2];
};

// the instantiated (CP, k-steps, stride) triples.  Streaming recipe (am_500ms_future_context.arch): TDS blocks (16, 9) (24, 15)
// (24, 18) (32, 22); sub-sampling convolutions 1 -> 15 -> 19 (kw 10, stride 2), 19 -> 23 (kw 12, stride 2), 23 -> 27 (kw 11):
// forward / filter (16, 10, s2) (24, 18, s2) (24, 18), backward phases (16, 5) (24, 9) (32, 22).  sota/2019 TDS-CTC (kw 21):
// blocks (16, 21) (24, 33); sub-sampling 1 -> 10 -> 14 -> 18 at stride 2: (16, 21, s2), phases (16, 11) (24, 18).
#define W2L_TB_FWD_LIST(X) X(16, 9, 1) X(16, 21, 1) X(24, 15, 1) X(24, 18, 1) X(24, 33, 1) X(32, 22, 1) \
                           X(16, 5, 1) X(16, 11, 1) X(24, 9, 1) X(16, 10, 2) X(24, 18, 2) X(16, 21, 2)
// (filter gradient: + the padded channel count of dy)
#define W2L_TB_FILTER_LIST(X) X(16, 9, 1, 16) X(16, 21, 1, 16) X(24, 15, 1, 24) X(24, 18, 1, 24) X(24, 33, 1, 24) X(32, 22, 1, 32) \
                              X(16, 10, 2, 16) X(16, 10, 2, 24) X(24, 18, 2, 24) X(24, 18, 1, 32) X(16, 21, 2, 16) X(16, 21, 2, 24)

static bool tb_has_fwd(int CP, int NS, int ST) {
#define X(a, b, c) if (CP == a && NS == b && ST == c) return true;
  W2L_TB_FWD_LIST(X)
#undef X
  return false;
}
static bool tb_has_filter(int CP, int NS, int ST, int CPO) {
#define X(a, b, c, d) if (CP == a && NS == b && ST == c && CPO == d) return true;
  W2L_TB_FILTER_LIST(X)
#undef X
  return false;
}

static bool tb_geometry(const w2l_conv_desc* d, TbGeom& g) {
  if (!d || (d->stride != 1 && d->stride != 2) || d->Cin < 1 || d->Cin > 32 || d->Cout < 1 || d->Cout > 32 || d->kw < d->stride) return false;
  if (d->B < 1 || d->T < 1 || d->padl < 0 || d->padr < 0 || d->H % kTbHB != 0) return false;   // whole mel blocks (16-byte aligned runs of 8 C floats)
  const int n = d->T + d->padl + d->padr - d->kw;
  if (n < 0) return false;
  const int st = d->stride;
  g.stride = st;
  g.To = n / st + 1;
  g.CPf = tb_cp(d->Cin); g.NSf = tb_nstep(d->kw, g.CPf);
  g.CPb = tb_cp(d->Cout); g.NSb = tb_nstep((d->kw + st - 1) / st, g.CPb);
  g.phases = 0;
  for (int f = 0; f < st; ++f) {   // input frames ti with (ti + padl) mod st == f are reached by the taps f, f + st, ... only
    TbPhase& q = g.ph[g.phases];
    q.kwf = (d->kw - f + st - 1) / st;
    q.c0 = (((f - d->padl) % st) + st) % st;
    if (q.c0 >= d->T) continue;
    q.U = (d->T - q.c0 + st - 1) / st;
    const int s0 = (q.c0 + d->padl - f) / st;   // dy frame of tap f at the phase's first input frame
    q.padl = q.kwf - 1 - s0;
    q.tapOff = f;
    ++g.phases;
  }
  // one answer for the three passes: a layer is either wholly on the bf16 kernels or wholly on the fp32 entry points (the
  // filter gradient stages 16-row mel blocks)
  return g.phases == st && tb_has_fwd(g.CPf, g.NSf, st) && tb_has_fwd(g.CPb, g.NSb, 1) && d->H % kTgHB == 0 &&
         tb_has_filter(g.CPf, g.NSf, st, g.CPb);
}

template <int CP, int NSTEP, int STRIDE>
static int tb_launch(const TdsBfP& p, hipStream_t s) {
  constexpr int KWP = NSTEP * 16 / CP;
  const size_t shmem = (size_t)((kTbTT - 1) * STRIDE + KWP) * tb_frame_pitch(CP);
  static bool attr[64] = {};
  if (shmem > 64 * 1024 && first_on_device(attr)) {
    W2L_HIP_CHECK(hipFuncSetAttribute((const void*)tds_conv_bf_k<CP, NSTEP, STRIDE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  }
  const dim3 grid((unsigned)(p.H / kTbHB), (unsigned)((p.Tout + kTbTT - 1) / kTbTT), (unsigned)p.B);
  hipLaunchKernelGGL((tds_conv_bf_k<CP, NSTEP, STRIDE>), grid, dim3(256), shmem, s, p);
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}

static int tb_dispatch(int CP, int NS, int ST, const TdsBfP& p, hipStream_t s) {
#define X(a, b, c) if (CP == a && NS == b && ST == c) return tb_launch<a, b, c>(p, s);
  W2L_TB_FWD_LIST(X)
#undef X
  return W2L_EUNSUPPORTED;
}

static int tb_dispatch_filter(int CP, int NS, int ST, int CPO, const TdsBfFilterP& p, float* dw, float* dbias, hipStream_t s) {
#define X(a, b, c, d) if (CP == a && NS == b && ST == c && CPO == d) return tb_launch_filter<a, b, c, d>(p, dw, dbias, s);
  W2L_TB_FILTER_LIST(X)
#undef X
  return W2L_EUNSUPPORTED;
}

}  // namespace w2l

using namespace w2l;

// bf16 elements of ONE weight image buffer of this geometry -- the forward image, or the backward-data images of all phases
// of a strided convolution, whichever is larger (0: the geometry has no bf16 kernel -- stay on w2l_conv_*)
W2L_API size_t w2l_tds_conv_bf16_image_elems(const w2l_conv_desc* d) {
  TbGeom g;
  if (!tb_geometry(d, g)) return 0;
  const size_t f = (size_t)32 * g.NSf * 16, b = (size_t)g.phases * 32 * g.NSb * 16;
  return f > b ? f : b;
}

// once per step: the forward and the backward-data weight images of w [kw][Cin][Cout] (fp32 master weights)
W2L_API int w2l_tds_conv_bf16_prepare(const w2l_conv_desc* d, const float* w, uint16_t* imgForward, uint16_t* imgBackward,
                                      w2l_stream_t stream) {
  TbGeom g;
  if (!w || (!imgForward && !imgBackward)) return W2L_EINVAL;
  if (!tb_geometry(d, g)) return W2L_EUNSUPPORTED;
  if (imgForward) {
    const int Kp = g.NSf * 16;
    hipLaunchKernelGGL(tds_bf_wprep_k, dim3((unsigned)((32 * Kp + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, d->kw, d->Cin, d->Cout,
                       g.CPf, Kp, d->kw, 0, 1, 0, imgForward);
  }
  if (imgBackward) {
    const int Kp = g.NSb * 16;
    for (int f = 0; f < g.phases; ++f)
      hipLaunchKernelGGL(tds_bf_wprep_k, dim3((unsigned)((32 * Kp + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, d->kw, d->Cin, d->Cout,
                         g.CPb, Kp, g.ph[f].kwf, g.ph[f].tapOff, g.stride, 1, imgBackward + (size_t)f * 32 * Kp);
  }
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}

// y [B][To][H][Cout] = (relu)(conv(bf16(x), imgForward) + bias)
W2L_API int w2l_tds_conv_bf16_forward(const w2l_conv_desc* d, const float* x, const uint16_t* imgForward, const float* bias, float* y,
                                      int relu, w2l_stream_t stream) {
  TbGeom g;
  if (!x || !imgForward || !y) return W2L_EINVAL;
  if (!tb_geometry(d, g)) return W2L_EUNSUPPORTED;
  TdsBfP p{x, imgForward, bias, nullptr, y, d->B, d->T, g.To, d->H, d->Cin, d->Cout, d->padl, relu, 0, 1, g.To, tb_magic(d->Cin), tb_abl()};
  prof_begin((hipStream_t)stream, 2.0 * d->B * (double)g.To * d->H * d->Cin * (double)d->Cout * d->kw, PROF_TDSCONV);
  const int st = tb_dispatch(g.CPf, g.NSf, g.stride, p, (hipStream_t)stream);
  prof_end((hipStream_t)stream);
  return st;
}

// dx [B][T][H][Cin] = (add +) conv^T(bf16(dy), w): forward passes over dy with the flipped images, one per phase of the stride
W2L_API int w2l_tds_conv_bf16_backward_data(const w2l_conv_desc* d, const float* dy, const uint16_t* imgBackward, const float* add,
                                            float* dx, w2l_stream_t stream) {
  TbGeom g;
  if (!dy || !imgBackward || !dx) return W2L_EINVAL;
  if (!tb_geometry(d, g)) return W2L_EUNSUPPORTED;
  prof_begin((hipStream_t)stream, 2.0 * d->B * (double)g.To * d->H * d->Cin * (double)d->Cout * d->kw, PROF_TDS_BWD_DATA);
  int st = W2L_OK;
  for (int f = 0; f < g.phases && st == W2L_OK; ++f) {
    const TbPhase& q = g.ph[f];
    TdsBfP p{dy, imgBackward + (size_t)f * 32 * g.NSb * 16, nullptr, add, dx, d->B, g.To, q.U, d->H, d->Cout, d->Cin, q.padl, 0,
             q.c0, g.stride, d->T, tb_magic(d->Cout), tb_abl()};
    st = tb_dispatch(g.CPb, g.NSb, 1, p, (hipStream_t)stream);
  }
  prof_end((hipStream_t)stream);
  return st;
}

// dw [kw][Cin][Cout] = x (*) dy on bf16-rounded operands (fp32 accumulation).  H must be a multiple of 16.
// dbias [Cout] (may be null): the column sums of the bf16-rounded dy over [B To H], summed from the dy slabs the kernel stages
// anyway (fixed order: deterministic) -- no column-sum launches for the convolution's bias.
W2L_API int w2l_tds_conv_bf16_backward_filter_bias(const w2l_conv_desc* d, const float* x, const float* dy, float* dw, float* dbias,
                                                   w2l_stream_t stream) {
  TbGeom g;
  if (!x || !dy || !dw) return W2L_EINVAL;
  if (!tb_geometry(d, g)) return W2L_EUNSUPPORTED;
  TdsBfFilterP p{x, dy, nullptr, d->B, d->T, g.To, d->H, d->Cin, d->Cout, d->kw, d->padl, tb_abl(), 0};
  prof_begin((hipStream_t)stream, 2.0 * d->B * (double)g.To * d->H * d->Cin * (double)d->Cout * d->kw, PROF_TDS_BWD_FILTER);
  const int st = tb_dispatch_filter(g.CPf, g.NSf, g.stride, g.CPb, p, dw, dbias, (hipStream_t)stream);
  prof_end((hipStream_t)stream);
  return st;
}

// (the weight gradient alone: the bias gradient by the caller's means, e.g. an fp32 w2l_colsum of dy over [B To H][Cout])
W2L_API int w2l_tds_conv_bf16_backward_filter(const w2l_conv_desc* d, const float* x, const float* dy, float* dw, w2l_stream_t stream) {
  return w2l_tds_conv_bf16_backward_filter_bias(d, x, dy, dw, nullptr, stream);
}
