// features.hip -- device side of the MFSC / log-mel front end (the input pipeline next to the hot path).
//
// Reference: the recipes' features are Flashlight's fl::lib::audio::Mfsc [UNVENDORED], configured as in
// recipes/streaming_convnets/inference/inference/module/feature/LogMelFeature.cpp:78-95 (25 ms frames, 10 ms stride,
// 0..fs/2, no dither, no frame mean removal, no energy, magnitude spectrum) and --filterbanks=80 / 40
// (recipes/sota/2019/librispeech/train_am_tds_ctc.cfg, recipes/conv_glu/librispeech/train.cfg).
//
// Everything per-frame before the magnitude is LINEAR in the samples (pre-emphasis, Hamming window, DFT), so it is
// folded on the host into one [400 x 2*257] matrix and the spectrum of ALL frames is ONE GEMM on overlapping rows of
// the audio (row t = samples [160 t, 160 t + 400): leading dimension 160 -- no framing copy; wav2letter_amd/features.py).
// The two kernels here are what is not a GEMM: |re + i im| (or its square) and log(max(mel, floor)) with the
// transposition to the network's input layout [B][NFEAT][T].
#include "common.hpp"

namespace w2l {

// spec[m][k] = sqrt(re^2 + im^2) (usePower: re^2 + im^2) for k < nbins, 0 for nbins <= k < ldOut
// reim [M][2*nbins]: columns [0, nbins) real parts, [nbins, 2 nbins) imaginary parts
__global__ __launch_bounds__(256) void mfsc_spectrum_k(const float* __restrict__ reim, float* __restrict__ spec, size_t M,
                                                       int nbins, int ldOut, int usePower) {
  const size_t n = M * (size_t)ldOut;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
    const size_t m = e / ldOut;
    const int k = (int)(e - m * ldOut);
    float v = 0.f;
    if (k < nbins) {
      const float re = reim[m * 2 * nbins + k], im = reim[m * 2 * nbins + nbins + k];
      const float pw = re * re + im * im;
      v = usePower ? pw : sqrtf(pw);
    }
    spec[e] = v;
  }
}

// out[b][f][t] = log(max(mel[b][t][f], floor)),  mel [B][Tp][F] (rows t >= T of an utterance are dropped)
__global__ __launch_bounds__(256) void mfsc_log_transpose_k(const float* __restrict__ mel, float* __restrict__ out, int Tp, int T,
                                                            int F, float floorv) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z, t0 = blockIdx.x * 32, f0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const int t = t0 + r, f = f0 + tx;
    tile[r][tx] = (t < T && f < F) ? mel[((size_t)b * Tp + t) * F + f] : 1.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int f = f0 + r, t = t0 + tx;
    if (f < F && t < T) out[((size_t)b * F + f) * T + t] = logf(fmaxf(tile[tx][r], floorv));
  }
}

}  // namespace w2l

using namespace w2l;

W2L_API int w2l_mfsc_spectrum(const float* reim, float* spec, size_t M, int nbins, int ldOut, int usePower,
                              w2l_stream_t stream) {
  if (!reim || !spec || M == 0 || nbins <= 0 || ldOut < nbins) return W2L_EINVAL;
  size_t g = (M * (size_t)ldOut + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(mfsc_spectrum_k, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, reim, spec, M, nbins, ldOut, usePower);
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}

W2L_API int w2l_mfsc_log_transpose(const float* mel, float* out, int B, int Tp, int T, int F, float floorv,
                                   w2l_stream_t stream) {
  if (!mel || !out || B <= 0 || T <= 0 || Tp < T || F <= 0 || !(floorv > 0.f)) return W2L_EINVAL;
  dim3 grid((unsigned)((T + 31) / 32), (unsigned)((F + 31) / 32), (unsigned)B);
  hipLaunchKernelGGL(mfsc_log_transpose_k, grid, dim3(256), 0, (hipStream_t)stream, mel, out, Tp, T, F, floorv);
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}
