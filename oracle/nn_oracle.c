/*
 * oracle/nn_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain loops, fp64 accumulation, fp32 storage) of the network
 * operators on the wav2letter acoustic-training hot path, in the REFERENCE's
 * memory layouts (ArrayFire column-major dims (W=T, H, C, N=B) == row-major C
 * array [B][C][H][T]):
 *
 *   Conv2D kw x 1 (time convolution)   arch grammar: recipes/joint_training_vox_populi/cpc/SequentialBuilder.cpp:203-252, :285-301
 *                                       arithmetic:   recipes/streaming_convnets/inference/inference/module/nn/backend/fbgemm/Conv1dFbGemm.cpp:104-185
 *                                       (cross-correlation: tap ts reads frame t*stride+ts, :113-114; bias pre-filled, :150-156)
 *   weight layout                       recipes/streaming_convnets/tools/StreamingTDSModelConverter.cpp:56-89
 *                                       (Flashlight (kw,1,cin,cout) == C array [cout][cin][1][kw])
 *   Linear                              SequentialBuilder.cpp:305-313; weight dims (out,in) == C array [in][out]
 *   LayerNorm                           SequentialBuilder.cpp:358-377; streaming form
 *                                       recipes/streaming_convnets/inference/inference/module/nn/LayerNorm.cpp:18-69
 *                                       + inference/common/Functions.cpp:15-33
 *   GatedLinearUnit / ReLU / Dropout    SequentialBuilder.cpp:467-473, :423-428, :388-394
 *   WeightNorm (param order v,g,bias)   SequentialBuilder.cpp:379-386; recipes/utilities/convlm_serializer/Utils.cpp:112-143
 *   TDSBlock                            SequentialBuilder.cpp:254-268; recipes/streaming_convnets/inference/inference/module/nn/TDSBlock.cpp:58-70
 *
 * Pinning: the two golden vectors the reference's own tests hold for this path,
 *   inference/module/test/Conv1dTest.cpp:30-104 and TDSBlockTest.cpp:27-188
 * (tests/golden/ fixtures, tests/test_oracle_nn.py), plus torch-CPU cross-checks of
 * every operator forward and backward.  The train-time Flashlight modules
 * themselves (fl::Conv2D, fl::LayerNorm eps, fl::TDSBlock) are un-vendored;
 * where the streaming library and the train-time module differ (LayerNorm
 * epsilon) both forms are provided.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define W2L_EXPORT __attribute__((visibility("default")))

/* output length of a time convolution */
W2L_EXPORT int w2l_oracle_conv_out_len(int T, int kw, int stride, int padl, int padr, int dil) {
  int eff = dil * (kw - 1) + 1;
  int n = T + padl + padr - eff;
  if (n < 0) return 0;
  return n / stride + 1;
}

/* Flashlight PaddingMode::SAME (pad = -1) for one axis: symmetric pad p on BOTH
 * sides (corroborated by StreamingTDSModelConverter.cpp:71-79) */
W2L_EXPORT int w2l_oracle_same_pad(int T, int kw, int stride, int dil) {
  int total;
  if (T % stride == 0) total = (kw - 1) * dil - stride + 1;
  else total = (kw - 1) * dil - (T % stride) + 1;
  if (total < 0) total = 0;
  return (total + 1) / 2;
}

/* ---- Conv2D (kw x 1), layout x [B][Cin][H][T], w [Cout][Cin][kw], y [B][Cout][H][To] */
W2L_EXPORT void w2l_oracle_conv_fwd(const float* x, const float* w, const float* bias, float* y,
                                    int B, int Cin, int Cout, int H, int T, int kw,
                                    int stride, int padl, int padr, int dil) {
  int To = w2l_oracle_conv_out_len(T, kw, stride, padl, padr, dil);
#pragma omp parallel for collapse(2)
  for (int b = 0; b < B; ++b)
    for (int co = 0; co < Cout; ++co)
      for (int h = 0; h < H; ++h)
        for (int to = 0; to < To; ++to) {
          double acc = bias ? (double)bias[co] : 0.0;
          for (int ci = 0; ci < Cin; ++ci)
            for (int k = 0; k < kw; ++k) {
              int t = to * stride + k * dil - padl;
              if (t < 0 || t >= T) continue;
              acc += (double)x[(((size_t)b * Cin + ci) * H + h) * T + t] *
                     (double)w[((size_t)co * Cin + ci) * kw + k];
            }
          y[(((size_t)b * Cout + co) * H + h) * To + to] = (float)acc;
        }
}

W2L_EXPORT void w2l_oracle_conv_bwd_data(const float* dy, const float* w, float* dx,
                                         int B, int Cin, int Cout, int H, int T, int kw,
                                         int stride, int padl, int padr, int dil) {
  int To = w2l_oracle_conv_out_len(T, kw, stride, padl, padr, dil);
#pragma omp parallel for collapse(2)
  for (int b = 0; b < B; ++b)
    for (int ci = 0; ci < Cin; ++ci)
      for (int h = 0; h < H; ++h)
        for (int t = 0; t < T; ++t) {
          double acc = 0;
          for (int co = 0; co < Cout; ++co)
            for (int k = 0; k < kw; ++k) {
              int num = t + padl - k * dil;
              if (num < 0 || num % stride) continue;
              int to = num / stride;
              if (to >= To) continue;
              acc += (double)dy[(((size_t)b * Cout + co) * H + h) * To + to] *
                     (double)w[((size_t)co * Cin + ci) * kw + k];
            }
          dx[(((size_t)b * Cin + ci) * H + h) * T + t] = (float)acc;
        }
}

W2L_EXPORT void w2l_oracle_conv_bwd_filter(const float* x, const float* dy, float* dw, float* dbias,
                                           int B, int Cin, int Cout, int H, int T, int kw,
                                           int stride, int padl, int padr, int dil) {
  int To = w2l_oracle_conv_out_len(T, kw, stride, padl, padr, dil);
#pragma omp parallel for collapse(2)
  for (int co = 0; co < Cout; ++co)
    for (int ci = 0; ci < Cin; ++ci)
      for (int k = 0; k < kw; ++k) {
        double acc = 0;
        for (int b = 0; b < B; ++b)
          for (int h = 0; h < H; ++h)
            for (int to = 0; to < To; ++to) {
              int t = to * stride + k * dil - padl;
              if (t < 0 || t >= T) continue;
              acc += (double)x[(((size_t)b * Cin + ci) * H + h) * T + t] *
                     (double)dy[(((size_t)b * Cout + co) * H + h) * To + to];
            }
        dw[((size_t)co * Cin + ci) * kw + k] = (float)acc;
      }
  if (dbias)
    for (int co = 0; co < Cout; ++co) {
      double acc = 0;
      for (int b = 0; b < B; ++b)
        for (size_t e = 0; e < (size_t)H * To; ++e)
          acc += dy[((size_t)b * Cout + co) * H * To + e];
      dbias[co] = (float)acc;
    }
}

/* ---- Linear: y[M][out] = x[M][in] . W[in][out] + b[out] ------------------- */
W2L_EXPORT void w2l_oracle_linear_fwd(const float* x, const float* w, const float* bias, float* y,
                                      int M, int in, int out) {
#pragma omp parallel for
  for (int m = 0; m < M; ++m) {
    double* acc = (double*)malloc(sizeof(double) * out);
    for (int o = 0; o < out; ++o) acc[o] = bias ? bias[o] : 0.0;
    for (int i = 0; i < in; ++i) {
      double xv = x[(size_t)m * in + i];
      const float* wr = w + (size_t)i * out;
      for (int o = 0; o < out; ++o) acc[o] += xv * (double)wr[o];
    }
    for (int o = 0; o < out; ++o) y[(size_t)m * out + o] = (float)acc[o];
    free(acc);
  }
}

W2L_EXPORT void w2l_oracle_linear_bwd(const float* x, const float* w, const float* dy,
                                      float* dx, float* dw, float* dbias, int M, int in, int out) {
  if (dx) {
#pragma omp parallel for
    for (int m = 0; m < M; ++m)
      for (int i = 0; i < in; ++i) {
        double acc = 0;
        const float* wr = w + (size_t)i * out;
        const float* dr = dy + (size_t)m * out;
        for (int o = 0; o < out; ++o) acc += (double)dr[o] * (double)wr[o];
        dx[(size_t)m * in + i] = (float)acc;
      }
  }
  if (dw) {
#pragma omp parallel for
    for (int i = 0; i < in; ++i) {
      double* acc = (double*)calloc(out, sizeof(double));
      for (int m = 0; m < M; ++m) {
        double xv = x[(size_t)m * in + i];
        const float* dr = dy + (size_t)m * out;
        for (int o = 0; o < out; ++o) acc[o] += xv * (double)dr[o];
      }
      for (int o = 0; o < out; ++o) dw[(size_t)i * out + o] = (float)acc[o];
      free(acc);
    }
  }
  if (dbias)
    for (int o = 0; o < out; ++o) {
      double acc = 0;
      for (int m = 0; m < M; ++m) acc += dy[(size_t)m * out + o];
      dbias[o] = (float)acc;
    }
}

/* ---- LayerNorm over `inner` contiguous elements per group (scalar affine) ---
 * train-time form: y = gamma * (x - mu) / sqrt(var + eps) + beta, biased var.
 * For axes {0,1,2} on [B][C][H][T]: groups = B, inner = C*H*T.
 * streaming == 1 selects the inference-library form (no eps; stddev clamp,
 * LayerNorm.cpp:65-67; var = E[x^2] - mu^2, Functions.cpp:15-20). */
W2L_EXPORT void w2l_oracle_layernorm_fwd(const float* x, float* y, float* mean, float* rstd,
                                         int groups, size_t inner, float gamma, float beta,
                                         float eps, int streaming) {
#pragma omp parallel for
  for (int g = 0; g < groups; ++g) {
    const float* xr = x + (size_t)g * inner;
    double s = 0, ss = 0;
    for (size_t e = 0; e < inner; ++e) { s += xr[e]; ss += (double)xr[e] * xr[e]; }
    double mu = s / inner, var;
    if (streaming) var = ss / inner - mu * mu;
    else { var = 0; for (size_t e = 0; e < inner; ++e) { double d = xr[e] - mu; var += d * d; } var /= inner; }
    double r;
    if (streaming) { double sd = sqrt(var > 0 ? var : 0); if (sd <= 1e-5) sd = 1.0; r = 1.0 / sd; }
    else r = 1.0 / sqrt(var + eps);
    if (mean) mean[g] = (float)mu;
    if (rstd) rstd[g] = (float)r;
    for (size_t e = 0; e < inner; ++e)
      y[(size_t)g * inner + e] = (float)(gamma * ((xr[e] - mu) * r) + beta);
  }
}

W2L_EXPORT void w2l_oracle_layernorm_bwd(const float* x, const float* dy, float* dx,
                                         double* dgamma, double* dbeta,
                                         int groups, size_t inner, float gamma, float eps) {
  double dgs = 0, dbs = 0;
#pragma omp parallel for reduction(+ : dgs, dbs)
  for (int g = 0; g < groups; ++g) {
    const float* xr = x + (size_t)g * inner;
    const float* dr = dy + (size_t)g * inner;
    double s = 0;
    for (size_t e = 0; e < inner; ++e) s += xr[e];
    double mu = s / inner, var = 0;
    for (size_t e = 0; e < inner; ++e) { double d = xr[e] - mu; var += d * d; }
    var /= inner;
    double r = 1.0 / sqrt(var + eps);
    double sdy = 0, sdyx = 0;
    for (size_t e = 0; e < inner; ++e) {
      double xh = (xr[e] - mu) * r;
      sdy += dr[e]; sdyx += dr[e] * xh;
    }
    dgs += sdyx; dbs += sdy;
    for (size_t e = 0; e < inner; ++e) {
      double xh = (xr[e] - mu) * r;
      dx[(size_t)g * inner + e] =
          (float)(gamma * r * (dr[e] - sdy / inner - xh * sdyx / inner));
    }
  }
  if (dgamma) *dgamma = dgs;
  if (dbeta) *dbeta = dbs;
}

/* ---- GLU along an axis: x viewed [outer][2*half][inner] -> y [outer][half][inner] */
W2L_EXPORT void w2l_oracle_glu_fwd(const float* x, float* y, size_t outer, size_t half, size_t inner) {
  for (size_t o = 0; o < outer; ++o)
    for (size_t c = 0; c < half; ++c)
      for (size_t i = 0; i < inner; ++i) {
        double a = x[(o * 2 * half + c) * inner + i];
        double g = x[(o * 2 * half + half + c) * inner + i];
        y[(o * half + c) * inner + i] = (float)(a / (1.0 + exp(-g)));
      }
}

W2L_EXPORT void w2l_oracle_glu_bwd(const float* x, const float* dy, float* dx,
                                   size_t outer, size_t half, size_t inner) {
  for (size_t o = 0; o < outer; ++o)
    for (size_t c = 0; c < half; ++c)
      for (size_t i = 0; i < inner; ++i) {
        double a = x[(o * 2 * half + c) * inner + i];
        double g = x[(o * 2 * half + half + c) * inner + i];
        double s = 1.0 / (1.0 + exp(-g));
        double d = dy[(o * half + c) * inner + i];
        dx[(o * 2 * half + c) * inner + i] = (float)(d * s);
        dx[(o * 2 * half + half + c) * inner + i] = (float)(d * a * s * (1.0 - s));
      }
}

/* ---- WeightNorm: w = v * g / ||v||, norm over everything except the kept
 * (output) axis; v viewed [outer][nout][inner] with the kept axis in the middle.
 *   Conv2D under "WN 3": v [Cout][Cin][kw]  -> outer = 1,  nout = Cout, inner = Cin*kw
 *   Linear under "WN 0": W [in][out]        -> outer = in, nout = out,  inner = 1 */
W2L_EXPORT void w2l_oracle_weightnorm_fwd(const float* v, const float* g, float* w, float* norm,
                                          size_t outer, size_t nout, size_t inner) {
  for (size_t o = 0; o < nout; ++o) {
    double ss = 0;
    for (size_t a = 0; a < outer; ++a)
      for (size_t i = 0; i < inner; ++i) { double t = v[(a * nout + o) * inner + i]; ss += t * t; }
    double n = sqrt(ss);
    if (norm) norm[o] = (float)n;
    for (size_t a = 0; a < outer; ++a)
      for (size_t i = 0; i < inner; ++i)
        w[(a * nout + o) * inner + i] = (float)(v[(a * nout + o) * inner + i] * ((double)g[o] / n));
  }
}

W2L_EXPORT void w2l_oracle_weightnorm_bwd(const float* v, const float* g, const float* dw,
                                          float* dv, float* dg,
                                          size_t outer, size_t nout, size_t inner) {
  for (size_t o = 0; o < nout; ++o) {
    double ss = 0, dot = 0;
    for (size_t a = 0; a < outer; ++a)
      for (size_t i = 0; i < inner; ++i) {
        double t = v[(a * nout + o) * inner + i];
        ss += t * t; dot += t * dw[(a * nout + o) * inner + i];
      }
    double n = sqrt(ss);
    dg[o] = (float)(dot / n);
    for (size_t a = 0; a < outer; ++a)
      for (size_t i = 0; i < inner; ++i) {
        size_t k = (a * nout + o) * inner + i;
        dv[k] = (float)((double)g[o] / n * (dw[k] - v[k] * dot / ss));
      }
  }
}

/* ---- Dropout with a stateless counter-hash mask (shared bit-exactly with the
 * device kernels: wav2letter_amd/csrc/common.hpp w2l_keep()).  Integer-only
 * decision => identical masks on CPU and GPU. */
static inline uint32_t w2l_hash32(uint32_t idx, uint32_t seed, uint32_t stream) {
  uint32_t h = idx * 0x9E3779B1u + seed;
  h ^= stream * 0x85EBCA77u;
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return h;
}

W2L_EXPORT uint32_t w2l_oracle_dropout_threshold(double p) {
  double t = p * 16777216.0;
  if (t < 0) t = 0;
  if (t > 16777216.0) t = 16777216.0;
  return (uint32_t)t;
}

W2L_EXPORT int w2l_oracle_keep(uint64_t idx, uint32_t seed, uint32_t stream, uint32_t thr) {
  uint32_t h = w2l_hash32((uint32_t)idx ^ (uint32_t)(idx >> 32) * 0x27D4EB2Fu, seed, stream);
  return (h >> 8) >= thr;
}

/* y = keep ? x/(1-p) : 0 ; element index = flat index in the DEVICE layout, so
 * the caller passes the permutation-free flat index space it wants compared. */
W2L_EXPORT void w2l_oracle_dropout(const float* x, float* y, size_t n, double p,
                                   uint32_t seed, uint32_t stream) {
  uint32_t thr = w2l_oracle_dropout_threshold(p);
  float sc = (float)(1.0 / (1.0 - p));
  for (size_t i = 0; i < n; ++i)
    y[i] = w2l_oracle_keep(i, seed, stream, thr) ? x[i] * sc : 0.0f;
}

/* ---- fl::SpecAugment (arch token SAUG, recipes/joint_training_vox_populi/cpc/SequentialBuilder.cpp:602-613;
 * recipe line recipes/sota/2019/am_arch/am_tds_ctc.arch:1 "SAUG 80 27 2 100 1.0 2").  The arithmetic is
 * un-vendored Flashlight (fl/contrib/modules/SpecAugment.cpp) => PARITY UNPINNED; restated as recalled:
 *   for i < nFMask: f  = randInt[0, fMaskF)  ; f0 = randInt[0, F - f) ; x(:, f0 .. f0+f, :, :) = 0
 *   Tm = min(tMaskT, (int)(T * tMaskP)); if Tm > 0, for i < nTMask:
 *                   t  = randInt[0, Tm)      ; t0 = randInt[0, T - t) ; x(t0 .. t0+t, :, :, :) = 0
 * af::seq(a, b) is INCLUSIVE, so a draw of f masks f + 1 channels; af::span over the batch dim => ONE set of
 * masks per batch; time warping (the first SAUG number) is not implemented by the reference module.
 * The random draws are the stateless hash shared with the device (stream 1000): draw k of the call is
 * hash32(k, seed, 1000) reduced modulo the range.  x is frame-major [B][T][F]; in place. */
W2L_EXPORT int w2l_oracle_specaugment(float* x, int B, int T, int F, int fMaskF, int nFMask, int tMaskT,
                                      float tMaskP, int nTMask, uint32_t seed, int* masks /* [4*8] f0,f1,t0,t1 incl. */) {
  if (F < fMaskF || nFMask > 8 || nTMask > 8) return 1;
  int f0s[8], f1s[8], t0s[8], t1s[8];
  for (int k = 0; k < 8; ++k) { f0s[k] = t0s[k] = 0; f1s[k] = t1s[k] = -1; }
  for (int k = 0; k < nFMask && fMaskF > 0; ++k) {
    int f = (int)(w2l_hash32(4 * k, seed, 1000) % (uint32_t)fMaskF);
    int f0 = (int)(w2l_hash32(4 * k + 1, seed, 1000) % (uint32_t)(F - f));
    f0s[k] = f0; f1s[k] = f0 + f;
  }
  int Tm = (int)((float)T * tMaskP);
  if (Tm > tMaskT) Tm = tMaskT;
  if (Tm > T) Tm = T;
  for (int k = 0; k < nTMask && Tm > 0; ++k) {
    int t = (int)(w2l_hash32(4 * k + 2, seed, 1000) % (uint32_t)Tm);
    int t0 = (int)(w2l_hash32(4 * k + 3, seed, 1000) % (uint32_t)(T - t));
    t0s[k] = t0; t1s[k] = t0 + t;
  }
  for (int b = 0; b < B; ++b)
    for (int t = 0; t < T; ++t)
      for (int f = 0; f < F; ++f) {
        int m = 0;
        for (int k = 0; k < 8; ++k) m |= (f >= f0s[k] && f <= f1s[k]) || (t >= t0s[k] && t <= t1s[k]);
        if (m) x[((size_t)b * T + t) * F + f] = 0.0f;
      }
  if (masks)
    for (int k = 0; k < 8; ++k) { masks[k] = f0s[k]; masks[8 + k] = f1s[k]; masks[16 + k] = t0s[k]; masks[24 + k] = t1s[k]; }
  return 0;
}

/* ---- streaming-library forms used only to replay the golden vectors ------- */
/* Conv1dFbGemm.cpp:104-185: frame-major x [T][groups][cin_g]; weights
 * [cout_g][kw][cin_g] shared by all groups; y [To][groups][cout_g]. */
W2L_EXPORT void w2l_oracle_streaming_conv1d(const float* x, const float* w, const float* bias,
                                            float* y, int T, int groups, int cin_g, int cout_g,
                                            int kw, int stride, int padl, int padr) {
  int Tp = T + padl + padr;
  int To = (Tp - kw) / stride + 1;
  for (int t = 0; t < To; ++t)
    for (int d = 0; d < groups; ++d)
      for (int co = 0; co < cout_g; ++co) {
        double acc = bias[co];
        for (int ts = 0; ts < kw; ++ts) {
          int ti = t * stride + ts - padl;
          if (ti < 0 || ti >= T) continue;
          for (int ci = 0; ci < cin_g; ++ci)
            acc += (double)x[((size_t)ti * groups + d) * cin_g + ci] *
                   (double)w[((size_t)co * kw + ts) * cin_g + ci];
        }
        y[((size_t)t * groups + d) * cout_g + co] = (float)acc;
      }
}
