/*
 * w2l_hip.h -- C ABI of libw2l_hip.so: the MI355X (gfx950) implementation of the
 * wav2letter acoustic-training hot path.
 *
 * Every pointer is a DEVICE pointer unless its name starts with h_.  All entry
 * points are asynchronous on `stream` (a hipStream_t passed as void*; NULL =
 * the default stream), re-entrant, keep no global state and return a
 * w2l_status (0 = OK) instead of throwing.  The caller owns every buffer,
 * including the workspaces sized by the *_workspace_size queries.
 *
 * Section 1 mirrors, one for one, the static functions of Flashlight's
 *   fl::lib::{cpu,cuda}::{ForceAlignmentCriterion, FullConnectionCriterion,
 *   ViterbiPath, ConnectionistTemporalClassificationCriterion}<float>
 *   and CriterionUtils (flashlight/lib/sequence/criterion/, un-vendored; call
 *   sites in /root/reference: recipes/slimIPL/src/Train.cpp:406-410, :1675,
 *   :838, :1375; recipes/joint_training_vox_populi/cpc/Train.cpp:813),
 * which is what fl::pkg::speech::{ASGLoss,CTCLoss} bind.
 * Layouts: emissions [B][T][N] (ArrayFire dims (N,T,B)); targets [B][L] int32
 * padded with negative values; transitions [N][N] indexed [to][from]; CTC
 * blank = N-1; paths [B][T] int32.
 */
#ifndef W2L_HIP_H_
#define W2L_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* w2l_stream_t; /* hipStream_t */

typedef enum {
  W2L_OK = 0,
  W2L_EINVAL = 1,       /* bad shape / null pointer (Flashlight throws std::invalid_argument) */
  W2L_EHIP = 2,         /* a HIP runtime call failed; see w2l_last_hip_error() */
  W2L_EUNSUPPORTED = 3  /* shape outside what this build implements */
} w2l_status;

/* fl::lib::seq::CriterionScaleMode (selected by --onorm / --sqnorm,
 * recipes/slimIPL/src/Train.cpp:389; recipes/conv_glu/librispeech/train.cfg:20-21) */
typedef enum {
  W2L_SCALE_NONE = 0,
  W2L_SCALE_INPUT_SZ = 1,
  W2L_SCALE_INPUT_SZ_SQRT = 2,
  W2L_SCALE_TARGET_SZ = 3,
  W2L_SCALE_TARGET_SZ_SQRT = 4
} w2l_scale_mode;

const char* w2l_version(void);
int w2l_last_hip_error(void); /* hipError_t of the last failing call on this thread */

/* ------------------------------------------------------------------------
 * 1. Sequence criteria
 * ---------------------------------------------------------------------- */

/* CriterionUtils::batchTargetSize: targetSize[b] = #leading non-negative labels
 * of target[b][0..L), clamped to maxSize (== T for ASG). */
int w2l_batch_target_size(int B, int L, int maxSize, const int* target, int* targetSize,
                          w2l_stream_t stream);
/* CTC flavour: with R = adjacent repeats, L <- min(L + R, T) - R */
int w2l_batch_ctc_target_size(int B, int L, int T, const int* target, int* targetSize,
                              w2l_stream_t stream);

/* FullConnectionCriterion<float>: log-partition over all N^T paths. */
size_t w2l_fcc_workspace_size(int B, int T, int N);
int w2l_fcc_forward(int B, int T, int N, int scaleMode, const float* input,
                    const int* targetSize, const float* trans, float* loss,
                    void* workspace, w2l_stream_t stream);
/* inputGrad [B][T][N] and transGrad [N][N] are OVERWRITTEN (transGrad summed over b). */
int w2l_fcc_backward(int B, int T, int N, const float* trans, const float* grad,
                     float* inputGrad, float* transGrad, void* workspace,
                     w2l_stream_t stream);

/* ForceAlignmentCriterion<float>: log-sum over monotone alignments of target. */
size_t w2l_fac_workspace_size(int B, int T, int N, int L);
int w2l_fac_forward(int B, int T, int N, int L, int scaleMode, const float* input,
                    const int* target, const int* targetSize, const float* trans,
                    float* loss, void* workspace, w2l_stream_t stream);
int w2l_fac_backward(int B, int T, int N, int L, const int* target, const int* targetSize,
                     const float* grad, float* inputGrad, float* transGrad,
                     void* workspace, w2l_stream_t stream);
/* forced alignment; bestPaths [B][T] holds target labels */
int w2l_fac_viterbi(int B, int T, int N, int L, const float* input, const int* target,
                    const int* targetSize, const float* trans, int* bestPaths,
                    void* workspace, w2l_stream_t stream);
/* AutoSegmentationCriterion in ONE call: loss[b] = FullConnectionCriterion - ForceAlignmentCriterion, the composition
 * fl::pkg::speech::ASGLoss::forward makes of the two criteria above (constructed at recipes/slimIPL/src/Train.cpp:408-410, run at
 * :1675, backward at :1720; math SURVEY App. B.1 / B.2).  target is the [B][L] batch as the Trainer hands it over (padded with
 * negative labels; the target sizes are counted on the device), trans the shared [N][N] transition parameter.  The two criteria
 * run side by side on `stream` and a library-owned side stream of the device; for the letter-sized label sets (N <= 31, L <= 320)
 * the launches that exist only to glue two calls together (target sizes, the loss / gradient differences) ride on their
 * neighbours (csrc/criterion_asg_fused.hpp).  backward OVERWRITES inputGrad [B][T][N] and transGrad [N][N] with the gradients of
 * sum_b grad[b] loss[b]; it needs the workspace of the forward call of the same batch.  Results equal the composed calls bit for bit. */
size_t w2l_asg_workspace_size(int B, int T, int N, int L);
int w2l_asg_forward(int B, int T, int N, int L, int scaleMode, const float* input, const int* target, const float* trans,
                    float* loss, void* workspace, w2l_stream_t stream);
int w2l_asg_backward(int B, int T, int N, int L, const int* target, const float* trans, const float* grad, float* inputGrad,
                     float* transGrad, void* workspace, w2l_stream_t stream);
/* Range check (diagnostics; no counterpart in the reference, whose log-domain recursion -- SURVEY App. B.1 / B.2, call sites
 * recipes/slimIPL/src/Train.cpp:408-410, :1675 -- is what the flagged utterances are recomputed with).  The fp32 / fp64
 * scaled-domain scans behind w2l_fcc_forward / w2l_fac_forward check per utterance that their inputs stay inside what they hold
 * exactly and hand the rest to log-domain kernels inside the same call: results are exact either way.  flags[b] (device, [B]) = 1
 * when utterance b of the LAST forward call on `workspace` (same B, T, N[, L]) took the log-domain path, else 0. */
int w2l_fcc_range_flags(int B, int T, int N, const void* workspace, int* flags, w2l_stream_t stream);
int w2l_fac_range_flags(int B, int T, int N, int L, const void* workspace, int* flags, w2l_stream_t stream);

/* LinSegCriterion (recipes/slimIPL/src/Train.cpp:589-617, --linseg): ASG on the target stretched linearly over the
 * T frames.  Flashlight getLinearTarget [UNVENDORED]: linTarget[b][t] = target[b][t * L_b / T], L_b = leading
 * non-negative entries of target[b][0..L); a row with L_b == 0 or L_b > T is filled with -1. */
int w2l_linear_target(int B, int L, int T, const int* target, int* linTarget /*[B][T]*/,
                      w2l_stream_t stream);
/* ForceAlignmentCriterion<float> on a length-T target (one alignment: label path[b][t] at frame t):
 * loss[b] = s_b * (sum_t input[b][t][y_t] + sum_{t>=1} trans[y_t][y_{t-1}]), target size = T in s_b.
 * A row of -1 gives loss 0 and zero gradients (as w2l_fac_* do for an empty target). */
int w2l_fac_fullpath_forward(int B, int T, int N, int scaleMode, const float* input,
                             const int* path, const float* trans, float* loss,
                             w2l_stream_t stream);
/* inputGrad [B][T][N] and transGrad [N][N] are OVERWRITTEN (transGrad summed over b; bit-reproducible
 * for N <= 64, float atomics above). */
int w2l_fac_fullpath_backward(int B, int T, int N, int scaleMode, const int* path,
                              const float* grad, float* inputGrad, float* transGrad,
                              w2l_stream_t stream);

/* ViterbiPath<float>: max-product path, first-max tie break, BIT-EXACT with the
 * CPU recursion (fp32 adds in the order (delta[j] + trans[i][j]) + x[t][i]). */
size_t w2l_viterbi_workspace_size(int B, int T, int N);
int w2l_viterbi_compute(int B, int T, int N, const float* input, const float* trans,
                        int* path, void* workspace, w2l_stream_t stream);

/* ConnectionistTemporalClassificationCriterion<float>; input are raw emissions
 * (log-softmax is applied inside), blank = N-1. */
size_t w2l_ctc_workspace_size(int B, int T, int N, int L);
int w2l_ctc_forward(int B, int T, int N, int L, int scaleMode, const float* input,
                    const int* target, const int* targetSize, float* loss,
                    void* workspace, w2l_stream_t stream);
/* needs `input` again (the softmax is recomputed instead of stored: 12*B*T*N
 * bytes of HBM traffic per fwd+bwd instead of 16). inputGrad OVERWRITTEN. */
int w2l_ctc_backward(int B, int T, int N, int L, const float* input, const int* target,
                     const int* targetSize, const float* grad, float* inputGrad,
                     void* workspace, w2l_stream_t stream);
/* CTCLoss::viterbiPath: per-frame argmax, first max wins */
int w2l_ctc_viterbi(int B, int T, int N, const float* input, int* path, w2l_stream_t stream);

/* ------------------------------------------------------------------------
 * 2. Network operators (fp32).  Activations are FRAME-MAJOR: a tensor the
 * reference holds as ArrayFire dims (T, H, C, B) lives here as x[B][T][H][C]
 * (C fastest), i.e. a row-major [M = B*T*H][C] matrix; Linear layers see
 * [M = B*T][H*C].  Replaces fl::conv2d / fl::linear / fl::LayerNorm /
 * fl::GatedLinearUnit / fl::Dropout autograd functions (un-vendored Flashlight;
 * grammar in recipes/joint_training_vox_populi/cpc/SequentialBuilder.cpp:92-626).
 * ---------------------------------------------------------------------- */

/* generic C[M][N] = op(A) op(B) (+bias[n]) (relu). a_kcontig: A stored [M][K];
 * else [K][M]. b_kcontig: B stored [N][K]; else [K][N]. splitk > 1: atomics. */
int w2l_gemm_f32(int M, int N, int K, const float* A, int lda, int a_kcontig, const float* B,
                 int ldb, int b_kcontig, float* C, int ldc, const float* bias, int relu,
                 int splitk, w2l_stream_t stream);

/* fl::Linear: y[M][out] = x[M][in] . w[in][out] + bias (w is Flashlight's (out,in)
 * column-major weight, byte-identical); optional fused ReLU. */
int w2l_linear_forward(int M, int in, int out, const float* x, const float* w,
                       const float* bias, float* y, int relu, w2l_stream_t stream);
int w2l_linear_backward_data(int M, int in, int out, const float* dy, const float* w, float* dx,
                             int accumulate, const float* maskSrc, float maskScale,
                             w2l_stream_t stream);
int w2l_linear_backward_weight(int M, int in, int out, const float* x, const float* dy, float* dw,
                               w2l_stream_t stream);
/* both parameter gradients of fl::Linear in one call: dw[in][out] = x^T dy and db[out] = sum_m dy[m][.] (db may be NULL: the
 * call above).  Where the product runs on the 160-wide LDS-DMA kernel the column sums ride on it -- the first tile row adds
 * up the dy fragments it multiplies, no second pass over dy --, otherwise w2l_colsum follows.  Both are deterministic. */
int w2l_linear_backward_weight_bias(int M, int in, int out, const float* x, const float* dy, float* dw, float* db,
                                    w2l_stream_t stream);
/* y = dropout(relu?(x w + b)): fl::Dropout behind a Linear(+ReLU) folded into the GEMM epilogue; bit-identical to
 * w2l_linear_forward followed by w2l_dropout_inplace(y, M*out, p, seed, rngStream) */
/* y = dropout(relu?(x w + b)) + add: the residual join behind a Linear in the same epilogue (fl::TDSBlock: r2 = dropout(lin2) + y1);
 * bit-identical to w2l_linear_forward_dropout followed by an elementwise add of `add` ([M][out]); p = 0: no dropout */
int w2l_linear_forward_dropout_add(int M, int in, int out, const float* x, const float* w, const float* bias, const float* add,
                                   float* y, int relu, double p, uint32_t seed, uint32_t rngStream, w2l_stream_t stream);
int w2l_linear_forward_dropout(int M, int in, int out, const float* x, const float* w, const float* bias,
                               float* y, int relu, double p, uint32_t seed, uint32_t rngStream,
                               w2l_stream_t stream);
/* dx = add + dy w^T (add laid out like dx): the residual join of a backward pass without copying add into dx first */
int w2l_linear_backward_data_add(int M, int in, int out, const float* dy, const float* w, const float* add,
                                 float* dx, w2l_stream_t stream);
/* Mixed precision (BASELINE config 3; fl's --fl_amp_use_mixed_precision, recipes/joint_training_vox_populi/cpc/Train.cpp:1184
 * keeps the criterion input in f32): mode 1 = every fl::Linear GEMM multiplies in bf16 (v_mfma_f32_32x32x16_bf16, operands
 * rounded to nearest even on the way into LDS) and accumulates in fp32; operands, results, master weights and the
 * criteria stay fp32.  Process-wide, returns the previous mode. */
int w2l_set_matmul_precision(int mode);
/* Mixed precision with bf16 OPERAND STORAGE (round 3): the activations, the output gradients and per-step copies of the fp32
 * master weights are kept as bf16 images in HBM and multiplied on v_mfma_f32_32x32x16_bf16 with fp32 accumulation; results,
 * bias, every non-GEMM operator and the criterion stay fp32 (cpc/Train.cpp:1184).
 * w2l_bf16_convert: x [rows][cols] fp32 (leading dimension ldx) -> rowMajor [rows][ldRows] and / or transposed [cols][ldTrans]
 *   bf16 (round to nearest even), zero-padded to the leading dimension; either may be NULL; ldRows >= cols, ldTrans >= rows,
 *   both multiples of 16 elements (use the reduction length rounded up to 64); 16-byte aligned outputs.
 * w2l_gemm_bf16: C[M][N] (fp32, ldc) = A[M][K] . B[N][K]^T (+ bias[n]) (ReLU) (dropout) (mask) (+ addend | + C); A and B are
 *   k-contiguous bf16 images whose rows are ZERO from column K up to K rounded to 64 (lda, ldb >= that, even). */
typedef struct {
  const float* mask;  /* v = mask[m][n] > 0 ? v * maskScale : 0  (layout of C) */
  float maskScale;
  const float* addend; /* v += addend[m][n] (layout of C) */
  int accumulate;      /* v += C[m][n] (ignored when addend is set) */
  double dropP;        /* > 0: dropout with the library's stateless hash of (m * ldc + n, dropSeed, dropStream) */
  uint32_t dropSeed, dropStream;
} w2l_gemm_epilogue;
int w2l_bf16_convert(const float* x, size_t rows, int cols, size_t ldx, uint16_t* rowMajor, size_t ldRows,
                     uint16_t* transposed, size_t ldTrans, w2l_stream_t stream);
/* the images of dropout(x): mask and scale of w2l_dropout_copy(p, seed, rngStream) over the dense [rows][ldx] matrix, applied on the
 * way (the masked copy is never written) -- a dropout layer's backward pass when the masked gradient is only a GEMM operand */
int w2l_bf16_convert_dropout(const float* x, size_t rows, int cols, size_t ldx, uint16_t* rowMajor, size_t ldRows,
                             uint16_t* transposed, size_t ldTrans, double p, uint32_t seed, uint32_t rngStream, w2l_stream_t stream);
/* n (1 .. 8) conversions in one launch (small matrices are launch-bound one at a time); each entry = the arguments of w2l_bf16_convert */
typedef struct {
  const float* x;
  size_t rows;
  int cols;
  size_t ldx;
  uint16_t* rowMajor;
  size_t ldRows;
  uint16_t* transposed;
  size_t ldTrans;
} w2l_bf16_convert_desc;
int w2l_bf16_convert_multi(int n, const w2l_bf16_convert_desc* descs, w2l_stream_t stream);
int w2l_gemm_bf16(int M, int N, int K, const uint16_t* A, int lda, const uint16_t* B, int ldb, float* C, int ldc,
                  const float* bias, int relu, const w2l_gemm_epilogue* epilogue, w2l_stream_t stream);
/* `groups` (1 .. 4) products of ONE shape in one launch, C_g [M][N] = A_g . B_g^T (+ bias_g): the tiles of all problems share the
 * persistent grid (a Transformer block's four C x C weight gradients at M = 3008 frames are 64 tiles each -- a quarter of the chip
 * one at a time).  A, B, C, bias are HOST arrays of device pointers (bias, or single entries of it, may be NULL). */
/* the same with k-MAJOR operands read in place: aKMajor -> A stored [K][lda] (lda >= M), bKMajor -> B stored [K][ldb] (ldb >= N);
 * lda / ldb multiples of 8, bases 16-byte aligned; rows k >= K are not read.  x^T dy from the row-major images of x and dy,
 * x w from the row-major image of w: no transposed bf16 image of either operand */
int w2l_gemm_bf16_ex(int M, int N, int K, const uint16_t* A, int lda, int aKMajor, const uint16_t* B, int ldb, int bKMajor, float* C,
                     int ldc, const float* bias, int relu, const w2l_gemm_epilogue* e, w2l_stream_t stream);
int w2l_gemm_bf16_grouped(int groups, int M, int N, int K, const uint16_t* const* A, int lda, const uint16_t* const* B, int ldb,
                          float* const* C, int ldc, const float* const* bias, w2l_stream_t stream);
int w2l_colsum(const float* x, float* out, size_t M, int N, w2l_stream_t stream); /* bias grads */

/* fl::Conv2D kw x 1 over time (arch tokens C / C2 / TDS). x [B][T][H][Cin],
 * w [kw][Cin][Cout], y [B][To][H][Cout]; cross-correlation, zero padding. */
typedef struct {
  int B, T, H, Cin, Cout, kw, stride, padl, padr;
} w2l_conv_desc;
int w2l_conv_out_len(int T, int kw, int stride, int padl, int padr);
int w2l_conv_same_pad(int T, int kw, int stride); /* PaddingMode::SAME (pad = -1) */
int w2l_conv_forward(const w2l_conv_desc* d, const float* x, const float* w, const float* bias,
                     float* y, int relu, w2l_stream_t stream);
int w2l_conv_backward_data(const w2l_conv_desc* d, const float* dy, const float* w, float* dx,
                           int accumulate, w2l_stream_t stream);
/* dx = add + backward-data(dy, w): fused residual join (add has the layout of dx) */
int w2l_conv_backward_data_add(const w2l_conv_desc* d, const float* dy, const float* w,
                               const float* add, float* dx, w2l_stream_t stream);
int w2l_conv_backward_filter(const w2l_conv_desc* d, const float* x, const float* dy, float* dw,
                             float* dbias, w2l_stream_t stream);
/* The kw x 1 convolutions over the mel rows of the TDS recipes in the mixed-precision mode (bf16 operand storage, see
 * w2l_gemm_bf16) -- fl::TDSBlock's time convolution (C -> C channels, stride 1) and the sub-sampling `C2 cin cout kw 1 s 1`
 * lines between the blocks (cin != cout <= 32 channels, stride 1 or 2, any padding): x / dy and the weights rounded to bf16,
 * fp32 accumulation on v_mfma_f32_32x32x16_bf16, fp32 bias / ReLU / addend / results (conv_tds_bf16.hip).  Layouts as
 * w2l_conv_*: x [B][T][H][Cin], w [kw][Cin][Cout], y / dy [B][To][H][Cout].
 * w2l_tds_conv_bf16_image_elems: bf16 elements of ONE weight image buffer of the geometry (the forward image, or the images
 *   of all phases of a strided backward-data pass, whichever is larger); 0 = no bf16 kernel for it (use w2l_conv_*).
 * w2l_tds_conv_bf16_prepare: once per step, the forward and the backward-data images of the fp32 weights.
 * backward_filter: dw only (H % 16 == 0); backward_filter_bias: dw and the bias gradient (sums of the ROUNDED dy) in one launch. */
size_t w2l_tds_conv_bf16_image_elems(const w2l_conv_desc* d);
int w2l_tds_conv_bf16_prepare(const w2l_conv_desc* d, const float* w, uint16_t* imgForward, uint16_t* imgBackward, w2l_stream_t stream);
int w2l_tds_conv_bf16_forward(const w2l_conv_desc* d, const float* x, const uint16_t* imgForward, const float* bias, float* y,
                              int relu, w2l_stream_t stream);
int w2l_tds_conv_bf16_backward_data(const w2l_conv_desc* d, const float* dy, const uint16_t* imgBackward, const float* add,
                                    float* dx, w2l_stream_t stream);
int w2l_tds_conv_bf16_backward_filter(const w2l_conv_desc* d, const float* x, const float* dy, float* dw, w2l_stream_t stream);
/* the same launch also produces dbias [Cout] = column sums of the bf16-rounded dy (summed from the slabs the kernel stages anyway) */
int w2l_tds_conv_bf16_backward_filter_bias(const w2l_conv_desc* d, const float* x, const float* dy, float* dw, float* dbias,
                                           w2l_stream_t stream);

/* bf16 operand images written by the kernel that PRODUCES a matrix [rows][cols] instead of a w2l_bf16_convert pass over an fp32
 * copy (mixed-precision mode): `rowMajor` [rows][ldRows] and / or `transposed` [cols][ldTrans], the layouts of w2l_bf16_convert;
 * either pointer may be NULL.  Only elements of the matrix are written (a kernel may write zeros into the padding): the zero
 * padding w2l_bf16_convert leaves beyond cols / rows must already be there -- images the caller zero-filled once. */
typedef struct {
  uint16_t* rowMajor;
  size_t ldRows;
  uint16_t* transposed;
  size_t ldTrans;
} w2l_bf16_image_sink;
/* w2l_gemm_bf16 whose RESULT leaves as the bf16 images the next products read (`images`: row-major [M][ldRows] and / or transposed
 * [N][ldTrans], what w2l_bf16_convert would make of C bit for bit; only elements of the matrix are written) instead of, or beside,
 * the fp32 C (C may be NULL when images are given); maskImage: the mask operand as a bf16 row-major image [M][ldMask] (> 0 test,
 * replaces epilogue->mask).  An activation that is only ever a GEMM operand and a ReLU / dropout mask then lives only as bf16,
 * as under the reference's AMP (recipes/slimIPL/src/Train.cpp:209-216).  N % 4 == 0; ldc (>= N) still names the flat index
 * m * ldc + n of the dropout hash.  W2L_EUNSUPPORTED where the wide epilogue cannot run. */
int w2l_gemm_bf16_images(int M, int N, int K, const uint16_t* A, int lda, const uint16_t* B, int ldb, float* C, int ldc,
                         const float* bias, int relu, const w2l_gemm_epilogue* epilogue, const w2l_bf16_image_sink* images,
                         const uint16_t* maskImage, size_t ldMask, float maskScale, w2l_stream_t stream);
/* r = dropout(a) + x ; y = LayerNorm(r) over `groups` contiguous chunks of `inner`
 * elements with scalar affine gammaBeta[2] (fl::LayerNorm axes {0,1,2}: groups = B).
 * a is updated in place to its dropped value; r, meanRstd[2*groups] are kept for
 * backward; stats / sums are double[w2l_layernorm_scratch_doubles(groups, inner)] scratch
 * (per-block partial sums, added in a fixed order: no atomics, run-to-run deterministic).  Dropout mask = stateless hash of
 * (flat index, seed, rngStream), reproduced bit-exactly by the oracle. */
size_t w2l_layernorm_scratch_doubles(int groups, size_t inner);
int w2l_residual_layernorm_forward(int groups, size_t inner, float* a, const float* x, float* r,
                                   float* y, const float* gammaBeta, float eps, double p,
                                   uint32_t seed, uint32_t rngStream, double* stats,
                                   float* meanRstd, w2l_stream_t stream);
int w2l_layernorm_backward(int groups, size_t inner, const float* r, const float* dy,
                           const float* gammaBeta, const float* meanRstd, float* dr,
                           float* dGammaBeta, const float* maskSrc, float* dmask, float maskScale,
                           double* sums, w2l_stream_t stream);
/* The per-frame LayerNorm of the mixed-precision mode with the images of its result written in the same pass
 * (layernorm_images.hip): as the two calls above for rows of inner <= 2304 floats (inner % 4 == 0; W2L_EUNSUPPORTED otherwise: run
 * the plain call and w2l_bf16_convert), plus yImages = images of y, resp. drImages = images of dr -- of dropout(dr) when
 * imageDropP > 0 (the mask of w2l_dropout_copy with imageDropSeed / imageDropStream over the flat index; dr itself stays
 * unmasked: what w2l_bf16_convert_dropout produced).  ldTrans must hold the rows rounded up to 16 (runs of 16 rows are written).
 * Replaces: fl::LayerNorm + the AMP casts of the next fl::Linear (recipes/slimIPL/src/Train.cpp:209-216). */
int w2l_residual_layernorm_forward_images(int groups, size_t inner, float* a, const float* x, float* r, float* y,
                                          const float* gammaBeta, float eps, double p, uint32_t seed, uint32_t rngStream,
                                          float* meanRstd, const w2l_bf16_image_sink* yImages, w2l_stream_t stream);
int w2l_layernorm_backward_images(int groups, size_t inner, const float* r, const float* dy, const float* gammaBeta,
                                  const float* meanRstd, float* dr, float* dGammaBeta, const float* maskSrc, float* dmask,
                                  float maskScale, double* sums, const w2l_bf16_image_sink* drImages, double imageDropP,
                                  uint32_t imageDropSeed, uint32_t imageDropStream, w2l_stream_t stream);
int w2l_dropout_inplace(float* x, size_t n, double p, uint32_t seed, uint32_t rngStream,
                        w2l_stream_t stream);
/* y = dropout(x) out of place, same mask as w2l_dropout_inplace (16-byte aligned x, y) */
int w2l_dropout_copy(float* y, const float* x, size_t n, double p, uint32_t seed, uint32_t rngStream,
                     w2l_stream_t stream);
int w2l_mask_backward(const float* dy, const float* src, float* dx, size_t n, float scale,
                      w2l_stream_t stream);

/* fl::Conv2D with a kh x kw kernel, kh > 1 (the "C2 cin cout kw kh sx 1 px -1" lines of
 * recipes/sota/2019/am_arch/am_tds_ctc_librivox.arch:3-26; builder recipes/joint_training_vox_populi/cpc/SequentialBuilder.cpp:285-300)
 * runs as a kw x 1 convolution over kh*C channels: xe[r][h][dh*C + c] = x[r][h + dh - padh][c], zero outside the mel
 * axis; rows = utterances x frames.  _backward is the adjoint (sum over dh). */
int w2l_hexpand_forward(const float* x, float* xe, size_t rows, int H, int C, int kh, int padh, w2l_stream_t stream);
int w2l_hexpand_backward(const float* dxe, float* dx, size_t rows, int H, int C, int kh, int padh, w2l_stream_t stream);
/* fl::Transformer's attention core (arch token `TR`, recipes/sota/2019/am_arch/am_transformer_ctc.arch:15-38; block:
 * recipes/joint_training_vox_populi/cpc/TransformerCPC.cpp:117-151; fl::multiheadAttention is [UNVENDORED] Flashlight).
 * Strided batched GEMM over G1 x G2 problems, C = (C +) A B: element (m, k) of A of problem (g1, g2) at
 * A[g1*a1 + g2*a2 + m*sam + k*sak], (k, n) of B at B[g1*b1 + g2*b2 + k*sbk + n*sbn], (m, n) of C at
 * C[g1*c1 + g2*c2 + m*ldc + n]; strides in floats.  With frame-major [B][T][heads*d] q / k / v this is QK^T, PV and their
 * gradients per (utterance, head) without a transpose. */
typedef struct {
  int M, N, K, G1, G2;
  long long sam, sak, a1, a2;
  long long sbk, sbn, b1, b2;
  long long ldc, c1, c2;
  int accumulate;
  /* banded A operand (w2l_bgemm_bf16 only; 0 = dense): the skewed score gradient dR of the relative-position products is non-zero,
   * in row (b, i, h), only at the bandT table rows w = j - i + bandOff, j in [0, bandT).  1: rows = (b, i, h) flattened (bandH heads),
   * k = w; 2: rows = w, k = (i, h) flattened.  K tiles outside the band are skipped (they multiply exact zeros). */
  int bandMode, bandT, bandH, bandOff;
} w2l_bgemm_desc;
int w2l_bgemm_f32(const w2l_bgemm_desc* d, const float* A, const float* B, float* C, w2l_stream_t stream);
/* the same product with bf16 multiplies: fp32 operands rounded to nearest-even bf16 on the way into LDS, v_mfma_f32_32x32x16_bf16,
 * fp32 accumulation and result (BASELINE config 5, "bf16 MFMA attention") */
int w2l_bgemm_bf16(const w2l_bgemm_desc* d, const float* A, const float* B, float* C, w2l_stream_t stream);
/* S[b][h][i][j] (in: q_i . k_j) -> P = softmax_j(scale * (S + R[(b*T + i)*H + h][j - i + n0 - rlo])) in place; R (may be
 * NULL: no position term) holds q_i . E[rlo + w] for w < W, row stride ldr; entries outside [0, W) count as 0
 * (relativePositionEmbeddingRotate pads with zeros).  keyLen (may be NULL): keys j >= keyLen[b] are padding and get
 * probability 0 (the log(padMask) term of TransformerCPC.cpp:138-144). */
int w2l_attn_softmax_forward(float* S, const float* R, const int* keyLen, int B, int H, int T, int ldr, int rlo, int W, int n0,
                             float scale, w2l_stream_t stream);
/* The attention core of one Transformer block's FORWARD pass in one launch (mixed-precision mode; attention_fused.hip):
 *   P = softmax_j(scale * (q_i . k_j + q_i . posTable[j - i + n0]) + log padMask),  Pd = dropout(P),  ctx_i = sum_j Pd[i][j] v_j
 * q, k, v: [B][T][ld] fp32, head h in columns h*d .. (h+1)*d (ld = C, or 3 C for an interleaved q|k|v buffer); posTable: the
 * [2 csz - 1][d] table in the library's internal layout or NULL; rows rlo .. rlo + W are the ones T frames reach (entries outside
 * count as 0); keyLen as in w2l_attn_softmax_forward.  Writes P [B][H][T][T] (the backward pass reads it), Pd (same shape, only when
 * dropP > 0: the same keep pattern as w2l_dropout_copy over P) and ctx [B][T][ldc].  Operands are rounded to bf16 where
 * w2l_bgemm_bf16 rounds them; the result equals the unfused sequence to fp32 summation order.
 * W2L_EUNSUPPORTED when the geometry has no fused kernel (d in {32, 256}, T <= 192, 16-byte aligned rows): run the unfused one.
 * Replaces: TransformerCPC.cpp:117-151 (selfAttention) for the forward pass. */
typedef struct {
  int B, H, T, d;
  int ld, ldc;
  int W, n0, rlo;
  float scale;
  double dropP;
  uint32_t dropSeed, dropStream;
} w2l_attn_fused_desc;
int w2l_attn_fused_forward(const w2l_attn_fused_desc* d, const float* q, const float* k, const float* v, const float* posTable,
                           const int* keyLen, float* P, float* Pd, float* ctx, w2l_stream_t stream);
/* The attention core of the same block's BACKWARD pass (mixed-precision mode; attention_fused_bwd.hip): from dctx [B][T][ldc], the
 * forward's P and its operands
 *   dP = dctx v^T with the forward's dropout mask (recomputed from dropP / dropSeed / dropStream: Pd is not read),
 *   dS = scale * P o (dP - rowsum(P o dP)),   dq_i = sum_j dS[i][j] (k_j + posTable[j - i + n0]),   dk_j = sum_i dS[i][j] q_i,
 *   dv_j = sum_i Pd[i][j] dctx_i,   dPosTable[w] = sum_(b, h, i) dS[i][i + w - n0] q_i
 * in four launches (E^T image, query side, key side, table-gradient reduce) instead of eleven.  dq, dk, dv: [B][T][ld]
 * (overwritten); dPosTable: the whole [2 n0 + 1][d] table gradient (overwritten, zero outside the rows T frames reach; ignored
 * when posTable is NULL).  Operands are rounded to bf16 where the unfused w2l_bgemm_bf16 sequence rounds them (dctx, v, Pd, dS, k,
 * q, posTable); the result equals that sequence to fp32 summation order.  workspace: w2l_attn_fused_backward_workspace(d,
 * posTable != NULL) bytes, 256-byte aligned; it holds the bf16 images dS^T / Pd^T [B H][32 NT][32 NT] (row = key, column = query,
 * NT = ceil(T / 32) rounded up to 2, 4 or 6) first, which tests read.  The key-padding mask needs no argument: P is zero there.
 * W2L_EUNSUPPORTED (workspace size 0) for a geometry without a fused kernel: the same set as the forward call.
 * Replaces: the gradient of TransformerCPC.cpp:117-151 (selfAttention). */
/* The same two calls writing the bf16 images the NEXT product reads (w2l_bf16_image_sink) instead of, or beside, the fp32 result
 * [B T][ldc or ld]: no fp32 copy, no w2l_bf16_convert pass.  The fp32 pointer may be NULL when images are given. */
int w2l_attn_fused_forward_images(const w2l_attn_fused_desc* d, const float* q, const float* k, const float* v, const float* posTable,
                                  const int* keyLen, float* P, float* Pd, float* ctx, const w2l_bf16_image_sink* ctxImages,
                                  w2l_stream_t stream);
int w2l_attn_fused_backward_images(const w2l_attn_fused_desc* d, const float* q, const float* k, const float* v, const float* posTable,
                                   const float* P, const float* dctx, float* dq, float* dk, float* dv,
                                   const w2l_bf16_image_sink* dqImages, const w2l_bf16_image_sink* dkImages,
                                   const w2l_bf16_image_sink* dvImages, float* dPosTable, void* workspace, size_t workspaceBytes,
                                   w2l_stream_t stream);
size_t w2l_attn_fused_backward_workspace(const w2l_attn_fused_desc* d, int withPosTable);
int w2l_attn_fused_backward(const w2l_attn_fused_desc* d, const float* q, const float* k, const float* v, const float* posTable,
                            const float* P, const float* dctx, float* dq, float* dk, float* dv, float* dPosTable, void* workspace,
                            size_t workspaceBytes, w2l_stream_t stream);
/* valid keys per utterance from the batch's input sizes (any unit), as forwardSequentialModuleWithPadMask builds the mask
 * (cpc/SequentialBuilder.cpp:58-81): n_b = ceil(size_b * Tin / max size) valid input frames, resized to Tk (nearest) */
int w2l_attn_key_lengths(const float* inputSizes, int B, int Tin, int Tk, int* keyLen, w2l_stream_t stream);
/* the same for a batch padded BEYOND its longest utterance: *fullSize (device scalar, same unit; NULL = the call above) is the
 * size the Tin input frames correspond to and replaces max size as the denominator */
int w2l_attn_key_lengths_full(const float* inputSizes, const float* fullSize, int B, int Tin, int Tk, int* keyLen,
                              w2l_stream_t stream);
/* dS (in: dL/dP) -> dL/dS (pre-scale scores) in place; dR (may be NULL) receives the skewed copy, zeros elsewhere */
int w2l_attn_softmax_backward(const float* P, float* dS, float* dR, int B, int H, int T, int ldr, int rlo, int W, int n0,
                              float scale, w2l_stream_t stream);
/* fl::Pool2D(w, 1, stride, 1, 0, 0, MAX) over time on frame-major x[B][T][F] -> y[B][(T-w)/stride+1][F] (arch token `M`,
 * SequentialBuilder.cpp:398-414); backward routes each dy to the first maximum of its window */
int w2l_pool_time_forward(const float* x, float* y, int B, int T, int F, int w, int stride, w2l_stream_t stream);
int w2l_pool_time_backward(const float* x, const float* dy, float* dx, int B, int T, int F, int w, int stride,
                           w2l_stream_t stream);
int w2l_axpy(float* y, const float* x, size_t n, float alpha, w2l_stream_t stream);
int w2l_fill(float* y, size_t n, float v, w2l_stream_t stream);
int w2l_transpose(const float* in, float* out, int G, int R, int C, w2l_stream_t stream);
/* MFSC / log-mel front end (fl::lib::audio::Mfsc as configured by LogMelFeature.cpp:78-95, [UNVENDORED]): the linear
 * per-frame part (pre-emphasis, window, DFT) is one w2l_gemm_f32 on overlapping rows of the audio (lda = frame stride);
 * spectrum: spec[m][k] = |re + i im| (usePower: squared), zero-padded to ldOut columns; then the mel GEMM; then
 * out[b][f][t] = log(max(mel[b][t][f], floor)) in the network's input layout. */
int w2l_mfsc_spectrum(const float* reim /*[M][2*nbins]*/, float* spec /*[M][ldOut]*/, size_t M, int nbins,
                      int ldOut, int usePower, w2l_stream_t stream);
int w2l_mfsc_log_transpose(const float* mel /*[B][Tp][F]*/, float* out /*[B][F][T]*/, int B, int Tp, int T,
                           int F, float floorv, w2l_stream_t stream);
int w2l_glu_forward(const float* x, float* y, size_t M, int half, w2l_stream_t stream);
int w2l_glu_backward(const float* x, const float* dy, float* dx, size_t M, int half,
                     w2l_stream_t stream);

/* fl::WeightNorm on an internal [K][N] weight (per-column norm); norm[N], dot[N] scratch */
int w2l_weightnorm_forward(const float* v, const float* g, float* w, float* norm, int K, int N,
                           w2l_stream_t stream);
int w2l_weightnorm_backward(const float* v, const float* g, const float* norm, const float* dw,
                            float* dv, float* dg, float* dot, int K, int N, w2l_stream_t stream);
/* fl::SpecAugment (SAUG token): frequency / time masking with zeros, in place on x[B][T][F] */
int w2l_specaugment_inplace(float* x, int B, int T, int F, int fMaskF, int nFMask, int tMaskT,
                            float tMaskP, int nTMask, uint32_t seed, w2l_stream_t stream);

/* fl::SGDOptimizer + fl::clipGradNorm over a flat parameter arena
 * (recipes/slimIPL/src/Train.cpp:1791-1804). */
int w2l_sumsq(const float* g, size_t n, double* out, int zeroFirst, w2l_stream_t stream);
int w2l_sgd_step(float* p, const float* g, float* v, size_t n, float lr, float momentum,
                 float gradScale, float maxGradNorm, const double* sumsq, w2l_stream_t stream);
/* Non-finite guard + device-side batch size of a data-parallel step (Train.cpp:1651-1660, :1743-1747).
 * acc[5] doubles: in  acc[0] = sum g^2 of the network gradients, acc[1] = of the criterion gradients;
 *                 out acc[2] = clip norm^2 (acc[0] + clampCrit*acc[1]), NaN when ANY gradient or the batch size is
 *                     non-finite, acc[3] = 1 / *batchDev (0 if batchDev is null), acc[4] += 1 per skipped update.
 * w2l_sgd_step_guarded(guard = acc + 2) leaves p and v untouched when guard[0] is non-finite -- with or without
 * clipping -- and scales the gradient by guard[1] instead of gradScale when guard[1] > 0. */
int w2l_grad_guard(double* acc, const float* batchDev, int clampCrit, w2l_stream_t stream);
int w2l_sgd_step_guarded(float* p, const float* g, float* v, size_t n, float lr, float momentum,
                         float gradScale, float maxGradNorm, const double* guard, w2l_stream_t stream);
/* fl::AdagradOptimizer::step (--netoptim=adagrad, recipes/sota/2019/librivox/train_am_transformer_ctc.cfg:25-26; un-vendored
 * Flashlight class): var += g'^2, p -= lr g' / (sqrt(var) + eps) with g' = g * gradScale * clip; guard as w2l_sgd_step_guarded
 * (NULL: no guard, no clip, gradScale as given) */
int w2l_adagrad_step_guarded(float* p, const float* g, float* var, size_t n, float lr, float eps, float gradScale,
                             float maxGradNorm, const double* guard, w2l_stream_t stream);
/* fl::AdadeltaOptimizer::step (--netoptim=adadelta, recipes/sota/2019/librispeech/train_am_transformer_ctc.cfg:23-26; un-vendored
 * Flashlight class): accGrad = rho accGrad + (1-rho) g'^2, delta = sqrt(accDelta + eps) / sqrt(accGrad + eps) g', p -= lr delta,
 * accDelta = rho accDelta + (1-rho) delta^2 */
int w2l_adadelta_step_guarded(float* p, const float* g, float* accGrad, float* accDelta, size_t n, float lr, float rho,
                              float eps, float gradScale, float maxGradNorm, const double* guard, w2l_stream_t stream);

/* ------------------------------------------------------------------------
 * 3. Trainer: arch file -> module graph -> one optimisation step.
 * Mirrors buildSequentialModule (recipes/joint_training_vox_populi/cpc/SequentialBuilder.h:23-26),
 * ASGLoss / CTCLoss construction (recipes/slimIPL/src/Train.cpp:406-410) and the hot loop
 * (Train.cpp:1454-1804).  The caller owns all device memory: query sizes, allocate, bind.
 * params / grads / momentum are flat arenas [network params | criterion params]; the
 * data-parallel all-reduce is ONE collective over `grads` between forward_backward and update.
 * ---------------------------------------------------------------------- */
const char* w2l_host_last_error(void);

/* ---- FLAC decoding (host side of the input pipeline).  Replaces: fl::pkg::speech::loadSound through libsndfile for
 * the .flac files the recipes' lists point at (data/librispeech/utils.py:36-46).  A from-specification decoder (RFC 9639): the native
 * container, every subframe type, Rice / Rice2 residuals, all stereo decorrelations, 4-32 bits; frame CRC-8 / CRC-16 and the
 * STREAMINFO MD5 of the decoded audio are verified.  `data` is the whole file in memory. */
int w2l_flac_info(const uint8_t* data, size_t bytes, int* sampleRate, int* channels, int* bitsPerSample, uint64_t* totalSamples);
/* out[sample][channel] interleaved int32, room for `capacity` inter-channel samples; *md5: 1 signature verified, -1 the file has none
 * (a mismatch returns W2L_EINVAL, as does any malformed or corrupted frame: w2l_flac_last_error() says which) */
int w2l_flac_decode(const uint8_t* data, size_t bytes, int32_t* out, uint64_t capacity, uint64_t* decoded, int* md5);
const char* w2l_flac_last_error(void);
void* w2l_trainer_create(const char* archText, int nFeat, int nLabel, const char* criterion,
                         int scaleMode, double transdiag);
void w2l_trainer_destroy(void* h);
const char* w2l_trainer_describe(void* h);
size_t w2l_trainer_param_floats(void* h);
size_t w2l_trainer_net_param_floats(void* h);
/* floats of the gradient arena to bind: parameters + a 4-float tail; tail[0] = this rank's batch size, written by
 * forward_backward, so that ONE all-reduce over [0, grad_floats) sums gradients and batch sizes
 * (recipes/slimIPL/src/Train.cpp:1743-1747 all-reduces the batch size separately) */
size_t w2l_trainer_grad_floats(void* h);
int w2l_trainer_num_params(void* h);
int w2l_trainer_param_info(void* h, int i, char* name, int nameCap, size_t* numel, size_t* offset);
int w2l_trainer_init_params(void* h, float* h_params, uint64_t seed);
int w2l_trainer_import_param(void* h, int i, const float* h_ref, float* h_params);
int w2l_trainer_export_param(void* h, int i, const float* h_arena, float* h_ref);
int w2l_trainer_plan(void* h, int B, int T, int L, size_t* arenaFloats, size_t* critWsBytes, int* Tout);
/* params / momentum: w2l_trainer_param_floats floats; grads: w2l_trainer_grad_floats floats (4-float tail) */
int w2l_trainer_bind(void* h, float* params, float* grads, float* momentum, float* arena, void* critWs);
/* x: [B][NFEAT][T] (the reference's (T,NFEAT,1,B) input); target [B][L] int32, -1 padded */
int w2l_trainer_forward(void* h, const float* x, int train, const float** emission, void* stream);
int w2l_trainer_forward_backward(void* h, const float* x, const int* target, float** lossDev,
                                 void* stream);
/* backward pass of the network alone from a caller-supplied gradient of the emissions [B][T'][N] (device): the fl::Module boundary
 * for a binder that keeps its own criterion.  After w2l_trainer_forward(train = 1) of the same step; leaves the network's
 * gradients in the gradient arena (unscaled).  Replaces: fl::Variable::backward through the module graph
 * (recipes/slimIPL/src/Train.cpp:1719-1721). */
int w2l_trainer_backward(void* handle, const float* dEmission, void* stream);
/* totalBatch > 0: scale the gradients by 1/totalBatch; totalBatch <= 0: by 1 / (the all-reduced batch size in the
 * gradient arena's tail).  A non-finite gradient (or batch size) skips the update on every rank -- with or without
 * clipping -- and counts it (w2l_trainer_skipped_updates). */
int w2l_trainer_update(void* h, float lr, float lrcrit, float momentum, float maxGradNorm,
                       float totalBatch, int clampCrit, void* stream);
int w2l_trainer_skipped_updates(void* h, uint64_t* count, void* stream);
/* optimizer of the network / criterion parameters for w2l_trainer_update: 0 = SGD with momentum (default), 1 = Adagrad
 * (eps 1e-8; the momentum arena holds the accumulated squared gradients), 2 = Adadelta (rho 0.9, eps 1e-8; accGrad in the
 * momentum arena, accDelta in a second arena of the same size: w2l_trainer_bind_state2).
 * Train.cpp:577-582 initOptimizer(--netoptim / --critoptim) */
int w2l_trainer_set_optimizer(void* h, int netKind, int critKind);
/* input sizes of the NEXT forward calls (device, [B] floats in any unit; NULL = every utterance fills the batch): the
 * Transformer blocks mask the padded keys as the reference's forwardSequentialModuleWithPadMask does.  Call after plan. */
int w2l_trainer_set_input_sizes(void* h, const float* inputSizesDev);
int w2l_trainer_bind_state2(void* h, float* state2);
int w2l_trainer_viterbi(void* h, const float* emission, int* path, void* stream);
int w2l_trainer_set_step(void* h, uint32_t step);
/* --fl_amp_use_mixed_precision restated for bf16 (BASELINE config 3): the network's fl::Linear GEMMs multiply in bf16
 * with fp32 accumulation (w2l_set_matmul_precision scoped to the network's calls); storage, master weights, convolutions,
 * LayerNorm, the criterion (recipes/joint_training_vox_populi/cpc/Train.cpp:1184) and the optimizer stay fp32 */
int w2l_trainer_set_mixed_precision(void* h, int on);
/* gradient norm seen by the last w2l_trainer_update (before the 1/totalBatch scale; taken on every update).  A
 * non-finite norm means the update was SKIPPED on every rank (the norm is taken on the all-reduced gradient):
 * the counterpart of the reference's NaN guards, recipes/slimIPL/src/Train.cpp:1651-1660, :1686-1698. */
int w2l_trainer_grad_norm(void* h, double* norm, void* stream);
/* --linseg=n (recipes/slimIPL/src/Train.cpp:589-617, :1866-1883): the first n updates of an ASG run use
 * LinSegCriterion on the ASG criterion's own transitions.  Call before w2l_trainer_plan. */
int w2l_trainer_set_linseg(void* h, uint32_t updates);
/* Data-parallel overlap (replaces fl::CoalescingReducer, recipes/slimIPL/src/Train.cpp:195, :1721-1735):
 * bucket k = [offsets[k], offsets[k+1]) of the flat gradient arena (ascending float offsets; the last
 * bucket runs to the end).  forward_backward records one event per bucket on its stream as soon as every
 * gradient at offset >= offsets[k] is final (backward walks the layers last to first);
 * wait_bucket makes `stream` (the collective's stream) wait for bucket k.  n = 0 removes the hooks. */
int w2l_trainer_set_grad_buckets(void* h, int n, const size_t* offsets);
int w2l_trainer_wait_bucket(void* h, int k, void* stream);
/* roofline instrumentation: bracket every MFMA GEMM launch with hipEvents on its stream */
int w2l_profile_enable(int on);
int w2l_profile_report(int* launches, double* totalMs, double* totalFlops); /* kind 0 */
/* kind: 0 = 128x128 MFMA GEMM, 1 = skinny implicit GEMM, 2 = TDS slab convolution (work = FLOPs),
 * 3 = FCC transition stream (work = algorithmic bytes), -1 = all */
int w2l_profile_report_kind(int kind, int* launches, double* totalMs, double* totalWork);
/* per-launch rows of one kind (bench.py's per-shape table of the dominant kernel): ms[i], work[i], dims[4 i ..] = M, N, K, kernel tag
 * (1 = gemm128_kernel, 2 = gemm128g_kernel, 3 = gemm160_kernel 128 x 160, 4 = 160 x 128; 0 = not recorded); returns the row count */
int w2l_profile_launches(int kind, int maxRows, double* ms, double* work, int* dims);
int w2l_arch_check(const char* archText, int nFeat, int nLabel, int* numLayers);
int w2l_flags_check(const char* flagsText, int* numFlags);

#ifdef __cplusplus
}
#endif
#endif /* W2L_HIP_H_ */
