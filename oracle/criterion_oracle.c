/*
 * oracle/criterion_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (fp64 internal arithmetic) of the wav2letter / Flashlight
 * sequence criteria on the acoustic-training hot path:
 *   ForceAlignmentCriterion (FAC), FullConnectionCriterion (FCC),
 *   ViterbiPath, ConnectionistTemporalClassificationCriterion (CTC),
 *   CriterionUtils (target size, scale modes).
 *
 * PARITY UNPINNED.  The arithmetic of these criteria lives in Flashlight <= 0.3.2
 * (flashlight/lib/sequence/criterion/cpu/), which is NOT vendored under
 * /root/reference (README.md:7-9,17; CMakeLists.txt:9-13) and cannot be built
 * here.  /root/reference holds only the call sites:
 *   recipes/slimIPL/src/Train.cpp:406-410  (CTCLoss / ASGLoss construction)
 *   recipes/slimIPL/src/Train.cpp:1675     (crit->forward({emission,target}))
 *   recipes/slimIPL/src/Train.cpp:834-838  (viterbiPath per sample, dims N,T,B)
 *   recipes/slimIPL/src/Train.cpp:248-251  (CTC blank appended LAST => N-1)
 *   recipes/slimIPL/src/Train.cpp:389      (scale mode from --onorm/--sqnorm)
 * and no golden vectors for them.  The recurrences below follow SURVEY.md
 * Appendix B (published ASG/CTC algorithms: Collobert et al. 2016
 * "Wav2Letter", Graves et al. 2006 "CTC") and are pinned in tests/ by
 *   (1) brute-force path enumeration, (2) analytic identities (App. B.6),
 *   (3) fp64 finite differences, (4) torch.nn.functional.ctc_loss on CPU.
 *
 * Layout conventions: emissions are ArrayFire dims
 * (N,T,B) == row-major C array [B][T][N]; targets [B][L] int32 padded with
 * negative values; transitions [N][N] indexed [to][from]; CTC blank = N-1;
 * Viterbi paths [B][T] int32.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.
 */
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define W2L_EXPORT __attribute__((visibility("default")))

enum { SCALE_NONE = 0, SCALE_INPUT_SZ = 1, SCALE_INPUT_SZ_SQRT = 2,
       SCALE_TARGET_SZ = 3, SCALE_TARGET_SZ_SQRT = 4 };

static const double NEG_INF = -INFINITY;

static inline double lse2(double a, double b) {
  if (a == NEG_INF) return b;
  if (b == NEG_INF) return a;
  double m = a > b ? a : b;
  return m + log(exp(a - m) + exp(b - m));
}

/* ---- CriterionUtils ------------------------------ */

/* number of leading non-negative labels, clamped to maxSize (== T for ASG) */
W2L_EXPORT int w2l_oracle_target_size(const int* target, int L, int maxSize) {
  int n = 0;
  while (n < L && target[n] >= 0) ++n;
  if (n > maxSize) n = maxSize;
  return n;
}

W2L_EXPORT void w2l_oracle_batch_target_size(int B, int L, int maxSize,
                                             const int* target, int* targetSize) {
  for (int b = 0; b < B; ++b)
    targetSize[b] = w2l_oracle_target_size(target + (size_t)b * L, L, maxSize);
}

/* CTC: truncate targets that cannot be emitted in T frames.
 * R = adjacent repeats in y[0..L); L <- min(L + R, T) - R.  (App. B.0) */
W2L_EXPORT int w2l_oracle_ctc_target_size(const int* target, int L, int T) {
  int n = 0;
  while (n < L && target[n] >= 0) ++n;
  int R = 0;
  for (int i = 1; i < n; ++i) R += (target[i] == target[i - 1]);
  int m = n + R < T ? n + R : T;
  m -= R;
  return m < 0 ? 0 : m;
}

W2L_EXPORT void w2l_oracle_batch_ctc_target_size(int B, int L, int T,
                                                 const int* target, int* targetSize) {
  for (int b = 0; b < B; ++b)
    targetSize[b] = w2l_oracle_ctc_target_size(target + (size_t)b * L, L, T);
}

static int wide_labels(int B, int N) {
#ifdef _OPENMP
  return N >= 512 && B < omp_get_max_threads();
#else
  (void)B; (void)N;
  return 0;
#endif
}

static double scale_of(int mode, int T, int L) {
  switch (mode) {
    case SCALE_NONE: return 1.0;
    case SCALE_INPUT_SZ: return T > 0 ? 1.0 / T : 1.0;
    case SCALE_INPUT_SZ_SQRT: return T > 0 ? sqrt(1.0 / T) : 1.0;
    case SCALE_TARGET_SZ: return L > 0 ? 1.0 / L : 1.0;
    case SCALE_TARGET_SZ_SQRT: return L > 0 ? sqrt(1.0 / L) : 1.0;
    default: return 1.0;
  }
}

W2L_EXPORT void w2l_oracle_compute_scale(int B, int T, int N, int scaleMode,
                                         const int* targetSize, double* scale) {
  (void)N;
  for (int b = 0; b < B; ++b) scale[b] = scale_of(scaleMode, T, targetSize[b]);
}

/* ---- FullConnectionCriterion (App. B.2) --------------------------------- */
/* alpha workspace: [B][T][N] doubles, kept by the caller between fwd and bwd */

W2L_EXPORT size_t w2l_oracle_fcc_workspace_size(int B, int T, int N) {
  return sizeof(double) * ((size_t)B * T * N + (size_t)B);
}

W2L_EXPORT void w2l_oracle_fcc_forward(int B, int T, int N, int scaleMode,
                                       const float* input, const int* targetSize,
                                       const float* trans, double* loss,
                                       void* workspace) {
  double* alphaAll = (double*)workspace;
  double* scaleAll = alphaAll + (size_t)B * T * N;
  /* few utterances x many labels (the N = 9998 checks): the rows i of one step are independent,
   * so the threads go to the i loop instead of the b loop; same arithmetic per (b, t, i) */
  const int inner = wide_labels(B, N);
#pragma omp parallel for if (!inner)
  for (int b = 0; b < B; ++b) {
    const float* x = input + (size_t)b * T * N;
    double* alpha = alphaAll + (size_t)b * T * N;
    double s = scale_of(scaleMode, T, targetSize[b]);
    scaleAll[b] = s;
    for (int i = 0; i < N; ++i) alpha[i] = x[i];
    for (int t = 1; t < T; ++t) {
      const double* ap = alpha + (size_t)(t - 1) * N;
      double* ac = alpha + (size_t)t * N;
#pragma omp parallel for if (inner)
      for (int i = 0; i < N; ++i) {
        double m = NEG_INF;
        for (int j = 0; j < N; ++j) {
          double v = ap[j] + (double)trans[(size_t)i * N + j];
          if (v > m) m = v;
        }
        double sum = 0;
        for (int j = 0; j < N; ++j)
          sum += exp(ap[j] + (double)trans[(size_t)i * N + j] - m);
        ac[i] = (double)x[(size_t)t * N + i] + m + log(sum);
      }
    }
    const double* al = alpha + (size_t)(T - 1) * N;
    double m = NEG_INF;
    for (int i = 0; i < N; ++i) if (al[i] > m) m = al[i];
    double sum = 0;
    for (int i = 0; i < N; ++i) sum += exp(al[i] - m);
    loss[b] = s * (m + log(sum));
  }
}

/* grad [B] upstream; inputGrad [B][T][N]; transGrad [N][N] (summed over b) */
W2L_EXPORT void w2l_oracle_fcc_backward(int B, int T, int N, const float* trans,
                                        const double* grad, double* inputGrad,
                                        double* transGrad, void* workspace) {
  double* alphaAll = (double*)workspace;
  double* scaleAll = alphaAll + (size_t)B * T * N;
  const int inner = wide_labels(B, N);
  /* inner mode runs the utterances one after the other: they can share one accumulator */
  double* tgBatch = (double*)calloc((size_t)(inner ? 1 : B) * N * N, sizeof(double));
#pragma omp parallel for if (!inner)
  for (int b = 0; b < B; ++b) {
    const double* alpha = alphaAll + (size_t)b * T * N;
    double* dx = inputGrad + (size_t)b * T * N;
    double* tg = tgBatch + (inner ? 0 : (size_t)b * N * N);
    double g = scaleAll[b] * grad[b];
    double* da = (double*)malloc(sizeof(double) * 2 * N);
    double* dprev = da + N;
    /* d loss / d alpha[T-1] = softmax(alpha[T-1]) */
    const double* al = alpha + (size_t)(T - 1) * N;
    double m = NEG_INF;
    for (int i = 0; i < N; ++i) if (al[i] > m) m = al[i];
    double sum = 0;
    for (int i = 0; i < N; ++i) sum += exp(al[i] - m);
    for (int i = 0; i < N; ++i) da[i] = exp(al[i] - m) / sum;
    for (int t = T - 1; t >= 1; --t) {
      const double* ap = alpha + (size_t)(t - 1) * N;
      for (int j = 0; j < N; ++j) dprev[j] = 0;
#pragma omp parallel for if (inner) reduction(+ : dprev[:N])
      for (int i = 0; i < N; ++i) {
        dx[(size_t)t * N + i] = g * da[i];
        /* lse_i = logsumexp_j(ap[j] + trans[i][j]) recomputed stably */
        double mm = NEG_INF;
        for (int j = 0; j < N; ++j) {
          double v = ap[j] + (double)trans[(size_t)i * N + j];
          if (v > mm) mm = v;
        }
        double ss = 0;
        for (int j = 0; j < N; ++j)
          ss += exp(ap[j] + (double)trans[(size_t)i * N + j] - mm);
        for (int j = 0; j < N; ++j) {
          double w = exp(ap[j] + (double)trans[(size_t)i * N + j] - mm) / ss;
          dprev[j] += da[i] * w;
          tg[(size_t)i * N + j] += g * da[i] * w;
        }
      }
      for (int j = 0; j < N; ++j) da[j] = dprev[j];
    }
    for (int i = 0; i < N; ++i) dx[i] = g * da[i];
    free(da);
  }
  for (size_t k = 0; k < (size_t)N * N; ++k) {
    double s = 0;
    for (int b = 0; b < (inner ? 1 : B); ++b) s += tgBatch[(size_t)b * N * N + k];
    transGrad[k] = s;
  }
  free(tgBatch);
}

/* ---- ForceAlignmentCriterion (App. B.1) --------------------------------- */
/* workspace: alpha [B][T][L] doubles + scale [B] */

W2L_EXPORT size_t w2l_oracle_fac_workspace_size(int B, int T, int N, int L) {
  (void)N;
  return sizeof(double) * ((size_t)B * T * L + (size_t)B);
}

W2L_EXPORT void w2l_oracle_fac_forward(int B, int T, int N, int L, int scaleMode,
                                       const float* input, const int* target,
                                       const int* targetSize, const float* trans,
                                       double* loss, void* workspace) {
  double* alphaAll = (double*)workspace;
  double* scaleAll = alphaAll + (size_t)B * T * L;
#pragma omp parallel for
  for (int b = 0; b < B; ++b) {
    const float* x = input + (size_t)b * T * N;
    const int* y = target + (size_t)b * L;
    double* alpha = alphaAll + (size_t)b * T * L;
    int S = targetSize[b];
    double s = scale_of(scaleMode, T, S);
    scaleAll[b] = s;
    if (S <= 0) { loss[b] = 0; continue; }
    for (size_t k = 0; k < (size_t)T * L; ++k) alpha[k] = NEG_INF;
    alpha[0] = x[y[0]];
    for (int t = 1; t < T; ++t) {
      const double* ap = alpha + (size_t)(t - 1) * L;
      double* ac = alpha + (size_t)t * L;
      int high = t < S ? t : S;
      int low = (T - t) < S ? S - (T - t) : 1;
      if (T - t >= S)
        ac[0] = ap[0] + (double)trans[(size_t)y[0] * N + y[0]] + (double)x[(size_t)t * N + y[0]];
      if (t < S)
        ac[high] = ap[high - 1] + (double)trans[(size_t)y[high] * N + y[high - 1]] +
                   (double)x[(size_t)t * N + y[high]];
      for (int i = low; i < high; ++i) {
        double s1 = ap[i] + (double)trans[(size_t)y[i] * N + y[i]];
        double s2 = ap[i - 1] + (double)trans[(size_t)y[i] * N + y[i - 1]];
        ac[i] = lse2(s1, s2) + (double)x[(size_t)t * N + y[i]];
      }
    }
    loss[b] = s * alpha[(size_t)(T - 1) * L + (S - 1)];
  }
}

W2L_EXPORT void w2l_oracle_fac_backward(int B, int T, int N, int L,
                                        const int* target, const int* targetSize,
                                        const float* trans, const double* grad,
                                        double* inputGrad, double* transGrad,
                                        void* workspace) {
  double* alphaAll = (double*)workspace;
  double* scaleAll = alphaAll + (size_t)B * T * L;
  const int inner = wide_labels(B, N);  /* few utterances x many labels: one after the other, one accumulator */
  double* tgBatch = (double*)calloc((size_t)(inner ? 1 : B) * N * N, sizeof(double));
  memset(inputGrad, 0, sizeof(double) * (size_t)B * T * N);
#pragma omp parallel for if (!inner)
  for (int b = 0; b < B; ++b) {
    const int* y = target + (size_t)b * L;
    const double* alpha = alphaAll + (size_t)b * T * L;
    double* dx = inputGrad + (size_t)b * T * N;
    double* tg = tgBatch + (inner ? 0 : (size_t)b * N * N);
    int S = targetSize[b];
    if (S <= 0) continue;
    double g = scaleAll[b] * grad[b];
    double* da = (double*)calloc((size_t)2 * L, sizeof(double));
    double* dprev = da + L;
    da[S - 1] = 1.0;
    for (int t = T - 1; t >= 1; --t) {
      const double* ap = alpha + (size_t)(t - 1) * L;
      int high = t < S ? t : S;
      int low = (T - t) < S ? S - (T - t) : 1;
      for (int i = 0; i < S; ++i) dprev[i] = 0;
      for (int i = 0; i < S; ++i)
        if (da[i] != 0) dx[(size_t)t * N + y[i]] += g * da[i];
      if (T - t >= S) {
        dprev[0] += da[0];
        tg[(size_t)y[0] * N + y[0]] += g * da[0];
      }
      if (t < S) {
        dprev[high - 1] += da[high];
        tg[(size_t)y[high] * N + y[high - 1]] += g * da[high];
      }
      for (int i = low; i < high; ++i) {
        double s1 = ap[i] + (double)trans[(size_t)y[i] * N + y[i]];
        double s2 = ap[i - 1] + (double)trans[(size_t)y[i] * N + y[i - 1]];
        double m = lse2(s1, s2);
        double w1 = (s1 == NEG_INF) ? 0.0 : exp(s1 - m);
        double w2 = (s2 == NEG_INF) ? 0.0 : exp(s2 - m);
        dprev[i] += da[i] * w1;
        dprev[i - 1] += da[i] * w2;
        tg[(size_t)y[i] * N + y[i]] += g * da[i] * w1;
        tg[(size_t)y[i] * N + y[i - 1]] += g * da[i] * w2;
      }
      for (int i = 0; i < S; ++i) da[i] = dprev[i];
    }
    dx[y[0]] += g * da[0];
    free(da);
  }
  for (size_t k = 0; k < (size_t)N * N; ++k) {
    double s = 0;
    for (int b = 0; b < (inner ? 1 : B); ++b) s += tgBatch[(size_t)b * N * N + k];
    transGrad[k] = s;
  }
  free(tgBatch);
}

/* forced alignment: max instead of LSE; ties: "stay" (s1) wins unless s2 > s1.
 * bestPaths [B][T] holds TARGET LABELS (not positions). fp32 arithmetic in the
 * order (alpha + trans) + x so a device implementation can be bit-exact. */
W2L_EXPORT void w2l_oracle_fac_viterbi(int B, int T, int N, int L,
                                       const float* input, const int* target,
                                       const int* targetSize, const float* trans,
                                       int* bestPaths) {
#pragma omp parallel for
  for (int b = 0; b < B; ++b) {
    const float* x = input + (size_t)b * T * N;
    const int* y = target + (size_t)b * L;
    int* path = bestPaths + (size_t)b * T;
    int S = targetSize[b];
    if (S <= 0) { for (int t = 0; t < T; ++t) path[t] = -1; continue; }
    float* alpha = (float*)malloc(sizeof(float) * (size_t)2 * S);
    unsigned char* bp = (unsigned char*)calloc((size_t)T * S, 1); /* 1 = came from i-1 */
    float* ap = alpha; float* ac = alpha + S;
    for (int i = 0; i < S; ++i) ap[i] = -INFINITY;
    ap[0] = x[y[0]];
    for (int t = 1; t < T; ++t) {
      int high = t < S ? t : S;
      int low = (T - t) < S ? S - (T - t) : 1;
      for (int i = 0; i < S; ++i) ac[i] = -INFINITY;
      if (T - t >= S)
        ac[0] = (ap[0] + trans[(size_t)y[0] * N + y[0]]) + x[(size_t)t * N + y[0]];
      if (t < S) {
        ac[high] = (ap[high - 1] + trans[(size_t)y[high] * N + y[high - 1]]) +
                   x[(size_t)t * N + y[high]];
        bp[(size_t)t * S + high] = 1;
      }
      for (int i = low; i < high; ++i) {
        float s1 = ap[i] + trans[(size_t)y[i] * N + y[i]];
        float s2 = ap[i - 1] + trans[(size_t)y[i] * N + y[i - 1]];
        if (s2 > s1) { ac[i] = s2 + x[(size_t)t * N + y[i]]; bp[(size_t)t * S + i] = 1; }
        else { ac[i] = s1 + x[(size_t)t * N + y[i]]; }
      }
      float* tmp = ap; ap = ac; ac = tmp;
    }
    int i = S - 1;
    for (int t = T - 1; t >= 0; --t) {
      path[t] = y[i];
      if (t > 0 && bp[(size_t)t * S + i]) --i;
    }
    free(alpha); free(bp);
  }
}

/* ---- ViterbiPath (App. B.3) --------------------------------------------- */
/* fp32 arithmetic: best_j (delta[t-1][j] + trans[i][j]) scanning j upward with
 * strict '>', then + x[t][i]; final state = first argmax. */
W2L_EXPORT void w2l_oracle_viterbi_compute(int B, int T, int N, const float* input,
                                           const float* trans, int* path) {
#pragma omp parallel for
  for (int b = 0; b < B; ++b) {
    const float* x = input + (size_t)b * T * N;
    int* p = path + (size_t)b * T;
    float* delta = (float*)malloc(sizeof(float) * (size_t)2 * N);
    int* psi = (int*)malloc(sizeof(int) * (size_t)T * N);
    float* dp = delta; float* dc = delta + N;
    for (int i = 0; i < N; ++i) dp[i] = x[i];
    for (int t = 1; t < T; ++t) {
      for (int i = 0; i < N; ++i) {
        float best = dp[0] + trans[(size_t)i * N];
        int arg = 0;
        for (int j = 1; j < N; ++j) {
          float v = dp[j] + trans[(size_t)i * N + j];
          if (v > best) { best = v; arg = j; }
        }
        dc[i] = best + x[(size_t)t * N + i];
        psi[(size_t)t * N + i] = arg;
      }
      float* tmp = dp; dp = dc; dc = tmp;
    }
    int arg = 0; float best = dp[0];
    for (int i = 1; i < N; ++i) if (dp[i] > best) { best = dp[i]; arg = i; }
    p[T - 1] = arg;
    for (int t = T - 1; t >= 1; --t) { arg = psi[(size_t)t * N + arg]; p[t - 1] = arg; }
    free(delta); free(psi);
  }
}

/* ---- CTC (App. B.4) ------------------------------------------------------ */
/* workspace: per b: logZ-per-frame lse [T], alpha [T][S], beta [T][S] with
 * S = 2L+1 (allocated for the padded L), + scale [B] + nll [B] */

W2L_EXPORT size_t w2l_oracle_ctc_workspace_size(int B, int T, int N, int L) {
  (void)N;
  size_t S = 2 * (size_t)L + 1;
  return sizeof(double) * ((size_t)B * (T + 2 * T * S) + 2 * (size_t)B);
}

W2L_EXPORT void w2l_oracle_ctc_forward(int B, int T, int N, int L, int scaleMode,
                                       const float* input, const int* target,
                                       const int* targetSize, double* loss,
                                       void* workspace) {
  const size_t Smax = 2 * (size_t)L + 1;
  double* ws = (double*)workspace;
  const size_t per = (size_t)T + 2 * (size_t)T * Smax;
  double* scaleAll = ws + (size_t)B * per;
  double* nllAll = scaleAll + B;
  const int blank = N - 1;
#pragma omp parallel for
  for (int b = 0; b < B; ++b) {
    const float* x = input + (size_t)b * T * N;
    const int* y = target + (size_t)b * L;
    double* lse = ws + (size_t)b * per;
    double* alpha = lse + T;
    int Lb = targetSize[b];
    int S = 2 * Lb + 1;
    double s = scale_of(scaleMode, T, Lb);
    scaleAll[b] = s;
    for (int t = 0; t < T; ++t) {
      double m = NEG_INF;
      for (int n = 0; n < N; ++n) if ((double)x[(size_t)t * N + n] > m) m = x[(size_t)t * N + n];
      double sum = 0;
      for (int n = 0; n < N; ++n) sum += exp((double)x[(size_t)t * N + n] - m);
      lse[t] = m + log(sum);
    }
#define EXT(sidx) (((sidx) & 1) ? y[(sidx) >> 1] : blank)
#define LP(t, sidx) ((double)x[(size_t)(t) * N + EXT(sidx)] - lse[t])
    for (size_t k = 0; k < (size_t)T * Smax; ++k) alpha[k] = NEG_INF;
    alpha[0] = LP(0, 0);
    if (S > 1) alpha[1] = LP(0, 1);
    for (int t = 1; t < T; ++t) {
      const double* ap = alpha + (size_t)(t - 1) * Smax;
      double* ac = alpha + (size_t)t * Smax;
      for (int si = 0; si < S; ++si) {
        double v = ap[si];
        if (si >= 1) v = lse2(v, ap[si - 1]);
        if (si >= 2 && (si & 1) && EXT(si) != EXT(si - 2)) v = lse2(v, ap[si - 2]);
        ac[si] = (v == NEG_INF) ? NEG_INF : v + LP(t, si);
      }
    }
    const double* al = alpha + (size_t)(T - 1) * Smax;
    double ll = al[S - 1];
    if (S > 1) ll = lse2(ll, al[S - 2]);
    nllAll[b] = -ll;
    loss[b] = -s * ll;
  }
}

W2L_EXPORT void w2l_oracle_ctc_backward(int B, int T, int N, int L,
                                        const float* input, const int* target,
                                        const int* targetSize, const double* grad,
                                        double* inputGrad, void* workspace) {
  const size_t Smax = 2 * (size_t)L + 1;
  double* ws = (double*)workspace;
  const size_t per = (size_t)T + 2 * (size_t)T * Smax;
  double* scaleAll = ws + (size_t)B * per;
  double* nllAll = scaleAll + B;
  const int blank = N - 1;
#pragma omp parallel for
  for (int b = 0; b < B; ++b) {
    const float* x = input + (size_t)b * T * N;
    const int* y = target + (size_t)b * L;
    double* lse = ws + (size_t)b * per;
    double* alpha = lse + T;
    double* beta = alpha + (size_t)T * Smax;
    double* dx = inputGrad + (size_t)b * T * N;
    int Lb = targetSize[b];
    int S = 2 * Lb + 1;
    double g = scaleAll[b] * grad[b];
    double ll = -nllAll[b];
    for (size_t k = 0; k < (size_t)T * Smax; ++k) beta[k] = NEG_INF;
    double* bl = beta + (size_t)(T - 1) * Smax;
    bl[S - 1] = LP(T - 1, S - 1);
    if (S > 1) bl[S - 2] = LP(T - 1, S - 2);
    for (int t = T - 2; t >= 0; --t) {
      const double* bn = beta + (size_t)(t + 1) * Smax;
      double* bc = beta + (size_t)t * Smax;
      for (int si = 0; si < S; ++si) {
        double v = bn[si];
        if (si + 1 < S) v = lse2(v, bn[si + 1]);
        if (si + 2 < S && (si & 1) && EXT(si) != EXT(si + 2)) v = lse2(v, bn[si + 2]);
        bc[si] = (v == NEG_INF) ? NEG_INF : v + LP(t, si);
      }
    }
    for (int t = 0; t < T; ++t) {
      for (int n = 0; n < N; ++n)
        dx[(size_t)t * N + n] = g * exp((double)x[(size_t)t * N + n] - lse[t]);
      if (ll == NEG_INF) continue; /* infeasible: grad = softmax (occupancy 0) */
      for (int si = 0; si < S; ++si) {
        double a = alpha[(size_t)t * Smax + si], be = beta[(size_t)t * Smax + si];
        if (a == NEG_INF || be == NEG_INF) continue;
        dx[(size_t)t * N + EXT(si)] -= g * exp(a + be - LP(t, si) - ll);
      }
    }
  }
#undef EXT
#undef LP
}

/* CTC viterbiPath = per-frame argmax, first max wins (App. B.4) */
W2L_EXPORT void w2l_oracle_ctc_viterbi(int B, int T, int N, const float* input, int* path) {
#pragma omp parallel for
  for (int b = 0; b < B; ++b)
    for (int t = 0; t < T; ++t) {
      const float* r = input + ((size_t)b * T + t) * N;
      int arg = 0; float best = r[0];
      for (int n = 1; n < N; ++n) if (r[n] > best) { best = r[n]; arg = n; }
      path[(size_t)b * T + t] = arg;
    }
}

W2L_EXPORT int w2l_oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

W2L_EXPORT void w2l_oracle_set_num_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}
