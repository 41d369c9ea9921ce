// trainer.cpp -- one optimisation step of the acoustic-model trainer and its C ABI.
// Step order restated from the reference's hot loop (recipes/slimIPL/src/Train.cpp):
//   forward network :1463-1470 -> criterion forward :1675 -> zeroGrad + loss.backward() :1718-1720
//   -> all-reduce of all gradients :1721-1735 (here: ONE collective over the flat arena,
//   issued by the caller between w2l_trainer_forward_backward and w2l_trainer_update)
//   -> grads / totalBatchSize :1748-1784 -> clipGradNorm :1791-1798 -> critopt/netopt step :1801-1802.
// The five af::sync() per step of the reference are gone: a step enqueues ~500 kernels
// on one stream and never blocks the host.
#include <cmath>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

#include <algorithm>

#include "w2l_host.hpp"

namespace w2l {

struct Trainer {
  std::shared_ptr<Sequential> net;
  std::shared_ptr<SequenceCriterion> crit;
  int nFeat = 0, nLabel = 0;
  std::string critName;
  // geometry
  int B = 0, T = 0, L = 0, Tout = 0;
  size_t netFloats = 0, critFloats = 0, arenaFloats = 0, critWsBytes = 0;
  // bound device memory (owned by the caller)
  float *params = nullptr, *grads = nullptr, *mom = nullptr, *arena = nullptr;
  void* critWs = nullptr;
  float *loss = nullptr, *gradLoss = nullptr, *dEm = nullptr;
  double* sumsq = nullptr;   // acc[8]: see w2l_grad_guard (include/w2l_hip.h)
  float* batchSlot = nullptr;  // grads[paramFloats]: this rank's utterance count, summed by the SAME all-reduce as the gradients
  const float* emission = nullptr;
  uint32_t step = 0;
  std::shared_ptr<SequenceCriterion> linseg;  // --linseg warm-up criterion (ASG only), used while step < linsegUpdates
  uint32_t linsegUpdates = 0;
  int scaleMode = 0;
  SequenceCriterion* activeCrit() { return (linseg && step < linsegUpdates) ? linseg.get() : crit.get(); }
  std::string lastError;
  bool guardZeroed = false;
  int netOptim = 0, critOptim = 0;  // 0 SGD(momentum), 1 Adagrad (variance in the momentum arena), 2 Adadelta (+ state2)
  const float* inputSizes = nullptr;  // w2l_trainer_set_input_sizes
  float* state2 = nullptr;          // Adadelta's accDelta, same size and layout as the momentum arena
  bool mixedPrecision = false;   // fl's --fl_amp_use_mixed_precision, restated for bf16: the network's fl::Linear GEMMs multiply
                                 // in bf16 (fp32 accumulate, fp32 storage, fp32 master weights); convolutions, LayerNorm, the
                                 // criterion and the optimizer stay fp32
  std::vector<hipEvent_t> bucketEvents;  // owned (w2l_trainer_set_grad_buckets)
  ~Trainer() { for (auto e : bucketEvents) (void)hipEventDestroy(e); }
};

}  // namespace w2l

using namespace w2l;

#define W2L_API extern "C" __attribute__((visibility("default")))
#define TRY(h, body)                                   \
  try { body; return W2L_OK; }                         \
  catch (const std::invalid_argument& e) { if (h) ((Trainer*)h)->lastError = e.what(); g_err = e.what(); return W2L_EINVAL; } \
  catch (const std::exception& e) { if (h) ((Trainer*)h)->lastError = e.what(); g_err = e.what(); return W2L_EHIP; }

static thread_local std::string g_err;

W2L_API const char* w2l_host_last_error(void) { return g_err.c_str(); }

// criterion: "ctc" | "asg" ; archText: contents of an .arch file (NFEAT/NLABEL substituted here)
W2L_API void* w2l_trainer_create(const char* archText, int nFeat, int nLabel, const char* criterion,
                                 int scaleMode, double transdiag) {
  try {
    auto t = new Trainer();
    t->nFeat = nFeat; t->nLabel = nLabel; t->critName = criterion ? criterion : "ctc"; t->scaleMode = scaleMode;
    t->net = buildSequentialFromText(archText, nFeat, nLabel);
    if (t->critName == "ctc") t->crit = makeCTCLoss(scaleMode);
    else if (t->critName == "asg") t->crit = makeASGLoss(nLabel, scaleMode, transdiag);
    else { delete t; throw std::invalid_argument("unsupported criterion: " + std::string(criterion)); }
    t->netFloats = t->net->paramFloats();
    t->critFloats = t->crit->paramFloats();
    return t;
  } catch (const std::exception& e) {
    g_err = e.what();
    return nullptr;
  }
}

W2L_API void w2l_trainer_destroy(void* h) { delete (Trainer*)h; }

W2L_API size_t w2l_trainer_param_floats(void* h) { Trainer* t = (Trainer*)h; return t->netFloats + t->critFloats; }
W2L_API size_t w2l_trainer_net_param_floats(void* h) { return ((Trainer*)h)->netFloats; }
// size of the GRADIENT arena the caller binds: parameters + a 4-float tail whose first element carries the local
// batch size through the gradient all-reduce (the reference all-reduces it separately, Train.cpp:1743-1747)
W2L_API size_t w2l_trainer_grad_floats(void* h) { Trainer* t = (Trainer*)h; return t->netFloats + t->critFloats + 4; }
W2L_API int w2l_trainer_num_params(void* h) { return (int)((Trainer*)h)->net->params().size(); }
W2L_API const char* w2l_trainer_describe(void* h) {
  Trainer* t = (Trainer*)h;
  t->lastError = t->net->prettyString() + " | criterion: " + t->crit->prettyString();
  return t->lastError.c_str();
}
// name / numel / offset of network parameter i (internal arena order = Flashlight params() order)
W2L_API int w2l_trainer_param_info(void* h, int i, char* name, int nameCap, size_t* numel, size_t* offset) {
  Trainer* t = (Trainer*)h;
  if (i < 0 || i >= (int)t->net->params().size()) return W2L_EINVAL;
  const ParamInfo& p = t->net->params()[i];
  if (name && nameCap > 0) { std::strncpy(name, p.name.c_str(), nameCap - 1); name[nameCap - 1] = 0; }
  if (numel) *numel = p.numel;
  if (offset) *offset = p.offset;
  return W2L_OK;
}

// host-side parameter helpers (h_* are HOST pointers)
W2L_API int w2l_trainer_init_params(void* h, float* h_params, uint64_t seed) {
  Trainer* t = (Trainer*)h;
  TRY(h, { t->net->initParams(h_params, seed); t->crit->initParams(h_params + t->netFloats); });
}
W2L_API int w2l_trainer_import_param(void* h, int i, const float* h_ref, float* h_params) {
  Trainer* t = (Trainer*)h;
  TRY(h, { if (i < 0 || i >= (int)t->net->params().size()) throw std::invalid_argument("bad param index"); t->net->importParam(i, h_ref, h_params); });
}
W2L_API int w2l_trainer_export_param(void* h, int i, const float* h_arena, float* h_ref) {
  Trainer* t = (Trainer*)h;
  TRY(h, { if (i < 0 || i >= (int)t->net->params().size()) throw std::invalid_argument("bad param index"); t->net->exportParam(i, h_arena, h_ref); });
}

// plan for a batch geometry: B utterances of T frames, targets padded to L.
W2L_API int w2l_trainer_plan(void* h, int B, int T, int L, size_t* arenaFloats, size_t* critWsBytes,
                             int* Tout) {
  Trainer* t = (Trainer*)h;
  TRY(h, {
    if (B <= 0 || T <= 0 || L <= 0) throw std::invalid_argument("plan: B, T, L must be positive");
    size_t used = t->net->plan(B, T, t->nFeat);
    t->B = B; t->T = T; t->L = L;
    t->Tout = t->net->outAct().T;
    if (t->net->outAct().F != t->nLabel) throw std::invalid_argument("network output width != NLABEL");
    const size_t bSlots = ((size_t)B + 63) / 64 * 64;
    size_t extra = ((size_t)B * t->Tout * t->nLabel + 63) / 64 * 64 + 2 * bSlots + 64;  // dEmission, loss, gradLoss, guard doubles
    t->arenaFloats = used + extra;
    t->critWsBytes = t->crit->workspaceBytes(B, t->Tout, t->nLabel, L);
    if (t->linseg) t->critWsBytes = std::max(t->critWsBytes, t->linseg->workspaceBytes(B, t->Tout, t->nLabel, L));
    if (arenaFloats) *arenaFloats = t->arenaFloats;
    if (critWsBytes) *critWsBytes = t->critWsBytes;
    if (Tout) *Tout = t->Tout;
  });
}

// all device pointers; params/grads/momentum hold net params followed by criterion params
W2L_API int w2l_trainer_bind(void* h, float* params, float* grads, float* momentum, float* arena,
                             void* critWs) {
  Trainer* t = (Trainer*)h;
  if (!params || !grads || !arena || !critWs || !t->arenaFloats) return W2L_EINVAL;
  t->params = params; t->grads = grads; t->mom = momentum; t->arena = arena; t->critWs = critWs;
  const size_t bSlots = ((size_t)t->B + 63) / 64 * 64;
  const size_t emSlots = ((size_t)t->B * t->Tout * t->nLabel + 63) / 64 * 64;
  size_t used = t->arenaFloats - (emSlots + 2 * bSlots + 64);
  t->dEm = arena + used;
  t->loss = t->dEm + emSlots;
  t->gradLoss = t->loss + bSlots;
  t->sumsq = (double*)(t->gradLoss + bSlots);  // 64 floats = 32 doubles, 256-byte aligned
  t->batchSlot = grads + t->netFloats + t->critFloats;
  t->guardZeroed = false;
  return W2L_OK;
}

static void requireBound(Trainer* t) {
  if (!t->arenaFloats || !t->params || !t->grads || !t->arena || !t->critWs)
    throw std::invalid_argument("trainer not planned / bound (w2l_trainer_plan + w2l_trainer_bind first)");
}

struct MatmulMode {   // scoped: the mode only covers the network's own calls (the criterion stays fp32)
  int prev;
  bool on;
  explicit MatmulMode(bool bf16) : prev(0), on(bf16) { if (on) prev = w2l_set_matmul_precision(1); }
  ~MatmulMode() { if (on) w2l_set_matmul_precision(prev); }
};

static Ctx makeCtx(Trainer* t, void* stream, bool train) {
  requireBound(t);
  Ctx c;
  c.stream = (hipStream_t)stream;
  c.train = train;
  c.seed = 0x9E3779B9u * (t->step + 1);
  c.params = t->params;
  c.grads = t->grads;
  c.inputSizes = t->inputSizes; c.inputT = t->T;
  c.bf16 = t->mixedPrecision;
  return c;
}

// network forward only (eval mode when train == 0); returns the device pointer of emissions [B][T'][N]
W2L_API int w2l_trainer_forward(void* h, const float* x, int train, const float** emission, void* stream) {
  Trainer* t = (Trainer*)h;
  TRY(h, {
    if (!x) throw std::invalid_argument("forward: null input");
    Ctx c = makeCtx(t, stream, train != 0);
    { MatmulMode mm(t->mixedPrecision); t->emission = t->net->forward(c, t->arena, x); }
    if (emission) *emission = t->emission;
  });
}

// forward + criterion + backward: leaves every gradient (network ‖ criterion) in the grads
// arena, UNSCALED (sum over the utterances of this rank).  loss_out (device, [B]) optional.
W2L_API int w2l_trainer_forward_backward(void* h, const float* x, const int* target, float** lossDev,
                                         void* stream) {
  Trainer* t = (Trainer*)h;
  TRY(h, {
    if (!x || !target) throw std::invalid_argument("forward_backward: null input");   // before anything is enqueued with them
    Ctx c = makeCtx(t, stream, true);
    { MatmulMode mm(t->mixedPrecision); t->emission = t->net->forward(c, t->arena, x); }
    float* cp = t->params + t->netFloats;
    float* cg = t->grads + t->netFloats;
    SequenceCriterion* crit = t->activeCrit();
    crit->forward(c, t->B, t->Tout, t->nLabel, t->L, t->emission, target, t->loss, t->critWs, cp);
    // d(sum_b loss_b)/d loss_b = 1
    hipCheck(hipMemsetAsync(t->gradLoss, 0, sizeof(float) * (((size_t)t->B + 63) / 64 * 64), c.stream), "memset");
    w2lCheck(w2l_fill(t->gradLoss, (size_t)t->B, 1.f, c.stream), "fill");
    w2lCheck(w2l_fill(t->batchSlot, 1, (float)t->B, c.stream), "batch slot");  // rides the gradient all-reduce
    hipCheck(hipMemsetAsync(t->batchSlot + 1, 0, sizeof(float) * 3, c.stream), "memset");
    crit->backward(c, t->B, t->Tout, t->nLabel, t->L, t->emission, target, t->gradLoss, t->dEm, t->critWs, cp, cg);
    { MatmulMode mm(t->mixedPrecision); t->net->backward(c, t->arena, t->dEm); }
    if (lossDev) *lossDev = t->loss;
  });
}

// backward pass of the NETWORK alone from a caller-supplied gradient of the emissions [B][T'][N] -- the fl::Module boundary for
// a binder that keeps its own criterion (and the hook of the teacher-forced per-layer parity tests).  Call after
// w2l_trainer_forward(train = 1) of the same step: leaves the network's gradients in the grads arena (unscaled), the
// criterion's part untouched.
W2L_API int w2l_trainer_backward(void* h, const float* dEmission, void* stream) {
  Trainer* t = (Trainer*)h;
  TRY(h, {
    requireBound(t);
    if (!dEmission) throw std::invalid_argument("backward: null gradient");
    if (!t->emission) throw std::invalid_argument("backward: no forward pass to differentiate");
    Ctx c = makeCtx(t, stream, true);
    { MatmulMode mm(t->mixedPrecision); t->net->backward(c, t->arena, dEmission); }
  });
}

// optimizer: grads *= 1/totalBatch, global-norm clip (network, and criterion if clampCrit), SGD+momentum
// on the network (lr, momentum), plain SGD on the criterion (lrcrit)  [Train.cpp:577-582, :1791-1802]
W2L_API int w2l_trainer_update(void* h, float lr, float lrcrit, float momentum, float maxGradNorm,
                               float totalBatch, int clampCrit, void* stream) {
  Trainer* t = (Trainer*)h;
  TRY(h, {
    requireBound(t);
    hipStream_t s = (hipStream_t)stream;
    // totalBatch > 0: the caller's number; <= 0: the all-reduced utterance count in the gradient arena's tail
    const float gs = totalBatch > 0.f ? 1.f / totalBatch : 0.f;
    if (!t->guardZeroed) { hipCheck(hipMemsetAsync(t->sumsq, 0, sizeof(double) * 8, s), "memset"); t->guardZeroed = true; }
    // --linseg: the reference trains the warm-up phase with its OWN optimizers (linNetoptim / linCritoptim,
    // Train.cpp:589-617), so the ASG phase proper starts from fresh optimizer state: clear the momentum arena -- SGD
    // velocity, or the Adagrad / Adadelta gradient statistics it holds -- and Adadelta's second arena at the switch
    if (t->linseg && t->linsegUpdates && t->step == t->linsegUpdates) {
      const bool stateful = momentum != 0.f || t->netOptim != 0 || t->critOptim != 0;
      if (t->mom && stateful)
        hipCheck(hipMemsetAsync(t->mom, 0, sizeof(float) * (t->netFloats + t->critFloats), s), "linseg optimizer-state reset");
      if (t->state2 && (t->netOptim == 2 || t->critOptim == 2))
        hipCheck(hipMemsetAsync(t->state2, 0, sizeof(float) * (t->netFloats + t->critFloats), s), "linseg optimizer-state reset");
    }
    // the norm is taken on EVERY update (also with --maxgradnorm=0): it is the non-finite guard of the step
    w2lCheck(w2l_sumsq(t->grads, t->netFloats, t->sumsq, 1, s), "sumsq");
    w2lCheck(w2l_sumsq(t->grads + t->netFloats, t->critFloats, t->sumsq + 1, 1, s), "sumsq");
    w2lCheck(w2l_grad_guard(t->sumsq, totalBatch > 0.f ? nullptr : t->batchSlot, clampCrit, s), "guard");
    const bool needState = t->netOptim != 0 || (t->critOptim != 0 && t->critFloats);
    const bool needState2 = t->netOptim == 2 || (t->critOptim == 2 && t->critFloats);
    if (needState && !t->mom) throw std::invalid_argument("Adagrad / Adadelta need the momentum arena bound (it holds the gradient statistics)");
    if (needState2 && !t->state2) throw std::invalid_argument("Adadelta needs w2l_trainer_bind_state2");
    auto stepOne = [&](int kind, size_t off, size_t n, float rate, float mom, float clip, const char* what) {
      float* v = t->mom ? t->mom + off : nullptr;
      if (kind == 1)
        w2lCheck(w2l_adagrad_step_guarded(t->params + off, t->grads + off, v, n, rate, 1e-8f, gs, clip, t->sumsq + 2, s), what);
      else if (kind == 2)
        w2lCheck(w2l_adadelta_step_guarded(t->params + off, t->grads + off, v, t->state2 + off, n, rate, 0.9f, 1e-8f, gs, clip, t->sumsq + 2, s), what);
      else
        w2lCheck(w2l_sgd_step_guarded(t->params + off, t->grads + off, mom != 0.f ? v : nullptr, n, rate, mom, gs, clip, t->sumsq + 2, s), what);
    };
    if (t->critFloats) stepOne(t->critOptim, t->netFloats, t->critFloats, lrcrit, 0.f, clampCrit ? maxGradNorm : 0.f, "criterion optimizer");
    stepOne(t->netOptim, 0, t->netFloats, lr, momentum, maxGradNorm, "network optimizer");
    t->step++;
  });
}

W2L_API int w2l_trainer_viterbi(void* h, const float* emission, int* path, void* stream) {
  Trainer* t = (Trainer*)h;
  TRY(h, {
    Ctx c = makeCtx(t, stream, false);
    t->crit->viterbiPath(c, t->B, t->Tout, t->nLabel, emission, path, t->critWs, t->params + t->netFloats);
  });
}

// Gradient buckets for the data-parallel overlap (fl's CoalescingReducer, recipes/slimIPL/src/Train.cpp:1721-1735,
// restated for one flat arena): offsets[0..n) ascending float offsets into the gradient arena, bucket k =
// [offsets[k], offsets[k+1]) and the last one runs to the end (criterion gradients included).  During
// w2l_trainer_forward_backward an event per bucket is recorded on the step's stream once every gradient at
// offset >= offsets[k] is final; w2l_trainer_wait_bucket makes another stream (the collective's) wait for it.
// n = 0 removes the hooks.
W2L_API int w2l_trainer_set_grad_buckets(void* h, int n, const size_t* offsets) {
  Trainer* t = (Trainer*)h;
  TRY(h, {
    if (n < 0 || (n > 0 && !offsets)) throw std::invalid_argument("set_grad_buckets: bad arguments");
    for (int k = 1; k < n; ++k)
      if (offsets[k] <= offsets[k - 1]) throw std::invalid_argument("set_grad_buckets: offsets must ascend");
    for (auto e : t->bucketEvents) (void)hipEventDestroy(e);
    t->bucketEvents.clear();
    std::vector<size_t> off(offsets, offsets + n);
    for (int k = 0; k < n; ++k) {
      hipEvent_t e;
      hipCheck(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate");
      t->bucketEvents.push_back(e);
    }
    t->net->setGradBuckets(off, t->bucketEvents);
  });
}
W2L_API int w2l_trainer_wait_bucket(void* h, int k, void* stream) {
  Trainer* t = (Trainer*)h;
  TRY(h, {
    if (k < 0 || k >= (int)t->bucketEvents.size()) throw std::invalid_argument("wait_bucket: no such bucket");
    hipCheck(hipStreamWaitEvent((hipStream_t)stream, t->bucketEvents[k], 0), "hipStreamWaitEvent");
  });
}

// --linseg=n (Train.cpp:589-617, :1866-1883): the first n updates of an ASG run use LinSegCriterion on the ASG
// criterion's own transitions.  Call before w2l_trainer_plan (the criterion workspace is sized for both).
W2L_API int w2l_trainer_set_linseg(void* h, uint32_t updates) {
  Trainer* t = (Trainer*)h;
  TRY(h, {
    if (updates && t->critName != "asg") throw std::invalid_argument("linseg may only be used with ASG criterion");
    t->linsegUpdates = updates;
    t->linseg = updates ? makeLinSegCriterion(t->nLabel, t->scaleMode) : nullptr;
    t->arenaFloats = 0;  // forces a new plan / bind
    t->params = t->grads = t->mom = t->arena = nullptr;
    t->critWs = nullptr;
  });
}

// gradient norm of the last w2l_trainer_update (sqrt of the clipped-norm accumulator, BEFORE the 1/totalBatch scale);
// NaN / Inf means that update was skipped.  Synchronises `stream`.
W2L_API int w2l_trainer_grad_norm(void* h, double* norm, void* stream) {
  Trainer* t = (Trainer*)h;
  TRY(h, {
    if (!t->sumsq || !norm) throw std::invalid_argument("trainer not bound");
    double s = 0.0;
    hipCheck(hipMemcpyAsync(&s, t->sumsq + 2, sizeof(double), hipMemcpyDeviceToHost, (hipStream_t)stream), "grad norm");
    hipCheck(hipStreamSynchronize((hipStream_t)stream), "grad norm");
    *norm = std::sqrt(s);
  });
}

// number of updates skipped so far because the (all-reduced) gradient or batch size was non-finite.  Synchronises.
W2L_API int w2l_trainer_skipped_updates(void* h, uint64_t* count, void* stream) {
  Trainer* t = (Trainer*)h;
  TRY(h, {
    if (!t->sumsq || !count) throw std::invalid_argument("trainer not bound");
    double s = 0.0;
    if (t->guardZeroed) {
      hipCheck(hipMemcpyAsync(&s, t->sumsq + 4, sizeof(double), hipMemcpyDeviceToHost, (hipStream_t)stream), "skipped");
      hipCheck(hipStreamSynchronize((hipStream_t)stream), "skipped");
    }
    *count = (uint64_t)s;
  });
}

W2L_API int w2l_trainer_set_optimizer(void* h, int netKind, int critKind) {
  if (!h || netKind < 0 || netKind > 2 || critKind < 0 || critKind > 2) return W2L_EINVAL;
  ((Trainer*)h)->netOptim = netKind; ((Trainer*)h)->critOptim = critKind;
  return W2L_OK;
}
W2L_API int w2l_trainer_set_input_sizes(void* h, const float* inputSizesDev) {
  if (!h) return W2L_EINVAL;
  ((Trainer*)h)->inputSizes = inputSizesDev;
  return W2L_OK;
}
W2L_API int w2l_trainer_bind_state2(void* h, float* state2) {
  if (!h) return W2L_EINVAL;
  ((Trainer*)h)->state2 = state2;
  return W2L_OK;
}
W2L_API int w2l_trainer_set_mixed_precision(void* h, int on) { ((Trainer*)h)->mixedPrecision = on != 0; return W2L_OK; }

W2L_API int w2l_trainer_set_step(void* h, uint32_t step) { ((Trainer*)h)->step = step; return W2L_OK; }

// arch / flags parsing without building kernels (recipes must load unchanged)
W2L_API int w2l_arch_check(const char* archText, int nFeat, int nLabel, int* numLayers) {
  TRY(nullptr, { auto v = parseArch(archText, nFeat, nLabel); if (numLayers) *numLayers = (int)v.size(); });
}
W2L_API int w2l_flags_check(const char* flagsText, int* numFlags) {
  TRY(nullptr, { auto f = parseFlagsText(flagsText); if (numFlags) *numFlags = (int)f.kv.size(); });
}
