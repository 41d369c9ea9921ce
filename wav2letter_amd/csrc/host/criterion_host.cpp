// criterion_host.cpp -- fl::pkg::speech::{CTCLoss, ASGLoss} over the kernel C ABI.
// Construction and use in the reference: recipes/slimIPL/src/Train.cpp:406-410 (ctor),
// :1675 (forward), :1720 (backward), :838 (viterbiPath).  ASGLoss = FullConnectionCriterion
// - ForceAlignmentCriterion sharing one N x N transition parameter initialised to
// transdiag * I (--transdiag, recipes/conv_glu/librispeech/train.cfg:25).
#include <cstring>
#include <mutex>
#include <stdexcept>

#include "w2l_host.hpp"
#include "../criterion_asg_fused.hpp"

namespace w2l {
namespace {

inline size_t up(size_t v) { return (v + 255) / 256 * 256; }

class CTCLossImpl : public SequenceCriterion {
 public:
  explicit CTCLossImpl(int mode) : mode_(mode) {}
  std::string prettyString() const override { return "ConnectionistTemporalClassificationCriterion"; }
  size_t workspaceBytes(int B, int T, int N, int L) const override {
    return up(sizeof(int) * B) + w2l_ctc_workspace_size(B, T, N, L);
  }
  void forward(Ctx& c, int B, int T, int N, int L, const float* em, const int* target, float* loss, void* ws,
               float*) override {
    int* ts = (int*)ws;
    void* kws = (char*)ws + up(sizeof(int) * B);
    w2lCheck(w2l_batch_ctc_target_size(B, L, T, target, ts, c.stream), "ctc target size");
    w2lCheck(w2l_ctc_forward(B, T, N, L, mode_, em, target, ts, loss, kws, c.stream), "ctc forward");
  }
  void backward(Ctx& c, int B, int T, int N, int L, const float* em, const int* target, const float* gradLoss,
                float* dEm, void* ws, float*, float*) override {
    int* ts = (int*)ws;
    void* kws = (char*)ws + up(sizeof(int) * B);
    w2lCheck(w2l_ctc_backward(B, T, N, L, em, target, ts, gradLoss, dEm, kws, c.stream), "ctc backward");
  }
  void viterbiPath(Ctx& c, int B, int T, int N, const float* em, int* path, void*, float*) override {
    w2lCheck(w2l_ctc_viterbi(B, T, N, em, path, c.stream), "ctc viterbi");
  }

 private:
  int mode_;
};

// FCC and FAC are independent given the emissions: each is a length-T serial scan that occupies B of the 256 CUs
// (one wave / workgroup per utterance), so the two run SIDE BY SIDE on the caller's stream and a library-owned side
// stream (fork / join with events); ASG forward = max(FCC, FAC) instead of their sum, same for backward.
//
// AsgSequence: that launch sequence, shared by the criterion object below and by the one-call C ABI (w2l_asg_forward /
// w2l_asg_backward at the end of this file).  Small label sets (the letter recipes: N <= 31, L <= 320) take the fused sequences of
// criterion_asg_fused.hpp: the launches that exist only because ASG composes two criterion calls (target sizes, the flagged-utterance
// launch, three axpy, the backward pass's position sort and partials fill) are folded into their neighbours, and with at most 64
// utterances the forward pass's four half scans are ONE launch on the caller's stream (criterion_asg.hip: no side stream, no
// events); everything else runs the composed calls.
struct AsgBuffers { int* ts; void* fcc; void* fac; float* dx2; float* dt2; float* loss2; };
class AsgSequence {
 public:
  ~AsgSequence() {
    if (side_) (void)hipStreamDestroy(side_);
    for (hipEvent_t e : {fork_, join_})
      if (e) (void)hipEventDestroy(e);
  }
  void forward(hipStream_t main, int B, int T, int N, int L, int mode, const float* em, const int* target, float* trans, float* loss,
               const AsgBuffers& w) {
    init();
    if (asg_forward_merged_ok(B, T, N, L)) {   // the four half scans of the pass in one launch: no side stream, no events
      w2lCheck(asg_forward_merged(B, T, N, L, mode, em, target, w.ts, trans, loss, w.loss2, w.fcc, w.fac, main), "asg forward");
      clearWs_ = w.fac;
      return;
    }
    if (fac_asg_fused_ok(B, T, N, L)) {
      FwdHook h{this, main, B, T, N, L, mode, em, target, trans, loss, &w};
      w2lCheck(fac_forward_asg(B, T, N, L, mode, em, target, w.ts, trans, w.loss2, loss, w.fac, main, fwdHook, &h), "asg forward");
      clearWs_ = w.fac;
      return;
    }
    clearWs_ = nullptr;
    w2lCheck(w2l_batch_target_size(B, L, T, target, w.ts, main), "asg target size");
    // the LONGER chain (FAC: label rows -> half scans -> finish) stays on the caller's stream, so that the fork / join events sit on
    // the chain that has the slack
    hipStream_t s2 = fork(main);
    w2lCheck(w2l_fcc_forward(B, T, N, mode, em, w.ts, trans, loss, w.fcc, s2), "fcc forward");
    w2lCheck(w2l_fac_forward(B, T, N, L, mode, em, target, w.ts, trans, w.loss2, w.fac, main), "fac forward");
    join(main);
    w2lCheck(w2l_axpy(loss, w.loss2, (size_t)B, -1.f, main), "asg loss");
  }
  void backward(hipStream_t main, int B, int T, int N, int L, const int* target, const float* gradLoss, float* dEm, float* trans,
                float* dTrans, const AsgBuffers& w) {
    init();
    if (fac_asg_fused_ok(B, T, N, L)) {
      const bool clear = clearWs_ == w.fac;   // forward's label-row launch cleared the transition-gradient partials of this workspace ...
      clearWs_ = nullptr;                     // ... and this pass uses them up (a second backward on the same forward fills them itself)
      // At these sizes FullConnectionCriterion's backward chain is the longer one (scan 167 us + 33 us of transition-gradient
      // launches against 150 + 42 us): it stays on the caller's stream and starts at once; ForceAlignmentCriterion's scan and scatter
      // go to the side stream, and the one combine launch behind the join sums both criteria's partials and subtracts.
      hipStream_t s2 = fork(main);
      int fccStride = 0;
      const float* fccPart = fcc_transgrad_partials(w.fcc, B, T, N, &fccStride);
      w2lCheck(fcc_backward_impl(B, T, N, trans, gradLoss, dEm, dTrans, w.fcc, main, fccPart != nullptr), "fcc backward");
      BwdHook h{this, main};
      w2lCheck(fac_backward_asg(B, T, N, L, target, w.ts, gradLoss, dEm, dTrans, w.dx2, w.fac, clear, fccPart, fccStride, s2, main, bwdHook, &h),
               "asg backward");
      return;
    }
    hipStream_t s2 = fork(main);
    w2lCheck(w2l_fcc_backward(B, T, N, trans, gradLoss, dEm, dTrans, w.fcc, s2), "fcc backward");
    w2lCheck(w2l_fac_backward(B, T, N, L, target, w.ts, gradLoss, w.dx2, w.dt2, w.fac, main), "fac backward");
    join(main);
    w2lCheck(w2l_axpy(dEm, w.dx2, (size_t)B * T * N, -1.f, main), "asg dx");
    w2lCheck(w2l_axpy(dTrans, w.dt2, (size_t)N * N, -1.f, main), "asg dtrans");
  }

 private:
  void init() {
    if (side_) return;
    hipCheck(hipStreamCreateWithFlags(&side_, hipStreamNonBlocking), "asg side stream");
    // (events between two streams of ONE device: no system-scope fence, i.e. no cache write-back towards the host, at the record)
    for (hipEvent_t* e : {&fork_, &join_})
      hipCheck(hipEventCreateWithFlags(e, hipEventDisableTiming | hipEventDisableSystemFence), "asg event");
  }
  hipStream_t fork(hipStream_t main) {  // side stream that has waited for everything enqueued on `main` so far
    hipCheck(hipEventRecord(fork_, main), "asg fork");
    hipCheck(hipStreamWaitEvent(side_, fork_, 0), "asg fork");
    return side_;
  }
  void join(hipStream_t main) {
    hipCheck(hipEventRecord(join_, side_), "asg join");
    hipCheck(hipStreamWaitEvent(main, join_, 0), "asg join");
  }
  struct FwdHook { AsgSequence* self; hipStream_t main; int B, T, N, L, mode; const float* em; const int* target; float* trans; float* loss; const AsgBuffers* w; };
  static void fwdHook(void* a, int what) {
    FwdHook& h = *(FwdHook*)a;
    if (what == ASG_TARGET_SIZES_QUEUED) {
      hipStream_t s2 = h.self->fork(h.main);
      w2lCheck(w2l_fcc_forward(h.B, h.T, h.N, h.mode, h.em, h.w->ts, h.trans, h.loss, h.w->fcc, s2), "fcc forward");
      hipCheck(hipEventRecord(h.self->join_, s2), "asg join");
    } else if (what == ASG_NEED_FCC_LOSS) {
      hipCheck(hipStreamWaitEvent(h.main, h.self->join_, 0), "asg join");
    }
  }
  struct BwdHook { AsgSequence* self; hipStream_t main; };
  static void bwdHook(void* a, int what) {
    BwdHook& h = *(BwdHook*)a;
    if (what == ASG_NEED_FCC_GRADS) h.self->join(h.main);
  }
  hipStream_t side_ = nullptr;
  hipEvent_t fork_ = nullptr, join_ = nullptr;
  const void* clearWs_ = nullptr;
};

class ASGLossImpl : public SequenceCriterion {
 public:
  ASGLossImpl(int N, int mode, double transdiag) : N_(N), mode_(mode), transdiag_(transdiag) {}
  std::string prettyString() const override { return "AutoSegmentationCriterion"; }
  size_t paramFloats() const override { return ((size_t)N_ * N_ + 3) / 4 * 4; }
  void initParams(float* host) const override {
    std::memset(host, 0, sizeof(float) * paramFloats());
    for (int i = 0; i < N_; ++i) host[(size_t)i * N_ + i] = (float)transdiag_;
  }
  struct Ws { AsgBuffers b; void* vit; };
  Ws carve(void* ws, int B, int T, int N, int L) const {
    char* p = (char*)ws;
    Ws w;
    w.b.ts = (int*)p; p += up(sizeof(int) * B);
    w.b.fcc = p; p += up(w2l_fcc_workspace_size(B, T, N));
    w.b.fac = p; p += up(w2l_fac_workspace_size(B, T, N, L));
    w.b.dx2 = (float*)p; p += up(sizeof(float) * (size_t)B * T * N);
    w.b.dt2 = (float*)p; p += up(sizeof(float) * (size_t)N * N);
    w.b.loss2 = (float*)p; p += up(sizeof(float) * B);
    w.vit = p;
    return w;
  }
  size_t workspaceBytes(int B, int T, int N, int L) const override {
    return up(sizeof(int) * B) + up(w2l_fcc_workspace_size(B, T, N)) + up(w2l_fac_workspace_size(B, T, N, L)) +
           up(sizeof(float) * (size_t)B * T * N) + up(sizeof(float) * (size_t)N * N) + up(sizeof(float) * B) +
           up(w2l_viterbi_workspace_size(B, T, N));
  }
  void forward(Ctx& c, int B, int T, int N, int L, const float* em, const int* target, float* loss, void* ws,
               float* trans) override {
    if (N != N_) throw std::invalid_argument("ASGLoss: N doesn't match with the letter size");
    seq_.forward(c.stream, B, T, N, L, mode_, em, target, trans, loss, carve(ws, B, T, N, L).b);
  }
  void backward(Ctx& c, int B, int T, int N, int L, const float*, const int* target, const float* gradLoss,
                float* dEm, void* ws, float* trans, float* dTrans) override {
    seq_.backward(c.stream, B, T, N, L, target, gradLoss, dEm, trans, dTrans, carve(ws, B, T, N, L).b);
  }
  void viterbiPath(Ctx& c, int B, int T, int N, const float* em, int* path, void* ws, float* trans) override {
    Ws w = carve(ws, B, T, N, 1);
    w2lCheck(w2l_viterbi_compute(B, T, N, em, trans, path, w.b.fcc, c.stream), "viterbi");
  }

 private:
  int N_, mode_;
  double transdiag_;
  AsgSequence seq_;
};

// LinearSegmentationCriterion: ASG on the linearly stretched target (first --linseg updates, Train.cpp:589-617).
// Uses the SAME parameter block as the ASG criterion it warms up (the caller passes the ASG transitions as
// critParams, the reference's `linseg->setParams(criterion->param(0), 0)`).
class LinSegCriterionImpl : public SequenceCriterion {
 public:
  LinSegCriterionImpl(int N, int mode) : N_(N), mode_(mode) {}
  std::string prettyString() const override { return "LinearSegmentationCriterion"; }
  size_t paramFloats() const override { return ((size_t)N_ * N_ + 3) / 4 * 4; }
  void initParams(float* host) const override { std::memset(host, 0, sizeof(float) * paramFloats()); }
  struct Ws { int* ts; int* lin; void* fcc; float* dx2; float* dt2; float* loss2; };
  Ws carve(void* ws, int B, int T, int N) const {
    char* p = (char*)ws;
    Ws w;
    w.ts = (int*)p; p += up(sizeof(int) * B);
    w.lin = (int*)p; p += up(sizeof(int) * (size_t)B * T);
    w.fcc = p; p += up(w2l_fcc_workspace_size(B, T, N));
    w.dx2 = (float*)p; p += up(sizeof(float) * (size_t)B * T * N);
    w.dt2 = (float*)p; p += up(sizeof(float) * (size_t)N * N);
    w.loss2 = (float*)p;
    return w;
  }
  size_t workspaceBytes(int B, int T, int N, int) const override {
    return up(sizeof(int) * B) + up(sizeof(int) * (size_t)B * T) + up(w2l_fcc_workspace_size(B, T, N)) +
           up(sizeof(float) * (size_t)B * T * N) + up(sizeof(float) * (size_t)N * N) + up(sizeof(float) * B) +
           up(w2l_viterbi_workspace_size(B, T, N));
  }
  void forward(Ctx& c, int B, int T, int N, int L, const float* em, const int* target, float* loss, void* ws,
               float* trans) override {
    if (N != N_) throw std::invalid_argument("LinSegCriterion: N doesn't match with the letter size");
    Ws w = carve(ws, B, T, N);
    w2lCheck(w2l_linear_target(B, L, T, target, w.lin, c.stream), "linear target");
    w2lCheck(w2l_batch_target_size(B, T, T, w.lin, w.ts, c.stream), "linseg target size");
    w2lCheck(w2l_fac_fullpath_forward(B, T, N, mode_, em, w.lin, trans, w.loss2, c.stream), "linseg path score");
    w2lCheck(w2l_fcc_forward(B, T, N, mode_, em, w.ts, trans, loss, w.fcc, c.stream), "fcc forward");
    w2lCheck(w2l_axpy(loss, w.loss2, (size_t)B, -1.f, c.stream), "linseg loss");
  }
  void backward(Ctx& c, int B, int T, int N, int, const float*, const int*, const float* gradLoss, float* dEm,
                void* ws, float* trans, float* dTrans) override {
    Ws w = carve(ws, B, T, N);
    w2lCheck(w2l_fac_fullpath_backward(B, T, N, mode_, w.lin, gradLoss, w.dx2, w.dt2, c.stream), "linseg path backward");
    w2lCheck(w2l_fcc_backward(B, T, N, trans, gradLoss, dEm, dTrans, w.fcc, c.stream), "fcc backward");
    w2lCheck(w2l_axpy(dEm, w.dx2, (size_t)B * T * N, -1.f, c.stream), "linseg dx");
    w2lCheck(w2l_axpy(dTrans, w.dt2, (size_t)N * N, -1.f, c.stream), "linseg dtrans");
  }
  void viterbiPath(Ctx& c, int B, int T, int N, const float* em, int* path, void* ws, float* trans) override {
    Ws w = carve(ws, B, T, N);
    w2lCheck(w2l_viterbi_compute(B, T, N, em, trans, path, w.fcc, c.stream), "viterbi");
  }

 private:
  int N_, mode_;
};

}  // namespace

std::shared_ptr<SequenceCriterion> makeLinSegCriterion(int N, int scaleMode) {
  return std::make_shared<LinSegCriterionImpl>(N, scaleMode);
}
std::shared_ptr<SequenceCriterion> makeCTCLoss(int scaleMode) { return std::make_shared<CTCLossImpl>(scaleMode); }
std::shared_ptr<SequenceCriterion> makeASGLoss(int N, int scaleMode, double transdiag) {
  return std::make_shared<ASGLossImpl>(N, scaleMode, transdiag);
}

}  // namespace w2l

// ---- ASGLoss in one call (include/w2l_hip.h): FullConnectionCriterion - ForceAlignmentCriterion on the caller's stream and a
// library-owned side stream per device, the sequence fl::pkg::speech::ASGLoss runs above
#define W2L_API extern "C" __attribute__((visibility("default")))
namespace {
w2l::AsgBuffers asg_carve(void* ws, int B, int T, int N, int L) {
  char* p = (char*)ws;
  w2l::AsgBuffers w;
  w.ts = (int*)p; p += w2l::up(sizeof(int) * B);
  w.loss2 = (float*)p; p += w2l::up(sizeof(float) * B);
  w.fcc = p; p += w2l::up(w2l_fcc_workspace_size(B, T, N));
  w.fac = p; p += w2l::up(w2l_fac_workspace_size(B, T, N, L));
  w.dx2 = (float*)p; p += w2l::up(sizeof(float) * (size_t)B * T * N);
  w.dt2 = (float*)p;
  return w;
}
std::mutex g_asgEnqueue;   // the device's sequence (its side stream, its events) is shared by every caller of w2l_asg_*: one enqueue at a time
w2l::AsgSequence* asg_sequence_of_device() {
  static std::mutex mu;
  static w2l::AsgSequence* seqs[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  std::lock_guard<std::mutex> g(mu);
  if (!seqs[dev]) seqs[dev] = new w2l::AsgSequence();   // lives as long as the process (its stream dies with the device)
  return seqs[dev];
}
template <class F>
int asg_guard(F&& f) {
  try { f(); return W2L_OK; }
  catch (const std::invalid_argument&) { return W2L_EINVAL; }
  catch (const std::exception&) { return W2L_EUNSUPPORTED; }
}
}  // namespace

W2L_API size_t w2l_asg_workspace_size(int B, int T, int N, int L) {
  if (B <= 0 || T <= 0 || N <= 0 || L <= 0) return 0;
  const size_t fcc = w2l_fcc_workspace_size(B, T, N), fac = w2l_fac_workspace_size(B, T, N, L);
  if (!fcc || !fac) return 0;
  return w2l::up(sizeof(int) * B) + w2l::up(sizeof(float) * B) + w2l::up(fcc) + w2l::up(fac) +
         w2l::up(sizeof(float) * (size_t)B * T * N) + w2l::up(sizeof(float) * (size_t)N * N);
}

W2L_API int w2l_asg_forward(int B, int T, int N, int L, int scaleMode, const float* input, const int* target, const float* trans,
                            float* loss, void* workspace, w2l_stream_t stream) {
  if (B <= 0 || T <= 0 || N <= 0 || L <= 0 || !input || !target || !trans || !loss || !workspace) return W2L_EINVAL;
  if (!w2l_asg_workspace_size(B, T, N, L)) return W2L_EUNSUPPORTED;
  w2l::AsgSequence* seq = asg_sequence_of_device();
  if (!seq) return W2L_EUNSUPPORTED;
  std::lock_guard<std::mutex> g(g_asgEnqueue);
  return asg_guard([&] {
    seq->forward((hipStream_t)stream, B, T, N, L, scaleMode, input, target, (float*)trans, loss, asg_carve(workspace, B, T, N, L));
  });
}

W2L_API int w2l_asg_backward(int B, int T, int N, int L, const int* target, const float* trans, const float* grad, float* inputGrad,
                             float* transGrad, void* workspace, w2l_stream_t stream) {
  if (B <= 0 || T <= 0 || N <= 0 || L <= 0 || !target || !trans || !grad || !inputGrad || !transGrad || !workspace) return W2L_EINVAL;
  if (!w2l_asg_workspace_size(B, T, N, L)) return W2L_EUNSUPPORTED;
  w2l::AsgSequence* seq = asg_sequence_of_device();
  if (!seq) return W2L_EUNSUPPORTED;
  std::lock_guard<std::mutex> g(g_asgEnqueue);
  return asg_guard([&] {
    seq->backward((hipStream_t)stream, B, T, N, L, target, grad, inputGrad, (float*)trans, transGrad, asg_carve(workspace, B, T, N, L));
  });
}
