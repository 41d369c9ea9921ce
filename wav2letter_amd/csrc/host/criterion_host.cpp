// criterion_host.cpp -- fl::pkg::speech::{CTCLoss, ASGLoss} over the kernel C ABI.
// Construction and use in the reference: recipes/slimIPL/src/Train.cpp:406-410 (ctor),
// :1675 (forward), :1720 (backward), :838 (viterbiPath).  ASGLoss = FullConnectionCriterion
// - ForceAlignmentCriterion sharing one N x N transition parameter initialised to
// transdiag * I (--transdiag, recipes/conv_glu/librispeech/train.cfg:25).
#include <cstring>

#include "w2l_host.hpp"

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
class ASGLossImpl : public SequenceCriterion {
 public:
  ASGLossImpl(int N, int mode, double transdiag) : N_(N), mode_(mode), transdiag_(transdiag) {}
  ~ASGLossImpl() override {
    if (side_) (void)hipStreamDestroy(side_);
    if (fork_) (void)hipEventDestroy(fork_);
    if (join_) (void)hipEventDestroy(join_);
  }
  hipStream_t fork(hipStream_t main) {  // side stream that has waited for everything enqueued on `main` so far
    if (!side_) {
      hipCheck(hipStreamCreateWithFlags(&side_, hipStreamNonBlocking), "asg side stream");
      hipCheck(hipEventCreateWithFlags(&fork_, hipEventDisableTiming), "asg event");
      hipCheck(hipEventCreateWithFlags(&join_, hipEventDisableTiming), "asg event");
    }
    hipCheck(hipEventRecord(fork_, main), "asg fork");
    hipCheck(hipStreamWaitEvent(side_, fork_, 0), "asg fork");
    return side_;
  }
  void join(hipStream_t main) {
    hipCheck(hipEventRecord(join_, side_), "asg join");
    hipCheck(hipStreamWaitEvent(main, join_, 0), "asg join");
  }
  std::string prettyString() const override { return "AutoSegmentationCriterion"; }
  size_t paramFloats() const override { return ((size_t)N_ * N_ + 3) / 4 * 4; }
  void initParams(float* host) const override {
    std::memset(host, 0, sizeof(float) * paramFloats());
    for (int i = 0; i < N_; ++i) host[(size_t)i * N_ + i] = (float)transdiag_;
  }
  struct Ws { int* ts; void* fcc; void* fac; float* dx2; float* dt2; float* loss2; void* vit; };
  Ws carve(void* ws, int B, int T, int N, int L) const {
    char* p = (char*)ws;
    Ws w;
    w.ts = (int*)p; p += up(sizeof(int) * B);
    w.fcc = p; p += up(w2l_fcc_workspace_size(B, T, N));
    w.fac = p; p += up(w2l_fac_workspace_size(B, T, N, L));
    w.dx2 = (float*)p; p += up(sizeof(float) * (size_t)B * T * N);
    w.dt2 = (float*)p; p += up(sizeof(float) * (size_t)N * N);
    w.loss2 = (float*)p; p += up(sizeof(float) * B);
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
    Ws w = carve(ws, B, T, N, L);
    w2lCheck(w2l_batch_target_size(B, L, T, target, w.ts, c.stream), "asg target size");
    // the two criteria side by side; the LONGER chain (FAC: label rows -> half scans -> finish, ~215 us at N = 30 against FCC's ~135)
    // stays on the caller's stream, so that the fork / join events sit on the chain that has the slack
    hipStream_t s2 = fork(c.stream);
    w2lCheck(w2l_fcc_forward(B, T, N, mode_, em, w.ts, trans, loss, w.fcc, s2), "fcc forward");
    w2lCheck(w2l_fac_forward(B, T, N, L, mode_, em, target, w.ts, trans, w.loss2, w.fac, c.stream), "fac forward");
    join(c.stream);
    w2lCheck(w2l_axpy(loss, w.loss2, (size_t)B, -1.f, c.stream), "asg loss");
  }
  void backward(Ctx& c, int B, int T, int N, int L, const float*, const int* target, const float* gradLoss,
                float* dEm, void* ws, float* trans, float* dTrans) override {
    Ws w = carve(ws, B, T, N, L);
    hipStream_t s2 = fork(c.stream);
    w2lCheck(w2l_fcc_backward(B, T, N, trans, gradLoss, dEm, dTrans, w.fcc, s2), "fcc backward");
    w2lCheck(w2l_fac_backward(B, T, N, L, target, w.ts, gradLoss, w.dx2, w.dt2, w.fac, c.stream), "fac backward");
    join(c.stream);
    w2lCheck(w2l_axpy(dEm, w.dx2, (size_t)B * T * N, -1.f, c.stream), "asg dx");
    w2lCheck(w2l_axpy(dTrans, w.dt2, (size_t)N * N, -1.f, c.stream), "asg dtrans");
  }
  void viterbiPath(Ctx& c, int B, int T, int N, const float* em, int* path, void* ws, float* trans) override {
    Ws w = carve(ws, B, T, N, 1);
    w2lCheck(w2l_viterbi_compute(B, T, N, em, trans, path, w.fcc, c.stream), "viterbi");
  }

 private:
  int N_, mode_;
  double transdiag_;
  hipStream_t side_ = nullptr;
  hipEvent_t fork_ = nullptr, join_ = nullptr;
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
