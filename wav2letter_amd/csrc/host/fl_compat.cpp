// fl_compat.cpp -- the fl:: surface of include/fl_compat/flashlight.h over the w2l:: host layer and the kernel C ABI.
// Reference shapes: see the header (CPCCriterion.h:30-50, 100h_supervised.cpp:84-87, Train.cpp:390-410, :1454-1804).
//
// Autograd here is Flashlight's tape in miniature: a Variable remembers its inputs and a gradFunc; backward() walks
// the DAG in reverse topological order.  Only two kinds of nodes exist on the hot path -- "criterion" and "planned
// network" -- so one loss.backward() is: criterion backward kernels -> dEmission -> the whole network's backward.
#include <dlfcn.h>
#include <time.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <sstream>
#include <unordered_map>
#include <unordered_set>

#include "../../../include/fl_compat/flashlight.h"
#include "w2l_host.hpp"

namespace {

hipStream_t g_stream = nullptr;
bool g_streamUser = false;

hipStream_t S() {
  if (!g_stream) w2l::hipCheck(hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking), "fl_compat stream");
  return g_stream;
}

std::shared_ptr<void> devAlloc(size_t bytes) {
  void* p = nullptr;
  w2l::hipCheck(hipMalloc(&p, bytes ? bytes : 4), "hipMalloc");
  return std::shared_ptr<void>(p, [](void* q) { (void)hipFree(q); });
}

}  // namespace

// ------------------------------------------------------------------------------------------------ af::
namespace af {

array::array(const dim4& dims, dtype ty) : dims_(dims), type_(ty) {
  owner_ = devAlloc((size_t)dims.elements() * 4);
  ptr_ = owner_.get();
}
array::array(const dim4& dims, const float* host) : array(dims, f32) {
  w2l::hipCheck(hipMemcpyAsync(ptr_, host, bytes(), hipMemcpyHostToDevice, S()), "array(host)");
  w2l::hipCheck(hipStreamSynchronize(S()), "array(host)");
}
array::array(const dim4& dims, const int* host) : array(dims, s32) {
  w2l::hipCheck(hipMemcpyAsync(ptr_, host, bytes(), hipMemcpyHostToDevice, S()), "array(host)");
  w2l::hipCheck(hipStreamSynchronize(S()), "array(host)");
}
array array::wrap(void* dev, const dim4& dims, dtype ty, std::shared_ptr<void> owner) {
  array a;
  a.ptr_ = dev; a.dims_ = dims; a.type_ = ty; a.owner_ = std::move(owner);
  return a;
}
void array::hostCopy(void* out) const {
  if (!elements()) return;
  w2l::hipCheck(hipMemcpyAsync(out, ptr_, bytes(), hipMemcpyDeviceToHost, S()), "array::host");
  w2l::hipCheck(hipStreamSynchronize(S()), "array::host");
}
void array::firstElement(void* out) const {
  w2l::hipCheck(hipMemcpyAsync(out, ptr_, 4, hipMemcpyDeviceToHost, S()), "array::scalar");
  w2l::hipCheck(hipStreamSynchronize(S()), "array::scalar");
}
array array::copy() const {
  array a(dims_, type_);
  if (elements()) w2l::hipCheck(hipMemcpyAsync(a.ptr_, ptr_, bytes(), hipMemcpyDeviceToDevice, S()), "array::copy");
  return a;
}
array constant(double v, const dim4& dims, dtype ty) {
  array a(dims, ty);
  if (ty == f32) w2l::w2lCheck(w2l_fill(a.device<float>(), (size_t)a.elements(), (float)v, S()), "constant");
  else {
    std::vector<int> h((size_t)a.elements(), (int)v);
    w2l::hipCheck(hipMemcpyAsync(a.device<int>(), h.data(), a.bytes(), hipMemcpyHostToDevice, S()), "constant");
    w2l::hipCheck(hipStreamSynchronize(S()), "constant");
  }
  return a;
}
void sync() { w2l::hipCheck(hipStreamSynchronize(S()), "af::sync"); }

}  // namespace af

// ------------------------------------------------------------------------------------------------ fl::
namespace fl {

void* currentStream() { return (void*)S(); }
void setCurrentStream(void* stream) { g_stream = (hipStream_t)stream; g_streamUser = true; }

struct VariableAccess {
  static Variable::Shared* sh(const Variable& v) { return v.s_.get(); }
};

Variable::Variable(af::array data, std::vector<Variable> inputs, GradFunc gradFunc) : s_(std::make_shared<Shared>()) {
  s_->data = std::move(data);
  s_->calcGrad = std::any_of(inputs.begin(), inputs.end(), [](const Variable& v) { return v.isCalcGrad(); });
  if (s_->calcGrad) {
    s_->inputs = std::move(inputs);
    s_->gradFunc = std::move(gradFunc);
  }
}

Variable& Variable::grad() const {
  if (!s_->calcGrad) throw std::logic_error("gradient calculation disabled for this Variable");
  if (!s_->grad) throw std::logic_error("gradient not calculated yet for this Variable");
  return *s_->grad;
}

void Variable::addGrad(const Variable& g) {
  if (!s_->calcGrad) return;
  if (g.elements() != elements()) throw std::invalid_argument("addGrad: gradient and variable sizes differ");
  if (!s_->grad) {
    s_->grad.reset(new Variable(g.array(), false));   // shares storage, like Flashlight
  } else {
    // accumulate: this happens for a parameter used twice (LinSeg + ASG transitions), never for activations here
    af::array sum = s_->grad->array().copy();
    w2l::w2lCheck(w2l_axpy(sum.device<float>(), g.array().device<float>(), (size_t)elements(), 1.f, S()), "addGrad");
    s_->grad.reset(new Variable(sum, false));
  }
}

void Variable::backward(const Variable& grad, bool retainGraph) {
  addGrad(grad);
  // reverse topological order from this node
  std::vector<Variable> order;
  std::unordered_set<Shared*> seen;
  std::function<void(const Variable&)> dfs = [&](const Variable& v) {
    Shared* p = VariableAccess::sh(v);
    if (seen.count(p)) return;
    seen.insert(p);
    for (const auto& in : p->inputs) dfs(in);
    order.push_back(v);
  };
  dfs(*this);
  for (auto it = order.rbegin(); it != order.rend(); ++it) {
    Shared* p = VariableAccess::sh(*it);
    if (p->gradFunc && p->grad) p->gradFunc(p->inputs, *p->grad);
    if (!retainGraph && p != s_.get() && !p->inputs.empty()) p->grad.reset();  // interior activations only
  }
  if (!retainGraph)
    for (auto& v : order) { Shared* p = VariableAccess::sh(v); p->inputs.clear(); p->gradFunc = nullptr; }
}

void Variable::backward(bool retainGraph) {
  backward(Variable(af::constant(1.0, dims(), af::f32), false), retainGraph);
}

Variable operator*(const Variable& v, double s) {
  af::array out = af::constant(0.0, v.dims(), af::f32);
  if (v.type() != af::f32) throw std::invalid_argument("Variable * scalar: f32 only");
  w2l::w2lCheck(w2l_axpy(out.device<float>(), v.array().device<float>(), (size_t)v.elements(), (float)s, S()), "scale");
  return Variable(out, false);
}
Variable operator/(const Variable& v, double s) { return v * (1.0 / s); }

void Module::setParams(const Variable& var, int position) { params_.at(position) = var; }

void Container::add(std::shared_ptr<Module> m) {
  if (!m) throw std::invalid_argument("can't add null Module to Container");
  const int child = (int)modules_.size();
  modules_.push_back(m);
  int k = 0;
  for (auto& p : m->params()) {
    params_.push_back(p);
    childOfParam_.push_back(child);
    childIndexOfParam_.push_back(k++);
  }
}
void Container::setParams(const Variable& var, int position) {
  Module::setParams(var, position);
  if (position < (int)childOfParam_.size() && childOfParam_[position] >= 0)
    modules_[childOfParam_[position]]->setParams(var, childIndexOfParam_[position]);
}

// ---- fl::Sequential of layer objects: the layers' arch lines, planned as one pipeline on first use
void Sequential::add(std::shared_ptr<Module> m) {
  if (!m) throw std::invalid_argument("can't add null Module to Sequential");
  if (auto* l = dynamic_cast<ArchLayer*>(m.get())) {
    if (planned_) throw std::logic_error("fl_compat: this fl::Sequential has been planned (params() / forward() was called): add every layer first");
    if (!modules_.empty() && archText_.empty()) throw std::logic_error("fl_compat: layer objects and other modules do not mix in one fl::Sequential");
    archText_ += l->archLine() + "\n";
    modules_.push_back(std::move(m));       // (listed for modules() / prettyString; the parameters appear when the pipeline is planned)
    return;
  }
  if (!archText_.empty()) throw std::logic_error("fl_compat: layer objects and other modules do not mix in one fl::Sequential");
  Container::add(std::move(m));
}
void Sequential::materialize() {
  if (planned_ || archText_.empty()) return;
  // Feature / label counts are not arguments of a layer list.  The label count is the last layer's output (taken after
  // planning); the feature count is needed to plan at all (the dry plan fixes the Linear row permutations), and a model that
  // starts the way the reference's do -- fl::View(af::dim4(-1, a, b, 0)) over the (T, NFEAT, 1, B) input -- says it: NFEAT = a b
  long d[4] = {0, 0, 0, 0};
  int nFeat = inputFeatures_;
  if (nFeat <= 0 && sscanf(archText_.c_str(), "V %ld %ld %ld %ld", &d[0], &d[1], &d[2], &d[3]) == 4 && d[0] == -1 && d[3] == 0 && d[1] > 0 && d[2] > 0)
    nFeat = (int)(d[1] * d[2]);
  if (nFeat <= 0)
    throw std::logic_error("fl_compat: an fl::Sequential of layer objects that does not start with fl::View(af::dim4(-1, a, b, 0)) needs "
                           "setInputFeatures(NFEAT) before params() / forward()");
  planned_ = pkg::speech::buildSequentialModuleFromText(archText_, nFeat, -1);
  params_ = planned_->params();
  childOfParam_.assign(params_.size(), -1);
  childIndexOfParam_.assign(params_.size(), 0);
  if (!train_) planned_->eval();
}
std::shared_ptr<Sequential> Sequential::planned() const {
  const_cast<Sequential*>(this)->materialize();
  return planned_;
}
std::vector<Variable> Sequential::params() const {
  const_cast<Sequential*>(this)->materialize();
  return params_;
}
void Sequential::setParams(const Variable& var, int position) {
  materialize();
  if (planned_) { planned_->setParams(var, position); params_ = planned_->params(); return; }
  Container::setParams(var, position);
}
void Sequential::train() { Container::train(); if (planned_) planned_->train(); }
void Sequential::eval() { Container::eval(); if (planned_) planned_->eval(); }
std::vector<Variable> Sequential::forward(const std::vector<Variable>& inputs) {
  materialize();
  if (planned_) return planned_->forward(inputs);
  std::vector<Variable> cur = inputs;
  for (auto& m : modules_) cur = m->forward(cur);
  return cur;
}
std::string Sequential::prettyString() const {
  if (planned_) return planned_->prettyString();
  std::ostringstream ss;
  ss << "Sequential [input";
  for (size_t i = 0; i < modules_.size(); ++i) ss << " -> (" << i << ")";
  ss << " -> output]";
  for (size_t i = 0; i < modules_.size(); ++i) ss << "\n\t(" << i << "): " << modules_[i]->prettyString();
  return ss.str();
}

// ------------------------------------------------------------------------------------------------ SpecAugment
SpecAugment::SpecAugment(int tWarpW, int fMaskF, int nFMask, int tMaskT, float tMaskP, int nTMask)
    : fMaskF_(fMaskF), nFMask_(nFMask), tMaskT_(tMaskT), nTMask_(nTMask), tMaskP_(tMaskP) {
  (void)tWarpW;
}
std::vector<Variable> SpecAugment::forward(const std::vector<Variable>& inputs) {
  if (inputs.empty()) throw std::invalid_argument("SpecAugment: no input");
  const Variable& in = inputs[0];
  if (!train_) return {in};
  if (in.type() != af::f32) throw std::invalid_argument("SpecAugment: features must be f32");
  // (T, F, 1, B) is time-fastest [B][F][T]; the masking kernel works on frames [B][T][F]
  const int T = (int)in.dims(0), F = (int)(in.dims(1) * in.dims(2)), B = (int)in.dims(3);
  af::array frames(af::dim4(F, T, B)), out(in.dims());
  w2l::w2lCheck(w2l_transpose(in.array().device<float>(), frames.device<float>(), B, F, T, S()), "saug transpose");
  w2l::w2lCheck(w2l_specaugment_inplace(frames.device<float>(), B, T, F, fMaskF_, nFMask_, tMaskT_, tMaskP_, nTMask_,
                                        0x9E3779B9u * (++calls_), S()), "saug");
  w2l::w2lCheck(w2l_transpose(frames.device<float>(), out.device<float>(), B, T, F, S()), "saug transpose back");
  return {Variable(out, false)};
}
std::string SpecAugment::prettyString() const {
  std::ostringstream ss;
  ss << "SpecAugment ( W: 0, F: " << fMaskF_ << ", mF: " << nFMask_ << ", T: " << tMaskT_ << ", p: " << tMaskP_ << ", mT: " << nTMask_ << " )";
  return ss.str();
}

// ------------------------------------------------------------------------------------------------ optimizers
namespace {
// Gradient (or parameter) storage of a parameter list as contiguous runs: the parameters of a planned network are slices of its
// flat arenas, and when ALL of them are in the list (at their own offsets) they collapse to the arena -- one launch instead of one
// per parameter (4-float alignment gaps between the slices are zero: nothing ever writes them).  Defined below, next to the arenas.
struct FlatRun { float* p; size_t n; };
std::vector<FlatRun> gradientRuns(const std::vector<Variable>& params);
bool flatParameterArena(const std::vector<Variable>& params, float*& base, size_t& floats, std::vector<size_t>& offsets);
bool gradientArenaOf(const float* g, float*& base);
}  // namespace

void scaleGradients(const std::vector<Variable>& params, double s) {
  if (s == 1.0) return;
  for (const FlatRun& r : gradientRuns(params)) w2l::w2lCheck(w2l_axpy(r.p, r.p, r.n, (float)(s - 1.0), S()), "scaleGradients");
}

void FirstOrderOptimizer::flatStateViews(int slot, std::vector<af::array>& views) {
  std::shared_ptr<void> owner = devAlloc(flatFloats_ * sizeof(float));
  w2l::hipCheck(hipMemsetAsync(owner.get(), 0, flatFloats_ * sizeof(float), S()), "optimizer state");
  flatState_[slot] = (float*)owner.get();
  for (size_t i = 0; i < parameters_.size(); ++i)
    views.push_back(af::array::wrap(flatState_[slot] + flatOffsets_[i], parameters_[i].dims(), af::f32, owner));
}
// every parameter, gradient and state array still at its arena offset?  then the whole network is one launch
bool FirstOrderOptimizer::flatNow(float*& gbase, const std::vector<af::array>* s0, const std::vector<af::array>* s1) const {
  if (!flatParams_ || parameters_.empty() || !parameters_[0].isGradAvailable()) return false;
  if (!gradientArenaOf(parameters_[0].grad().array().device<float>(), gbase)) return false;
  for (size_t i = 0; i < parameters_.size(); ++i) {
    const auto& p = parameters_[i];
    if (!p.isGradAvailable() || p.array().device<float>() != flatParams_ + flatOffsets_[i] ||
        p.grad().array().device<float>() != gbase + flatOffsets_[i])
      return false;
    if (s0 && (*s0)[i].device<float>() != flatState_[0] + flatOffsets_[i]) return false;
    if (s1 && (*s1)[i].device<float>() != flatState_[1] + flatOffsets_[i]) return false;
  }
  return true;
}

SGDOptimizer::SGDOptimizer(const std::vector<Variable>& params, double lr, double momentum, double weightDecay, bool useNesterov)
    : FirstOrderOptimizer(params, lr), mu_(momentum), wd_(weightDecay), nesterov_(useNesterov) {
  if (wd_ != 0 || nesterov_) throw std::invalid_argument("fl_compat SGDOptimizer: weight decay / Nesterov are not on the hot path of the recipes");
  if (flatParameterArena(parameters_, flatParams_, flatFloats_, flatOffsets_)) {
    if (mu_ != 0) flatStateViews(0, velocities_);   // one flat velocity buffer, the per-parameter arrays (checkpoints read them) are views of it
    return;
  }
  flatParams_ = nullptr;
  if (mu_ != 0)
    for (auto& p : parameters_) velocities_.push_back(af::constant(0.0, p.dims(), af::f32));
}
void SGDOptimizer::step() {
  float* gbase = nullptr;
  if (flatNow(gbase, mu_ != 0 ? &velocities_ : nullptr, nullptr)) {
    w2l::w2lCheck(w2l_sgd_step(flatParams_, gbase, mu_ != 0 ? flatState_[0] : nullptr, flatFloats_, (float)lr_, (float)mu_, 1.f, 0.f, nullptr, S()), "sgd");
    return;
  }
  for (size_t i = 0; i < parameters_.size(); ++i) {
    auto& p = parameters_[i];
    if (!p.isGradAvailable()) continue;
    w2l::w2lCheck(w2l_sgd_step(p.array().device<float>(), p.grad().array().device<float>(),
                               mu_ != 0 ? velocities_[i].device<float>() : nullptr, (size_t)p.elements(), (float)lr_, (float)mu_,
                               1.f, 0.f, nullptr, S()), "sgd");
  }
}
std::string SGDOptimizer::prettyString() const {
  std::ostringstream ss;
  ss << "SGD";
  if (mu_ != 0) ss << " (momentum=" << mu_ << ")";
  return ss.str();
}

AdagradOptimizer::AdagradOptimizer(const std::vector<Variable>& params, double lr, double eps, double weightDecay)
    : FirstOrderOptimizer(params, lr), eps_(eps) {
  if (weightDecay != 0) throw std::invalid_argument("fl_compat AdagradOptimizer: weight decay is not on the hot path of the recipes");
  if (flatParameterArena(parameters_, flatParams_, flatFloats_, flatOffsets_)) { flatStateViews(0, variance_); return; }
  flatParams_ = nullptr;
  for (auto& p : parameters_) variance_.push_back(af::constant(0.0, p.dims(), af::f32));
}
void AdagradOptimizer::step() {
  float* gbase = nullptr;
  if (flatNow(gbase, &variance_, nullptr)) {
    w2l::w2lCheck(w2l_adagrad_step_guarded(flatParams_, gbase, flatState_[0], flatFloats_, (float)lr_, (float)eps_, 1.f, 0.f, nullptr, S()), "adagrad");
    return;
  }
  for (size_t i = 0; i < parameters_.size(); ++i) {
    auto& p = parameters_[i];
    if (!p.isGradAvailable()) continue;
    w2l::w2lCheck(w2l_adagrad_step_guarded(p.array().device<float>(), p.grad().array().device<float>(), variance_[i].device<float>(),
                                           (size_t)p.elements(), (float)lr_, (float)eps_, 1.f, 0.f, nullptr, S()), "adagrad");
  }
}
std::string AdagradOptimizer::prettyString() const {
  std::ostringstream ss;
  ss << "Adagrad (epsilon=" << eps_ << ")";
  return ss.str();
}

AdadeltaOptimizer::AdadeltaOptimizer(const std::vector<Variable>& params, double lr, double rho, double eps, double weightDecay)
    : FirstOrderOptimizer(params, lr), rho_(rho), eps_(eps) {
  if (weightDecay != 0) throw std::invalid_argument("fl_compat AdadeltaOptimizer: weight decay is not on the hot path of the recipes");
  if (flatParameterArena(parameters_, flatParams_, flatFloats_, flatOffsets_)) {
    flatStateViews(0, accGrad_);
    flatStateViews(1, accDelta_);
    return;
  }
  flatParams_ = nullptr;
  for (auto& p : parameters_) {
    accGrad_.push_back(af::constant(0.0, p.dims(), af::f32));
    accDelta_.push_back(af::constant(0.0, p.dims(), af::f32));
  }
}
void AdadeltaOptimizer::step() {
  float* gbase = nullptr;
  if (flatNow(gbase, &accGrad_, &accDelta_)) {
    w2l::w2lCheck(w2l_adadelta_step_guarded(flatParams_, gbase, flatState_[0], flatState_[1], flatFloats_, (float)lr_, (float)rho_, (float)eps_, 1.f,
                                            0.f, nullptr, S()), "adadelta");
    return;
  }
  for (size_t i = 0; i < parameters_.size(); ++i) {
    auto& p = parameters_[i];
    if (!p.isGradAvailable()) continue;
    w2l::w2lCheck(w2l_adadelta_step_guarded(p.array().device<float>(), p.grad().array().device<float>(), accGrad_[i].device<float>(),
                                            accDelta_[i].device<float>(), (size_t)p.elements(), (float)lr_, (float)rho_, (float)eps_, 1.f,
                                            0.f, nullptr, S()), "adadelta");
  }
}
std::string AdadeltaOptimizer::prettyString() const {
  std::ostringstream ss;
  ss << "Adadelta (rho=" << rho_ << ") (epsilon=" << eps_ << ")";
  return ss.str();
}

double clipGradNorm(const std::vector<Variable>& params, double maxNorm) {
  static std::shared_ptr<void> acc = devAlloc(sizeof(double));
  double* d = (double*)acc.get();
  bool first = true;
  const std::vector<FlatRun> runs = gradientRuns(params);
  for (const FlatRun& r : runs) {
    w2l::w2lCheck(w2l_sumsq(r.p, r.n, d, first ? 1 : 0, S()), "clipGradNorm");
    first = false;
  }
  if (first) return 0.0;
  double ss = 0;
  w2l::hipCheck(hipMemcpyAsync(&ss, d, sizeof(double), hipMemcpyDeviceToHost, S()), "clipGradNorm");
  w2l::hipCheck(hipStreamSynchronize(S()), "clipGradNorm");
  const double norm = std::sqrt(ss);
  const double scale = maxNorm / (norm + 1e-6);
  if (scale < 1.0)
    for (const FlatRun& r : runs) w2l::w2lCheck(w2l_axpy(r.p, r.p, r.n, (float)(scale - 1.0), S()), "clipGradNorm");
  return norm;
}

// ------------------------------------------------------------------------------------------------ data parallelism
// RCCL through dlopen: the types come from <rccl/rccl.h>, the entry points are resolved when initDistributed runs, so
// neither the library nor a single-GPU process depends on librccl.so (a Python process that already carries torch's
// RCCL gets that copy: RTLD_NOLOAD first).
}  // namespace fl
#include <rccl/rccl.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <atomic>
#include <fstream>
namespace fl {
namespace {
struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclComm_t comm = nullptr;
  int rank = 0, size = 1;
  struct ShmColl* shm = nullptr;   // the host-memory test collective (initDistributed("shm:<file>")): no RCCL involved
  bool on() const { return comm != nullptr || shm != nullptr; }
};
Rccl& rccl() { static Rccl r; return r; }

// ---- host-memory collective for TESTS ("host-memory fake collective"): N processes that share ONE GPU -- RCCL
// refuses two ranks on one device -- reduce through a file in /dev/shm.  Selected explicitly by a rendezvous path of the form
// shm:<file>; same call sites, same stream semantics as the RCCL path (the collective is ordered behind everything enqueued on
// its stream before it, and everything enqueued after it sees the result): the stream is drained, the slice goes to the
// rank's slot, all slots are summed IN RANK ORDER by every rank (bit-identical results on all ranks), and the sum is copied
// back.  Not a performance path: it exists so that allReduceParameters, the bucketed event-gated CoalescingReducer, the batch
// size reduce and the barrier of the data-parallel Train run with world_size > 1 on a one-GPU box.
struct ShmColl {
  struct Header { std::atomic<unsigned long long> magic; std::atomic<unsigned> arrived, generation; };
  static constexpr unsigned long long kMagic = 0x77326c73686d3031ull;   // "w2lshm01"
  static constexpr size_t kChunk = (size_t)4 << 20;                       // floats per rank slot (16 MiB)
  Header* hdr = nullptr;
  float* slots = nullptr;
  std::vector<float> acc;
  int rank = 0, size = 1;
  void barrier() {
    const unsigned gen = hdr->generation.load(std::memory_order_acquire);
    if (hdr->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == (unsigned)size) {
      hdr->arrived.store(0, std::memory_order_relaxed);
      hdr->generation.store(gen + 1, std::memory_order_release);
      return;
    }
    for (long spins = 0; hdr->generation.load(std::memory_order_acquire) == gen; ++spins) {
      if (spins > 2400000) throw std::runtime_error("host-memory collective: a rank did not arrive within two minutes");
      usleep(50);
    }
  }
  void allReduce(float* p, size_t n, hipStream_t s) {
    w2l::hipCheck(hipStreamSynchronize(s), "shm collective: drain");
    acc.resize(std::min(n, kChunk));
    for (size_t off = 0; off < n; off += kChunk) {
      const size_t m = std::min(kChunk, n - off);
      w2l::hipCheck(hipMemcpy(slots + (size_t)rank * kChunk, p + off, m * sizeof(float), hipMemcpyDeviceToHost), "shm collective: out");
      barrier();
      for (size_t i = 0; i < m; ++i) acc[i] = slots[i];
      for (int r = 1; r < size; ++r) {
        const float* q = slots + (size_t)r * kChunk;
        for (size_t i = 0; i < m; ++i) acc[i] += q[i];
      }
      barrier();   // every rank has read every slot: they may be overwritten
      w2l::hipCheck(hipMemcpyAsync(p + off, acc.data(), m * sizeof(float), hipMemcpyHostToDevice, s), "shm collective: in");
      w2l::hipCheck(hipStreamSynchronize(s), "shm collective: in");
    }
  }
};
ShmColl* shmOpen(const std::string& file, int rank, int size) {
  const size_t bytes = 4096 + (size_t)size * ShmColl::kChunk * sizeof(float);
  int fd = -1;
  if (rank == 0) {
    (void)unlink(file.c_str());
    fd = open(file.c_str(), O_RDWR | O_CREAT | O_EXCL, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)bytes) != 0) throw std::runtime_error("host-memory collective: cannot create " + file);
  } else {
    for (int tries = 0; tries < 1200 && fd < 0; ++tries) {   // up to two minutes
      fd = open(file.c_str(), O_RDWR);
      struct stat st;
      if (fd >= 0 && (fstat(fd, &st) != 0 || (size_t)st.st_size < bytes)) { close(fd); fd = -1; }
      if (fd < 0) usleep(100000);
    }
    if (fd < 0) throw std::runtime_error("host-memory collective: " + file + " did not appear");
  }
  void* base = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (base == MAP_FAILED) throw std::runtime_error("host-memory collective: mmap failed");
  auto* c = new ShmColl;
  c->hdr = (ShmColl::Header*)base;
  c->slots = (float*)((char*)base + 4096);
  c->rank = rank; c->size = size;
  if (rank == 0) {
    c->hdr->arrived.store(0); c->hdr->generation.store(0);
    c->hdr->magic.store(ShmColl::kMagic, std::memory_order_release);
  } else {
    for (long spins = 0; c->hdr->magic.load(std::memory_order_acquire) != ShmColl::kMagic; ++spins) {
      if (spins > 1200) throw std::runtime_error("host-memory collective: rank 0 never initialised " + file);
      usleep(100000);
    }
  }
  c->barrier();
  if (rank == 0) (void)unlink(file.c_str());   // everybody has it mapped: nothing stale survives the run
  return c;
}
void ncclCheck(ncclResult_t st, const char* what) {
  if (st != ncclSuccess) throw std::runtime_error(std::string(what) + ": " + (rccl().GetErrorString ? rccl().GetErrorString(st) : "RCCL error"));
}
template <class F> void bindSym(void* lib, const char* name, F& f) {
  f = (F)dlsym(lib, name);
  if (!f) throw std::runtime_error(std::string("librccl.so lacks ") + name);
}
// one all-reduce(sum) of p[0..n) ordered on stream s, through whichever backend is up
void collectiveSum(float* p, size_t n, hipStream_t s) {
  Rccl& r = rccl();
  if (r.shm) { r.shm->allReduce(p, n, s); return; }
  ncclCheck(r.AllReduce(p, p, n, ncclFloat32, ncclSum, r.comm, s), "ncclAllReduce");
}
void allReduceRaw(float* p, size_t n, double scale) {
  Rccl& r = rccl();
  if (!r.on() || !n) return;   // (a communicator of ONE rank still goes through RCCL: the plumbing is what a 1-GPU box can test)
  collectiveSum(p, n, (hipStream_t)S());
  if (scale != 1.0) w2l::w2lCheck(w2l_axpy(p, p, n, (float)(scale - 1.0), S()), "allReduce scale");
}
}  // namespace

int getWorldRank() { return rccl().rank; }
int getWorldSize() { return rccl().size; }
void allReduce(af::array& arr, double scale) {
  if (arr.type() != af::f32) throw std::invalid_argument("allReduce: f32 arrays only");
  allReduceRaw(arr.device<float>(), (size_t)arr.elements(), scale);
}
void allReduce(Variable& var, double scale) { allReduce(var.array(), scale); }
void allReduceParameters(const std::shared_ptr<const Module>& module) {
  if (!rccl().on()) return;
  for (auto& p : module->params()) allReduceRaw(p.array().device<float>(), (size_t)p.elements(), 1.0 / rccl().size);
}
void barrier() {
  if (!rccl().on()) return;
  static std::shared_ptr<void> one = devAlloc(sizeof(float));
  w2l::hipCheck(hipMemsetAsync(one.get(), 0, sizeof(float), (hipStream_t)S()), "barrier");
  allReduceRaw((float*)one.get(), 1, 1.0);
  w2l::hipCheck(hipStreamSynchronize((hipStream_t)S()), "barrier");
}

// ---- the planned networks' gradient arenas, as the reducer sees them (defined next to PlannedNet further down)
namespace {
struct ArenaInfo { float* base = nullptr; size_t floats = 0; void* net = nullptr; };
bool findArena(const float* p, ArenaInfo& out);
std::vector<size_t> arenaParamOffsets(void* net);
const std::vector<size_t>& arenaBucketOffsets(void* net);
hipEvent_t arenaBucketEvent(void* net, size_t k);
void arenaInstallBuckets(void* net, std::vector<size_t> offs);

// cut [0, total) into <= nBuckets pieces of roughly equal size at parameter boundaries (wav2letter_amd/parallel.bucket_offsets)
std::vector<size_t> cutBuckets(const std::vector<size_t>& bounds, size_t total, int nBuckets) {
  std::vector<size_t> out{0};
  for (int k = 1; k < nBuckets; ++k) {
    const size_t want = total * (size_t)k / (size_t)nBuckets;
    size_t best = 0, bestD = (size_t)-1;
    for (size_t b : bounds) {
      const size_t d = b > want ? b - want : want - b;
      if (d < bestD) { bestD = d; best = b; }
    }
    if (best > out.back()) out.push_back(best);
  }
  return out;
}
hipStream_t commStream() {
  static hipStream_t s = nullptr;
  if (!s) w2l::hipCheck(hipStreamCreateWithFlags(&s, hipStreamNonBlocking), "reducer stream");
  return s;
}
}  // namespace

CoalescingReducer::CoalescingReducer(double scale, bool async, bool contiguous) : scale_(scale), async_(async), contiguous_(contiguous) {}
CoalescingReducer::~CoalescingReducer() {}
void CoalescingReducer::add(Variable& var) {
  if (var.type() != af::f32) throw std::invalid_argument("CoalescingReducer: f32 gradients only");
  float* p = var.array().device<float>();
  const size_t n = (size_t)var.elements();
  if (!n) return;
  if (!spans_.empty() && spans_.back().ptr + spans_.back().n == p) spans_.back().n += n;   // the next piece of the same buffer
  else spans_.push_back({p, n});
}
// Gradients that are views of a planned network's flat gradient arena are reduced as that ARENA (its alignment padding
// included: a handful of floats), in a few large buckets: from the second step on, backward() has recorded one event per
// bucket on the compute stream as soon as every gradient at that offset or beyond was final (w2l::Sequential::backward),
// and finalize() -- called right after loss.backward() has been ENQUEUED -- walks the buckets last to first, makes the side
// stream wait for the bucket's event and issues the ncclAllReduce there: the last layers' gradients cross xGMI while the
// GPU still computes the first layers' backward (the reference's CoalescingReducer overlaps the same way, bucket by
// bucket, recipes/slimIPL/src/Train.cpp:1721-1735).  The first step (no events yet) reduces the arena in one piece and
// installs the buckets.  Everything else (the criterion's transitions) goes through one collective per buffer.
void CoalescingReducer::finalize() {
  lastCollectives_ = lastOverlapped_ = 0;
  std::vector<ArenaInfo> arenas;
  for (auto& sp : spans_) {
    ArenaInfo ai;
    if (rccl().on() && findArena(sp.ptr, ai)) {
      bool seen = false;
      for (auto& a : arenas) seen = seen || a.net == ai.net;
      if (!seen) arenas.push_back(ai);
      continue;
    }
    allReduceRaw(sp.ptr, sp.n, scale_);
    if (rccl().on()) ++lastCollectives_;
  }
  spans_.clear();
  Rccl& r = rccl();
  for (auto& ai : arenas) {
    const std::vector<size_t>& offs = arenaBucketOffsets(ai.net);
    if (offs.empty() || scale_ != 1.0) {
      allReduceRaw(ai.base, ai.floats, scale_);
      ++lastCollectives_;
      if (offs.empty()) arenaInstallBuckets(ai.net, cutBuckets(arenaParamOffsets(ai.net), ai.floats, 4));
      continue;
    }
    hipStream_t cs = commStream();
    for (size_t k = offs.size(); k-- > 0;) {
      const size_t lo = offs[k], hi = k + 1 < offs.size() ? offs[k + 1] : ai.floats;
      w2l::hipCheck(hipStreamWaitEvent(cs, arenaBucketEvent(ai.net, k), 0), "bucket wait");
      collectiveSum(ai.base + lo, hi - lo, cs);
      ++lastCollectives_;
      ++lastOverlapped_;
    }
    static hipEvent_t done = nullptr;
    if (!done) w2l::hipCheck(hipEventCreateWithFlags(&done, hipEventDisableTiming), "reducer event");
    w2l::hipCheck(hipEventRecord(done, cs), "reducer event");
    w2l::hipCheck(hipStreamWaitEvent(S(), done, 0), "reducer join");   // the optimizer step waits for the reduced gradients
  }
}

namespace pkg {
namespace runtime {
void initDistributed(int worldRank, int worldSize, int maxDevicesPerNode, const std::string& rndvFilepath) {
  Rccl& r = rccl();
  if (r.on()) throw std::runtime_error("initDistributed called twice");
  if (worldSize < 1 || worldRank < 0 || worldRank >= worldSize) throw std::invalid_argument("initDistributed: bad rank / world size");
  int ndev = 0;
  w2l::hipCheck(hipGetDeviceCount(&ndev), "hipGetDeviceCount");
  const int perNode = std::max(1, std::min(maxDevicesPerNode > 0 ? maxDevicesPerNode : ndev, ndev));
  w2l::hipCheck(hipSetDevice(worldRank % perNode), "hipSetDevice");
  if (rndvFilepath.rfind("shm:", 0) == 0) {   // the host-memory test collective: ranks may share a device
    r.shm = shmOpen(rndvFilepath.substr(4), worldRank, worldSize);
    r.rank = worldRank;
    r.size = worldSize;
    return;
  }
  r.lib = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
  if (!r.lib) r.lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!r.lib) r.lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!r.lib) throw std::runtime_error(std::string("cannot load librccl.so: ") + dlerror());
  bindSym(r.lib, "ncclGetUniqueId", r.GetUniqueId);
  bindSym(r.lib, "ncclCommInitRank", r.CommInitRank);
  bindSym(r.lib, "ncclAllReduce", r.AllReduce);
  bindSym(r.lib, "ncclCommDestroy", r.CommDestroy);
  bindSym(r.lib, "ncclGetErrorString", r.GetErrorString);
  // File rendezvous (the reference's --rndv_filepath).  The record carries a magic word and rank 0's wall-clock time of
  // publication: a reader accepts a record published no earlier than two minutes before its own start at once (an older one
  // only after it has stayed unchanged for 30 s, below), rank 0
  // removes whatever an earlier run left under the name before it publishes, and removes its own record once every rank
  // has joined (ncclCommInitRank is collective) -- so a `continue` / re-run with the same --rndv_filepath never picks up
  // the previous run's ncclUniqueId (which would hang ncclCommInitRank).
  // A launcher that can name its launch closes the remaining window (round-4 advice: a left-over record stays "unchanged for
  // 30 s" too, and is accepted when THIS run's rank 0 starts more than 30 s late): `--rndv_filepath=<dir>#<launch id>` -- any text
  // that is the same on every rank of one launch and differs between launches (a job id, rank 0's start time).  The record then
  // carries a hash of it, readers take a record with their own launch id at once and no other, whatever its age.
  struct RndvRecord { unsigned long long magic; long long publishedNs; unsigned long long launch; ncclUniqueId id; };
  constexpr unsigned long long kRndvMagic = 0x7732'6c72'6e64'7632ull;   // "w2lrndv2"
  std::string rndvDir = rndvFilepath;
  unsigned long long launch = 0;   // 0 = no launch id given: the time rules below
  {
    // ('#' separates the id only when the whole text does not name an existing directory: a rendezvous directory whose own path
    //  contains '#' keeps working as before -- round-5 advice)
    struct stat sb;
    const bool wholeIsDir = stat(rndvFilepath.c_str(), &sb) == 0 && S_ISDIR(sb.st_mode);
    const size_t h = wholeIsDir ? std::string::npos : rndvFilepath.rfind('#');
    if (h != std::string::npos) {
      rndvDir = rndvFilepath.substr(0, h);
      launch = 1469598103934665603ull;   // FNV-1a of the id text (never 0 for a non-empty id; an empty id counts as none)
      for (size_t i = h + 1; i < rndvFilepath.size(); ++i) launch = (launch ^ (unsigned char)rndvFilepath[i]) * 1099511628211ull;
      if (h + 1 == rndvFilepath.size()) launch = 0;
      else if (launch == 0) launch = 1;
    }
  }
  auto nowNs = [] { timespec ts; clock_gettime(CLOCK_REALTIME, &ts); return (long long)ts.tv_sec * 1000000000ll + ts.tv_nsec; };
  const long long startNs = nowNs();
  RndvRecord rec;
  std::memset(&rec, 0, sizeof rec);
  ncclUniqueId& id = rec.id;
  const std::string path = rndvDir + "/w2l_nccl_id." + std::to_string(worldSize);
  if (worldSize > 1 && rndvDir.empty()) throw std::invalid_argument("initDistributed: --rndv_filepath is needed for world_size > 1");
  if (worldRank == 0) {
    ncclCheck(r.GetUniqueId(&id), "ncclGetUniqueId");
    if (worldSize > 1) {   // write to a temporary name, then rename: readers never see a partial file
      (void)unlink(path.c_str());   // a record left behind by a run that died during its rendezvous
      rec.magic = kRndvMagic;
      rec.publishedNs = nowNs();
      rec.launch = launch;
      const std::string tmp = path + ".tmp";
      { std::ofstream f(tmp, std::ios::binary); f.write((const char*)&rec, sizeof rec); if (!f) throw std::runtime_error("cannot write " + tmp); }
      if (rename(tmp.c_str(), path.c_str()) != 0) throw std::runtime_error("cannot publish " + path);
    }
  } else {
    // A record published within two minutes of this rank's start is taken at once.  An OLDER one is either a rank 0 that started
    // long before this rank (staggered container start-up, clock skew between nodes) or the left-over of a run that died during
    // its rendezvous: it is accepted once it has stayed unchanged for 30 s -- a live rank 0 of THIS run would have replaced a
    // left-over by then (it unlinks and republishes at start-up).  (Round-3 advice: the old rule rejected the valid record of a
    // rank 0 more than two minutes ahead and failed after ten.)
    bool got = false;
    long long oldSince = 0, oldPublished = 0;
    for (int tries = 0; tries < 6000 && !got; ++tries) {   // up to 10 minutes
      std::ifstream f(path, std::ios::binary);
      RndvRecord in;
      if (f && f.read((char*)&in, sizeof in) && in.magic == kRndvMagic) {
        const long long now = nowNs();
        if (launch != 0 || in.launch != 0) {
          if (in.launch == launch) { rec = in; got = true; }   // (a record of another launch, or of a run without an id: keep waiting)
          else if (tries % 300 == 299)                          // every 30 s: say what is being waited for
            std::fprintf(stderr, "initDistributed: rank %d waits for a rendezvous record of launch %016llx in %s; the record there is of launch %016llx\n",
                         worldRank, launch, path.c_str(), in.launch);
        } else if (in.publishedNs >= startNs - 120ll * 1000000000ll) {
          rec = in; got = true;
        } else if (oldPublished == in.publishedNs && now - oldSince >= 30ll * 1000000000ll) {
          rec = in; got = true;
        } else if (oldPublished != in.publishedNs) {
          oldPublished = in.publishedNs; oldSince = now;
        }
      } else {
        oldPublished = 0;
      }
      if (!got) usleep(100000);
    }
    if (!got) throw std::runtime_error("rendezvous file " + path + " did not appear (or only a stale one from an earlier run)");
  }
  ncclCheck(r.CommInitRank(&r.comm, worldSize, id, worldRank), "ncclCommInitRank");
  if (worldRank == 0 && worldSize > 1) (void)unlink(path.c_str());   // every rank has read it: nothing stale survives a successful start
  r.rank = worldRank;
  r.size = worldSize;
}
}  // namespace runtime
}  // namespace pkg

// ------------------------------------------------------------------------------------------------ plugin loader
namespace pkg {
namespace runtime {
ModulePlugin::ModulePlugin(const std::string& name) : name_(name) {
  const bool isArch = name.size() >= 5 && name.compare(name.size() - 5, 5, ".arch") == 0;
  if (isArch) return;
  handle_ = dlopen(name.c_str(), RTLD_LAZY);
  if (!handle_) throw std::runtime_error("unable to load library <" + name + ">: " + std::string(dlerror()));
  dlerror();
  create_ = (fl::Module * (*)(int64_t, int64_t)) dlsym(handle_, "createModule");
  if (!create_) { const char* e = dlerror(); throw std::runtime_error("unable to resolve symbol <createModule>: " + std::string(e ? e : "?")); }
}
ModulePlugin::~ModulePlugin() {}  // the handle stays open: modules created by the plugin may outlive the loader
std::shared_ptr<fl::Module> ModulePlugin::arch(int64_t nFeatures, int64_t nClasses) {
  if (!create_) return speech::buildSequentialModule(name_, nFeatures, nClasses);
  return std::shared_ptr<fl::Module>(create_(nFeatures, nClasses));
}
}  // namespace runtime

// ------------------------------------------------------------------------------------------------ network
namespace speech {

CriterionScaleMode getCriterionScaleMode(const std::string& onorm, bool sqnorm) {
  return (CriterionScaleMode)w2l::criterionScaleMode(onorm, sqnorm);
}

namespace {

struct ArchLine : public fl::Module {  // one line of the arch file: bookkeeping only (the pipeline runs as a whole)
  explicit ArchLine(std::string s) : text(std::move(s)) {}
  std::vector<Variable> forward(const std::vector<Variable>&) override {
    throw std::logic_error("arch-file layers run as one planned pipeline: call forward() on the Sequential");
  }
  std::string prettyString() const override { return text; }
  std::string text;
};

// the planned pipeline behind a Sequential built from an arch file
class PlannedNet : public fl::Sequential {
 public:
  PlannedNet(std::shared_ptr<w2l::Sequential> net, int64_t nFeat, int64_t nLabel) : net_(std::move(net)), nFeat_((int)nFeat), nLabel_((int)nLabel) {
    const size_t n = net_->paramFloats();
    paramArena_ = devAlloc((n + 4) * sizeof(float));
    gradArena_ = devAlloc((n + 4) * sizeof(float));
    // (the 4-float alignment gaps between the parameters are never written again: whole-arena norms / updates read zeros there)
    w2l::hipCheck(hipMemsetAsync(paramArena_.get(), 0, (n + 4) * sizeof(float), S()), "params");
    w2l::hipCheck(hipMemsetAsync(gradArena_.get(), 0, (n + 4) * sizeof(float), S()), "grads");
    std::vector<float> host(n);
    net_->initParams(host.data(), 1);
    w2l::hipCheck(hipMemcpyAsync(paramArena_.get(), host.data(), n * sizeof(float), hipMemcpyHostToDevice, S()), "params");
    w2l::hipCheck(hipStreamSynchronize(S()), "params");
    for (const auto& pi : net_->params()) {
      // a view into the flat arena, in the module's INTERNAL layout (import / export of reference-ordered tensors:
      // w2l_trainer_import_param); dims carry the reference's ArrayFire shape
      af::dim4 d(1, 1, 1, 1);
      for (size_t k = 0; k < pi.refShape.size() && k < 4; ++k) d[(int)k] = pi.refShape[k];
      if (d.elements() != (af::dim_t)pi.numel) d = af::dim4((af::dim_t)pi.numel);
      params_.push_back(Variable(af::array::wrap((float*)paramArena_.get() + pi.offset, d, af::f32, paramArena_), true));
      childOfParam_.push_back(-1);
      childIndexOfParam_.push_back(0);
    }
    std::istringstream ss(net_->prettyString());
    std::string line;
    while (std::getline(ss, line)) modules_.push_back(std::make_shared<ArchLine>(line));
  }

  // {features (T, NFEAT, 1, B) [, inputSizes (1, B)]} -> {emissions (NLABEL, T', B)}
  std::vector<Variable> forward(const std::vector<Variable>& inputs) override {
    if (inputs.empty()) throw std::invalid_argument("network forward: no input");
    const Variable& in = inputs[0];
    if (in.type() != af::f32) throw std::invalid_argument("network forward: features must be f32");
    const int T = (int)in.dims(0), B = (int)in.dims(3);
    if (nFeat_ < 0) nFeat_ = (int)(in.dims(1) * in.dims(2));   // (a Sequential of layer objects: the first input says)
    if (in.dims(1) * in.dims(2) != nFeat_) throw std::invalid_argument("network forward: feature dimension != NFEAT");
    for (size_t i = 0; i < params_.size(); ++i)
      if (params_[i].array().device<float>() != (float*)paramArena_.get() + net_->params()[i].offset)
        throw std::logic_error("setParams on a planned network must write INTO the parameter (copy), not rebind it");
    if (B != B_ || T != T_) {
      const size_t fl = net_->plan(B, T, nFeat_);
      if (nLabel_ < 0) nLabel_ = net_->outAct().F;           // (... and the last layer the label count)
      arena_ = devAlloc(fl * sizeof(float));
      arenaFloats_ = fl;
      B_ = B; T_ = T;
      dEm_ = devAlloc(((size_t)B * net_->outAct().T * nLabel_ + 64) * sizeof(float));
    }
    w2l::Ctx c;
    c.stream = S(); c.train = train_; c.seed = 0x9E3779B9u * (++step_);
    c.params = (float*)paramArena_.get(); c.grads = (float*)gradArena_.get();
    c.bf16 = mixed_;
    if (inputs.size() >= 2 && !inputs[1].isempty()) {  // inputSizes (1, B): padding mask of the Transformer blocks
      // (1, B + 1): the extra entry is the size the T input frames correspond to -- a batch padded beyond its longest utterance
      if (inputs[1].type() != af::f32 || (inputs[1].elements() != B && inputs[1].elements() != B + 1))
        throw std::invalid_argument("network forward: inputSizes must be f32 (1, B)");
      c.inputSizes = inputs[1].array().device<float>();
      if (inputs[1].elements() == B + 1) c.inputSizeFull = c.inputSizes + B;
      c.inputT = T;
    }
    const int prevMode = mixed_ ? w2l_set_matmul_precision(1) : 0;
    const float* em = net_->forward(c, (float*)arena_.get(), in.array().device<float>());
    if (mixed_) w2l_set_matmul_precision(prevMode);
    const int To = net_->outAct().T;
    af::array out = af::array::wrap((void*)em, af::dim4(nLabel_, To, B), af::f32, arena_);
    auto self = this;
    auto ctx = c;
    std::vector<Variable> deps(params_.begin(), params_.end());
    deps.push_back(in);
    return {Variable(out, deps, [self, ctx](std::vector<Variable>& ins, const Variable& gradOut) mutable {
      // whole-network backward: overwrites every parameter gradient in the flat gradient arena, then hands each
      // parameter a VIEW of its slice (no per-parameter copies; netoptim steps on the views)
      const int prevMode = self->mixed_ ? w2l_set_matmul_precision(1) : 0;
      self->net_->backward(ctx, (float*)self->arena_.get(), gradOut.array().device<float>());
      if (self->mixed_) w2l_set_matmul_precision(prevMode);
      for (size_t i = 0; i + 1 < ins.size(); ++i) {
        const auto& pi = self->net_->params()[i];
        ins[i].zeroGrad();
        ins[i].addGrad(Variable(af::array::wrap((float*)self->gradArena_.get() + pi.offset, ins[i].dims(), af::f32, self->gradArena_), false));
      }
    })};
  }
  std::string prettyString() const override { return net_->prettyString(); }
  w2l::Sequential& impl() { return *net_; }
  float* paramPtr() { return (float*)paramArena_.get(); }
  float* gradPtr() { return (float*)gradArena_.get(); }
  bool mixed_ = false;
  std::string archSha_;             // sha-256 of the arch text this network was built from (checkpoint header)
  int nFeat() const { return nFeat_; }
  int nLabel() const { return nLabel_; }
  uint32_t step() const { return step_; }      // forwards so far = position of the dropout-seed stream
  void setStep(uint32_t s) { step_ = s; }
  // gradient buckets of the data-parallel overlap (CoalescingReducer): events recorded by backward() on the compute stream
  std::vector<size_t> bucketOffsets_;
  std::vector<hipEvent_t> bucketEvents_;
  void installBuckets(std::vector<size_t> offs) {
    for (auto e : bucketEvents_) (void)hipEventDestroy(e);
    bucketEvents_.clear();
    for (size_t k = 0; k < offs.size(); ++k) {
      hipEvent_t e;
      w2l::hipCheck(hipEventCreateWithFlags(&e, hipEventDisableTiming), "bucket event");
      bucketEvents_.push_back(e);
    }
    bucketOffsets_ = offs;
    net_->setGradBuckets(offs, bucketEvents_);
  }

 private:
  std::shared_ptr<w2l::Sequential> net_;
  int nFeat_, nLabel_;
  std::shared_ptr<void> paramArena_, gradArena_, arena_, dEm_;
  size_t arenaFloats_ = 0;
  int B_ = 0, T_ = 0;
  uint32_t step_ = 0;
};

}  // namespace

namespace {
// ---- sha-256 (FIPS 180-4) of the arch text: the checkpoint header's "arch_sha256" (wav2letter_amd/checkpoint.py hashes the same bytes)
std::string sha256Hex(const std::string& msg) {
  static const uint32_t K[64] = {
      0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be,
      0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa,
      0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85,
      0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3,
      0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f,
      0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
  uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
  std::string m = msg;
  const uint64_t bits = (uint64_t)msg.size() * 8;
  m.push_back((char)0x80);
  while (m.size() % 64 != 56) m.push_back('\0');
  for (int i = 7; i >= 0; --i) m.push_back((char)((bits >> (8 * i)) & 0xff));
  auto rotr = [](uint32_t x, int n) { return (x >> n) | (x << (32 - n)); };
  for (size_t off = 0; off < m.size(); off += 64) {
    uint32_t w[64];
    for (int i = 0; i < 16; ++i)
      w[i] = ((uint32_t)(uint8_t)m[off + 4 * i] << 24) | ((uint32_t)(uint8_t)m[off + 4 * i + 1] << 16) | ((uint32_t)(uint8_t)m[off + 4 * i + 2] << 8) |
             (uint32_t)(uint8_t)m[off + 4 * i + 3];
    for (int i = 16; i < 64; ++i) {
      const uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3), s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
      w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    for (int i = 0; i < 64; ++i) {
      const uint32_t S1 = rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25), ch = (e & f) ^ (~e & g), t1 = hh + S1 + ch + K[i] + w[i];
      const uint32_t S0 = rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22), mj = (a & b) ^ (a & c) ^ (b & c), t2 = S0 + mj;
      hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
  }
  char out[65];
  for (int i = 0; i < 8; ++i) snprintf(out + 8 * i, 9, "%08x", h[i]);
  return std::string(out, 64);
}
std::vector<PlannedNet*>& plannedNets() { static std::vector<PlannedNet*> v; return v; }
}  // namespace

std::shared_ptr<fl::Sequential> buildSequentialModuleFromText(const std::string& archText, int64_t nFeatures, int64_t nClasses) {
  auto net = std::make_shared<PlannedNet>(w2l::buildSequentialFromText(archText, nFeatures, nClasses), nFeatures, nClasses);
  net->archSha_ = sha256Hex(archText);
  plannedNets().push_back(net.get());   // (networks live as long as the trainer process: no removal)
  return net;
}
std::shared_ptr<fl::Sequential> buildSequentialModule(const std::string& archfile, int64_t nFeatures, int64_t nClasses) {
  return buildSequentialModuleFromText(w2l::readFile(archfile), nFeatures, nClasses);
}

}  // namespace speech
}  // namespace pkg
namespace {
bool findArena(const float* p, ArenaInfo& out) {
  for (auto* n : pkg::speech::plannedNets()) {
    float* b = n->gradPtr();
    const size_t fl = n->impl().paramFloats();
    if (p >= b && p < b + fl) { out.base = b; out.floats = fl; out.net = n; return true; }
  }
  return false;
}
bool gradientArenaOf(const float* g, float*& base) {
  ArenaInfo ai;
  if (!findArena(g, ai)) return false;
  base = ai.base;
  return true;
}
std::vector<FlatRun> gradientRuns(const std::vector<Variable>& params) {
  std::vector<FlatRun> runs;
  std::vector<std::pair<void*, size_t>> seen;   // (network, parameters of it found at their own offsets)
  std::vector<ArenaInfo> info(params.size());
  std::vector<char> inArena(params.size(), 0);
  for (size_t i = 0; i < params.size(); ++i) {
    if (!params[i].isGradAvailable()) continue;
    float* g = params[i].grad().array().device<float>();
    if (params[i].grad().type() != af::f32 || !findArena(g, info[i])) continue;
    const auto& pis = ((pkg::speech::PlannedNet*)info[i].net)->impl().params();
    bool slice = false;
    for (const auto& pi : pis) slice = slice || (g == info[i].base + pi.offset && (size_t)params[i].elements() == pi.numel);
    if (!slice) continue;
    inArena[i] = 1;
    bool found = false;
    for (auto& sp : seen)
      if (sp.first == info[i].net) { ++sp.second; found = true; }
    if (!found) seen.push_back({info[i].net, 1});
  }
  std::vector<void*> whole;
  for (auto& sp : seen)
    if (sp.second == ((pkg::speech::PlannedNet*)sp.first)->impl().params().size()) whole.push_back(sp.first);
  std::vector<void*> emitted;
  for (size_t i = 0; i < params.size(); ++i) {
    if (!params[i].isGradAvailable()) continue;
    const bool w = inArena[i] && std::find(whole.begin(), whole.end(), info[i].net) != whole.end();
    if (!w) { runs.push_back({params[i].grad().array().device<float>(), (size_t)params[i].elements()}); continue; }
    if (std::find(emitted.begin(), emitted.end(), info[i].net) != emitted.end()) continue;
    emitted.push_back(info[i].net);
    runs.push_back({info[i].base, info[i].floats});
  }
  return runs;
}
bool flatParameterArena(const std::vector<Variable>& params, float*& base, size_t& floats, std::vector<size_t>& offsets) {
  if (params.empty()) return false;
  for (auto* n : pkg::speech::plannedNets()) {
    const auto& pis = n->impl().params();
    if (pis.size() != params.size()) continue;
    bool all = true;
    for (size_t i = 0; all && i < pis.size(); ++i)
      all = params[i].type() == af::f32 && params[i].array().device<float>() == n->paramPtr() + pis[i].offset && (size_t)params[i].elements() == pis[i].numel;
    if (!all) continue;
    base = n->paramPtr(); floats = n->impl().paramFloats();
    offsets.clear();
    for (const auto& pi : pis) offsets.push_back(pi.offset);
    return true;
  }
  return false;
}
std::vector<size_t> arenaParamOffsets(void* net) {
  std::vector<size_t> v;
  for (const auto& pi : ((pkg::speech::PlannedNet*)net)->impl().params()) v.push_back(pi.offset);
  return v;
}
const std::vector<size_t>& arenaBucketOffsets(void* net) { return ((pkg::speech::PlannedNet*)net)->bucketOffsets_; }
hipEvent_t arenaBucketEvent(void* net, size_t k) { return ((pkg::speech::PlannedNet*)net)->bucketEvents_.at(k); }
void arenaInstallBuckets(void* net, std::vector<size_t> offs) { ((pkg::speech::PlannedNet*)net)->installBuckets(std::move(offs)); }
}  // namespace
namespace pkg {
namespace speech {

namespace {
// the planned pipeline behind a network handle: the PlannedNet itself, or the one a Sequential of layer objects plans
PlannedNet* plannedOf(fl::Module* m) {
  if (auto* p = dynamic_cast<PlannedNet*>(m)) return p;
  if (auto* s = dynamic_cast<fl::Sequential*>(m)) return dynamic_cast<PlannedNet*>(s->planned().get());
  return nullptr;
}
}  // namespace

FlatView flatParameters(const std::shared_ptr<fl::Module>& network) {
  auto* p = plannedOf(network.get());
  return p ? FlatView{p->paramPtr(), p->impl().paramFloats()} : FlatView{nullptr, 0};
}
void setMixedPrecision(const std::shared_ptr<fl::Module>& network, bool on) {
  if (auto* p = plannedOf(network.get())) p->mixed_ = on;
}
uint32_t networkStep(const std::shared_ptr<fl::Module>& network) {
  auto* p = plannedOf(network.get());
  return p ? p->step() : 0;
}
void setNetworkStep(const std::shared_ptr<fl::Module>& network, uint32_t step) {
  if (auto* p = plannedOf(network.get())) p->setStep(step);
}
FlatView flatGradients(const std::shared_ptr<fl::Module>& network) {
  auto* p = plannedOf(network.get());
  return p ? FlatView{p->gradPtr(), p->impl().paramFloats()} : FlatView{nullptr, 0};
}

// ------------------------------------------------------------------------------------------------ criteria
namespace {

struct CritState {
  std::shared_ptr<w2l::SequenceCriterion> impl;
  std::shared_ptr<void> ws;
  size_t wsBytes = 0;
  void* workspace(int B, int T, int N, int L) {
    const size_t need = impl->workspaceBytes(B, T, N, L) + 256;
    if (need > wsBytes) { ws = devAlloc(need); wsBytes = need; }
    return ws.get();
  }
  // viterbiPath has a workspace of ITS OWN: the Trainer decodes between the criterion's forward and loss.backward() on every
  // report iteration (Train.cpp:1699-1716), and the backward kernels read what forward left in `ws`
  std::shared_ptr<void> vws;
  size_t vwsBytes = 0;
  void* viterbiWorkspace(int B, int T, int N) {
    const size_t need = impl->workspaceBytes(B, T, N, 1) + 256;
    if (need > vwsBytes) { vws = devAlloc(need); vwsBytes = need; }
    return vws.get();
  }
};

void checkCritInputs(const std::vector<Variable>& inputs, int& N, int& T, int& B, int& L) {
  if (inputs.size() < 2) throw std::invalid_argument("Invalid inputs size");
  const Variable& em = inputs[0];
  const Variable& tg = inputs[1];
  if (em.type() != af::f32) throw std::invalid_argument("criterion: emission must be f32");
  if (tg.type() != af::s32) throw std::invalid_argument("criterion: target must be s32");
  N = (int)em.dims(0); T = (int)em.dims(1); B = (int)em.dims(2); L = (int)tg.dims(0);
  if (tg.dims(1) != B) throw std::invalid_argument("criterion: target batch size != emission batch size");
  if (N <= 0 || T <= 0 || B <= 0 || L <= 0) throw std::invalid_argument("criterion: empty input");
}

// one forward: loss (B) Variable whose gradFunc runs the criterion's backward kernels
std::vector<Variable> critForward(std::shared_ptr<CritState> st, const std::vector<Variable>& inputs, const Variable* trans) {
  int N, T, B, L;
  checkCritInputs(inputs, N, T, B, L);
  w2l::Ctx c;
  c.stream = S();
  void* ws = st->workspace(B, T, N, L);
  af::array loss(af::dim4(B), af::f32);
  float* tp = trans ? trans->array().device<float>() : nullptr;
  st->impl->forward(c, B, T, N, L, inputs[0].array().device<float>(), inputs[1].array().device<int>(), loss.device<float>(), ws, tp);
  std::vector<Variable> deps{inputs[0]};
  if (trans) deps.push_back(*trans);
  Variable target = inputs[1];
  return {Variable(loss, deps, [st, target, N, T, B, L, c](std::vector<Variable>& ins, const Variable& gradOut) mutable {
    af::array dEm(af::dim4(N, T, B), af::f32);
    af::array dTr;
    float* tp = nullptr;
    float* dtp = nullptr;
    if (ins.size() > 1) {
      dTr = af::array(ins[1].dims(), af::f32);
      tp = ins[1].array().device<float>();
      dtp = dTr.device<float>();
    }
    st->impl->backward(c, B, T, N, L, ins[0].array().device<float>(), target.array().device<int>(), gradOut.array().device<float>(),
                       dEm.device<float>(), st->ws.get(), tp, dtp);
    ins[0].addGrad(Variable(dEm, false));
    if (ins.size() > 1) ins[1].addGrad(Variable(dTr, false));
  })};
}

}  // namespace

struct ASGState : CritState {};
static std::shared_ptr<CritState>& stateOf(const void* key) {
  static std::unordered_map<const void*, std::shared_ptr<CritState>> m;
  return m[key];
}

ASGLoss::ASGLoss(int N, CriterionScaleMode scalemode, double transdiag) : N_(N), scaleMode_(scalemode) {
  if (N <= 0) throw std::invalid_argument("ASGLoss: N must be positive");
  auto st = std::make_shared<CritState>();
  st->impl = w2l::makeASGLoss(N, (int)scalemode, transdiag);
  stateOf(this) = st;
  std::vector<float> host(st->impl->paramFloats());
  st->impl->initParams(host.data());
  af::array tr(af::dim4(N, N), af::f32);   // exactly N*N floats (the kernels read N*N)
  w2l::hipCheck(hipMemcpyAsync(tr.device<float>(), host.data(), (size_t)N * N * sizeof(float), hipMemcpyHostToDevice, S()), "transitions");
  w2l::hipCheck(hipStreamSynchronize(S()), "transitions");
  params_ = {Variable(tr, true)};
}
std::vector<Variable> ASGLoss::forward(const std::vector<Variable>& inputs) {
  if (!inputs.empty() && inputs[0].dims(0) != N_) throw std::invalid_argument("ASGLoss: N doesn't match with the letter size");
  return critForward(stateOf(this), inputs, &params_[0]);
}
af::array ASGLoss::viterbiPath(const af::array& input, const af::array&) {
  const int N = (int)input.dims(0), T = (int)input.dims(1), B = (int)input.dims(2);
  if (N != N_) throw std::invalid_argument("ASGLoss: N doesn't match with the letter size");
  auto st = stateOf(this);
  af::array path(af::dim4(T, B), af::s32);
  w2l::Ctx c;
  c.stream = S();
  st->impl->viterbiPath(c, B, T, N, input.device<float>(), path.device<int>(), st->viterbiWorkspace(B, T, N), params_[0].array().device<float>());
  return path;
}
af::array ASGLoss::viterbiPathWithTarget(const af::array& input, const af::array& target, const af::array&, const af::array&) {
  const int N = (int)input.dims(0), T = (int)input.dims(1), B = (int)input.dims(2), L = (int)target.dims(0);
  if (N != N_) throw std::invalid_argument("ASGLoss: N doesn't match with the letter size");
  if (target.type() != af::s32 || target.dims(1) != B) throw std::invalid_argument("viterbiPathWithTarget: bad target");
  af::array path(af::dim4(T, B), af::s32);
  af::array ts(af::dim4(B), af::s32);
  auto ws = devAlloc(w2l_fac_workspace_size(B, T, N, L) + 256);
  w2l::w2lCheck(w2l_batch_target_size(B, L, T, target.device<int>(), ts.device<int>(), S()), "target size");
  w2l::w2lCheck(w2l_fac_viterbi(B, T, N, L, input.device<float>(), target.device<int>(), ts.device<int>(), params_[0].array().device<float>(),
                                path.device<int>(), ws.get(), S()), "fac viterbi");
  af::sync();  // ws is released at return
  return path;
}
std::string ASGLoss::prettyString() const { return "AutoSegmentationCriterion"; }

CTCLoss::CTCLoss(CriterionScaleMode scalemode) : scaleMode_(scalemode) {
  auto st = std::make_shared<CritState>();
  st->impl = w2l::makeCTCLoss((int)scalemode);
  stateOf(this) = st;
}
std::vector<Variable> CTCLoss::forward(const std::vector<Variable>& inputs) { return critForward(stateOf(this), inputs, nullptr); }
af::array CTCLoss::viterbiPath(const af::array& input, const af::array&) {
  const int N = (int)input.dims(0), T = (int)input.dims(1), B = (int)input.dims(2);
  auto st = stateOf(this);
  af::array path(af::dim4(T, B), af::s32);
  w2l::Ctx c;
  c.stream = S();
  st->impl->viterbiPath(c, B, T, N, input.device<float>(), path.device<int>(), nullptr, nullptr);
  return path;
}
af::array CTCLoss::viterbiPathWithTarget(const af::array&, const af::array&, const af::array&, const af::array&) {
  throw std::logic_error("CTCLoss::viterbiPathWithTarget is not used by the recipes' training loop (forced alignment is an ASG / tools feature)");
}
std::string CTCLoss::prettyString() const { return "ConnectionistTemporalClassificationCriterion"; }

}  // namespace speech
}  // namespace pkg
}  // namespace fl

// ------------------------------------------------------------------------------------------------ Serializer
// The W2LAMD01 container of wav2letter_amd/checkpoint.py (see the layout there): 8-byte magic, uint32 JSON-header length,
// JSON header, then every tensor's float32 data 16-byte aligned in header order.  Network tensors are stored in the
// REFERENCE's layouts (w2l::Sequential::exportParam / importParam), the ASG transitions (N, N) raw, the optimizer state
// as flat arenas [network parameters in arena order | criterion parameters] ("momentum": SGD velocity / Adagrad variance /
// Adadelta accGrad; "state2": Adadelta accDelta) -- the same bytes the Python front-end writes, so either side resumes
// the other's run.
namespace fl {
namespace pkg {
namespace runtime {
namespace {

struct JVal {   // the subset of JSON the header uses
  enum Kind { NUL, NUM, STR, ARR, OBJ } kind = NUL;
  double num = 0;
  std::string str;
  std::vector<JVal> arr;
  std::vector<std::pair<std::string, JVal>> obj;
  const JVal* get(const std::string& k) const {
    for (auto& kv : obj) if (kv.first == k) return &kv.second;
    return nullptr;
  }
};
struct JParser {
  const std::string& s;
  size_t i = 0;
  explicit JParser(const std::string& t) : s(t) {}
  void ws() { while (i < s.size() && (s[i] == ' ' || s[i] == '\n' || s[i] == '\t' || s[i] == '\r')) ++i; }
  [[noreturn]] void fail(const char* what) { throw std::runtime_error(std::string("checkpoint header: ") + what + " at byte " + std::to_string(i)); }
  std::string parseString() {
    if (s[i] != '"') fail("expected a string");
    ++i;
    std::string out;
    while (i < s.size() && s[i] != '"') {
      char c = s[i++];
      if (c == '\\') {
        if (i >= s.size()) fail("bad escape");
        char e = s[i++];
        switch (e) {
          case 'n': out += '\n'; break;
          case 't': out += '\t'; break;
          case 'r': out += '\r'; break;
          case 'b': out += '\b'; break;
          case 'f': out += '\f'; break;
          case 'u': {
            if (i + 4 > s.size()) fail("bad \\u escape");
            unsigned cp = (unsigned)std::stoul(s.substr(i, 4), nullptr, 16);
            i += 4;
            if (cp < 0x80) out += (char)cp;
            else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
            else { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
            break;
          }
          default: out += e;
        }
      } else {
        out += c;
      }
    }
    if (i >= s.size()) fail("unterminated string");
    ++i;
    return out;
  }
  JVal parse() {
    ws();
    if (i >= s.size()) fail("unexpected end");
    JVal v;
    const char c = s[i];
    if (c == '{') {
      v.kind = JVal::OBJ;
      ++i; ws();
      if (s[i] == '}') { ++i; return v; }
      for (;;) {
        ws();
        std::string k = parseString();
        ws();
        if (s[i] != ':') fail("expected ':'");
        ++i;
        v.obj.emplace_back(std::move(k), parse());
        ws();
        if (s[i] == ',') { ++i; continue; }
        if (s[i] == '}') { ++i; break; }
        fail("expected ',' or '}'");
      }
    } else if (c == '[') {
      v.kind = JVal::ARR;
      ++i; ws();
      if (s[i] == ']') { ++i; return v; }
      for (;;) {
        v.arr.push_back(parse());
        ws();
        if (s[i] == ',') { ++i; continue; }
        if (s[i] == ']') { ++i; break; }
        fail("expected ',' or ']'");
      }
    } else if (c == '"') {
      v.kind = JVal::STR;
      v.str = parseString();
    } else if (s.compare(i, 4, "true") == 0) { v.kind = JVal::NUM; v.num = 1; i += 4; }
    else if (s.compare(i, 5, "false") == 0) { v.kind = JVal::NUM; v.num = 0; i += 5; }
    else if (s.compare(i, 4, "null") == 0) { i += 4; }
    else {
      size_t used = 0;
      v.kind = JVal::NUM;
      try { v.num = std::stod(s.substr(i, 64), &used); } catch (...) { fail("bad number"); }
      i += used;
    }
    return v;
  }
};
std::string jstr(const std::string& t) {
  std::string o = "\"";
  for (unsigned char c : t) {
    if (c == '"') o += "\\\"";
    else if (c == '\\') o += "\\\\";
    else if (c == '\n') o += "\\n";
    else if (c == '\t') o += "\\t";
    else if (c == '\r') o += "\\r";
    else if (c < 0x20) { char b[8]; snprintf(b, sizeof b, "\\u%04x", c); o += b; }
    else o += (char)c;
  }
  return o + "\"";
}

struct Tensor { std::string name, kind; std::vector<long> shape; std::vector<float> data; };

std::vector<float> toHost(const float* dev, size_t n) {
  std::vector<float> h(n);
  if (n) {
    w2l::hipCheck(hipMemcpyAsync(h.data(), dev, n * sizeof(float), hipMemcpyDeviceToHost, S()), "checkpoint copy");
    w2l::hipCheck(hipStreamSynchronize(S()), "checkpoint copy");
  }
  return h;
}
void toDevice(float* dev, const float* host, size_t n) {
  if (!n) return;
  w2l::hipCheck(hipMemcpyAsync(dev, host, n * sizeof(float), hipMemcpyHostToDevice, S()), "checkpoint copy");
  w2l::hipCheck(hipStreamSynchronize(S()), "checkpoint copy");
}
speech::PlannedNet& planned(const std::shared_ptr<fl::Module>& network) {
  auto* p = dynamic_cast<speech::PlannedNet*>(network.get());
  if (!p) throw std::invalid_argument("Serializer: the network must come from an arch file (buildSequentialModule / ModulePlugin::arch)");
  return *p;
}
std::string critName(const std::shared_ptr<fl::Module>& criterion) {
  if (dynamic_cast<speech::ASGLoss*>(criterion.get())) return "asg";
  if (dynamic_cast<speech::CTCLoss*>(criterion.get())) return "ctc";
  throw std::invalid_argument("Serializer: unsupported criterion (this build: ASGLoss, CTCLoss)");
}
size_t critSlot(const std::shared_ptr<fl::Module>& criterion) {   // floats the criterion occupies in the flat arenas (16-byte slots)
  size_t n = 0;
  for (auto& p : criterion->params()) n += ((size_t)p.elements() + 3) / 4 * 4;
  return n;
}
// flat optimizer arena [network | criterion] <- per-parameter state arrays (slot 0 or 1); false when neither optimizer has that slot
bool gatherState(speech::PlannedNet& net, const std::shared_ptr<fl::Module>& criterion, fl::FirstOrderOptimizer* no, fl::FirstOrderOptimizer* co,
                 size_t slot, std::vector<float>& flat) {
  const size_t nNet = net.impl().paramFloats();
  flat.assign(nNet + critSlot(criterion), 0.f);
  bool any = false;
  if (no) {
    auto st = no->state();
    if (slot < st.size()) {
      any = true;
      const auto& table = net.impl().params();
      if (st[slot]->size() != table.size()) throw std::logic_error("Serializer: network optimizer state does not match the network");
      for (size_t i = 0; i < table.size(); ++i) {
        auto h = toHost((*st[slot])[i].device<float>(), table[i].numel);
        std::copy(h.begin(), h.end(), flat.begin() + table[i].offset);
      }
    }
  }
  if (co) {
    auto st = co->state();
    if (slot < st.size()) {
      any = true;
      size_t off = nNet;
      for (size_t i = 0; i < st[slot]->size(); ++i) {
        const size_t n = (size_t)(*st[slot])[i].elements();
        auto h = toHost((*st[slot])[i].device<float>(), n);
        std::copy(h.begin(), h.end(), flat.begin() + off);
        off += (n + 3) / 4 * 4;
      }
    }
  }
  return any;
}
void scatterState(speech::PlannedNet& net, fl::FirstOrderOptimizer* no, fl::FirstOrderOptimizer* co, size_t slot, const std::vector<float>& flat) {
  const size_t nNet = net.impl().paramFloats();
  if (no) {
    auto st = no->state();
    if (slot < st.size()) {
      const auto& table = net.impl().params();
      for (size_t i = 0; i < table.size(); ++i) toDevice((*st[slot])[i].device<float>(), flat.data() + table[i].offset, table[i].numel);
    }
  }
  if (co) {
    auto st = co->state();
    if (slot < st.size()) {
      size_t off = nNet;
      for (size_t i = 0; i < st[slot]->size(); ++i) {
        const size_t n = (size_t)(*st[slot])[i].elements();
        if (off + n > flat.size()) throw std::runtime_error("checkpoint: optimizer arena too short for the criterion state");
        toDevice((*st[slot])[i].device<float>(), flat.data() + off, n);
        off += (n + 3) / 4 * 4;
      }
    }
  }
}

struct Loaded { JVal header; std::vector<std::vector<float>> arrays; };
Loaded readFile(const std::string& path, bool headerOnly) {
  std::ifstream f(path, std::ios::binary);
  if (!f) throw std::runtime_error("cannot open checkpoint " + path);
  char magic[8];
  uint32_t n = 0;
  if (!f.read(magic, 8) || std::memcmp(magic, "W2LAMD01", 8) != 0) throw std::runtime_error(path + ": not a W2LAMD01 checkpoint");
  if (!f.read((char*)&n, 4)) throw std::runtime_error(path + ": truncated");
  std::string hdr(n, '\0');
  if (!f.read(&hdr[0], n)) throw std::runtime_error(path + ": truncated header");
  Loaded L;
  L.header = JParser(hdr).parse();
  if (headerOnly) return L;
  const JVal* ts = L.header.get("tensors");
  if (!ts || ts->kind != JVal::ARR) throw std::runtime_error(path + ": header without a tensor list");
  size_t pos = 12 + (size_t)n;
  for (auto& t : ts->arr) {
    const JVal* ne = t.get("numel");
    if (!ne) throw std::runtime_error(path + ": tensor without numel");
    const size_t numel = (size_t)ne->num;
    const size_t at = (pos + 15) / 16 * 16;
    f.seekg((std::streamoff)at);
    std::vector<float> a(numel);
    if (numel && !f.read((char*)a.data(), (std::streamsize)(numel * 4))) throw std::runtime_error(path + ": truncated tensor data");
    L.arrays.push_back(std::move(a));
    pos = at + numel * 4;
  }
  return L;
}
void configFromHeader(const JVal& h, std::string& version, Serializer::Config& config) {
  config.clear();
  if (const JVal* fl = h.get("flags"))
    for (auto& kv : fl->obj) config[kv.first] = kv.second.kind == JVal::STR ? kv.second.str : (kv.second.kind == JVal::NUM ? std::to_string((long long)kv.second.num) : "");
  auto it = config.find("version");
  version = it == config.end() ? "" : it->second;
}
void loadModel(const std::string& path, std::string& version, Serializer::Config& config, const std::shared_ptr<fl::Module>& network,
               const std::shared_ptr<fl::Module>& criterion, fl::FirstOrderOptimizer* no, fl::FirstOrderOptimizer* co) {
  Loaded L = readFile(path, false);
  configFromHeader(L.header, version, config);
  speech::PlannedNet& net = planned(network);
  auto num = [&](const char* k) { const JVal* v = L.header.get(k); return v ? (long)v->num : -1L; };
  if (num("nfeat") != net.nFeat() || num("nlabel") != net.nLabel()) throw std::runtime_error("checkpoint was written for a different NFEAT / NLABEL");
  const JVal* sha = L.header.get("arch_sha256");
  if (sha && sha->kind == JVal::STR && !sha->str.empty() && sha->str != net.archSha_) throw std::runtime_error("checkpoint was written for a different architecture file");
  const JVal* cn = L.header.get("criterion");
  if (cn && cn->kind == JVal::STR && cn->str != critName(criterion)) throw std::runtime_error("checkpoint was written for criterion '" + cn->str + "'");
  const auto& table = net.impl().params();
  const JVal& ts = *L.header.get("tensors");
  std::vector<float> host = toHost(net.paramPtr(), net.impl().paramFloats());
  size_t ni = 0;
  const std::vector<float>* mom = nullptr;
  const std::vector<float>* st2 = nullptr;
  bool haveCrit = false;
  for (size_t k = 0; k < ts.arr.size(); ++k) {
    const JVal* kind = ts.arr[k].get("kind");
    const std::string kd = kind ? kind->str : "";
    if (kd == "network") {
      if (ni >= table.size() || L.arrays[k].size() != table[ni].numel) throw std::runtime_error("checkpoint: parameter list does not match this network");
      net.impl().importParam(ni, L.arrays[k].data(), host.data());
      ++ni;
    } else if (kd == "criterion") {
      auto cp = criterion->params();
      if (cp.empty() || (size_t)cp[0].elements() != L.arrays[k].size()) throw std::runtime_error("checkpoint: criterion parameters do not match");
      toDevice(cp[0].array().device<float>(), L.arrays[k].data(), L.arrays[k].size());
      haveCrit = true;
    } else if (kd == "momentum") {
      mom = &L.arrays[k];
    } else if (kd == "state2") {
      st2 = &L.arrays[k];
    }
  }
  if (ni != table.size()) throw std::runtime_error("checkpoint: parameter list does not match this network");
  if (haveCrit != !criterion->params().empty()) throw std::runtime_error("checkpoint: criterion parameters do not match (one side has transitions, the other has none)");
  toDevice(net.paramPtr(), host.data(), host.size());
  if (no || co) {
    const JVal* op = L.header.get("optim");
    const std::string wantNet = no ? no->kind() : "sgd", wantCrit = co ? co->kind() : "sgd";
    if (op && op->kind == JVal::ARR && op->arr.size() == 2 && (mom || st2) && (op->arr[0].str != wantNet || op->arr[1].str != wantCrit))
      throw std::runtime_error("checkpoint holds the state of optimizers (" + op->arr[0].str + ", " + op->arr[1].str + "), this run uses (" + wantNet + ", " + wantCrit + ")");
    const size_t want = net.impl().paramFloats() + critSlot(criterion);
    if (mom) { if (mom->size() != want) throw std::runtime_error("checkpoint: optimizer arena does not match this trainer"); scatterState(net, no, co, 0, *mom); }
    if (st2) { if (st2->size() != want) throw std::runtime_error("checkpoint: optimizer arena does not match this trainer"); scatterState(net, no, co, 1, *st2); }
  }
  net.setStep((uint32_t)std::max(0L, num("step")));
}

}  // namespace

void Serializer::save(const std::string& path, const std::string& version, const Config& config, const std::shared_ptr<fl::Module>& network,
                      const std::shared_ptr<fl::Module>& criterion, const std::shared_ptr<fl::FirstOrderOptimizer>& netoptim,
                      const std::shared_ptr<fl::FirstOrderOptimizer>& critoptim) {
  speech::PlannedNet& net = planned(network);
  const auto& table = net.impl().params();
  std::vector<Tensor> tensors;
  const std::vector<float> host = toHost(net.paramPtr(), net.impl().paramFloats());
  for (size_t i = 0; i < table.size(); ++i) {
    Tensor t;
    t.name = table[i].name; t.kind = "network"; t.shape = {(long)table[i].numel};
    t.data.resize(table[i].numel);
    net.impl().exportParam(i, host.data(), t.data.data());
    tensors.push_back(std::move(t));
  }
  for (auto& p : criterion->params()) {
    Tensor t;
    t.name = "criterion.transitions"; t.kind = "criterion";
    t.shape = {(long)p.dims(0), (long)p.dims(1)};
    t.data = toHost(p.array().device<float>(), (size_t)p.elements());
    tensors.push_back(std::move(t));
  }
  std::vector<float> flat;
  if (gatherState(net, criterion, netoptim.get(), critoptim.get(), 0, flat)) {
    Tensor t; t.name = "netoptim.momentum(internal arena)"; t.kind = "momentum"; t.shape = {(long)flat.size()}; t.data = flat;
    tensors.push_back(std::move(t));
  }
  if (gatherState(net, criterion, netoptim.get(), critoptim.get(), 1, flat)) {
    Tensor t; t.name = "netoptim.accDelta(internal arena)"; t.kind = "state2"; t.shape = {(long)flat.size()}; t.data = flat;
    tensors.push_back(std::move(t));
  }
  std::ostringstream h;
  h << "{\"nfeat\": " << net.nFeat() << ", \"nlabel\": " << net.nLabel() << ", \"criterion\": " << jstr(critName(criterion))
    << ", \"optim\": [" << jstr(netoptim ? netoptim->kind() : "sgd") << ", " << jstr(critoptim ? critoptim->kind() : "sgd") << "]"
    << ", \"arch_sha256\": " << jstr(net.archSha_) << ", \"step\": " << net.step() << ", \"flags\": {" << jstr("version") << ": " << jstr(version);
  for (auto& kv : config)
    if (kv.first != "version") h << ", " << jstr(kv.first) << ": " << jstr(kv.second);
  h << "}, \"tensors\": [";
  for (size_t i = 0; i < tensors.size(); ++i) {
    h << (i ? ", " : "") << "{\"name\": " << jstr(tensors[i].name) << ", \"kind\": " << jstr(tensors[i].kind) << ", \"numel\": " << tensors[i].data.size()
      << ", \"shape\": [";
    for (size_t k = 0; k < tensors[i].shape.size(); ++k) h << (k ? ", " : "") << tensors[i].shape[k];
    h << "]}";
  }
  h << "]}";
  const std::string hdr = h.str();
  const std::string tmp = path + ".tmp";
  {
    std::ofstream f(tmp, std::ios::binary);
    if (!f) throw std::runtime_error("cannot write " + tmp);
    f.write("W2LAMD01", 8);
    const uint32_t n = (uint32_t)hdr.size();
    f.write((const char*)&n, 4);
    f.write(hdr.data(), (std::streamsize)hdr.size());
    size_t pos = 12 + hdr.size();
    static const char zeros[16] = {0};
    for (auto& t : tensors) {
      const size_t pad = (pos + 15) / 16 * 16 - pos;
      f.write(zeros, (std::streamsize)pad);
      f.write((const char*)t.data.data(), (std::streamsize)(t.data.size() * 4));
      pos += pad + t.data.size() * 4;
    }
    if (!f) throw std::runtime_error("cannot write " + tmp);
  }
  if (rename(tmp.c_str(), path.c_str()) != 0) throw std::runtime_error("cannot publish " + path);   // never a half-written model_last.bin
}

void Serializer::load(const std::string& path, std::string& version, Config& config) {
  configFromHeader(readFile(path, true).header, version, config);
}
void Serializer::load(const std::string& path, std::string& version, Config& config, const std::shared_ptr<fl::Module>& network,
                      const std::shared_ptr<fl::Module>& criterion) {
  loadModel(path, version, config, network, criterion, nullptr, nullptr);
}
void Serializer::load(const std::string& path, std::string& version, Config& config, const std::shared_ptr<fl::Module>& network,
                      const std::shared_ptr<fl::Module>& criterion, const std::shared_ptr<fl::FirstOrderOptimizer>& netoptim,
                      const std::shared_ptr<fl::FirstOrderOptimizer>& critoptim) {
  loadModel(path, version, config, network, criterion, netoptim.get(), critoptim.get());
}

}  // namespace runtime
}  // namespace pkg
}  // namespace fl
