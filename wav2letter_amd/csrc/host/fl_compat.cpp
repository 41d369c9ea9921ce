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

std::vector<Variable> Sequential::forward(const std::vector<Variable>& inputs) {
  std::vector<Variable> cur = inputs;
  for (auto& m : modules_) cur = m->forward(cur);
  return cur;
}
std::string Sequential::prettyString() const {
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
SGDOptimizer::SGDOptimizer(const std::vector<Variable>& params, double lr, double momentum, double weightDecay, bool useNesterov)
    : FirstOrderOptimizer(params, lr), mu_(momentum), wd_(weightDecay), nesterov_(useNesterov) {
  if (wd_ != 0 || nesterov_) throw std::invalid_argument("fl_compat SGDOptimizer: weight decay / Nesterov are not on the hot path of the recipes");
  if (mu_ != 0)
    for (auto& p : parameters_) velocities_.push_back(af::constant(0.0, p.dims(), af::f32));
}
void SGDOptimizer::step() {
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
  for (auto& p : parameters_) variance_.push_back(af::constant(0.0, p.dims(), af::f32));
}
void AdagradOptimizer::step() {
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
  for (auto& p : parameters_) {
    accGrad_.push_back(af::constant(0.0, p.dims(), af::f32));
    accDelta_.push_back(af::constant(0.0, p.dims(), af::f32));
  }
}
void AdadeltaOptimizer::step() {
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
  for (auto& p : params) {
    if (!p.isGradAvailable()) continue;
    w2l::w2lCheck(w2l_sumsq(p.grad().array().device<float>(), (size_t)p.elements(), d, first ? 1 : 0, S()), "clipGradNorm");
    first = false;
  }
  if (first) return 0.0;
  double ss = 0;
  w2l::hipCheck(hipMemcpyAsync(&ss, d, sizeof(double), hipMemcpyDeviceToHost, S()), "clipGradNorm");
  w2l::hipCheck(hipStreamSynchronize(S()), "clipGradNorm");
  const double norm = std::sqrt(ss);
  const double scale = maxNorm / (norm + 1e-6);
  if (scale < 1.0)
    for (auto& p : params)
      if (p.isGradAvailable()) {
        float* g = p.grad().array().device<float>();
        w2l::w2lCheck(w2l_axpy(g, g, (size_t)p.elements(), (float)(scale - 1.0), S()), "clipGradNorm");
      }
  return norm;
}

// ------------------------------------------------------------------------------------------------ data parallelism
// RCCL through dlopen: the types come from <rccl/rccl.h>, the entry points are resolved when initDistributed runs, so
// neither the library nor a single-GPU process depends on librccl.so (a Python process that already carries torch's
// RCCL gets that copy: RTLD_NOLOAD first).
}  // namespace fl
#include <rccl/rccl.h>
#include <unistd.h>
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
};
Rccl& rccl() { static Rccl r; return r; }
void ncclCheck(ncclResult_t st, const char* what) {
  if (st != ncclSuccess) throw std::runtime_error(std::string(what) + ": " + (rccl().GetErrorString ? rccl().GetErrorString(st) : "RCCL error"));
}
template <class F> void bindSym(void* lib, const char* name, F& f) {
  f = (F)dlsym(lib, name);
  if (!f) throw std::runtime_error(std::string("librccl.so lacks ") + name);
}
void allReduceRaw(float* p, size_t n, double scale) {
  Rccl& r = rccl();
  if (!r.comm || !n) return;   // (a communicator of ONE rank still goes through RCCL: the plumbing is what a 1-GPU box can test)
  ncclCheck(r.AllReduce(p, p, n, ncclFloat32, ncclSum, r.comm, (hipStream_t)S()), "ncclAllReduce");
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
  if (!rccl().comm) return;
  for (auto& p : module->params()) allReduceRaw(p.array().device<float>(), (size_t)p.elements(), 1.0 / rccl().size);
}
void barrier() {
  if (!rccl().comm) return;
  static std::shared_ptr<void> one = devAlloc(sizeof(float));
  w2l::hipCheck(hipMemsetAsync(one.get(), 0, sizeof(float), (hipStream_t)S()), "barrier");
  allReduceRaw((float*)one.get(), 1, 1.0);
  w2l::hipCheck(hipStreamSynchronize((hipStream_t)S()), "barrier");
}

CoalescingReducer::CoalescingReducer(double scale, bool async, bool contiguous) : scale_(scale), async_(async), contiguous_(contiguous) {}
CoalescingReducer::~CoalescingReducer() {}
void CoalescingReducer::add(Variable& var) {
  if (var.type() != af::f32) throw std::invalid_argument("CoalescingReducer: f32 gradients only");
  float* p = var.array().device<float>();
  const size_t n = (size_t)var.elements();
  if (!n) return;
  if (!spans_.empty() && spans_.back().ptr + spans_.back().n == p) spans_.back().n += n;   // the next piece of the same arena
  else spans_.push_back({p, n});
}
void CoalescingReducer::finalize() {
  lastCollectives_ = 0;
  for (auto& sp : spans_) {
    allReduceRaw(sp.ptr, sp.n, scale_);
    ++lastCollectives_;
  }
  spans_.clear();
}

namespace pkg {
namespace runtime {
void initDistributed(int worldRank, int worldSize, int maxDevicesPerNode, const std::string& rndvFilepath) {
  Rccl& r = rccl();
  if (r.comm) throw std::runtime_error("initDistributed called twice");
  if (worldSize < 1 || worldRank < 0 || worldRank >= worldSize) throw std::invalid_argument("initDistributed: bad rank / world size");
  int ndev = 0;
  w2l::hipCheck(hipGetDeviceCount(&ndev), "hipGetDeviceCount");
  const int perNode = std::max(1, std::min(maxDevicesPerNode > 0 ? maxDevicesPerNode : ndev, ndev));
  w2l::hipCheck(hipSetDevice(worldRank % perNode), "hipSetDevice");
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
  // publication: a reader only accepts a record published no earlier than two minutes before its own start, rank 0
  // removes whatever an earlier run left under the name before it publishes, and removes its own record once every rank
  // has joined (ncclCommInitRank is collective) -- so a `continue` / re-run with the same --rndv_filepath never picks up
  // the previous run's ncclUniqueId (which would hang ncclCommInitRank).
  struct RndvRecord { unsigned long long magic; long long publishedNs; ncclUniqueId id; };
  constexpr unsigned long long kRndvMagic = 0x7732'6c72'6e64'7631ull;   // "w2lrndv1"
  auto nowNs = [] { timespec ts; clock_gettime(CLOCK_REALTIME, &ts); return (long long)ts.tv_sec * 1000000000ll + ts.tv_nsec; };
  const long long startNs = nowNs();
  RndvRecord rec;
  std::memset(&rec, 0, sizeof rec);
  ncclUniqueId& id = rec.id;
  const std::string path = rndvFilepath + "/w2l_nccl_id." + std::to_string(worldSize);
  if (worldSize > 1 && rndvFilepath.empty()) throw std::invalid_argument("initDistributed: --rndv_filepath is needed for world_size > 1");
  if (worldRank == 0) {
    ncclCheck(r.GetUniqueId(&id), "ncclGetUniqueId");
    if (worldSize > 1) {   // write to a temporary name, then rename: readers never see a partial file
      (void)unlink(path.c_str());   // a record left behind by a run that died during its rendezvous
      rec.magic = kRndvMagic;
      rec.publishedNs = nowNs();
      const std::string tmp = path + ".tmp";
      { std::ofstream f(tmp, std::ios::binary); f.write((const char*)&rec, sizeof rec); if (!f) throw std::runtime_error("cannot write " + tmp); }
      if (rename(tmp.c_str(), path.c_str()) != 0) throw std::runtime_error("cannot publish " + path);
    }
  } else {
    bool got = false;
    for (int tries = 0; tries < 6000 && !got; ++tries) {   // up to 10 minutes
      std::ifstream f(path, std::ios::binary);
      RndvRecord in;
      if (f && f.read((char*)&in, sizeof in) && in.magic == kRndvMagic && in.publishedNs >= startNs - 120ll * 1000000000ll) {
        rec = in;
        got = true;
      } else {
        usleep(100000);
      }
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
    if (in.dims(1) * in.dims(2) != nFeat_) throw std::invalid_argument("network forward: feature dimension != NFEAT");
    for (size_t i = 0; i < params_.size(); ++i)
      if (params_[i].array().device<float>() != (float*)paramArena_.get() + net_->params()[i].offset)
        throw std::logic_error("setParams on a planned network must write INTO the parameter (copy), not rebind it");
    if (B != B_ || T != T_) {
      const size_t fl = net_->plan(B, T, nFeat_);
      arena_ = devAlloc(fl * sizeof(float));
      arenaFloats_ = fl;
      B_ = B; T_ = T;
      dEm_ = devAlloc(((size_t)B * net_->outAct().T * nLabel_ + 64) * sizeof(float));
    }
    w2l::Ctx c;
    c.stream = S(); c.train = train_; c.seed = 0x9E3779B9u * (++step_);
    c.params = (float*)paramArena_.get(); c.grads = (float*)gradArena_.get();
    if (inputs.size() >= 2 && !inputs[1].isempty()) {  // inputSizes (1, B): padding mask of the Transformer blocks
      if (inputs[1].type() != af::f32 || inputs[1].elements() != B) throw std::invalid_argument("network forward: inputSizes must be f32 (1, B)");
      c.inputSizes = inputs[1].array().device<float>();
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

 private:
  std::shared_ptr<w2l::Sequential> net_;
  int nFeat_, nLabel_;
  std::shared_ptr<void> paramArena_, gradArena_, arena_, dEm_;
  size_t arenaFloats_ = 0;
  int B_ = 0, T_ = 0;
  uint32_t step_ = 0;
};

}  // namespace

std::shared_ptr<fl::Sequential> buildSequentialModuleFromText(const std::string& archText, int64_t nFeatures, int64_t nClasses) {
  return std::make_shared<PlannedNet>(w2l::buildSequentialFromText(archText, nFeatures, nClasses), nFeatures, nClasses);
}
std::shared_ptr<fl::Sequential> buildSequentialModule(const std::string& archfile, int64_t nFeatures, int64_t nClasses) {
  return buildSequentialModuleFromText(w2l::readFile(archfile), nFeatures, nClasses);
}

FlatView flatParameters(const std::shared_ptr<fl::Module>& network) {
  auto* p = dynamic_cast<PlannedNet*>(network.get());
  return p ? FlatView{p->paramPtr(), p->impl().paramFloats()} : FlatView{nullptr, 0};
}
void setMixedPrecision(const std::shared_ptr<fl::Module>& network, bool on) {
  if (auto* p = dynamic_cast<PlannedNet*>(network.get())) p->mixed_ = on;
}
FlatView flatGradients(const std::shared_ptr<fl::Module>& network) {
  auto* p = dynamic_cast<PlannedNet*>(network.get());
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
  st->impl->viterbiPath(c, B, T, N, input.device<float>(), path.device<int>(), st->workspace(B, T, N, 1), params_[0].array().device<float>());
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
