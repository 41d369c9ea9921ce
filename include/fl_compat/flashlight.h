/*
 * fl_compat/flashlight.h -- the Flashlight (<= 0.3.2) surface the wav2letter recipes' Trainer touches on the
 * acoustic-training hot path, re-hosted on libw2l_hip.so (hand-written HIP for gfx950; no ArrayFire).
 *
 * BASELINE.json north_star: "host code stays C++ and exposes the same fl::Module / SequenceCriterion /
 * fl::app::asr::Trainer surface so recipes/ configs load unchanged".  What is mirrored, with the reference file:line
 * that shows each shape (Flashlight itself is un-vendored, so the in-repo USES are the witnesses):
 *
 *   af::dim4, af::array (device buffer + dims in ArrayFire order, d0 fastest)    every call site below
 *   fl::Variable, fl::input / fl::noGrad, .array() .dims() .grad() .backward()   recipes/slimIPL/src/Train.cpp:1454-1470, :1718-1720
 *   fl::Module / Container / Sequential: forward(std::vector<Variable>),
 *       params(), param(i), setParams(), train(), eval(), prettyString()         recipes/slimIPL/100h_supervised.cpp:37-75; Train.cpp:396-401
 *   fl::pkg::runtime::ModulePlugin(path).arch(nFeature, nLabel)
 *       -> dlopen + extern "C" fl::Module* createModule(int64_t, int64_t)        recipes/slimIPL/100h_supervised.cpp:84-87; Train.cpp:390-395
 *   fl::pkg::speech::buildSequentialModule(archfile, nFeatures, nClasses)        recipes/joint_training_vox_populi/cpc/SequentialBuilder.h:23-26
 *   fl::pkg::speech::SequenceCriterion { forward({emission (N,T,B), target (L,B)}) -> {loss (B)},
 *       viterbiPath(input, inputSize), viterbiPathWithTarget, prettyString }     recipes/joint_training_vox_populi/cpc/CPCCriterion.h:30-50
 *   ASGLoss(N, scaleMode, transdiag), CTCLoss(scaleMode), getCriterionScaleMode  recipes/slimIPL/src/Train.cpp:389, :406-410
 *   fl::SGDOptimizer(params, lr, momentum, wd), zeroGrad(), step(), setLr();
 *       fl::clipGradNorm(params, maxNorm)                                         recipes/slimIPL/src/Train.cpp:577-582, :1718, :1791-1802
 *
 * Not ArrayFire: af::array here is only a ref-counted device buffer with dims and a dtype (f32 / s32) -- enough to
 * carry tensors across this boundary; arithmetic on arrays belongs to the kernels behind the modules.  Memory order
 * is ArrayFire's (column-major, d0 fastest), so an (N, T, B) emission array IS the [B][T][N] buffer of the C ABI.
 *
 * The network built from an arch file is ONE planned pipeline (static layer list, activations in one arena, the
 * parameters / gradients in flat arenas: w2l_host.hpp); fl::Sequential::modules() lists its lines for prettyString
 * and parameter bookkeeping, the forward / backward run as a whole.  A plugin module (createModule) composes such
 * pipelines -- from arch text, or (round 5) from LAYER OBJECTS the way the reference's plugins are written:
 * fl::View / Reorder / Dropout / ReLU / GatedLinearUnit / LayerNorm / Linear / Conv2D / Pool2D / Transformer / TDSBlock /
 * WeightNorm, constructed with the reference's arguments and add()ed to an fl::Sequential, are their lines of the arch
 * grammar, and the Sequential plans one pipeline out of them on first use (tests/cpp/plugin_layers.cpp).  What is NOT
 * provided is per-layer execution (`modules()[i]->forward(x)`, a custom forward() that interleaves af:: arithmetic, as
 * recipes/slimIPL/100h_supervised.cpp:44-66 does for its padding mask): a free-form per-op autograd is outside the hot
 * path; the padding mask that forward builds is what the planned pipeline's Transformer blocks derive from inputSizes.
 */
#pragma once
#include <stdint.h>
#include <cstdio>
#include <initializer_list>

#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <unordered_map>
#include <vector>

#define FL_COMPAT_API __attribute__((visibility("default")))

namespace af {
typedef long long dim_t;

class FL_COMPAT_API dim4 {
 public:
  dim_t dims[4];
  dim4(dim_t d0 = 1, dim_t d1 = 1, dim_t d2 = 1, dim_t d3 = 1) : dims{d0, d1, d2, d3} {}
  dim_t elements() const { return dims[0] * dims[1] * dims[2] * dims[3]; }
  dim_t ndims() const {
    if (elements() == 0) return 0;
    for (int i = 3; i > 0; --i) if (dims[i] != 1) return i + 1;
    return 1;
  }
  dim_t& operator[](int i) { return dims[i]; }
  const dim_t& operator[](int i) const { return dims[i]; }
  bool operator==(const dim4& o) const { return dims[0] == o.dims[0] && dims[1] == o.dims[1] && dims[2] == o.dims[2] && dims[3] == o.dims[3]; }
  bool operator!=(const dim4& o) const { return !(*this == o); }
};

enum dtype { f32 = 0, s32 = 5 };  // ArrayFire's enumerator values

// ref-counted device buffer; copies share storage (like af::array handles)
class FL_COMPAT_API array {
 public:
  array() {}
  explicit array(const dim4& dims, dtype ty = f32);                       // uninitialised
  array(const dim4& dims, const float* host);                             // afHost source
  array(const dim4& dims, const int* host);
  // fl_compat extension: a view over memory somebody else owns (`owner` keeps it alive; may be null)
  static array wrap(void* dev, const dim4& dims, dtype ty, std::shared_ptr<void> owner = nullptr);
  dim4 dims() const { return dims_; }
  dim_t dims(int i) const { return dims_[i]; }
  dim_t elements() const { return ptr_ ? dims_.elements() : 0; }
  dtype type() const { return type_; }
  bool isempty() const { return elements() == 0; }
  size_t bytes() const { return (size_t)elements() * 4; }
  template <class T> T* device() const { return (T*)ptr_; }                // raw device pointer (no lock bookkeeping)
  void unlock() const {}
  template <class T> void host(T* out) const { hostCopy((void*)out); }     // synchronous device -> host
  template <class T> T scalar() const { T v[1]; if (elements() < 1) throw std::invalid_argument("scalar() of an empty array"); firstElement((void*)v); return v[0]; }
  array copy() const;                                                     // deep copy

 private:
  void hostCopy(void* out) const;
  void firstElement(void* out) const;
  std::shared_ptr<void> owner_;
  void* ptr_ = nullptr;
  dim4 dims_{0, 0, 0, 0};
  dtype type_ = f32;
};

FL_COMPAT_API array constant(double v, const dim4& dims, dtype ty = f32);
FL_COMPAT_API void sync();          // wait for the stream every fl_compat call enqueues on
}  // namespace af

namespace fl {

FL_COMPAT_API void* currentStream();                 // hipStream_t the facade enqueues on (default: a private non-blocking stream)
FL_COMPAT_API void setCurrentStream(void* stream);

class FL_COMPAT_API Variable {
 public:
  using GradFunc = std::function<void(std::vector<Variable>& inputs, const Variable& gradOutput)>;
  Variable() : s_(std::make_shared<Shared>()) {}
  Variable(af::array data, bool calcGrad) : s_(std::make_shared<Shared>()) { s_->data = std::move(data); s_->calcGrad = calcGrad; }
  Variable(af::array data, std::vector<Variable> inputs, GradFunc gradFunc);
  af::array& array() const { return s_->data; }
  Variable& grad() const;
  bool isCalcGrad() const { return s_->calcGrad; }
  bool isGradAvailable() const { return s_->grad != nullptr; }
  af::dim4 dims() const { return s_->data.dims(); }
  af::dim_t dims(int i) const { return s_->data.dims(i); }
  af::dim_t elements() const { return s_->data.elements(); }
  af::dtype type() const { return s_->data.type(); }
  bool isempty() const { return s_->data.isempty(); }
  void setCalcGrad(bool b) { s_->calcGrad = b; }
  void addGrad(const Variable& g);            // accumulates (axpy on the device) like Flashlight
  void zeroGrad() { s_->grad.reset(); }
  void backward(bool retainGraph = false);    // d/d(this) with an all-ones seed (a loss vector of B utterances: sum)
  void backward(const Variable& grad, bool retainGraph = false);
  template <class T> T scalar() const { return s_->data.scalar<T>(); }
  template <class T> void host(T* p) const { s_->data.host(p); }

 private:
  struct Shared {
    af::array data;
    std::unique_ptr<Variable> grad;
    bool calcGrad = false;
    std::vector<Variable> inputs;
    GradFunc gradFunc;
  };
  std::shared_ptr<Shared> s_;
  friend struct VariableAccess;
};

// scalar arithmetic the Trainer uses on losses and gradients (`p.grad() = p.grad() / totalBatchSize`,
// recipes/slimIPL/src/Train.cpp:1748-1784): results are plain (no-grad) Variables
FL_COMPAT_API Variable operator*(const Variable& v, double s);
FL_COMPAT_API Variable operator/(const Variable& v, double s);

inline Variable input(const af::array& a) { return Variable(a, false); }
inline Variable noGrad(const af::array& a) { return Variable(a, false); }
inline Variable param(const af::array& a) { return Variable(a, true); }

class FL_COMPAT_API Module {
 public:
  virtual ~Module() {}
  virtual std::vector<Variable> params() const { return params_; }   // (virtual: a Sequential of layer objects plans itself on first use)
  Variable param(int i) const { return params().at(i); }   // (through params(): a Sequential of layer objects plans itself on first use)
  virtual void setParams(const Variable& var, int position);
  virtual void train() { train_ = true; }
  virtual void eval() { train_ = false; }
  bool isTrainMode() const { return train_; }
  void zeroGrad() { for (auto& p : params_) p.zeroGrad(); }
  virtual std::vector<Variable> forward(const std::vector<Variable>& inputs) = 0;
  std::vector<Variable> operator()(const std::vector<Variable>& inputs) { return forward(inputs); }
  virtual std::string prettyString() const = 0;

 protected:
  std::vector<Variable> params_;
  bool train_ = true;
};

class FL_COMPAT_API Container : public Module {
 public:
  virtual void add(std::shared_ptr<Module> m);               // the child's parameters join params() (virtual: a layer object added through a
                                                             //  Container reference to an fl::Sequential still contributes its arch line)
  std::shared_ptr<Module> module(int i) const { return modules_.at(i); }
  std::vector<std::shared_ptr<Module>> modules() const { return modules_; }
  void train() override { train_ = true; for (auto& m : modules_) m->train(); }
  void eval() override { train_ = false; for (auto& m : modules_) m->eval(); }
  void setParams(const Variable& var, int position) override;

 protected:
  std::vector<std::shared_ptr<Module>> modules_;
  std::vector<int> childOfParam_, childIndexOfParam_;
};

// A layer AS AN OBJECT, the way a reference model plugin builds its network (recipes/slimIPL/100h_supervised.cpp:24-43:
// `convFrontend_->add(std::make_shared<fl::Conv2D>(nFeature, 1536, 7, 1, 3, 1, -1, 0, 1, 1))`, `fl::View`, `fl::LayerNorm`,
// `fl::GatedLinearUnit`, `fl::Dropout`, `fl::Reorder`, `fl::Transformer`, `fl::Linear`).  Here a layer object IS its line of the
// arch grammar (cpc/SequentialBuilder.cpp:92-626, the same constructor arguments in the same order): add()ed to an
// fl::Sequential it appends that line, and the Sequential plans ONE pipeline out of its lines on first use (params() /
// forward()), exactly as buildSequentialModule does for an arch file.  A layer object cannot run on its own: forward() on it
// throws -- per-layer execution with a free-form autograd is outside the hot path (see the header comment).
class FL_COMPAT_API ArchLayer : public Module {
 public:
  explicit ArchLayer(std::string line) : line_(std::move(line)) {}
  const std::string& archLine() const { return line_; }
  std::vector<Variable> forward(const std::vector<Variable>&) override {
    throw std::logic_error("fl_compat: a layer object runs as part of the fl::Sequential it was added to (one planned pipeline): " + line_);
  }
  std::string prettyString() const override { return line_; }

 protected:
  static std::string join(const char* tok, std::initializer_list<double> v) {
    std::string s(tok);
    for (double x : v) {
      char b[40];
      if (x == (double)(long long)x) snprintf(b, sizeof b, " %lld", (long long)x); else snprintf(b, sizeof b, " %.9g", x);
      s += b;
    }
    return s;
  }
  std::string line_;
};
class View : public ArchLayer { public: explicit View(const af::dim4& d) : ArchLayer(join("V", {(double)d[0], (double)d[1], (double)d[2], (double)d[3]})) {} };
class Reorder : public ArchLayer { public: Reorder(int d0, int d1, int d2 = 2, int d3 = 3) : ArchLayer(join("RO", {(double)d0, (double)d1, (double)d2, (double)d3})) {} };
class Dropout : public ArchLayer { public: explicit Dropout(double p = 0.5) : ArchLayer(join("DO", {p})) {} };
class ReLU : public ArchLayer { public: ReLU() : ArchLayer("R") {} };
class GatedLinearUnit : public ArchLayer { public: explicit GatedLinearUnit(int dim) : ArchLayer(join("GLU", {(double)dim})) {} };
class LayerNorm : public ArchLayer {
 public:
  explicit LayerNorm(const std::vector<int>& axes) : ArchLayer(lineOf(axes)) {}
  explicit LayerNorm(int axis) : ArchLayer(lineOf({axis})) {}   // fl::LayerNorm(int axis, ...)
 private:
  static std::string lineOf(const std::vector<int>& axes) { std::string s("LN"); for (int a : axes) s += " " + std::to_string(a); return s; }
};
class Linear : public ArchLayer {
 public:
  Linear(int in, int out, bool bias = true) : ArchLayer(join("L", {(double)in, (double)out})) {
    if (!bias) throw std::invalid_argument("fl_compat: fl::Linear without bias is not in the arch grammar");
  }
};
// fl::Conv2D(nIn, nOut, wx, wy, sx, sy, px, py, dx, dy, bias, groups): wx along time (ArrayFire dim 0); px / py = -1 is PaddingMode::SAME
class Conv2D : public ArchLayer {
 public:
  Conv2D(int nIn, int nOut, int wx, int wy, int sx = 1, int sy = 1, int px = 0, int py = 0, int dx = 1, int dy = 1, bool bias = true, int groups = 1)
      : ArchLayer(dx == 1 && dy == 1 ? join("C2", {(double)nIn, (double)nOut, (double)wx, (double)wy, (double)sx, (double)sy, (double)px, (double)py})
                                     : join("C2", {(double)nIn, (double)nOut, (double)wx, (double)wy, (double)sx, (double)sy, (double)px, (double)py, (double)dx, (double)dy})) {
    if (!bias || groups != 1) throw std::invalid_argument("fl_compat: fl::Conv2D without bias / with groups is not in the arch grammar");
  }
};
enum class PoolingMode { MAX = 0 };
class Pool2D : public ArchLayer {
 public:
  Pool2D(int wx, int wy, int sx = 1, int sy = 1, int px = 0, int py = 0, PoolingMode mode = PoolingMode::MAX)
      : ArchLayer(px == 0 && py == 0 ? join("M", {(double)wx, (double)wy, (double)sx, (double)sy}) : join("M", {(double)wx, (double)wy, (double)sx, (double)sy, (double)px, (double)py})) { (void)mode; }
};
// fl::Transformer(modelDim, headDim, mlpDim, nHeads, bptt, pDropout, pLayerdrop, useMask, preLN)
class Transformer : public ArchLayer {
 public:
  Transformer(int modelDim, int headDim, int mlpDim, int nHeads, int bptt, float pDropout, float pLayerdrop, bool useMask = false, bool preLN = false)
      : ArchLayer(join("TR", {(double)modelDim, (double)mlpDim, (double)nHeads, (double)bptt, (double)pDropout, (double)pLayerdrop})) {
    if (headDim * nHeads != modelDim || useMask || preLN)
      throw std::invalid_argument("fl_compat: fl::Transformer with headDim * nHeads != modelDim, a causal mask or pre-LN is not in the arch grammar");
  }
};
// fl::TDSBlock(channels, kernelSize, width, dropout, innerLinearDim)
class TDSBlock : public ArchLayer {
 public:
  TDSBlock(int c, int kw, int h, double dropout = 0, int innerLinearDim = 0)
      : ArchLayer(innerLinearDim ? join("TDS", {(double)c, (double)kw, (double)h, dropout, (double)innerLinearDim}) : join("TDS", {(double)c, (double)kw, (double)h, dropout})) {}
};
// fl::WeightNorm(module, dim)
class WeightNorm : public ArchLayer {
 public:
  WeightNorm(const std::shared_ptr<ArchLayer>& child, int dim) : ArchLayer("WN " + std::to_string(dim) + " " + child->archLine()) {}
  WeightNorm(const ArchLayer& child, int dim) : ArchLayer("WN " + std::to_string(dim) + " " + child.archLine()) {}
};

// user-composed chain: output of module i feeds module i + 1.  With layer objects (above) the chain is one planned pipeline:
// the lines of the layers added so far, planned on first use; layer objects and other modules do not mix in one Sequential.
class FL_COMPAT_API Sequential : public Container {
 public:
  void add(std::shared_ptr<Module> m) override;               // a layer object contributes its arch line
  template <class T, class = typename std::enable_if<std::is_base_of<Module, T>::value>::type>
  void add(const T& layer) { add(std::static_pointer_cast<Module>(std::make_shared<T>(layer))); }   // fl's add(const T&)
  std::vector<Variable> params() const override;
  void setParams(const Variable& var, int position) override;
  void train() override;
  void eval() override;
  std::vector<Variable> forward(const std::vector<Variable>& inputs) override;
  Variable forward(const Variable& in) { return forward(std::vector<Variable>{in}).front(); }
  std::string prettyString() const override;
  std::shared_ptr<Sequential> planned() const;                // the pipeline behind a Sequential of layer objects (null otherwise)
  void setInputFeatures(int nFeat) { inputFeatures_ = nFeat; } // fl_compat extension: NFEAT of the (T, NFEAT, 1, B) input when the first layer does not say it

 private:
  void materialize();
  std::string archText_;                                      // lines of the layer objects added so far
  std::shared_ptr<Sequential> planned_;
  int inputFeatures_ = 0;
};

// fl::SpecAugment as the Trainer instantiates it from --saug_start_update (recipes/slimIPL/src/Train.cpp:1026-1048, applied
// at :1453-1461): frequency / time masking with zeros on the features (T, NFEAT, 1, B); identity in eval mode.  Same kernel as
// the SAUG arch token (w2l_specaugment_inplace; one mask set per batch).  Time warping (tWarpW) is not applied by the
// reference's own module either in this configuration (the first argument is the filterbank count).
class FL_COMPAT_API SpecAugment : public Module {
 public:
  SpecAugment(int tWarpW, int fMaskF, int nFMask, int tMaskT, float tMaskP, int nTMask);
  std::vector<Variable> forward(const std::vector<Variable>& inputs) override;
  std::string prettyString() const override;
  uint32_t calls() const { return calls_; }          // fl_compat extension: the mask stream position (checkpoints)
  void setCalls(uint32_t n) { calls_ = n; }

 private:
  int fMaskF_, nFMask_, tMaskT_, nTMask_;
  float tMaskP_;
  uint32_t calls_ = 0;
};

// ---- optimizers (recipes/slimIPL/src/Train.cpp:577-582: SGDOptimizer(params, lr, momentum, weightdecay))
class FL_COMPAT_API FirstOrderOptimizer {
 public:
  FirstOrderOptimizer(const std::vector<Variable>& params, double lr) : parameters_(params), lr_(lr) {}
  virtual ~FirstOrderOptimizer() {}
  virtual void step() = 0;
  double getLr() const { return lr_; }
  void setLr(double lr) { lr_ = lr; }
  virtual void zeroGrad() { for (auto& p : parameters_) p.zeroGrad(); }
  virtual std::string prettyString() const = 0;
  // fl_compat extension (checkpoints, fl::pkg::runtime::Serializer): "sgd" | "adagrad" | "adadelta", and the per-parameter
  // state arrays -- slot 0: SGD velocity / Adagrad variance / Adadelta accGrad, slot 1: Adadelta accDelta (empty: stateless)
  virtual const char* kind() const = 0;
  virtual std::vector<std::vector<af::array>*> state() { return {}; }
  const std::vector<Variable>& parameters() const { return parameters_; }

 protected:
  std::vector<Variable> parameters_;
  double lr_;
  // parameters that are exactly the slices of ONE planned network's flat parameter arena: the optimizer's state arrays are views of
  // flat buffers with the same offsets and step() is a single launch over (parameters, gradients, state) -- the reference walks
  // the parameters one by one (a few hundred launches per update); anything else falls back to that walk
  float* flatParams_ = nullptr;
  float* flatState_[2] = {nullptr, nullptr};
  size_t flatFloats_ = 0;
  std::vector<size_t> flatOffsets_;
  void flatStateViews(int slot, std::vector<af::array>& views);   // allocates flatState_[slot] (zeros) and fills `views`
  bool flatNow(float*& gradBase, const std::vector<af::array>* s0, const std::vector<af::array>* s1) const;
};

// every gradient of `params` multiplied by `s` IN PLACE (Train.cpp:1748-1760 writes `p.grad() = p.grad() / totalBatchSize`, one
// temporary per parameter); gradients that are the slices of one planned network's gradient arena are scaled as the arena
FL_COMPAT_API void scaleGradients(const std::vector<Variable>& params, double s);

class FL_COMPAT_API SGDOptimizer : public FirstOrderOptimizer {
 public:
  SGDOptimizer(const std::vector<Variable>& params, double lr, double momentum = 0, double weightDecay = 0, bool useNesterov = false);
  void step() override;
  std::string prettyString() const override;
  const char* kind() const override { return "sgd"; }
  std::vector<std::vector<af::array>*> state() override { return velocities_.empty() ? std::vector<std::vector<af::array>*>{} : std::vector<std::vector<af::array>*>{&velocities_}; }

 private:
  double mu_, wd_;
  bool nesterov_;
  std::vector<af::array> velocities_;
};

// --netoptim=adagrad (recipes/sota/2019/librivox/train_am_transformer_ctc.cfg:25-26): variance += g^2, p -= lr g / (sqrt(variance) + eps)
class FL_COMPAT_API AdagradOptimizer : public FirstOrderOptimizer {
 public:
  AdagradOptimizer(const std::vector<Variable>& params, double lr, double eps = 1e-8, double weightDecay = 0);
  void step() override;
  std::string prettyString() const override;
  const char* kind() const override { return "adagrad"; }
  std::vector<std::vector<af::array>*> state() override { return {&variance_}; }

 private:
  double eps_;
  std::vector<af::array> variance_;
};

// --netoptim=adadelta (recipes/sota/2019/librispeech/train_am_transformer_ctc.cfg:23-26)
class FL_COMPAT_API AdadeltaOptimizer : public FirstOrderOptimizer {
 public:
  AdadeltaOptimizer(const std::vector<Variable>& params, double lr = 1.0, double rho = 0.9, double eps = 1e-8, double weightDecay = 0);
  void step() override;
  std::string prettyString() const override;
  const char* kind() const override { return "adadelta"; }
  std::vector<std::vector<af::array>*> state() override { return {&accGrad_, &accDelta_}; }

 private:
  double rho_, eps_;
  std::vector<af::array> accGrad_, accDelta_;
};

// global-norm clipping over the gradients that are available; returns the norm before clipping
FL_COMPAT_API double clipGradNorm(const std::vector<Variable>& params, double maxNorm);

// ---- data parallelism (recipes/slimIPL/src/Train.cpp:188-199, :1078-1079, :1721-1747): one process per GPU, RCCL over xGMI.
// librccl.so is dlopen()ed by initDistributed only: a single-GPU run never touches it.
FL_COMPAT_API int getWorldRank();
FL_COMPAT_API int getWorldSize();
FL_COMPAT_API void allReduce(af::array& arr, double scale = 1.0);           // in place, sum over ranks (then * scale)
FL_COMPAT_API void allReduce(Variable& var, double scale = 1.0);
FL_COMPAT_API void allReduceParameters(const std::shared_ptr<const Module>& module);   // average: replicas start identical
FL_COMPAT_API void barrier();

class FL_COMPAT_API Reducer {
 public:
  virtual ~Reducer() {}
  virtual void add(Variable& var) = 0;
  virtual void finalize() = 0;
};
// fl::CoalescingReducer(scale, async, contiguous): gradients are added as backward produces them and reduced in
// buckets; here the parameters of a planned network already sit in ONE flat arena, so adjacent gradients coalesce into
// a single ncclAllReduce over the whole arena (the reference issues dozens of ~20 MB buckets)
class FL_COMPAT_API CoalescingReducer : public Reducer {
 public:
  CoalescingReducer(double scale, bool async, bool contiguous);
  ~CoalescingReducer() override;
  void add(Variable& var) override;
  void finalize() override;
  size_t lastCollectives() const { return lastCollectives_; }   // fl_compat extension: ncclAllReduce calls of the last finalize()
  size_t lastOverlapped() const { return lastOverlapped_; }     // ... of which were issued on the side stream behind a bucket event

 private:
  double scale_;
  bool async_, contiguous_;
  struct Span { float* ptr; size_t n; };
  std::vector<Span> spans_;
  size_t lastCollectives_ = 0, lastOverlapped_ = 0;
};

namespace pkg {
namespace runtime {
// fl::pkg::runtime::initDistributed(worldRank, worldSize, maxDevicesPerNode, rndvFilepath) (Train.cpp:189-194): binds
// this process to GPU worldRank % maxDevicesPerNode and creates the RCCL communicator; the ncclUniqueId travels through
// the file <rndvFilepath>/w2l_nccl_id.<worldSize> written by rank 0 (the reference's file-system rendezvous)
FL_COMPAT_API void initDistributed(int worldRank, int worldSize, int maxDevicesPerNode, const std::string& rndvFilepath);
// fl::pkg::runtime::Serializer::save(filename, version, config, network, criterion, netoptim, critoptim) / load(...)
// (recipes/slimIPL/src/Train.cpp:132-173 header-only load of `continue` / `fork`, :452-463 model loads, :767-790 saves).
// The reference's container is cereal (needs Flashlight to read); this one is the documented W2LAMD01 layout of
// wav2letter_amd/checkpoint.py -- network tensors in the REFERENCE's parameter order and array layouts, the ASG
// transitions, the optimizer state as flat arenas, `config` in the JSON header -- written and read by both languages.
struct FL_COMPAT_API Serializer {
  using Config = std::unordered_map<std::string, std::string>;
  static void save(const std::string& path, const std::string& version, const Config& config, const std::shared_ptr<fl::Module>& network,
                   const std::shared_ptr<fl::Module>& criterion, const std::shared_ptr<fl::FirstOrderOptimizer>& netoptim,
                   const std::shared_ptr<fl::FirstOrderOptimizer>& critoptim);
  static void load(const std::string& path, std::string& version, Config& config);
  static void load(const std::string& path, std::string& version, Config& config, const std::shared_ptr<fl::Module>& network,
                   const std::shared_ptr<fl::Module>& criterion);
  static void load(const std::string& path, std::string& version, Config& config, const std::shared_ptr<fl::Module>& network,
                   const std::shared_ptr<fl::Module>& criterion, const std::shared_ptr<fl::FirstOrderOptimizer>& netoptim,
                   const std::shared_ptr<fl::FirstOrderOptimizer>& critoptim);
};
// dlopen(path, RTLD_LAZY) + dlsym("createModule"): extern "C" fl::Module* createModule(int64_t nFeature, int64_t nLabel)
// returns an OWNING raw pointer (recipes/slimIPL/100h_supervised.cpp:84-87; loader call Train.cpp:390-395).
// A name that ends in ".arch" is not a plugin: arch() then goes through buildSequentialModule, which is what the
// reference's `--arch` flag does when it names an arch file.
class FL_COMPAT_API ModulePlugin {
 public:
  explicit ModulePlugin(const std::string& name);
  ~ModulePlugin();
  std::shared_ptr<fl::Module> arch(int64_t nFeatures, int64_t nClasses);

 private:
  std::string name_;
  void* handle_ = nullptr;
  fl::Module* (*create_)(int64_t, int64_t) = nullptr;
};
}  // namespace runtime

namespace speech {
enum class CriterionScaleMode { NONE = 0, INPUT_SZ = 1, INPUT_SZ_SQRT = 2, TARGET_SZ = 3, TARGET_SZ_SQRT = 4 };
FL_COMPAT_API CriterionScaleMode getCriterionScaleMode(const std::string& onorm, bool sqnorm);

// arch file -> network.  forward({features (T, NFEAT, 1, B) f32 [, inputSizes (1, B)]}) -> {emissions (NLABEL, T', B)}
// (inputSizes may carry one more entry, (1, B + 1): the size the T frames correspond to when the batch is padded beyond its
// longest utterance -- the denominator of the Transformer padding mask; the reference pads to the longest only)
FL_COMPAT_API std::shared_ptr<fl::Sequential> buildSequentialModule(const std::string& archfile, int64_t nFeatures, int64_t nClasses);
FL_COMPAT_API std::shared_ptr<fl::Sequential> buildSequentialModuleFromText(const std::string& archText, int64_t nFeatures, int64_t nClasses);

class FL_COMPAT_API SequenceCriterion : public fl::Container {
 public:
  // inputs {emission (N, T, B) f32, target (L, B) s32 (padded with negative values)} -> {loss (B)}
  std::vector<Variable> forward(const std::vector<Variable>& inputs) override = 0;
  virtual af::array viterbiPath(const af::array& input, const af::array& inputSize = af::array()) = 0;              // (T, B) s32
  virtual af::array viterbiPathWithTarget(const af::array& input, const af::array& target,
                                          const af::array& inputSizes = af::array(), const af::array& targetSizes = af::array()) = 0;
  std::string prettyString() const override = 0;
};

// fl_compat extension: the flat gradient arena behind a network built from an arch file (one data-parallel
// all-reduce / one fused optimizer launch instead of one per parameter); {nullptr, 0} for any other module
struct FlatView { float* ptr; size_t floats; };
FL_COMPAT_API FlatView flatParameters(const std::shared_ptr<fl::Module>& network);
FL_COMPAT_API FlatView flatGradients(const std::shared_ptr<fl::Module>& network);
// --fl_amp_use_mixed_precision, restated for bf16: in a network built from an arch file every fl::Linear product, the TDS and
// sub-sampling convolutions and the attention products (scores, position term, P V and their gradients) multiply bf16 operand
// images with fp32 accumulation; activations, master weights, LayerNorm, the optimizer and the criterion stay fp32 (no-op for
// any other module)
FL_COMPAT_API void setMixedPrecision(const std::shared_ptr<fl::Module>& network, bool on);
// fl_compat extension: forwards so far of a network built from an arch file = the position of its dropout-seed stream
// (restored by Serializer::load; `Train fork` starts it from zero)
FL_COMPAT_API uint32_t networkStep(const std::shared_ptr<fl::Module>& network);
FL_COMPAT_API void setNetworkStep(const std::shared_ptr<fl::Module>& network, uint32_t step);

class FL_COMPAT_API ASGLoss : public SequenceCriterion {
 public:
  ASGLoss(int N, CriterionScaleMode scalemode = CriterionScaleMode::NONE, double transdiag = 0.0);
  std::vector<Variable> forward(const std::vector<Variable>& inputs) override;
  af::array viterbiPath(const af::array& input, const af::array& inputSize = af::array()) override;
  af::array viterbiPathWithTarget(const af::array& input, const af::array& target, const af::array& inputSizes = af::array(),
                                  const af::array& targetSizes = af::array()) override;
  std::string prettyString() const override;
  Variable transitions() const { return params_[0]; }   // (N, N), [to][from]
  CriterionScaleMode scaleMode() const { return scaleMode_; }

 private:
  int N_;
  CriterionScaleMode scaleMode_;
};

class FL_COMPAT_API CTCLoss : public SequenceCriterion {
 public:
  explicit CTCLoss(CriterionScaleMode scalemode = CriterionScaleMode::NONE);
  std::vector<Variable> forward(const std::vector<Variable>& inputs) override;
  af::array viterbiPath(const af::array& input, const af::array& inputSize = af::array()) override;   // per-frame argmax
  af::array viterbiPathWithTarget(const af::array& input, const af::array& target, const af::array& inputSizes = af::array(),
                                  const af::array& targetSizes = af::array()) override;
  std::string prettyString() const override;
  CriterionScaleMode scaleMode() const { return scaleMode_; }

 private:
  CriterionScaleMode scaleMode_;
};
}  // namespace speech
}  // namespace pkg
}  // namespace fl
