// w2l_host.hpp -- C++ host layer above the kernel C ABI: arch-file parser, module
// graph with explicit forward/backward, sequence criteria, trainer step.
//
// Mirrors the surface the reference's Trainer uses (Flashlight <= 0.3.2, un-vendored):
//   buildSequentialModule(archfile, nFeat, nLabel)   recipes/joint_training_vox_populi/cpc/SequentialBuilder.h:23-26
//   fl::Module zoo instantiated by the arch grammar   .../cpc/SequentialBuilder.cpp:92-626
//   ASGLoss / CTCLoss                                  recipes/slimIPL/src/Train.cpp:406-410
//   one optimisation step                              recipes/slimIPL/src/Train.cpp:1454-1804
// Design (MI355X-first, not a translation of fl::Variable autograd): the arch files
// only ever build a fl::Sequential, so the graph is a static layer list with hand-written
// backward passes; activations live frame-major [B][T][F] in one arena planned once per
// (B, T); View/Reorder are metadata (a logical ArrayFire-dims view over the physical
// buffer) and never move data -- permutations they imply are absorbed into the weights
// of the next Linear.  Parameters, gradients and momentum are three flat arenas so that
// data-parallel training needs exactly ONE all-reduce and ONE fused optimizer launch.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/w2l_hip.h"

namespace w2l {

// ----------------------------------------------------------------------------
// arch grammar
struct LayerSpec {
  std::string tok;
  std::vector<std::string> args;  // without the token itself
  std::string line;
  int lineNo = 0;
  std::shared_ptr<LayerSpec> child;  // WN <dim> <child line>
};
// trims, drops blank / '#' lines, substitutes NFEAT / NLABEL, validates arity per token
std::vector<LayerSpec> parseArch(const std::string& text, int64_t nFeat, int64_t nLabel);
std::string readFile(const std::string& path);

// ----------------------------------------------------------------------------
// logical (ArrayFire-dims) view over a physical frame-major buffer [B][T][F]
enum FKind { F_FEAT = 0, F_TIME = 1, F_BATCH = 2 };
struct Factor {
  int size;
  int stride;  // F_FEAT: element stride inside a frame; F_TIME / F_BATCH: 1
  FKind kind;
};
struct LDim {
  std::vector<Factor> f;  // fastest first
  long size() const { long s = 1; for (auto& x : f) s *= x.size; return s; }
};
struct Act {
  LDim d[4];
  int B = 0, T = 0, F = 0;
  size_t numel() const { return (size_t)B * T * F; }
  std::string str() const;
};
Act actInput(int B, int T, int nFeat);          // Flashlight input (T, NFEAT, 1, B)
Act actView(const Act& a, const long dims[4]);  // fl::View semantics (-1 infer, 0 keep)
Act actReorder(const Act& a, const int perm[4]);

// ----------------------------------------------------------------------------
struct ParamInfo {
  std::string name;
  size_t numel = 0;
  size_t offset = 0;             // into the flat arenas (floats)
  std::vector<int> refShape;     // ArrayFire dims of the Flashlight parameter
  int kind = 0;                  // 0 plain, 1 conv weight, 2 linear weight (rows may be permuted), 3 scalar pair, 4 WeightNorm g, 5 2-D table stored transposed
  std::vector<int> rowPerm;      // linear: internal row r holds reference row rowPerm[r]
  double initBound = 0;          // uniform(-b, b); 0 => constant initConst
  float initConst = 0;
  int kw = 0, kh = 1, cin = 0, cout = 0;
};

struct Ctx {
  hipStream_t stream = nullptr;
  bool train = true;
  uint32_t seed = 0;     // dropout seed of this step
  float* params = nullptr;
  float* grads = nullptr;
  const float* inputSizes = nullptr;  // device [B] (any unit) or null: padding mask of the Transformer blocks
  int inputT = 0;                     // frames of the padded network input the sizes refer to
  const float* inputSizeFull = nullptr;  // device scalar or null: the size inputT frames correspond to (a batch padded beyond its longest utterance)
  bool bf16 = false;                  // mixed precision: the fl::Linear products run on bf16 operand images (gemm_bf16g.hpp)
  // mixed precision: bf16 images of the activation `imgOf` that the layer producing it has already written (row-major
  // [imgRows][pad64(imgCols)] at arena float offset imgRowsOff, transposed [imgCols + 1 ones row][pad64(imgRows)] at imgTransOff).
  // Sequential::forward clears the note unless the layer that just ran left it for its own output; a consumer whose input matrix
  // has another shape ignores it.
  const float* imgOf = nullptr;
  size_t imgRowsOff = 0, imgTransOff = 0;
  int imgRows = 0, imgCols = 0;
};

class Planner {  // bump allocator over the activation arena (sizes only until bound)
 public:
  size_t alloc(size_t floats) { size_t o = used_; used_ += (floats + 63) / 64 * 64; return o; }
  size_t allocBf16(size_t elements) { return alloc((elements + 1) / 2); }   // a bf16 image, addressed in float slots
  size_t used() const { return used_; }
 private:
  size_t used_ = 0;
};

class Layer {
 public:
  virtual ~Layer() {}
  virtual std::string name() const = 0;
  virtual void registerParams(std::vector<ParamInfo>& table) { (void)table; }
  // shape inference + arena planning for a given input; returns the output activation
  virtual Act plan(const Act& in, Planner& pl) = 0;
  virtual void forward(Ctx& c, float* arena, const float* x, float*& y) = 0;
  // dy: gradient wrt output; writes dx unless !needDx
  virtual void backward(Ctx& c, float* arena, const float* dy, float*& dx, bool needDx) = 0;
  // true when forward() hands its input through UNCHANGED (same pointer, same values) under this context: only then may
  // a note about the bf16 images of the input (Ctx::imgOf) go on describing the output
  virtual bool passesInputThrough(const Ctx& c) const { (void)c; return false; }
  int rngStream = 0;  // distinct dropout stream per layer
};

class Sequential {
 public:
  void add(std::shared_ptr<Layer> l) { layers_.push_back(std::move(l)); }
  size_t size() const { return layers_.size(); }
  Layer& at(size_t i) { return *layers_[i]; }
  std::string prettyString() const;

  void finalize();  // registers params, assigns offsets and rng streams
  const std::vector<ParamInfo>& params() const { return params_; }
  size_t paramFloats() const { return paramFloats_; }

  // plan for input (T, nFeat, 1, B); returns activation-arena floats needed
  size_t plan(int B, int T, int nFeat);
  const Act& outAct() const { return out_; }
  const Act& inAct() const { return in_; }

  // x: device [B][nFeat][T] (the reference's (T,NFEAT,1,B) array, time fastest)
  const float* forward(Ctx& c, float* arena, const float* xRef);
  // dEmission: [B][T'][N]; accumulates nothing, overwrites all parameter gradients
  void backward(Ctx& c, float* arena, const float* dOut);

  // Data-parallel overlap: parameters sit in the flat arena in layer order and backward walks the layers
  // last to first, so after layer i every gradient at offset >= firstOffset(i) is final.  Bucket k =
  // [offsets[k], offsets[k+1]) (ascending); events[k] is recorded on the step's stream as soon as all
  // gradients at offsets >= offsets[k] are complete, so the caller's reduce of the LAST buckets runs
  // under the backward pass of the earlier layers.  Empty vectors switch the hooks off.
  void setGradBuckets(std::vector<size_t> offsets, std::vector<hipEvent_t> events) { bOff_ = std::move(offsets); bEv_ = std::move(events); }

  void initParams(float* hostParams, uint64_t seed) const;            // Flashlight-style init, internal layout
  void importParam(size_t i, const float* ref, float* hostParams) const;  // reference layout -> internal
  void exportParam(size_t i, const float* hostArena, float* ref) const;   // internal -> reference layout

 private:
  std::vector<std::shared_ptr<Layer>> layers_;
  std::vector<ParamInfo> params_;
  size_t paramFloats_ = 0;
  Act in_, out_;
  size_t inOff_ = 0;
  std::vector<size_t> dOff_;  // gradient buffer per layer boundary
  std::vector<Act> acts_;
  std::vector<float*> ys_;
  std::vector<size_t> layerLo_;  // per layer: arena offset of its first parameter (paramFloats_ if it has none after it)
  std::vector<size_t> bOff_;
  std::vector<hipEvent_t> bEv_;
};

// arch file -> Sequential (throws std::invalid_argument like the reference's builder)
std::shared_ptr<Sequential> buildSequentialModule(const std::string& archfile, int64_t nFeatures, int64_t nClasses);
std::shared_ptr<Sequential> buildSequentialFromText(const std::string& archText, int64_t nFeatures, int64_t nClasses);

// ----------------------------------------------------------------------------
// sequence criteria (fl::pkg::speech::SequenceCriterion)
class SequenceCriterion {
 public:
  virtual ~SequenceCriterion() {}
  virtual std::string prettyString() const = 0;
  virtual size_t paramFloats() const { return 0; }
  virtual void initParams(float* host) const { (void)host; }
  virtual size_t workspaceBytes(int B, int T, int N, int L) const = 0;
  // emission [B][T][N], target [B][L] (device, int32, -1 padded); loss [B] device
  virtual void forward(Ctx& c, int B, int T, int N, int L, const float* emission, const int* target,
                       float* loss, void* ws, float* critParams) = 0;
  virtual void backward(Ctx& c, int B, int T, int N, int L, const float* emission, const int* target,
                        const float* gradLoss, float* dEmission, void* ws, float* critParams,
                        float* critGrads) = 0;
  virtual void viterbiPath(Ctx& c, int B, int T, int N, const float* emission, int* path, void* ws,
                           float* critParams) = 0;
};
std::shared_ptr<SequenceCriterion> makeCTCLoss(int scaleMode);
std::shared_ptr<SequenceCriterion> makeASGLoss(int N, int scaleMode, double transdiag);
std::shared_ptr<SequenceCriterion> makeLinSegCriterion(int N, int scaleMode);  // shares the ASG transitions (caller passes them)

// gflags-style flag files (--flagsfile, --k=v lines, '#' comments), recipes/*/train.cfg
struct Flags {
  std::vector<std::pair<std::string, std::string>> kv;
  bool has(const std::string& k) const;
  std::string get(const std::string& k, const std::string& def = "") const;
  double getd(const std::string& k, double def) const;
  long geti(const std::string& k, long def) const;
  bool getb(const std::string& k, bool def) const;
  void set(const std::string& k, const std::string& v);
};
Flags parseFlagsText(const std::string& text);
Flags parseFlagsFile(const std::string& path);
int criterionScaleMode(const std::string& onorm, bool sqnorm);  // getCriterionScaleMode

void hipCheck(hipError_t e, const char* what);
void w2lCheck(int status, const char* what);

}  // namespace w2l
