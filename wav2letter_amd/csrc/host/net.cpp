// net.cpp -- module graph: logical-layout tracking, layers with explicit forward /
// backward over the kernel C ABI, arch-spec -> Sequential builder.
// Module semantics restated from the arch grammar
// (recipes/joint_training_vox_populi/cpc/SequentialBuilder.cpp:92-626) and, for TDS,
// recipes/streaming_convnets/inference/inference/module/nn/TDSBlock.cpp:58-70 and
// recipes/streaming_convnets/tools/StreamingTDSModelConverter.cpp:103-136 (10 parameters:
// conv w,b; LN1 gamma,beta; lin1 w,b; lin2 w,b; LN2 gamma,beta).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <sstream>

#include "w2l_host.hpp"

namespace w2l {

// ============================================================================ layout
std::string Act::str() const {
  std::ostringstream os;
  os << "(";
  for (int i = 0; i < 4; ++i) {
    os << (i ? "," : "");
    if (d[i].f.size() == 1 && d[i].f[0].kind == F_TIME) os << "T";
    else if (d[i].f.size() == 1 && d[i].f[0].kind == F_BATCH) os << "B";
    else os << d[i].size();
  }
  os << ") phys[" << B << "][" << T << "][" << F << "]";
  return os.str();
}

// The time and batch axes are dynamic: in the layout algebra they carry SYMBOLIC sizes (two
// primes that never divide a feature count), so View / Reorder bookkeeping is independent of
// the actual B and T (which live in Act::B / Act::T) and never degenerates when B or T is 1.
static constexpr int kSymT = 1000003, kSymB = 1000033;

Act actInput(int B, int T, int nFeat) {
  Act a;
  a.B = B; a.T = T; a.F = nFeat;
  a.d[0].f = {{kSymT, 1, F_TIME}};
  if (nFeat > 1) a.d[1].f = {{nFeat, 1, F_FEAT}};
  a.d[3].f = {{kSymB, 1, F_BATCH}};
  return a;
}

Act actView(const Act& a, const long dimsIn[4]) {
  std::vector<Factor> flat;
  long total = 1;
  for (int i = 0; i < 4; ++i) {
    for (auto& f : a.d[i].f) flat.push_back(f);
    total *= a.d[i].size();
  }
  long dims[4];
  long known = 1;
  int infer = -1;
  for (int i = 0; i < 4; ++i) {
    dims[i] = dimsIn[i];
    if (dims[i] == 0) dims[i] = a.d[i].size();
    if (dims[i] == -1) {
      if (infer >= 0) throw std::invalid_argument("View: more than one -1");
      infer = i;
    } else {
      known *= dims[i];
    }
  }
  if (infer >= 0) {
    if (known == 0 || total % known) throw std::invalid_argument("View: cannot infer dimension");
    dims[infer] = total / known;
  } else if (known != total) {
    throw std::invalid_argument("View: element count mismatch");
  }
  Act o = a;
  size_t pos = 0;
  for (int i = 0; i < 4; ++i) {
    o.d[i].f.clear();
    long need = dims[i];
    while (need > 1) {
      if (pos >= flat.size()) throw std::invalid_argument("View: ran out of factors");
      Factor& f = flat[pos];
      if (f.size == 1) { ++pos; continue; }
      if (need % f.size == 0) {
        o.d[i].f.push_back(f);
        need /= f.size;
        ++pos;
      } else if (f.size % need == 0) {
        // split the factor: low part (size need) stays here, high part continues
        Factor lo = f, hi = f;
        lo.size = (int)need;
        hi.size = f.size / (int)need;
        if (f.kind == F_FEAT) hi.stride = f.stride * (int)need;
        else throw std::invalid_argument("View: cannot split the time/batch axis");
        o.d[i].f.push_back(lo);
        f = hi;
        need = 1;
      } else {
        throw std::invalid_argument("View: incompatible factorisation");
      }
    }
  }
  return o;
}

Act actReorder(const Act& a, const int perm[4]) {
  Act o = a;
  bool seen[4] = {false, false, false, false};
  for (int i = 0; i < 4; ++i) {
    if (perm[i] < 0 || perm[i] > 3 || seen[perm[i]]) throw std::invalid_argument("Reorder: invalid permutation");
    seen[perm[i]] = true;
    o.d[i] = a.d[perm[i]];
  }
  return o;
}

static bool isKind(const LDim& d, FKind k) { return d.f.size() == 1 && d.f[0].kind == k; }
static bool allFeat(const LDim& d) {
  for (auto& f : d.f) if (f.kind != F_FEAT) return false;
  return true;
}
// (T, H, C, B) with physical frame = [H][C], C fastest
static void requireConvLayout(const Act& a, int cin, const char* who, int& H) {
  if (!isKind(a.d[0], F_TIME) || !isKind(a.d[3], F_BATCH))
    throw std::invalid_argument(std::string(who) + ": input must be (T, H, C, B), got " + a.str());
  long c = a.d[2].size(), h = a.d[1].size();
  if (c != cin) throw std::invalid_argument(std::string(who) + ": channel mismatch, input " + a.str());
  if (!allFeat(a.d[1]) || !allFeat(a.d[2]) || h * c != a.F)
    throw std::invalid_argument(std::string(who) + ": unsupported layout " + a.str());
  if (c > 1 && !(a.d[2].f.size() == 1 && a.d[2].f[0].stride == 1))
    throw std::invalid_argument(std::string(who) + ": channels must be contiguous, layout " + a.str());
  if (h > 1 && !(a.d[1].f.size() == 1 && a.d[1].f[0].stride == (int)c))
    throw std::invalid_argument(std::string(who) + ": rows must be frame-major, layout " + a.str());
  H = (int)h;
}
static Act convOutAct(const Act& in, int To, int H, int cout) {
  Act o;
  o.B = in.B; o.T = To; o.F = H * cout;
  o.d[0].f = {{kSymT, 1, F_TIME}};
  if (H > 1) o.d[1].f = {{H, cout, F_FEAT}};
  if (cout > 1) o.d[2].f = {{cout, 1, F_FEAT}};
  o.d[3].f = {{kSymB, 1, F_BATCH}};
  return o;
}

static int samePad(int T, int kw, int stride) { return w2l_conv_same_pad(T, kw, stride); }

// ============================================================================ layers
namespace {

struct P {  // bound parameter handle
  size_t idx = 0;
  const std::vector<ParamInfo>* table = nullptr;
  float* w(Ctx& c) const { return c.params + (*table)[idx].offset; }
  float* g(Ctx& c) const { return c.grads + (*table)[idx].offset; }
};

static size_t addParam(std::vector<ParamInfo>& t, ParamInfo pi, P& h) {
  t.push_back(std::move(pi));
  h.idx = t.size() - 1;
  h.table = &t;
  return h.idx;
}

class ViewLayer : public Layer {
 public:
  long dims[4];
  std::string name() const override { return "View"; }
  Act plan(const Act& in, Planner&) override { return actView(in, dims); }
  bool passesInputThrough(const Ctx&) const override { return true; }
  void forward(Ctx&, float*, const float* x, float*& y) override { y = const_cast<float*>(x); }
  void backward(Ctx&, float*, const float* dy, float*& dx, bool) override { dx = const_cast<float*>(dy); }
};

class ReorderLayer : public Layer {
 public:
  int perm[4];
  std::string name() const override { return "Reorder"; }
  Act plan(const Act& in, Planner&) override { return actReorder(in, perm); }
  bool passesInputThrough(const Ctx&) const override { return true; }
  void forward(Ctx&, float*, const float* x, float*& y) override { y = const_cast<float*>(x); }
  void backward(Ctx&, float*, const float* dy, float*& dx, bool) override { dx = const_cast<float*>(dy); }
};

class ReLULayer : public Layer {
 public:
  size_t n = 0;
  float* yPtr = nullptr;
  std::string name() const override { return "ReLU"; }
  Act plan(const Act& in, Planner&) override { n = in.numel(); return in; }
  void forward(Ctx& c, float*, const float* x, float*& y) override {
    y = const_cast<float*>(x);
    w2lCheck(w2l_mask_backward(x, x, y, n, 1.f, c.stream), "relu");
    yPtr = y;
  }
  void backward(Ctx& c, float*, const float* dy, float*& dx, bool) override {
    dx = const_cast<float*>(dy);
    w2lCheck(w2l_mask_backward(dy, yPtr, dx, n, 1.f, c.stream), "relu bwd");
  }
};

class DropoutLayer : public Layer {
 public:
  double p = 0;
  size_t n = 0;
  std::string name() const override { return "Dropout"; }
  Act plan(const Act& in, Planner&) override { n = in.numel(); return in; }
  bool passesInputThrough(const Ctx& c) const override { return !(c.train && p > 0); }
  void forward(Ctx& c, float*, const float* x, float*& y) override {
    y = const_cast<float*>(x);
    if (c.train && p > 0) w2lCheck(w2l_dropout_inplace(y, n, p, c.seed, rngStream, c.stream), "dropout");
  }
  void backward(Ctx& c, float*, const float* dy, float*& dx, bool) override {
    dx = const_cast<float*>(dy);
    if (c.train && p > 0) w2lCheck(w2l_dropout_inplace(dx, n, p, c.seed, rngStream, c.stream), "dropout bwd");
  }
};

class SpecAugmentLayer : public Layer {
 public:
  int fMaskF = 0, nFMask = 0, tMaskT = 0, nTMask = 0;
  float tMaskP = 1.f;
  int B = 0, T = 0, F = 0;
  std::string name() const override { return "SpecAugment"; }
  Act plan(const Act& in, Planner&) override {
    if (!isKind(in.d[0], F_TIME)) throw std::invalid_argument("SAUG expects (T, F, 1, B) input");
    B = in.B; T = in.T; F = in.F;
    return in;
  }
  bool passesInputThrough(const Ctx& c) const override { return !c.train; }
  void forward(Ctx& c, float*, const float* x, float*& y) override {
    y = const_cast<float*>(x);
    if (c.train)
      w2lCheck(w2l_specaugment_inplace(y, B, T, F, fMaskF, nFMask, tMaskT, tMaskP, nTMask, c.seed ^ 0x5a5a5a5au, c.stream),
               "specaugment");
  }
  void backward(Ctx&, float*, const float* dy, float*& dx, bool) override { dx = const_cast<float*>(dy); }
};


// ---- mixed precision with bf16 operand storage (Ctx::bf16; gemm_bf16g.hpp, convert.hip) -------------------------------
// fl's AMP (recipes/slimIPL/src/Train.cpp:211, :1681-1760; cpc/Train.cpp:365) casts the operands of fl::linear to half
// precision and keeps fp32 master weights; here the half type is bf16 and every operand of the three products of an
// fl::Linear is a bf16 IMAGE in the activation arena, written once per step:
//   activation x [M][in]   -> rows image [M][inP] (forward A) + transposed image [in][MP] (weight-gradient A)
//   gradient  dy [M][out]  -> rows image [M][outP] (backward-data A) + transposed image [out][MP] (weight-gradient B)
//   weight     w [in][out] -> rows image [in][outP] (backward-data B) + transposed image [out][inP] (forward B)
// (inP / outP / MP: rounded up to 64, zero-filled: the GEMM runs whole 64-k tiles).  Results, bias, LayerNorm, residual,
// dropout and the criterion stay fp32.
static inline int pad64(int n) { return (n + 63) / 64 * 64; }
static inline uint16_t* bfp(float* arena, size_t off) { return (uint16_t*)(arena + off); }

struct BfImage {   // the two bf16 images of an fp32 matrix [rows][cols]
  size_t rowsOff = 0, transOff = 0;
  int rows = 0, cols = 0, colsP = 0, rowsP = 0;
  // onesRow: the transposed image carries one more row, index `cols`, of bf16 ones.  As the A operand of a weight-gradient
  // product x^T dy it makes the product's row `cols` the column sums of dy: the BIAS gradient, which sits right behind the
  // weight gradient in the arena, comes out of the same launch (no column-sum kernels).  Written once per plan.
  bool onesRow = false;
  mutable const float* onesArena = nullptr;   // the arena the ones row was written into (a re-bound arena gets it again)
  void plan(Planner& pl, int r, int c, bool ones = false) {
    rows = r; cols = c; colsP = pad64(c); rowsP = pad64(r);
    onesRow = ones; onesArena = nullptr; sinkArena = nullptr;
    rowsOff = pl.allocBf16((size_t)rows * colsP);
    transOff = pl.allocBf16((size_t)(cols + (ones ? 1 : 0)) * rowsP);
  }
  void convert(Ctx& c, float* arena, const float* x, const char* what) const {
    w2lCheck(w2l_bf16_convert(x, (size_t)rows, cols, (size_t)cols, bfp(arena, rowsOff), (size_t)colsP, bfp(arena, transOff), (size_t)rowsP,
                              c.stream), what);
    ensureOnes(c, arena);
  }
  void ensureOnes(Ctx& c, float* arena) const {
    if (onesRow && onesArena != arena) {   // two bf16 ones per float slot; columns past `rows` multiply the zero padding of dy's image
      float pair;
      const uint32_t bits = 0x3F803F80u;
      std::memcpy(&pair, &bits, 4);
      w2lCheck(w2l_fill((float*)(bfp(arena, transOff) + (size_t)cols * rowsP), (size_t)rowsP / 2, pair, c.stream), "bf16 ones row");
      onesArena = arena;
    }
  }
  // the images of dropout(x) (mask of w2l_dropout_copy with the same p / seed / stream), without the masked copy
  void convertDropout(Ctx& c, float* arena, const float* x, double p, uint32_t seed, uint32_t stream, const char* what) const {
    w2lCheck(w2l_bf16_convert_dropout(x, (size_t)rows, cols, (size_t)cols, bfp(arena, rowsOff), (size_t)colsP, bfp(arena, transOff),
                                      (size_t)rowsP, p, seed, stream, c.stream), what);
  }
  // the images as the OUTPUT of the kernel that produces the matrix (w2l_bf16_image_sink: only elements of the matrix are written).
  // The zero padding a conversion pass would leave is written once per arena -- the planner never aliases an image.
  mutable const float* sinkArena = nullptr;
  w2l_bf16_image_sink sink(Ctx& c, float* arena) const {
    if (sinkArena != arena) {
      if (colsP != cols) w2lCheck(w2l_fill((float*)bfp(arena, rowsOff), ((size_t)rows * colsP + 1) / 2, 0.f, c.stream), "bf16 image padding");
      if (rowsP != rows) {
        w2lCheck(w2l_fill((float*)bfp(arena, transOff), ((size_t)cols * rowsP + 1) / 2, 0.f, c.stream), "bf16 image padding");
      }
      sinkArena = arena;
    }
    ensureOnes(c, arena);
    w2l_bf16_image_sink k;
    k.rowMajor = bfp(arena, rowsOff); k.ldRows = (size_t)colsP; k.transposed = bfp(arena, transOff); k.ldTrans = (size_t)rowsP;
    return k;
  }
  const uint16_t* r(float* arena) const { return bfp(arena, rowsOff); }
  const uint16_t* t(float* arena) const { return bfp(arena, transOff); }
  // entry of a w2l_bf16_convert_multi call (an image with a ones row: ensureOnes() after the call)
  w2l_bf16_convert_desc desc(float* arena, const float* x) const {
    w2l_bf16_convert_desc d;
    d.x = x; d.rows = (size_t)rows; d.cols = cols; d.ldx = (size_t)cols;
    d.rowMajor = bfp(arena, rowsOff); d.ldRows = (size_t)colsP; d.transposed = bfp(arena, transOff); d.ldTrans = (size_t)rowsP;
    return d;
  }
};

struct BfLinear {   // the weight images of one fl::Linear(in, out) and its three products on M frames
  int M = 0, in = 0, out = 0;
  BfImage w;        // w [in][out]: rows image = B of backward-data, transposed image [out][inP] = B of forward
  void plan(Planner& pl, int M_, int in_, int out_) { M = M_; in = in_; out = out_; w.plan(pl, in, out); }
  void convertWeight(Ctx& c, float* arena, const float* wf) const { w.convert(c, arena, wf, "bf16 weight images"); }
  // y [M][out] = x w + b (ReLU) (dropout); xImg: images of x [M][in]
  void forward(Ctx& c, float* arena, const BfImage& xImg, const float* bias, float* y, int relu, double dropP, uint32_t seed,
               uint32_t stream) const {
    w2l_gemm_epilogue e{};
    e.dropP = dropP; e.dropSeed = seed; e.dropStream = stream;
    w2lCheck(w2l_gemm_bf16(M, out, in, xImg.r(arena), xImg.colsP, w.t(arena), w.rowsP, y, out, bias, relu, dropP > 0 ? &e : nullptr,
                           c.stream), "bf16 linear fwd");
  }
  // y [M][out] = dropout(x w + b) + add: the residual join behind the product in its epilogue (w2l_linear_forward_dropout_add)
  void forwardAdd(Ctx& c, float* arena, const BfImage& xImg, const float* bias, const float* add, float* y, double dropP, uint32_t seed,
                  uint32_t stream) const {
    w2l_gemm_epilogue e{};
    e.dropP = dropP; e.dropSeed = seed; e.dropStream = stream; e.addend = add;
    w2lCheck(w2l_gemm_bf16(M, out, in, xImg.r(arena), xImg.colsP, w.t(arena), w.rowsP, y, out, bias, 0, &e, c.stream), "bf16 linear fwd + add");
  }
  // y = dropout(relu?(x w + b)) leaving ONLY as the bf16 images yImg (w2l_gemm_bf16_images: no fp32 y, no conversion pass): for an
  // activation that is only ever a GEMM operand and a ReLU / dropout mask.  false: this geometry has no such epilogue (caller: the
  // fp32 result + conversion)
  bool forwardImages(Ctx& c, float* arena, const BfImage& xImg, const float* bias, const BfImage& yImg, int relu, double dropP, uint32_t seed,
                     uint32_t stream) const {
    w2l_gemm_epilogue e{};
    e.dropP = dropP; e.dropSeed = seed; e.dropStream = stream;
    const w2l_bf16_image_sink k = yImg.sink(c, arena);
    const int st = w2l_gemm_bf16_images(M, out, in, xImg.r(arena), xImg.colsP, w.t(arena), w.rowsP, nullptr, out, bias, relu,
                                        dropP > 0 ? &e : nullptr, &k, nullptr, 0, 1.f, c.stream);
    if (st == W2L_EUNSUPPORTED) return false;
    w2lCheck(st, "bf16 linear fwd -> images");
    return true;
  }
  // dx = (dy w^T) masked by (maskImg > 0) * maskScale, maskImg the bf16 row image of the forward activation; dx leaves as the images
  // dxImg only (dx == nullptr) or as fp32 too
  bool backwardDataImages(Ctx& c, float* arena, const BfImage& dyImg, float* dx, const BfImage* dxImg, const BfImage& maskImg,
                          float maskScale) const {
    w2l_bf16_image_sink k{};
    if (dxImg) k = dxImg->sink(c, arena);
    const int st = w2l_gemm_bf16_images(M, in, out, dyImg.r(arena), dyImg.colsP, w.r(arena), w.colsP, dx, in, nullptr, 0, nullptr,
                                        dxImg ? &k : nullptr, maskImg.r(arena), (size_t)maskImg.colsP, maskScale, c.stream);
    if (st == W2L_EUNSUPPORTED) return false;
    w2lCheck(st, "bf16 linear bwd data -> images");
    return true;
  }
  // dx [M][in] = dy w^T (mask) (+ addend | += dx); dyImg: images of dy [M][out]
  void backwardData(Ctx& c, float* arena, const BfImage& dyImg, float* dx, const float* mask, float maskScale, const float* addend,
                    int accumulate) const {
    w2l_gemm_epilogue e{};
    e.mask = mask; e.maskScale = maskScale; e.addend = addend; e.accumulate = accumulate;
    w2lCheck(w2l_gemm_bf16(M, in, out, dyImg.r(arena), dyImg.colsP, w.r(arena), w.colsP, dx, in, nullptr, 0,
                           (mask || addend || accumulate) ? &e : nullptr, c.stream), "bf16 linear bwd data");
  }
  // dw [in][out] = x^T dy
  // withBias: xImg carries the ones row and db == dw + in * out (the caller checked): row `in` of the product is the bias gradient
  void backwardWeight(Ctx& c, float* arena, const BfImage& xImg, const BfImage& dyImg, float* dw, bool withBias = false) const {
    w2lCheck(w2l_gemm_bf16(in + (withBias ? 1 : 0), out, M, xImg.t(arena), xImg.rowsP, dyImg.t(arena), dyImg.rowsP, dw, out, nullptr, 0,
                           nullptr, c.stream), "bf16 linear bwd weight");
  }
  // the bias gradient can ride on the weight-gradient product: its slot follows the weight's in the gradient arena
  bool biasRides(const BfImage& xImg, const float* dw, const float* db) const {
    return xImg.onesRow && db == dw + (size_t)in * out;
  }
};

// LayerNorm of the mixed-precision mode: where the normalised rows are the rows of the next product's operand (`img` non-null,
// rows of at most 2304 floats) the images come out of the LayerNorm kernel itself (layernorm_images.hip) -- returns true then;
// otherwise the plain kernels run and the caller converts.
// Measured on one box (profiles/r04_run24_ln_images_ab.log): at config 5's 3008 x 1024 matrices the image-writing kernels replace a
// launch-bound conversion; at config 3's 11968 x 1200 ... 2160 they stream at 3.5 TB/s against 5.4 for LayerNorm + conversion and
// the step LOSES 0.8 ms -- so they are used below kLnImageElems only.
constexpr size_t kLnImageElems = (size_t)1 << 22;
static bool lnForward(Ctx& cx, float* ar, int groups, size_t inner, float* a, const float* x, float* r, float* y, const float* gb, double p,
                      uint32_t seed, uint32_t stream, double* stats, float* mr, const BfImage* img, const char* what) {
  if (img && img->rows == groups && (size_t)img->cols == inner && (size_t)groups * inner <= kLnImageElems) {
    const w2l_bf16_image_sink k = img->sink(cx, ar);
    const int st = w2l_residual_layernorm_forward_images(groups, inner, a, x, r, y, gb, 1e-5f, p, seed, stream, mr, &k, cx.stream);
    if (st == W2L_OK) return true;
    if (st != W2L_EUNSUPPORTED) w2lCheck(st, what);
  }
  w2lCheck(w2l_residual_layernorm_forward(groups, inner, a, x, r, y, gb, 1e-5f, p, seed, stream, stats, mr, cx.stream), what);
  return false;
}
// imgP > 0: the images are those of dropout(dr) (hash over the flat index with imgSeed / imgStream), dr stays unmasked
static bool lnBackward(Ctx& cx, float* ar, int groups, size_t inner, const float* r, const float* dy, const float* gb, const float* mr, float* dr,
                       float* dgb, const float* maskSrc, float* dmask, float maskScale, double* sums, const BfImage* img, double imgP,
                       uint32_t imgSeed, uint32_t imgStream, const char* what) {
  if (img && img->rows == groups && (size_t)img->cols == inner && (size_t)groups * inner <= kLnImageElems) {
    const w2l_bf16_image_sink k = img->sink(cx, ar);
    const int st = w2l_layernorm_backward_images(groups, inner, r, dy, gb, mr, dr, dgb, maskSrc, dmask, maskScale, sums, &k, imgP, imgSeed,
                                                 imgStream, cx.stream);
    if (st == W2L_OK) return true;
    if (st != W2L_EUNSUPPORTED) w2lCheck(st, what);
  }
  w2lCheck(w2l_layernorm_backward(groups, inner, r, dy, gb, mr, dr, dgb, maskSrc, dmask, maskScale, sums, cx.stream), what);
  return false;
}

// optional WeightNorm state shared by Conv2D / Linear
struct WNState {
  bool on = false;
  P v, g;
  size_t wOff = 0, dwOff = 0, normOff = 0, dotOff = 0;
  int K = 0, N = 0;
};

class Conv2DLayer : public Layer {
 public:
  int cin, cout, kw, stride, pad;  // pad -1 = SAME
  int kh = 1, padH = 0;            // kh > 1: the mel axis is unrolled into kh*cin channels (w2l_hexpand_*), SAME padding on it
  size_t xeOff = 0, dxeOff = 0, nIn = 0;
  bool hasBias = true, fuseRelu = false;
  int extraPadL = 0, extraPadR = 0;  // from a preceding "PD" (time axis)
  WNState wn;
  P w, b;
  w2l_conv_desc d{};
  size_t yOff = 0, dxOff = 0;
  size_t nOut = 0;
  const float* xSaved = nullptr;
  // mixed precision: the sub-sampling convolutions of the TDS recipes (<= 32 channels over the mel rows) on the bf16 kernels
  size_t convImgElems = 0, convImgFOff = 0, convImgBOff = 0;

  std::string name() const override { return wn.on ? "WeightNorm(Conv2D)" : "Conv2D"; }
  void registerParams(std::vector<ParamInfo>& t) override {
    ParamInfo pw;
    pw.numel = (size_t)kw * kh * cin * cout;
    pw.refShape = {kw, kh, cin, cout};
    pw.kind = 1; pw.kw = kw; pw.kh = kh; pw.cin = cin; pw.cout = cout;
    pw.initBound = std::sqrt(1.0 / ((double)cin * kw * kh));
    if (wn.on) {
      pw.name = "wn.v";
      addParam(t, pw, wn.v);
      ParamInfo pg;
      pg.name = "wn.g"; pg.numel = cout; pg.refShape = {1, 1, 1, cout}; pg.kind = 4;  // init = ||v||
      addParam(t, pg, wn.g);
      wn.K = kw * kh * cin; wn.N = cout;
    } else {
      pw.name = "conv.w";
      addParam(t, pw, w);
    }
    if (hasBias) {
      ParamInfo pb;
      pb.name = "conv.b"; pb.numel = cout; pb.refShape = {1, 1, cout, 1};
      pb.initBound = std::sqrt(1.0 / ((double)cin * kw * kh));
      addParam(t, pb, b);
    }
  }
  Act plan(const Act& in, Planner& pl) override {
    int H;
    requireConvLayout(in, cin, "Conv2D", H);
    int pL, pR;
    if (pad == -1) pL = pR = samePad(in.T, kw, stride); else pL = pR = pad;
    pL += extraPadL; pR += extraPadR;
    d = {in.B, in.T, H, kh * cin, cout, kw, stride, pL, pR};
    int To = w2l_conv_out_len(in.T, kw, stride, pL, pR);
    if (To <= 0) throw std::invalid_argument("Conv2D: input too short: " + in.str());
    Act o = convOutAct(in, To, H, cout);
    nOut = o.numel();
    yOff = pl.alloc(nOut);
    dxOff = pl.alloc(in.numel());
    nIn = in.numel();
    if (kh > 1) { xeOff = pl.alloc(nIn * kh); dxeOff = pl.alloc(nIn * kh); }
    convImgElems = H % 16 == 0 ? w2l_tds_conv_bf16_image_elems(&d) : 0;
    if (convImgElems) { convImgFOff = pl.allocBf16(convImgElems); convImgBOff = pl.allocBf16(convImgElems); }
    if (wn.on) {
      wn.wOff = pl.alloc((size_t)wn.K * wn.N);
      wn.dwOff = pl.alloc((size_t)wn.K * wn.N);
      wn.normOff = pl.alloc(wn.N);
      wn.dotOff = pl.alloc(wn.N);
    }
    return o;
  }
  const float* weight(Ctx& c, float* arena) {
    if (!wn.on) return w.w(c);
    w2lCheck(w2l_weightnorm_forward(wn.v.w(c), wn.g.w(c), arena + wn.wOff, arena + wn.normOff, wn.K, wn.N, c.stream), "wn fwd");
    return arena + wn.wOff;
  }
  void forward(Ctx& c, float* arena, const float* x, float*& y) override {
    y = arena + yOff;
    if (kh > 1) {
      w2lCheck(w2l_hexpand_forward(x, arena + xeOff, (size_t)d.B * d.T, d.H, cin, kh, padH, c.stream), "conv H unroll");
      x = arena + xeOff;
    }
    xSaved = x;
    const float* wt = weight(c, arena);
    if (c.bf16 && convImgElems) {   // bf16 operands, fp32 accumulation / bias / ReLU (conv_tds_bf16.hip); images once per step
      w2lCheck(w2l_tds_conv_bf16_prepare(&d, wt, bfp(arena, convImgFOff), bfp(arena, convImgBOff), c.stream), "conv images");
      w2lCheck(w2l_tds_conv_bf16_forward(&d, x, bfp(arena, convImgFOff), hasBias ? b.w(c) : nullptr, y, fuseRelu ? 1 : 0, c.stream), "conv fwd bf16");
      return;
    }
    w2lCheck(w2l_conv_forward(&d, x, wt, hasBias ? b.w(c) : nullptr, y, fuseRelu ? 1 : 0, c.stream), "conv fwd");
  }
  void backward(Ctx& c, float* arena, const float* dy, float*& dx, bool needDx) override {
    float* dym = const_cast<float*>(dy);
    if (fuseRelu) w2lCheck(w2l_mask_backward(dy, arena + yOff, dym, nOut, 1.f, c.stream), "conv relu bwd");
    float* dwt = wn.on ? arena + wn.dwOff : w.g(c);
    const bool bf = c.bf16 && convImgElems;
    if (bf) {
      w2lCheck(w2l_tds_conv_bf16_backward_filter_bias(&d, xSaved, dym, dwt, hasBias ? b.g(c) : nullptr, c.stream), "conv bwd filter + bias bf16");
    } else {
      w2lCheck(w2l_conv_backward_filter(&d, xSaved, dym, dwt, hasBias ? b.g(c) : nullptr, c.stream), "conv bwd filter");
    }
    const float* wt = wn.on ? arena + wn.wOff : w.w(c);
    if (needDx) {
      dx = arena + dxOff;
      if (bf && kh > 1) {
        w2lCheck(w2l_tds_conv_bf16_backward_data(&d, dym, bfp(arena, convImgBOff), nullptr, arena + dxeOff, c.stream), "conv bwd data bf16");
        w2lCheck(w2l_hexpand_backward(arena + dxeOff, dx, (size_t)d.B * d.T, d.H, cin, kh, padH, c.stream), "conv H fold");
      } else if (bf) {
        w2lCheck(w2l_tds_conv_bf16_backward_data(&d, dym, bfp(arena, convImgBOff), nullptr, dx, c.stream), "conv bwd data bf16");
      } else if (kh > 1) {
        w2lCheck(w2l_conv_backward_data(&d, dym, wt, arena + dxeOff, 0, c.stream), "conv bwd data");
        w2lCheck(w2l_hexpand_backward(arena + dxeOff, dx, (size_t)d.B * d.T, d.H, cin, kh, padH, c.stream), "conv H fold");
      } else {
        w2lCheck(w2l_conv_backward_data(&d, dym, wt, dx, 0, c.stream), "conv bwd data");
      }
    }
    if (wn.on)
      w2lCheck(w2l_weightnorm_backward(wn.v.w(c), wn.g.w(c), arena + wn.normOff, dwt, wn.v.g(c), wn.g.g(c),
                                       arena + wn.dotOff, wn.K, wn.N, c.stream), "wn bwd");
  }
};

class LinearLayer : public Layer {
 public:
  int in, out;
  bool hasBias = true, fuseRelu = false;
  WNState wn;
  P w, b;
  int M = 0;
  size_t yOff = 0, dxOff = 0;
  const float* xSaved = nullptr;
  std::vector<ParamInfo>* table = nullptr;
  BfLinear bl;          // mixed precision: weight images and the bf16 products
  BfImage xImg, dyImg;  // images of this layer's input and output gradient
  BfImage xSeen;        // the images of x this step's products read: xImg, or the producer's (Ctx::imgOf)

  std::string name() const override { return wn.on ? "WeightNorm(Linear)" : "Linear"; }
  void registerParams(std::vector<ParamInfo>& t) override {
    table = &t;
    ParamInfo pw;
    pw.numel = (size_t)in * out;
    pw.refShape = {out, in};
    pw.kind = 2; pw.cin = in; pw.cout = out;
    pw.initBound = std::sqrt(1.0 / (double)in);
    if (wn.on) {
      pw.name = "wn.v";
      addParam(t, pw, wn.v);
      ParamInfo pg;
      pg.name = "wn.g"; pg.numel = out; pg.refShape = {out, 1}; pg.kind = 4;
      addParam(t, pg, wn.g);
      wn.K = in; wn.N = out;
    } else {
      pw.name = "linear.w";
      addParam(t, pw, w);
    }
    if (hasBias) {
      ParamInfo pb;
      pb.name = "linear.b"; pb.numel = out; pb.refShape = {out};
      pb.initBound = std::sqrt(1.0 / (double)in);
      addParam(t, pb, b);
    }
  }
  Act plan(const Act& a, Planner& pl) override {
    if (a.d[0].size() != in || !allFeat(a.d[0]) || a.F != in)
      throw std::invalid_argument("Linear(" + std::to_string(in) + "," + std::to_string(out) +
                                  "): dim 0 must be the whole frame, input " + a.str());
    for (int i = 1; i < 4; ++i)
      for (auto& f : a.d[i].f)
        if (f.kind == F_FEAT) throw std::invalid_argument("Linear: feature factors outside dim 0: " + a.str());
    // logical feature index -> physical offset in the frame; absorbed into the weight rows
    std::vector<int> perm(in);
    for (int fidx = 0; fidx < in; ++fidx) {
      int rem = fidx, off = 0;
      for (auto& f : a.d[0].f) { off += (rem % f.size) * f.stride; rem /= f.size; }
      perm[fidx] = off;
    }
    std::vector<int> rowPerm(in, -1);  // internal row (physical offset) -> reference row (logical index)
    for (int fidx = 0; fidx < in; ++fidx) {
      if (perm[fidx] < 0 || perm[fidx] >= in || rowPerm[perm[fidx]] != -1)
        throw std::invalid_argument("Linear: input view is not a permutation of the frame: " + a.str());
      rowPerm[perm[fidx]] = fidx;
    }
    bool ident = true;
    for (int r = 0; r < in; ++r) ident = ident && rowPerm[r] == r;
    ParamInfo& pi = (*table)[wn.on ? wn.v.idx : w.idx];
    if (pi.rowPerm.empty() && !ident) pi.rowPerm = rowPerm;
    else if (!pi.rowPerm.empty() && pi.rowPerm != rowPerm && !ident) throw std::invalid_argument("Linear: layout changed between plans");
    M = a.B * a.T;
    Act o = a;
    o.F = out;
    o.d[0].f.clear();
    if (out > 1) o.d[0].f = {{out, 1, F_FEAT}};
    yOff = pl.alloc((size_t)M * out);
    dxOff = pl.alloc((size_t)M * in);
    if (wn.on) {
      wn.wOff = pl.alloc((size_t)wn.K * wn.N);
      wn.dwOff = pl.alloc((size_t)wn.K * wn.N);
      wn.normOff = pl.alloc(wn.N);
      wn.dotOff = pl.alloc(wn.N);
    }
    bl.plan(pl, M, in, out);
    xImg.plan(pl, M, in, hasBias && !wn.on);
    dyImg.plan(pl, M, out);
    return o;
  }
  void forward(Ctx& c, float* arena, const float* x, float*& y) override {
    y = arena + yOff;
    xSaved = x;
    const float* wt = w.table ? w.w(c) : nullptr;
    if (wn.on) {
      w2lCheck(w2l_weightnorm_forward(wn.v.w(c), wn.g.w(c), arena + wn.wOff, arena + wn.normOff, wn.K, wn.N, c.stream), "wn fwd");
      wt = arena + wn.wOff;
    }
    if (c.bf16) {
      xSeen = xImg;
      if (c.imgOf == x && c.imgRows == M && c.imgCols == in) {   // written by the layer that produced x (a Transformer block's LayerNorm)
        xSeen.rowsOff = c.imgRowsOff; xSeen.transOff = c.imgTransOff;
        xSeen.onesRow = true;
      } else {
        xImg.convert(c, arena, x, "linear input images");
      }
      bl.convertWeight(c, arena, wt);
      bl.forward(c, arena, xSeen, hasBias ? b.w(c) : nullptr, y, fuseRelu ? 1 : 0, 0.0, 0, 0);
      return;
    }
    w2lCheck(w2l_linear_forward(M, in, out, x, wt, hasBias ? b.w(c) : nullptr, y, fuseRelu ? 1 : 0, c.stream), "linear fwd");
  }
  void backward(Ctx& c, float* arena, const float* dy, float*& dx, bool needDx) override {
    float* dym = const_cast<float*>(dy);
    if (fuseRelu) w2lCheck(w2l_mask_backward(dy, arena + yOff, dym, (size_t)M * out, 1.f, c.stream), "linear relu bwd");
    float* dwt = wn.on ? arena + wn.dwOff : w.g(c);
    const float* wt = wn.on ? arena + wn.wOff : w.w(c);
    if (c.bf16) {   // the images of x and of the weight were written by forward
      dyImg.convert(c, arena, dym, "linear output-gradient images");
      const bool rides = hasBias && bl.biasRides(xSeen, dwt, b.g(c));
      bl.backwardWeight(c, arena, xSeen, dyImg, dwt, rides);
      if (hasBias && !rides) w2lCheck(w2l_colsum(dym, b.g(c), (size_t)M, out, c.stream), "linear bwd b");
      if (needDx) {
        dx = arena + dxOff;
        bl.backwardData(c, arena, dyImg, dx, nullptr, 1.f, nullptr, 0);
      }
    } else {
    w2lCheck(w2l_linear_backward_weight_bias(M, in, out, xSaved, dym, dwt, hasBias ? b.g(c) : nullptr, c.stream), "linear bwd w + b");
    if (needDx) {
      dx = arena + dxOff;
      w2lCheck(w2l_linear_backward_data(M, in, out, dym, wt, dx, 0, nullptr, 1.f, c.stream), "linear bwd x");
    }
    }
    if (wn.on)
      w2lCheck(w2l_weightnorm_backward(wn.v.w(c), wn.g.w(c), arena + wn.normOff, dwt, wn.v.g(c), wn.g.g(c),
                                       arena + wn.dotOff, wn.K, wn.N, c.stream), "wn bwd");
  }
};

class LayerNormLayer : public Layer {
 public:
  std::vector<int> axes;
  P gb;
  int groups = 0;
  size_t inner = 0, yOff = 0, dxOff = 0, statOff = 0, mrOff = 0;
  const float* rSaved = nullptr;
  std::string name() const override { return "LayerNorm"; }
  void registerParams(std::vector<ParamInfo>& t) override {
    ParamInfo p;
    p.name = "ln.weight+bias"; p.numel = 2; p.refShape = {1}; p.kind = 3;
    addParam(t, p, gb);
  }
  Act plan(const Act& in, Planner& pl) override {
    std::vector<int> ax = axes;
    std::sort(ax.begin(), ax.end());
    bool all = ax == std::vector<int>{0, 1, 2}, frame = ax == std::vector<int>{1, 2};
    if (!isKind(in.d[3], F_BATCH) || !(all || frame) || (frame && !isKind(in.d[0], F_TIME)))
      throw std::invalid_argument("LayerNorm: supported axes are {0,1,2} and {1,2} on (T,H,C,B); input " + in.str());
    groups = all ? in.B : in.B * in.T;
    inner = in.numel() / groups;
    if (inner % 4) throw std::invalid_argument("LayerNorm: normalised size must be a multiple of 4");
    yOff = pl.alloc(in.numel());
    dxOff = pl.alloc(in.numel());
    statOff = pl.alloc(2 * w2l_layernorm_scratch_doubles(groups, inner));  // doubles
    mrOff = pl.alloc(2 * (size_t)groups);
    return in;
  }
  void forward(Ctx& c, float* arena, const float* x, float*& y) override {
    y = arena + yOff;
    rSaved = x;
    float* xm = const_cast<float*>(x);
    w2lCheck(w2l_residual_layernorm_forward(groups, inner, xm, nullptr, xm, y, gb.w(c), 1e-5f, 0.0, 0, 0,
                                            (double*)(arena + statOff), arena + mrOff, c.stream), "layernorm fwd");
  }
  void backward(Ctx& c, float* arena, const float* dy, float*& dx, bool) override {
    dx = arena + dxOff;
    w2lCheck(w2l_layernorm_backward(groups, inner, rSaved, dy, gb.w(c), arena + mrOff, dx, gb.g(c), nullptr, nullptr, 1.f,
                                    (double*)(arena + statOff), c.stream), "layernorm bwd");
  }
};

class GLULayer : public Layer {
 public:
  int dim = 0;
  size_t M = 0;
  int half = 0;
  size_t yOff = 0, dxOff = 0;
  const float* xSaved = nullptr;
  std::string name() const override { return "GatedLinearUnit"; }
  Act plan(const Act& in, Planner& pl) override {
    const LDim& d = in.d[dim];
    if (!(d.f.size() == 1 && d.f[0].kind == F_FEAT && d.f[0].stride == 1 && d.f[0].size == in.F) || (in.F & 1))
      throw std::invalid_argument("GLU: the gated dimension must be the (contiguous) frame; input " + in.str());
    M = (size_t)in.B * in.T;
    half = in.F / 2;
    Act o = in;
    o.F = half;
    o.d[dim].f[0].size = half;
    yOff = pl.alloc(M * half);
    dxOff = pl.alloc(M * in.F);
    return o;
  }
  void forward(Ctx& c, float* arena, const float* x, float*& y) override {
    y = arena + yOff;
    xSaved = x;
    w2lCheck(w2l_glu_forward(x, y, M, half, c.stream), "glu fwd");
  }
  void backward(Ctx& c, float* arena, const float* dy, float*& dx, bool) override {
    dx = arena + dxOff;
    w2lCheck(w2l_glu_backward(xSaved, dy, dx, M, half, c.stream), "glu bwd");
  }
};

// fl::TDSBlock(c, kw, h, dropout, l2, rPad, lNormIncludeTime):
//   y1  = LN(dropout(relu(conv(x))) + x)
//   out = LN(dropout(lin2(dropout(relu(lin1(y1))))) + y1)
class TDSLayer : public Layer {
 public:
  int c, kw, h, l, l2, rPad;
  double p;
  bool lnTime;
  P wc, bc, gb1, w1, b1, w2, b2, gb2;
  w2l_conv_desc d{};
  int B = 0, T = 0, M = 0, groups = 0;
  size_t inner = 0, n = 0;
  size_t aOff, r1Off, y1Off, uOff, vOff, outOff, st1Off, mr1Off, st2Off, mr2Off;   // forward (r2 aliases v)
  size_t dsOff, duOff, dy1Off, dr1Off, daOff, dxOff;                        // backward
  const float* xSaved = nullptr;
  bool uOnlyImages = false;             // mixed precision, this step: u exists only as uImg (set by forward, read by backward)
  BfLinear bl1, bl2;                    // mixed precision: lin1 (l -> l2), lin2 (l2 -> l)
  BfImage y1Img, uImg, dvImg, duImg;    // images of the two Linear inputs and of the two output gradients
  size_t convImgElems = 0, convImgFOff = 0, convImgBOff = 0;   // bf16 weight images of the convolution (0: fp32 kernels only)

  std::string name() const override { return "TDSBlock"; }
  void registerParams(std::vector<ParamInfo>& t) override {
    ParamInfo q;
    q = ParamInfo(); q.name = "tds.conv.w"; q.numel = (size_t)kw * c * c; q.refShape = {kw, 1, c, c}; q.kind = 1;
    q.kw = kw; q.cin = c; q.cout = c; q.initBound = std::sqrt(1.0 / ((double)c * kw)); addParam(t, q, wc);
    q = ParamInfo(); q.name = "tds.conv.b"; q.numel = c; q.refShape = {1, 1, c, 1}; q.initBound = std::sqrt(1.0 / ((double)c * kw)); addParam(t, q, bc);
    q = ParamInfo(); q.name = "tds.ln1.weight+bias"; q.numel = 2; q.refShape = {1}; q.kind = 3; addParam(t, q, gb1);
    q = ParamInfo(); q.name = "tds.lin1.w"; q.numel = (size_t)l * l2; q.refShape = {l2, l}; q.kind = 2; q.cin = l; q.cout = l2;
    q.initBound = std::sqrt(1.0 / (double)l); addParam(t, q, w1);
    q = ParamInfo(); q.name = "tds.lin1.b"; q.numel = l2; q.refShape = {l2}; q.initBound = std::sqrt(1.0 / (double)l); addParam(t, q, b1);
    q = ParamInfo(); q.name = "tds.lin2.w"; q.numel = (size_t)l2 * l; q.refShape = {l, l2}; q.kind = 2; q.cin = l2; q.cout = l;
    q.initBound = std::sqrt(1.0 / (double)l2); addParam(t, q, w2);
    q = ParamInfo(); q.name = "tds.lin2.b"; q.numel = l; q.refShape = {l}; q.initBound = std::sqrt(1.0 / (double)l2); addParam(t, q, b2);
    q = ParamInfo(); q.name = "tds.ln2.weight+bias"; q.numel = 2; q.refShape = {1}; q.kind = 3; addParam(t, q, gb2);
  }
  Act plan(const Act& in, Planner& pl) override {
    int H;
    requireConvLayout(in, c, "TDSBlock", H);
    if (H != h) throw std::invalid_argument("TDSBlock: expected H=" + std::to_string(h) + ", input " + in.str());
    int pL, pR;
    if (rPad < 0) { pL = pR = samePad(in.T, kw, 1); }
    else { if (rPad > kw - 1) throw std::invalid_argument("TDSBlock: invalid right padding"); pR = rPad; pL = kw - 1 - rPad; }
    d = {in.B, in.T, h, c, c, kw, 1, pL, pR};
    if (w2l_conv_out_len(in.T, kw, 1, pL, pR) != in.T) throw std::invalid_argument("TDSBlock: conv must preserve T (odd kw)");
    B = in.B; T = in.T; M = B * T; n = in.numel();
    groups = lnTime ? B : B * T;
    inner = n / groups;
    if (inner % 4) throw std::invalid_argument("TDSBlock: LayerNorm size must be a multiple of 4");
    aOff = pl.alloc(n); r1Off = pl.alloc(n); y1Off = pl.alloc(n); uOff = pl.alloc((size_t)M * l2); vOff = pl.alloc(n); outOff = pl.alloc(n);
    st1Off = pl.alloc(2 * w2l_layernorm_scratch_doubles(groups, inner)); mr1Off = pl.alloc(2 * (size_t)groups);
    st2Off = pl.alloc(2 * w2l_layernorm_scratch_doubles(groups, inner)); mr2Off = pl.alloc(2 * (size_t)groups);
    dsOff = pl.alloc(n); duOff = pl.alloc((size_t)M * l2); dy1Off = pl.alloc(n); dr1Off = pl.alloc(n);
    daOff = pl.alloc(n); dxOff = pl.alloc(n);
    bl1.plan(pl, M, l, l2); bl2.plan(pl, M, l2, l);
    y1Img.plan(pl, M, l, true); uImg.plan(pl, M, l2, true); dvImg.plan(pl, M, l); duImg.plan(pl, M, l2);
    convImgElems = h % 16 == 0 ? w2l_tds_conv_bf16_image_elems(&d) : 0;
    if (convImgElems) { convImgFOff = pl.allocBf16(convImgElems); convImgBOff = pl.allocBf16(convImgElems); }
    return in;
  }
  void forward(Ctx& cx, float* ar, const float* x, float*& y) override {
    hipStream_t s = cx.stream;
    const double pd = cx.train ? p : 0.0;
    xSaved = x;
    float *a = ar + aOff, *r1 = ar + r1Off, *y1 = ar + y1Off, *u = ar + uOff, *v = ar + vOff, *out = ar + outOff;
    if (cx.bf16 && convImgElems) {   // bf16 operands, fp32 accumulation / bias / ReLU (conv_tds_bf16.hip); images once per step
      w2lCheck(w2l_tds_conv_bf16_prepare(&d, wc.w(cx), bfp(ar, convImgFOff), bfp(ar, convImgBOff), s), "tds conv images");
      w2lCheck(w2l_tds_conv_bf16_forward(&d, x, bfp(ar, convImgFOff), bc.w(cx), a, 1, s), "tds conv bf16");
    } else {
      w2lCheck(w2l_conv_forward(&d, x, wc.w(cx), bc.w(cx), a, 1, s), "tds conv");
    }
    // a <- dropout(relu(conv)) in place (kept: its sign pattern is the ReLU+dropout mask); r1 = a + x; y1 = LN(r1)
    const bool y1Images = lnForward(cx, ar, groups, inner, a, x, r1, y1, gb1.w(cx), pd, cx.seed, rngStream, (double*)(ar + st1Off), ar + mr1Off,
                                    cx.bf16 ? &y1Img : nullptr, "tds ln1");
    if (cx.bf16) {
      // the same two products on bf16 images: y1 and u as images (row image for this product, transposed image for the weight
      // gradient in backward), the weights once per step (both orientations).  y1's images come out of the LayerNorm kernel when
      // it normalises per frame; otherwise y1 joins the weights' conversion launch (10 us each on their own: launch-bound)
      if (y1Images) {
        const w2l_bf16_convert_desc wd[2] = {bl1.w.desc(ar, w1.w(cx)), bl2.w.desc(ar, w2.w(cx))};
        w2lCheck(w2l_bf16_convert_multi(2, wd, s), "tds weight images");
      } else {
        const w2l_bf16_convert_desc wd[3] = {y1Img.desc(ar, y1), bl1.w.desc(ar, w1.w(cx)), bl2.w.desc(ar, w2.w(cx))};
        w2lCheck(w2l_bf16_convert_multi(3, wd, s), "tds y1 + weight images");
        y1Img.ensureOnes(cx, ar);
      }
      // u = dropout(relu(lin1(y1))) is only ever an operand of lin2's products and the mask of lin2's backward-data product: it lives
      // ONLY as its two bf16 images, written by lin1's epilogue (uOnlyImages; where that epilogue cannot run: fp32 u + conversion)
      uOnlyImages = bl1.forwardImages(cx, ar, y1Img, b1.w(cx), uImg, 1, pd, cx.seed, rngStream + 1);
      if (!uOnlyImages) {
        bl1.forward(cx, ar, y1Img, b1.w(cx), u, 1, pd, cx.seed, rngStream + 1);
        uImg.convert(cx, ar, u, "tds u images");
      }
      // (as in the fp32 branch below: second dropout + residual join in lin2's epilogue, a plain LayerNorm behind it)
      bl2.forwardAdd(cx, ar, uImg, b2.w(cx), y1, v, pd, cx.seed, rngStream + 2);
      w2lCheck(w2l_residual_layernorm_forward(groups, inner, v, nullptr, v, out, gb2.w(cx), 1e-5f, 0.0, 0, 0,
                                              (double*)(ar + st2Off), ar + mr2Off, s), "tds ln2");
      y = out;
      return;
    } else {
    // lin1 + ReLU + dropout in one GEMM epilogue (same mask bits as a separate dropout pass over u)
    if (pd > 0) w2lCheck(w2l_linear_forward_dropout(M, l, l2, y1, w1.w(cx), b1.w(cx), u, 1, pd, cx.seed, rngStream + 1, s), "tds lin1+do");
    else w2lCheck(w2l_linear_forward(M, l, l2, y1, w1.w(cx), b1.w(cx), u, 1, s), "tds lin1");
    // lin2 with the second dropout and the residual join in its epilogue: v <- r2 = dropout(lin2(u)) + y1 (the same mask bits as a
    // dropout pass over v: backward re-derives them from the hash), then a plain LayerNorm of r2 -- two tensor passes fewer than
    // residual_layernorm(v, y1) behind a plain lin2.  Where the product runs on the LDS-DMA kernels (l2 % 32 == 0: their epilogues
    // fetch the addend ahead of the LDS turn); the register-staged kernel pays more for the addend than the LayerNorm saves
    // (config 3 in fp32, l2 = 3600 ... 6480: profiles/r06_run24_c3_lin2_epilogue_ab.log).
    if (l2 % 32 == 0) {
    w2lCheck(w2l_linear_forward_dropout_add(M, l2, l, u, w2.w(cx), b2.w(cx), y1, v, 0, pd, cx.seed, rngStream + 2, s), "tds lin2+do+res");
    w2lCheck(w2l_residual_layernorm_forward(groups, inner, v, nullptr, v, out, gb2.w(cx), 1e-5f, 0.0, 0, 0,
                                            (double*)(ar + st2Off), ar + mr2Off, s), "tds ln2");
    y = out;
    return;
    }
    w2lCheck(w2l_linear_forward(M, l2, l, u, w2.w(cx), b2.w(cx), v, 0, s), "tds lin2");
    }
    // r2 = dropout(v) + y1 (stored over v), out = LN(r2)
    w2lCheck(w2l_residual_layernorm_forward(groups, inner, v, y1, v, out, gb2.w(cx), 1e-5f, pd, cx.seed, rngStream + 2,
                                            (double*)(ar + st2Off), ar + mr2Off, s), "tds ln2");
    y = out;
  }
  void backward(Ctx& cx, float* ar, const float* dy, float*& dx, bool needDx) override {
    hipStream_t s = cx.stream;
    const double pd = cx.train ? p : 0.0;
    const float sc = (float)(1.0 / (1.0 - pd));
    float *a = ar + aOff, *y1 = ar + y1Off, *u = ar + uOff, *v = ar + vOff;
    float *ds = ar + dsOff, *du = ar + duOff, *dy1 = ar + dy1Off, *dr1 = ar + dr1Off, *da = ar + daOff;
    // LN2 backward: ds = d r2 ; dv = ds masked by the dropout of v
    const bool rides2 = cx.bf16 && bl2.biasRides(uImg, w2.g(cx), b2.g(cx));
    // mixed precision with the bias gradient riding on the weight-gradient product: dv = dropout-masked ds is only ever a GEMM
    // operand -- its images are taken straight from ds with the mask applied on the way, no masked copy; where the LayerNorm
    // normalises per frame they come out of its backward kernel and there is no conversion launch either
    const bool maskInConvert = pd > 0 && rides2;
    const bool dvImages = lnBackward(cx, ar, groups, inner, v, dy, gb2.w(cx), ar + mr2Off, ds, gb2.g(cx), nullptr, nullptr, 1.f,
                                     (double*)(ar + st2Off), cx.bf16 && (pd == 0 || maskInConvert) ? &dvImg : nullptr, pd, cx.seed,
                                     rngStream + 2, "tds ln2 bwd");
    const float* dv = ds;
    if (pd > 0 && !maskInConvert) {
      // dy1 buffer doubles as scratch for the masked copy (one out-of-place pass)
      w2lCheck(w2l_dropout_copy(dy1, ds, n, pd, cx.seed, rngStream + 2, s), "tds do2 bwd");
      dv = dy1;
    }
    if (cx.bf16) {
      // (dv may live in dy1 -- the masked copy above --, which the last product overwrites: its images are taken first)
      if (dvImages) {}
      else if (maskInConvert) dvImg.convertDropout(cx, ar, ds, pd, cx.seed, rngStream + 2, "tds dv images (masked)");
      else dvImg.convert(cx, ar, dv, "tds dv images");
      bl2.backwardWeight(cx, ar, uImg, dvImg, w2.g(cx), rides2);
      if (!rides2) w2lCheck(w2l_colsum(dv, b2.g(cx), (size_t)M, l, s), "tds lin2 bwd b");
      // du = (dv W2^T) masked by u > 0: mask from u's bf16 image, and du itself -- an operand of lin1's two backward products and
      // nothing else once its bias gradient rides on the weight-gradient product -- leaves only as its images
      const bool rides1 = bl1.biasRides(y1Img, w1.g(cx), b1.g(cx));
      if (!(uOnlyImages && bl2.backwardDataImages(cx, ar, dvImg, rides1 ? nullptr : du, &duImg, uImg, sc))) {
        if (uOnlyImages) throw std::runtime_error("TDSBlock: u was kept as images only, but the image-writing backward-data product refused");
        bl2.backwardData(cx, ar, dvImg, du, u, sc, nullptr, 0);
        duImg.convert(cx, ar, du, "tds du images");
      }
      bl1.backwardWeight(cx, ar, y1Img, duImg, w1.g(cx), rides1);
      if (!rides1) w2lCheck(w2l_colsum(du, b1.g(cx), (size_t)M, l2, s), "tds lin1 bwd b");
      bl1.backwardData(cx, ar, duImg, dy1, nullptr, 1.f, ds, 0);
    } else {
    // lin2: dW2 = u^T dv, db2, du = (dv W2^T) masked by relu+dropout of u (u holds the dropped value)
    w2lCheck(w2l_linear_backward_weight_bias(M, l2, l, u, dv, w2.g(cx), b2.g(cx), s), "tds lin2 bwd w + b");
    w2lCheck(w2l_linear_backward_data(M, l2, l, dv, w2.w(cx), du, 0, u, sc, s), "tds lin2 bwd x");
    // lin1: dW1 = y1^T du, db1, dy1 = ds + du W1^T
    w2lCheck(w2l_linear_backward_weight_bias(M, l, l2, y1, du, w1.g(cx), b1.g(cx), s), "tds lin1 bwd w + b");
    // dy1 = ds + du W1^T: the residual join rides in the GEMM epilogue as a separate addend (no copy of ds into dy1)
    w2lCheck(w2l_linear_backward_data_add(M, l, l2, du, w1.w(cx), ds, dy1, s), "tds lin1 bwd x");
    }
    // LN1 backward: dr1, and in the same pass da = dr1 masked by the ReLU+dropout pattern of a
    w2lCheck(w2l_layernorm_backward(groups, inner, ar + r1Off, dy1, gb1.w(cx), ar + mr1Off, dr1, gb1.g(cx), a, da, sc,
                                    (double*)(ar + st1Off), s), "tds ln1 bwd");
    if (cx.bf16 && convImgElems) {
      w2lCheck(w2l_tds_conv_bf16_backward_filter_bias(&d, xSaved, da, wc.g(cx), bc.g(cx), s), "tds conv bwd filter + bias bf16");
      if (needDx) {
        dx = ar + dxOff;
        w2lCheck(w2l_tds_conv_bf16_backward_data(&d, da, bfp(ar, convImgBOff), dr1, dx, s), "tds conv bwd data bf16");
      }
      return;
    }
    w2lCheck(w2l_conv_backward_filter(&d, xSaved, da, wc.g(cx), bc.g(cx), s), "tds conv bwd filter");
    if (needDx) {
      dx = ar + dxOff;
      w2lCheck(w2l_conv_backward_data_add(&d, da, wc.w(cx), dr1, dx, s), "tds conv bwd data");
    }
  }
};

// fl::Pool2D(wx, 1, sx, 1, 0, 0, MAX) on (T, H, C, B): the `M 1 1 2 1` lines between the front end's convolutions
// (recipes/sota/2019/am_arch/am_transformer_ctc.arch:5; grammar SequentialBuilder.cpp:398-414)
class PoolTimeLayer : public Layer {
 public:
  int w = 1, stride = 1;
  int B = 0, T = 0, To = 0, F = 0;
  size_t yOff = 0, dxOff = 0;
  const float* xSaved = nullptr;
  std::string name() const override { return "Pool2D"; }
  Act plan(const Act& in, Planner& pl) override {
    if (!isKind(in.d[0], F_TIME) || !isKind(in.d[3], F_BATCH))
      throw std::invalid_argument("Pool2D: input must be (T, H, C, B), got " + in.str());
    if (in.T < w) throw std::invalid_argument("Pool2D: fewer frames than the window");
    B = in.B; T = in.T; F = in.F;
    To = (T - w) / stride + 1;
    Act o = in;
    o.T = To;
    yOff = pl.alloc((size_t)B * To * F);
    dxOff = pl.alloc(in.numel());
    return o;
  }
  void forward(Ctx& c, float* arena, const float* x, float*& y) override {
    y = arena + yOff;
    xSaved = x;
    w2lCheck(w2l_pool_time_forward(x, y, B, T, F, w, stride, c.stream), "pool fwd");
  }
  void backward(Ctx& c, float* arena, const float* dy, float*& dx, bool needDx) override {
    if (!needDx) return;
    dx = arena + dxOff;
    w2lCheck(w2l_pool_time_backward(xSaved, dy, dx, B, T, F, w, stride, c.stream), "pool bwd");
  }
};

// fl::Transformer(modelDim, modelDim / nHeads, mlpDim, nHeads, csz, pDropout, pLayerdrop) -- arch token `TR`
// (SequentialBuilder.cpp:136-158).  Block as the in-repo copy states it, recipes/joint_training_vox_populi/cpc/
// TransformerCPC.cpp: parameters and their order :41-95 (position table, w1, w2, wq, wk, wv, wf, norm1, norm2), post-LN data
// flow :153-182, mlp :97-101 (no dropout inside), selfAttention :117-151 (q scaled by 1/sqrt(d), dropout on the attention
// probabilities only; padding mask of the keys when the caller supplies the batch's input sizes, Ctx::inputSizes).
//   h   = LN1(f * Wf attention(x) + x)          f = 0 with probability pLayerdrop while training, else 1
//   out = LN2(f * W2 relu(W1 h) + h)
// Input (C, T, B, 1) == frame-major [B*T][C]; heads are contiguous slices of a frame (moddims(T, d, heads*B)).
class TransformerLayer : public Layer {
 public:
  int C = 0, mlp = 0, nH = 0, csz = 0;
  double p = 0, pLayerDrop = 0;
  int d = 0;
  P pe, w1, b1, w2, b2, wq, bq, wk, bk, wv, bv, wf, bf, gb1, gb2;
  int B = 0, T = 0, M = 0, W = 0, rlo = 0, ldr = 0, n0 = 0;
  size_t qOff, kOff, vOff, sOff, pdOff, rOff, ctxOff, oOff, hOff, uOff, m2Off, outOff, st1Off, mr1Off, st2Off, mr2Off;
  size_t ds2Off, duOff, dhOff, dr1Off, dctxOff, dsOff, dRoff, dqOff, dkOff, dvOff, dxOff, dEpOff, klOff;
  size_t abwOff = 0, abwBytes = 0;   // workspace of the fused attention backward (0: geometry without a fused kernel)
  const float* xSaved = nullptr;
  bool dropped = false, fusedFwd = false, uOnlyImages = false;
  // mixed precision: the six fl::Linear of the block on bf16 images (the attention products stay on the fp32 batched GEMM)
  BfLinear blq, blk, blv, blf, bl1, bl2;
  BfImage xImg, ctxImg, hImg, uImg, dqImg, dkImg, dvImg, dr1Img, duImg, ds2Img;
  BfImage outImg;           // images of the block's output, written by its last LayerNorm: the next block's x images (Ctx::imgOf)
  BfImage xSeen;            // the images of x this step's products read: xImg, or the previous block's outImg

  std::string name() const override { return "Transformer"; }
  void registerParams(std::vector<ParamInfo>& t) override {
    d = C / nH;
    auto lin = [&](const char* nm, int in, int out, double wBound, double bBound, P& w, P& b) {
      ParamInfo q;
      q.name = std::string(nm) + ".w"; q.numel = (size_t)in * out; q.refShape = {out, in}; q.kind = 2; q.cin = in; q.cout = out;
      q.initBound = wBound; addParam(t, q, w);
      q = ParamInfo(); q.name = std::string(nm) + ".b"; q.numel = out; q.refShape = {out}; q.initBound = bBound; addParam(t, q, b);
    };
    if (csz > 0) {
      ParamInfo q;  // ArrayFire (2 csz - 1, d), column-major == memory [d][2 csz - 1]; internal [2 csz - 1][d]
      q.name = "tr.posemb"; q.numel = (size_t)(2 * csz - 1) * d; q.refShape = {2 * csz - 1, d}; q.kind = 5;
      q.cin = 2 * csz - 1; q.cout = d; q.initBound = 0.1; addParam(t, q, pe);
    }
    const double g = 0.707 * std::sqrt(6.0 / (2.0 * C));
    lin("tr.w1", C, mlp, std::sqrt(1.0 / C), std::sqrt(1.0 / C), w1, b1);
    lin("tr.w2", mlp, C, std::sqrt(1.0 / mlp), std::sqrt(1.0 / mlp), w2, b2);
    lin("tr.wq", C, C, g, std::sqrt(1.0 / C), wq, bq);
    lin("tr.wk", C, C, g, std::sqrt(1.0 / C), wk, bk);
    lin("tr.wv", C, C, g, std::sqrt(1.0 / C), wv, bv);
    lin("tr.wf", C, C, std::sqrt(6.0 / (2.0 * C)), 0.0, wf, bf);
    ParamInfo q;
    q.name = "tr.norm1.weight+bias"; q.numel = 2; q.refShape = {1}; q.kind = 3; addParam(t, q, gb1);
    q = ParamInfo(); q.name = "tr.norm2.weight+bias"; q.numel = 2; q.refShape = {1}; q.kind = 3; addParam(t, q, gb2);
  }
  Act plan(const Act& in, Planner& pl) override {
    bool ok = in.d[0].f.size() == 1 && in.d[0].f[0].kind == F_FEAT && in.d[0].f[0].stride == 1 && in.d[0].size() == C && in.F == C &&
              isKind(in.d[1], F_TIME) && isKind(in.d[2], F_BATCH) && in.d[3].f.empty();
    if (!ok) throw std::invalid_argument("Transformer(" + std::to_string(C) + "): input must be (C, T, B, 1), got " + in.str());
    if (C % nH || (C / nH) % 4) throw std::invalid_argument("Transformer: head size must be a multiple of 4");
    B = in.B; T = in.T; M = B * T;
    n0 = csz - 1;
    if (csz > 0) {  // table rows j - i + n0 a T-frame utterance reaches, clipped to the table
      rlo = std::max(0, n0 - (T - 1));
      W = std::min(2 * csz - 1, n0 + T) - rlo;
      ldr = (W + 3) / 4 * 4;
    } else { rlo = W = ldr = 0; }
    const size_t n = (size_t)M * C, ns = (size_t)B * nH * T * T, nr = (size_t)M * nH * ldr;
    qOff = pl.alloc(n); kOff = pl.alloc(n); vOff = pl.alloc(n); sOff = pl.alloc(ns); pdOff = p > 0 ? pl.alloc(ns) : sOff;
    rOff = pl.alloc(nr); ctxOff = pl.alloc(n); oOff = pl.alloc(n); hOff = pl.alloc(n); uOff = pl.alloc((size_t)M * mlp);
    m2Off = pl.alloc(n); outOff = pl.alloc(n);
    st1Off = pl.alloc(2 * w2l_layernorm_scratch_doubles(M, C)); mr1Off = pl.alloc(2 * (size_t)M);
    st2Off = pl.alloc(2 * w2l_layernorm_scratch_doubles(M, C)); mr2Off = pl.alloc(2 * (size_t)M);
    ds2Off = pl.alloc(n); duOff = pl.alloc((size_t)M * mlp); dhOff = pl.alloc(n); dr1Off = pl.alloc(n); dctxOff = pl.alloc(n);
    dsOff = pl.alloc(ns); dRoff = pl.alloc(nr); dqOff = pl.alloc(n); dkOff = pl.alloc(n); dvOff = pl.alloc(n); dxOff = pl.alloc(n);
    dEpOff = pl.alloc((size_t)B * W * d);
    klOff = pl.alloc((size_t)B);
    {
      const w2l_attn_fused_desc fd = fusedDesc(0.0, 0);
      abwBytes = w2l_attn_fused_backward_workspace(&fd, csz > 0 ? 1 : 0);
      if (abwBytes) abwOff = pl.alloc((abwBytes + 3) / 4);   // (arena slots are 256-byte aligned: 64 floats)
    }
    blq.plan(pl, M, C, C); blk.plan(pl, M, C, C); blv.plan(pl, M, C, C); blf.plan(pl, M, C, C); bl1.plan(pl, M, C, mlp); bl2.plan(pl, M, mlp, C);
    xImg.plan(pl, M, C, true); ctxImg.plan(pl, M, C, true); hImg.plan(pl, M, C, true); uImg.plan(pl, M, mlp, true);
    dqImg.plan(pl, M, C); dkImg.plan(pl, M, C); dvImg.plan(pl, M, C); dr1Img.plan(pl, M, C); duImg.plan(pl, M, mlp); ds2Img.plan(pl, M, C);
    outImg.plan(pl, M, C, true);
    return in;
  }
  w2l_attn_fused_desc fusedDesc(double pd, uint32_t seed) const {
    w2l_attn_fused_desc fd{};
    fd.B = B; fd.H = nH; fd.T = T; fd.d = d; fd.ld = C; fd.ldc = C; fd.W = W; fd.n0 = n0; fd.rlo = rlo;
    fd.scale = (float)(1.0 / std::sqrt((double)d));
    fd.dropP = pd; fd.dropSeed = seed; fd.dropStream = (uint32_t)rngStream;
    return fd;
  }
  // per-(utterance, head) product over frame-major operands; see w2l_bgemm_desc
  w2l_bgemm_desc heads(int Mm, int Nn, int Kk) const {
    w2l_bgemm_desc g{};
    g.M = Mm; g.N = Nn; g.K = Kk; g.G1 = B; g.G2 = nH;
    return g;
  }
  void forward(Ctx& cx, float* ar, const float* x, float*& y) override {
    hipStream_t s = cx.stream;
    auto bg = cx.bf16 ? w2l_bgemm_bf16 : w2l_bgemm_f32;   // mixed precision: the attention products multiply in bf16 too
    xSaved = x;
    const double pd = cx.train ? p : 0.0;
    dropped = false;
    if (cx.train && pLayerDrop > 0) {  // one draw per block and step (af::randu(1) in the reference)
      uint64_t st = ((uint64_t)cx.seed << 32) ^ (uint64_t)(rngStream + 3) * 0x9E3779B97F4A7C15ull;
      st = (st ^ (st >> 30)) * 0xBF58476D1CE4E5B9ull; st = (st ^ (st >> 27)) * 0x94D049BB133111EBull; st ^= st >> 31;
      dropped = (double)(st >> 11) * (1.0 / 9007199254740992.0) < pLayerDrop;
    }
    float *q = ar + qOff, *k = ar + kOff, *v = ar + vOff, *S = ar + sOff, *Pd = ar + pdOff, *R = ar + rOff, *ctx = ar + ctxOff;
    float *o = ar + oOff, *h = ar + hOff, *u = ar + uOff, *m2 = ar + m2Off, *out = ar + outOff;
    float* xm = const_cast<float*>(x);
    if (dropped) {  // f = 0: both sublayers vanish, the two LayerNorms remain
      w2lCheck(w2l_residual_layernorm_forward(M, C, xm, nullptr, xm, h, gb1.w(cx), 1e-5f, 0.0, 0, 0, (double*)(ar + st1Off), ar + mr1Off, s), "tr ln1");
      w2lCheck(w2l_residual_layernorm_forward(M, C, h, nullptr, h, out, gb2.w(cx), 1e-5f, 0.0, 0, 0, (double*)(ar + st2Off), ar + mr2Off, s), "tr ln2");
      y = out;
      return;
    }
    const bool mixed = cx.bf16;
    if (mixed) {   // one pair of images of x serves the three projections (and their weight gradients in backward)
      if (cx.imgOf == x && cx.imgRows == M && cx.imgCols == C) {   // the previous block's LayerNorm wrote them already
        xSeen = xImg;
        xSeen.rowsOff = cx.imgRowsOff; xSeen.transOff = cx.imgTransOff;
      } else {
        xImg.convert(cx, ar, x, "tr x images");
        xSeen = xImg;
      }
      {   // the six weights of the block in ONE conversion launch (8 us each on their own: launch-bound)
        const w2l_bf16_convert_desc wd[6] = {blq.w.desc(ar, wq.w(cx)), blk.w.desc(ar, wk.w(cx)), blv.w.desc(ar, wv.w(cx)),
                                             blf.w.desc(ar, wf.w(cx)), bl1.w.desc(ar, w1.w(cx)), bl2.w.desc(ar, w2.w(cx))};
        w2lCheck(w2l_bf16_convert_multi(6, wd, s), "tr weight images");
      }
      {   // the three projections in ONE grouped launch: 3 x 192 tiles share the grid (one at a time each fills 3/8 of the slots)
        const uint16_t* A3[3] = {xSeen.r(ar), xSeen.r(ar), xSeen.r(ar)};
        const uint16_t* B3[3] = {blq.w.t(ar), blk.w.t(ar), blv.w.t(ar)};
        float* C3[3] = {q, k, v};
        const float* b3[3] = {bq.w(cx), bk.w(cx), bv.w(cx)};
        w2lCheck(w2l_gemm_bf16_grouped(3, M, C, C, A3, xSeen.colsP, B3, blq.w.rowsP, C3, C, b3, s), "tr q k v");
      }
    } else {
    w2lCheck(w2l_linear_forward(M, C, C, x, wq.w(cx), bq.w(cx), q, 0, s), "tr q");
    w2lCheck(w2l_linear_forward(M, C, C, x, wk.w(cx), bk.w(cx), k, 0, s), "tr k");
    w2lCheck(w2l_linear_forward(M, C, C, x, wv.w(cx), bv.w(cx), v, 0, s), "tr v");
    }
    const long long TC = (long long)T * C, TT = (long long)T * T;
    const int* keyLen = nullptr;
    if (cx.inputSizes) {  // padding mask of the keys (cpc/SequentialBuilder.cpp:58-81, TransformerCPC.cpp:138-144)
      int* kl = (int*)(ar + klOff);
      w2lCheck(w2l_attn_key_lengths_full(cx.inputSizes, cx.inputSizeFull, B, cx.inputT, T, kl, s), "tr key lengths");
      keyLen = kl;
    }
    const float scale = (float)(1.0 / std::sqrt((double)d));
    bool fused = false;
    if (mixed) {   // scores, position term, softmax, dropout and P V in one launch where the geometry has a fused kernel
      // (ctx is only ever the operand of the wf products in this mode: the kernel writes its bf16 images, no fp32 copy)
      const w2l_attn_fused_desc fd = fusedDesc(pd, cx.seed);
      const w2l_bf16_image_sink ctxSink = ctxImg.sink(cx, ar);
      const int st = w2l_attn_fused_forward_images(&fd, q, k, v, csz > 0 ? pe.w(cx) : nullptr, keyLen, S, pd > 0 ? Pd : nullptr, nullptr,
                                                   &ctxSink, s);
      if (st != W2L_EUNSUPPORTED) w2lCheck(st, "tr fused attention");
      fused = st == W2L_OK;
    }
    fusedFwd = fused;
    if (!fused) {
    {  // S[b][h][i][j] = q_i . k_j
      w2l_bgemm_desc g = heads(T, T, d);
      g.sam = C; g.sak = 1; g.a1 = TC; g.a2 = d; g.sbk = 1; g.sbn = C; g.b1 = TC; g.b2 = d; g.ldc = T; g.c1 = nH * TT; g.c2 = TT;
      w2lCheck(bg(&g, q, k, S, s), "tr qk");
    }
    if (csz > 0) {  // R[(b, i, h)][w] = q_i . E[rlo + w]
      w2l_bgemm_desc g{};
      g.M = M * nH; g.N = W; g.K = d; g.G1 = g.G2 = 1; g.sam = d; g.sak = 1; g.sbk = 1; g.sbn = d; g.ldc = ldr;
      w2lCheck(bg(&g, q, pe.w(cx) + (size_t)rlo * d, R, s), "tr qE");
    }
    w2lCheck(w2l_attn_softmax_forward(S, csz > 0 ? R : nullptr, keyLen, B, nH, T, ldr, rlo, W, n0, scale, s), "tr softmax");
    if (pd > 0) w2lCheck(w2l_dropout_copy(Pd, S, (size_t)B * nH * TT, pd, cx.seed, rngStream, s), "tr attn dropout");
    {  // ctx_i = sum_j P[i][j] v_j
      w2l_bgemm_desc g = heads(T, d, T);
      g.sam = T; g.sak = 1; g.a1 = nH * TT; g.a2 = TT; g.sbk = C; g.sbn = 1; g.b1 = TC; g.b2 = d; g.ldc = C; g.c1 = TC; g.c2 = d;
      w2lCheck(bg(&g, pd > 0 ? Pd : S, v, ctx, s), "tr pv");
    }
    }
    if (mixed) {
      if (!fused) ctxImg.convert(cx, ar, ctx, "tr ctx images");
      blf.forwardAdd(cx, ar, ctxImg, bf.w(cx), x, o, 0.0, 0, 0);
    } else {
      w2lCheck(w2l_linear_forward_dropout_add(M, C, C, ctx, wf.w(cx), bf.w(cx), x, o, 0, 0.0, 0, 0, s), "tr wf + res");
    }
    // o = r1 = wf(ctx) + x (the residual join in the product's epilogue), h = LN1(r1)
    const bool hImages = lnForward(cx, ar, M, C, o, nullptr, o, h, gb1.w(cx), 0.0, 0, 0, (double*)(ar + st1Off), ar + mr1Off, mixed ? &hImg : nullptr, "tr ln1");
    if (mixed) {
      if (!hImages) hImg.convert(cx, ar, h, "tr h images");
      // u = relu(w1 h) is only ever an operand of w2's products and the mask of w2's backward-data product: bf16 images only
      uOnlyImages = bl1.forwardImages(cx, ar, hImg, b1.w(cx), uImg, 1, 0.0, 0, 0);
      if (!uOnlyImages) {
        bl1.forward(cx, ar, hImg, b1.w(cx), u, 1, 0.0, 0, 0);
        uImg.convert(cx, ar, u, "tr u images");
      }
      bl2.forwardAdd(cx, ar, uImg, b2.w(cx), h, m2, 0.0, 0, 0);
    } else {
    w2lCheck(w2l_linear_forward(M, C, mlp, h, w1.w(cx), b1.w(cx), u, 1, s), "tr w1");
    w2lCheck(w2l_linear_forward_dropout_add(M, mlp, C, u, w2.w(cx), b2.w(cx), h, m2, 0, 0.0, 0, 0, s), "tr w2 + res");
    }
    // m2 = r2 = w2(u) + h
    const bool outImages = lnForward(cx, ar, M, C, m2, nullptr, m2, out, gb2.w(cx), 0.0, 0, 0, (double*)(ar + st2Off), ar + mr2Off,
                                     mixed ? &outImg : nullptr, "tr ln2");
    if (outImages) {   // a following Transformer block of the same width reads them as its x images
      cx.imgOf = out; cx.imgRowsOff = outImg.rowsOff; cx.imgTransOff = outImg.transOff; cx.imgRows = M; cx.imgCols = C;
    }
    y = out;
  }
  // the four projection bias gradients are row C of the grouped weight-gradient product (ones rows in the x / ctx images)
  bool projectionBiasRides(Ctx& cx) const {
    return blq.biasRides(xImg, wq.g(cx), bq.g(cx)) && blk.biasRides(xImg, wk.g(cx), bk.g(cx)) && blv.biasRides(xImg, wv.g(cx), bv.g(cx)) &&
           blf.biasRides(ctxImg, wf.g(cx), bf.g(cx));
  }
  void zeroGrad(Ctx& cx, const P& w, hipStream_t s) {
    const ParamInfo& pi = (*w.table)[w.idx];
    w2lCheck(w2l_fill(w.g(cx), pi.numel, 0.f, s), "tr zero grad");
  }
  void backward(Ctx& cx, float* ar, const float* dy, float*& dx, bool needDx) override {
    hipStream_t s = cx.stream;
    auto bg = cx.bf16 ? w2l_bgemm_bf16 : w2l_bgemm_f32;
    const double pd = cx.train ? p : 0.0;
    float *q = ar + qOff, *k = ar + kOff, *v = ar + vOff, *S = ar + sOff, *Pd = ar + pdOff, *ctx = ar + ctxOff;
    float *o = ar + oOff, *h = ar + hOff, *u = ar + uOff, *m2 = ar + m2Off;
    float *ds2 = ar + ds2Off, *du = ar + duOff, *dh = ar + dhOff, *dr1 = ar + dr1Off, *dctx = ar + dctxOff, *dS = ar + dsOff;
    float *dR = ar + dRoff, *dq = ar + dqOff, *dk = ar + dkOff, *dv = ar + dvOff, *dEp = ar + dEpOff;
    (void)needDx;  // a Transformer block is never the first parameterised layer
    dx = ar + dxOff;
    if (dropped) {
      w2lCheck(w2l_layernorm_backward(M, C, h, dy, gb2.w(cx), ar + mr2Off, dh, gb2.g(cx), nullptr, nullptr, 1.f, (double*)(ar + st2Off), s), "tr ln2 bwd");
      w2lCheck(w2l_layernorm_backward(M, C, xSaved, dh, gb1.w(cx), ar + mr1Off, dx, gb1.g(cx), nullptr, nullptr, 1.f, (double*)(ar + st1Off), s), "tr ln1 bwd");
      {   // the block's weight / bias / table gradients are one contiguous run of the arena: one fill, not thirteen
        size_t lo = (size_t)-1, hi = 0;
        std::vector<const P*> ps = {&w1, &b1, &w2, &b2, &wq, &bq, &wk, &bk, &wv, &bv, &wf, &bf};
        if (csz > 0) ps.push_back(&pe);
        size_t total = 0;
        for (const P* w : ps) {
          const ParamInfo& pi = (*w->table)[w->idx];
          lo = std::min(lo, pi.offset); hi = std::max(hi, pi.offset + pi.numel);
          total += (pi.numel + 3) / 4 * 4;
        }
        // gb1 / gb2 (the LayerNorm pairs, written by the two LayerNorm backward calls above) may lie inside the run: only fill
        // the run when nothing else does (slots are 4-float aligned, so a gap-free run has hi - lo within the padded total)
        if (hi - lo <= total) w2lCheck(w2l_fill(cx.grads + lo, hi - lo, 0.f, s), "tr zero grads");
        else for (const P* w : ps) zeroGrad(cx, *w, s);
      }
      return;
    }
    const long long TC = (long long)T * C, TT = (long long)T * T;
    const bool mixed = cx.bf16;
    const bool ds2Images = lnBackward(cx, ar, M, C, m2, dy, gb2.w(cx), ar + mr2Off, ds2, gb2.g(cx), nullptr, nullptr, 1.f, (double*)(ar + st2Off),
                                      mixed ? &ds2Img : nullptr, 0.0, 0, 0, "tr ln2 bwd");
    if (mixed) {
      if (!ds2Images) ds2Img.convert(cx, ar, ds2, "tr ds2 images");
      const bool rides2 = bl2.biasRides(uImg, w2.g(cx), b2.g(cx));
      bl2.backwardWeight(cx, ar, uImg, ds2Img, w2.g(cx), rides2);
      if (!rides2) w2lCheck(w2l_colsum(ds2, b2.g(cx), (size_t)M, C, s), "tr w2 bwd b");
      const bool rides1 = bl1.biasRides(hImg, w1.g(cx), b1.g(cx));
      if (!(uOnlyImages && bl2.backwardDataImages(cx, ar, ds2Img, rides1 ? nullptr : du, &duImg, uImg, 1.f))) {
        if (uOnlyImages) throw std::runtime_error("Transformer: u was kept as images only, but the image-writing backward-data product refused");
        bl2.backwardData(cx, ar, ds2Img, du, u, 1.f, nullptr, 0);
        duImg.convert(cx, ar, du, "tr du images");
      }
      bl1.backwardWeight(cx, ar, hImg, duImg, w1.g(cx), rides1);
      if (!rides1) w2lCheck(w2l_colsum(du, b1.g(cx), (size_t)M, mlp, s), "tr w1 bwd b");
      bl1.backwardData(cx, ar, duImg, dh, nullptr, 1.f, ds2, 0);
    } else {
    w2lCheck(w2l_linear_backward_weight(M, mlp, C, u, ds2, w2.g(cx), s), "tr w2 bwd w");
    w2lCheck(w2l_colsum(ds2, b2.g(cx), (size_t)M, C, s), "tr w2 bwd b");
    w2lCheck(w2l_linear_backward_data(M, mlp, C, ds2, w2.w(cx), du, 0, u, 1.f, s), "tr w2 bwd x");
    w2lCheck(w2l_linear_backward_weight(M, C, mlp, h, du, w1.g(cx), s), "tr w1 bwd w");
    w2lCheck(w2l_colsum(du, b1.g(cx), (size_t)M, mlp, s), "tr w1 bwd b");
    w2lCheck(w2l_linear_backward_data_add(M, C, mlp, du, w1.w(cx), ds2, dh, s), "tr w1 bwd x");
    }
    const bool dr1Images = lnBackward(cx, ar, M, C, o, dh, gb1.w(cx), ar + mr1Off, dr1, gb1.g(cx), nullptr, nullptr, 1.f, (double*)(ar + st1Off),
                                      mixed ? &dr1Img : nullptr, 0.0, 0, 0, "tr ln1 bwd");
    if (mixed) {
      if (!dr1Images) dr1Img.convert(cx, ar, dr1, "tr dr1 images");
      // (wf's weight and bias gradients join those of wq / wk / wv in one grouped launch at the end of this function)
      blf.backwardData(cx, ar, dr1Img, dctx, nullptr, 1.f, nullptr, 0);
    } else {
    w2lCheck(w2l_linear_backward_weight(M, C, C, ctx, dr1, wf.g(cx), s), "tr wf bwd w");
    w2lCheck(w2l_colsum(dr1, bf.g(cx), (size_t)M, C, s), "tr wf bwd b");
    w2lCheck(w2l_linear_backward_data(M, C, C, dr1, wf.w(cx), dctx, 0, nullptr, 1.f, s), "tr wf bwd x");
    }
    bool fusedBwd = false;
    if (mixed && fusedFwd && abwBytes) {   // dP, dropout mask, softmax backward, dq (+ position term), dk, dv, table gradient: four launches
      // dq / dk / dv are only ever GEMM operands once their bias gradients ride on the weight-gradient products: bf16 images from the
      // kernels, no fp32 copies, no conversion launch
      const w2l_attn_fused_desc fd = fusedDesc(pd, cx.seed);
      const bool rides = projectionBiasRides(cx);
      const w2l_bf16_image_sink sq = dqImg.sink(cx, ar), sk = dkImg.sink(cx, ar), sv = dvImg.sink(cx, ar);
      const int st = w2l_attn_fused_backward_images(&fd, q, k, v, csz > 0 ? pe.w(cx) : nullptr, S, dctx, rides ? nullptr : dq,
                                                    rides ? nullptr : dk, rides ? nullptr : dv, &sq, &sk, &sv,
                                                    csz > 0 ? pe.g(cx) : nullptr, ar + abwOff, abwBytes, s);
      if (st != W2L_EUNSUPPORTED) w2lCheck(st, "tr fused attention backward");
      fusedBwd = st == W2L_OK;
    }
    if (!fusedBwd) {
    {  // dPd[i][j] = dctx_i . v_j
      w2l_bgemm_desc g = heads(T, T, d);
      g.sam = C; g.sak = 1; g.a1 = TC; g.a2 = d; g.sbk = 1; g.sbn = C; g.b1 = TC; g.b2 = d; g.ldc = T; g.c1 = nH * TT; g.c2 = TT;
      w2lCheck(bg(&g, dctx, v, dS, s), "tr dP");
    }
    {  // dv_j = sum_i Pd[i][j] dctx_i
      w2l_bgemm_desc g = heads(T, d, T);
      g.sam = 1; g.sak = T; g.a1 = nH * TT; g.a2 = TT; g.sbk = C; g.sbn = 1; g.b1 = TC; g.b2 = d; g.ldc = C; g.c1 = TC; g.c2 = d;
      w2lCheck(bg(&g, pd > 0 ? Pd : S, dctx, dv, s), "tr dv");
    }
    if (pd > 0) w2lCheck(w2l_dropout_inplace(dS, (size_t)B * nH * TT, pd, cx.seed, rngStream, s), "tr attn dropout bwd");
    w2lCheck(w2l_attn_softmax_backward(S, dS, csz > 0 ? dR : nullptr, B, nH, T, ldr, rlo, W, n0, (float)(1.0 / std::sqrt((double)d)), s), "tr softmax bwd");
    {  // dq_i = sum_j dS[i][j] k_j
      w2l_bgemm_desc g = heads(T, d, T);
      g.sam = T; g.sak = 1; g.a1 = nH * TT; g.a2 = TT; g.sbk = C; g.sbn = 1; g.b1 = TC; g.b2 = d; g.ldc = C; g.c1 = TC; g.c2 = d;
      w2lCheck(bg(&g, dS, k, dq, s), "tr dq");
    }
    {  // dk_j = sum_i dS[i][j] q_i
      w2l_bgemm_desc g = heads(T, d, T);
      g.sam = 1; g.sak = T; g.a1 = nH * TT; g.a2 = TT; g.sbk = C; g.sbn = 1; g.b1 = TC; g.b2 = d; g.ldc = C; g.c1 = TC; g.c2 = d;
      w2lCheck(bg(&g, dS, q, dk, s), "tr dk");
    }
    if (csz > 0) {
      const float* Ew = pe.w(cx) + (size_t)rlo * d;
      {  // dq_(b,i,h) += sum_w dR[(b,i,h)][w] E[rlo + w]
        w2l_bgemm_desc g{};
        g.M = M * nH; g.N = d; g.K = W; g.G1 = g.G2 = 1; g.sam = ldr; g.sak = 1; g.sbk = d; g.sbn = 1; g.ldc = d; g.accumulate = 1;
        g.bandMode = 1; g.bandT = T; g.bandH = nH; g.bandOff = n0 - rlo;   // row (b, i, h) of dR is zero outside w = j - i + n0 - rlo
        w2lCheck(bg(&g, dR, Ew, dq, s), "tr dq rel");
      }
      {  // dE[rlo + w] = sum_(b,i,h) dR[(b,i,h)][w] q_(b,i,h): one partial per utterance, then a column sum
        w2l_bgemm_desc g{};
        g.M = W; g.N = d; g.K = T * nH; g.G1 = B; g.G2 = 1; g.sam = 1; g.sak = ldr; g.a1 = (long long)T * nH * ldr;
        g.sbk = d; g.sbn = 1; g.b1 = TC; g.ldc = d; g.c1 = (long long)W * d;
        g.bandMode = 2; g.bandT = T; g.bandH = nH; g.bandOff = n0 - rlo;
        w2lCheck(bg(&g, dR, q, dEp, s), "tr dE");
        zeroGrad(cx, pe, s);
        w2lCheck(w2l_colsum(dEp, pe.g(cx) + (size_t)rlo * d, (size_t)B, W * d, s), "tr dE sum");
      }
    }
    }
    if (mixed) {
      if (!fusedBwd) {
        const w2l_bf16_convert_desc gd[3] = {dqImg.desc(ar, dq), dkImg.desc(ar, dk), dvImg.desc(ar, dv)};
        w2lCheck(w2l_bf16_convert_multi(3, gd, s), "tr dq / dk / dv images");
      }
      {   // the four C x C weight gradients (64 tiles each at the recipe's width) in ONE grouped launch
        const uint16_t* A4[4] = {xSeen.t(ar), xSeen.t(ar), xSeen.t(ar), ctxImg.t(ar)};
        const uint16_t* B4[4] = {dqImg.t(ar), dkImg.t(ar), dvImg.t(ar), dr1Img.t(ar)};
        float* C4[4] = {wq.g(cx), wk.g(cx), wv.g(cx), wf.g(cx)};
        // with the ones rows of the x / ctx images the four bias gradients are row C of the four products
        const bool rides = projectionBiasRides(cx);
        w2lCheck(w2l_gemm_bf16_grouped(4, C + (rides ? 1 : 0), C, M, A4, xSeen.rowsP, B4, dqImg.rowsP, C4, C, nullptr, s),
                 "tr projection weight gradients");
        if (!rides) {
          w2lCheck(w2l_colsum(dr1, bf.g(cx), (size_t)M, C, s), "tr wf bwd b");
          w2lCheck(w2l_colsum(dq, bq.g(cx), (size_t)M, C, s), "tr wq bwd b");
          w2lCheck(w2l_colsum(dk, bk.g(cx), (size_t)M, C, s), "tr wk bwd b");
          w2lCheck(w2l_colsum(dv, bv.g(cx), (size_t)M, C, s), "tr wv bwd b");
        }
      }
      blq.backwardData(cx, ar, dqImg, dx, nullptr, 1.f, dr1, 0);
      blk.backwardData(cx, ar, dkImg, dx, nullptr, 1.f, nullptr, 1);
      blv.backwardData(cx, ar, dvImg, dx, nullptr, 1.f, nullptr, 1);
      return;
    }
    w2lCheck(w2l_linear_backward_weight(M, C, C, xSaved, dq, wq.g(cx), s), "tr wq bwd w");
    w2lCheck(w2l_colsum(dq, bq.g(cx), (size_t)M, C, s), "tr wq bwd b");
    w2lCheck(w2l_linear_backward_weight(M, C, C, xSaved, dk, wk.g(cx), s), "tr wk bwd w");
    w2lCheck(w2l_colsum(dk, bk.g(cx), (size_t)M, C, s), "tr wk bwd b");
    w2lCheck(w2l_linear_backward_weight(M, C, C, xSaved, dv, wv.g(cx), s), "tr wv bwd w");
    w2lCheck(w2l_colsum(dv, bv.g(cx), (size_t)M, C, s), "tr wv bwd b");
    w2lCheck(w2l_linear_backward_data_add(M, C, C, dq, wq.w(cx), dr1, dx, s), "tr wq bwd x");
    w2lCheck(w2l_linear_backward_data(M, C, C, dk, wk.w(cx), dx, 1, nullptr, 1.f, s), "tr wk bwd x");
    w2lCheck(w2l_linear_backward_data(M, C, C, dv, wv.w(cx), dx, 1, nullptr, 1.f, s), "tr wv bwd x");
  }
};

}  // namespace

// ============================================================================ Sequential
std::string Sequential::prettyString() const {
  std::ostringstream os;
  os << "Sequential [";
  for (size_t i = 0; i < layers_.size(); ++i) os << (i ? " -> " : "") << "(" << i << ") " << layers_[i]->name();
  os << "]";
  return os.str();
}

void Sequential::finalize() {
  params_.clear();
  params_.reserve(layers_.size() * 10 + 16);  // P handles keep pointers to this vector: never reallocate later
  int stream = 1;
  std::vector<size_t> firstIdx;
  for (auto& l : layers_) {
    l->rngStream = stream;
    stream += 4;
    firstIdx.push_back(params_.size());
    l->registerParams(params_);
  }
  size_t off = 0;
  for (auto& p : params_) {
    p.offset = off;
    off += (p.numel + 3) / 4 * 4;  // 16-byte aligned slots (float4 optimizer)
  }
  paramFloats_ = off;
  layerLo_.clear();
  for (size_t i = 0; i < layers_.size(); ++i) layerLo_.push_back(firstIdx[i] < params_.size() ? params_[firstIdx[i]].offset : paramFloats_);
}

size_t Sequential::plan(int B, int T, int nFeat) {
  Planner pl;
  in_ = actInput(B, T, nFeat);
  inOff_ = pl.alloc(in_.numel());
  Act a = in_;
  acts_.clear();
  for (auto& l : layers_) {
    a = l->plan(a, pl);
    acts_.push_back(a);
  }
  out_ = a;
  // emissions must be (N, T', B, 1) == physical [B][T'][N]
  // (dimension 0 may be several feature factors -- a View over a convolution layout -- as long as they tile the frame contiguously,
  // fastest first: the flat feature index is then the logical one)
  bool ok = out_.d[0].size() == out_.F;
  {
    long run = 1;
    for (auto& f : out_.d[0].f) {
      ok = ok && f.kind == F_FEAT && f.stride == run;
      run *= f.size;
    }
  }
  std::vector<FKind> rest;
  for (int i = 1; i < 4; ++i)
    for (auto& f : out_.d[i].f) rest.push_back(f.kind);
  ok = ok && rest == std::vector<FKind>{F_TIME, F_BATCH};
  if (!ok) throw std::invalid_argument("network output must be (NLABEL, T, B, 1); got " + out_.str());
  ys_.assign(layers_.size(), nullptr);
  return pl.used();
}

const float* Sequential::forward(Ctx& c, float* arena, const float* xRef) {
  // reference input (T, NFEAT, 1, B) is time-fastest [B][NFEAT][T]; internal layout is frame-major
  float* x = arena + inOff_;
  w2lCheck(w2l_transpose(xRef, x, in_.B, in_.F, in_.T, c.stream), "input transpose");
  const float* cur = x;
  c.imgOf = nullptr;
  for (size_t i = 0; i < layers_.size(); ++i) {
    float* y = nullptr;
    const float* noteBefore = c.imgOf;
    layers_[i]->forward(c, arena, cur, y);
    // a note about bf16 images only ever describes the activation handed to the NEXT layer, and only its CURRENT values: a
    // layer that rewrites its input in place (Dropout in training, ReLU, SpecAugment) returns the same pointer with other
    // contents -- the images the producer wrote are stale then, and the next fl::Linear must convert again
    if (c.imgOf == noteBefore && !layers_[i]->passesInputThrough(c)) c.imgOf = nullptr;
    if (c.imgOf != y) c.imgOf = nullptr;
    ys_[i] = y;
    cur = y;
  }
  c.imgOf = nullptr;
  return cur;
}

void Sequential::backward(Ctx& c, float* arena, const float* dOut) {
  const float* dy = dOut;
  // first layer with parameters: nothing before it needs a data gradient
  size_t firstParam = 0;
  for (size_t i = 0; i < layers_.size(); ++i) {
    const std::string nm = layers_[i]->name();
    if (nm != "View" && nm != "Reorder" && nm != "SpecAugment" && nm != "Dropout" && nm != "ReLU") { firstParam = i; break; }
  }
  size_t nb = bOff_.size();  // buckets [nb, ...) already signalled
  for (size_t ii = layers_.size(); ii-- > 0;) {
    float* dx = nullptr;
    bool needDx = ii > firstParam;
    layers_[ii]->backward(c, arena, dy, dx, needDx);
    for (; nb > 0 && bOff_[nb - 1] >= layerLo_[ii]; --nb) hipCheck(hipEventRecord(bEv_[nb - 1], c.stream), "bucket event");
    if (!needDx) break;
    dy = dx;
  }
  for (; nb > 0; --nb) hipCheck(hipEventRecord(bEv_[nb - 1], c.stream), "bucket event");
}

// ---- parameter init / import / export ----------------------------------------
static inline uint64_t splitmix(uint64_t& s) {
  uint64_t z = (s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

void Sequential::importParam(size_t i, const float* ref, float* host) const {
  const ParamInfo& p = params_[i];
  float* dst = host + p.offset;
  if (p.kind == 1) {  // reference memory [cout][cin][kh][kw] (ArrayFire dims (kw, kh, cin, cout)) -> internal [kw][kh][cin][cout]
    for (int co = 0; co < p.cout; ++co)
      for (int ci = 0; ci < p.cin; ++ci)
        for (int dh = 0; dh < p.kh; ++dh)
          for (int k = 0; k < p.kw; ++k)
            dst[(((size_t)k * p.kh + dh) * p.cin + ci) * p.cout + co] = ref[(((size_t)co * p.cin + ci) * p.kh + dh) * p.kw + k];
  } else if (p.kind == 2 && !p.rowPerm.empty()) {  // reference memory [in][out]; internal row r = reference row rowPerm[r]
    for (int r = 0; r < p.cin; ++r) std::memcpy(dst + (size_t)r * p.cout, ref + (size_t)p.rowPerm[r] * p.cout, sizeof(float) * p.cout);
  } else if (p.kind == 5) {  // ArrayFire (rows, cols) column-major == memory [cols][rows] -> internal [rows][cols]
    for (int r = 0; r < p.cin; ++r)
      for (int c = 0; c < p.cout; ++c) dst[(size_t)r * p.cout + c] = ref[(size_t)c * p.cin + r];
  } else {
    std::memcpy(dst, ref, sizeof(float) * p.numel);
  }
}

void Sequential::exportParam(size_t i, const float* host, float* ref) const {
  const ParamInfo& p = params_[i];
  const float* src = host + p.offset;
  if (p.kind == 1) {
    for (int co = 0; co < p.cout; ++co)
      for (int ci = 0; ci < p.cin; ++ci)
        for (int dh = 0; dh < p.kh; ++dh)
          for (int k = 0; k < p.kw; ++k)
            ref[(((size_t)co * p.cin + ci) * p.kh + dh) * p.kw + k] = src[(((size_t)k * p.kh + dh) * p.cin + ci) * p.cout + co];
  } else if (p.kind == 2 && !p.rowPerm.empty()) {
    for (int r = 0; r < p.cin; ++r) std::memcpy(ref + (size_t)p.rowPerm[r] * p.cout, src + (size_t)r * p.cout, sizeof(float) * p.cout);
  } else if (p.kind == 5) {
    for (int r = 0; r < p.cin; ++r)
      for (int c = 0; c < p.cout; ++c) ref[(size_t)c * p.cin + r] = src[(size_t)r * p.cout + c];
  } else {
    std::memcpy(ref, src, sizeof(float) * p.numel);
  }
}

void Sequential::initParams(float* host, uint64_t seed) const {
  std::memset(host, 0, sizeof(float) * paramFloats_);
  uint64_t st = seed * 0x2545F4914F6CDD1Dull + 12345;
  for (size_t i = 0; i < params_.size(); ++i) {
    const ParamInfo& p = params_[i];
    float* dst = host + p.offset;
    if (p.kind == 3) { dst[0] = 1.f; dst[1] = 0.f; continue; }  // LayerNorm gamma, beta
    if (p.kind == 4) {  // WeightNorm g = ||v|| per output column of the preceding v [K][N]
      const ParamInfo& v = params_[i - 1];
      const float* vs = host + v.offset;
      size_t N = p.numel, K = v.numel / N;
      for (size_t n = 0; n < N; ++n) {
        double s = 0;
        for (size_t k = 0; k < K; ++k) s += (double)vs[k * N + n] * vs[k * N + n];
        dst[n] = (float)std::sqrt(s);
      }
      continue;
    }
    for (size_t e = 0; e < p.numel; ++e) {
      double u = (double)(splitmix(st) >> 11) * (1.0 / 9007199254740992.0);
      dst[e] = (float)((2.0 * u - 1.0) * p.initBound);
    }
  }
}

// ============================================================================ builder
static int toI(const std::string& s) { return std::stoi(s); }
static double toD(const std::string& s) { return std::stod(s); }

static std::shared_ptr<Layer> buildOne(const LayerSpec& s, const LayerSpec* wnParent) {
  const auto& a = s.args;
  auto unsupported = [&]() -> std::shared_ptr<Layer> {
    throw std::invalid_argument("layer '" + s.tok + "' is outside the hot path this build implements: " + s.line);
  };
  if (s.tok == "V") {
    auto l = std::make_shared<ViewLayer>();
    for (int i = 0; i < 4; ++i) l->dims[i] = std::stol(a[i]);
    return l;
  }
  if (s.tok == "RO") {
    auto l = std::make_shared<ReorderLayer>();
    for (int i = 0; i < 4; ++i) l->perm[i] = toI(a[i]);
    return l;
  }
  if (s.tok == "C" || s.tok == "C1") {
    auto l = std::make_shared<Conv2DLayer>();
    l->cin = toI(a[0]); l->cout = toI(a[1]); l->kw = toI(a[2]); l->stride = toI(a[3]);
    l->pad = a.size() >= 5 ? toI(a[4]) : 0;
    if (a.size() >= 6 && toI(a[5]) != 1) throw std::invalid_argument("dilation != 1 not supported: " + s.line);
    l->hasBias = a.size() >= 7 ? toI(a[6]) != 0 : true;
    if (a.size() >= 8 && toI(a[7]) != 1) throw std::invalid_argument("groups != 1 not supported: " + s.line);
    return l;
  }
  if (s.tok == "C2") {
    auto l = std::make_shared<Conv2DLayer>();
    l->cin = toI(a[0]); l->cout = toI(a[1]); l->kw = toI(a[2]);
    l->kh = toI(a[3]);
    if (l->kh < 1 || toI(a[5]) != 1) throw std::invalid_argument("stride on the H axis is not supported: " + s.line);
    l->stride = toI(a[4]);
    l->pad = a.size() >= 7 ? toI(a[6]) : 0;
    const int py = a.size() >= 8 ? toI(a[7]) : 0;
    if (l->kh == 1) {
      if (py != 0 && py != -1) throw std::invalid_argument("padding on the H axis not supported: " + s.line);
    } else {
      // kh > 1 (recipes/sota/2019/am_arch/am_tds_ctc_librivox.arch: "C2 1 16 21 3 2 1 -1 -1"): SAME on the mel axis only
      if (l->kh % 2 == 0 || (py != -1 && py != (l->kh - 1) / 2))
        throw std::invalid_argument("kh x kw convolutions keep the H axis (odd kh, padding -1 or (kh-1)/2): " + s.line);
      l->padH = (l->kh - 1) / 2;
    }
    if ((a.size() >= 9 && toI(a[8]) != 1) || (a.size() >= 10 && toI(a[9]) != 1)) throw std::invalid_argument("dilation not supported: " + s.line);
    return l;
  }
  if (s.tok == "L") {
    auto l = std::make_shared<LinearLayer>();
    l->in = toI(a[0]); l->out = toI(a[1]);
    l->hasBias = !(a.size() == 3 && a[2] == "0");
    return l;
  }
  if (s.tok == "TDS") {
    auto l = std::make_shared<TDSLayer>();
    l->c = toI(a[0]); l->kw = toI(a[1]); l->h = toI(a[2]);
    l->p = a.size() >= 4 ? toD(a[3]) : 0.0;
    l->l = l->c * l->h;
    l->l2 = a.size() >= 5 ? toI(a[4]) : 0;
    if (l->l2 == 0) l->l2 = l->l;
    l->rPad = a.size() >= 6 ? toI(a[5]) : -1;
    l->lnTime = !(a.size() >= 7 && toI(a[6]) == 0);
    return l;
  }
  if (s.tok == "M") {
    if (a.size() < 4) throw std::invalid_argument("Failed parsing - " + s.line);
    if (toI(a[1]) != 1 || toI(a[3]) != 1 || (a.size() > 4 && toI(a[4]) != 0) || (a.size() > 5 && toI(a[5]) != 0))
      throw std::invalid_argument("M: max pooling is supported over the time axis, unpadded: " + s.line);
    auto l = std::make_shared<PoolTimeLayer>();
    l->w = toI(a[0]); l->stride = toI(a[2]);
    if (l->w < 1 || l->stride < 1) throw std::invalid_argument("Failed parsing - " + s.line);
    return l;
  }
  if (s.tok == "TR") {
    if (a.size() < 5 || a.size() > 8) throw std::invalid_argument("Failed parsing - " + s.line);
    auto l = std::make_shared<TransformerLayer>();
    l->C = toI(a[0]); l->mlp = toI(a[1]); l->nH = toI(a[2]); l->csz = toI(a[3]); l->p = toD(a[4]);
    l->pLayerDrop = a.size() >= 6 ? toD(a[5]) : 0.0;
    if ((a.size() >= 7 && toI(a[6]) != 0) || (a.size() >= 8 && toI(a[7]) != 0))
      throw std::invalid_argument("TR: pre-LayerNorm / future-mask variants are outside this build: " + s.line);
    if (l->nH < 1 || l->C % l->nH) throw std::invalid_argument("TR: heads must divide the model size: " + s.line);
    return l;
  }
  if (s.tok == "LN") {
    auto l = std::make_shared<LayerNormLayer>();
    for (auto& x : a) l->axes.push_back(toI(x));
    return l;
  }
  if (s.tok == "R") return std::make_shared<ReLULayer>();
  if (s.tok == "DO") { auto l = std::make_shared<DropoutLayer>(); l->p = toD(a[0]); return l; }
  if (s.tok == "GLU") { auto l = std::make_shared<GLULayer>(); l->dim = toI(a[0]); return l; }
  if (s.tok == "SAUG") {
    auto l = std::make_shared<SpecAugmentLayer>();
    l->fMaskF = toI(a[1]); l->nFMask = toI(a[2]); l->tMaskT = toI(a[3]); l->tMaskP = (float)toD(a[4]); l->nTMask = toI(a[5]);
    return l;
  }
  if (s.tok == "WN") {
    int dim = toI(a[0]);
    auto child = buildOne(*s.child, &s);
    if (auto c = std::dynamic_pointer_cast<Conv2DLayer>(child)) {
      if (dim != 3) throw std::invalid_argument("WN on a convolution is supported for dim 3 (per output channel): " + s.line);
      c->wn.on = true;
      return c;
    }
    if (auto l = std::dynamic_pointer_cast<LinearLayer>(child)) {
      if (dim != 0) throw std::invalid_argument("WN on Linear is supported for dim 0 (per output row): " + s.line);
      l->wn.on = true;
      return l;
    }
    throw std::invalid_argument("WN must wrap C / C2 / L: " + s.line);
  }
  (void)wnParent;
  return unsupported();
}

std::shared_ptr<Sequential> buildSequentialFromText(const std::string& text, int64_t nFeat, int64_t nLabel) {
  auto specs = parseArch(text, nFeat, nLabel);
  auto net = std::make_shared<Sequential>();
  int padL = 0, padR = 0;
  bool pendingPad = false;
  std::shared_ptr<Layer> last;
  for (auto& s : specs) {
    if (s.tok == "PD") {
      // only zero padding of the time axis (dim 0), folded into the next convolution
      if (toD(s.args[0]) != 0.0) throw std::invalid_argument("PD: only zero padding is supported: " + s.line);
      for (size_t i = 3; i < s.args.size(); ++i)
        if (toI(s.args[i]) != 0) throw std::invalid_argument("PD: only the time axis can be padded: " + s.line);
      padL = toI(s.args[1]); padR = toI(s.args[2]);
      pendingPad = true;
      continue;
    }
    auto l = buildOne(s, nullptr);
    if (pendingPad) {
      auto c = std::dynamic_pointer_cast<Conv2DLayer>(l);
      if (!c || c->pad != 0) throw std::invalid_argument("PD must be followed by an unpadded convolution: " + s.line);
      c->extraPadL = padL; c->extraPadR = padR;
      pendingPad = false;
    }
    // peephole: Conv/Linear + ReLU -> fused epilogue
    if (s.tok == "R" && last) {
      if (auto c = std::dynamic_pointer_cast<Conv2DLayer>(last)) { if (!c->fuseRelu) { c->fuseRelu = true; continue; } }
      if (auto li = std::dynamic_pointer_cast<LinearLayer>(last)) { if (!li->fuseRelu) { li->fuseRelu = true; continue; } }
    }
    net->add(l);
    last = l;
  }
  if (pendingPad) throw std::invalid_argument("dangling PD at the end of the architecture");
  net->finalize();
  // dry plan: validates shapes / layouts and fixes the Linear row permutations
  net->plan(1, 4096, (int)nFeat);
  return net;
}

std::shared_ptr<Sequential> buildSequentialModule(const std::string& archfile, int64_t nFeat, int64_t nLabel) {
  return buildSequentialFromText(readFile(archfile), nFeat, nLabel);
}

}  // namespace w2l
