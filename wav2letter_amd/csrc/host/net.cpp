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
  void forward(Ctx&, float*, const float* x, float*& y) override { y = const_cast<float*>(x); }
  void backward(Ctx&, float*, const float* dy, float*& dx, bool) override { dx = const_cast<float*>(dy); }
};

class ReorderLayer : public Layer {
 public:
  int perm[4];
  std::string name() const override { return "Reorder"; }
  Act plan(const Act& in, Planner&) override { return actReorder(in, perm); }
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
  void forward(Ctx& c, float*, const float* x, float*& y) override {
    y = const_cast<float*>(x);
    if (c.train)
      w2lCheck(w2l_specaugment_inplace(y, B, T, F, fMaskF, nFMask, tMaskT, tMaskP, nTMask, c.seed ^ 0x5a5a5a5au, c.stream),
               "specaugment");
  }
  void backward(Ctx&, float*, const float* dy, float*& dx, bool) override { dx = const_cast<float*>(dy); }
};

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
    w2lCheck(w2l_conv_forward(&d, x, wt, hasBias ? b.w(c) : nullptr, y, fuseRelu ? 1 : 0, c.stream), "conv fwd");
  }
  void backward(Ctx& c, float* arena, const float* dy, float*& dx, bool needDx) override {
    float* dym = const_cast<float*>(dy);
    if (fuseRelu) w2lCheck(w2l_mask_backward(dy, arena + yOff, dym, nOut, 1.f, c.stream), "conv relu bwd");
    float* dwt = wn.on ? arena + wn.dwOff : w.g(c);
    w2lCheck(w2l_conv_backward_filter(&d, xSaved, dym, dwt, hasBias ? b.g(c) : nullptr, c.stream), "conv bwd filter");
    const float* wt = wn.on ? arena + wn.wOff : w.w(c);
    if (needDx) {
      dx = arena + dxOff;
      if (kh > 1) {
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
    w2lCheck(w2l_linear_forward(M, in, out, x, wt, hasBias ? b.w(c) : nullptr, y, fuseRelu ? 1 : 0, c.stream), "linear fwd");
  }
  void backward(Ctx& c, float* arena, const float* dy, float*& dx, bool needDx) override {
    float* dym = const_cast<float*>(dy);
    if (fuseRelu) w2lCheck(w2l_mask_backward(dy, arena + yOff, dym, (size_t)M * out, 1.f, c.stream), "linear relu bwd");
    float* dwt = wn.on ? arena + wn.dwOff : w.g(c);
    const float* wt = wn.on ? arena + wn.wOff : w.w(c);
    w2lCheck(w2l_linear_backward_weight(M, in, out, xSaved, dym, dwt, c.stream), "linear bwd w");
    if (hasBias) w2lCheck(w2l_colsum(dym, b.g(c), (size_t)M, out, c.stream), "linear bwd b");
    if (needDx) {
      dx = arena + dxOff;
      w2lCheck(w2l_linear_backward_data(M, in, out, dym, wt, dx, 0, nullptr, 1.f, c.stream), "linear bwd x");
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
    return in;
  }
  void forward(Ctx& cx, float* ar, const float* x, float*& y) override {
    hipStream_t s = cx.stream;
    const double pd = cx.train ? p : 0.0;
    xSaved = x;
    float *a = ar + aOff, *r1 = ar + r1Off, *y1 = ar + y1Off, *u = ar + uOff, *v = ar + vOff, *out = ar + outOff;
    w2lCheck(w2l_conv_forward(&d, x, wc.w(cx), bc.w(cx), a, 1, s), "tds conv");
    // a <- dropout(relu(conv)) in place (kept: its sign pattern is the ReLU+dropout mask); r1 = a + x; y1 = LN(r1)
    w2lCheck(w2l_residual_layernorm_forward(groups, inner, a, x, r1, y1, gb1.w(cx), 1e-5f, pd, cx.seed, rngStream,
                                            (double*)(ar + st1Off), ar + mr1Off, s), "tds ln1");
    // lin1 + ReLU + dropout in one GEMM epilogue (same mask bits as a separate dropout pass over u)
    if (pd > 0) w2lCheck(w2l_linear_forward_dropout(M, l, l2, y1, w1.w(cx), b1.w(cx), u, 1, pd, cx.seed, rngStream + 1, s), "tds lin1+do");
    else w2lCheck(w2l_linear_forward(M, l, l2, y1, w1.w(cx), b1.w(cx), u, 1, s), "tds lin1");
    w2lCheck(w2l_linear_forward(M, l2, l, u, w2.w(cx), b2.w(cx), v, 0, s), "tds lin2");
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
    w2lCheck(w2l_layernorm_backward(groups, inner, v, dy, gb2.w(cx), ar + mr2Off, ds, gb2.g(cx), nullptr, nullptr, 1.f,
                                    (double*)(ar + st2Off), s), "tds ln2 bwd");
    const float* dv = ds;
    if (pd > 0) {
      // dy1 buffer doubles as scratch for the masked copy (one out-of-place pass)
      w2lCheck(w2l_dropout_copy(dy1, ds, n, pd, cx.seed, rngStream + 2, s), "tds do2 bwd");
      dv = dy1;
    }
    // lin2: dW2 = u^T dv, db2, du = (dv W2^T) masked by relu+dropout of u (u holds the dropped value)
    w2lCheck(w2l_linear_backward_weight(M, l2, l, u, dv, w2.g(cx), s), "tds lin2 bwd w");
    w2lCheck(w2l_colsum(dv, b2.g(cx), (size_t)M, l, s), "tds lin2 bwd b");
    w2lCheck(w2l_linear_backward_data(M, l2, l, dv, w2.w(cx), du, 0, u, sc, s), "tds lin2 bwd x");
    // lin1: dW1 = y1^T du, db1, dy1 = ds + du W1^T
    w2lCheck(w2l_linear_backward_weight(M, l, l2, y1, du, w1.g(cx), s), "tds lin1 bwd w");
    w2lCheck(w2l_colsum(du, b1.g(cx), (size_t)M, l2, s), "tds lin1 bwd b");
    // dy1 = ds + du W1^T: the residual join rides in the GEMM epilogue as a separate addend (no copy of ds into dy1)
    w2lCheck(w2l_linear_backward_data_add(M, l, l2, du, w1.w(cx), ds, dy1, s), "tds lin1 bwd x");
    // LN1 backward: dr1, and in the same pass da = dr1 masked by the ReLU+dropout pattern of a
    w2lCheck(w2l_layernorm_backward(groups, inner, ar + r1Off, dy1, gb1.w(cx), ar + mr1Off, dr1, gb1.g(cx), a, da, sc,
                                    (double*)(ar + st1Off), s), "tds ln1 bwd");
    w2lCheck(w2l_conv_backward_filter(&d, xSaved, da, wc.g(cx), bc.g(cx), s), "tds conv bwd filter");
    if (needDx) {
      dx = ar + dxOff;
      w2lCheck(w2l_conv_backward_data_add(&d, da, wc.w(cx), dr1, dx, s), "tds conv bwd data");
    }
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
  bool ok = out_.d[0].f.size() <= 1 && (out_.d[0].f.empty() || (out_.d[0].f[0].kind == F_FEAT && out_.d[0].f[0].stride == 1)) &&
            out_.d[0].size() == out_.F;
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
  for (size_t i = 0; i < layers_.size(); ++i) {
    float* y = nullptr;
    layers_[i]->forward(c, arena, cur, y);
    ys_[i] = y;
    cur = y;
  }
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
