// Compiled C++ caller of the fl:: surface (include/fl_compat/flashlight.h): the round-1 verdict's "every test enters
// through ctypes" gap.  Built by __graft_entry__.build() with plain g++ against libw2l_hip.so, driven by
// tests/test_gpu_fl_compat.py, which writes the inputs, runs this binary and checks its outputs against the oracle.
//
//   fl_compat_test crit <asg|ctc> <scalemode> <in.bin> <out.bin>
//       in : int32 N T B L | float em[B][T][N] | int32 target[B][L] | float trans[N][N] | float gradWeights[B]
//       out: float loss[B] | float dEm[B][T][N] | float dTrans[N][N] (asg) | int32 path[B][T] | int32 fpath[B][T] (asg)
//   fl_compat_test net <plugin.so|file.arch> <nfeat> <nlabel> <in.bin> <out.bin>
//       in : int32 T B L | float x[B][nfeat][T] | int32 target[B][L]
//       out: float loss0[B] | float loss1[B] (after one SGD step) | float emission checksum | int32 nparams | float gradnorm
//   fl_compat_test mfsc <audio file> <nfilters> <out.bin>
//       fl::pkg::speech::loadSound + fl::lib::audio::Mfsc (include/fl_compat/audio.h) of one utterance
//       out: int32 T F rate nsamples | float feat[F][T]
// The step is the reference's (recipes/slimIPL/src/Train.cpp:1454-1804): forward, criterion forward, zeroGrad,
// loss.backward(), grads / batch, clipGradNorm, critopt->step(), netopt->step().
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

#include "fl_compat/flashlight.h"
#include "fl_compat/audio.h"

using namespace fl;
using namespace fl::pkg::speech;

static std::vector<char> slurp(const char* path) {
  FILE* f = fopen(path, "rb");
  if (!f) { perror(path); exit(2); }
  fseek(f, 0, SEEK_END);
  long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  std::vector<char> b((size_t)n);
  if (fread(b.data(), 1, (size_t)n, f) != (size_t)n) { perror("read"); exit(2); }
  fclose(f);
  return b;
}

static int runCrit(const std::string& kind, int mode, const char* in, const char* outp) {
  auto buf = slurp(in);
  const int* hd = (const int*)buf.data();
  const int N = hd[0], T = hd[1], B = hd[2], L = hd[3];
  const float* em = (const float*)(hd + 4);
  const int* tgt = (const int*)(em + (size_t)B * T * N);
  const float* trans = (const float*)(tgt + (size_t)B * L);
  const float* gw = trans + (size_t)N * N;

  std::shared_ptr<SequenceCriterion> crit;
  if (kind == "asg") {
    auto asg = std::make_shared<ASGLoss>(N, (CriterionScaleMode)mode, 4.0);
    asg->setParams(Variable(af::array(af::dim4(N, N), trans), true), 0);   // like Train.cpp's setParams on the transitions
    crit = asg;
  } else {
    crit = std::make_shared<CTCLoss>((CriterionScaleMode)mode);
  }
  std::cout << crit->prettyString() << std::endl;
  Variable emission(af::array(af::dim4(N, T, B), em), true);
  Variable target(af::array(af::dim4(L, B), tgt), false);
  auto loss = crit->forward({emission, target}).front();
  if (loss.dims(0) != B) { std::cerr << "loss dims\n"; return 1; }
  // the Trainer decodes BETWEEN the criterion's forward and loss.backward() on every report iteration (Train.cpp:1699-1716):
  // the decode must not disturb what backward reads (it once shared the criterion's workspace)
  std::vector<int> hp0((size_t)B * T);
  crit->viterbiPath(emission.array()).host(hp0.data());
  loss.backward(Variable(af::array(af::dim4(B), gw), false));

  FILE* f = fopen(outp, "wb");
  std::vector<float> h((size_t)B * T * N);
  std::vector<float> hl(B);
  loss.host(hl.data());
  fwrite(hl.data(), 4, B, f);
  emission.grad().host(h.data());
  fwrite(h.data(), 4, h.size(), f);
  if (kind == "asg") {
    std::vector<float> ht((size_t)N * N);
    crit->param(0).grad().host(ht.data());
    fwrite(ht.data(), 4, ht.size(), f);
  }
  std::vector<int> hp((size_t)B * T);
  crit->viterbiPath(emission.array()).host(hp.data());
  if (hp != hp0) { std::cerr << "viterbiPath before and after backward differ\n"; return 1; }
  fwrite(hp.data(), 4, hp.size(), f);
  if (kind == "asg") {
    crit->viterbiPathWithTarget(emission.array(), target.array()).host(hp.data());
    fwrite(hp.data(), 4, hp.size(), f);
  }
  fclose(f);
  // error behaviour mirrors Flashlight: std::invalid_argument on a shape / dtype mismatch
  try {
    crit->forward({Variable(af::array(af::dim4(N, T, B), tgt), true), target});
    std::cerr << "expected invalid_argument\n";
    return 1;
  } catch (const std::invalid_argument&) {}
  return 0;
}

static int runNet(const std::string& arch, int nfeat, int nlabel, const char* in, const char* outp) {
  auto buf = slurp(in);
  const int* hd = (const int*)buf.data();
  const int T = hd[0], B = hd[1], L = hd[2];
  const float* x = (const float*)(hd + 3);
  const int* tgt = (const int*)(x + (size_t)B * nfeat * T);

  // --arch names either an arch file or a plugin library (Train.cpp:390-395)
  std::shared_ptr<fl::Module> network = fl::pkg::runtime::ModulePlugin(arch).arch(nfeat, nlabel);
  std::cout << network->prettyString() << std::endl;
  auto criterion = std::make_shared<CTCLoss>(getCriterionScaleMode("target", true));
  auto netoptim = std::make_shared<SGDOptimizer>(network->params(), 0.05, 0.5, 0.0);
  auto critoptim = std::make_shared<SGDOptimizer>(criterion->params(), 0.0, 0.0, 0.0);
  network->train();  // the arch of the parity run is dropout-free, so the step is deterministic in train mode

  Variable input = fl::input(af::array(af::dim4(T, nfeat, 1, B), x));
  Variable target(af::array(af::dim4(L, B), tgt), false);
  std::vector<float> l0(B), l1(B);
  float checksum = 0, gnorm = 0;
  for (int it = 0; it < 2; ++it) {
    auto output = network->forward({input, fl::noGrad(af::constant(T, af::dim4(1, B)))}).front();
    auto loss = criterion->forward({output, target}).front();
    if (it == 0) {
      std::vector<float> e((size_t)output.elements());
      output.host(e.data());
      double s = 0;
      for (float v : e) s += v;
      checksum = (float)s;
      loss.host(l0.data());
    } else {
      loss.host(l1.data());
      break;
    }
    netoptim->zeroGrad();
    critoptim->zeroGrad();
    loss.backward();
    // grads / totalBatchSize  (Train.cpp:1743-1784)
    for (auto& p : network->params()) {
      if (!p.isGradAvailable()) continue;
      std::vector<float> g((size_t)p.elements());
      p.grad().host(g.data());
      for (auto& v : g) v /= (float)B;
      p.grad().array() = af::array(p.dims(), g.data());
    }
    gnorm = (float)clipGradNorm(network->params(), 1.0);
    critoptim->step();
    netoptim->step();
  }
  FILE* f = fopen(outp, "wb");
  fwrite(l0.data(), 4, B, f);
  fwrite(l1.data(), 4, B, f);
  fwrite(&checksum, 4, 1, f);
  int np = (int)network->params().size();
  fwrite(&np, 4, 1, f);
  fwrite(&gnorm, 4, 1, f);
  fclose(f);
  return 0;
}

static int runMfsc(const char* path, int nfilters, const char* outp) {
  Sound snd = loadSound(path);
  fl::lib::audio::FeatureParams fp;
  fp.samplingFreq = snd.rate;
  fp.numFilterbankChans = nfilters;
  fl::lib::audio::Mfsc mfsc(fp);
  const long S = mfsc.frameStride(), n = (long)snd.samples.size(), nP = (n + S - 1) / S * S;
  std::vector<float> padded((size_t)nP, 0.f);
  std::copy(snd.samples.begin(), snd.samples.end(), padded.begin());
  af::array feats = mfsc.apply(af::array(af::dim4(nP, 1), padded.data()));
  const int Tall = (int)feats.dims(0), T = mfsc.numFrames(n);
  std::vector<float> h((size_t)feats.elements());
  feats.host(h.data());
  FILE* f = fopen(outp, "wb");
  if (!f) { perror(outp); return 2; }
  const int hd[4] = {T, nfilters, snd.rate, (int)n};
  fwrite(hd, 4, 4, f);
  for (int k = 0; k < nfilters; ++k) fwrite(h.data() + (size_t)k * Tall, 4, (size_t)T, f);
  fclose(f);
  return 0;
}

int main(int argc, char** argv) {
  try {
    if (argc == 5 && std::string(argv[1]) == "mfsc") return runMfsc(argv[2], atoi(argv[3]), argv[4]);
    if (argc == 6 && std::string(argv[1]) == "crit") return runCrit(argv[2], atoi(argv[3]), argv[4], argv[5]);
    if (argc == 7 && std::string(argv[1]) == "net") return runNet(argv[2], atoi(argv[3]), atoi(argv[4]), argv[5], argv[6]);
  } catch (const std::exception& e) {
    std::cerr << "fl_compat_test: " << e.what() << std::endl;
    return 3;
  }
  std::cerr << "usage: fl_compat_test crit <asg|ctc> <mode> in out | net <arch|plugin.so> nfeat nlabel in out\n";
  return 2;
}
