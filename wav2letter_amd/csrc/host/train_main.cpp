// train_main.cpp -- `Train train --flagsfile=... [--k=v ...]`: the reference Trainer's command line
// (recipes/slimIPL/src/Train.cpp:110-179: `train [flags]` | `continue [directory] [flags]` | `fork [model] [flags]`)
// over the fl:: surface of include/fl_compat/flashlight.h.
//
// It reads the recipes' own train.cfg / *.arch files UNCHANGED (gflags `--flagsfile`, later flags win), builds
//   network   = fl::pkg::runtime::ModulePlugin(FLAGS_arch).arch(numFeatures, numClasses)   (Train.cpp:390-395)
//   criterion = CTCLoss(scalemode) | ASGLoss(numClasses, scalemode, FLAGS_transdiag)         (:406-410)
//   netoptim / critoptim = initOptimizer(--netoptim / --critoptim: sgd | adagrad | adadelta)                                (:577-582)
// and runs the hot loop of Train.cpp:1454-1804 (forward, criterion, zeroGrad, backward, grads / batch, clipGradNorm,
// critopt->step, netopt->step) with the reference's meters, printing the log line of MyLogger.cpp:40-106
// (`epoch | nupdates | lr | lrcriterion | runtime | bch(ms) | smp(ms) | fwd(ms) | crit-fwd(ms) | bwd(ms) | optim(ms) |
// loss | train-TER | train-WER | ...`) to stdout and to <rundir>/<runname>/001_log, next to 001_config (:644-651).
//
// `Train continue <directory>` / `Train fork <model>` (Train.cpp:124-179, :452-463): the run directory holds NNN_log, NNN_config and
// NNN_model_last.bin (getRunFile, :644-651, :767-790); `continue` looks for the highest NNN_model_last.bin, re-reads the flags
// stored in it (the command line overrides them), restores network, criterion, both optimizers, the update counter and the
// position of every random stream, and goes on as run NNN+1 -- bit for bit where an uninterrupted run would be; `fork` takes
// flags + network + criterion from a model file and starts a fresh run in --rundir.  The container is the documented
// W2LAMD01 layout of wav2letter_amd/checkpoint.py (fl::pkg::runtime::Serializer), not cereal.
//
// Data (Train.cpp:277-339): when --train names list files that exist (relative to --datadir, comma separated), the step consumes
// THEM -- `id path duration transcript` lines, audio decoded on the host (WAV / FLAC / raw PCM: fl_compat/audio.h), MFSC features
// and the per-utterance normalisation on the device, targets from --tokens / --lexicon through fl_compat/text.h (replabels for
// ASG), batches of --batchsize in list order, rank r taking its share of every global batch (partitionByRoundRobin); train-TER /
// train-WER of the log line come from the Viterbi path through tknPrediction2Ltr / tkn2Wrd (Train.cpp:829-872).  Without lists the
// data is SYNTHETIC: LibriSpeech-shaped padded batches (--w2l_synth_frames frames of --filterbanks features, random targets), which
// is exactly what bench.py times.  Not here (out of scope): the decoder, cereal checkpoints, validation sets.
// Flags of this driver that the reference does not have are prefixed w2l_.
// Data parallelism is the reference's: --enable_distributed --world_rank --world_size --max_devices_per_node
// --rndv_filepath (Train.cpp:188-199; RANK / WORLD_SIZE / LOCAL_WORLD_SIZE of a torchrun-style launcher are read when the
// flags are absent): one process per GPU, fl::CoalescingReducer over RCCL, batch size all-reduced with the gradients.
#include <limits>
#include <sys/stat.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <atomic>
#include <fstream>
#include <future>
#include <iostream>
#include <memory>
#include <random>
#include <sstream>
#include <thread>

#include "../../../include/fl_compat/flashlight.h"
#include "../../../include/fl_compat/audio.h"
#include "../../../include/fl_compat/data.h"
#include "../../../include/fl_compat/text.h"
#include "w2l_host.hpp"

using namespace fl;
using namespace fl::pkg::speech;

namespace {

struct Timer {
  double total = 0;
  int n = 0;
  std::chrono::steady_clock::time_point t0;
  void resume() { t0 = std::chrono::steady_clock::now(); }
  void stop() { total += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
  void stopAndIncUnit() { stop(); ++n; }
  double value() const { return n ? total / n : 0.0; }  // seconds per unit, like fl::TimeMeter(true)
  void reset() { total = 0; n = 0; }
};

std::string fmt(const char* f, double v) { char b[64]; snprintf(b, sizeof b, f, v); return b; }
std::string fmti(const char* f, long v) { char b[64]; snprintf(b, sizeof b, f, v); return b; }

int editDistance(const std::vector<int>& a, const std::vector<int>& b) {
  std::vector<int> prev(b.size() + 1), cur(b.size() + 1);
  for (size_t j = 0; j <= b.size(); ++j) prev[j] = (int)j;
  for (size_t i = 1; i <= a.size(); ++i) {
    cur[0] = (int)i;
    for (size_t j = 1; j <= b.size(); ++j)
      cur[j] = std::min({prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (a[i - 1] != b[j - 1])});
    std::swap(prev, cur);
  }
  return prev[b.size()];
}

bool fileExists(const std::string& p) { struct stat st; return stat(p.c_str(), &st) == 0; }
void mkdirs(const std::string& p) {
  std::string cur;
  for (size_t i = 0; i <= p.size(); ++i) {
    if (i == p.size() || p[i] == '/') { if (!cur.empty()) mkdir(cur.c_str(), 0755); }
    if (i < p.size()) cur += p[i];
  }
}
std::string pathJoin(const std::string& a, const std::string& b) {
  if (a.empty() || (!b.empty() && b[0] == '/')) return b;
  return a.back() == '/' ? a + b : a + "/" + b;
}

int countTokens(const std::string& path) {
  std::ifstream f(path);
  if (!f) return -1;
  int n = 0;
  std::string line;
  while (std::getline(f, line)) if (!line.empty()) ++n;
  return n;
}

// ---- list files -> padded device batches (Train.cpp:277-339)
struct ListData {
  std::vector<fl::pkg::speech::ListSample> samples;
  std::vector<long> mine;                       // this rank's sample indices, in list order
  fl::lib::text::Dictionary dict;
  fl::lib::text::LexiconMap lexicon;
  std::string wordsep, criterion;
  int replabel = 0, nFeat = 0, batch = 1, padFrames = 64, rate = 16000;
  std::unique_ptr<fl::lib::audio::Mfsc> mfsc;
  af::array unit;                               // LayerNorm (gamma, beta) = (1, 0): the per-utterance normalisation
  long batches() const { return ((long)mine.size() + batch - 1) / batch; }
  // the batch an update trains on: every epoch walks this rank's batches in a fresh order, the SAME order on every rank (the
  // global batch g stays the union of the ranks' batches g), a function of (--seed, epoch) only so that `continue` resumes on the
  // batch the uninterrupted run would have taken (the reference reshuffles per epoch with the epoch as seed:
  // loadPrefetchDataset(trainset, nthread, true /*shuffle*/, curEpoch), recipes/slimIPL/src/Train.cpp:1183)
  uint64_t shuffleSeed = 0;
  mutable long permEpoch = -1;
  mutable std::vector<long> perm;
  long batchOfUpdate(long update) const {   // update = 1, 2, ...
    const long nb = batches(), epoch = (update - 1) / nb, pos = (update - 1) % nb;
    if (epoch != permEpoch) {
      perm.resize((size_t)nb);
      for (long i = 0; i < nb; ++i) perm[(size_t)i] = i;
      std::mt19937_64 g(0x9E3779B97F4A7C15ull * (shuffleSeed + 1) + (uint64_t)epoch);
      for (long i = nb - 1; i > 0; --i) std::swap(perm[(size_t)i], perm[(size_t)(g() % (uint64_t)(i + 1))]);   // (std::shuffle is not portable across libraries)
      permEpoch = epoch;
    }
    return perm[(size_t)pos];
  }

  // host half of a batch: decoded audio + target rows.  Prepared AHEAD of the step by `nthread` decode threads (the reference's
  // --nthread prefetch workers, Train.cpp:331): a LibriSpeech batch is 32 FLAC files x ~10 ms each
  struct HostBatch {
    std::vector<std::vector<float>> audio;
    std::vector<std::vector<int>> rows;
    std::vector<float> sizes;
    std::string error;
  };
  int nthread = 6;
  std::future<std::shared_ptr<HostBatch>> pending;
  long pendingK = -1;

  std::shared_ptr<HostBatch> decode(long k) const {
    auto hb = std::make_shared<HostBatch>();
    const long lo = k * batch, hi = std::min<long>(lo + batch, (long)mine.size());
    const int B = (int)(hi - lo);
    hb->audio.resize((size_t)B); hb->rows.resize((size_t)B); hb->sizes.assign((size_t)B, 0.f);
    std::vector<std::string> errs((size_t)B);
    auto one = [&](int b) {
      try {
        const auto& smp = samples[(size_t)mine[(size_t)(lo + b)]];
        fl::pkg::speech::Sound snd = fl::pkg::speech::loadSound(smp.path);
        if (snd.rate != rate) throw std::runtime_error(smp.path + ": sample rate " + std::to_string(snd.rate) + ", --samplerate is " + std::to_string(rate));
        if ((long)snd.samples.size() < mfsc->frameSize()) throw std::runtime_error(smp.path + ": shorter than one analysis frame");
        hb->sizes[(size_t)b] = (float)snd.samples.size();
        hb->audio[(size_t)b] = std::move(snd.samples);
        hb->rows[(size_t)b] = fl::pkg::speech::targetIndices(smp.transcript, lexicon, dict, criterion, replabel, wordsep);
      } catch (const std::exception& e) { errs[(size_t)b] = e.what(); }
    };
    const int nt = std::max(1, std::min(nthread, B));
    std::vector<std::thread> pool;
    std::atomic<int> next{0};
    for (int t = 0; t < nt; ++t)
      pool.emplace_back([&]() { for (int b = next++; b < B; b = next++) one(b); });
    for (auto& th : pool) th.join();
    for (auto& e : errs) if (!e.empty()) { hb->error = e; break; }
    return hb;
  }

  // batch k -> features (T, NFEAT, 1, B) on the device, zero beyond every utterance's own frames; targets [B][L] (-1 padded);
  // sizes [B] in samples.  Returns B (the last batch of an epoch may be short).  The NEXT batch's files are decoded in the
  // background while the caller trains on this one.
  int get(long k, long kNext, af::array& input, std::vector<int>& tgt, int& L, std::vector<float>& sizes, int& T) {
    std::shared_ptr<HostBatch> hb;
    if (pending.valid() && pendingK == k) hb = pending.get();
    else { if (pending.valid()) pending.get(); hb = decode(k); }
    pendingK = kNext;
    pending = std::async(std::launch::async, [this]() { return decode(pendingK); });
    if (!hb->error.empty()) throw std::runtime_error(hb->error);
    const int B = (int)hb->audio.size();
    const int S = mfsc->frameStride();
    auto& audio = hb->audio;
    auto& rows = hb->rows;
    sizes = hb->sizes;   // (+ one entry below: the sample count the T padded frames stand for)
    long nsMax = 0;
    L = 1;
    for (int b = 0; b < B; ++b) {
      nsMax = std::max<long>(nsMax, (long)audio[(size_t)b].size());
      L = std::max<int>(L, (int)rows[(size_t)b].size());
    }
    // pad to whole strides and to a multiple of padFrames frames: few distinct (B, T) plans of the network
    int Tb = mfsc->numFrames(nsMax);
    Tb = (Tb + padFrames - 1) / padFrames * padFrames;
    const long ns = (long)(Tb - 1) * S + mfsc->frameSize();
    const long nsP = (ns + S - 1) / S * S;
    // T is rounded up beyond the longest utterance: the Transformer blocks' padding mask must divide by what T frames span, not by
    // the longest utterance (the reference pads to the longest only; cpc/SequentialBuilder.cpp:58-81)
    sizes.push_back((float)ns);
    std::vector<float> host((size_t)B * nsP, 0.f);
    for (int b = 0; b < B; ++b) memcpy(host.data() + (size_t)b * nsP, audio[(size_t)b].data(), audio[(size_t)b].size() * sizeof(float));
    af::array dev(af::dim4(nsP, B), host.data());
    af::array feats = mfsc->apply(dev);          // (Tall, NFEAT, 1, B), Tall >= Tb
    const int Tall = (int)feats.dims(0);
    T = Tb;
    input = af::constant(0.0, af::dim4(T, nFeat, 1, B));
    // per-utterance normalisation over the utterance's OWN frames (fl::lib::audio normalize(): zero mean, unit variance; applied
    // before padding in the reference): gather [NFEAT][T_b] -> LayerNorm with (1, 0) -> scatter into the zeroed batch
    hipStream_t st = (hipStream_t)fl::currentStream();
    const size_t cap = (size_t)nFeat * Tall;
    af::array tmp(af::dim4((af::dim_t)cap)), nrm(af::dim4((af::dim_t)cap)), mr(af::dim4(2));
    af::array stats(af::dim4((af::dim_t)(2 * w2l_layernorm_scratch_doubles(1, cap) + 2)));
    for (int b = 0; b < B; ++b) {
      const int tb = std::min(mfsc->numFrames((long)sizes[(size_t)b]), T);
      const float* src = feats.device<float>() + (size_t)b * nFeat * Tall;
      w2l::hipCheck(hipMemcpy2DAsync(tmp.device<float>(), (size_t)tb * 4, src, (size_t)Tall * 4, (size_t)tb * 4, (size_t)nFeat, hipMemcpyDeviceToDevice, st), "gather features");
      w2l::w2lCheck(w2l_residual_layernorm_forward(1, (size_t)nFeat * tb, tmp.device<float>(), nullptr, tmp.device<float>(), nrm.device<float>(),
                                                   unit.device<float>(), 1e-10f, 0.0, 0, 0, (double*)stats.device<float>(), mr.device<float>(), st), "normalise features");
      w2l::hipCheck(hipMemcpy2DAsync(input.device<float>() + (size_t)b * nFeat * T, (size_t)T * 4, nrm.device<float>(), (size_t)tb * 4, (size_t)tb * 4, (size_t)nFeat,
                                     hipMemcpyDeviceToDevice, st), "scatter features");
    }
    tgt.assign((size_t)B * L, -1);
    for (int b = 0; b < B; ++b) std::copy(rows[(size_t)b].begin(), rows[(size_t)b].end(), tgt.begin() + (size_t)b * L);
    af::sync();   // tmp / nrm / stats go out of scope
    return B;
  }
};

int usage(const char* exe) {
  std::cerr << "Usage: \n " << exe << " train [flags]\n or " << exe << " continue [directory] [flags]\n or " << exe
            << " fork [directory/model] [flags]" << std::endl;
  return 2;
}

}  // namespace

int main(int argc, char** argv) {
  if (argc <= 1) return usage(argv[0]);
  const std::string runStatus = argv[1];
  try {
    int first = 2;
    int runIdx = 1;                 // current #runs in this path (Train.cpp:124)
    std::string runPath, reloadPath;
    long startUpdate = 0;
    using Serializer = fl::pkg::runtime::Serializer;
    Serializer::Config reloadCfg;
    auto getRunFile = [](const std::string& name, int idx, const std::string& path) {
      char b[16];
      snprintf(b, sizeof b, "%03d_", idx);
      return pathJoin(path, std::string(b) + name);
    };
    w2l::Flags flags;
    if (runStatus == "continue") {
      if (argc <= 2) return usage(argv[0]);
      runPath = argv[2];
      first = 3;
      while (fileExists(getRunFile("model_last.bin", runIdx, runPath))) ++runIdx;
      if (runIdx == 1) throw std::invalid_argument("continue: no 001_model_last.bin in '" + runPath + "'");
      reloadPath = getRunFile("model_last.bin", runIdx - 1, runPath);
      std::cout << "reload path is " << reloadPath << std::endl;
    } else if (runStatus == "fork") {
      if (argc <= 2) return usage(argv[0]);
      reloadPath = argv[2];
      first = 3;
    } else if (runStatus != "train") {
      return usage(argv[0]);
    }
    if (!reloadPath.empty()) {
      std::string version;
      Serializer::load(reloadPath, version, reloadCfg);
      auto it = reloadCfg.find("gflags");
      if (it == reloadCfg.end()) throw std::invalid_argument("Invalid config loaded from " + reloadPath);
      std::cout << "Reading flags from config file " << reloadPath << std::endl;
      flags = w2l::parseFlagsText(it->second);
      if (runStatus == "continue") {
        auto up = reloadCfg.find("nbupdates");
        if (up == reloadCfg.end()) std::cout << "Did not find #updates to start from, starting from 0." << std::endl;
        else startUpdate = std::stol(up->second);
      }
    }
    // ---- flags: (the checkpoint's,) --flagsfile, then the command line in order (gflags semantics: the last definition wins)
    for (int i = first; i < argc; ++i) {
      w2l::Flags one = w2l::parseFlagsText(argv[i]);
      for (auto& kv : one.kv) flags.kv.push_back(kv);
    }
    const std::string criterionName = flags.get("criterion", "asg");    // the reference default (FLAGS_criterion)
    const int batch = (int)flags.geti("batchsize", 1);
    const int nFeat = flags.getb("mfcc", false) ? (int)flags.geti("mfcccoeffs", 13) * 3
                      : flags.getb("pow", false) ? (int)flags.geti("framesizems", 25) * 8 + 1 : (int)flags.geti("filterbanks", 40);
    const double lr0 = flags.getd("lr", 1.0), lrcrit0 = flags.getd("lrcrit", 0.0), momentum = flags.getd("momentum", 0.0);
    const double maxgradnorm = flags.getd("maxgradnorm", 0.0);
    const long iters = flags.geti("w2l_synth_updates", flags.geti("iter", 8));
    const long reportiters = flags.geti("reportiters", 0);
    const int T = (int)flags.geti("w2l_synth_frames", 1500);
    const int Lmax = (int)flags.geti("w2l_synth_target_len", criterionName == "ctc" ? 80 : 300);
    const long linseg = flags.geti("linseg", 0);
    const long warmup = flags.geti("warmup", 1);
    const uint64_t seed = (uint64_t)flags.geti("seed", 0);

    // ---- number of classes: the token dictionary (+ replabels for ASG, + blank for CTC: Train.cpp:230-251)
    int numClasses = (int)flags.geti("w2l_nlabel", 0);
    if (!numClasses) {
      const std::string tok = pathJoin(flags.get("tokensdir", ""), flags.get("tokens", "tokens.txt"));
      const int n = countTokens(tok);
      if (n <= 0) throw std::invalid_argument("cannot read the token dictionary '" + tok + "' (--tokensdir / --tokens); pass --w2l_nlabel=N for a synthetic run");
      numClasses = n;
      if (criterionName == "asg") numClasses += (int)flags.geti("replabel", 0);
      if (criterionName == "ctc") numClasses += 1;  // blank, appended LAST
    }

    // ---- run directory: NNN_log, NNN_config, NNN_model_last.bin (Train.cpp:644-651, :767)
    if (runStatus != "continue") runPath = pathJoin(flags.get("rundir", ""), flags.get("runname", ""));
    const bool haveRunDir = runStatus == "continue" || (!flags.get("rundir", "").empty() && flags.get("rundir", "") != "[...]");
    std::string gflagsText;   // what `continue` / `fork` read back (the reference serialises its gflags the same way)
    {
      // later definitions win, so the effective value of every flag once, in first-appearance order
      std::vector<std::string> order;
      for (auto& kv : flags.kv) if (std::find(order.begin(), order.end(), kv.first) == order.end()) order.push_back(kv.first);
      for (auto& k : order) if (k != "flagsfile") gflagsText += "--" + k + "=" + flags.get(k) + "\n";
    }
    std::ofstream logFile;

    // ---- network / criterion / optimizers
    const std::string archPath = pathJoin(flags.get("archdir", ""), flags.get("arch", ""));
    if (!fileExists(archPath)) throw std::invalid_argument("arch file / plugin '" + archPath + "' not found (--archdir / --arch)");
    auto scalemode = getCriterionScaleMode(flags.get("onorm", "none"), flags.getb("sqnorm", false));
    std::cout << "Loading architecture file from " << archPath << std::endl;
    std::shared_ptr<fl::Module> network = fl::pkg::runtime::ModulePlugin(archPath).arch(nFeat, numClasses);
    if (flags.getb("fl_amp_use_mixed_precision", false)) {
      setMixedPrecision(network, true);   // bf16 multiplies in the fl::Linear GEMMs, fp32 master weights / criterion
      std::cout << "Mixed precision training enabled (bf16 matrix multiplies, fp32 accumulation and storage)" << std::endl;
    }
    std::shared_ptr<SequenceCriterion> criterion;
    if (criterionName == "ctc") criterion = std::make_shared<CTCLoss>(scalemode);
    else if (criterionName == "asg") criterion = std::make_shared<ASGLoss>(numClasses, scalemode, flags.getd("transdiag", 0.0));
    else throw std::invalid_argument("unsupported criterion '" + criterionName + "' (this build: ctc, asg)");
    if (linseg > 0 && criterionName != "asg") throw std::invalid_argument("linseg may only be used with ASG criterion");  // Train.cpp:593
    size_t nparams = 0;
    for (auto& p : network->params()) nparams += (size_t)p.elements();
    std::cout << "[Network] " << network->prettyString() << std::endl;
    std::cout << "[Network Params: " << nparams << "]" << std::endl;
    std::cout << "[Criterion] " << criterion->prettyString() << std::endl;
    // initOptimizer(nets, --netoptim, lr, momentum, weightdecay) (Train.cpp:577-582): sgd, adadelta (librispeech/train_am_transformer_ctc.cfg:23-24)
    // and adagrad (librivox/train_am_transformer_ctc.cfg:25-26) are the ones the BASELINE recipes name
    const double optimrho = flags.getd("optimrho", 0.9), optimepsilon = flags.getd("optimepsilon", 1e-8);
    auto initOptimizer = [&](const std::vector<fl::Variable>& params, const std::string& kind, double lr, double mom) -> std::shared_ptr<fl::FirstOrderOptimizer> {
      if (kind == "sgd") return std::make_shared<SGDOptimizer>(params, lr, mom, 0.0);
      if (kind == "adagrad") return std::make_shared<fl::AdagradOptimizer>(params, lr);
      if (kind == "adadelta") return std::make_shared<fl::AdadeltaOptimizer>(params, lr, optimrho, optimepsilon);
      throw std::invalid_argument("unsupported optimizer '" + kind + "' (this build: sgd, adagrad, adadelta)");
    };
    auto netoptim = initOptimizer(network->params(), flags.get("netoptim", "sgd"), lr0, momentum);
    auto critoptim = initOptimizer(criterion->params(), flags.get("critoptim", "sgd"), lrcrit0, 0.0);
    std::cout << "[Network Optimizer] " << netoptim->prettyString() << std::endl;
    std::cout << "[Criterion Optimizer] " << critoptim->prettyString() << std::endl;
    if (runStatus == "fork") {            // Train.cpp:452-459: network + criterion, fresh optimizers
      std::string version;
      Serializer::Config unused;
      Serializer::load(reloadPath, version, unused, network, criterion);
      fl::pkg::speech::setNetworkStep(network, 0);
      std::cout << "Loaded model " << reloadPath << " for fork" << std::endl;
    } else if (runStatus == "continue") { // Train.cpp:460-467
      std::string version;
      Serializer::Config unused;
      Serializer::load(reloadPath, version, unused, network, criterion, netoptim, critoptim);
      std::cout << "Loaded model for continue training" << std::endl;
    }

    // ---- data parallelism (Train.cpp:188-199, :1078-1079)
    std::shared_ptr<fl::Reducer> reducer;
    if (flags.getb("enable_distributed", false)) {
      auto envi = [](const char* k, long dflt) { const char* v = getenv(k); return v ? atol(v) : dflt; };
      const int worldRank = (int)flags.geti("world_rank", envi("RANK", 0));
      const int worldSize = (int)flags.geti("world_size", envi("WORLD_SIZE", 1));
      fl::pkg::runtime::initDistributed(worldRank, worldSize, (int)flags.geti("max_devices_per_node", envi("LOCAL_WORLD_SIZE", 8)),
                                        flags.get("rndv_filepath", ""));
      reducer = std::make_shared<fl::CoalescingReducer>(1.0, true, true);
      fl::allReduceParameters(network);     // replicas start identical
      fl::allReduceParameters(criterion);
      std::cout << "[Distributed] world rank " << fl::getWorldRank() << " of " << fl::getWorldSize()
                << (flags.get("rndv_filepath", "").rfind("shm:", 0) == 0 ? " (host-memory test collective)" : " (RCCL)") << std::endl;
    }
    const bool isMaster = fl::getWorldRank() == 0;
    if (haveRunDir && isMaster) {
      mkdirs(runPath);
      logFile.open(getRunFile("log", runIdx, runPath));
      if (!logFile) throw std::runtime_error("failed to open log file for writing");
      std::ofstream cfg(getRunFile("config", runIdx, runPath));
      cfg << gflagsText;
    }

    // ---- data
    std::string trainLists = flags.get("train", "");
    const std::string dataDir = flags.get("datadir", "");
    std::vector<std::string> listPaths;
    {
      std::istringstream ls(trainLists);
      for (std::string one; std::getline(ls, one, ',');) if (!one.empty()) listPaths.push_back(pathJoin(dataDir, one));
    }
    const bool haveLists = !listPaths.empty() && trainLists.find("[DATA_DST]") == std::string::npos && fileExists(listPaths[0]);
    ListData data;
    if (haveLists) {
      for (auto& lp : listPaths) {
        std::ifstream lf(lp);
        if (!lf) throw std::invalid_argument("cannot read the list file '" + lp + "' (--train / --datadir)");
        std::stringstream buf;
        buf << lf.rdbuf();
        for (auto& smp : fl::pkg::speech::parseList(buf.str())) {
          data.samples.push_back(smp);
          auto& pth = data.samples.back().path;   // the recipes' lists hold absolute paths; a relative one is taken from --datadir
          if (!pth.empty() && pth[0] != '/' && !fileExists(pth)) pth = pathJoin(dataDir, pth);
        }
      }
      if (data.samples.empty()) throw std::invalid_argument("the --train lists hold no samples");
      data.criterion = criterionName;
      data.replabel = criterionName == "asg" ? (int)flags.geti("replabel", 0) : 0;
      data.wordsep = flags.get("wordseparator", "|");
      data.dict = fl::pkg::speech::createTokenDict(fl::lib::text::Dictionary(pathJoin(flags.get("tokensdir", ""), flags.get("tokens", "tokens.txt"))),
                                                   criterionName, data.replabel);
      if ((int)data.dict.indexSize() != numClasses) throw std::invalid_argument("token dictionary size != number of classes");
      const std::string lexPath = flags.get("lexicon", "");
      if (!lexPath.empty() && fileExists(lexPath)) data.lexicon = fl::lib::text::loadWords(lexPath, (int)flags.geti("maxword", -1));
      data.nFeat = nFeat;
      data.batch = batch;
      data.shuffleSeed = seed;
      data.rate = (int)flags.geti("samplerate", 16000);
      data.padFrames = (int)flags.geti("w2l_pad_frames", 64);
      data.nthread = (int)flags.geti("nthread", 6);
      fl::lib::audio::FeatureParams fp;
      fp.samplingFreq = data.rate; fp.frameSizeMs = (int)flags.geti("framesizems", 25); fp.frameStrideMs = (int)flags.geti("framestridems", 10);
      fp.numFilterbankChans = nFeat; fp.preemCoef = (float)flags.getd("preemcoef", 0.97); fp.melFloor = (float)flags.getd("melfloor", 1.0);
      if (flags.getb("mfcc", false) || flags.getb("pow", false)) throw std::invalid_argument("list data: only --mfsc features are built (--mfcc / --pow are not)");
      data.mfsc.reset(new fl::lib::audio::Mfsc(fp));
      const float unit[2] = {1.f, 0.f};
      data.unit = af::array(af::dim4(2), unit);
      for (long i : fl::lib::partitionByRoundRobin((long)data.samples.size(), fl::getWorldRank(), fl::getWorldSize(), batch)) data.mine.push_back(i);
      if (data.mine.empty()) throw std::invalid_argument("this rank has no samples (fewer samples than world_size * batchsize)");
      std::cout << "[Data] " << data.samples.size() << " samples in " << listPaths.size() << " list(s), " << data.mine.size() << " on this rank, "
                << data.batches() << " batches of " << batch << " per epoch; " << nFeat << " MFSC features; " << data.dict.indexSize() << " classes" << std::endl;
    }
    std::mt19937_64 rng(2026 + seed + 7919ull * (uint64_t)fl::getWorldRank());   // every rank draws its own shard of the (synthetic) minibatch
    std::normal_distribution<float> gauss(0.f, 1.f);
    // --w2l_synth_emulate_world=W (one process): the batch is the concatenation of the shards W ranks of --batchsize / W would
    // draw -- the reference run a data-parallel run of W ranks must reproduce (tests)
    const int emuWorld = (int)flags.geti("w2l_synth_emulate_world", 1);
    if (emuWorld < 1 || batch % emuWorld) throw std::invalid_argument("--w2l_synth_emulate_world must divide --batchsize");
    std::vector<std::mt19937_64> emuRng;
    std::vector<std::normal_distribution<float>> emuGauss((size_t)emuWorld, std::normal_distribution<float>(0.f, 1.f));
    for (int r = 0; r < emuWorld; ++r) emuRng.emplace_back(2026 + seed + 7919ull * (uint64_t)r);
    const int nTok = criterionName == "ctc" ? numClasses - 1 : std::max(1, numClasses - (int)flags.geti("replabel", 0));
    std::vector<float> hx((size_t)batch * nFeat * T);
    std::vector<int> ht((size_t)batch * Lmax);
    const long synthPool = flags.geti("w2l_synth_pool", 0);
    std::vector<float> hxPool;
    std::vector<int> htPool;
    if (runStatus == "continue") {   // the sample stream goes on where the saved run stopped (per rank)
      auto it = reloadCfg.find("w2l_data_rng." + std::to_string(fl::getWorldRank()));
      if (it != reloadCfg.end()) { std::istringstream is(it->second); is >> rng >> gauss; }
    }

    // ---- meters (MyLogger.cpp:40-106)
    Timer runtime, timer, sampletimer, fwdtimer, critfwdtimer, bwdtimer, optimtimer;
    double lossSum = 0;
    long lossN = 0, editErr = 0, editLen = 0, tszTotal = 0, tszMax = 0, nsamples = 0, nbatches = 0;
    fl::EditDistanceMeter wordMeter;   // list data: train-WER over words
    long framesTotal = 0;              // list data: padded input frames consumed (avg-isz, hrs)
    runtime.resume();
    network->train();
    criterion->train();
    const bool clampCrit = true;

    auto logStatus = [&](long epoch, long nupdates, double lr, double lrcrit) {
      runtime.stop();
      std::ostringstream s;
      auto item = [&](const std::string& k, const std::string& v) { s << (s.tellp() > 0 ? " | " : "") << k << ": " << v; };
      const int rt = (int)runtime.total;
      char rtb[32];
      snprintf(rtb, sizeof rtb, "%02d:%02d:%02d", rt / 3600, (rt / 60) % 60, rt % 60);
      item("epoch", fmti("%8ld", epoch));
      item("nupdates", fmti("%12ld", nupdates));
      item("lr", fmt("%4.6lf", lr));
      item("lrcriterion", fmt("%4.6lf", lrcrit));
      item("runtime", rtb);
      item("bch(ms)", fmt("%.2f", timer.value() * 1000));
      item("smp(ms)", fmt("%.2f", sampletimer.value() * 1000));
      item("fwd(ms)", fmt("%.2f", fwdtimer.value() * 1000));
      item("crit-fwd(ms)", fmt("%.2f", critfwdtimer.value() * 1000));
      item("bwd(ms)", fmt("%.2f", bwdtimer.value() * 1000));
      item("optim(ms)", fmt("%.2f", optimtimer.value() * 1000));
      item("loss", fmt("%10.5f", lossN ? lossSum / lossN : 0.0));
      const double ter = editLen ? 100.0 * editErr / editLen : 0.0;
      item("train-TER", fmt("%5.2f", ter));
      item("train-WER", fmt("%5.2f", haveLists ? wordMeter.value() : ter));  // synthetic targets: every token is its own word
      const double framesPerSample = haveLists && nsamples ? (double)framesTotal / nsamples : T;
      item("avg-isz", fmti("%03ld", (long)framesPerSample));
      item("avg-tsz", fmti("%03ld", nsamples ? tszTotal / nsamples : 0));
      item("max-tsz", fmti("%03ld", tszMax));
      item("avr-batchsz", fmt("%7.2f", nbatches ? (double)nsamples / nbatches : 0.0));
      const double audioSec = nsamples * framesPerSample * flags.getd("framestridems", 10) / 1000.0;
      item("hrs", fmt("%7.2f", audioSec / 3600.0));
      const double timeTaken = timer.value() * nbatches;
      item("thrpt(sec/sec)", timeTaken > 0 ? fmt("%.2f", audioSec / timeTaken) : std::string("n/a"));
      std::time_t now = std::time(nullptr);
      char ts[64];
      std::strftime(ts, sizeof ts, "%Y-%m-%d %H:%M:%S", std::localtime(&now));
      item("timestamp", ts);
      std::cout << s.str() << std::endl;
      if (logFile.is_open()) logFile << s.str() << std::endl;
      runtime.resume();
    };

    const long batchesPerEpoch = haveLists ? data.batches() : std::max<long>(1, flags.geti("w2l_synth_batches_per_epoch", iters));
    const long lrDecay = flags.geti("lr_decay", std::numeric_limits<int>::max());
    const long lrDecayStep = std::max<long>(1, flags.geti("lr_decay_step", std::numeric_limits<int>::max()));
    // --saug_start_update (Train.cpp:1026-1048): SpecAugment on the features from that update on (archs without a SAUG line)
    const long saugStart = flags.geti("saug_start_update", -1);
    std::shared_ptr<fl::SpecAugment> saug;
    if (saugStart >= 0)
      saug = std::make_shared<fl::SpecAugment>((int)flags.geti("filterbanks", 40), (int)flags.geti("saug_fmaskf", 27), (int)flags.geti("saug_fmaskn", 2),
                                               (int)flags.geti("saug_tmaskt", 100), (float)flags.getd("saug_tmaskp", 1.0), (int)flags.geti("saug_tmaskn", 2));
    if (saug) std::cout << "[SpecAugment from update " << saugStart << "] " << saug->prettyString() << std::endl;
    if (saug && runStatus == "continue") {
      auto it = reloadCfg.find("w2l_saug_calls");
      if (it != reloadCfg.end()) saug->setCalls((uint32_t)std::stoul(it->second));
    }

    // ---- checkpoints (saveModels, Train.cpp:718-790): NNN_model_last.bin after every epoch of the run and at its end
    auto saveModels = [&](long epoch, long totalUpdates) {
      if (!haveRunDir) return;
      // every rank's sample-stream position travels in rank 0's file: the others hand theirs over through the run directory
      std::ostringstream rs;
      rs << rng << " " << gauss;
      const std::string mine = getRunFile("rng." + std::to_string(fl::getWorldRank()), runIdx, runPath);
      if (!isMaster) { std::ofstream f(mine); f << rs.str(); }
      if (fl::getWorldSize() > 1) fl::barrier();
      if (!isMaster) {
        if (flags.getb("w2l_save_all_ranks", false)) {   // tests: the replica of a rank other than 0 (the replicas must stay in step)
          mkdirs(runPath);
          Serializer::Config rc;
          rc["nbupdates"] = std::to_string(totalUpdates);
          Serializer::save(getRunFile("model_last.bin.rank" + std::to_string(fl::getWorldRank()), runIdx, runPath), "0.1", rc, network, criterion,
                           netoptim, critoptim);
        }
        return;
      }
      Serializer::Config config;
      config["gflags"] = gflagsText;
      config["epoch"] = std::to_string(epoch);
      config["nbupdates"] = std::to_string(totalUpdates);
      config["runIdx"] = std::to_string(runIdx);
      config["w2l_data_rng.0"] = rs.str();
      for (int r = 1; r < fl::getWorldSize(); ++r) {
        std::ifstream f(getRunFile("rng." + std::to_string(r), runIdx, runPath));
        std::stringstream b;
        b << f.rdbuf();
        config["w2l_data_rng." + std::to_string(r)] = b.str();
      }
      if (saug) config["w2l_saug_calls"] = std::to_string(saug->calls());
      Serializer::save(getRunFile("model_last.bin", runIdx, runPath), "0.1", config, network, criterion, netoptim, critoptim);
    };

    // ---- the hot loop (Train.cpp:1454-1804)
    double lr = lr0, lrcrit = lrcrit0;
    for (long curBatch = startUpdate + 1; curBatch <= iters; ++curBatch) {
      // learning rate (Train.cpp:1170-1175, :1334-1348): 0.5^(epoch steps after --lr_decay) * (cosine | gamma^(batch / stepsize)) * warm-up;
      // an epoch of the synthetic run is --w2l_synth_batches_per_epoch updates (default: the whole run is epoch 1)
      const long curEpoch = 1 + (curBatch - 1) / batchesPerEpoch;
      const long afterDecay = curEpoch - lrDecay;
      const double lrDecayScale = std::pow(0.5, afterDecay < 0 ? 0.0 : (double)(1 + afterDecay / lrDecayStep));
      const double lrScheduleScale = flags.getb("lrcosine", false)
                                         ? std::cos((double)curBatch / (double)iters * std::acos(-1.0) / 2.0)
                                         : std::pow(flags.getd("gamma", 1.0), (double)curBatch / flags.getd("stepsize", 1e18));
      const double sched = lrDecayScale * lrScheduleScale * std::min((double)curBatch / std::max<long>(1, warmup), 1.0);
      lr = lr0 * sched;
      lrcrit = lrcrit0 * sched;
      netoptim->setLr(lr);
      critoptim->setLr(lrcrit);

      timer.resume();
      sampletimer.resume();
      int curB = batch, curT = T, curL = Lmax;   // this batch's shape (list data: per batch)
      fl::Variable input;
      af::array inputSizes;
      if (haveLists) {
        af::array feats;
        std::vector<float> sizes;
        curB = data.get(data.batchOfUpdate(curBatch), data.batchOfUpdate(curBatch + 1), feats, ht, curL, sizes, curT);
        for (int b = 0; b < curB; ++b) {
          long len = 0;
          while (len < curL && ht[(size_t)b * curL + len] >= 0) ++len;
          tszTotal += len;
          tszMax = std::max<long>(tszMax, len);
        }
        input = fl::input(feats);
        inputSizes = af::array(af::dim4(1, (af::dim_t)sizes.size()), sizes.data());   // curB sizes + the padded length
        if (curBatch <= 2 && !flags.get("w2l_dump_features", "").empty()) {   // debugging aid: [B][NFEAT][T] float32 of the first two batches
          std::vector<float> hf((size_t)feats.elements());
          feats.host(hf.data());
          std::ofstream df(flags.get("w2l_dump_features") + "." + std::to_string(curBatch), std::ios::binary);
          const int hd[3] = {curB, nFeat, curT};
          df.write((const char*)hd, sizeof hd);
          df.write((const char*)hf.data(), (std::streamsize)(hf.size() * 4));
        }
      } else {
      // --w2l_synth_pool=K: K batches are drawn once (before the first timed update) and update u trains on batch (u - 1) % K --
      // what a prefetching loader hides in a real run; the default draws every batch inside smp(ms) (3.8 M normal deviates)
      auto draw = [&](float* hxd, int* htd) {
        const int shard = batch / emuWorld;
        for (int r = 0; r < emuWorld; ++r) {   // (emuWorld = 1: this rank's own stream)
          auto& rg = emuWorld > 1 ? emuRng[(size_t)r] : rng;
          auto& gs = emuWorld > 1 ? emuGauss[(size_t)r] : gauss;
          float* hxr = hxd + (size_t)r * shard * nFeat * T;
          for (size_t k = 0; k < (size_t)shard * nFeat * T; ++k) hxr[k] = gs(rg);
          for (int bb = 0; bb < shard; ++bb) {
            const int b = r * shard + bb;
            const int lo = criterionName == "ctc" ? 20 : 60;
            const int len = lo + (int)(rg() % (uint64_t)std::max(1, Lmax - lo + 1));
            int prev = -1;
            for (int i = 0; i < Lmax; ++i) {
              int y = -1;
              if (i < len) {
                y = (int)(rg() % (uint64_t)nTok);
                if (criterionName == "asg" && y == prev) y = (y + 1) % nTok;  // replabel convention: no identical neighbours
                prev = y;
              }
              htd[(size_t)b * Lmax + i] = y;
            }
          }
        }
      };
      if (synthPool > 0) {
        if (hxPool.empty()) {
          sampletimer.stop(); timer.stop();
          hxPool.resize((size_t)synthPool * hx.size());
          htPool.resize((size_t)synthPool * ht.size());
          for (long k = 0; k < synthPool; ++k) draw(hxPool.data() + (size_t)k * hx.size(), htPool.data() + (size_t)k * ht.size());
          timer.resume(); sampletimer.resume();
        }
        const size_t slot = (size_t)((curBatch - 1) % synthPool);
        std::memcpy(hx.data(), hxPool.data() + slot * hx.size(), hx.size() * sizeof(float));
        std::memcpy(ht.data(), htPool.data() + slot * ht.size(), ht.size() * sizeof(int));
      } else {
        draw(hx.data(), ht.data());
      }
      for (int b = 0; b < batch; ++b) {
        long len = 0;
        while (len < Lmax && ht[(size_t)b * Lmax + len] >= 0) ++len;
        tszTotal += len;
        tszMax = std::max<long>(tszMax, len);
      }
      input = fl::input(af::array(af::dim4(T, nFeat, 1, batch), hx.data()));
      inputSizes = af::constant(T, af::dim4(1, batch));
      }
      if (saug && curBatch >= saugStart) input = saug->forward({input}).front();   // Train.cpp:1453-1461
      fl::Variable target(af::array(af::dim4(curL, curB), ht.data()), false);
      af::sync();
      sampletimer.stopAndIncUnit();

      // forward
      fwdtimer.resume();
      auto output = network->forward({input, fl::noGrad(inputSizes)}).front();
      af::sync();
      critfwdtimer.resume();
      auto loss = criterion->forward({output, target}).front();
      af::sync();
      fwdtimer.stopAndIncUnit();
      critfwdtimer.stopAndIncUnit();
      std::vector<float> hl((size_t)curB);
      loss.host(hl.data());
      for (float v : hl) {
        if (!std::isfinite(v)) {   // LOG(FATAL), Train.cpp:1686-1698
          std::ostringstream m;
          m << "Loss has NaN values (update " << curBatch << ", per-utterance losses:";
          for (float q : hl) m << " " << q;
          std::vector<float> he((size_t)output.elements());
          output.array().host(he.data());
          float mx = 0.f;
          size_t bad = 0;
          for (float q : he) { if (std::isfinite(q)) mx = std::max(mx, std::fabs(q)); else ++bad; }
          m << "; emissions: max |x| " << mx << ", " << bad << " non-finite of " << he.size() << ")";
          throw std::runtime_error(m.str());
        }
        lossSum += v;
        ++lossN;
      }
      if (reportiters > 0 && curBatch % reportiters == 0) {  // token error of the Viterbi path (evalOutput, Train.cpp:1699-1716)
        std::vector<int> path((size_t)curB * output.dims(1));
        criterion->viterbiPath(output.array()).host(path.data());
        const int To = (int)output.dims(1);
        for (int b = 0; b < curB; ++b) {
          std::vector<int> hyp, ref;
          int prev = -1;
          for (int t = 0; t < To; ++t) {
            const int y = path[(size_t)b * To + t];
            if (y != prev && !(criterionName == "ctc" && y == numClasses - 1)) hyp.push_back(y);
            prev = y;
          }
          for (int i = 0; i < curL && ht[(size_t)b * curL + i] >= 0; ++i) ref.push_back(ht[(size_t)b * curL + i]);
          editErr += editDistance(hyp, ref);
          editLen += (long)ref.size();
          if (haveLists) {   // words: path -> letters (replabels undone, blank dropped) -> split at the word separator (Train.cpp:829-872)
            std::vector<int> pv(path.begin() + (size_t)b * To, path.begin() + (size_t)(b + 1) * To);
            const bool wp = flags.getb("usewordpiece", false);
            auto hw = tkn2Wrd(tknPrediction2Ltr(pv, data.dict, criterionName, flags.get("surround", ""), data.replabel, wp, data.wordsep), data.wordsep);
            auto rw = tkn2Wrd(tknTarget2Ltr(ref, data.dict, criterionName, flags.get("surround", ""), data.replabel, wp, data.wordsep), data.wordsep);
            wordMeter.add(hw, rw);
          }
        }
      }

      // backward
      bwdtimer.resume();
      netoptim->zeroGrad();
      critoptim->zeroGrad();
      loss.backward();
      if (reducer) {   // Train.cpp:1721-1735
        for (auto& p : network->params()) {
          if (!p.isGradAvailable()) p.addGrad(fl::Variable(af::constant(0.0, p.dims(), p.type()), false));
          reducer->add(p.grad());
        }
        for (auto& p : criterion->params()) {
          if (!p.isGradAvailable()) p.addGrad(fl::Variable(af::constant(0.0, p.dims(), p.type()), false));
          reducer->add(p.grad());
        }
        reducer->finalize();
      }
      af::sync();
      bwdtimer.stopAndIncUnit();

      // optimizer: scale down gradients by batchsize, clamp, update
      optimtimer.resume();
      af::array totalBatchSizeArr = af::constant((double)loss.dims(0), af::dim4(1), af::f32);   // Train.cpp:1743-1747
      if (reducer) fl::allReduce(totalBatchSizeArr);
      const double totalBatchSize = (double)totalBatchSizeArr.scalar<float>();
      // (Train.cpp:1748-1760: `p.grad() = p.grad() / totalBatchSize` per parameter -- here in place, the planned network's
      // gradient arena in one launch)
      fl::scaleGradients(network->params(), 1.0 / totalBatchSize);
      fl::scaleGradients(criterion->params(), 1.0 / totalBatchSize);
      if (maxgradnorm > 0) {
        auto params = network->params();
        if (clampCrit) {
          auto cp = criterion->params();
          params.insert(params.end(), cp.begin(), cp.end());
        }
        const bool dbg = flags.getb("w2l_debug_gradnorm", false);
        double gEm = 0, gCrit = 0, gNet = 0;
        if (dbg) {   // (norms after the division by the batch size; 1e30 never clips)
          if (output.isGradAvailable()) gEm = fl::clipGradNorm({output}, 1e30);
          gCrit = fl::clipGradNorm(criterion->params(), 1e30);
          gNet = fl::clipGradNorm(network->params(), 1e30);
        }
        const double gnorm = fl::clipGradNorm(params, maxgradnorm);
        if (dbg) std::cout << "[debug] update " << curBatch << " gradient norm " << gnorm << " (network " << gNet << ", criterion " << gCrit
                           << ", emissions (unscaled) " << gEm << ")" << std::endl;
      }
      critoptim->step();
      netoptim->step();
      af::sync();
      optimtimer.stopAndIncUnit();
      timer.stopAndIncUnit();
      nsamples += curB;
      framesTotal += (long)curB * curT;
      ++nbatches;

      if (isMaster && ((reportiters > 0 && curBatch % reportiters == 0) || curBatch == iters)) logStatus(curEpoch, curBatch, lr, lrcrit);
      if (reportiters > 0 && curBatch % reportiters == 0) {
        // every report starts the next readings afresh (resetTimeStatMeters + the train meters, Train.cpp:1081-1090, :1125-1131,
        // :1844-1847): a report's timers are those of its own window, not of the run so far
        timer.reset(); sampletimer.reset(); fwdtimer.reset(); critfwdtimer.reset(); bwdtimer.reset(); optimtimer.reset();
        runtime.total = 0;
        lossSum = 0; lossN = 0; editErr = 0; editLen = 0; tszTotal = 0; tszMax = 0; nsamples = 0; nbatches = 0; framesTotal = 0;
        wordMeter = fl::EditDistanceMeter();
      }
      if (curBatch % batchesPerEpoch == 0 || curBatch == iters) saveModels(curEpoch, curBatch);
    }
    if (auto* cr = dynamic_cast<fl::CoalescingReducer*>(reducer.get()))
      std::cout << "[Distributed] gradient collectives of the last update: " << cr->lastCollectives() << " (" << cr->lastOverlapped()
                << " issued on the side stream behind bucket events of the backward pass)" << std::endl;
    std::cout << "Finished training" << std::endl;
    return 0;
  } catch (const std::exception& e) {
    std::cerr << "Train: " << e.what() << std::endl;
    return 1;
  }
}
