// fl_compat/audio.h -- the audio side of the input pipeline on the reference's C++ names (header only):
//
//   fl::pkg::speech::loadSound(path)        libsndfile in the reference (un-vendored); here RIFF/WAV PCM (8/16/24/32 bit), native
//                                           FLAC through the library's decoder (w2l_flac_*), headerless .raw/.pcm (int16) and .f32
//   fl::lib::audio::Mfsc                    log-mel filterbank features as the Trainer configures them (recipes/slimIPL/src/Train.cpp:
//                                           277-290; LogMelFeature.cpp:78-95): pre-emphasis . Hamming window . DFT folded into one
//                                           matrix (fp64, host, once), the frames of an utterance = overlapping rows of its samples,
//                                           so the spectra are ONE w2l_gemm_f32; |.|, the mel GEMM, log(max(., floor)) and the
//                                           transposition to the network input layout (T, NFEAT, 1, B) are three more launches
//   fl::pkg::speech::featurize(...)         a padded batch of utterances -> normalised features on the device + input sizes
//
// Python mirror: wav2letter_amd/features.py, wav2letter_amd/data.py (same arithmetic; tests/test_gpu_fl_compat.py holds the Train
// binary's features to it).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../w2l_hip.h"
#include "flashlight.h"

namespace fl {
namespace pkg {
namespace speech {

struct Sound {
  std::vector<float> samples;   // mono, [-1, 1)
  int rate = 0;
};

inline std::vector<uint8_t> readFileBytes(const std::string& path) {
  std::ifstream f(path, std::ios::binary);
  if (!f) throw std::runtime_error("loadSound: cannot open " + path);
  return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

inline bool endsWith(const std::string& s, const char* suffix) {
  const size_t n = strlen(suffix);
  if (s.size() < n) return false;
  for (size_t i = 0; i < n; ++i)
    if (std::tolower((unsigned char)s[s.size() - n + i]) != suffix[i]) return false;
  return true;
}

inline Sound loadSound(const std::string& path) {
  Sound out;
  const std::vector<uint8_t> d = readFileBytes(path);
  auto le = [&](size_t off, int bytes) { uint32_t v = 0; for (int i = 0; i < bytes; ++i) v |= (uint32_t)d[off + i] << (8 * i); return v; };
  if (endsWith(path, ".flac")) {
    int rate = 0, ch = 0, bps = 0;
    uint64_t total = 0, done = 0;
    int md5 = 0;
    if (w2l_flac_info(d.data(), d.size(), &rate, &ch, &bps, &total) != W2L_OK) throw std::runtime_error(path + ": " + w2l_flac_last_error());
    const uint64_t cap = total ? total : (uint64_t)d.size() * 8;
    std::vector<int32_t> pcm((size_t)cap * ch);
    if (w2l_flac_decode(d.data(), d.size(), pcm.data(), cap, &done, &md5) != W2L_OK) throw std::runtime_error(path + ": " + w2l_flac_last_error());
    out.rate = rate;
    out.samples.resize((size_t)done);
    const float sc = 1.0f / (float)(1u << (bps - 1));
    for (size_t i = 0; i < (size_t)done; ++i) {
      float a = 0.f;
      for (int c = 0; c < ch; ++c) a += (float)pcm[i * ch + c] * sc;
      out.samples[i] = ch > 1 ? a / (float)ch : a;
    }
    return out;
  }
  if (endsWith(path, ".wav")) {
    if (d.size() < 12 || memcmp(d.data(), "RIFF", 4) != 0 || memcmp(d.data() + 8, "WAVE", 4) != 0) throw std::runtime_error(path + ": not a RIFF/WAVE file");
    int ch = 0, bits = 0, fmt = 0;
    size_t off = 12;
    while (off + 8 <= d.size()) {
      const uint32_t len = le(off + 4, 4);
      if (memcmp(d.data() + off, "fmt ", 4) == 0 && len >= 16 && off + 24 <= d.size()) {   // (a truncated header must not be read past the file)
        fmt = (int)le(off + 8, 2); ch = (int)le(off + 10, 2); out.rate = (int)le(off + 12, 4); bits = (int)le(off + 22, 2);
      } else if (memcmp(d.data() + off, "data", 4) == 0) {
        if (fmt != 1 || ch < 1 || (bits != 8 && bits != 16 && bits != 24 && bits != 32)) throw std::runtime_error(path + ": unsupported WAV encoding (PCM 8/16/24/32 only)");
        const size_t bytesPer = (size_t)bits / 8, avail = std::min<size_t>(len, d.size() - off - 8), n = avail / (bytesPer * ch);
        out.samples.resize(n);
        for (size_t i = 0; i < n; ++i) {
          float a = 0.f;
          for (int c = 0; c < ch; ++c) {
            const size_t p = off + 8 + (i * ch + c) * bytesPer;
            float v;
            if (bits == 8) v = ((float)d[p] - 128.f) / 128.f;
            else if (bits == 16) v = (float)(int16_t)le(p, 2) / 32768.f;
            else if (bits == 24) { int32_t x = (int32_t)(le(p, 3) << 8) >> 8; v = (float)x / 8388608.f; }
            else v = (float)(int32_t)le(p, 4) / 2147483648.f;
            a += v;
          }
          out.samples[i] = ch > 1 ? a / (float)ch : a;
        }
        return out;
      }
      off += 8 + len + (len & 1);
    }
    throw std::runtime_error(path + ": no data chunk");
  }
  if (endsWith(path, ".raw") || endsWith(path, ".pcm")) {
    out.rate = 16000;
    out.samples.resize(d.size() / 2);
    for (size_t i = 0; i < out.samples.size(); ++i) out.samples[i] = (float)(int16_t)le(2 * i, 2) / 32768.f;
    return out;
  }
  if (endsWith(path, ".f32")) {
    out.rate = 16000;
    out.samples.resize(d.size() / 4);
    memcpy(out.samples.data(), d.data(), out.samples.size() * 4);
    return out;
  }
  throw std::runtime_error(path + ": no decoder for this container (WAV, FLAC and raw PCM are read; the reference uses libsndfile)");
}

}  // namespace speech
}  // namespace pkg

namespace lib {
namespace audio {

struct FeatureParams {   // the fields of fl::lib::audio::FeatureParams the Trainer sets (Train.cpp:277-290)
  int samplingFreq = 16000, frameSizeMs = 25, frameStrideMs = 10, numFilterbankChans = 80;
  float preemCoef = 0.97f, melFloor = 1.0f;
  bool usePower = false;
};

class Mfsc {
 public:
  explicit Mfsc(const FeatureParams& p) : p_(p) {
    N_ = (int)std::lround(1e-3 * p.frameSizeMs * p.samplingFreq);
    S_ = (int)std::lround(1e-3 * p.frameStrideMs * p.samplingFreq);
    nfft_ = 1;
    while (nfft_ < N_) nfft_ <<= 1;
    nb_ = nfft_ / 2 + 1;
    ld_ = (nb_ + 31) / 32 * 32;
    const int F = p.numFilterbankChans;
    const double pi = std::acos(-1.0);
    // frame -> windowed pre-emphasised frame: y = W P x, (P x)[i] = x[i] - a x[i-1], (P x)[0] = (1 - a) x[0]; then the DFT
    std::vector<double> WP((size_t)N_ * N_, 0.0);   // WP[i][j]
    for (int i = 0; i < N_; ++i) {
      const double w = 0.54 - 0.46 * std::cos(2.0 * pi * i / (N_ - 1));
      WP[(size_t)i * N_ + i] = w * (i == 0 ? 1.0 - p.preemCoef : 1.0);
      if (i > 0) WP[(size_t)i * N_ + i - 1] = -w * p.preemCoef;
    }
    std::vector<float> G((size_t)N_ * 2 * nb_);     // G[j][k] = sum_i WP[i][j] cos(2 pi i k / nfft) | -sin
    for (int j = 0; j < N_; ++j)
      for (int k = 0; k < nb_; ++k) {
        double re = 0, im = 0;
        for (int i = j; i <= std::min(j + 1, N_ - 1); ++i) {   // WP is bidiagonal: column j has rows j and j + 1
          const double a = WP[(size_t)i * N_ + j], ang = 2.0 * pi * (double)i * k / nfft_;
          re += a * std::cos(ang);
          im -= a * std::sin(ang);
        }
        G[(size_t)j * 2 * nb_ + k] = (float)re;
        G[(size_t)j * 2 * nb_ + nb_ + k] = (float)im;
      }
    std::vector<float> H((size_t)ld_ * F, 0.f);     // mel filterbank (HTK mel scale), rows padded to whole K tiles
    auto mel = [](double f) { return 2595.0 * std::log10(1.0 + f / 700.0); };
    auto imel = [](double m) { return 700.0 * (std::pow(10.0, m / 2595.0) - 1.0); };
    std::vector<double> pts((size_t)F + 2);
    for (int i = 0; i < F + 2; ++i) pts[i] = imel(mel(0.0) + (mel(p.samplingFreq / 2.0) - mel(0.0)) * i / (F + 1)) * (nb_ - 1) * 2.0 / p.samplingFreq;
    for (int k = 0; k < nb_; ++k)
      for (int f = 0; f < F; ++f) {
        const double hi = (k - pts[f]) / (pts[f + 1] - pts[f]), lo = (pts[f + 2] - k) / (pts[f + 2] - pts[f + 1]);
        H[(size_t)k * F + f] = (float)std::max(std::min(hi, lo), 0.0);
      }
    G_ = af::array(af::dim4(2 * nb_, N_), G.data());
    H_ = af::array(af::dim4(F, ld_), H.data());
  }
  int frameSize() const { return N_; }
  int frameStride() const { return S_; }
  int numFrames(long nSamples) const { return nSamples < N_ ? 0 : (int)(1 + (nSamples - N_) / S_); }

  // audio: device, dims (ns, B) = memory [B][ns] with ns a multiple of the frame stride -> features dims (T, F, 1, B)
  af::array apply(const af::array& audio) const {
    const long ns = audio.dims(0);
    const int B = (int)audio.dims(1), F = p_.numFilterbankChans;
    if (ns % S_ != 0) throw std::invalid_argument("Mfsc::apply: the padded utterance length must be a multiple of the frame stride");
    const int T = numFrames(ns), Tp = (int)(ns / S_);
    if (T <= 0) throw std::invalid_argument("Mfsc::apply: utterance shorter than one frame");
    void* s = fl::currentStream();
    const long rows = ((long)B * ns - N_) / S_ + 1;   // rows that straddle two utterances are computed and dropped
    af::array reim(af::dim4(2 * nb_, (long)B * Tp)), spec(af::dim4(ld_, (long)B * Tp)), melv(af::dim4(F, (long)B * Tp)), out(af::dim4(T, F, 1, B));
    chk(w2l_gemm_f32((int)rows, 2 * nb_, N_, audio.device<float>(), S_, 1, G_.device<float>(), 2 * nb_, 0, reim.device<float>(), 2 * nb_, nullptr, 0, 1, s), "mfsc spectrum gemm");
    chk(w2l_mfsc_spectrum(reim.device<float>(), spec.device<float>(), (size_t)rows, nb_, ld_, p_.usePower ? 1 : 0, s), "mfsc spectrum");
    chk(w2l_gemm_f32((int)rows, F, ld_, spec.device<float>(), ld_, 1, H_.device<float>(), F, 0, melv.device<float>(), F, nullptr, 0, 1, s), "mfsc mel gemm");
    chk(w2l_mfsc_log_transpose(melv.device<float>(), out.device<float>(), B, Tp, T, F, p_.melFloor, s), "mfsc log");
    return out;
  }

 private:
  static void chk(int st, const char* what) { if (st != W2L_OK) throw std::runtime_error(std::string(what) + ": w2l status " + std::to_string(st)); }
  FeatureParams p_;
  int N_ = 0, S_ = 0, nfft_ = 0, nb_ = 0, ld_ = 0;
  af::array G_, H_;
};

}  // namespace audio
}  // namespace lib
}  // namespace fl
