// arch.cpp -- arch-file grammar and gflags-style flag files.
// Grammar restated from recipes/joint_training_vox_populi/cpc/SequentialBuilder.cpp:
//   pre-processing (trim, NFEAT/NLABEL substitution, '#' and blank lines)  :39-48
//   token arities                                                          :92-626
// Every token the 30 arch files of the reference use is recognised and arity-checked;
// building is limited to the tokens the hot path needs (net.cpp).
#include <algorithm>
#include <cstdlib>
#include <fstream>
#include <map>
#include <sstream>

#include "w2l_host.hpp"

namespace w2l {

std::string readFile(const std::string& path) {
  std::ifstream f(path);
  if (!f) throw std::invalid_argument("cannot open file: " + path);
  std::stringstream ss;
  ss << f.rdbuf();
  return ss.str();
}

static std::string trim(const std::string& s) {
  size_t a = s.find_first_not_of(" \t\r\n");
  if (a == std::string::npos) return "";
  size_t b = s.find_last_not_of(" \t\r\n");
  return s.substr(a, b - a + 1);
}

static void replaceAll(std::string& s, const std::string& from, const std::string& to) {
  size_t pos = 0;
  while ((pos = s.find(from, pos)) != std::string::npos) {
    s.replace(pos, from.size(), to);
    pos += to.size();
  }
}

static std::vector<std::string> splitWs(const std::string& s) {
  std::vector<std::string> out;
  std::istringstream is(s);
  std::string w;
  while (is >> w) out.push_back(w);
  return out;
}

struct Arity { int lo, hi; bool evenOnly; };
// number of whitespace-separated fields INCLUDING the token (as the reference counts params.size())
static const std::map<std::string, Arity>& arities() {
  static const std::map<std::string, Arity> m = {
      {"RO", {5, 5, false}},   {"V", {5, 5, false}},    {"PD", {4, 10, true}},  {"TR", {6, 9, false}},
      {"CFR", {7, 8, false}},  {"POSEMB", {3, 4, false}}, {"SINPOSEMB", {2, 3, false}}, {"C", {5, 9, false}},
      {"C1", {5, 9, false}},   {"TDS", {4, 8, false}},  {"AC", {5, 8, false}},  {"C2", {7, 11, false}},
      {"L", {3, 4, false}},    {"E", {3, 3, false}},    {"ADAPTIVEE", {3, 3, false}}, {"BN", {3, 5, false}},
      {"LN", {2, 4, false}},   {"WN", {3, 99, false}},  {"DO", {2, 2, false}},  {"M", {5, 7, false}},
      {"A", {5, 7, false}},    {"ELU", {1, 1, false}},  {"R", {1, 1, false}},   {"R6", {1, 1, false}},
      {"PR", {1, 3, false}},   {"LG", {1, 1, false}},   {"HT", {1, 1, false}},  {"T", {1, 1, false}},
      {"GLU", {2, 2, false}},  {"LSM", {2, 2, false}},  {"SH", {1, 2, false}},  {"RNN", {3, 99, false}},
      {"GRU", {3, 99, false}}, {"LSTM", {3, 99, false}}, {"RES", {4, 99, false}}, {"SKIP", {3, 4, false}},
      {"SKIPL", {4, 5, false}}, {"SAUG", {7, 7, false}}, {"PC", {2, 2, false}},
  };
  return m;
}

static LayerSpec parseLine(const std::string& line, int lineNo) {
  auto f = splitWs(line);
  if (f.empty()) throw std::invalid_argument("Failed parsing - " + line);
  LayerSpec s;
  s.tok = f[0];
  s.line = line;
  s.lineNo = lineNo;
  auto it = arities().find(s.tok);
  if (it == arities().end()) throw std::invalid_argument("Failed parsing - unknown layer: " + line);
  int n = (int)f.size();
  if (n < it->second.lo || n > it->second.hi || (it->second.evenOnly && (n & 1)))
    throw std::invalid_argument("Failed parsing - " + line);
  if (s.tok == "WN") {
    // WN <dim> <child layer line>
    s.args = {f[1]};
    std::string rest;
    for (size_t i = 2; i < f.size(); ++i) rest += (i > 2 ? " " : "") + f[i];
    s.child = std::make_shared<LayerSpec>(parseLine(rest, lineNo));
    return s;
  }
  if (s.tok == "LN") {
    // the reference's builder rejects the pre-migration form `LN 3` (SequentialBuilder.cpp:366-375)
    if (f.size() == 2 && f[1] == "3")
      throw std::invalid_argument(
          "Failed parsing - flashlight LayerNorm API for specifying `featAxes` is modified recently. "
          "You probably would want to specify LN 0 1 2 instead of LN 3: " + line);
  }
  s.args.assign(f.begin() + 1, f.end());
  return s;
}

std::vector<LayerSpec> parseArch(const std::string& text, int64_t nFeat, int64_t nLabel) {
  std::vector<LayerSpec> out;
  std::istringstream is(text);
  std::string raw;
  int lineNo = 0;
  while (std::getline(is, raw)) {
    ++lineNo;
    std::string line = trim(raw);
    if (line.empty() || line[0] == '#') continue;
    replaceAll(line, "NFEAT", std::to_string(nFeat));
    replaceAll(line, "NLABEL", std::to_string(nLabel));
    out.push_back(parseLine(line, lineNo));
  }
  if (out.empty()) throw std::invalid_argument("empty architecture");
  return out;
}

// ---------------------------------------------------------------------------- flags
bool Flags::has(const std::string& k) const {
  for (auto& p : kv) if (p.first == k) return true;
  return false;
}
std::string Flags::get(const std::string& k, const std::string& def) const {
  std::string v = def;
  for (auto& p : kv) if (p.first == k) v = p.second;  // last one wins (command line overrides file)
  return v;
}
double Flags::getd(const std::string& k, double def) const { return has(k) ? std::atof(get(k).c_str()) : def; }
long Flags::geti(const std::string& k, long def) const { return has(k) ? std::atol(get(k).c_str()) : def; }
bool Flags::getb(const std::string& k, bool def) const {
  if (!has(k)) return def;
  std::string v = get(k);
  std::transform(v.begin(), v.end(), v.begin(), ::tolower);
  return v == "true" || v == "1" || v == "yes" || v == "t" || v == "y" || v.empty();
}
void Flags::set(const std::string& k, const std::string& v) { kv.emplace_back(k, v); }

Flags parseFlagsText(const std::string& text) {
  Flags fl;
  std::istringstream is(text);
  std::string raw;
  while (std::getline(is, raw)) {
    std::string line = trim(raw);
    if (line.empty() || line[0] == '#') continue;
    if (line.rfind("--", 0) == 0) line = line.substr(2);
    else if (line.rfind("-", 0) == 0) line = line.substr(1);
    else throw std::invalid_argument("flags file: expected --flag=value, got: " + raw);
    size_t eq = line.find('=');
    std::string k = eq == std::string::npos ? line : line.substr(0, eq);
    std::string v = eq == std::string::npos ? "" : line.substr(eq + 1);
    if (k.rfind("no", 0) == 0 && eq == std::string::npos && k.size() > 2) { fl.set(k.substr(2), "false"); continue; }
    if (k == "flagsfile") {
      Flags sub = parseFlagsFile(v);
      for (auto& p : sub.kv) fl.kv.push_back(p);
      continue;
    }
    fl.set(k, v);
  }
  return fl;
}

Flags parseFlagsFile(const std::string& path) { return parseFlagsText(readFile(path)); }

// getCriterionScaleMode(FLAGS_onorm, FLAGS_sqnorm)  (recipes/slimIPL/src/Train.cpp:389)
int criterionScaleMode(const std::string& onorm, bool sqnorm) {
  if (onorm == "none" || onorm.empty()) return W2L_SCALE_NONE;
  if (onorm == "input") return sqnorm ? W2L_SCALE_INPUT_SZ_SQRT : W2L_SCALE_INPUT_SZ;
  if (onorm == "target") return sqnorm ? W2L_SCALE_TARGET_SZ_SQRT : W2L_SCALE_TARGET_SZ;
  throw std::invalid_argument("invalid onorm option: " + onorm);
}

void hipCheck(hipError_t e, const char* what) {
  if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}
void w2lCheck(int status, const char* what) {
  if (status == W2L_OK) return;
  if (status == W2L_EINVAL) throw std::invalid_argument(std::string(what) + ": invalid argument");
  if (status == W2L_EUNSUPPORTED) throw std::runtime_error(std::string(what) + ": shape unsupported by this build");
  throw std::runtime_error(std::string(what) + ": HIP error " + std::to_string(w2l_last_hip_error()));
}

}  // namespace w2l
