// fl_compat/text.h -- the Trainer's token dictionary, target packing and evaluation remap on the reference's own names
// (header only, host code, no device dependency).  Python mirror: wav2letter_amd/text.py; tests: tests/test_text.py (Python)
// and tests/cpp/text_test.cpp (this header, compiled with g++ by the CPU test suite).
//
// In-repo witnesses: class inventory of a run recipes/slimIPL/src/Train.cpp:235-251 (tokens file, `<1>`..`<replabel>`, the CTC
// blank LAST); evaluation :829-872 (viterbiPath -> tknPrediction2Ltr / tknTarget2Ltr -> tkn2Wrd -> edit distances).  The
// functions themselves are un-vendored Flashlight (fl::lib::text::Dictionary, fl::pkg::speech::{packReplabels,
// unpackReplabels, tknPrediction2Ltr, tknTarget2Ltr, tknIdx2Ltr, tkn2Wrd}, fl::EditDistanceMeter): restated from their
// published behaviour.
#pragma once
#include <algorithm>
#include <cstdint>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

namespace fl {
namespace lib {
namespace text {

class Dictionary {
 public:
  Dictionary() {}
  explicit Dictionary(const std::string& path) {
    std::ifstream f(path);
    if (!f) throw std::runtime_error("Dictionary: cannot open " + path);
    std::string line;
    while (std::getline(f, line)) addLine(line);
  }
  explicit Dictionary(const std::vector<std::string>& lines) { for (auto& l : lines) addLine(l); }

  // one index per line; further entries on the same line are aliases of it
  void addLine(const std::string& line) {
    std::istringstream ss(line);
    std::string tok;
    int idx = -1;
    while (ss >> tok) {
      if (entry2idx_.count(tok)) throw std::invalid_argument("Dictionary: duplicate entry " + tok);
      if (idx < 0) { idx = (int)idx2entry_.size(); idx2entry_.push_back(tok); }
      entry2idx_[tok] = idx;
    }
  }
  void addEntry(const std::string& entry) { addLine(entry); }
  size_t indexSize() const { return idx2entry_.size(); }
  bool contains(const std::string& e) const { return entry2idx_.count(e) != 0; }
  int getIndex(const std::string& e) const {
    auto it = entry2idx_.find(e);
    if (it == entry2idx_.end()) throw std::invalid_argument("Dictionary: unknown entry " + e);
    return it->second;
  }
  const std::string& getEntry(int idx) const {
    if (idx < 0 || (size_t)idx >= idx2entry_.size()) throw std::invalid_argument("Dictionary: index out of range");
    return idx2entry_[(size_t)idx];
  }

 private:
  std::unordered_map<std::string, int> entry2idx_;
  std::vector<std::string> idx2entry_;
};

// word -> spellings (token strings), in file order: `word<TAB or space>tok tok ...`, one spelling per line
using LexiconMap = std::unordered_map<std::string, std::vector<std::vector<std::string>>>;

// maxWords: the reference's second argument (loadWords(FLAGS_lexicon, FLAGS_maxword), e.g. recipes/slimIPL/src/Train.cpp): at most
// that many distinct WORDS are kept (-1: all); reading STOPS at the first new word beyond the limit, as the reference's loop does
// (spellings of kept words further down the file are not collected)
inline LexiconMap loadWordsFromLines(const std::vector<std::string>& lines, int maxWords = -1) {
  LexiconMap lex;
  for (auto& line : lines) {
    std::istringstream ss(line);
    std::string word, tok;
    if (!(ss >> word)) continue;
    std::vector<std::string> sp;
    while (ss >> tok) sp.push_back(tok);
    if (sp.empty()) continue;
    // [UNVENDORED] fl::lib::text::loadWords is not in /root/reference: the limit is applied as "at most maxWords distinct
    // words; reading stops at the first NEW word beyond it", so further spellings of already-kept words that precede that
    // word in the file are kept.  If upstream breaks right after the insertion that reaches the limit, those spellings are
    // the one difference (recipes sort the lexicon by word, where the two rules coincide).
    if (lex.find(word) == lex.end() && maxWords >= 0 && (int)lex.size() >= maxWords) break;
    lex[word].push_back(sp);
  }
  return lex;
}
inline LexiconMap loadWords(const std::string& path, int maxWords = -1) {
  std::ifstream f(path);
  if (!f) throw std::runtime_error("loadWords: cannot open " + path);
  std::vector<std::string> lines;
  std::string line;
  while (std::getline(f, line)) lines.push_back(line);
  return loadWordsFromLines(lines, maxWords);
}

}  // namespace text
}  // namespace lib

namespace pkg {
namespace speech {

constexpr const char* kBlankToken = "#";
constexpr const char* kCtcCriterion = "ctc";
constexpr const char* kAsgCriterion = "asg";

inline std::string replabelToken(int r) { return "<" + std::to_string(r) + ">"; }

// Train.cpp:235-251
inline lib::text::Dictionary createTokenDict(lib::text::Dictionary d, const std::string& criterion, int replabel) {
  for (int r = 1; r <= replabel; ++r) d.addEntry(replabelToken(r));
  if (criterion == kCtcCriterion) d.addEntry(kBlankToken);
  return d;
}

// UTF-8 code points of a token
inline std::vector<std::string> splitWrd(const std::string& w) {
  std::vector<std::string> out;
  for (size_t i = 0; i < w.size();) {
    const unsigned char c = (unsigned char)w[i];
    const size_t n = c < 0x80 ? 1 : (c >> 5) == 0x6 ? 2 : (c >> 4) == 0xE ? 3 : (c >> 3) == 0x1E ? 4 : 1;
    out.push_back(w.substr(i, n));
    i += n;
  }
  return out;
}

// transcription words -> token strings, argument order of the reference's wrd2Target(words, lexicon, dict, wordSeparator,
// targetSamplePct, fallback2LtrWordSepLeft, fallback2LtrWordSepRight, skipUnk): the first spelling of the lexicon, or with
// probability targetSamplePct one of the word's spellings drawn uniformly (--sampletarget); an out-of-lexicon word falls back to
// its letters (word separator on the chosen sides) when every letter is a token, else it is skipped (skipUnk) or an error
inline std::vector<std::string> wrd2Target(const std::vector<std::string>& words, const lib::text::LexiconMap& lexicon,
                                           const lib::text::Dictionary& dict, const std::string& wordSeparator = "",
                                           float targetSamplePct = 0.f, bool fallback2LtrWordSepLeft = false,
                                           bool fallback2LtrWordSepRight = true, bool skipUnk = false) {
  static thread_local uint64_t rng = 0x9E3779B97F4A7C15ull;
  auto draw = [&]() { rng = rng * 6364136223846793005ull + 1442695040888963407ull; return (double)(rng >> 11) * (1.0 / 9007199254740992.0); };
  std::vector<std::string> out;
  for (auto& w : words) {
    auto it = lexicon.find(w);
    if (it != lexicon.end() && !it->second.empty()) {
      size_t pick = 0;
      if (targetSamplePct > 0.f && draw() < (double)targetSamplePct) pick = (size_t)(draw() * (double)it->second.size()) % it->second.size();
      out.insert(out.end(), it->second[pick].begin(), it->second[pick].end());
      continue;
    }
    auto letters = splitWrd(w);
    const bool spellable = std::all_of(letters.begin(), letters.end(), [&](const std::string& c) { return dict.contains(c); });
    if (spellable) {
      if (fallback2LtrWordSepLeft && !wordSeparator.empty()) out.push_back(wordSeparator);
      out.insert(out.end(), letters.begin(), letters.end());
      if (fallback2LtrWordSepRight && !wordSeparator.empty()) out.push_back(wordSeparator);
    } else if (!skipUnk) {
      throw std::invalid_argument("wrd2Target: word '" + w + "' is not in the lexicon and cannot be spelled with the token set");
    }
  }
  return out;
}

// `a a a b` -> `a <2> b`: a run becomes the token + the replabel counting the EXTRA repetitions
inline std::vector<int> packReplabels(const std::vector<int>& tokens, const lib::text::Dictionary& dict, int maxReps) {
  if (tokens.empty() || maxReps <= 0) return tokens;
  std::vector<int> repIdx((size_t)maxReps + 1);
  for (int r = 1; r <= maxReps; ++r) repIdx[(size_t)r] = dict.getIndex(replabelToken(r));
  std::vector<int> out;
  int prev = -1, reps = 0;
  for (int t : tokens) {
    if (t == prev && reps < maxReps) { ++reps; continue; }
    if (reps > 0) { out.push_back(repIdx[(size_t)reps]); reps = 0; }
    out.push_back(t);
    prev = t;
  }
  if (reps > 0) out.push_back(repIdx[(size_t)reps]);
  return out;
}

inline std::vector<int> unpackReplabels(const std::vector<int>& tokens, const lib::text::Dictionary& dict, int maxReps) {
  if (tokens.empty() || maxReps <= 0) return tokens;
  std::unordered_map<int, int> value;
  for (int r = 1; r <= maxReps; ++r) value[dict.getIndex(replabelToken(r))] = r;
  std::vector<int> out;
  int prev = -1;
  for (int t : tokens) {
    auto it = value.find(t);
    if (it == value.end()) { out.push_back(t); prev = t; }
    else if (prev != -1) { out.insert(out.end(), (size_t)it->second, prev); prev = -1; }
  }
  return out;
}

// one transcription -> the int32 target row of the criterion (before -1 padding)
inline std::vector<int> targetIndices(const std::vector<std::string>& words, const lib::text::LexiconMap& lexicon,
                                      const lib::text::Dictionary& dict, const std::string& criterion, int replabel,
                                      const std::string& wordSeparator) {
  std::vector<int> idx;
  for (auto& t : wrd2Target(words, lexicon, dict, wordSeparator, 0.f)) idx.push_back(dict.getIndex(t));
  return criterion == kAsgCriterion && replabel > 0 ? packReplabels(idx, dict, replabel) : idx;
}

inline void uniq(std::vector<int>& v) { v.erase(std::unique(v.begin(), v.end()), v.end()); }

inline std::vector<std::string> tknIdx2Ltr(const std::vector<int>& labels, const lib::text::Dictionary& d, bool useWordPiece,
                                           const std::string& wordSep) {
  std::vector<std::string> out;
  for (int id : labels) {
    const std::string& tok = d.getEntry(id);
    if (useWordPiece) { for (auto& c : splitWrd(tok)) out.push_back(c); }
    else out.push_back(tok);
  }
  if (!out.empty() && !wordSep.empty()) {
    if (out.front() == wordSep) out.erase(out.begin());
    if (!out.empty() && out.back() == wordSep) out.pop_back();
  }
  return out;
}

inline void remapLabels(std::vector<int>& labels, const lib::text::Dictionary& dict, const std::string& surround, int replabel) {
  if (replabel > 0) labels = unpackReplabels(labels, dict, replabel);
  if (!surround.empty() && dict.contains(surround)) {
    const int s = dict.getIndex(surround);
    if (!labels.empty() && labels.back() == s) labels.pop_back();
    if (!labels.empty() && labels.front() == s) labels.erase(labels.begin());
  }
}

// a Viterbi path (one label per frame) -> letters: collapse repeated frames, drop the CTC blank, undo replabels
inline std::vector<std::string> tknPrediction2Ltr(std::vector<int> tokens, const lib::text::Dictionary& dict,
                                                  const std::string& criterion, const std::string& surround, int replabel,
                                                  bool useWordPiece, const std::string& wordSep) {
  tokens.erase(std::remove_if(tokens.begin(), tokens.end(), [](int t) { return t < 0; }), tokens.end());
  if (tokens.empty()) return {};
  if (criterion == kCtcCriterion || criterion == kAsgCriterion) uniq(tokens);
  if (criterion == kCtcCriterion) {
    const int blank = dict.getIndex(kBlankToken);
    tokens.erase(std::remove(tokens.begin(), tokens.end(), blank), tokens.end());
  }
  remapLabels(tokens, dict, surround, criterion == kAsgCriterion ? replabel : 0);
  return tknIdx2Ltr(tokens, dict, useWordPiece, wordSep);
}

inline std::vector<std::string> tknTarget2Ltr(std::vector<int> tokens, const lib::text::Dictionary& dict, const std::string& criterion,
                                              const std::string& surround, int replabel, bool useWordPiece,
                                              const std::string& wordSep) {
  tokens.erase(std::remove_if(tokens.begin(), tokens.end(), [](int t) { return t < 0; }), tokens.end());   // batch padding
  if (tokens.empty()) return {};
  remapLabels(tokens, dict, surround, criterion == kAsgCriterion ? replabel : 0);
  return tknIdx2Ltr(tokens, dict, useWordPiece, wordSep);
}

inline std::vector<std::string> tkn2Wrd(const std::vector<std::string>& letters, const std::string& wordSep) {
  std::vector<std::string> words;
  std::string cur;
  for (auto& t : letters) {
    if (t == wordSep) { if (!cur.empty()) { words.push_back(cur); cur.clear(); } }
    else cur += t;
  }
  if (!cur.empty()) words.push_back(cur);
  return words;
}

}  // namespace speech
}  // namespace pkg

// fl::EditDistanceMeter as the Trainer logs it: 100 * (ins + del + sub) / reference length over everything added
class EditDistanceMeter {
 public:
  template <class T>
  void add(const std::vector<T>& hyp, const std::vector<T>& ref) {
    std::vector<size_t> prev(ref.size() + 1), cur(ref.size() + 1);
    for (size_t j = 0; j <= ref.size(); ++j) prev[j] = j;
    for (size_t i = 1; i <= hyp.size(); ++i) {
      cur[0] = i;
      for (size_t j = 1; j <= ref.size(); ++j)
        cur[j] = std::min({prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (hyp[i - 1] == ref[j - 1] ? 0u : 1u)});
      std::swap(prev, cur);
    }
    errors_ += prev[ref.size()];
    length_ += ref.size();
  }
  double value() const { return length_ ? 100.0 * (double)errors_ / (double)length_ : 0.0; }
  size_t errors() const { return errors_; }
  size_t length() const { return length_; }
  void reset() { errors_ = length_ = 0; }

 private:
  size_t errors_ = 0, length_ = 0;
};

}  // namespace fl
