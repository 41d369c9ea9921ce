// CPU test of include/fl_compat/text.h (compiled with g++ by tests/test_text.py): the same worked examples as the Python
// mirror, so the two stay one specification.
#include <cassert>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/fl_compat/text.h"

using namespace fl::pkg::speech;
using fl::lib::text::Dictionary;

static std::vector<std::string> names(const std::vector<int>& v, const Dictionary& d) {
  std::vector<std::string> out;
  for (int i : v) out.push_back(d.getEntry(i));
  return out;
}
static std::vector<std::string> chars(const std::string& s) {
  std::vector<std::string> out;
  for (char c : s) out.push_back(std::string(1, c));
  return out;
}

int main() {
  std::vector<std::string> letters = {"|", "'"};
  for (char c = 'a'; c <= 'z'; ++c) letters.push_back(std::string(1, c));
  // class counts of the recipes (Train.cpp:235-251)
  Dictionary asg = createTokenDict(Dictionary(letters), "asg", 2);
  assert(asg.indexSize() == 30 && asg.getIndex("<1>") == 28 && asg.getIndex("<2>") == 29 && !asg.contains(kBlankToken));
  std::vector<std::string> pieces;
  for (int i = 0; i < 9997; ++i) pieces.push_back("_w" + std::to_string(i));
  Dictionary ctcBig = createTokenDict(Dictionary(pieces), "ctc", 0);
  assert(ctcBig.indexSize() == 9998 && ctcBig.getIndex(kBlankToken) == 9997);
  // replabels
  auto idx = [&](const std::string& s) { std::vector<int> v; for (char c : s) v.push_back(asg.getIndex(std::string(1, c))); return v; };
  auto packed = packReplabels(idx("hello|aaa|"), asg, 2);
  assert((names(packed, asg) == std::vector<std::string>{"h", "e", "l", "<1>", "o", "|", "a", "<2>", "|"}));
  assert(unpackReplabels(packed, asg, 2) == idx("hello|aaa|"));
  assert((names(packReplabels(idx("aaaa"), asg, 2), asg) == std::vector<std::string>{"a", "<2>", "a"}));
  // ASG prediction: frames | h h e l <1> <1> o | | -> hello
  std::vector<int> path = {asg.getIndex("|"), asg.getIndex("h"), asg.getIndex("h"), asg.getIndex("e"), asg.getIndex("l"),
                           asg.getIndex("<1>"), asg.getIndex("<1>"), asg.getIndex("o"), asg.getIndex("|"), asg.getIndex("|")};
  assert(tknPrediction2Ltr(path, asg, "asg", "|", 2, false, "|") == chars("hello"));
  // CTC with word pieces
  Dictionary wp = createTokenDict(Dictionary(std::vector<std::string>{"_the", "_c", "at", "_cat", "s"}), "ctc", 0);
  const int b = wp.getIndex(kBlankToken);
  std::vector<int> p2 = {b, wp.getIndex("_the"), wp.getIndex("_the"), b, b, wp.getIndex("_c"), wp.getIndex("at"), wp.getIndex("at"), b, wp.getIndex("s")};
  auto ltr = tknPrediction2Ltr(p2, wp, "ctc", "", 0, true, "_");
  assert(ltr == chars("the_cats"));
  assert((tkn2Wrd(ltr, "_") == std::vector<std::string>{"the", "cats"}));
  std::vector<int> tgt = {wp.getIndex("_the"), wp.getIndex("_cat"), -1, -1};
  auto lt = tknTarget2Ltr(tgt, wp, "ctc", "", 0, true, "_");
  assert((tkn2Wrd(lt, "_") == std::vector<std::string>{"the", "cat"}));
  fl::EditDistanceMeter ter, wer;
  ter.add(ltr, lt);
  wer.add(tkn2Wrd(ltr, "_"), tkn2Wrd(lt, "_"));
  assert(ter.errors() == 1 && ter.length() == 7 && wer.errors() == 1 && wer.length() == 2 && wer.value() == 50.0);
  // lexicon + target generation
  auto lex = fl::lib::text::loadWordsFromLines({"hello\th e l l o |", "aaa\ta a a |", "bee\tb e e |", "bee\tb e |"});
  assert(lex["bee"].size() == 2 && lex["bee"][1].size() == 3);
  assert((names(targetIndices({"hello", "aaa"}, lex, asg, "asg", 2, "|"), asg) ==
          std::vector<std::string>{"h", "e", "l", "<1>", "o", "|", "a", "<2>", "|"}));
  assert(targetIndices({"hello", "aaa"}, lex, asg, "ctc", 0, "|") == idx("hello|aaa|"));
  assert(targetIndices({"zed"}, lex, asg, "ctc", 0, "|") == idx("zed|"));                     // out of lexicon: letters + separator
  bool threw = false;
  try { targetIndices({"z3d"}, lex, asg, "ctc", 0, "|"); } catch (const std::invalid_argument&) { threw = true; }
  assert(threw);
  assert(wrd2Target({"z3d", "bee"}, lex, asg, "|", 0.f, false, true, true) == chars("bee|"));   // the reference's argument order
  assert(lex.at("bee").size() == 2);
  assert(fl::lib::text::loadWordsFromLines({"a\ta |", "b\tb |", "a\ta a |", "c\tc |"}, 2).size() == 2);   // maxWords caps WORDS
  // UTF-8 code points
  assert((splitWrd("a\xC3\xA9z") == std::vector<std::string>{"a", "\xC3\xA9", "z"}));
  std::printf("text ok\n");
  return 0;
}
