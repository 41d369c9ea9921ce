// CPU test of include/fl_compat/data.h (compiled with g++ by tests/test_data.py)
#include <cassert>
#include <cstdio>
#include <set>

#include "../../include/fl_compat/data.h"

int main() {
  auto s = fl::pkg::speech::parseList("u1 /a/b.flac 1234.5 hello world\n\nu2 /c.wav 10\n");
  assert(s.size() == 2 && s[0].id == "u1" && s[0].path == "/a/b.flac" && s[0].durationMs == 1234.5);
  assert((s[0].transcript == std::vector<std::string>{"hello", "world"}) && s[1].transcript.empty());
  bool threw = false;
  try { fl::pkg::speech::parseList("u1 /a x12 hi\n"); } catch (const std::invalid_argument&) { threw = true; }
  assert(threw);
  // every rank gets the same number of samples without allowEmpty; the union is everything with it
  for (long n : {0L, 5L, 37L, 64L, 101L})
    for (int world : {1, 2, 3, 8}) {
      std::set<long> all;
      size_t first = 0;
      for (int r = 0; r < world; ++r) {
        auto p = fl::lib::partitionByRoundRobin(n, r, world, 4, false);
        if (r == 0) first = p.size();
        assert(p.size() == first);
        auto q = fl::lib::partitionByRoundRobin(n, r, world, 4, true);
        for (long i : q) { assert(i >= 0 && i < n && !all.count(i)); all.insert(i); }
      }
      assert((long)all.size() == n);
    }
  auto p = fl::lib::partitionByRoundRobin(20, 1, 2, 4);   // global batches of 8: rank 1 takes 4..7, 12..15, then the tail 18, 19
  assert((p == std::vector<long>{4, 5, 6, 7, 12, 13, 14, 15, 18, 19}));
  std::printf("data ok\n");
  return 0;
}
