// fl_compat/data.h -- list files and rank partitioning of the input pipeline on the host (header only; Python mirror:
// wav2letter_amd/data.py, which documents the formats; tests: tests/cpp/data_test.cpp compiled with g++ by tests/test_data.py).
//
//   .lst line           `id path duration_ms transcript...`   data/librispeech/utils.py:36-46, read back at :49-57;
//                       consumed through --train / --valid (recipes/slimIPL/src/Train.cpp:327-339)
//   partitionByRoundRobin   Flashlight's rank partitioning [UNVENDORED, recalled]: global batches of world * batchsize
//                       consecutive samples, rank r takes the r-th slice of every global batch; the tail is split evenly,
//                       and with allowEmpty (validation sets) the first `rest % world` ranks take one sample more
#pragma once
#include <algorithm>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace fl {
namespace pkg {
namespace speech {

struct ListSample {
  std::string id, path;
  double durationMs = 0;
  std::vector<std::string> transcript;
};

inline std::vector<ListSample> parseList(const std::string& text) {
  std::vector<ListSample> out;
  std::istringstream lines(text);
  std::string line;
  int ln = 0;
  while (std::getline(lines, line)) {
    ++ln;
    std::istringstream ss(line);
    ListSample s;
    std::string dur, w;
    if (!(ss >> s.id)) continue;  // blank line
    if (!(ss >> s.path >> dur)) throw std::invalid_argument("list line " + std::to_string(ln) + ": expected 'id path duration [transcript]'");
    try {
      size_t used = 0;
      s.durationMs = std::stod(dur, &used);
      if (used != dur.size()) throw std::invalid_argument(dur);
    } catch (const std::exception&) {
      throw std::invalid_argument("list line " + std::to_string(ln) + ": duration '" + dur + "' is not a number");
    }
    while (ss >> w) s.transcript.push_back(w);
    out.push_back(std::move(s));
  }
  return out;
}

}  // namespace speech
}  // namespace pkg

namespace lib {

// sample indices of rank `rank`
inline std::vector<long> partitionByRoundRobin(long nSamples, int rank, int world, long batchSize, bool allowEmpty = false) {
  if (rank < 0 || rank >= world || batchSize <= 0 || nSamples < 0) throw std::invalid_argument("partitionByRoundRobin: bad arguments");
  const long perGlobal = (long)world * batchSize, nGlobal = nSamples / perGlobal;
  std::vector<long> out;
  for (long g = 0; g < nGlobal; ++g) {
    const long base = g * perGlobal + (long)rank * batchSize;
    for (long i = 0; i < batchSize; ++i) out.push_back(base + i);
  }
  const long rest = nSamples - nGlobal * perGlobal;
  if (rest >= world || (allowEmpty && rest > 0)) {
    long per = rest / world;
    const long remaining = rest % world;
    long base = nGlobal * perGlobal + (long)rank * per;
    if (allowEmpty) {
      base += std::min<long>(rank, remaining);
      per += rank < remaining ? 1 : 0;
    }
    for (long i = 0; i < per; ++i) out.push_back(base + i);
  }
  return out;
}

}  // namespace lib
}  // namespace fl
