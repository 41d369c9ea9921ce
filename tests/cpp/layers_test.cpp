// The layer classes of include/fl_compat/flashlight.h are their lines of the arch grammar: the front end and blocks of the
// reference's plugin (recipes/slimIPL/100h_supervised.cpp:24-43) and of its arch files, constructor arguments in the
// reference's order.  Header-only part: plain g++, no library, no GPU (tests/test_recipes.py).
#include <iostream>

#include "fl_compat/flashlight.h"

static int fails = 0;
static void expect(const fl::ArchLayer& l, const std::string& want) {
  if (l.archLine() != want) { std::cerr << "got `" << l.archLine() << "` want `" << want << "`\n"; ++fails; }
}

int main() {
  // 100h_supervised.cpp:16-27
  expect(fl::View(af::dim4(-1, 1, 80, 0)), "V -1 1 80 0");
  expect(fl::LayerNorm(std::vector<int>{0, 1, 2}), "LN 0 1 2");
  expect(fl::Conv2D(80, 1536, 7, 1, 3, 1, -1, 0, 1, 1), "C2 80 1536 7 1 3 1 -1 0");
  expect(fl::GatedLinearUnit(2), "GLU 2");
  expect(fl::Dropout(0.3), "DO 0.3");
  expect(fl::Reorder(2, 0, 3, 1), "RO 2 0 3 1");
  expect(fl::Reorder(1, 0), "RO 1 0 2 3");          // fl::Reorder(dim0, dim1, dim2 = 2, dim3 = 3)
  expect(fl::LayerNorm(3), "LN 3");                  // fl::LayerNorm(int axis)
  expect(fl::Transformer(768, 192, 3072, 4, 920, 0.3f, 0.3f, false, false), "TR 768 3072 4 920 0.300000012 0.300000012");
  expect(fl::Linear(768, 31), "L 768 31");
  // am_tds_ctc.arch / am_transformer_ctc.arch
  expect(fl::TDSBlock(10, 21, 80, 0.05, 2400), "TDS 10 21 80 0.05 2400");
  expect(fl::Pool2D(2, 1, 2, 1), "M 2 1 2 1");
  expect(fl::WeightNorm(fl::Conv2D(80, 200, 13, 1, 1, 1, 170, 0), 3), "WN 3 C2 80 200 13 1 1 1 170 0");
  expect(fl::Conv2D(1, 16, 21, 3, 2, 1, -1, -1, 1, 2), "C2 1 16 21 3 2 1 -1 -1 1 2");
  bool threw = false;
  try { fl::Transformer(768, 100, 3072, 4, 920, 0.f, 0.f); } catch (const std::invalid_argument&) { threw = true; }
  if (!threw) { std::cerr << "headDim * nHeads != modelDim accepted\n"; ++fails; }
  threw = false;
  try { fl::View(af::dim4(1, 1, 1, 1)).forward({}); } catch (const std::logic_error&) { threw = true; }
  if (!threw) { std::cerr << "a lone layer object ran\n"; ++fails; }
  std::cout << (fails ? "FAIL" : "ok") << "\n";
  return fails ? 1 : 0;
}
