// A model plugin in the shape of the reference's (recipes/slimIPL/100h_supervised.cpp:24-87): a fl::Container subclass
// with its own forward({features (T, NFEAT, 1, B), inputSizes (1, B)}), exported through
//     extern "C" fl::Module* createModule(int64_t nFeature, int64_t nLabel)
// and loaded by fl::pkg::runtime::ModulePlugin (dlopen).  The layers come from an arch text (the planned pipeline of
// libw2l_hip.so); the plugin contract -- symbol name, signature, owning raw pointer, forward inputs -- is the reference's.
#include <sstream>

#include "fl_compat/flashlight.h"

namespace {
class MyModel : public fl::Container {
 public:
  MyModel(int64_t nFeature, int64_t nLabel) {
    std::ostringstream a;
    a << "V -1 NFEAT 1 0\n"
      << "C2 1 4 5 1 2 1 -1 -1\nR\nDO 0.0\nLN 0 1 2\n"
      << "TDS 4 5 " << nFeature << " 0.0 " << 4 * nFeature * 2 << "\n"
      << "TDS 4 5 " << nFeature << " 0.0 0\n"
      << "V 0 " << 4 * nFeature << " 1 0\nRO 1 0 3 2\nL " << 4 * nFeature << " NLABEL\n";
    encoder_ = fl::pkg::speech::buildSequentialModuleFromText(a.str(), nFeature, nLabel);
    add(encoder_);
  }
  std::vector<fl::Variable> forward(const std::vector<fl::Variable>& input) override {
    // expected input dims T x C x 1 x B, input[1] = sizes (unused by this all-padded test model)
    if (input.size() < 2) throw std::invalid_argument("MyModel expects {features, inputSizes}");
    return {encoder_->forward(input[0])};  // fl::Sequential::forward(const Variable&), as in Flashlight
  }
  std::string prettyString() const override { return "Model: " + encoder_->prettyString(); }

 private:
  std::shared_ptr<fl::Sequential> encoder_;
};
}  // namespace

extern "C" __attribute__((visibility("default"))) fl::Module* createModule(int64_t nFeature, int64_t nLabel) {
  auto m = std::make_unique<MyModel>(nFeature, nLabel);
  return m.release();
}
