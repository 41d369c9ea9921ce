// A model plugin that builds its network out of LAYER OBJECTS, the way the reference's plugins do
// (recipes/slimIPL/100h_supervised.cpp:24-43: `convFrontend_->add(std::make_shared<fl::Conv2D>(...))`, `fl::View`,
// `fl::LayerNorm`, `fl::Dropout`, `fl::Reorder`, `fl::Linear`, a loop of blocks), exported through
//     extern "C" fl::Module* createModule(int64_t nFeature, int64_t nLabel)
// and loaded by fl::pkg::runtime::ModulePlugin.  The layers are the ones of tests/cpp/plugin_model.cpp's arch text, so the
// two plugins and the arch file must train identically (tests/test_gpu_fl_compat.py).
#include "fl_compat/flashlight.h"

namespace {
class MyModel : public fl::Container {
 public:
  MyModel(int64_t nFeature, int64_t nLabel) {
    const int nf = (int)nFeature;
    encoder_->add(std::make_shared<fl::View>(af::dim4(-1, nf, 1, 0)));
    // Time x nFeature x 1 x Batch
    encoder_->add(std::make_shared<fl::Conv2D>(1, 4, 5, 1, 2, 1, -1, -1));
    fl::Container& asContainer = *encoder_;   // Container::add is virtual: through a base reference a layer object still contributes its line
    asContainer.add(std::make_shared<fl::ReLU>());
    encoder_->add(std::make_shared<fl::Dropout>(0.0));
    std::vector<int> lnDims = {0, 1, 2};
    encoder_->add(std::make_shared<fl::LayerNorm>(lnDims));
    for (int blk = 0; blk < 2; ++blk) {
      auto layer = std::make_shared<fl::TDSBlock>(4, 5, nf, 0.0, blk == 0 ? 4 * nf * 2 : 0);
      blocks_.push_back(layer);
      encoder_->add(layer);
    }
    encoder_->add(fl::View(af::dim4(0, 4 * nf, 1, 0)));          // fl's add(const T&)
    encoder_->add(std::make_shared<fl::Reorder>(1, 0, 3, 2));
    encoder_->add(std::make_shared<fl::Linear>(4 * nf, (int)nLabel));
    if (encoder_->param(0).elements() != 4 * 5) throw std::logic_error("param(0) of an unplanned Sequential");   // param(i) plans, like params()
    add(encoder_);
  }
  std::vector<fl::Variable> forward(const std::vector<fl::Variable>& input) override {
    if (input.size() < 2) throw std::invalid_argument("MyModel expects {features, inputSizes}");
    return {encoder_->forward(input[0])};
  }
  std::string prettyString() const override { return "Model: " + encoder_->prettyString(); }

 private:
  std::shared_ptr<fl::Sequential> encoder_{std::make_shared<fl::Sequential>()};
  std::vector<std::shared_ptr<fl::TDSBlock>> blocks_;
};
}  // namespace

extern "C" __attribute__((visibility("default"))) fl::Module* createModule(int64_t nFeature, int64_t nLabel) {
  auto m = std::make_unique<MyModel>(nFeature, nLabel);
  return m.release();
}
