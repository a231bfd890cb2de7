/* decoder/lm/ZeroLM.h -- flashlight/lib/text/decoder/lm/ZeroLM.{h,cpp}. */
#pragma once
#include "flashlight/lib/text/Defines.h"
#include "flashlight/lib/text/decoder/lm/LM.h"

namespace fl {
namespace lib {
namespace text {

class FL_TEXT_API ZeroLM : public LM {
 public:
  ZeroLM() { detail::check(fltx_lm_zero_create(nullptr, &h_)); }
  ~ZeroLM() override { fltx_lm_destroy(h_); }
  ZeroLM(const ZeroLM&) = delete;
  ZeroLM& operator=(const ZeroLM&) = delete;

  LMStatePtr start(bool /* unused */) override { return std::make_shared<LMState>(); }
  std::pair<LMStatePtr, float> score(const LMStatePtr& state, const int usrTokenIdx) override {
    return std::make_pair(state->child<LMState>(usrTokenIdx), 0.0f);
  }
  std::pair<LMStatePtr, float> finish(const LMStatePtr& state) override {
    return std::make_pair(state, 0.0f);
  }
  fltx_lm* deviceHandle() const override { return h_; }

 private:
  fltx_lm* h_ = nullptr;
};

using ZeroLMPtr = std::shared_ptr<ZeroLM>;

} // namespace text
} // namespace lib
} // namespace fl
