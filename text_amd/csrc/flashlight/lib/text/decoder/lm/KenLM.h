/*
 * decoder/lm/KenLM.h -- the KenLM adapter's interface
 * (flashlight/lib/text/decoder/lm/KenLM.{h,cpp}) backed by this repo's flat
 * back-off n-gram tables instead of libkenlm (third party, not vendored by the
 * reference and absent here).  ARPA text models only.
 */
#pragma once
#include <string>

#include "flashlight/lib/text/Defines.h"
#include "flashlight/lib/text/decoder/lm/LM.h"
#include "flashlight/lib/text/dictionary/Dictionary.h"

namespace fl {
namespace lib {
namespace text {

struct FL_TEXT_API KenLMState : LMState {
  std::vector<int32_t> ctx; /* suffix n-gram node ids (the role of lm::ngram::State) */
};

class FL_TEXT_API KenLM : public LM {
 public:
  KenLM(const std::string& path, const Dictionary& usrTknDict) {
    std::string words;
    for (size_t i = 0; i < usrTknDict.indexSize(); ++i) { /* KenLM.cpp:44-49 */
      words += usrTknDict.getEntry((int)i);
      words += '\n';
    }
    if (!words.empty()) {
      words.pop_back();
    }
    int rc = fltx_lm_arpa_load(path.c_str(), words.c_str(), &h_);
    if (rc != FLTX_OK) {
      throw std::runtime_error(std::string("[KenLM] LM loading failed: ") + fltx_last_error());
    }
    detail::check(fltx_lm_state_size(h_, &stateSize_));
  }
  ~KenLM() override { fltx_lm_destroy(h_); }
  KenLM(const KenLM&) = delete;
  KenLM& operator=(const KenLM&) = delete;

  LMStatePtr start(bool startWithNothing) override {
    auto out = std::make_shared<KenLMState>();
    out->ctx.assign(stateSize_, 0);
    detail::check(fltx_lm_start(h_, startWithNothing ? 1 : 0, out->ctx.data()));
    return out;
  }
  std::pair<LMStatePtr, float> score(const LMStatePtr& state, const int usrTokenIdx) override {
    auto in = std::static_pointer_cast<KenLMState>(state);
    std::vector<int32_t> nxt(stateSize_, 0);
    float s = 0;
    int rc = fltx_lm_step(h_, in->ctx.data(), usrTokenIdx, nxt.data(), &s);
    if (rc != FLTX_OK) { /* KenLM.cpp:66-69 */
      throw std::runtime_error("[KenLM] Invalid user token index: " + std::to_string(usrTokenIdx));
    }
    auto out = in->child<KenLMState>(usrTokenIdx);
    out->ctx = std::move(nxt);
    return std::make_pair(std::move(out), s);
  }
  std::pair<LMStatePtr, float> finish(const LMStatePtr& state) override {
    auto in = std::static_pointer_cast<KenLMState>(state);
    std::vector<int32_t> nxt(stateSize_, 0);
    float s = 0;
    detail::check(fltx_lm_step(h_, in->ctx.data(), -1, nxt.data(), &s));
    auto out = in->child<KenLMState>(-1);
    out->ctx = std::move(nxt);
    return std::make_pair(std::move(out), s);
  }
  fltx_lm* deviceHandle() const override { return h_; }

 private:
  fltx_lm* h_ = nullptr;
  int32_t stateSize_ = 0;
};

using KenLMPtr = std::shared_ptr<KenLM>;

} // namespace text
} // namespace lib
} // namespace fl
