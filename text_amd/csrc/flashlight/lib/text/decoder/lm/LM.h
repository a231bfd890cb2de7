/*
 * decoder/lm/LM.h -- LMState / LM with the reference's interface
 * (flashlight/lib/text/decoder/lm/LM.h:21-87).
 *
 * Host-side objects: they serve user code that scores words outside the
 * decoder (e.g. trie label scores, DecoderTest.cpp:137-146).  During decoding
 * ZeroLM and KenLM are evaluated on the device from flat tables (deviceHandle());
 * any other subclass -- a user's own LM, C++ or Python -- keeps its start /
 * score / finish on the host: the beam search still runs in the kernels and
 * asks the LM once per frame about the frame's distinct (state, index) pairs
 * (decoder/lm/HostLM.h).  LM::score must therefore be a function of its
 * arguments, which LMState::child's memo already makes it in the reference.
 */
#pragma once
#include <memory>
#include <stdexcept>
#include <unordered_map>
#include <utility>
#include <vector>

#include "flashlight/lib/text/decoder/Fltx.h"

namespace fl {
namespace lib {
namespace text {

struct LMState {
  std::unordered_map<int, std::shared_ptr<LMState>> children;

  template <typename T>
  std::shared_ptr<T> child(int usrIdx) {
    auto s = children.find(usrIdx);
    if (s == children.end()) {
      auto state = std::make_shared<T>();
      children[usrIdx] = state;
      return state;
    }
    return std::static_pointer_cast<T>(s->second);
  }

  int compare(const std::shared_ptr<LMState>& state) const {
    LMState* inState = state.get();
    if (!state) {
      throw std::runtime_error("a state is null");
    }
    return this == inState ? 0 : (this < inState ? -1 : 1);
  }
};

using LMStatePtr = std::shared_ptr<LMState>;

class LM {
 public:
  virtual LMStatePtr start(bool startWithNothing) = 0;
  virtual std::pair<LMStatePtr, float> score(const LMStatePtr& state, const int usrTokenIdx) = 0;
  virtual std::pair<LMStatePtr, float> finish(const LMStatePtr& state) = 0;
  virtual void updateCache(std::vector<LMStatePtr>) {}
  virtual ~LM() = default;

  /* additive: the device tables of this LM, or nullptr when it has none */
  virtual fltx_lm* deviceHandle() const { return nullptr; }

 protected:
  std::vector<int> usrToLmIdxMap_;
};

using LMPtr = std::shared_ptr<LM>;

} // namespace text
} // namespace lib
} // namespace fl
