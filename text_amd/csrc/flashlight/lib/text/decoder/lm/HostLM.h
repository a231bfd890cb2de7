/*
 * decoder/lm/HostLM.h -- a user-defined LM (any subclass of LM without device tables, e.g. the Python trampoline
 * PyLM: reference bindings/python/flashlight/lib/text/_decoder.cpp:39-56, interface decoder/lm/LM.h:61-85) behind
 * the C ABI's host-LM callbacks (include/fltx.h, fltx_lm_host_create).
 *
 * The beam search stays in the HIP kernels.  Once per frame the library asks this bridge about the DISTINCT
 * (LM state, index) pairs of the whole frame; the bridge calls the user's LM::score / LM::finish and names the
 * LMState objects it gets back by small integers -- the same integer for the same OBJECT (the reference merges
 * hypotheses on the state's address, lm/LM.h:37-49), a fresh one otherwise.  The states are kept alive here for as
 * long as a hypothesis can hold them: until decodeBegin, or until prune() reports that the beam no longer does
 * (in the reference the pruned hypotheses' shared_ptrs go away, decoder/Utils.h:312-342).
 *
 * Exceptions thrown by the user's LM (C++ or Python) are parked, the decode call fails with FLTX_ERR_CALLBACK and
 * the facade rethrows the original exception.
 */
#pragma once
#include <exception>
#include <unordered_map>
#include <vector>

#include "flashlight/lib/text/decoder/lm/LM.h"

namespace fl {
namespace lib {
namespace text {
namespace detail {

class HostLmBridge {
 public:
  explicit HostLmBridge(LMPtr lm) : lm_(std::move(lm)) {
    fltx_host_lm cb;
    cb.user = this;
    cb.start = &HostLmBridge::startCb;
    cb.score = &HostLmBridge::scoreCb;
    cb.update_cache = &HostLmBridge::cacheCb;
    cb.retain = &HostLmBridge::retainCb;
    check(fltx_lm_host_create(&cb, &h_));
  }
  ~HostLmBridge() { fltx_lm_destroy(h_); }
  HostLmBridge(const HostLmBridge&) = delete;
  HostLmBridge& operator=(const HostLmBridge&) = delete;

  fltx_lm* handle() const { return h_; }
  /* LM states held for utterance b (tests: prune() bounds it) */
  size_t liveStates(int b) const { return b < (int)utt_.size() ? utt_[(size_t)b].states.size() : 0; }
  /* the exception the user's LM threw inside a callback, if any */
  void rethrow() {
    if (error_) {
      std::exception_ptr e = error_;
      error_ = nullptr;
      std::rethrow_exception(e);
    }
  }

 private:
  struct Utt {
    std::unordered_map<int32_t, LMStatePtr> states;
    std::unordered_map<const LMState*, int32_t> ids;
    int32_t next = 1;
  };

  const LMStatePtr& stateOf(int b, int32_t id) const {
    const auto& m = utt_.at((size_t)b).states;
    auto it = m.find(id);
    if (it == m.end()) {
      throw std::runtime_error("[HostLM] unknown LM state id");
    }
    return it->second;
  }
  int32_t idOf(int b, const LMStatePtr& s) {
    if (!s) {
      throw std::runtime_error("a state is null"); /* LMState::compare, lm/LM.h:38-41 */
    }
    Utt& u = utt_[(size_t)b];
    auto it = u.ids.find(s.get());
    if (it != u.ids.end()) {
      return it->second;
    }
    const int32_t id = u.next++;
    u.ids.emplace(s.get(), id);
    u.states.emplace(id, s);
    return id;
  }

  template <class F>
  int32_t guarded(F&& f) noexcept {
    try {
      f();
      return 0;
    } catch (...) {
      error_ = std::current_exception();
      return 1;
    }
  }

  static int32_t startCb(void* user, int32_t nUtt) {
    auto* self = static_cast<HostLmBridge*>(user);
    return self->guarded([&] {
      self->utt_.assign((size_t)nUtt, Utt());
      for (int b = 0; b < nUtt; ++b) { /* lm_->start(0), LexiconFreeDecoder.cpp:24 / LexiconDecoder.cpp:24 */
        LMStatePtr s0 = self->lm_->start(false);
        if (!s0) {
          throw std::runtime_error("a state is null");
        }
        Utt& u = self->utt_[(size_t)b];
        u.ids.emplace(s0.get(), 0);
        u.states.emplace(0, std::move(s0));
      }
    });
  }
  static int32_t scoreCb(void* user, int32_t n, const int32_t* utt, const int32_t* state, const int32_t* idx,
                         int32_t* outState, float* outScore) {
    auto* self = static_cast<HostLmBridge*>(user);
    return self->guarded([&] {
      for (int32_t i = 0; i < n; ++i) {
        const LMStatePtr in = self->stateOf(utt[i], state[i]); /* (a copy: the maps may rehash below) */
        const std::pair<LMStatePtr, float> r = idx[i] < 0 ? self->lm_->finish(in) : self->lm_->score(in, idx[i]);
        outState[i] = self->idOf(utt[i], r.first);
        outScore[i] = r.second;
      }
    });
  }
  static int32_t cacheCb(void* user, int32_t b, int32_t n, const int32_t* states) {
    auto* self = static_cast<HostLmBridge*>(user);
    return self->guarded([&] { /* updateLMCache, decoder/Utils.h:346-354 */
      std::vector<LMStatePtr> v;
      v.reserve((size_t)n);
      for (int32_t i = 0; i < n; ++i) {
        v.push_back(self->stateOf(b, states[i]));
      }
      self->lm_->updateCache(std::move(v));
    });
  }
  static int32_t retainCb(void* user, int32_t b, int32_t n, const int32_t* states) {
    auto* self = static_cast<HostLmBridge*>(user);
    return self->guarded([&] {
      Utt& u = self->utt_.at((size_t)b);
      Utt kept;
      kept.next = u.next;
      for (int32_t i = 0; i < n; ++i) {
        auto it = u.states.find(states[i]);
        if (it != u.states.end() && !kept.states.count(states[i])) {
          kept.ids.emplace(it->second.get(), states[i]);
          kept.states.emplace(states[i], it->second);
        }
      }
      u = std::move(kept);
    });
  }

  LMPtr lm_;
  fltx_lm* h_ = nullptr;
  std::vector<Utt> utt_;
  std::exception_ptr error_;
};

} // namespace detail
} // namespace text
} // namespace lib
} // namespace fl
