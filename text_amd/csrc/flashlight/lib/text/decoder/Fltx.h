/*
 * decoder/Fltx.h -- glue between the fl::lib::text C++ facade and the C ABI
 * (include/fltx.h): status -> exception mapping and the shared device context.
 *
 * The reference reports errors as C++ exceptions: std::runtime_error
 * (lm/LM.h:40, lm/KenLM.cpp:36,40,67), std::out_of_range (Trie.cpp:32,54),
 * std::invalid_argument (dictionary/Dictionary.cpp:64).  The C ABI returns
 * status codes; this header turns them back into the same exception types.
 */
#pragma once
#include <memory>
#include <stdexcept>
#include <string>

#include "fltx.h"

namespace fl {
namespace lib {
namespace text {
namespace detail {

inline void check(int rc) {
  if (rc == FLTX_OK) {
    return;
  }
  const std::string msg = fltx_last_error();
  switch (rc) {
    case FLTX_ERR_INVALID:
      throw std::invalid_argument(msg);
    case FLTX_ERR_RANGE:
      throw std::out_of_range(msg);
    default:
      throw std::runtime_error(msg);
  }
}

/* one context (HIP device + stream) shared by the decoders a THREAD creates; the device is the current HIP device
 * at first use.  The reference's safe pattern is one decoder per thread over a shared read-only Trie and LM
 * (Utils.h:60-62): with a context per thread those decoders run on streams of their own, so several utterances --
 * one per thread -- are on the device at a time. */
struct Context {
  fltx_ctx* h = nullptr;
  Context() { check(fltx_ctx_create(-1, nullptr, &h)); }
  ~Context() { fltx_ctx_destroy(h); }
  static std::shared_ptr<Context> get() {
    static thread_local std::weak_ptr<Context> cached;
    auto sp = cached.lock();
    if (!sp) {
      sp = std::make_shared<Context>();
      cached = sp;
    }
    return sp;
  }
};

} // namespace detail
} // namespace text
} // namespace lib
} // namespace fl
