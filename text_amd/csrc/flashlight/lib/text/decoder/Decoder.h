/*
 * decoder/Decoder.h -- the abstract Decoder API of the reference
 * (flashlight/lib/text/decoder/Decoder.h:16-74), unchanged.
 */
#pragma once
#include "flashlight/lib/text/decoder/Utils.h"

namespace fl {
namespace lib {
namespace text {

enum class CriterionType { ASG = 0, CTC = 1, S2S = 2 };

class Decoder {
 public:
  Decoder() = default;
  virtual ~Decoder() = default;

  virtual void decodeBegin() {}
  virtual void decodeStep(const float* emissions, int T, int N) = 0;
  virtual void decodeEnd() {}
  virtual std::vector<DecodeResult> decode(const float* emissions, int T, int N) {
    decodeBegin();
    decodeStep(emissions, T, N);
    decodeEnd();
    return getAllFinalHypothesis();
  }
  virtual void prune(int lookBack = 0) = 0;
  virtual int nDecodedFramesInBuffer() const = 0;
  virtual DecodeResult getBestHypothesis(int lookBack = 0) const = 0;
  virtual std::vector<DecodeResult> getAllFinalHypothesis() const = 0;
};

} // namespace text
} // namespace lib
} // namespace fl
