/* decoder/Utils.h -- DecodeResult (flashlight/lib/text/decoder/Utils.h:26-39). */
#pragma once
#include <limits>
#include <vector>

#include "flashlight/lib/text/decoder/lm/LM.h"

namespace fl {
namespace lib {
namespace text {

const double kNegativeInfinity = -std::numeric_limits<double>::infinity();
const int kLookBackLimit = 100;

struct DecodeResult {
  double score;
  double emittingModelScore;
  double lmScore;
  std::vector<int> words;
  std::vector<int> tokens;

  explicit DecodeResult(int length = 0)
      : score(0), emittingModelScore(0), lmScore(0), words(length, -1), tokens(length, -1) {}
};

} // namespace text
} // namespace lib
} // namespace fl
