/*
 * decoder/LexiconFreeDecoder.h -- LexiconFreeDecoder with the reference's
 * interface (flashlight/lib/text/decoder/LexiconFreeDecoder.h:20-28,102-138),
 * running on the MI355X kernels.  decodeBatch() is additive: B independent
 * utterances in one launch (the reference has no batch API).
 */
#pragma once
#include "flashlight/lib/text/Defines.h"
#include "flashlight/lib/text/decoder/DeviceDecoder.h"

namespace fl {
namespace lib {
namespace text {

struct LexiconFreeDecoderOptions {
  int beamSize;
  int beamSizeToken;
  double beamThreshold;
  double lmWeight;
  double silScore;
  bool logAdd;
  CriterionType criterionType;
};

class FL_TEXT_API LexiconFreeDecoder : public Decoder {
 public:
  LexiconFreeDecoder(LexiconFreeDecoderOptions opt, const LMPtr& lm, const int sil, const int blank,
                     const std::vector<float>& transitions)
      : opt_(std::move(opt)), lm_(lm), transitions_(transitions), sil_(sil), blank_(blank) {
    if (opt_.criterionType == CriterionType::S2S) {
      throw std::runtime_error("[LexiconFreeDecoder] S2S criterion is not supported");
    }
    fltx_options o{opt_.beamSize, opt_.beamSizeToken, opt_.beamThreshold, opt_.lmWeight, 0.0, 0.0,
                   opt_.silScore, opt_.logAdd ? 1 : 0, (int32_t)opt_.criterionType};
    dev_.create(FLTX_DECODER_LEXFREE, o, nullptr, lm_, sil_, blank_, -1, transitions_, false);
  }

  void decodeBegin() override { dev_.begin(); }
  void decodeStep(const float* emissions, int T, int N) override { dev_.step(emissions, T, N); }
  void decodeEnd() override { dev_.end(); }
  std::vector<DecodeResult> decode(const float* emissions, int T, int N) override {
    return dev_.decodeOne(emissions, T, N);
  }
  int nHypothesis() const { return dev_.nHypothesis(); }
  void prune(int lookBack = 0) override { dev_.prune(lookBack); }
  int nDecodedFramesInBuffer() const override { return dev_.framesInBuffer(); }
  DecodeResult getBestHypothesis(int lookBack = 0) const override { return dev_.best(lookBack); }
  std::vector<DecodeResult> getAllFinalHypothesis() const override { return dev_.results(0); }

  const LMPtr& getLMPtr() const { return lm_; }
  int getSilIdx() const { return sil_; }
  int getBlankIdx() const { return blank_; }
  const LexiconFreeDecoderOptions& getOptions() const { return opt_; }
  const std::vector<float>& getTransitions() const { return transitions_; }

  /* additive: utterance b reads T[b]*N floats at emissions + offsets[b]
   * (offsets empty = packed back to back); onDevice = emissions is HBM resident */
  std::vector<std::vector<DecodeResult>> decodeBatch(const float* emissions, const std::vector<int>& T, int N,
                                                     const std::vector<int64_t>& offsets = {},
                                                     bool onDevice = false) {
    return dev_.decodeBatch(emissions, packed(offsets, T, N), T, N, onDevice);
  }
  /* additive: the same, leaving the n-best as arrays in pinned host memory (valid until the next
   * decode); materialise(view, b) builds the DecodeResult objects of one utterance on demand */
  detail::BatchView decodeBatchView(const float* emissions, const std::vector<int>& T, int N,
                                    const std::vector<int64_t>& offsets = {}, bool onDevice = false) {
    return dev_.decodeBatchView(emissions, packed(offsets, T, N), T, N, onDevice);
  }
  std::vector<DecodeResult> materialise(const detail::BatchView& v, int b) const { return dev_.materialise(v, b); }
  /* additive: frames one stream may buffer between prune() calls (decodeStep path) */
  void setMaxStreamFrames(int n) { dev_.setMaxStreamFrames(n); }
  /* additive: LM states a user-defined LM (no device tables) has alive for this decoder's stream; prune() bounds it */
  size_t hostLmStates() const { return dev_.hostLmStates(); }
  /* additive: the batch sharded over `devices` (one host thread + stream per device, results in
   * input order) */
  std::vector<std::vector<DecodeResult>> decodeBatch(const float* emissions, const std::vector<int>& T, int N,
                                                     const std::vector<int>& devices,
                                                     const std::vector<int64_t>& offsets = {}) {
    return dev_.decodeBatchOn(devices, emissions, offsets, T, N);
  }

 protected:
  static std::vector<int64_t> packed(const std::vector<int64_t>& offsets, const std::vector<int>& T, int N) {
    if (!offsets.empty()) {
      return offsets;
    }
    std::vector<int64_t> o(T.size(), 0);
    for (size_t b = 1; b < T.size(); ++b) {
      o[b] = o[b - 1] + (int64_t)T[b - 1] * N;
    }
    return o;
  }
  LexiconFreeDecoderOptions opt_;
  LMPtr lm_;
  std::vector<float> transitions_;
  int sil_;
  int blank_;
  detail::DeviceDecoder dev_;
};

} // namespace text
} // namespace lib
} // namespace fl
