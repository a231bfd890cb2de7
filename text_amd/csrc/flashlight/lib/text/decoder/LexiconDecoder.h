/*
 * decoder/LexiconDecoder.h -- LexiconDecoder with the reference's interface
 * (flashlight/lib/text/decoder/LexiconDecoder.h:21-31,117-156), running on the
 * MI355X kernels.  The trie is flattened into HBM at construction; later
 * mutation of the Trie object is picked up only by a new decoder.
 */
#pragma once
#include "flashlight/lib/text/Defines.h"
#include "flashlight/lib/text/decoder/DeviceDecoder.h"
#include "flashlight/lib/text/decoder/Trie.h"

namespace fl {
namespace lib {
namespace text {

struct LexiconDecoderOptions {
  int beamSize;
  int beamSizeToken;
  double beamThreshold;
  double lmWeight;
  double wordScore;
  double unkScore;
  double silScore;
  bool logAdd;
  CriterionType criterionType;
};

class FL_TEXT_API LexiconDecoder : public Decoder {
 public:
  LexiconDecoder(LexiconDecoderOptions opt, const TriePtr& lexicon, const LMPtr& lm, const int sil,
                 const int blank, const int unk, const std::vector<float>& transitions, const bool isLmToken)
      : opt_(std::move(opt)), lexicon_(lexicon), lm_(lm), sil_(sil), blank_(blank), unk_(unk),
        transitions_(transitions), isLmToken_(isLmToken) {
    if (opt_.criterionType == CriterionType::S2S) {
      throw std::runtime_error("[LexiconDecoder] S2S criterion is not supported");
    }
    if (!lexicon_) {
      throw std::invalid_argument("[LexiconDecoder] null lexicon");
    }
    fltx_options o{opt_.beamSize, opt_.beamSizeToken, opt_.beamThreshold, opt_.lmWeight, opt_.wordScore,
                   opt_.unkScore, opt_.silScore, opt_.logAdd ? 1 : 0, (int32_t)opt_.criterionType};
    devTrie_ = lexicon_->deviceHandle(dev_.ctx()); /* this decoder's reference keeps the upload alive */
    dev_.create(FLTX_DECODER_LEXICON, o, devTrie_.get(), lm_, sil_, blank_, unk_, transitions_, isLmToken_,
                lexicon_->hostHandle());
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

  std::vector<std::vector<DecodeResult>> decodeBatch(const float* emissions, const std::vector<int>& T, int N,
                                                     const std::vector<int64_t>& offsets = {},
                                                     bool onDevice = false) {
    std::vector<int64_t> o = offsets;
    if (o.empty()) {
      o.assign(T.size(), 0);
      for (size_t b = 1; b < T.size(); ++b) {
        o[b] = o[b - 1] + (int64_t)T[b - 1] * N;
      }
    }
    return dev_.decodeBatch(emissions, o, T, N, onDevice);
  }
  /* additive: the same, leaving the n-best as arrays in pinned host memory (valid until the next
   * decode); materialise(view, b) builds the DecodeResult objects of one utterance on demand */
  detail::BatchView decodeBatchView(const float* emissions, const std::vector<int>& T, int N,
                                    const std::vector<int64_t>& offsets = {}, bool onDevice = false) {
    return dev_.decodeBatchView(emissions, packedOffsets(offsets, T, N), T, N, onDevice);
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
  static std::vector<int64_t> packedOffsets(const std::vector<int64_t>& offsets, const std::vector<int>& T, int N) {
    if (!offsets.empty()) {
      return offsets;
    }
    std::vector<int64_t> o(T.size(), 0);
    for (size_t b = 1; b < T.size(); ++b) {
      o[b] = o[b - 1] + (int64_t)T[b - 1] * N;
    }
    return o;
  }
  LexiconDecoderOptions opt_;
  TriePtr lexicon_;
  LMPtr lm_;
  int sil_;
  int blank_;
  int unk_;
  std::vector<float> transitions_;
  bool isLmToken_;
  std::shared_ptr<const fltx_trie> devTrie_;
  detail::DeviceDecoder dev_;
};

} // namespace text
} // namespace lib
} // namespace fl
