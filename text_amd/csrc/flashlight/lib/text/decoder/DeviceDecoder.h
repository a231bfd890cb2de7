/*
 * decoder/DeviceDecoder.h -- shared implementation of the two decoder facades
 * on top of the C ABI: one stream (B = 1) for the reference's single-utterance
 * calls, plus the additive batched entry point decodeBatch().
 */
#pragma once
#include <algorithm>
#include <cstdint>
#include <thread>
#include <vector>

#include "flashlight/lib/text/decoder/Decoder.h"
#include "flashlight/lib/text/decoder/Fltx.h"
#include "flashlight/lib/text/decoder/lm/HostLM.h"

namespace fl {
namespace lib {
namespace text {
namespace detail {

/* the n-best of a whole batch where the library put it (pinned host memory, valid until the next
 * decode on the decoder): hypothesis i of utterance b has scores[(b * K + i) * 3 + {0,1,2}] and the
 * length[b] tokens at tokens + offsets[b] + i * length[b] (words likewise, null for lexicon-free) */
struct BatchView {
  int B = 0, K = 0;
  const int32_t* nHyp = nullptr;
  const int32_t* length = nullptr;
  const double* scores = nullptr;
  const int32_t* tokens = nullptr;
  const int32_t* words = nullptr;
  const int64_t* offsets = nullptr;
};

class DeviceDecoder {
 public:
  /* frames a single stream may hold between prune() calls: 2^15, or fewer for big beams -- the
   * history is sized up front (beam x frames records) and the engines index at most 2^22 / beam
   * frames; setMaxStreamFrames() overrides */
  static constexpr int kDefaultMaxStreamFrames = 1 << 15;
  static int defaultMaxStreamFrames(int beamSize) {
    const long long byBeam = (1ll << 22) / (beamSize > 0 ? beamSize : 1) - 4;
    return (int)std::max<long long>(64, std::min<long long>(kDefaultMaxStreamFrames, byBeam));
  }

  DeviceDecoder() = default;
  ~DeviceDecoder() {
    if (group_) {
      fltx_group_destroy(group_);
    }
    if (h_) {
      fltx_decoder_destroy(h_);
    }
  }
  DeviceDecoder(const DeviceDecoder&) = delete;
  DeviceDecoder& operator=(const DeviceDecoder&) = delete;

  void create(int kind, const fltx_options& opt, const fltx_trie* trie, const LMPtr& lm, int sil, int blank,
              int unk, const std::vector<float>& transitions, bool isLmToken, fltx_htrie* hostTrie = nullptr) {
    kind_ = kind;
    opt_ = opt;
    lmKeep_ = lm;
    unk_ = unk;
    transitions_ = transitions;
    isLmToken_ = isLmToken;
    hostTrie_ = hostTrie;
    if (!lm) {
      throw std::runtime_error("[decoder] the LM is null");
    }
    /* an LM without device tables (a user subclass, PyLM): the search still runs on the device, its
     * start / score / finish are answered on the host once per frame (decoder/lm/HostLM.h) */
    if (!lm->deviceHandle()) {
      bridge_ = std::make_unique<HostLmBridge>(lm);
    }
    lmHandle_ = bridge_ ? bridge_->handle() : lm->deviceHandle();
    check(fltx_decoder_create(ctx_->h, kind, &opt, trie, lmHandle_, sil, blank, unk,
                              transitions.empty() ? nullptr : transitions.data(), (int32_t)transitions.size(),
                              isLmToken ? 1 : 0, &h_));
    sil_ = sil;
    blank_ = blank;
    nTrans_ = (int)transitions.size();
    beamSize_ = opt.beam_size;
    maxFrames_ = defaultMaxStreamFrames(opt.beam_size);
  }

  fltx_ctx* ctx() const { return ctx_->h; }

  void begin() {
    pendingBegin_ = true;
    open_ = false;
    oneWithoutScores_ = false;
  }

  void step(const float* emissions, int T, int N) {
    ensureOpen(N);
    const int64_t off = 0;
    const int32_t t32 = T;
    chk(fltx_stream_step(h_, emissions, 0, &off, &t32));
  }

  void end() {
    ensureOpen(guessN());
    chk(fltx_stream_end(h_));
  }

  std::vector<DecodeResult> decodeOne(const float* emissions, int T, int N) {
    const int64_t off = 0;
    const int32_t t32 = T;
    /* getBestHypothesis(lookBack) after decode() reports an ancestor's scores (Utils.h:236-238), which needs the
     * per-frame score history -- and keeping it takes the utterance off the lane engines (4.8 ms instead of 2.0 for
     * T = 1000, beam 50).  Few callers ask: decode() runs without it, and best() decodes the utterance again WITH the
     * history from the library's own device copy of the emissions the first time somebody does (redoWithScores) */
    /* (a user LM decodes on the generic engine either way, and a second decode would call its start / score / finish
     * again -- twice the Python time, side effects of a stateful LM repeated: such a decoder keeps the history) */
    const bool keep = bridge_ != nullptr;
    check(fltx_decoder_set(h_, "keep_scores", keep ? 1 : 0));
    chk(fltx_decode_batch(h_, emissions, 0, &off, &t32, 1, N));
    open_ = true;
    pendingBegin_ = false;
    oneT_ = T;
    oneN_ = N;
    oneWithoutScores_ = !keep;
    return results(0);
  }

  /* decode the batch and leave the n-best as arrays (no per-hypothesis objects) */
  BatchView decodeBatchView(const float* emissions, const std::vector<int64_t>& offsets, const std::vector<int>& T,
                            int N, bool onDevice) {
    std::vector<int32_t> t32(T.begin(), T.end());
    oneWithoutScores_ = false;
    check(fltx_decoder_set(h_, "keep_scores", 0));
    chk(fltx_decode_batch(h_, emissions, onDevice ? 1 : 0, offsets.empty() ? nullptr : offsets.data(),
                            t32.data(), (int32_t)t32.size(), N));
    open_ = true;
    pendingBegin_ = false;
    /* the whole n-best crosses PCIe once (pinned staging inside the library) */
    BatchView v;
    v.B = (int)T.size();
    v.K = beamSize_;
    check(fltx_result_fetch_batch(h_, &v.nHyp, &v.length, &v.scores, &v.tokens, &v.words, &v.offsets));
    return v;
  }

  /* DecodeResult objects of utterance b of a view */
  std::vector<DecodeResult> materialise(const BatchView& v, int b) const {
    std::vector<DecodeResult> out;
    fill(out, b, v.nHyp, v.length, v.scores, v.tokens, v.words, v.offsets);
    return out;
  }

  std::vector<std::vector<DecodeResult>> decodeBatch(const float* emissions, const std::vector<int64_t>& offsets,
                                                     const std::vector<int>& T, int N, bool onDevice) {
    const BatchView v = decodeBatchView(emissions, offsets, T, N, onDevice);
    std::vector<std::vector<DecodeResult>> out(T.size());
    /* 12 800 DecodeResult objects of two 1 002-entry vectors each for a C2 batch (100 MB of rows): filled by up to
     * eight host threads, utterances interleaved */
    const size_t nThreads = std::min<size_t>({(size_t)8, (size_t)std::max(1u, std::thread::hardware_concurrency()), T.size() / 16 + 1});
    auto work = [&](size_t first) {
      for (size_t b = first; b < T.size(); b += nThreads) {
        fill(out[b], (int)b, v.nHyp, v.length, v.scores, v.tokens, v.words, v.offsets);
      }
    };
    if (nThreads <= 1) {
      work(0);
    } else {
      std::vector<std::thread> pool;
      for (size_t i = 1; i < nThreads; ++i) {
        pool.emplace_back(work, i);
      }
      work(0);
      for (auto& th : pool) {
        th.join();
      }
    }
    return out;
  }

  /* The same batch over several devices: one context, decoder and host
   * thread per entry of `devices` (an index may repeat), trie and LM tables replicated, the
   * utterances cut into contiguous shards of about equal frame count, results in input order.
   * `emissions` is host memory. */
  std::vector<std::vector<DecodeResult>> decodeBatchOn(const std::vector<int>& devices, const float* emissions,
                                                       const std::vector<int64_t>& offsets,
                                                       const std::vector<int>& T, int N) {
    if (devices.empty()) {
      throw std::invalid_argument("[decoder] decodeBatch: empty device list");
    }
    if (!group_ || devices != groupDevices_) {
      if (group_) {
        fltx_group_destroy(group_);
        group_ = nullptr;
      }
      std::vector<int32_t> dv(devices.begin(), devices.end());
      check(fltx_group_create(dv.data(), (int32_t)dv.size(), kind_, &opt_, hostTrie_, lmHandle_, sil_,
                              blank_, unk_, transitions_.empty() ? nullptr : transitions_.data(),
                              (int32_t)transitions_.size(), isLmToken_ ? 1 : 0, &group_));
      groupDevices_ = devices;
    }
    std::vector<int32_t> t32(T.begin(), T.end());
    std::vector<const float*> ptrs(devices.size(), emissions);
    chk(fltx_group_decode_batch(group_, ptrs.data(), nullptr, offsets.empty() ? nullptr : offsets.data(),
                                  t32.data(), (int32_t)t32.size(), N));
    std::vector<std::vector<DecodeResult>> out(T.size());
    for (size_t i = 0; i < devices.size(); ++i) {
      fltx_decoder* part = nullptr;
      int32_t first = 0, count = 0;
      check(fltx_group_decoder(group_, (int32_t)i, &part, &first, &count));
      if (count == 0) {
        continue;
      }
      const int32_t *nHyp = nullptr, *len = nullptr, *tok = nullptr, *wrd = nullptr;
      const double* sc = nullptr;
      const int64_t* off = nullptr;
      check(fltx_result_fetch_batch(part, &nHyp, &len, &sc, &tok, &wrd, &off));
      for (int k = 0; k < count; ++k) {
        fill(out[(size_t)(first + k)], k, nHyp, len, sc, tok, wrd, off);
      }
    }
    return out;
  }

  void prune(int lookBack) {
    if (!open_) {
      return;
    }
    if (oneWithoutScores_) {
      redoWithScores(); /* (prune() after decode(): as before, on the utterance decoded with its score history) */
    }
    chk(fltx_stream_prune(h_, lookBack));
  }

  int framesInBuffer() const {
    if (!open_) {
      return pendingBegin_ ? 1 : 0;
    }
    int32_t n = 0;
    check(fltx_stream_frames_in_buffer(h_, 0, &n));
    return n;
  }

  int nHypothesis() const {
    if (!open_) {
      return pendingBegin_ ? 1 : 0;
    }
    int32_t n = 0, len = 0;
    check(fltx_result_count(h_, 0, &n, &len));
    return n;
  }

  DecodeResult best(int lookBack) const {
    if (!open_) {
      return DecodeResult();
    }
    if (oneWithoutScores_) {
      redoWithScores();
    }
    int32_t n = 0, len = 0;
    check(fltx_result_count(h_, 0, &n, &len));
    std::vector<int32_t> tok(len), wrd(len);
    double sc[3] = {0, 0, 0};
    int32_t got = 0;
    check(fltx_result_best(h_, 0, lookBack, sc, tok.data(), wrd.data(), len, &got));
    DecodeResult r(got);
    if (got > 0) {
      r.score = sc[0];
      r.emittingModelScore = sc[1];
      r.lmScore = sc[2];
      std::copy(tok.begin(), tok.begin() + got, r.tokens.begin());
      std::copy(wrd.begin(), wrd.begin() + got, r.words.begin());
    }
    return r;
  }

  std::vector<DecodeResult> results(int b) const {
    std::vector<DecodeResult> out;
    if (!open_) {
      return out;
    }
    int32_t n = 0, len = 0;
    check(fltx_result_count(h_, b, &n, &len));
    if (n <= 0) {
      return out;
    }
    std::vector<double> sc(3 * (size_t)n);
    std::vector<int32_t> tok((size_t)n * len), wrd((size_t)n * len);
    int32_t got = 0;
    check(fltx_result_fetch(h_, b, n, sc.data(), tok.data(), wrd.data(), &got));
    out.reserve(got);
    for (int i = 0; i < got; ++i) {
      DecodeResult r(len);
      r.score = sc[3 * i];
      r.emittingModelScore = sc[3 * i + 1];
      r.lmScore = sc[3 * i + 2];
      std::copy(tok.begin() + (size_t)i * len, tok.begin() + (size_t)(i + 1) * len, r.tokens.begin());
      std::copy(wrd.begin() + (size_t)i * len, wrd.begin() + (size_t)(i + 1) * len, r.words.begin());
      out.push_back(std::move(r));
    }
    return out;
  }

  void setMaxStreamFrames(int n) { maxFrames_ = n; }
  /* a user LM's states held on the host for the single-utterance stream (0 for LMs with device tables) */
  size_t hostLmStates() const { return bridge_ ? bridge_->liveStates(0) : 0; }

 private:
  /* like check(); a failure inside a user LM's start / score / finish surfaces as what the LM threw */
  void chk(int rc) const {
    if (rc == FLTX_ERR_CALLBACK && bridge_) {
      bridge_->rethrow();
    }
    check(rc);
  }
  std::unique_ptr<HostLmBridge> bridge_;
  fltx_lm* lmHandle_ = nullptr;
  /* n-best of utterance b of a fetched batch -> DecodeResult objects */
  void fill(std::vector<DecodeResult>& dst, int b, const int32_t* nHyp, const int32_t* len, const double* sc,
            const int32_t* tok, const int32_t* wrd, const int64_t* off) const {
    const int n = nHyp[b], L = len[b];
    dst.reserve((size_t)n);
    for (int i = 0; i < n; ++i) {
      DecodeResult r(L);
      const double* s3 = sc + ((size_t)b * beamSize_ + (size_t)i) * 3;
      r.score = s3[0];
      r.emittingModelScore = s3[1];
      r.lmScore = s3[2];
      const int32_t* tp = tok + off[b] + (int64_t)i * L;
      std::copy(tp, tp + L, r.tokens.begin());
      if (wrd) {
        const int32_t* wp = wrd + off[b] + (int64_t)i * L;
        std::copy(wp, wp + L, r.words.begin());
      } /* else: DecodeResult(L) leaves words at -1 (LexiconFreeDecoder.h:80-82) */
      dst.push_back(std::move(r));
    }
  }
  int beamSize_ = 0;
  int kind_ = 0, unk_ = -1;
  fltx_options opt_{};
  LMPtr lmKeep_;
  std::vector<float> transitions_;
  bool isLmToken_ = false;
  fltx_htrie* hostTrie_ = nullptr;
  fltx_group* group_ = nullptr;
  std::vector<int> groupDevices_;
  int guessN() const {
    if (nTrans_ > 0) {
      int n = 1;
      while (n * n < nTrans_) {
        ++n;
      }
      return n;
    }
    return std::max(sil_, blank_) + 1;
  }
  /* decode() ran without the per-frame score history (decodeOne): decode the utterance again with it, from the
   * device copy of the emissions the library still holds -- the n-best is the same, getBestHypothesis(lookBack) and the
   * calls behind it find what they need */
  void redoWithScores() const {
    int64_t staged = 0;
    check(fltx_decoder_get(h_, "staged_emissions", &staged));
    if (!staged) {
      throw std::runtime_error("[decoder] getBestHypothesis: the utterance's emissions are no longer on the device");
    }
    const int64_t off = 0;
    const int32_t t32 = oneT_;
    check(fltx_decoder_set(h_, "keep_scores", 1));
    chk(fltx_decode_batch(h_, (const float*)(uintptr_t)staged, 1, &off, &t32, 1, oneN_));
    oneWithoutScores_ = false;
  }
  void ensureOpen(int N) {
    oneWithoutScores_ = false;
    if (pendingBegin_ || !open_) {
      chk(fltx_stream_begin(h_, 1, N, maxFrames_));
      pendingBegin_ = false;
      open_ = true;
    }
  }

  std::shared_ptr<Context> ctx_ = Context::get();
  fltx_decoder* h_ = nullptr;
  bool pendingBegin_ = false, open_ = false;
  mutable bool oneWithoutScores_ = false; /* the last call was decode() and the score history has not been asked for yet */
  int oneT_ = 0, oneN_ = 0;
  int maxFrames_ = kDefaultMaxStreamFrames;
  int sil_ = 0, blank_ = 0, nTrans_ = 0;
};

} // namespace detail
} // namespace text
} // namespace lib
} // namespace fl
