/*
 * oracle/oracle.cpp -- CPU RESTATEMENT of the flashlight/text beam-search
 * decoders (LexiconFreeDecoder, LexiconDecoder, Trie, LM state identity,
 * candidate merge / prune, back-trace, streaming prune).
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the checker for the HIP path in
 * text_amd/csrc; it is never linked into, loaded by, or called from the
 * product (see oracle/orc_api.h).  It is written from the behaviour of the
 * reference, not copied from it: hypotheses live in per-frame arrays and point
 * to their parent by slot index, LM states are integer ids in a per-decoder
 * arena keyed by (parent id, edge), the trie is an index arena.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this restatement
 * against (a) golden vectors produced by the unmodified reference compiled in
 * the dev container (oracle/_ref, tests/golden/make_golden.py) and (b) the
 * known answers of the reference's own DecoderTest.cpp.
 *
 * Build: g++ -O2 -ffp-contract=off (no -march=native: the reference is built
 * without FMA contraction; score chains must round identically).
 *
 * Each function cites the reference lines (relative to
 * /root/reference/flashlight/lib/text/) it restates.
 */
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <numeric>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

#include "arpa_lm.h"
#include "orc_api.h"

namespace orc {

static const double kNegInf = -std::numeric_limits<double>::infinity();
static const int kLookBackLimit = 100; // decoder/Utils.h:28

/* ------------------------------------------------------------------------ */
/* LM models.  The reference couples model and state trie (lm/LM.h:21-85);   */
/* here the model is a pure function and the per-utterance state trie lives  */
/* in the decoder (StateArena) so one model can serve many decoders.         */
/* ------------------------------------------------------------------------ */
struct LMModel {
  virtual ~LMModel() = default;
  /* true: finish() moves to child(state, -1) (lm/KenLM.cpp:77-83);
   * false: finish() returns the state itself (lm/ZeroLM.cpp:24-26). */
  virtual bool finishMakesChild() const = 0;
  virtual void startCtx(bool nothing, std::vector<int32_t>& ctx) const = 0;
  virtual float scoreCtx(const std::vector<int32_t>& ctx, int usr,
                         std::vector<int32_t>& out) const = 0;
  virtual float finishCtx(const std::vector<int32_t>& ctx,
                          std::vector<int32_t>& out) const = 0;
  /* true: the LM hands out ONE state object per input, whatever the history (a user LM whose state is its last
   * input): child(state, key) is then the same node for every `state` -- the reference compares LM states by
   * address (lm/LM.h:37-49), so hypotheses with different histories merge. */
  virtual bool sharesStatesByInput() const { return false; }
};

/* lm/ZeroLM.cpp:14-26 */
struct ZeroModel : LMModel {
  bool finishMakesChild() const override { return false; }
  void startCtx(bool, std::vector<int32_t>& ctx) const override { ctx.clear(); }
  float scoreCtx(const std::vector<int32_t>&, int,
                 std::vector<int32_t>& out) const override {
    out.clear();
    return 0.0f;
  }
  float finishCtx(const std::vector<int32_t>&,
                  std::vector<int32_t>& out) const override {
    out.clear();
    return 0.0f;
  }
};

/* lm/KenLM.cpp:32-83 with oracle/arpa_lm.h standing in for libkenlm */
struct ArpaLM : LMModel {
  ArpaModel model;
  std::vector<int32_t> usrToLm; // KenLM.cpp:44-49
  bool finishMakesChild() const override { return true; }
  void startCtx(bool nothing, std::vector<int32_t>& ctx) const override {
    ctx.clear();
    if (!nothing) {
      ctx.push_back(model.bos); // BeginSentenceWrite, KenLM.cpp:57
    }
  }
  float scoreCtx(const std::vector<int32_t>& ctx, int usr,
                 std::vector<int32_t>& out) const override {
    if (usr < 0 || usr >= (int)usrToLm.size()) {
      throw std::runtime_error("[ArpaLM] Invalid user token index"); // KenLM.cpp:66-69
    }
    return model.score(ctx, usrToLm[usr], out);
  }
  float finishCtx(const std::vector<int32_t>& ctx,
                  std::vector<int32_t>& out) const override {
    return model.score(ctx, model.eos, out); // KenLM.cpp:80-81
  }
};

/* orc_api.h lm_lastword_create: score = integer hash of (previous input, input, seed) / 2^13 */
static float pairScore(int prev, int w, int seed) {
  uint32_t h = (uint32_t)prev * 1000003u + (uint32_t)w * 7919u + (uint32_t)seed;
  h ^= h >> 13;
  h *= 0x5BD1E995u;
  h ^= h >> 15;
  return -(float)(h & 0xFFFFu) / 8192.0f;
}
struct LastWordModel : LMModel {
  int seed = 0;
  bool finishMakesChild() const override { return false; }
  bool sharesStatesByInput() const override { return true; }
  void startCtx(bool, std::vector<int32_t>& ctx) const override { ctx.assign(1, -1); }
  float scoreCtx(const std::vector<int32_t>& ctx, int usr, std::vector<int32_t>& out) const override {
    out.assign(1, usr);
    return pairScore(ctx.at(0) + 2, usr + 2, seed);
  }
  float finishCtx(const std::vector<int32_t>& ctx, std::vector<int32_t>& out) const override {
    out = ctx;
    return pairScore(ctx.at(0) + 2, 1, seed);
  }
};

/* LMState trie (lm/LM.h:21-50): child(parent, key) is memoised, identity is
 * the node (here: the id). */
struct StateArena {
  std::unordered_map<uint64_t, int32_t> kids;
  std::vector<std::vector<int32_t>> ctx;
  void clear() {
    kids.clear();
    ctx.clear();
  }
  int32_t root(const LMModel& lm, bool nothing) {
    byInput = lm.sharesStatesByInput();
    ctx.emplace_back();
    lm.startCtx(nothing, ctx.back());
    return (int32_t)ctx.size() - 1;
  }
  /* LM.h:24-34 */
  bool byInput = false; /* LMModel::sharesStatesByInput */
  int32_t child(int32_t s, int32_t key, bool& fresh) {
    uint64_t k = ((uint64_t)(byInput ? 0xFFFFFFFFu : (uint32_t)s) << 32) | (uint32_t)key;
    auto it = kids.find(k);
    if (it != kids.end()) {
      fresh = false;
      return it->second;
    }
    fresh = true;
    ctx.emplace_back();
    int32_t id = (int32_t)ctx.size() - 1;
    kids.emplace(k, id);
    return id;
  }
  std::pair<int32_t, float> score(const LMModel& lm, int32_t s, int key) {
    bool fresh;
    std::vector<int32_t> out;
    float sc = lm.scoreCtx(ctx[s], key, out);
    int32_t c = child(s, key, fresh);
    if (fresh) {
      ctx[c] = std::move(out);
    }
    return {c, sc};
  }
  std::pair<int32_t, float> finish(const LMModel& lm, int32_t s) {
    std::vector<int32_t> out;
    float sc = lm.finishCtx(ctx[s], out);
    if (!lm.finishMakesChild()) {
      return {s, sc};
    }
    bool fresh;
    int32_t c = child(s, -1, fresh);
    if (fresh) {
      ctx[c] = std::move(out);
    }
    return {c, sc};
  }
};

/* ------------------------------------------------------------------------ */
/* Trie (decoder/Trie.h:30-92, Trie.cpp:20-101)                              */
/* ------------------------------------------------------------------------ */
static const int kTrieMaxLabel = 6; // Trie.h:19
static const double kMinusLogThreshold = -39.14; // Trie.cpp:20

struct TrieNodeO {
  // same container type / insertion sequence as the reference so that LOGADD
  // smearing visits children in the same order (Trie.cpp:84-94).
  std::unordered_map<int, int32_t> children;
  int idx;
  std::vector<int> labels;
  std::vector<float> scores;
  float maxScore = 0;
};

struct TrieO {
  std::vector<TrieNodeO> nodes;
  int maxChildren;
  TrieO(int maxChildren_, int rootIdx) : maxChildren(maxChildren_) {
    nodes.emplace_back();
    nodes[0].idx = rootIdx;
  }
  /* Trie.cpp:26-48 */
  int insert(const int32_t* indices, int n, int label, float score) {
    int32_t node = 0;
    for (int i = 0; i < n; ++i) {
      int idx = indices[i];
      if (idx < 0 || idx >= maxChildren) {
        return -1;
      }
      auto it = nodes[node].children.find(idx);
      if (it == nodes[node].children.end()) {
        int32_t id = (int32_t)nodes.size();
        nodes.emplace_back();
        nodes[id].idx = idx;
        nodes[node].children[idx] = id;
        node = id;
      } else {
        node = it->second;
      }
    }
    if ((int)nodes[node].labels.size() < kTrieMaxLabel) {
      nodes[node].labels.push_back(label);
      nodes[node].scores.push_back(score);
    }
    return 0;
  }
  /* Trie.cpp:50-63 */
  int32_t search(const int32_t* indices, int n) const {
    int32_t node = 0;
    for (int i = 0; i < n; ++i) {
      auto it = nodes[node].children.find(indices[i]);
      if (it == nodes[node].children.end()) {
        return -1;
      }
      node = it->second;
    }
    return node;
  }
  /* Trie.cpp:66-77 */
  static double logAdd(double a, double b) {
    if (a < b) {
      std::swap(a, b);
    }
    double d = b - a;
    if (d < kMinusLogThreshold) {
      return a;
    }
    return a + log1p(exp(d));
  }
  /* Trie.cpp:79-95: a node's own scores are always log-added; maxScore is
   * narrowed to float after every step. */
  void smearNode(int32_t id, int mode) {
    TrieNodeO& nd = nodes[id];
    nd.maxScore = -std::numeric_limits<float>::infinity();
    for (float s : nd.scores) {
      nd.maxScore = (float)logAdd(nd.maxScore, s);
    }
    for (auto& kv : nd.children) {
      int32_t c = kv.second;
      smearNode(c, mode);
      if (mode == 2) {
        nodes[id].maxScore = (float)logAdd(nodes[id].maxScore, nodes[c].maxScore);
      } else if (mode == 1 && nodes[c].maxScore > nodes[id].maxScore) {
        nodes[id].maxScore = nodes[c].maxScore;
      }
    }
  }
  void smear(int mode) {
    if (mode != 0) {
      smearNode(0, mode);
    }
  }
};

/* ------------------------------------------------------------------------ */
/* Hypotheses and the candidate machinery (decoder/Utils.h:121-225)          */
/* ------------------------------------------------------------------------ */
struct Hyp {
  double score = 0;
  int32_t lmState = -1;
  int32_t lex = 0; // trie node (lexicon decoder only)
  int32_t parent = -1; // slot in the previous frame, -1 = none
  int32_t token = -1;
  int32_t word = -1;
  bool prevBlank = false;
  double am = 0;
  double lm = 0;
};

/* LexiconFreeDecoder.h:68-78 / LexiconDecoder.h:79-91; `lex` is 0 for every
 * lexicon-free hypothesis so one comparator serves both. */
static inline int keyCompare(const Hyp& a, const Hyp& b) {
  if (a.lmState != b.lmState) {
    return a.lmState > b.lmState ? 1 : -1;
  }
  if (a.lex != b.lex) {
    return a.lex > b.lex ? 1 : -1;
  }
  if (a.token != b.token) {
    return a.token > b.token ? 1 : -1;
  }
  if (a.prevBlank != b.prevBlank) {
    return a.prevBlank > b.prevBlank ? 1 : -1;
  }
  return 0;
}

/* Where the reference stops being a function of its inputs: it orders candidates with std::sort /
 * std::nth_element / std::partial_sort over pointers, so which of several EQUAL-scoring candidates is kept, or comes
 * first, depends on addresses.  This restatement counts every such place it passes, so that a test can tell "the input
 * had a tie" (either answer is the reference's) from "the oracle is wrong" (never excusable):
 *   merge : two candidates of one merge group with the same score (Utils.h:176-198: the survivor keeps the
 *           parent / emitting-model / LM fields of whichever the sort put first);
 *   cut   : the K-th and the (K+1)-th best merged candidate score the same (nth_element, Utils.h:206-214);
 *   order : equal scores inside the sorted final n-best (partial_sort in decodeEnd, Utils.h:216-220);
 *   token : an emission equal to the smallest one kept by the token beam (partial_sort, LexiconFreeDecoder.cpp:42-51);
 *   best  : two hypotheses share the best score where one is picked (findBestAncestor, Utils.h:268-283). */
struct TieCounts {
  long long merge = 0, cut = 0, order = 0, token = 0, best = 0;
};

struct Candidates {
  TieCounts* ties = nullptr;
  double best = kNegInf;
  std::vector<Hyp> cands;
  std::vector<Hyp*> ptrs;
  /* Utils.h:121-129 */
  void reset() {
    best = kNegInf;
    cands.clear();
    ptrs.clear();
  }
  /* Utils.h:131-144 */
  void add(double beamThreshold, const Hyp& h) {
    if (h.score >= best) {
      best = h.score;
    }
    if (h.score >= best - beamThreshold) {
      cands.push_back(h);
    }
  }
  /* Utils.h:146-225 */
  void store(std::vector<Hyp>& out, int beamSize, double threshold,
             bool logAdd, bool returnSorted) {
    out.clear();
    if (cands.empty()) {
      return;
    }
    for (auto& c : cands) { // 1. select (:160-165)
      if (c.score >= threshold) {
        ptrs.push_back(&c);
      }
    }
    // 2. merge (:167-198)
    std::sort(ptrs.begin(), ptrs.end(), [](const Hyp* a, const Hyp* b) {
      int c = keyCompare(*a, *b);
      return c == 0 ? a->score > b->score : c > 0;
    });
    size_t n = 1;
    double groupTop = ptrs.empty() ? 0.0 : ptrs[0]->score; /* the score of the group's best member (its first) */
    for (size_t i = 1; i < ptrs.size(); ++i) {
      if (keyCompare(*ptrs[i], *ptrs[n - 1]) != 0) {
        ptrs[n++] = ptrs[i];
        groupTop = ptrs[i]->score;
      } else {
        if (ties && ptrs[i]->score == groupTop) {
          ++ties->merge; /* (which of the two lends its back-pointer and scores is the sort's choice) */
        }
        double mx = std::max(ptrs[n - 1]->score, ptrs[i]->score);
        if (logAdd) {
          double mn = std::min(ptrs[n - 1]->score, ptrs[i]->score);
          ptrs[n - 1]->score = mx + std::log1p(std::exp(mn - mx));
        } else {
          ptrs[n - 1]->score = mx;
        }
      }
    }
    ptrs.resize(n);
    // 3. prune (:200-220)
    auto byScore = [](const Hyp* a, const Hyp* b) { return a->score > b->score; };
    int nValid = (int)ptrs.size();
    int finalSize = std::min(nValid, beamSize);
    if (ties && nValid > beamSize && beamSize > 0) { /* equal scores across the cut */
      std::vector<double> sc(ptrs.size());
      for (size_t i = 0; i < ptrs.size(); ++i) {
        sc[i] = ptrs[i]->score;
      }
      std::nth_element(sc.begin(), sc.begin() + beamSize, sc.end(), std::greater<double>());
      const double firstOut = sc[beamSize];
      const double lastIn = *std::min_element(sc.begin(), sc.begin() + beamSize);
      if (firstOut == lastIn) {
        ++ties->cut;
      }
    }
    if (!returnSorted && nValid > beamSize) {
      std::nth_element(ptrs.begin(), ptrs.begin() + finalSize, ptrs.end(), byScore);
    } else if (returnSorted) {
      std::partial_sort(ptrs.begin(), ptrs.begin() + finalSize, ptrs.end(), byScore);
    }
    for (int i = 0; i < finalSize; ++i) { // 4. (:222-224)
      out.push_back(*ptrs[i]);
      if (ties && returnSorted && i > 0 && ptrs[i]->score == ptrs[i - 1]->score) {
        ++ties->order;
      }
    }
  }
};

struct Result {
  double score = 0, am = 0, lm = 0;
  std::vector<int> words, tokens;
};

/* ------------------------------------------------------------------------ */
/* Decoder base: frame buffer, back-trace, streaming helpers                 */
/* ------------------------------------------------------------------------ */
struct DecoderO {
  orc_options opt;
  const LMModel* lm;
  int sil, blank;
  std::vector<float> transitions;
  StateArena states;
  Candidates cand;
  mutable TieCounts ties;
  std::vector<std::vector<Hyp>> hyp; // hyp[frame][slot]
  int nDecoded = 0, nPruned = 0;
  bool lexicon = false;

  DecoderO() { cand.ties = &ties; }
  DecoderO(const DecoderO&) = delete;
  virtual ~DecoderO() = default;
  virtual void begin() = 0;
  virtual void step(const float* e, int T, int N) = 0;
  virtual void end() = 0;

  void ensureFrames(size_t n) {
    if (hyp.size() < n) {
      hyp.resize(n);
    }
  }

  /* Utils.h:229-250 (getHypothesis): walk parents from (frame, slot). */
  Result backtrace(int frame, int slot, int finalFrame) const {
    Result r;
    if (slot < 0) {
      return r;
    }
    r.words.assign(finalFrame + 1, -1);
    r.tokens.assign(finalFrame + 1, -1);
    const Hyp* h = &hyp[frame][slot];
    r.score = h->score;
    r.am = h->am;
    r.lm = h->lm;
    int i = 0;
    int f = frame;
    while (true) {
      r.words[finalFrame - i] = lexicon ? h->word : -1;
      r.tokens[finalFrame - i] = h->token;
      if (h->parent < 0 || f == 0) {
        break;
      }
      --f;
      h = &hyp[f][h->parent];
      ++i;
    }
    return r;
  }

  bool isComplete(int frame, int slot) const {
    if (!lexicon) {
      return true; // LexiconFreeDecoder.h:84-86
    }
    const Hyp& h = hyp[frame][slot]; // LexiconDecoder.h:97-99
    return h.parent < 0 || frame == 0 || hyp[frame - 1][h.parent].word >= 0;
  }

  /* Utils.h:268-310 (findBestAncestor) */
  bool bestAncestor(int finalFrame, int& lookBack, int& frameOut, int& slotOut) const {
    const auto& fin = hyp[finalFrame];
    if (fin.empty()) {
      return false;
    }
    int bestSlot = 0;
    double bestScore = fin[0].score;
    for (int r = 1; r < (int)fin.size(); ++r) {
      if (fin[r].score > bestScore) {
        bestScore = fin[r].score;
        bestSlot = r;
      }
    }
    for (int r = 0; r < (int)fin.size(); ++r) {
      if (r != bestSlot && fin[r].score == bestScore) {
        ++ties.best; /* (the reference takes the first in ITS order of the beam, which nth_element left unspecified) */
        break;
      }
    }
    int f = finalFrame, s = bestSlot, n = 0;
    auto up = [&]() {
      int p = (f == 0) ? -1 : hyp[f][s].parent;
      s = p;
      --f;
    };
    while (s >= 0 && n < lookBack) {
      ++n;
      up();
    }
    const int maxLookBack = lookBack + kLookBackLimit;
    while (s >= 0) {
      if (isComplete(f, s)) {
        break;
      }
      ++n;
      up();
      if (n == maxLookBack) {
        break;
      }
    }
    lookBack = n;
    frameOut = f;
    slotOut = s;
    return s >= 0;
  }

  int finalFrame() const { return nDecoded - nPruned; }

  /* LexiconFreeDecoder.cpp:160-166 / LexiconDecoder.cpp:276-283 */
  std::vector<Result> allFinal() const {
    std::vector<Result> out;
    int ff = finalFrame();
    if (lexicon ? (ff < 1) : hyp.empty()) {
      return out;
    }
    for (int r = 0; r < (int)hyp[ff].size(); ++r) {
      out.push_back(backtrace(ff, r, ff));
    }
    return out;
  }

  /* LexiconFreeDecoder.cpp:188-194 / LexiconDecoder.cpp:285-293 */
  Result best(int lookBack) const {
    int ff = finalFrame();
    if (lexicon && ff - lookBack < 1) {
      return Result();
    }
    int f, s;
    int lb = lookBack;
    if (!bestAncestor(ff, lb, f, s)) {
      return Result();
    }
    return backtrace(f, s, ff - lb);
  }

  /* LexiconFreeDecoder.cpp:205-227 / LexiconDecoder.cpp:304-325,
   * Utils.h:312-342 (pruneAndNormalize) */
  void prune(int lookBack) {
    int ff = finalFrame();
    if (ff - lookBack < 1) {
      return;
    }
    int f, s, lb = lookBack;
    if (!bestAncestor(ff, lb, f, s)) {
      return;
    }
    int startFrame = ff - lb;
    if (startFrame < 1) {
      return;
    }
    ensureFrames(startFrame + lb + 1);
    for (int i = 0; i < (int)hyp.size(); ++i) {
      if (i <= lb) {
        hyp[i].swap(hyp[i + startFrame]);
      } else {
        hyp[i].clear();
      }
    }
    for (auto& h : hyp[0]) {
      h.parent = -1;
    }
    double largest = hyp[lb].front().score;
    for (size_t i = 1; i < hyp[lb].size(); ++i) {
      if (largest < hyp[lb][i].score) {
        largest = hyp[lb][i].score;
      }
    }
    for (auto& h : hyp[lb]) {
      h.score -= largest;
    }
    nPruned = nDecoded - lb;
  }

  /* per-frame token short-list (LexiconFreeDecoder.cpp:42-51) */
  void tokenShortlist(const float* row, int N, std::vector<size_t>& idx) const {
    idx.resize(N);
    std::iota(idx.begin(), idx.end(), 0);
    if (N > opt.beam_size_token) {
      std::partial_sort(idx.begin(), idx.begin() + opt.beam_size_token, idx.end(),
                        [row](size_t l, size_t r) { return row[l] > row[r]; });
      if (opt.beam_size_token > 0) {
        const float lastIn = row[idx[opt.beam_size_token - 1]];
        for (int i = opt.beam_size_token; i < N; ++i) {
          if (row[idx[i]] == lastIn) {
            ++ties.token;
            break;
          }
        }
      }
    }
  }
};

/* ------------------------------------------------------------------------ */
/* LexiconFreeDecoder (decoder/LexiconFreeDecoder.cpp:20-158)                */
/* ------------------------------------------------------------------------ */
struct LexFreeO : DecoderO {
  /* :20-28 */
  void begin() override {
    hyp.clear();
    states.clear();
    hyp.emplace_back();
    Hyp h;
    h.score = 0.0;
    h.lmState = states.root(*lm, false);
    h.parent = -1;
    h.token = sil;
    hyp[0].push_back(h);
    nDecoded = 0;
    nPruned = 0;
  }
  /* :30-125 */
  void step(const float* emissions, int T, int N) override {
    int startFrame = nDecoded - nPruned;
    ensureFrames(startFrame + T + 2);
    std::vector<size_t> idx;
    const bool asg = opt.criterion == 0, ctc = opt.criterion == 1;
    const int nTok = std::min(opt.beam_size_token, N);
    for (int t = 0; t < T; ++t) {
      const float* row = emissions + (size_t)t * N;
      tokenShortlist(row, N, idx);
      cand.reset();
      const auto& prevs = hyp[startFrame + t];
      for (int pi = 0; pi < (int)prevs.size(); ++pi) {
        const Hyp& prev = prevs[pi];
        const int prevIdx = prev.token;
        for (int r = 0; r < nTok; ++r) {
          int n = (int)idx[r];
          double amScore = row[n];
          if (nDecoded + t > 0 && asg) {
            amScore += transitions[(size_t)n * N + prevIdx]; // :60-62
          }
          double score = prev.score + row[n]; // :64 (transition NOT in score)
          if (n == sil) {
            score += opt.sil_score;
          }
          Hyp c;
          c.parent = pi;
          c.token = n;
          c.am = prev.am + amScore;
          if ((asg && n != prevIdx) ||
              (ctc && n != blank && (n != prevIdx || prev.prevBlank))) { // :69-85
            auto sp = states.score(*lm, prev.lmState, n);
            float lmScore = sp.second;
            c.score = score + opt.lm_weight * lmScore;
            c.lmState = sp.first;
            c.prevBlank = false;
            c.lm = prev.lm + lmScore;
          } else if (ctc && n == blank) { // :86-97
            c.score = score;
            c.lmState = prev.lmState;
            c.prevBlank = true;
            c.lm = prev.lm;
          } else { // :98-110
            c.score = score;
            c.lmState = prev.lmState;
            c.prevBlank = false;
            c.lm = prev.lm;
          }
          cand.add(opt.beam_threshold, c);
        }
      }
      cand.store(hyp[startFrame + t + 1], opt.beam_size,
                 cand.best - opt.beam_threshold, opt.log_add != 0, false);
    }
    nDecoded += T;
  }
  /* :127-158 */
  void end() override {
    cand.reset();
    int ff = finalFrame();
    ensureFrames(ff + 2);
    const auto& prevs = hyp[ff];
    for (int pi = 0; pi < (int)prevs.size(); ++pi) {
      const Hyp& prev = prevs[pi];
      auto sp = states.finish(*lm, prev.lmState);
      float lmScore = sp.second;
      Hyp c;
      c.score = prev.score + opt.lm_weight * lmScore;
      c.lmState = sp.first;
      c.parent = pi;
      c.token = sil;
      c.prevBlank = false;
      c.am = prev.am;
      c.lm = prev.lm + lmScore;
      cand.add(opt.beam_threshold, c);
    }
    cand.store(hyp[ff + 1], opt.beam_size, cand.best - opt.beam_threshold,
               opt.log_add != 0, true);
    ++nDecoded;
  }
};

/* ------------------------------------------------------------------------ */
/* LexiconDecoder (decoder/LexiconDecoder.cpp:21-274)                        */
/* ------------------------------------------------------------------------ */
struct LexiconO : DecoderO {
  const TrieO* trie = nullptr;
  int unk = -1;
  bool isLmToken = false;

  /* :21-30 */
  void begin() override {
    hyp.clear();
    states.clear();
    hyp.emplace_back();
    Hyp h;
    h.score = 0.0;
    h.lmState = states.root(*lm, false);
    h.lex = 0;
    h.parent = -1;
    h.token = sil;
    h.word = -1;
    hyp[0].push_back(h);
    nDecoded = 0;
    nPruned = 0;
  }
  /* :32-229 */
  void step(const float* emissions, int T, int N) override {
    int startFrame = nDecoded - nPruned;
    ensureFrames(startFrame + T + 2);
    std::vector<size_t> idx;
    const bool asg = opt.criterion == 0, ctc = opt.criterion == 1;
    const int nTok = std::min(opt.beam_size_token, N);
    for (int t = 0; t < T; ++t) {
      const float* row = emissions + (size_t)t * N;
      tokenShortlist(row, N, idx);
      cand.reset();
      const auto& prevs = hyp[startFrame + t];
      for (int pi = 0; pi < (int)prevs.size(); ++pi) {
        const Hyp& prev = prevs[pi];
        const TrieNodeO& prevLex = trie->nodes[prev.lex];
        const int prevIdx = prev.token;
        const bool atRoot = prev.lex == 0;
        const float lexMaxScore = atRoot ? 0 : prevLex.maxScore; // :58-59

        /* (1) children, :62-165 */
        for (int r = 0; r < nTok; ++r) {
          int n = (int)idx[r];
          auto it = prevLex.children.find(n);
          if (it == prevLex.children.end()) {
            continue;
          }
          const int32_t lexId = it->second;
          const TrieNodeO& lex = trie->nodes[lexId];
          double amScore = row[n];
          if (nDecoded + t > 0 && asg) {
            amScore += transitions[(size_t)n * N + prevIdx];
          }
          double score = prev.score + amScore;
          if (n == sil) {
            score += opt.sil_score;
          }
          int32_t lmState = -1;
          double lmScore = 0.;
          if (isLmToken) { // :82-86
            auto sp = states.score(*lm, prev.lmState, n);
            lmState = sp.first;
            lmScore = sp.second;
          }
          /* (1a) eat a new token, :89-110 */
          if (!ctc || prev.prevBlank || n != prevIdx) {
            if (!lex.children.empty()) {
              if (!isLmToken) {
                lmState = prev.lmState;
                lmScore = lex.maxScore - lexMaxScore; // float subtraction
              }
              Hyp c;
              c.score = score + opt.lm_weight * lmScore;
              c.lmState = lmState;
              c.lex = lexId;
              c.parent = pi;
              c.token = n;
              c.word = -1;
              c.prevBlank = false;
              c.am = prev.am + amScore;
              c.lm = prev.lm + lmScore;
              cand.add(opt.beam_threshold, c);
            }
          }
          /* (1b) a true word, :113-142 */
          for (int label : lex.labels) {
            if (atRoot && prev.token == n) {
              continue; // :114-122
            }
            if (!isLmToken) {
              auto sp = states.score(*lm, prev.lmState, label);
              lmState = sp.first;
              lmScore = sp.second - lexMaxScore; // float subtraction
            }
            Hyp c;
            c.score = score + opt.lm_weight * lmScore + opt.word_score;
            c.lmState = lmState;
            c.lex = 0;
            c.parent = pi;
            c.token = n;
            c.word = label;
            c.prevBlank = false;
            c.am = prev.am + amScore;
            c.lm = prev.lm + lmScore;
            cand.add(opt.beam_threshold, c);
          }
          /* (1c) unknown word, :145-164 */
          if (lex.labels.empty() && opt.unk_score > kNegInf) {
            if (!isLmToken) {
              auto sp = states.score(*lm, prev.lmState, unk);
              lmState = sp.first;
              lmScore = sp.second - lexMaxScore;
            }
            Hyp c;
            c.score = score + opt.lm_weight * lmScore + opt.unk_score;
            c.lmState = lmState;
            c.lex = 0;
            c.parent = pi;
            c.token = n;
            c.word = unk;
            c.prevBlank = false;
            c.am = prev.am + amScore;
            c.lm = prev.lm + lmScore;
            cand.add(opt.beam_threshold, c);
          }
        }
        /* (2) same lexicon node, :168-194 */
        if (!ctc || !prev.prevBlank || atRoot) {
          int n = atRoot ? sil : prevIdx;
          double amScore = row[n];
          if (nDecoded + t > 0 && asg) {
            amScore += transitions[(size_t)n * N + prevIdx];
          }
          double score = prev.score + amScore;
          if (n == sil) {
            score += opt.sil_score;
          }
          Hyp c;
          c.score = score;
          c.lmState = prev.lmState;
          c.lex = prev.lex;
          c.parent = pi;
          c.token = n;
          c.word = -1;
          c.prevBlank = false;
          c.am = prev.am + amScore;
          c.lm = prev.lm;
          cand.add(opt.beam_threshold, c);
        }
        /* (3) CTC blank, :197-213 */
        if (ctc) {
          int n = blank;
          double amScore = row[n];
          Hyp c;
          c.score = prev.score + amScore;
          c.lmState = prev.lmState;
          c.lex = prev.lex;
          c.parent = pi;
          c.token = n;
          c.word = -1;
          c.prevBlank = true;
          c.am = prev.am + amScore;
          c.lm = prev.lm;
          cand.add(opt.beam_threshold, c);
        }
      }
      cand.store(hyp[startFrame + t + 1], opt.beam_size,
                 cand.best - opt.beam_threshold, opt.log_add != 0, false);
    }
    nDecoded += T;
  }
  /* :231-274 */
  void end() override {
    cand.reset();
    int ff = finalFrame();
    ensureFrames(ff + 2);
    const auto& prevs = hyp[ff];
    bool niceEnding = false;
    for (const Hyp& p : prevs) {
      if (p.lex == 0) {
        niceEnding = true;
        break;
      }
    }
    for (int pi = 0; pi < (int)prevs.size(); ++pi) {
      const Hyp& prev = prevs[pi];
      if (!niceEnding || prev.lex == 0) {
        auto sp = states.finish(*lm, prev.lmState);
        float lmScore = sp.second;
        Hyp c;
        c.score = prev.score + opt.lm_weight * lmScore;
        c.lmState = sp.first;
        c.lex = prev.lex;
        c.parent = pi;
        c.token = sil;
        c.word = -1;
        c.prevBlank = false;
        c.am = prev.am;
        c.lm = prev.lm + lmScore;
        cand.add(opt.beam_threshold, c);
      }
    }
    cand.store(hyp[ff + 1], opt.beam_size, cand.best - opt.beam_threshold,
               opt.log_add != 0, true);
    ++nDecoded;
  }
};

} // namespace orc

/* ------------------------------------------------------------------------ */
/* C interface                                                               */
/* ------------------------------------------------------------------------ */
using namespace orc;

extern "C" {

void* ORC_FN(lm_zero_create)(void) { return new ZeroModel(); }

void* ORC_FN(lm_arpa_create)(const char* arpa_path, const char* usr_words) {
  try {
    auto* m = new ArpaLM();
    m->model.load(arpa_path);
    std::istringstream ss(usr_words ? usr_words : "");
    std::string w;
    while (std::getline(ss, w, '\n')) {
      m->usrToLm.push_back(m->model.index(w));
    }
    return m;
  } catch (...) {
    return nullptr;
  }
}

void* ORC_FN(lm_lastword_create)(int32_t, int32_t seed) {
  auto* m = new LastWordModel();
  m->seed = seed;
  return m;
}

void ORC_FN(lm_destroy)(void* lm) { delete (LMModel*)lm; }

float ORC_FN(lm_score_sequence)(void* lm, const int32_t* words, int32_t n,
                                int32_t with_finish, float* per_word) {
  const LMModel& m = *(LMModel*)lm;
  StateArena st;
  int32_t s = st.root(m, false);
  float total = 0;
  for (int i = 0; i < n; ++i) {
    auto sp = st.score(m, s, words[i]);
    s = sp.first;
    if (per_word) {
      per_word[i] = sp.second;
    }
    total += sp.second;
  }
  if (with_finish) {
    total += st.finish(m, s).second;
  }
  return total;
}

void* ORC_FN(trie_create)(int32_t max_children, int32_t root_idx) {
  return new TrieO(max_children, root_idx);
}
int32_t ORC_FN(trie_insert)(void* trie, const int32_t* indices, int32_t n,
                            int32_t label, float score) {
  return ((TrieO*)trie)->insert(indices, n, label, score);
}
void ORC_FN(trie_smear)(void* trie, int32_t mode) { ((TrieO*)trie)->smear(mode); }
int32_t ORC_FN(trie_search)(void* trie, const int32_t* indices, int32_t n,
                            float* max_score, int32_t* n_labels) {
  TrieO* t = (TrieO*)trie;
  int32_t id = t->search(indices, n);
  if (id < 0) {
    return 0;
  }
  if (max_score) {
    *max_score = t->nodes[id].maxScore;
  }
  if (n_labels) {
    *n_labels = (int32_t)t->nodes[id].labels.size();
  }
  return 1;
}
int64_t ORC_FN(trie_num_nodes)(void* trie) { return (int64_t)((TrieO*)trie)->nodes.size(); }
void ORC_FN(trie_destroy)(void* trie) { delete (TrieO*)trie; }

static void fillCommon(DecoderO* d, const orc_options* opt, void* lm, int sil,
                       int blank, const float* tr, int ntr) {
  d->opt = *opt;
  d->lm = (const LMModel*)lm;
  d->sil = sil;
  d->blank = blank;
  if (tr && ntr > 0) {
    d->transitions.assign(tr, tr + ntr);
  }
}

void* ORC_FN(decoder_create_lexfree)(const orc_options* opt, void* lm,
                                     int32_t sil, int32_t blank,
                                     const float* transitions,
                                     int32_t n_transitions) {
  auto* d = new LexFreeO();
  fillCommon(d, opt, lm, sil, blank, transitions, n_transitions);
  d->lexicon = false;
  return (DecoderO*)d;
}

void* ORC_FN(decoder_create_lexicon)(const orc_options* opt, void* trie,
                                     void* lm, int32_t sil, int32_t blank,
                                     int32_t unk, const float* transitions,
                                     int32_t n_transitions,
                                     int32_t is_lm_token) {
  auto* d = new LexiconO();
  fillCommon(d, opt, lm, sil, blank, transitions, n_transitions);
  d->lexicon = true;
  d->trie = (const TrieO*)trie;
  d->unk = unk;
  d->isLmToken = is_lm_token != 0;
  return (DecoderO*)d;
}

void ORC_FN(decoder_destroy)(void* dec) { delete (DecoderO*)dec; }
void ORC_FN(decoder_begin)(void* dec) { ((DecoderO*)dec)->begin(); }
void ORC_FN(decoder_step)(void* dec, const float* e, int32_t T, int32_t N) {
  ((DecoderO*)dec)->step(e, T, N);
}
void ORC_FN(decoder_end)(void* dec) { ((DecoderO*)dec)->end(); }
void ORC_FN(decoder_prune)(void* dec, int32_t lb) { ((DecoderO*)dec)->prune(lb); }
int32_t ORC_FN(decoder_n_frames_in_buffer)(void* dec) {
  return ((DecoderO*)dec)->finalFrame() + 1;
}

int32_t ORC_FN(decoder_n_final)(void* dec, int32_t* length) {
  DecoderO* d = (DecoderO*)dec;
  int ff = d->finalFrame();
  if (length) {
    *length = ff + 1;
  }
  if (d->lexicon ? (ff < 1) : d->hyp.empty()) {
    return 0;
  }
  return (int32_t)d->hyp[ff].size();
}

static void copyResult(const Result& r, double* scores, int32_t* tokens,
                       int32_t* words) {
  scores[0] = r.score;
  scores[1] = r.am;
  scores[2] = r.lm;
  if (tokens) {
    std::copy(r.tokens.begin(), r.tokens.end(), tokens);
  }
  if (words) {
    std::copy(r.words.begin(), r.words.end(), words);
  }
}

int32_t ORC_FN(decoder_get_all)(void* dec, int32_t max_hyp, double* scores,
                                int32_t* tokens, int32_t* words) {
  DecoderO* d = (DecoderO*)dec;
  auto all = d->allFinal();
  int n = std::min<int>(max_hyp, (int)all.size());
  for (int i = 0; i < n; ++i) {
    size_t len = all[i].tokens.size();
    copyResult(all[i], scores + 3 * i, tokens ? tokens + i * len : nullptr,
               words ? words + i * len : nullptr);
  }
  return n;
}

int32_t ORC_FN(decoder_get_best)(void* dec, int32_t look_back, double* scores,
                                 int32_t* tokens, int32_t* words,
                                 int32_t capacity) {
  DecoderO* d = (DecoderO*)dec;
  Result r = d->best(look_back);
  if ((int)r.tokens.size() > capacity) {
    return -1;
  }
  copyResult(r, scores, tokens, words);
  return (int32_t)r.tokens.size();
}

/* ties the decoder has passed since it was created (or since the last call with reset != 0): out[0..4] = merge, cut,
 * order, token, best (see TieCounts).  Oracle only: the reference build has no such counters. */
void orc_decoder_ties(void* dec, int64_t* out, int32_t reset) {
  DecoderO* d = (DecoderO*)dec;
  out[0] = d->ties.merge;
  out[1] = d->ties.cut;
  out[2] = d->ties.order;
  out[3] = d->ties.token;
  out[4] = d->ties.best;
  if (reset) {
    d->ties = TieCounts();
  }
}

} // extern "C"
