/*
 * oracle/ref_driver.cpp -- C driver around the UNMODIFIED reference library.
 *
 * TEST INFRASTRUCTURE ONLY.  Compiled only by oracle/Makefile target `ref`,
 * against the headers and sources where they lie under /root/reference (never
 * copied into this repo); output goes to oracle/_ref/ (git-ignored, travels to
 * the GPU box as a prebuilt .so).  It exposes the same C interface as the
 * restatement (oracle/orc_api.h) under the prefix ref_, so that
 *   - tests/golden/make_golden.py can generate golden vectors from the true
 *     reference, and
 *   - bench.py's cpu_baseline leg can time the true reference
 *     (cpu_baseline.kind == "reference").
 *
 * The only non-reference arithmetic in here is ArpaRefLM: KenLM is an absent
 * third-party dependency (see oracle/arpa_lm.h), so the n-gram LM plugged into
 * the reference decoder is the same ARPA back-off model the restatement uses,
 * wrapped as an fl::lib::text::LM subclass exactly the way KenLM.cpp wraps
 * libkenlm (state trie via LMState::child, decoder/lm/KenLM.cpp:63-83).
 */
#define ORC_PREFIX_REF 1
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <sstream>
#include <string>
#include <vector>

#include "flashlight/lib/text/decoder/LexiconDecoder.h"
#include "flashlight/lib/text/decoder/LexiconFreeDecoder.h"
#include "flashlight/lib/text/decoder/Trie.h"
#include "flashlight/lib/text/decoder/lm/LM.h"
#include "flashlight/lib/text/decoder/lm/ZeroLM.h"
#include "flashlight/lib/text/dictionary/Defines.h"
#include "flashlight/lib/text/dictionary/Dictionary.h"
#include "flashlight/lib/text/dictionary/Utils.h"

#include "arpa_lm.h"
#include "orc_api.h"

using namespace fl::lib::text;

namespace {

struct ArpaRefState : LMState {
  std::vector<int32_t> ctx;
};

class ArpaRefLM : public LM {
 public:
  orc::ArpaModel model;
  std::vector<int32_t> usrToLm;

  LMStatePtr start(bool startWithNothing) override {
    auto s = std::make_shared<ArpaRefState>();
    if (!startWithNothing) {
      s->ctx.push_back(model.bos);
    }
    return s;
  }
  std::pair<LMStatePtr, float> score(const LMStatePtr& state,
                                     const int usrTokenIdx) override {
    if (usrTokenIdx < 0 || usrTokenIdx >= (int)usrToLm.size()) {
      throw std::runtime_error("[ArpaRefLM] Invalid user token index");
    }
    auto in = std::static_pointer_cast<ArpaRefState>(state);
    auto out = in->child<ArpaRefState>(usrTokenIdx);
    std::vector<int32_t> ctx;
    float s = model.score(in->ctx, usrToLm[usrTokenIdx], ctx);
    out->ctx = std::move(ctx);
    return {out, s};
  }
  std::pair<LMStatePtr, float> finish(const LMStatePtr& state) override {
    auto in = std::static_pointer_cast<ArpaRefState>(state);
    auto out = in->child<ArpaRefState>(-1);
    std::vector<int32_t> ctx;
    float s = model.score(in->ctx, model.eos, ctx);
    out->ctx = std::move(ctx);
    return {out, s};
  }
};

static float pairScore(int prev, int w, int seed) {
  uint32_t h = (uint32_t)prev * 1000003u + (uint32_t)w * 7919u + (uint32_t)seed;
  h ^= h >> 13;
  h *= 0x5BD1E995u;
  h ^= h >> 15;
  return -(float)(h & 0xFFFFu) / 8192.0f;
}

/* a user LM with one shared state object per last input (orc_api.h lm_lastword_create) */
struct LastWordState : LMState {
  int last = -1;
};
class LastWordRefLM : public LM {
 public:
  LastWordRefLM(int n, int seed) : seed_(seed) {
    begin_ = std::make_shared<LastWordState>();
    for (int i = 0; i < n; ++i) {
      states_.push_back(std::make_shared<LastWordState>());
      states_.back()->last = i;
    }
  }
  LMStatePtr start(bool) override { return begin_; }
  std::pair<LMStatePtr, float> score(const LMStatePtr& state, const int idx) override {
    const int prev = std::static_pointer_cast<LastWordState>(state)->last;
    return {states_.at((size_t)idx), pairScore(prev + 2, idx + 2, seed_)};
  }
  std::pair<LMStatePtr, float> finish(const LMStatePtr& state) override {
    const int prev = std::static_pointer_cast<LastWordState>(state)->last;
    return {state, pairScore(prev + 2, 1, seed_)}; /* (the state itself, as ZeroLM::finish: the final n-best keeps one entry per last input) */
  }

 private:
  int seed_;
  std::shared_ptr<LastWordState> begin_;
  std::vector<std::shared_ptr<LastWordState>> states_;
};

struct LMBox {
  LMPtr lm;
};
struct TrieBox {
  TriePtr trie;
  int64_t nNodes = 1;
};
struct DecBox {
  std::unique_ptr<Decoder> dec;
  bool lexicon = false;
};

int64_t countNodes(const TrieNode* n) {
  int64_t c = 1;
  for (auto& kv : n->children) {
    c += countNodes(kv.second.get());
  }
  return c;
}

CriterionType crit(int c) {
  return c == 0 ? CriterionType::ASG : CriterionType::CTC;
}

void copyResult(const DecodeResult& r, double* scores, int32_t* tokens,
                int32_t* words) {
  scores[0] = r.score;
  scores[1] = r.emittingModelScore;
  scores[2] = r.lmScore;
  if (tokens) {
    std::copy(r.tokens.begin(), r.tokens.end(), tokens);
  }
  if (words) {
    std::copy(r.words.begin(), r.words.end(), words);
  }
}

} // namespace

extern "C" {

void* ref_lm_zero_create(void) {
  auto* b = new LMBox();
  b->lm = std::make_shared<ZeroLM>();
  return b;
}

void* ref_lm_arpa_create(const char* arpa_path, const char* usr_words) {
  try {
    auto lm = std::make_shared<ArpaRefLM>();
    lm->model.load(arpa_path);
    std::istringstream ss(usr_words ? usr_words : "");
    std::string w;
    while (std::getline(ss, w, '\n')) {
      lm->usrToLm.push_back(lm->model.index(w));
    }
    auto* b = new LMBox();
    b->lm = lm;
    return b;
  } catch (...) {
    return nullptr;
  }
}

void* ref_lm_lastword_create(int32_t n_idx, int32_t seed) {
  auto* b = new LMBox();
  b->lm = std::make_shared<LastWordRefLM>(n_idx, seed);
  return b;
}

void ref_lm_destroy(void* lm) { delete (LMBox*)lm; }

float ref_lm_score_sequence(void* lm, const int32_t* words, int32_t n,
                            int32_t with_finish, float* per_word) {
  LMPtr m = ((LMBox*)lm)->lm;
  auto s = m->start(false);
  float total = 0, sc = 0;
  for (int i = 0; i < n; ++i) {
    std::tie(s, sc) = m->score(s, words[i]);
    if (per_word) {
      per_word[i] = sc;
    }
    total += sc;
  }
  if (with_finish) {
    std::tie(s, sc) = m->finish(s);
    total += sc;
  }
  return total;
}

void* ref_trie_create(int32_t max_children, int32_t root_idx) {
  auto* b = new TrieBox();
  b->trie = std::make_shared<Trie>(max_children, root_idx);
  return b;
}
int32_t ref_trie_insert(void* trie, const int32_t* indices, int32_t n,
                        int32_t label, float score) {
  try {
    ((TrieBox*)trie)->trie->insert(std::vector<int>(indices, indices + n), label, score);
    return 0;
  } catch (const std::out_of_range&) {
    return -1;
  }
}
void ref_trie_smear(void* trie, int32_t mode) {
  ((TrieBox*)trie)->trie->smear((SmearingMode)mode);
}
int32_t ref_trie_search(void* trie, const int32_t* indices, int32_t n,
                        float* max_score, int32_t* n_labels) {
  auto node = ((TrieBox*)trie)->trie->search(std::vector<int>(indices, indices + n));
  if (!node) {
    return 0;
  }
  if (max_score) {
    *max_score = node->maxScore;
  }
  if (n_labels) {
    *n_labels = (int32_t)node->labels.size();
  }
  return 1;
}
int64_t ref_trie_num_nodes(void* trie) {
  return countNodes(((TrieBox*)trie)->trie->getRoot());
}
void ref_trie_destroy(void* trie) { delete (TrieBox*)trie; }

void* ref_decoder_create_lexfree(const orc_options* o, void* lm, int32_t sil,
                                 int32_t blank, const float* transitions,
                                 int32_t n_transitions) {
  LexiconFreeDecoderOptions opt{o->beam_size,  o->beam_size_token,
                                o->beam_threshold, o->lm_weight,
                                o->sil_score,  o->log_add != 0,
                                crit(o->criterion)};
  std::vector<float> tr;
  if (transitions && n_transitions > 0) {
    tr.assign(transitions, transitions + n_transitions);
  }
  auto* b = new DecBox();
  b->dec.reset(new LexiconFreeDecoder(opt, ((LMBox*)lm)->lm, sil, blank, tr));
  b->lexicon = false;
  return b;
}

void* ref_decoder_create_lexicon(const orc_options* o, void* trie, void* lm,
                                 int32_t sil, int32_t blank, int32_t unk,
                                 const float* transitions,
                                 int32_t n_transitions, int32_t is_lm_token) {
  LexiconDecoderOptions opt{o->beam_size,  o->beam_size_token,
                            o->beam_threshold, o->lm_weight,
                            o->word_score, o->unk_score,
                            o->sil_score,  o->log_add != 0,
                            crit(o->criterion)};
  std::vector<float> tr;
  if (transitions && n_transitions > 0) {
    tr.assign(transitions, transitions + n_transitions);
  }
  auto* b = new DecBox();
  b->dec.reset(new LexiconDecoder(opt, ((TrieBox*)trie)->trie, ((LMBox*)lm)->lm,
                                  sil, blank, unk, tr, is_lm_token != 0));
  b->lexicon = true;
  return b;
}

void ref_decoder_destroy(void* dec) { delete (DecBox*)dec; }
void ref_decoder_begin(void* dec) { ((DecBox*)dec)->dec->decodeBegin(); }
void ref_decoder_step(void* dec, const float* e, int32_t T, int32_t N) {
  ((DecBox*)dec)->dec->decodeStep(e, T, N);
}
void ref_decoder_end(void* dec) { ((DecBox*)dec)->dec->decodeEnd(); }
void ref_decoder_prune(void* dec, int32_t lb) { ((DecBox*)dec)->dec->prune(lb); }
int32_t ref_decoder_n_frames_in_buffer(void* dec) {
  return ((DecBox*)dec)->dec->nDecodedFramesInBuffer();
}

int32_t ref_decoder_n_final(void* dec, int32_t* length) {
  auto all = ((DecBox*)dec)->dec->getAllFinalHypothesis();
  if (length) {
    *length = all.empty() ? ((DecBox*)dec)->dec->nDecodedFramesInBuffer()
                          : (int32_t)all[0].tokens.size();
  }
  return (int32_t)all.size();
}

int32_t ref_decoder_get_all(void* dec, int32_t max_hyp, double* scores,
                            int32_t* tokens, int32_t* words) {
  auto all = ((DecBox*)dec)->dec->getAllFinalHypothesis();
  int n = std::min<int>(max_hyp, (int)all.size());
  for (int i = 0; i < n; ++i) {
    size_t len = all[i].tokens.size();
    copyResult(all[i], scores + 3 * i, tokens ? tokens + i * len : nullptr,
               words ? words + i * len : nullptr);
  }
  return n;
}

int32_t ref_decoder_get_best(void* dec, int32_t look_back, double* scores,
                             int32_t* tokens, int32_t* words,
                             int32_t capacity) {
  DecodeResult r = ((DecBox*)dec)->dec->getBestHypothesis(look_back);
  if ((int)r.tokens.size() > capacity) {
    return -1;
  }
  copyResult(r, scores, tokens, words);
  return (int32_t)r.tokens.size();
}

/* DecoderTest.cpp:94-98,137-146 setup, run through the reference's own
 * loadWords / createWordDict / Dictionary / tkn2Idx.  Returns a malloc'd text:
 *   line 0: "<n_tokens> <n_words> <sil_idx> <unk_idx>"
 *   then one line per (word, spelling): "<word_idx>\t<word>\t<i0> <i1> ..."
 * in the reference's lexicon iteration order (= its trie insertion order).
 * Words are listed in wordDict index order in a trailing block
 *   "#words\n<word0>\n<word1>..." so the caller can rebuild usr->LM maps. */
char* ref_lexicon_dump(const char* words_path, const char* tokens_path,
                       const char* extra_token, int32_t max_reps) {
  try {
    auto lexicon = loadWords(words_path);
    Dictionary tokenDict{std::string(tokens_path)};
    if (extra_token && extra_token[0]) {
      tokenDict.addEntry(extra_token);
    }
    auto wordDict = createWordDict(lexicon);
    std::ostringstream os;
    os << tokenDict.indexSize() << ' ' << wordDict.indexSize() << ' '
       << tokenDict.getIndex("|") << ' ' << wordDict.getIndex(kUnkToken) << '\n';
    for (const auto& it : lexicon) {
      int usrIdx = wordDict.getIndex(it.first);
      for (const auto& tokens : it.second) {
        auto idx = tkn2Idx(tokens, tokenDict, max_reps);
        os << usrIdx << '\t' << it.first << '\t';
        for (size_t i = 0; i < idx.size(); ++i) {
          os << (i ? " " : "") << idx[i];
        }
        os << '\n';
      }
    }
    os << "#words\n";
    for (size_t i = 0; i < wordDict.indexSize(); ++i) {
      os << wordDict.getEntry((int)i) << '\n';
    }
    std::string s = os.str();
    char* out = (char*)malloc(s.size() + 1);
    memcpy(out, s.c_str(), s.size() + 1);
    return out;
  } catch (...) {
    return nullptr;
  }
}
void ref_free(void* p) { free(p); }

} // extern "C"
