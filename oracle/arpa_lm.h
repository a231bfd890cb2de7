/*
 * oracle/arpa_lm.h -- ARPA back-off n-gram model, CPU, float arithmetic.
 * TEST INFRASTRUCTURE ONLY (see oracle/orc_api.h).
 *
 * Why this exists: the reference's KenLM adapter (decoder/lm/KenLM.cpp:32-83)
 * delegates all arithmetic to the third-party library KenLM, pinned at
 * https://github.com/jacobkahn/kenlm.git @ 5bf7b46558e1c5595bf3b8c9b0b1f9d8d257040a
 * (cmake/BuildKenlm.cmake:6-7), which is NOT under /root/reference and not
 * installed in this image.  This header restates KenLM's published scoring
 * rule for ARPA models (lm/model.cc GenericModel::FullScore): the score of
 * word w after context c is the log10 probability of the longest n-gram
 * (c[-k:], w) present in the model, plus the back-off weights of every
 * context c[-j:] (j = k+1 .. |c|) that was skipped, accumulated in `float`
 * from the shortest skipped context to the longest.  Out-of-vocabulary words
 * map to <unk> (KenLM vocabulary index 0).
 *
 * Pinned against the reference's own known answers
 * (flashlight/lib/text/test/decoder/DecoderTest.cpp:107-120,148-155,184-194):
 * see tests/test_oracle_golden.py.  Bit-level agreement with real KenLM's
 * binary formats is unverifiable here ("parity unpinned" beyond those 19
 * numbers @1e-5 / 1e-3).
 */
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

namespace orc {

struct NgramKeyHash {
  size_t operator()(const std::vector<int32_t>& v) const {
    uint64_t h = 1469598103934665603ull;
    for (int32_t x : v) {
      h ^= (uint32_t)x;
      h *= 1099511628211ull;
    }
    return (size_t)h;
  }
};

struct ArpaModel {
  struct Entry {
    float prob;
    float backoff;
  };
  int order = 0;
  std::unordered_map<std::string, int32_t> vocab; // word -> LM id
  std::vector<std::string> words;
  // table[k] holds (k+1)-grams keyed by ids, oldest word first
  std::vector<std::unordered_map<std::vector<int32_t>, Entry, NgramKeyHash>>
      table;
  int32_t unk = 0, bos = -1, eos = -1;

  int32_t addWord(const std::string& w) {
    auto it = vocab.find(w);
    if (it != vocab.end()) {
      return it->second;
    }
    int32_t id = (int32_t)words.size();
    vocab.emplace(w, id);
    words.push_back(w);
    return id;
  }

  /* KenLM vocab->Index(): unknown strings map to <unk> == 0. */
  int32_t index(const std::string& w) const {
    auto it = vocab.find(w);
    return it == vocab.end() ? unk : it->second;
  }

  void load(const std::string& path) {
    std::ifstream in(path);
    if (!in) {
      throw std::runtime_error("[ArpaModel] cannot open " + path);
    }
    addWord("<unk>"); // KenLM reserves index 0 for <unk>
    std::string line;
    std::vector<size_t> counts;
    while (std::getline(in, line)) {
      if (line.rfind("ngram ", 0) == 0) {
        auto eq = line.find('=');
        counts.push_back((size_t)std::strtoull(line.c_str() + eq + 1, nullptr, 10));
      } else if (line.rfind("\\1-grams:", 0) == 0) {
        break;
      }
    }
    order = (int)counts.size();
    table.resize(order);
    int cur = 1;
    std::vector<std::string> f;
    while (std::getline(in, line)) {
      if (line.empty()) {
        continue;
      }
      if (line[0] == '\\') {
        if (line.rfind("\\end\\", 0) == 0) {
          break;
        }
        cur = std::atoi(line.c_str() + 1);
        continue;
      }
      f.clear();
      size_t p = 0;
      while (p < line.size()) {
        size_t q = line.find_first_of(" \t", p);
        if (q == std::string::npos) {
          q = line.size();
        }
        if (q > p) {
          f.emplace_back(line.substr(p, q - p));
        }
        p = q + 1;
      }
      if ((int)f.size() < cur + 1) {
        continue;
      }
      Entry e;
      e.prob = std::strtof(f[0].c_str(), nullptr);
      e.backoff = ((int)f.size() > cur + 1) ? std::strtof(f[cur + 1].c_str(), nullptr) : 0.0f;
      std::vector<int32_t> key(cur);
      for (int i = 0; i < cur; ++i) {
        key[i] = (cur == 1) ? addWord(f[1 + i]) : index(f[1 + i]);
      }
      table[cur - 1][key] = e;
    }
    bos = index("<s>");
    eos = index("</s>");
  }

  /* ctx: LM ids, oldest first, at most order-1 of them.  Returns log10 p and
   * writes the successor context. */
  float score(const std::vector<int32_t>& ctx, int32_t w,
              std::vector<int32_t>& out) const {
    int L = (int)ctx.size();
    if (L > order - 1) {
      L = order - 1;
    }
    float prob = 0.0f;
    int matched = 0; // number of context words in the longest match
    std::vector<int32_t> key;
    bool found = false;
    for (int k = L; k >= 0; --k) {
      key.assign(ctx.end() - k, ctx.end());
      key.push_back(w);
      auto it = table[k].find(key);
      if (it != table[k].end()) {
        prob = it->second.prob;
        matched = k;
        found = true;
        break;
      }
    }
    if (!found) { // word unseen even as unigram: score <unk>
      key.assign(1, unk);
      auto it = table[0].find(key);
      prob = it == table[0].end() ? -100.0f : it->second.prob;
      matched = 0;
    }
    // back-off weights of skipped contexts, shortest first
    for (int j = matched + 1; j <= L; ++j) {
      key.assign(ctx.end() - j, ctx.end());
      auto it = table[j - 1].find(key);
      if (it != table[j - 1].end()) {
        prob += it->second.backoff;
      }
    }
    out = ctx;
    out.push_back(found ? w : unk);
    if ((int)out.size() > order - 1) {
      out.erase(out.begin(), out.end() - (order - 1));
    }
    return prob;
  }
};

} // namespace orc
