/*
 * fltx_host_trie.cpp -- host-side lexicon trie with the reference semantics
 * (flashlight/lib/text/decoder/Trie.h:30-92, Trie.cpp:20-101) and its
 * flattening into the HBM layout the kernels gather from.
 *
 * Setup code: runs once per lexicon on the CPU, exactly like the reference.
 * The smear is done here and never on the device because its result depends on
 * float narrowing after every step (Trie.h:54) and, for LOGADD, on the child
 * visiting order of a std::unordered_map<int, ...> (Trie.cpp:84-94); the same
 * container type and insertion sequence are used so the order is reproduced.
 */
#include <cmath>
#include <cstdint>
#include <limits>
#include <unordered_map>
#include <utility>
#include <vector>

#include "fltx.h"

namespace {
constexpr int kMaxLabel = 6;            /* kTrieMaxLabel, Trie.h:19 */
constexpr double kMinusLogThr = -39.14; /* kMinusLogThreshold, Trie.cpp:20 */

struct HNode {
  std::unordered_map<int, int32_t> kids;
  int token = 0;
  std::vector<int32_t> labels;
  std::vector<float> scores;
  float maxScore = 0.0f;
};

double logAdd(double a, double b) { /* TrieLogAdd, Trie.cpp:66-77 */
  if (a < b) {
    std::swap(a, b);
  }
  const double d = b - a;
  return d < kMinusLogThr ? a : a + std::log1p(std::exp(d));
}
} // namespace

struct fltx_htrie {
  std::vector<HNode> nodes;
  int maxChildren;

  void smearNode(int32_t id, int mode) { /* smearNode, Trie.cpp:79-95 */
    nodes[id].maxScore = -std::numeric_limits<float>::infinity();
    for (float s : nodes[id].scores) {
      nodes[id].maxScore = (float)logAdd(nodes[id].maxScore, s);
    }
    for (const auto& kv : nodes[id].kids) {
      const int32_t c = kv.second;
      smearNode(c, mode);
      if (mode == FLTX_SMEAR_LOGADD) {
        nodes[id].maxScore = (float)logAdd(nodes[id].maxScore, nodes[c].maxScore);
      } else if (mode == FLTX_SMEAR_MAX && nodes[c].maxScore > nodes[id].maxScore) {
        nodes[id].maxScore = nodes[c].maxScore;
      }
    }
  }
};

/* defined in fltx_api.cpp */
extern "C" int fltx_set_error_(int code, const char* msg);

extern "C" {

int fltx_htrie_create(int32_t maxChildren, int32_t rootIdx, fltx_htrie** out) {
  if (!out || maxChildren <= 0) {
    return fltx_set_error_(FLTX_ERR_INVALID, "fltx_htrie_create: bad argument");
  }
  auto* t = new fltx_htrie();
  t->maxChildren = maxChildren;
  t->nodes.emplace_back();
  t->nodes[0].token = rootIdx;
  *out = t;
  return FLTX_OK;
}

int fltx_htrie_destroy(fltx_htrie* t) {
  delete t;
  return FLTX_OK;
}

int fltx_htrie_insert(fltx_htrie* t, const int32_t* idx, int32_t n, int32_t label, float score) {
  if (!t || (!idx && n > 0)) {
    return fltx_set_error_(FLTX_ERR_INVALID, "fltx_htrie_insert: null argument");
  }
  for (int i = 0; i < n; ++i) { /* validate first: the reference throws mid-way, but
                                   leaves only label-free nodes behind */
    if (idx[i] < 0 || idx[i] >= t->maxChildren) {
      return fltx_set_error_(FLTX_ERR_RANGE, "[Trie] Invalid letter index");
    }
  }
  int32_t node = 0;
  for (int i = 0; i < n; ++i) {
    auto it = t->nodes[node].kids.find(idx[i]);
    if (it == t->nodes[node].kids.end()) {
      const int32_t id = (int32_t)t->nodes.size();
      t->nodes.emplace_back();
      t->nodes[id].token = idx[i];
      t->nodes[node].kids[idx[i]] = id;
      node = id;
    } else {
      node = it->second;
    }
  }
  if ((int)t->nodes[node].labels.size() < kMaxLabel) {
    t->nodes[node].labels.push_back(label);
    t->nodes[node].scores.push_back(score);
  }
  return FLTX_OK;
}

int fltx_htrie_search(fltx_htrie* t, const int32_t* idx, int32_t n, int32_t* found, float* maxScore,
                      int32_t* nLabels, int32_t* labels, float* scores) {
  if (!t || !found || (!idx && n > 0)) {
    return fltx_set_error_(FLTX_ERR_INVALID, "fltx_htrie_search: null argument");
  }
  int32_t node = 0;
  *found = 0;
  for (int i = 0; i < n; ++i) {
    if (idx[i] < 0 || idx[i] >= t->maxChildren) {
      return fltx_set_error_(FLTX_ERR_RANGE, "[Trie] Invalid letter index");
    }
    auto it = t->nodes[node].kids.find(idx[i]);
    if (it == t->nodes[node].kids.end()) {
      return FLTX_OK;
    }
    node = it->second;
  }
  *found = 1;
  const HNode& nd = t->nodes[node];
  if (maxScore) {
    *maxScore = nd.maxScore;
  }
  if (nLabels) {
    *nLabels = (int32_t)nd.labels.size();
  }
  for (size_t i = 0; i < nd.labels.size(); ++i) {
    if (labels) {
      labels[i] = nd.labels[i];
    }
    if (scores) {
      scores[i] = nd.scores[i];
    }
  }
  return FLTX_OK;
}

int fltx_htrie_smear(fltx_htrie* t, int32_t mode) {
  if (!t || mode < 0 || mode > 2) {
    return fltx_set_error_(FLTX_ERR_INVALID, "fltx_htrie_smear: bad argument");
  }
  if (mode != FLTX_SMEAR_NONE) {
    t->smearNode(0, mode);
  }
  return FLTX_OK;
}

int fltx_htrie_num_nodes(fltx_htrie* t, int64_t* n) {
  if (!t || !n) {
    return fltx_set_error_(FLTX_ERR_INVALID, "null argument");
  }
  *n = (int64_t)t->nodes.size();
  return FLTX_OK;
}

int fltx_htrie_node(fltx_htrie* t, int64_t node, int32_t* token, float* maxScore, int32_t* nLabels,
                    int32_t* labels, float* scores, int32_t* nChildren, int32_t* childTokens,
                    int64_t* childNodes, int32_t childCapacity) {
  if (!t) {
    return fltx_set_error_(FLTX_ERR_INVALID, "fltx_htrie_node: null trie");
  }
  if (node < 0 || node >= (int64_t)t->nodes.size()) {
    return fltx_set_error_(FLTX_ERR_RANGE, "fltx_htrie_node: node out of range");
  }
  const HNode& nd = t->nodes[(size_t)node];
  if (token) {
    *token = nd.token;
  }
  if (maxScore) {
    *maxScore = nd.maxScore;
  }
  if (nLabels) {
    *nLabels = (int32_t)nd.labels.size();
  }
  for (size_t i = 0; i < nd.labels.size(); ++i) {
    if (labels) {
      labels[i] = nd.labels[i];
    }
    if (scores) {
      scores[i] = nd.scores[i];
    }
  }
  if (nChildren) {
    *nChildren = (int32_t)nd.kids.size();
  }
  if (childTokens || childNodes) {
    if ((int32_t)nd.kids.size() > childCapacity) {
      return fltx_set_error_(FLTX_ERR_RANGE, "fltx_htrie_node: child_capacity too small");
    }
    int32_t i = 0;
    for (const auto& kv : nd.kids) {
      if (childTokens) {
        childTokens[i] = kv.first;
      }
      if (childNodes) {
        childNodes[i] = kv.second;
      }
      ++i;
    }
  }
  return FLTX_OK;
}

int fltx_htrie_upload(fltx_htrie* t, fltx_ctx* ctx, fltx_trie** out) {
  if (!t || !ctx || !out) {
    return fltx_set_error_(FLTX_ERR_INVALID, "fltx_htrie_upload: null argument");
  }
  const size_t nn = t->nodes.size();
  const int N = t->maxChildren;
  std::vector<int32_t> child(nn * (size_t)N, -1), labOff(nn + 1, 0), labels;
  std::vector<float> maxScore(nn);
  for (size_t i = 0; i < nn; ++i) {
    const HNode& nd = t->nodes[i];
    for (const auto& kv : nd.kids) {
      child[i * N + kv.first] = kv.second;
    }
    maxScore[i] = nd.maxScore;
    labOff[i] = (int32_t)labels.size();
    labels.insert(labels.end(), nd.labels.begin(), nd.labels.end());
  }
  labOff[nn] = (int32_t)labels.size();
  return fltx_trie_create(ctx, (int64_t)nn, N, child.data(), maxScore.data(), labOff.data(),
                          labels.empty() ? nullptr : labels.data(), out);
}

} // extern "C"
