/*
 * decoder/Trie.h -- Trie / TrieNode with the reference's interface
 * (flashlight/lib/text/decoder/Trie.h:19-94), backed by the host trie of the C
 * ABI (fltx_htrie_*), which restates Trie.cpp:26-101 and flattens into HBM.
 * getRoot()/search() expose the node tree (children, labels, scores, maxScore)
 * as the reference does; it is materialised from the host trie on first use.
 */
#pragma once
#include <iterator>
#include <map>
#include <memory>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "flashlight/lib/text/Defines.h"
#include "flashlight/lib/text/decoder/Fltx.h"

namespace fl {
namespace lib {
namespace text {

constexpr int kTrieMaxLabel = 6;

enum class SmearingMode { NONE = 0, MAX = 1, LOGADD = 2 };

struct TrieNode {
  explicit TrieNode(int idx) : idx(idx), maxScore(0) {}
  std::unordered_map<int, std::shared_ptr<TrieNode>> children;
  int idx;
  std::vector<int> labels;
  std::vector<float> scores;
  float maxScore;
};
using TrieNodePtr = std::shared_ptr<TrieNode>;

class FL_TEXT_API Trie {
 public:
  Trie(int maxChildren, int rootIdx) : maxChildren_(maxChildren), root_(std::make_shared<TrieNode>(rootIdx)) {
    detail::check(fltx_htrie_create(maxChildren, rootIdx, &h_));
  }
  ~Trie() { fltx_htrie_destroy(h_); }
  Trie(const Trie&) = delete;
  Trie& operator=(const Trie&) = delete;

  /* the node tree (children, labels, scores, maxScore) as the host trie holds it now; built on
   * first use after a modification (O(nodes)) and shared with search() */
  const TrieNode* getRoot() const {
    materialise();
    return root_.get();
  }

  /* returns a snapshot of the node the word ends in (idx, labels, scores, maxScore; children left
   * empty: building a lexicon must not pay for the node tree -- use search() for that) */
  TrieNodePtr insert(const std::vector<int>& indices, int label, float score) {
    std::vector<int32_t> idx(indices.begin(), indices.end());
    detail::check(fltx_htrie_insert(h_, idx.data(), (int32_t)idx.size(), label, score)); /* out_of_range */
    dirty_ = true;
    treeStale_ = true;
    int32_t found = 0, n = 0, labels[kTrieMaxLabel];
    float ms = 0, scores[kTrieMaxLabel];
    detail::check(fltx_htrie_search(h_, idx.data(), (int32_t)idx.size(), &found, &ms, &n, labels, scores));
    auto node = std::make_shared<TrieNode>(indices.empty() ? root_->idx : indices.back());
    node->labels.assign(labels, labels + n);
    node->scores.assign(scores, scores + n);
    node->maxScore = ms;
    return node;
  }

  TrieNodePtr search(const std::vector<int>& indices) {
    materialise();
    TrieNodePtr node = root_;
    for (int idx : indices) {
      auto it = node->children.find(idx);
      if (it == node->children.end()) {
        return nullptr;
      }
      node = it->second;
    }
    return node;
  }

  void smear(const SmearingMode smearMode) {
    detail::check(fltx_htrie_smear(h_, (int32_t)smearMode));
    dirty_ = true;
    treeStale_ = true;
  }

  /* additive: the host trie behind this object (fltx_group_create replicates it per device) */
  fltx_htrie* hostHandle() const { return h_; }

  /* additive: the flattened trie in HBM.  Uploaded on first use; insert()/smear() afterwards make
   * the next call upload a NEW copy -- decoders created earlier keep the one they were built with
   * (each holds a reference), so a shared Trie stays valid for every decoder. */
  std::shared_ptr<const fltx_trie> deviceHandle(fltx_ctx* ctx) const {
    std::lock_guard<std::mutex> lock(devMu_); /* (decoders of several threads may be built over one Trie) */
    /* one copy per CONTEXT (the facade's contexts are per thread), keyed by the context's uid -- an address may be handed
     * out again after a thread's context died -- and held weakly: the decoders own their copy, the Trie only remembers it
     * for the next decoder built on the same context */
    if (dirty_) {
      dev_.clear();
      last_.reset();
      dirty_ = false;
    }
    const uint64_t uid = fltx_ctx_uid(ctx);
    auto it = dev_.find(uid);
    if (it != dev_.end()) {
      if (auto alive = it->second.lock()) {
        return alive;
      }
      dev_.erase(it);
    }
    for (auto j = dev_.begin(); j != dev_.end();) { /* (copies whose decoders are all gone) */
      j = j->second.expired() ? dev_.erase(j) : std::next(j);
    }
    fltx_trie* t = nullptr;
    detail::check(fltx_htrie_upload(h_, ctx, &t));
    std::shared_ptr<const fltx_trie> up(t, [](const fltx_trie* p) { fltx_trie_destroy(const_cast<fltx_trie*>(p)); });
    dev_[uid] = up;
    last_ = up; /* (the newest copy stays alive with the Trie: build decoder, drop it, build another -- no second upload) */
    return up;
  }

 private:
  void materialise() const {
    if (!treeStale_) {
      return;
    }
    std::vector<int32_t> toks((size_t)maxChildren_);
    std::vector<int64_t> ids((size_t)maxChildren_);
    /* (node id in the host trie, TrieNode to fill) -- iterative: words can be long */
    std::vector<std::pair<int64_t, TrieNode*>> stack;
    root_->children.clear();
    stack.emplace_back(0, root_.get());
    while (!stack.empty()) {
      const auto cur = stack.back();
      stack.pop_back();
      int32_t tok = 0, nl = 0, nc = 0, labels[kTrieMaxLabel];
      float ms = 0, scores[kTrieMaxLabel];
      detail::check(fltx_htrie_node(h_, cur.first, &tok, &ms, &nl, labels, scores, &nc, toks.data(), ids.data(),
                                    maxChildren_));
      TrieNode* nd = cur.second;
      nd->labels.assign(labels, labels + nl);
      nd->scores.assign(scores, scores + nl);
      nd->maxScore = ms;
      for (int i = 0; i < nc; ++i) {
        auto child = std::make_shared<TrieNode>(toks[(size_t)i]);
        nd->children[toks[(size_t)i]] = child;
        stack.emplace_back(ids[(size_t)i], child.get());
      }
    }
    treeStale_ = false;
  }

  int maxChildren_;
  TrieNodePtr root_;
  fltx_htrie* h_ = nullptr;
  mutable std::mutex devMu_;
  mutable std::map<uint64_t, std::weak_ptr<const fltx_trie>> dev_;
  mutable std::shared_ptr<const fltx_trie> last_;
  mutable bool dirty_ = true;
  mutable bool treeStale_ = true;
};

using TriePtr = std::shared_ptr<Trie>;

} // namespace text
} // namespace lib
} // namespace fl
