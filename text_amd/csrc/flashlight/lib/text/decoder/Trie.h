/*
 * decoder/Trie.h -- Trie / TrieNode with the reference's interface
 * (flashlight/lib/text/decoder/Trie.h:19-94), backed by the host trie of the C
 * ABI (fltx_htrie_*), which restates Trie.cpp:26-101 and flattens into HBM.
 * insert()/search() return snapshots of the node (idx, labels, scores,
 * maxScore); `children` is not materialised on the host side of the facade.
 */
#pragma once
#include <memory>
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
  Trie(int maxChildren, int rootIdx) : root_(std::make_shared<TrieNode>(rootIdx)) {
    detail::check(fltx_htrie_create(maxChildren, rootIdx, &h_));
  }
  ~Trie() {
    if (dev_) {
      fltx_trie_destroy(dev_);
    }
    fltx_htrie_destroy(h_);
  }
  Trie(const Trie&) = delete;
  Trie& operator=(const Trie&) = delete;

  const TrieNode* getRoot() const { return root_.get(); }

  TrieNodePtr insert(const std::vector<int>& indices, int label, float score) {
    std::vector<int32_t> idx(indices.begin(), indices.end());
    detail::check(fltx_htrie_insert(h_, idx.data(), (int32_t)idx.size(), label, score)); /* out_of_range */
    dirty_ = true;
    return search(indices);
  }

  TrieNodePtr search(const std::vector<int>& indices) {
    std::vector<int32_t> idx(indices.begin(), indices.end());
    int32_t found = 0, n = 0, labels[kTrieMaxLabel];
    float ms = 0, scores[kTrieMaxLabel];
    detail::check(fltx_htrie_search(h_, idx.data(), (int32_t)idx.size(), &found, &ms, &n, labels, scores));
    if (!found) {
      return nullptr;
    }
    auto node = std::make_shared<TrieNode>(indices.empty() ? root_->idx : indices.back());
    node->labels.assign(labels, labels + n);
    node->scores.assign(scores, scores + n);
    node->maxScore = ms;
    return node;
  }

  void smear(const SmearingMode smearMode) {
    detail::check(fltx_htrie_smear(h_, (int32_t)smearMode));
    dirty_ = true;
  }

  /* additive: the flattened trie in HBM (uploaded on first use, re-uploaded
   * after later insert()/smear() calls) */
  const fltx_trie* deviceHandle(fltx_ctx* ctx) const {
    if (dirty_ || !dev_) {
      if (dev_) {
        fltx_trie_destroy(dev_);
        dev_ = nullptr;
      }
      detail::check(fltx_htrie_upload(h_, ctx, &dev_));
      dirty_ = false;
    }
    return dev_;
  }

 private:
  TrieNodePtr root_;
  fltx_htrie* h_ = nullptr;
  mutable fltx_trie* dev_ = nullptr;
  mutable bool dirty_ = true;
};

using TriePtr = std::shared_ptr<Trie>;

} // namespace text
} // namespace lib
} // namespace fl
