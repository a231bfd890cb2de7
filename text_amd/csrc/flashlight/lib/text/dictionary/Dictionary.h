/*
 * dictionary/Dictionary.h -- the subset of fl::lib::text::Dictionary
 * (flashlight/lib/text/dictionary/Dictionary.{h,cpp}) that the KenLM adapter
 * needs: a string <-> index bimap with a default index.
 */
#pragma once
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

namespace fl {
namespace lib {
namespace text {

class Dictionary {
 public:
  Dictionary() = default;
  explicit Dictionary(const std::vector<std::string>& tkns) {
    for (const auto& t : tkns) {
      addEntry(t);
    }
  }
  size_t entrySize() const { return entry2idx_.size(); }
  size_t indexSize() const { return idx2entry_.size(); }
  void addEntry(const std::string& entry, int idx) {
    if (entry2idx_.find(entry) != entry2idx_.end()) {
      throw std::invalid_argument("Duplicate entry name in dictionary '" + entry + "'");
    }
    entry2idx_[entry] = idx;
    if (idx2entry_.find(idx) == idx2entry_.end()) {
      idx2entry_[idx] = entry;
    }
  }
  void addEntry(const std::string& entry) {
    int idx = (int)idx2entry_.size();
    while (idx2entry_.find(idx) != idx2entry_.end()) {
      ++idx;
    }
    addEntry(entry, idx);
  }
  std::string getEntry(int idx) const {
    auto it = idx2entry_.find(idx);
    if (it == idx2entry_.end()) {
      throw std::invalid_argument("Unknown index in dictionary '" + std::to_string(idx) + "'");
    }
    return it->second;
  }
  void setDefaultIndex(int idx) { defaultIndex_ = idx; }
  int getIndex(const std::string& entry) const {
    auto it = entry2idx_.find(entry);
    if (it == entry2idx_.end()) {
      if (defaultIndex_ < 0) {
        throw std::invalid_argument("Unknown entry in dictionary: '" + entry + "'");
      }
      return defaultIndex_;
    }
    return it->second;
  }
  bool contains(const std::string& entry) const { return entry2idx_.find(entry) != entry2idx_.end(); }

 private:
  std::unordered_map<std::string, int> entry2idx_;
  std::unordered_map<int, std::string> idx2entry_;
  int defaultIndex_ = -1;
};

} // namespace text
} // namespace lib
} // namespace fl
