/*
 * dictionary/Dictionary.h -- fl::lib::text::Dictionary
 * (flashlight/lib/text/dictionary/Dictionary.{h,cpp}) as the decoder path and its
 * Python surface use it: a string <-> index bimap with a default index (the file /
 * stream constructors are loadDictionary in dictionary/Utils.h).
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
  /* no hole in 0 .. indexSize() - 1 and every entry's index is known (Dictionary.cpp:125-137) */
  bool isContiguous() const {
    const size_t n = indexSize();
    for (size_t i = 0; i < n; ++i) {
      if (idx2entry_.count((int)i) == 0) {
        return false;
      }
    }
    for (const auto& kv : entry2idx_) {
      if (idx2entry_.count(kv.second) == 0) {
        return false;
      }
    }
    return true;
  }
  std::vector<int> mapEntriesToIndices(const std::vector<std::string>& entries) const {
    std::vector<int> out;
    out.reserve(entries.size());
    for (const auto& e : entries) {
      out.push_back(getIndex(e));
    }
    return out;
  }
  std::vector<std::string> mapIndicesToEntries(const std::vector<int>& indices) const {
    std::vector<std::string> out;
    out.reserve(indices.size());
    for (int i : indices) {
      out.push_back(getEntry(i));
    }
    return out;
  }

 private:
  std::unordered_map<std::string, int> entry2idx_;
  std::unordered_map<int, std::string> idx2entry_;
  int defaultIndex_ = -1;
};

} // namespace text
} // namespace lib
} // namespace fl
