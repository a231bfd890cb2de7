/*
 * dictionary/Utils.h -- lexicon loading for the decoder path: the step right
 * before the hot path.  Restates
 * flashlight/lib/text/dictionary/Utils.cpp:19-62 (createWordDict, loadWords),
 * :64-88 (splitWrd), :90-124 (packReplabels), :152-162 (tkn2Idx) and the
 * file constructor of Dictionary (Dictionary.cpp:24-60).
 *
 * createWordDict assigns word ids in the iteration order of the LexiconMap (an
 * unordered_map): the same container type and insertion sequence are used here
 * so the ids come out identical on the same standard library
 * (tests/test_cpp_facade.py checks against the reference's own dump).
 */
#pragma once
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "flashlight/lib/text/dictionary/Dictionary.h"

namespace fl {
namespace lib {
namespace text {

constexpr const char* kUnkToken = "<unk>";

using LexiconMap = std::unordered_map<std::string, std::vector<std::vector<std::string>>>;

namespace detail {
inline std::vector<std::string> splitWs(const std::string& line) {
  std::vector<std::string> out;
  size_t p = 0;
  const std::string ws = "\t\n\v\f\r ";
  while (p < line.size()) {
    const size_t a = line.find_first_not_of(ws, p);
    if (a == std::string::npos) {
      break;
    }
    size_t b = line.find_first_of(ws, a);
    if (b == std::string::npos) {
      b = line.size();
    }
    out.emplace_back(line, a, b - a);
    p = b;
  }
  return out;
}
} // namespace detail

/* Dictionary(filename): every whitespace-separated entry of a line maps to the
 * line's index (Dictionary.cpp:40-60) */
inline Dictionary loadDictionary(const std::string& filename) {
  std::ifstream in(filename);
  if (!in) {
    throw std::runtime_error("Dictionary - cannot open file  " + filename);
  }
  Dictionary d;
  std::string line;
  while (std::getline(in, line)) {
    if (line.empty()) {
      continue;
    }
    const int idx = (int)d.indexSize();
    for (const auto& t : detail::splitWs(line)) {
      d.addEntry(t, idx);
    }
  }
  return d;
}

/* Utils.cpp:28-62 */
inline LexiconMap loadWords(const std::string& filename, int maxWords = -1) {
  LexiconMap lexicon;
  std::ifstream in(filename);
  if (!in) {
    throw std::invalid_argument("text::loadWords - can't open file " + filename);
  }
  std::string line;
  while (maxWords != (int)lexicon.size() && std::getline(in, line)) {
    auto fields = detail::splitWs(line);
    if (fields.size() < 2) {
      throw std::runtime_error("[loadWords] Invalid line: " + line);
    }
    const std::string word = fields[0];
    std::vector<std::string> spelling(fields.begin() + 1, fields.end());
    if (lexicon.find(word) == lexicon.end()) {
      lexicon[word] = {};
    }
    lexicon[word].push_back(spelling);
  }
  lexicon[kUnkToken] = {};
  return lexicon;
}

/* Utils.cpp:19-26 */
inline Dictionary createWordDict(const LexiconMap& lexicon) {
  Dictionary dict;
  for (const auto& it : lexicon) {
    dict.addEntry(it.first);
  }
  dict.setDefaultIndex(dict.getIndex(kUnkToken));
  return dict;
}

/* Utils.cpp:64-88: split a word into UTF-8 characters */
inline std::vector<std::string> splitWrd(const std::string& word) {
  std::vector<std::string> tokens;
  const int len = (int)word.length();
  for (int i = 0; i < len;) {
    const auto c = static_cast<unsigned char>(word[i]);
    int n = -1;
    if ((c & 0x80) == 0) {
      n = 1;
    } else if ((c & 0xE0) == 0xC0) {
      n = 2;
    } else if ((c & 0xF0) == 0xE0) {
      n = 3;
    } else if ((c & 0xF8) == 0xF0) {
      n = 4;
    }
    if (n == -1 || i + n > len) {
      throw std::runtime_error("splitWrd: invalid UTF-8 : " + word);
    }
    tokens.emplace_back(word.begin() + i, word.begin() + i + n);
    i += n;
  }
  return tokens;
}

/* Utils.cpp:90-124: "abbccc" -> "ab1c2" */
inline std::vector<int> packReplabels(const std::vector<int>& tokens, const Dictionary& dict, int maxReps) {
  if (tokens.empty() || maxReps <= 0) {
    return tokens;
  }
  std::vector<int> repIdx(maxReps + 1);
  for (int i = 1; i <= maxReps; ++i) {
    repIdx[i] = dict.getIndex("<" + std::to_string(i) + ">");
  }
  std::vector<int> result;
  int prev = -1, reps = 0;
  for (int t : tokens) {
    if (t == prev && reps < maxReps) {
      ++reps;
    } else {
      if (reps > 0) {
        result.push_back(repIdx[reps]);
        reps = 0;
      }
      result.push_back(t);
      prev = t;
    }
  }
  if (reps > 0) {
    result.push_back(repIdx[reps]);
  }
  return result;
}

/* Utils.cpp:152-162 */
inline std::vector<int> tkn2Idx(const std::vector<std::string>& spelling, const Dictionary& tokenDict,
                                int maxReps) {
  std::vector<int> ret;
  ret.reserve(spelling.size());
  for (const auto& t : spelling) {
    ret.push_back(tokenDict.getIndex(t));
  }
  return packReplabels(ret, tokenDict, maxReps);
}

} // namespace text
} // namespace lib
} // namespace fl
