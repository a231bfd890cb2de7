/*
 * dictionary/Utils.h -- lexicon loading for the decoder path: the step right
 * before the hot path.  Same functions and results as
 * flashlight/lib/text/dictionary/Utils.cpp:19-62 (createWordDict, loadWords),
 * :64-88 (splitWrd), :90-124 (packReplabels), :126-150 (unpackReplabels), :152-162 (tkn2Idx) and the file
 * constructor of Dictionary (Dictionary.cpp:24-60); written for this tree.
 *
 * One thing is dictated by the reference: word ids are assigned in the iteration
 * order of the LexiconMap (an unordered_map keyed by the word, Utils.cpp:19-26),
 * so the map type, its key insertion sequence (first sight of a word in the file,
 * "<unk>" last) and nothing else about it must match for the ids to come out
 * identical on the same standard library.  tests/test_cpp_facade.py checks the
 * result against the reference's own dump, byte for byte.
 */
#pragma once
#include <fstream>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "flashlight/lib/text/dictionary/Dictionary.h"

namespace fl {
namespace lib {
namespace text {

constexpr const char* kUnkToken = "<unk>";

using LexiconMap = std::unordered_map<std::string, std::vector<std::vector<std::string>>>;

namespace detail {

/* the whitespace-separated fields of one text line, without copying the line */
class Fields {
 public:
  explicit Fields(const std::string& line) : s_(line) { advance(0); }
  bool done() const { return begin_ == std::string::npos; }
  std::string next() {
    std::string f = s_.substr(begin_, end_ - begin_);
    advance(end_);
    return f;
  }

 private:
  static bool blank(char c) { return c == ' ' || (c >= '\t' && c <= '\r'); }
  void advance(size_t from) {
    begin_ = from;
    while (begin_ < s_.size() && blank(s_[begin_])) {
      ++begin_;
    }
    if (begin_ >= s_.size()) {
      begin_ = end_ = std::string::npos;
      return;
    }
    end_ = begin_;
    while (end_ < s_.size() && !blank(s_[end_])) {
      ++end_;
    }
  }
  const std::string& s_;
  size_t begin_ = 0, end_ = 0;
};

/* calls onLine(line) for the lines of a text file until it returns false */
template <class F>
void forEachLine(const std::string& filename, const char* what, bool invalidArgument, F&& onLine) {
  std::ifstream in(filename);
  if (!in) {
    if (invalidArgument) {
      throw std::invalid_argument(std::string(what) + filename);
    }
    throw std::runtime_error(std::string(what) + filename);
  }
  for (std::string line; std::getline(in, line);) {
    if (!onLine(line)) {
      break;
    }
  }
}

} // namespace detail

/* Dictionary(filename): all the entries of line i get index i (Dictionary.cpp:40-60);
 * empty lines do not count */
inline Dictionary loadDictionary(const std::string& filename) {
  Dictionary dict;
  detail::forEachLine(filename, "Dictionary - cannot open file  ", false, [&](const std::string& line) {
    if (!line.empty()) {
      const int lineIndex = (int)dict.indexSize();
      for (detail::Fields f(line); !f.done();) {
        dict.addEntry(f.next(), lineIndex);
      }
    }
    return true;
  });
  if (!dict.isContiguous()) { /* (Dictionary.cpp:57-59) */
    throw std::runtime_error("Invalid dictionary format - not contiguous");
  }
  return dict;
}

/* "word tok tok ..." per line -> word -> list of spellings; reading stops once maxWords
 * distinct words are held; "<unk>" is added with no spelling (Utils.cpp:28-62) */
inline LexiconMap loadWords(const std::string& filename, int maxWords = -1) {
  LexiconMap lexicon;
  {
    detail::forEachLine(filename, "text::loadWords - can't open file ", true, [&](const std::string& line) {
      if (maxWords == (int)lexicon.size()) {
        return false;
      }
      detail::Fields f(line);
      std::vector<std::string> spelling;
      std::string word;
      if (!f.done()) {
        word = f.next();
      }
      while (!f.done()) {
        spelling.push_back(f.next());
      }
      if (spelling.empty()) { /* fewer than two fields */
        throw std::runtime_error("[loadWords] Invalid line: " + line);
      }
      lexicon[word].push_back(std::move(spelling)); /* operator[] inserts the key on first sight */
      return true;
    });
  }
  lexicon[kUnkToken] = {};
  return lexicon;
}

/* word ids in the map's iteration order; unknown words map to <unk> (Utils.cpp:19-26) */
inline Dictionary createWordDict(const LexiconMap& lexicon) {
  Dictionary dict;
  for (const auto& entry : lexicon) {
    dict.addEntry(entry.first);
  }
  dict.setDefaultIndex(dict.getIndex(kUnkToken));
  return dict;
}

/* the UTF-8 characters of a word (Utils.cpp:64-88) */
inline std::vector<std::string> splitWrd(const std::string& word) {
  std::vector<std::string> chars;
  for (size_t i = 0; i < word.size();) {
    /* length of the sequence = number of leading one bits of its first byte (none: ASCII) */
    const unsigned lead = static_cast<unsigned char>(word[i]);
    int ones = 0;
    while (ones < 5 && (lead & (0x80u >> ones))) {
      ++ones;
    }
    const size_t n = ones == 0 ? 1 : (size_t)ones;
    if (ones == 1 || ones > 4 || i + n > word.size()) {
      throw std::runtime_error("splitWrd: invalid UTF-8 : " + word);
    }
    chars.push_back(word.substr(i, n));
    i += n;
  }
  return chars;
}

/* replabels: a run of r + 1 equal tokens becomes the token followed by "<r>", runs longer
 * than maxReps + 1 restart ("abbccc" -> "a b <1> c <2>"; Utils.cpp:90-124) */
inline std::vector<int> packReplabels(const std::vector<int>& tokens, const Dictionary& dict, int maxReps) {
  if (tokens.empty() || maxReps <= 0) {
    return tokens;
  }
  std::vector<int> label((size_t)maxReps + 1, -1); /* label[r] = index of "<r>" */
  for (int r = 1; r <= maxReps; ++r) {
    label[(size_t)r] = dict.getIndex("<" + std::to_string(r) + ">");
  }
  std::vector<int> out;
  for (size_t i = 0; i < tokens.size();) {
    size_t run = 1;
    while (i + run < tokens.size() && tokens[i + run] == tokens[i]) {
      ++run;
    }
    for (size_t left = run; left > 0;) { /* pieces of at most maxReps + 1 */
      const size_t piece = left < (size_t)maxReps + 1 ? left : (size_t)maxReps + 1;
      out.push_back(tokens[i]);
      if (piece > 1) {
        out.push_back(label[piece - 1]);
      }
      left -= piece;
    }
    i += run;
  }
  return out;
}

/* the inverse: "<r>" after a token stands for r more copies of it; a replabel with nothing to repeat
 * (at the start, or right after another replabel) is dropped; indices beyond maxReps are ordinary tokens
 * (Utils.cpp:126-150; DictionaryTest.cpp:103-147 are the known answers) */
inline std::vector<int> unpackReplabels(const std::vector<int>& tokens, const Dictionary& dict, int maxReps) {
  if (tokens.empty() || maxReps <= 0) {
    return tokens;
  }
  std::vector<std::pair<int, int>> reps; /* (index of "<r>", r); the smallest r wins when two share an index */
  for (int r = 1; r <= maxReps; ++r) {
    const int idx = dict.getIndex("<" + std::to_string(r) + ">");
    bool known = false;
    for (const auto& p : reps) {
      known = known || p.first == idx;
    }
    if (!known) {
      reps.emplace_back(idx, r);
    }
  }
  std::vector<int> out;
  bool canRepeat = false; /* the last thing seen was an ordinary token */
  for (int tok : tokens) {
    int r = 0;
    for (const auto& p : reps) {
      if (p.first == tok) {
        r = p.second;
      }
    }
    if (r == 0) {
      out.push_back(tok);
      canRepeat = true;
    } else {
      /* (a token index of -1 cannot be repeated in the reference either: it is its "nothing pending" mark) */
      if (canRepeat && out.back() != -1) {
        out.insert(out.end(), (size_t)r, out.back());
      }
      canRepeat = false;
    }
  }
  return out;
}

/* spelling -> token indices with replabels (Utils.cpp:152-162) */
inline std::vector<int> tkn2Idx(const std::vector<std::string>& spelling, const Dictionary& tokenDict,
                                int maxReps) {
  std::vector<int> idx(spelling.size());
  for (size_t i = 0; i < spelling.size(); ++i) {
    idx[i] = tokenDict.getIndex(spelling[i]);
  }
  return packReplabels(idx, tokenDict, maxReps);
}

} // namespace text
} // namespace lib
} // namespace fl
