/*
 * fltx_arpa.cpp -- ARPA text model -> flat n-gram tables (host side, runs once).
 *
 * Replaces the constructor of the reference's KenLM adapter,
 * KenLM::KenLM(path, usrTknDict) (flashlight/lib/text/decoder/lm/KenLM.cpp:32-50):
 * load the model, then map every entry of the user dictionary to an LM word
 * id with vocab->Index(token), unknown strings going to <unk> (KenLM.cpp:44-49).
 * KenLM itself (third-party, not vendored by the reference and absent from this
 * image) also reads its own binary formats; only ARPA text is supported here.
 *
 * Scoring semantics are those of an ARPA back-off model evaluated in float
 * (see ngScore in fltx_kernels.h); the decode-time lookups run on the device.
 */
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <unordered_map>
#include <vector>

#include "fltx.h"

extern "C" int fltx_set_error_(int code, const char* msg);

namespace {

/* split on blanks / tabs */
void splitFields(const std::string& line, std::vector<std::string>& out) {
  out.clear();
  size_t p = 0;
  const size_t n = line.size();
  while (p < n) {
    while (p < n && (line[p] == ' ' || line[p] == '\t' || line[p] == '\r')) {
      ++p;
    }
    size_t q = p;
    while (q < n && line[q] != ' ' && line[q] != '\t' && line[q] != '\r') {
      ++q;
    }
    if (q > p) {
      out.emplace_back(line, p, q - p);
    }
    p = q;
  }
}

} // namespace

extern "C" int fltx_lm_arpa_load(const char* path, const char* usrWords, fltx_lm** out) {
  if (!path || !out) {
    return fltx_set_error_(FLTX_ERR_INVALID, "fltx_lm_arpa_load: null argument");
  }
  std::ifstream in(path);
  if (!in) {
    return fltx_set_error_(FLTX_ERR_INVALID, "[ngram LM] LM loading failed: cannot open file"); /* KenLM.cpp:35-37 */
  }
  std::string line;
  std::vector<long long> counts;
  bool inData = false;
  while (std::getline(in, line)) {
    if (line.compare(0, 6, "\\data\\") == 0) {
      inData = true;
    } else if (inData && line.compare(0, 6, "ngram ") == 0) {
      const size_t eq = line.find('=');
      if (eq == std::string::npos) {
        return fltx_set_error_(FLTX_ERR_INVALID, "[ngram LM] malformed ARPA header");
      }
      counts.push_back(std::atoll(line.c_str() + eq + 1));
    } else if (line.compare(0, 9, "\\1-grams:") == 0) {
      break;
    }
  }
  const int order = (int)counts.size();
  if (order < 1 || order > 6) { /* FL_TEXT_KENLM_MAX_ORDER = 6 (lm/CMakeLists.txt:3) */
    return fltx_set_error_(FLTX_ERR_UNSUPPORTED, "[ngram LM] ARPA order must be 1..6");
  }
  std::unordered_map<std::string, int32_t> vocab;
  vocab.emplace("<unk>", 0); /* KenLM keeps <unk> at index 0 */
  int32_t nWords = 1;
  std::vector<int32_t> ngOrder, ngWords;
  std::vector<float> prob, backoff;
  long long total = 0;
  for (long long c : counts) {
    total += c;
  }
  ngOrder.reserve((size_t)total);
  ngWords.reserve((size_t)total * order);
  prob.reserve((size_t)total);
  backoff.reserve((size_t)total);
  int cur = 1;
  std::vector<std::string> f;
  while (std::getline(in, line)) {
    if (line.empty() || line[0] == '\r') {
      continue;
    }
    if (line[0] == '\\') {
      if (line.compare(0, 5, "\\end\\") == 0) {
        break;
      }
      cur = std::atoi(line.c_str() + 1);
      if (cur < 1 || cur > order) {
        return fltx_set_error_(FLTX_ERR_INVALID, "[ngram LM] unexpected ARPA section");
      }
      continue;
    }
    splitFields(line, f);
    if ((int)f.size() < cur + 1) {
      return fltx_set_error_(FLTX_ERR_INVALID, "[ngram LM] malformed n-gram line");
    }
    ngOrder.push_back(cur);
    prob.push_back(std::strtof(f[0].c_str(), nullptr));
    backoff.push_back((int)f.size() > cur + 1 ? std::strtof(f[cur + 1].c_str(), nullptr) : 0.0f);
    for (int i = 0; i < order; ++i) {
      if (i >= cur) {
        ngWords.push_back(-1);
        continue;
      }
      const std::string& w = f[1 + i];
      auto it = vocab.find(w);
      if (it == vocab.end()) {
        if (cur == 1) {
          it = vocab.emplace(w, nWords++).first;
        } else {
          it = vocab.find("<unk>"); /* a word unseen as unigram */
        }
      }
      ngWords.push_back(it->second);
    }
  }
  if (ngOrder.empty()) {
    return fltx_set_error_(FLTX_ERR_INVALID, "[ngram LM] LM loading failed: no n-grams");
  }
  auto idx = [&](const std::string& w) {
    auto it = vocab.find(w);
    return it == vocab.end() ? 0 : it->second;
  };
  std::vector<int32_t> usrToLm;
  if (usrWords) {
    const char* p = usrWords;
    while (*p) {
      const char* q = std::strchr(p, '\n');
      const size_t len = q ? (size_t)(q - p) : std::strlen(p);
      usrToLm.push_back(idx(std::string(p, len)));
      if (!q) {
        break;
      }
      p = q + 1;
    }
  }
  return fltx_lm_ngram_create(nullptr, order, (int64_t)ngOrder.size(), ngOrder.data(), ngWords.data(),
                              prob.data(), backoff.data(), usrToLm.data(), (int32_t)usrToLm.size(),
                              idx("<s>"), idx("</s>"), 0, out);
}
