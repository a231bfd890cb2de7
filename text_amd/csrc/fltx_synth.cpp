/*
 * fltx_synth.cpp -- bit-reproducible synthetic measurement inputs
 * (SURVEY.md Appendix A): emissions in three distributions and the 90k-word
 * sil-terminated lexicon.  Integer PRNG (splitmix64) -> 24-bit integers ->
 * exact float32 ops, no libm, so every IEEE-754 host produces identical bytes.
 *
 * Host-only helper (g++), shared by tests/, bench.py and
 * tests/golden/make_golden.py.  Not part of the decode path.
 */
#include <cstdint>
#include <cstring>
#include <string>
#include <unordered_set>
#include <vector>

namespace {

struct SplitMix64 {
  uint64_t s;
  explicit SplitMix64(uint64_t seed) : s(seed) {}
  uint64_t next() {
    s += 0x9E3779B97F4A7C15ull;
    uint64_t z = s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  uint32_t r24() { return (uint32_t)(next() >> 40); }
};

/* one frame of N distinct 24-bit draws -> x_n = r_n * 2^-24 (exact) */
void frameDraw(SplitMix64& g, int N, std::vector<uint32_t>& r, float* x) {
  r.resize(N);
  for (int n = 0; n < N; ++n) {
    uint32_t v;
    bool dup;
    do {
      v = g.r24();
      dup = false;
      for (int m = 0; m < n; ++m) {
        if (r[m] == v) {
          dup = true;
          break;
        }
      }
    } while (dup);
    r[n] = v;
    x[n] = (float)v * 0x1p-24f;
  }
}

} // namespace

extern "C" {

/* dist: 0 = uniform, 1 = ctc, 2 = lexspell (needs spell_flat/spell_off/W).
 * out: [T*N] frame-major f32.  Returns 0, or -1 on bad arguments. */
int fltx_synth_emissions(int dist, uint64_t S, uint64_t u, int T, int N,
                         const int32_t* spell_flat, const int64_t* spell_off,
                         int64_t W, float* out) {
  if (N < 2 || T < 0 || !out) {
    return -1;
  }
  const int blank = N - 1;
  std::vector<uint32_t> r;
  std::vector<float> x(N);
  if (dist == 0) {
    SplitMix64 g(S * 1000003ull + u);
    for (int t = 0; t < T; ++t) {
      frameDraw(g, N, r, x.data());
      for (int n = 0; n < N; ++n) {
        out[(size_t)t * N + n] = -(x[n]) * 10.0f;
      }
    }
    return 0;
  }
  if (dist == 1) {
    SplitMix64 g(S * 1000003ull + u + 0x5EEDull);
    int prevTok = -1;
    for (int t = 0; t < T; ++t) {
      int w;
      uint32_t a = g.r24();
      if (a < 10066330u) {
        w = blank;
      } else {
        uint32_t b = g.r24();
        if (prevTok >= 0 && b < 6710886u) {
          w = prevTok;
        } else {
          w = (int)(g.next() % (uint64_t)(N - 1));
        }
        prevTok = w;
      }
      frameDraw(g, N, r, x.data());
      for (int n = 0; n < N; ++n) {
        out[(size_t)t * N + n] = (n == w) ? -(x[n] * 0.5f) : (-3.0f - x[n] * 8.0f);
      }
    }
    return 0;
  }
  if (dist == 2) {
    if (!spell_flat || !spell_off || W <= 0) {
      return -1;
    }
    SplitMix64 g(S * 1000003ull + u + 0xA11CEull);
    std::vector<int> win;
    int lastTok = -1;
    bool lastWasBlank = true;
    while ((int)win.size() < T) {
      int64_t wi = (int64_t)(g.next() % (uint64_t)W);
      for (int64_t p = spell_off[wi]; p < spell_off[wi + 1]; ++p) {
        int tok = spell_flat[p];
        if (tok == lastTok && !lastWasBlank) {
          win.push_back(blank);
          lastWasBlank = true;
        }
        int hold = 1 + (int)(g.next() % 3);
        for (int h = 0; h < hold; ++h) {
          win.push_back(tok);
        }
        lastTok = tok;
        lastWasBlank = false;
        if ((g.next() >> 63) == 1) {
          int nb = 1 + (int)(g.next() % 2);
          for (int h = 0; h < nb; ++h) {
            win.push_back(blank);
          }
          lastWasBlank = true;
        }
      }
    }
    for (int t = 0; t < T; ++t) {
      int w = win[t];
      frameDraw(g, N, r, x.data());
      for (int n = 0; n < N; ++n) {
        out[(size_t)t * N + n] = (n == w) ? -(x[n] * 0.5f) : (-3.0f - x[n] * 8.0f);
      }
    }
    return 0;
  }
  return -1;
}

/* n floats lo + (hi-lo) * r24 * 2^-24 (float ops), e.g. ASG transitions */
void fltx_synth_floats(uint64_t seed, int64_t n, float lo, float hi, float* out) {
  SplitMix64 g(seed);
  for (int64_t i = 0; i < n; ++i) {
    out[i] = lo + (hi - lo) * ((float)g.r24() * 0x1p-24f);
  }
}

/* Synthetic lexicon: W unique spellings of 2..12 letters (tokens 1..27)
 * followed by token 0 (sil).  spell_off must hold W+1 entries; spell_flat
 * capacity `cap` tokens.  Returns the total number of tokens written, or -1
 * if cap is too small. */
int64_t fltx_synth_lexicon(uint64_t seed, int64_t W, int32_t* spell_flat,
                           int64_t cap, int64_t* spell_off) {
  SplitMix64 g(seed);
  std::unordered_set<std::string> seen;
  seen.reserve((size_t)W * 2);
  int64_t w = 0, pos = 0;
  spell_off[0] = 0;
  std::string key;
  while (w < W) {
    int len = 2 + (int)(g.next() % 11);
    key.clear();
    for (int i = 0; i < len; ++i) {
      key.push_back((char)(1 + (int)(g.next() % 27)));
    }
    key.push_back((char)0);
    if (!seen.insert(key).second) {
      continue;
    }
    if (pos + (int64_t)key.size() > cap) {
      return -1;
    }
    for (char c : key) {
      spell_flat[pos++] = (int32_t)c;
    }
    ++w;
    spell_off[w] = pos;
  }
  return pos;
}

} // extern "C"
