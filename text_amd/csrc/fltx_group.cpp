/*
 * fltx_group.cpp -- one batch over several devices.
 *
 * Utterances are independent, so a batch of B shards over D devices with no
 * exchange at all: one context + one decoder per device, one host thread per
 * device for the life of the group, trie and LM tables replicated (tens of MB),
 * results read back in the caller's utterance order.  Written against the public
 * C ABI only (include/fltx.h); wraps the per-utterance loop a multi-GPU user of
 * Decoder::decode (decoder/Decoder.h:51-57) would write by hand.
 */
#include <algorithm>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "fltx.h"

extern "C" int fltx_set_error_(int code, const char* msg);
extern "C" int fltx_lm_kind_(const fltx_lm* lm);

struct fltx_group {
  /* one host thread per device, alive for as long as the group (a thread per call would cost
   * tens of microseconds of create + join against a 2.7 ms batch) */
  struct Worker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<void()> job;
    bool busy = false, quit = false;
    Worker() {
      th = std::thread([this]() {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
          cv.wait(lk, [this]() { return busy || quit; });
          if (quit) {
            return;
          }
          lk.unlock();
          job();
          lk.lock();
          busy = false;
          cv.notify_all();
        }
      });
    }
    void post(std::function<void()> f) {
      std::lock_guard<std::mutex> lk(mu);
      job = std::move(f);
      busy = true;
      cv.notify_all();
    }
    void wait() {
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [this]() { return !busy; });
    }
    ~Worker() {
      {
        std::lock_guard<std::mutex> lk(mu);
        quit = true;
        cv.notify_all();
      }
      th.join();
    }
  };
  struct Part {
    fltx_ctx* ctx = nullptr;
    fltx_trie* trie = nullptr;
    fltx_decoder* dec = nullptr;
    int first = 0, count = 0; /* utterances [first, first + count) of the last batch */
    int rc = 0;
    std::string err;
    std::unique_ptr<Worker> worker; /* started by the first batch that gives this part work */
  };
  std::vector<Part> parts;
  int B = 0;
};

namespace {
int gfail(int code, const std::string& msg) { return fltx_set_error_(code, msg.c_str()); }

/* run f(part) on every part that has work, each on its part's host thread */
template <class F>
int forEachPart(fltx_group* g, F&& f) {
  for (auto& p : g->parts) {
    p.rc = 0;
    if (p.count == 0) {
      continue;
    }
    if (!p.worker) {
      p.worker.reset(new fltx_group::Worker());
    }
    fltx_group::Part* pp = &p;
    p.worker->post([pp, &f]() {
      pp->rc = f(*pp);
      if (pp->rc) {
        pp->err = fltx_last_error(); /* (the message is thread-local) */
      }
    });
  }
  for (auto& p : g->parts) {
    if (p.count != 0 && p.worker) {
      p.worker->wait();
    }
  }
  for (auto& p : g->parts) {
    if (p.rc) {
      return gfail(p.rc, p.err);
    }
  }
  return FLTX_OK;
}

const fltx_group::Part* partOf(const fltx_group* g, int b) {
  for (const auto& p : g->parts) {
    if (b >= p.first && b < p.first + p.count) {
      return &p;
    }
  }
  return nullptr;
}
} // namespace

extern "C" {

int fltx_group_create(const int32_t* devices, int32_t nDevices, int32_t kind, const fltx_options* opt,
                      fltx_htrie* htrie, const fltx_lm* lm, int32_t sil, int32_t blank, int32_t unk,
                      const float* transitions, int32_t nTransitions, int32_t isLmToken, fltx_group** out) {
  if (!devices || nDevices <= 0 || !opt || !lm || !out) {
    return gfail(FLTX_ERR_INVALID, "fltx_group_create: null or empty argument");
  }
  if (kind == FLTX_DECODER_LEXICON && !htrie) {
    return gfail(FLTX_ERR_INVALID, "fltx_group_create: the lexicon decoder needs a host trie to replicate");
  }
  if (nDevices > 1 && fltx_lm_kind_(lm) == 2) {
    /* its callbacks are made on the caller's thread and never concurrently (decoder/Utils.h:60-62); a group drives
     * its devices from one host thread each */
    return gfail(FLTX_ERR_UNSUPPORTED, "fltx_group_create: a host LM (fltx_lm_host_create) serves one device");
  }
  auto* g = new fltx_group();
  g->parts.resize((size_t)nDevices);
  for (int i = 0; i < nDevices; ++i) {
    auto& p = g->parts[(size_t)i];
    int rc = fltx_ctx_create(devices[i], nullptr, &p.ctx);
    if (!rc && kind == FLTX_DECODER_LEXICON) {
      rc = fltx_htrie_upload(htrie, p.ctx, &p.trie);
    }
    if (!rc) {
      rc = fltx_decoder_create(p.ctx, kind, opt, p.trie, lm, sil, blank, unk, transitions, nTransitions, isLmToken,
                               &p.dec);
    }
    if (rc) {
      const std::string msg = fltx_last_error();
      fltx_group_destroy(g);
      return gfail(rc, msg);
    }
  }
  *out = g;
  return FLTX_OK;
}

int fltx_group_destroy(fltx_group* g) {
  if (!g) {
    return FLTX_OK;
  }
  for (auto& p : g->parts) {
    p.worker.reset();
    if (p.dec) {
      fltx_decoder_destroy(p.dec);
    }
    if (p.trie) {
      fltx_trie_destroy(p.trie);
    }
    if (p.ctx) {
      fltx_ctx_destroy(p.ctx);
    }
  }
  delete g;
  return FLTX_OK;
}

int fltx_group_size(fltx_group* g, int32_t* nDevices) {
  if (!g || !nDevices) {
    return gfail(FLTX_ERR_INVALID, "fltx_group_size: null argument");
  }
  *nDevices = (int32_t)g->parts.size();
  return FLTX_OK;
}

int fltx_group_decoder(fltx_group* g, int32_t i, fltx_decoder** dec, int32_t* first, int32_t* count) {
  if (!g || i < 0 || i >= (int32_t)g->parts.size()) {
    return gfail(FLTX_ERR_RANGE, "fltx_group_decoder: part out of range");
  }
  const auto& p = g->parts[(size_t)i];
  if (dec) {
    *dec = p.dec;
  }
  if (first) {
    *first = p.first;
  }
  if (count) {
    *count = p.count;
  }
  return FLTX_OK;
}

int fltx_group_decode_batch(fltx_group* g, const float* const* emissions, const int32_t* onDevice,
                            const int64_t* offsets, const int32_t* T, int32_t B, int32_t N) {
  if (!g || !emissions || !T || B <= 0 || N <= 0) {
    return gfail(FLTX_ERR_INVALID, "fltx_group_decode_batch: bad argument");
  }
  /* contiguous shards balanced by frames (the frame step is serial in T: a shard takes as long
   * as the sum over its launch rounds of the longest utterance, so equal frame counts are what
   * keeps ragged batches level) */
  const int D = (int)g->parts.size();
  long long total = 0;
  for (int b = 0; b < B; ++b) {
    if (T[b] < 0) {
      return gfail(FLTX_ERR_INVALID, "fltx_group_decode_batch: negative T");
    }
    total += T[b] + 1;
  }
  int b0 = 0;
  long long done = 0;
  for (int i = 0; i < D; ++i) {
    auto& p = g->parts[(size_t)i];
    p.first = b0;
    const long long want = total * (i + 1) / D;
    while (b0 < B && (i == D - 1 || done < want)) {
      done += T[b0] + 1;
      ++b0;
    }
    p.count = b0 - p.first;
  }
  g->B = B;
  return forEachPart(g, [&](fltx_group::Part& p) {
    const int i = (int)(&p - g->parts.data());
    /* emissions[i] is device i's pointer (host memory, or that device's HBM when on_device[i]);
     * offsets are relative to it, utterance numbering is the caller's */
    std::vector<int64_t> offs((size_t)p.count, 0);
    if (offsets) {
      for (int k = 0; k < p.count; ++k) {
        offs[(size_t)k] = offsets[p.first + k];
      }
    } else {
      int64_t o = 0;
      for (int b = 0; b < p.first; ++b) {
        o += (int64_t)T[b] * N;
      }
      for (int k = 0; k < p.count; ++k) {
        offs[(size_t)k] = o;
        o += (int64_t)T[p.first + k] * N;
      }
    }
    const int od = onDevice ? onDevice[i] : 0;
    return fltx_decode_batch(p.dec, emissions[i], od, offs.data(), T + p.first, p.count, N);
  });
}

int fltx_group_result_count(fltx_group* g, int32_t b, int32_t* nHyp, int32_t* length) {
  const auto* p = g ? partOf(g, b) : nullptr;
  if (!p) {
    return gfail(FLTX_ERR_RANGE, "fltx_group_result_count: no such utterance in the last batch");
  }
  return fltx_result_count(p->dec, b - p->first, nHyp, length);
}

int fltx_group_result_fetch(fltx_group* g, int32_t b, int32_t maxHyp, double* scores, int32_t* tokens,
                            int32_t* words, int32_t* nCopied) {
  const auto* p = g ? partOf(g, b) : nullptr;
  if (!p) {
    return gfail(FLTX_ERR_RANGE, "fltx_group_result_fetch: no such utterance in the last batch");
  }
  return fltx_result_fetch(p->dec, b - p->first, maxHyp, scores, tokens, words, nCopied);
}

int fltx_group_synchronize(fltx_group* g) {
  if (!g) {
    return gfail(FLTX_ERR_INVALID, "null group");
  }
  for (auto& p : g->parts) {
    int rc = fltx_ctx_synchronize(p.ctx);
    if (rc) {
      return rc;
    }
  }
  return FLTX_OK;
}

} /* extern "C" */
