/*
 * tests/emu/hip_emu.h -- minimal host emulation of the HIP constructs used by
 * text_amd/csrc/fltx_kernels.h, for debugging kernel LOGIC without a GPU.
 *
 * TEST INFRASTRUCTURE ONLY.  A workgroup runs as one host thread per WAVE; the 64 lanes of a wave are
 * cooperative fibers on that thread (their own stacks, a register-only switch): a wave collective is a
 * round of switches, __syncthreads a barrier between the wave threads.  (The first version ran every
 * lane as a host thread with pthread barriers: 576 .. 1 024 threads per workgroup, most of the time in
 * futex wake-ups.)  Wave collectives exchange through a per-wave scratch.  It is slow (a fraction of a
 * millisecond per frame) and is only ever built into tests/emu/libfltx_emu.so by tests/emu/build.sh;
 * the product library (text_amd/lib/libfltx.so) has no CPU path and never loads this.
 */
#pragma once
#include <pthread.h>
#include <stdint.h>

#include <cmath>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

#define FLTX_DEV static inline
#define __device__
#define __global__
#define __host__
#define __forceinline__ inline

struct uint4 {
  uint32_t x, y, z, w;
};
struct int2 {
  int x, y;
};
static inline uint4 make_uint4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) { return uint4{a, b, c, d}; }
static inline int2 make_int2(int a, int b) { return int2{a, b}; }
struct uint2 {
  uint32_t x, y;
};
static inline uint2 make_uint2(uint32_t a, uint32_t b) { return uint2{a, b}; }

struct EmuDim {
  unsigned x;
};

struct EmuWave {
  unsigned long long slot[64];
  void* sp[64];      /* saved stack pointers of the lanes' fibers */
  void* schedSp;     /* ... of the wave thread itself */
  bool done[64];
  int cur;           /* lane that runs */
  int nDone;
  unsigned waveGen, blockGen; /* generations of the wave / workgroup barrier this wave has passed */
  int waveArrive, blockArrive;
  unsigned base;     /* threadIdx.x of lane 0 */
};
struct EmuBlock {
  pthread_barrier_t bar; /* one participant per wave */
  std::vector<EmuWave> waves;
  const std::function<void(char*)>* fn;
  char* lds;
};

extern thread_local EmuDim threadIdx, blockIdx, blockDim;
extern thread_local EmuBlock* emuBlock;

/* let the next lane of this wave that has not finished run (returns when this lane's turn comes again) */
void emuYield();

static inline void __syncthreads() {
  EmuWave& w = emuBlock->waves[threadIdx.x >> 6];
  const unsigned g = w.blockGen;
  if (++w.blockArrive == 64) { /* the wave's last lane: the wave thread meets the other waves */
    pthread_barrier_wait(&emuBlock->bar);
    w.blockArrive = 0;
    w.blockGen = g + 1;
  } else {
    while (w.blockGen == g) {
      emuYield();
    }
  }
}
static inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
static inline float __uint_as_float(uint32_t u) {
  float f;
  memcpy(&f, &u, 4);
  return f;
}
static inline uint32_t __float_as_uint(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  return u;
}
static inline long long __double_as_longlong(double d) {
  long long r;
  memcpy(&r, &d, 8);
  return r;
}
static inline double __longlong_as_double(long long v) {
  double r;
  memcpy(&r, &v, 8);
  return r;
}

namespace fltx {
FLTX_DEV int laneId() { return (int)(threadIdx.x & 63); }
FLTX_DEV int waveId() { return (int)(threadIdx.x >> 6); }

FLTX_DEV uint32_t atomCas32(uint32_t* p, uint32_t cmp, uint32_t val) {
  __atomic_compare_exchange_n(p, &cmp, val, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST);
  return cmp;
}
FLTX_DEV uint32_t atomExch32(uint32_t* p, uint32_t v) { return __atomic_exchange_n(p, v, __ATOMIC_SEQ_CST); }
FLTX_DEV uint32_t atomAdd32(uint32_t* p, uint32_t v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
FLTX_DEV uint32_t atomOr32(uint32_t* p, uint32_t v) { return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST); }
FLTX_DEV uint32_t atomMin32(uint32_t* p, uint32_t v) {
  uint32_t cur = __atomic_load_n(p, __ATOMIC_SEQ_CST);
  while (cur > v && !__atomic_compare_exchange_n(p, &cur, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {
  }
  return cur;
}
FLTX_DEV unsigned long long atomMax64(unsigned long long* p, unsigned long long v) {
  unsigned long long cur = __atomic_load_n(p, __ATOMIC_SEQ_CST);
  while (cur < v && !__atomic_compare_exchange_n(p, &cur, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {
  }
  return cur;
}
FLTX_DEV unsigned long long atomMin64(unsigned long long* p, unsigned long long v) {
  unsigned long long cur = __atomic_load_n(p, __ATOMIC_SEQ_CST);
  while (cur > v && !__atomic_compare_exchange_n(p, &cur, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {
  }
  return cur;
}
FLTX_DEV unsigned long long atomOr64(unsigned long long* p, unsigned long long v) {
  return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST);
}
FLTX_DEV void atomAddF64(double* p, double v) {
  unsigned long long cur = __atomic_load_n((unsigned long long*)p, __ATOMIC_SEQ_CST);
  for (;;) {
    double d;
    memcpy(&d, &cur, 8);
    d += v;
    unsigned long long nxt;
    memcpy(&nxt, &d, 8);
    if (__atomic_compare_exchange_n((unsigned long long*)p, &cur, nxt, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {
      return;
    }
  }
}
FLTX_DEV uint32_t loadCoherent32(const uint32_t* p) { return __atomic_load_n(p, __ATOMIC_SEQ_CST); }
FLTX_DEV void storeCoherent32(uint32_t* p, uint32_t v) { __atomic_store_n(p, v, __ATOMIC_SEQ_CST); }
FLTX_DEV unsigned long long atomCas64(unsigned long long* p, unsigned long long cmp, unsigned long long val) {
  __atomic_compare_exchange_n(p, &cmp, val, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST);
  return cmp;
}
FLTX_DEV unsigned long long loadCoherent64(const unsigned long long* p) {
  return __atomic_load_n(p, __ATOMIC_SEQ_CST);
}
FLTX_DEV uint32_t ldsLoad32(const uint32_t* p) { return __atomic_load_n(p, __ATOMIC_SEQ_CST); }
FLTX_DEV void compilerFence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
FLTX_DEV void ldsBarrier() { __syncthreads(); }

/* wave collectives: publish, barrier, read, barrier */
FLTX_DEV EmuWave& emuWave() { return emuBlock->waves[threadIdx.x >> 6]; }
FLTX_DEV void waveSync() {
  EmuWave& w = emuWave();
  const unsigned g = w.waveGen;
  if (++w.waveArrive == 64) {
    w.waveArrive = 0;
    w.waveGen = g + 1;
  } else {
    while (w.waveGen == g) {
      emuYield();
    }
  }
}
template <class F>
FLTX_DEV auto emuExchange(unsigned long long mine, F&& f) {
  EmuWave& w = emuWave();
  w.slot[threadIdx.x & 63] = mine;
  waveSync();
  auto r = f(w.slot);
  waveSync();
  return r;
}
FLTX_DEV unsigned long long waveBallot(bool p) {
  return emuExchange(p ? 1ull : 0ull, [](const unsigned long long* s) {
    unsigned long long m = 0;
    for (int i = 0; i < 64; ++i) {
      m |= (s[i] & 1ull) << i;
    }
    return m;
  });
}
FLTX_DEV int popc64(unsigned long long m) { return __builtin_popcountll(m); }
FLTX_DEV void ldsRowLoad(float* ldsRow, const float* src, bool active) {
  if (active) {
    ldsRow[threadIdx.x & 63] = *src;
  }
}
FLTX_DEV void ldsRowWait() {}
FLTX_DEV int waveUniform(int v) { return v; }
FLTX_DEV int wavePrefixCount(unsigned long long m) {
  const int lane = (int)(threadIdx.x & 63);
  return __builtin_popcountll(lane ? (m & (~0ull >> (64 - lane))) : 0ull);
}
FLTX_DEV uint32_t waveShfl32(uint32_t v, int src) {
  return emuExchange(v, [src](const unsigned long long* s) { return (uint32_t)s[src & 63]; });
}
FLTX_DEV uint32_t waveReadLane32(uint32_t v, int src) { return waveShfl32(v, src); }
FLTX_DEV uint32_t waveGather32(uint32_t v, int src) { return waveShfl32(v, src); }
FLTX_DEV unsigned long long waveShfl64(unsigned long long v, int src) {
  return emuExchange(v, [src](const unsigned long long* s) { return s[src & 63]; });
}
/* lane i of a row of 16 takes the value of lane (i - R) mod 16 of its row (DPP row_ror) */
template <int R>
FLTX_DEV unsigned long long waveRowRor64(unsigned long long v) {
  const int lane = (int)(threadIdx.x & 63);
  return waveShfl64(v, (lane & ~15) | ((lane - R) & 15));
}
FLTX_DEV unsigned long long waveMax64(unsigned long long v) {
  return emuExchange(v, [](const unsigned long long* s) {
    unsigned long long m = 0;
    for (int i = 0; i < 64; ++i) {
      m = s[i] > m ? s[i] : m;
    }
    return m;
  });
}
FLTX_DEV uint32_t waveMax32(uint32_t v) { return (uint32_t)waveMax64((unsigned long long)v); }
FLTX_DEV unsigned long long waveMin64(unsigned long long v) {
  return emuExchange(v, [](const unsigned long long* s) {
    unsigned long long m = ~0ull;
    for (int i = 0; i < 64; ++i) {
      m = s[i] < m ? s[i] : m;
    }
    return m;
  });
}
FLTX_DEV int waveInclusiveScan(int v) {
  const int lane = (int)(threadIdx.x & 63);
  return emuExchange((unsigned long long)(uint32_t)v, [lane](const unsigned long long* s) {
    int acc = 0;
    for (int i = 0; i <= lane; ++i) {
      acc += (int)(uint32_t)s[i];
    }
    return acc;
  });
}
} // namespace fltx

/* run fn() as a grid of nBlocks workgroups of W threads, with `ldsBytes` of
 * zero-initialised "LDS" per workgroup */
void emuLaunch(int nBlocks, int W, size_t ldsBytes, const std::function<void(char*)>& fn);
