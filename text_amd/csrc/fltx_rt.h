/*
 * fltx_rt.h -- the handful of device-runtime primitives the kernels use.
 *
 * Product build (hipcc, gfx950): thin inline wrappers over HIP / AMDGCN
 * builtins.  Wave width is 64 (hard-coded: gfx950 is wave64 only).
 *
 * FLTX_EMU build (g++, tests/emu only): the same names are provided by
 * tests/emu/hip_emu.h, which runs each workgroup as W host threads with real
 * barriers so the kernel LOGIC can be debugged without a GPU.  The emulator is
 * test infrastructure: nothing in text_amd/ loads it and the product library
 * has no CPU path.
 */
#pragma once
#include <stdint.h>

#ifdef FLTX_EMU
#include "hip_emu.h"
#else
#include <hip/hip_runtime.h>
#endif

namespace fltx {

constexpr int kWave = 64;

#ifndef FLTX_EMU
#define FLTX_DEV __device__ __forceinline__

FLTX_DEV int laneId() { return (int)(threadIdx.x & 63); }
FLTX_DEV int waveId() { return (int)(threadIdx.x >> 6); }

/* ---- LDS / global atomics (relaxed; ordering comes from barriers) -------- */
FLTX_DEV uint32_t atomCas32(uint32_t* p, uint32_t cmp, uint32_t val) {
  return atomicCAS(p, cmp, val);
}
FLTX_DEV uint32_t atomExch32(uint32_t* p, uint32_t val) { return atomicExch(p, val); }
FLTX_DEV uint32_t atomAdd32(uint32_t* p, uint32_t val) { return atomicAdd(p, val); }
FLTX_DEV uint32_t atomOr32(uint32_t* p, uint32_t val) { return atomicOr(p, val); }
FLTX_DEV unsigned long long atomMax64(unsigned long long* p, unsigned long long v) {
  return atomicMax(p, v);
}
FLTX_DEV unsigned long long atomMin64(unsigned long long* p, unsigned long long v) {
  return atomicMin(p, v);
}
FLTX_DEV unsigned long long atomCas64(unsigned long long* p, unsigned long long cmp,
                                      unsigned long long val) {
  return atomicCAS(p, cmp, val);
}
/* L1-bypassing load (global_load ... sc1): the LM-state table is written with
 * L2 atomics by this workgroup, so it must never be read through the
 * non-coherent vector L1 (MI355X_MICROARCH.md, inter-workgroup visibility). */
FLTX_DEV unsigned long long loadCoherent64(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
FLTX_DEV uint32_t ldsLoad32(const uint32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
/* LDS instructions of one wave execute in program order; this only stops the
 * compiler from reordering the record write past the publishing atomic. */
FLTX_DEV void compilerFence() { __asm__ volatile("" ::: "memory"); }

/* ---- wave-level primitives (wave64) --------------------------------------- */
FLTX_DEV unsigned long long waveBallot(bool p) { return __ballot(p); }
FLTX_DEV int popc64(unsigned long long m) { return __popcll(m); }
FLTX_DEV uint32_t waveShfl32(uint32_t v, int src) { return (uint32_t)__shfl((int)v, src, 64); }
FLTX_DEV int waveShflUpI(int v, int d) { return __shfl_up(v, d, 64); }
FLTX_DEV unsigned long long waveShflXor64(unsigned long long v, int m) {
  uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
  lo = (uint32_t)__shfl_xor((int)lo, m, 64);
  hi = (uint32_t)__shfl_xor((int)hi, m, 64);
  return ((unsigned long long)hi << 32) | lo;
}
FLTX_DEV int waveFirstLaneI(int v) { return __builtin_amdgcn_readfirstlane(v); }
#endif /* !FLTX_EMU */

/* ---- portable helpers ------------------------------------------------------ */
/* order-preserving map double -> u64 (a > b  <=>  key(a) > key(b)) */
FLTX_DEV unsigned long long f64Key(double d) {
  unsigned long long b = (unsigned long long)__double_as_longlong(d);
  return (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);
}
FLTX_DEV double f64FromKey(unsigned long long k) {
  unsigned long long b = (k & 0x8000000000000000ull) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
  return __longlong_as_double((long long)b);
}

#ifndef FLTX_EMU
FLTX_DEV unsigned long long waveMax64(unsigned long long v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    unsigned long long o = waveShflXor64(v, m);
    v = o > v ? o : v;
  }
  return v;
}
FLTX_DEV unsigned long long waveMin64(unsigned long long v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    unsigned long long o = waveShflXor64(v, m);
    v = o < v ? o : v;
  }
  return v;
}
/* inclusive scan of an int across the wave */
FLTX_DEV int waveInclusiveScan(int v) {
  int lane = laneId();
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    int o = waveShflUpI(v, d);
    if (lane >= d) {
      v += o;
    }
  }
  return v;
}
#endif /* !FLTX_EMU */

} // namespace fltx
