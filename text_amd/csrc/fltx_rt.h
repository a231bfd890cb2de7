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
FLTX_DEV uint32_t atomMin32(uint32_t* p, uint32_t val) { return atomicMin(p, val); }
FLTX_DEV unsigned long long atomMax64(unsigned long long* p, unsigned long long v) {
  return atomicMax(p, v);
}
FLTX_DEV unsigned long long atomMin64(unsigned long long* p, unsigned long long v) {
  return atomicMin(p, v);
}
FLTX_DEV unsigned long long atomOr64(unsigned long long* p, unsigned long long v) { return atomicOr(p, v); }
FLTX_DEV void atomAddF64(double* p, double v) { (void)atomicAdd(p, v); } /* (LDS: ds_add_f64) */
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
FLTX_DEV uint32_t loadCoherent32(const uint32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
FLTX_DEV void storeCoherent32(uint32_t* p, uint32_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
FLTX_DEV uint32_t ldsLoad32(const uint32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
/* LDS instructions of one wave execute in program order; this only stops the
 * compiler from reordering the record write past the publishing atomic. */
FLTX_DEV void compilerFence() { __asm__ volatile("" ::: "memory"); }
/* Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains
 * vmcnt, i.e. waits for every global load/store in flight (~2k clocks for the
 * emission prefetch, the back-pointer stores ...); the frame step exchanges
 * data between threads through LDS only, so those stay in flight. */
FLTX_DEV void ldsBarrier() { __asm__ volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

/* ---- wave-level primitives (wave64) --------------------------------------- */
/* LDS writes of this wave issued before the call are visible to its reads after
 * it: the LDS queue of a wave is in order, this only stops the compiler from
 * moving the accesses across */
FLTX_DEV void waveSync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
/* One dword per active lane from HBM straight into LDS (global_load_lds_dword: no register in
 * between, so nothing in the instruction stream waits for it): lane l's word lands at ldsRow[l].
 * ldsRowWait() before the row is read. */
FLTX_DEV void ldsRowLoad(float* ldsRow, const float* src, bool active) {
  if (active) {
    __builtin_amdgcn_global_load_lds(src, ldsRow, 4, 0, 0);
  }
}
FLTX_DEV void ldsRowWait() { __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
FLTX_DEV unsigned long long waveBallot(bool p) { return __ballot(p); }
FLTX_DEV int popc64(unsigned long long m) { return __popcll(m); }
/* src must be wave-uniform: v_readlane_b32 (VALU) instead of ds_bpermute (LDS crossbar) */
FLTX_DEV uint32_t waveShfl32(uint32_t v, int src) {
  return (uint32_t)__builtin_amdgcn_readlane((int)v, __builtin_amdgcn_readfirstlane(src));
}
/* broadcast lane `src` (must be wave-uniform) -- v_readlane_b32 */
FLTX_DEV uint32_t waveReadLane32(uint32_t v, int src) {
  return (uint32_t)__builtin_amdgcn_readlane((int)v, src);
}
FLTX_DEV int waveShflUpI(int v, int d) { return __shfl_up(v, d, 64); }
/* per-lane source (ds_bpermute) */
FLTX_DEV uint32_t waveGather32(uint32_t v, int src) { return (uint32_t)__shfl((int)v, src, 64); }
FLTX_DEV unsigned long long waveShfl64(unsigned long long v, int src) {
  const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)v, src, 64);
  const uint32_t hi = (uint32_t)__shfl((int)(uint32_t)(v >> 32), src, 64);
  return ((unsigned long long)hi << 32) | lo;
}
FLTX_DEV unsigned long long waveShflXor64(unsigned long long v, int m) {
  uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
  lo = (uint32_t)__shfl_xor((int)lo, m, 64);
  hi = (uint32_t)__shfl_xor((int)hi, m, 64);
  return ((unsigned long long)hi << 32) | lo;
}
FLTX_DEV int waveFirstLaneI(int v) { return __builtin_amdgcn_readfirstlane(v); }
/* a value the caller knows to be the same in every lane, moved to a scalar register */
FLTX_DEV int waveUniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
/* number of set bits of `m` below this lane (v_mbcnt_lo/hi) */
FLTX_DEV int wavePrefixCount(unsigned long long m) {
  return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
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

/* order-preserving map float -> u32 */
FLTX_DEV uint32_t f32Key(float x) {
  const uint32_t b = __float_as_uint(x);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
FLTX_DEV float f32FromKey(uint32_t k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

#ifndef FLTX_EMU
/* Wave64 scans/reductions on the DPP data path (VALU speed) instead of
 * ds_bpermute shuffles (LDS crossbar, ~100 clocks per step on a latency-bound
 * wave).  gfx9-family pattern: Hillis-Steele inside each row of 16 lanes with
 * row_shr:1,2,4,8, then row_bcast:15 into rows 1 and 3 and row_bcast:31 into
 * rows 2 and 3.  Lanes without a source read the identity (old = 0). */
template <int CTRL, int ROWMASK>
FLTX_DEV uint32_t dppTake(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWMASK, 0xf, true);
}
template <int CTRL, int ROWMASK>
FLTX_DEV unsigned long long dppTake64(unsigned long long v) {
  const uint32_t lo = dppTake<CTRL, ROWMASK>((uint32_t)v);
  const uint32_t hi = dppTake<CTRL, ROWMASK>((uint32_t)(v >> 32));
  return ((unsigned long long)hi << 32) | lo;
}
#define FLTX_DPP_SCAN(v, OP, TAKE)                 \
  v = OP(v, TAKE<0x111, 0xf>(v)); /* row_shr:1 */  \
  v = OP(v, TAKE<0x112, 0xf>(v)); /* row_shr:2 */  \
  v = OP(v, TAKE<0x114, 0xf>(v)); /* row_shr:4 */  \
  v = OP(v, TAKE<0x118, 0xf>(v)); /* row_shr:8 */  \
  v = OP(v, TAKE<0x142, 0xa>(v)); /* row_bcast:15 -> rows 1,3 */ \
  v = OP(v, TAKE<0x143, 0xc>(v)); /* row_bcast:31 -> rows 2,3 */
/* lane i of a row of 16 takes the value of lane (i - R) mod 16 of its row */
template <int R>
FLTX_DEV unsigned long long waveRowRor64(unsigned long long v) {
  if constexpr (R == 0) {
    return v;
  } else {
    return dppTake64<0x120 + R, 0xf>(v);
  }
}
FLTX_DEV unsigned long long umax64(unsigned long long a, unsigned long long b) { return a > b ? a : b; }
FLTX_DEV uint32_t uadd32(uint32_t a, uint32_t b) { return a + b; }
FLTX_DEV unsigned long long waveBcastLast64(unsigned long long v) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, 63);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), 63);
  return ((unsigned long long)hi << 32) | lo;
}
/* max over the wave, result in every lane (0 is the identity: keys are unsigned) */
FLTX_DEV unsigned long long waveMax64(unsigned long long v) {
  FLTX_DPP_SCAN(v, umax64, dppTake64)
  return waveBcastLast64(v);
}
FLTX_DEV uint32_t umax32(uint32_t a, uint32_t b) { return a > b ? a : b; }
FLTX_DEV uint32_t waveMax32(uint32_t v) {
  FLTX_DPP_SCAN(v, umax32, dppTake)
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
/* min over the wave: max of the complement */
FLTX_DEV unsigned long long waveMin64(unsigned long long v) { return ~waveMax64(~v); }
/* inclusive prefix sum of a non-negative int across the wave */
FLTX_DEV int waveInclusiveScan(int v) {
  uint32_t x = (uint32_t)v;
  FLTX_DPP_SCAN(x, uadd32, dppTake)
  return (int)x;
}
#endif /* !FLTX_EMU */

} // namespace fltx
