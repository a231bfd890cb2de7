/*
 * fltx_kernel_entry.h -- the __global__ entry points of the decode kernels
 * (templates over workgroup size and engine variant).  Every instantiation
 * the host launches is listed once in fltx_instances.h; fltx_api.cpp declares
 * them `extern template` and the instance translation units (fltx_kinst.cpp,
 * one object per group, compiled in parallel by __graft_entry__.build) define
 * them, so the 85 kernels do not have to be compiled by one hipcc process.
 */
#pragma once
#include "fltx_kernels.h"

#ifndef FLTX_EMU
using namespace fltx;
/* kernels: global scope, external linkage (the runtime resolves them by name) */
/* One instantiation per workgroup size so that __launch_bounds__ gives the
 * register allocator the real budget (256 threads = 1 wave per SIMD = up to
 * 512 VGPRs; the 1024-thread default would cap it at 128 and spill). */
template <int W, int GMAX>
__global__ void __launch_bounds__(W) fltx_decode_kernel_lds(DecodeParams P) {
  extern __shared__ __attribute__((aligned(16))) char fltx_smem[];
  decodeUtterance<GMAX>(P, fltx_smem);
}
/* generic engine specialised by assumption: LEX = lexicon decoder with a word LM
 * (no lexicon-free generation, no token-LM paths), ZLM = ZeroLM (no n-gram
 * scoring: a third of the kernel and a good part of its register pressure).
 * The facts are stated to the compiler, which then drops the dead paths. */
template <int W, bool LEX, bool ZLM>
__global__ void __launch_bounds__(W) fltx_decode_kernel_lds_spec(DecodeParams P) {
  extern __shared__ __attribute__((aligned(16))) char fltx_smem[];
  if (LEX) {
    __builtin_assume(P.kind == 1);
    __builtin_assume(P.isLmToken == 0);
    __builtin_assume(P.dense == 0);
  }
  if (ZLM) {
    __builtin_assume(P.lmKind == 0);
    __builtin_assume(P.isLmToken == 0);
  } else {
    __builtin_assume(P.lmKind == 1);
  }
  decodeUtterance<0>(P, fltx_smem);
}
template <int W, int GT, bool LOGADD, bool FULLTOK>
__global__ void __launch_bounds__(W) fltx_decode_kernel_lane(DecodeParams P) {
  extern __shared__ __attribute__((aligned(16))) char fltx_smem[];
  decodeUtterance<1, GT, LOGADD, FULLTOK>(P, fltx_smem);
}
/* lane = LM state decode of a whole utterance (fltx_slane.h): the headline configuration */
template <int W, int GT, bool LA, bool PROF>
__global__ void __launch_bounds__(W) fltx_decode_kernel_slane(DecodeParams P) {
  extern __shared__ __attribute__((aligned(16))) char fltx_smem[];
  slaneUtterance<GT, LA, false, PROF>(P, fltx_smem);
}
/* ... with a token-level n-gram LM flattened to a dense (context, token) table (TL) */
/* (512 threads: the geometry of which two workgroups share a CU -- 75 KB of LDS each with the small memos, and at most
 * 128 VGPRs: four waves per SIMD) */
template <int W, int GT, bool LA, bool PROF = false>
__global__ void __launch_bounds__(W, W == 512 && GT == 5 ? 4 : 1) fltx_decode_kernel_tlane(DecodeParams P) {
  extern __shared__ __attribute__((aligned(16))) char fltx_smem[];
  slaneUtterance<GT, LA, false, PROF, true>(P, fltx_smem);
}
/* ... with NG groups of 64 lanes: beams 65 .. 64 * NG (fltx_mlane.h) */
template <int W, int GT, int NG, int GPW, int SPW, bool LA>
__global__ void __launch_bounds__(W) fltx_decode_kernel_mlane(DecodeParams P) {
  extern __shared__ __attribute__((aligned(16))) char fltx_smem[];
  mlaneUtterance<GT, NG, GPW, SPW, LA>(P, fltx_smem);
}
/* ... and a token-level n-gram LM (dense table; state ids from a table in HBM) */
template <int W, int GT, int NG, int GPW, int SPW, bool LA>
__global__ void __launch_bounds__(W) fltx_decode_kernel_tmlane(DecodeParams P) {
  extern __shared__ __attribute__((aligned(16))) char fltx_smem[];
  mlaneUtterance<GT, NG, GPW, SPW, LA, true>(P, fltx_smem);
}
/* ... for token sets beyond 64 (word pieces) with a token beam of at most 64 (fltx_wlane.h) */
template <int W, int GT>
__global__ void __launch_bounds__(W) fltx_decode_kernel_wlane(DecodeParams P) {
  extern __shared__ __attribute__((aligned(16))) char fltx_smem[];
  wlaneUtterance<GT>(P, fltx_smem);
}
/* the frames of one decodeStep chunk of a stream on the same engine: beam in and out in the parked format of the
 * lane-per-slot step, whose kernels do decodeBegin / decodeEnd / prune / getBestHypothesis */
template <int W, int GT>
__global__ void __launch_bounds__(W) fltx_decode_kernel_slane_stream(DecodeParams P) {
  extern __shared__ __attribute__((aligned(16))) char fltx_smem[];
  slaneUtterance<GT, false, true, false>(P, fltx_smem);
}
/* ... and with a token-level n-gram LM (state ids from the generic engine's table, whose kernels do begin / end / prune) */
template <int W, int GT>
__global__ void __launch_bounds__(W) fltx_decode_kernel_tlane_stream(DecodeParams P) {
  extern __shared__ __attribute__((aligned(16))) char fltx_smem[];
  slaneUtterance<GT, false, true, false, true>(P, fltx_smem);
}
/* lane = (LM state, trie node) decode of a whole utterance (fltx_xlane.h): lexicon + ZeroLM */
/* HM = 1: the LM-state memo in HBM, 28 KB of LDS -- several workgroups share a CU (batches beyond the CUs) */
template <int W, int GT, int HM, bool PROF, bool LA = false> /* LA: logAdd merges */
__global__ void __launch_bounds__(W, HM ? 4 : 1) fltx_decode_kernel_xlane(DecodeParams P) {
  extern __shared__ __attribute__((aligned(16))) char fltx_smem[];
  xlaneUtterance<GT, HM, PROF, LA>(P, fltx_smem);
}
/* ... with a word LM and / or a smeared trie, beams up to 128 (fltx_ylane.h) */
/* HM = 1: the geometry that shares a CU (LM-state memo in HBM, 77 KB of LDS, at most 128 VGPRs: four
 * waves per SIMD = two workgroups of 512 threads).  LMK bit 2 (several words per spelling): the larger merge table leaves
 * room for one workgroup per CU, which may then take all the registers its waves can address */
template <int W, int NG, int R, int LMK, int HM, bool PROF>
__global__ void __launch_bounds__(W, (HM && !(LMK & 4)) ? 4 : 1) fltx_decode_kernel_ylane(DecodeParams P) {
  extern __shared__ __attribute__((aligned(16))) char fltx_smem[];
  ylaneUtterance<NG, R, LMK, HM, PROF>(P, fltx_smem);
}
template <int W>
__global__ void __launch_bounds__(W) fltx_decode_kernel_gwslean(DecodeParams P) { /* streaming lean step, HBM workspace */
  extern __shared__ __attribute__((aligned(16))) char fltx_hot[]; /* histogram & block scalars stay in LDS */
  decodeUtterance<255>(P, P.gws + (size_t)blockIdx.x * P.gwsStride, fltx_hot);
}
template <int W>
__global__ void __launch_bounds__(W) fltx_decode_kernel_gws(DecodeParams P) {
  /* HBM workspace; the histogram and block scalars -- and, when they fit, the
   * candidate records and the merge hash -- stay in LDS (carveWs splitHot) */
  extern __shared__ __attribute__((aligned(16))) char fltx_hot2[];
  decodeUtterance<0>(P, P.gws + (size_t)blockIdx.x * P.gwsStride, fltx_hot2);
}
#endif /* !FLTX_EMU */
