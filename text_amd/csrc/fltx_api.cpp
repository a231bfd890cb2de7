/*
 * fltx_api.cpp -- host side of the C ABI declared in include/fltx.h:
 * HBM allocation and layout, table flattening (n-gram LM, trie), kernel
 * launches, result retrieval.  Compiled by hipcc (-x hip) into
 * text_amd/lib/libfltx.so together with the kernels of fltx_kernels.h.
 *
 * There is no CPU decode path in this file: every decode launches the HIP
 * kernels and fails with FLTX_ERR_HIP if that is impossible.  (The FLTX_EMU
 * branches below exist only for tests/emu/libfltx_emu.so, a host-thread
 * emulation used to debug kernel logic; see tests/emu/hip_emu.h.)
 */
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <atomic>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "fltx.h"
#include "fltx_kernel_entry.h"

using namespace fltx;

/* Streams on the optimistic (LDS-sized) geometry: the parked beam of every stream is saved before a chunk
 * and put back for the streams whose chunk has to be decoded again on the general path (dir 1, map). */
struct SnapParams {
  int32_t K, dir;
  const int32_t* map; /* utterances to restore, null = all */
  double *gScore, *gAm, *gLm;
  float* gLexMax;
  uint32_t *gState, *gSPar, *gLex, *gTokPb;
  int32_t* gSEdge;
  int32_t *uttNBeam, *uttFrame, *uttTotal, *uttStatus;
  char* snap;
  int64_t B;
};
template <class T>
FLTX_HD void snapMove(int dir, T* live, char* snap, size_t& off, int64_t B, int32_t K, int b, int i) {
  T* s = (T*)(snap + off) + (size_t)b * K + i;
  T* l = live + (size_t)b * K + i;
  if (dir) {
    *l = *s;
  } else {
    *s = *l;
  }
  off += sizeof(T) * (size_t)B * K;
}
FLTX_HD void snapUtterance(const SnapParams& Q, int b, int i0, int step) {
  for (int i = i0; i < Q.K; i += step) {
    size_t off = 0;
    snapMove(Q.dir, Q.gScore, Q.snap, off, Q.B, Q.K, b, i);
    snapMove(Q.dir, Q.gAm, Q.snap, off, Q.B, Q.K, b, i);
    snapMove(Q.dir, Q.gLm, Q.snap, off, Q.B, Q.K, b, i);
    snapMove(Q.dir, Q.gLexMax, Q.snap, off, Q.B, Q.K, b, i);
    snapMove(Q.dir, Q.gState, Q.snap, off, Q.B, Q.K, b, i);
    snapMove(Q.dir, Q.gSPar, Q.snap, off, Q.B, Q.K, b, i);
    snapMove(Q.dir, Q.gLex, Q.snap, off, Q.B, Q.K, b, i);
    snapMove(Q.dir, Q.gTokPb, Q.snap, off, Q.B, Q.K, b, i);
    snapMove(Q.dir, Q.gSEdge, Q.snap, off, Q.B, Q.K, b, i);
  }
  if (i0 == 0) {
    int32_t* u = (int32_t*)(Q.snap + (size_t)Q.B * Q.K * (3 * 8 + 6 * 4)) + (size_t)b * 4;
    int32_t* live[4] = {Q.uttNBeam, Q.uttFrame, Q.uttTotal, Q.uttStatus};
    for (int k = 0; k < 4; ++k) {
      if (Q.dir) {
        live[k][b] = u[k];
      } else {
        u[k] = live[k][b];
      }
    }
  }
}
#ifndef FLTX_EMU
#define FLTX_INST(...) extern template __global__ void __VA_ARGS__(DecodeParams);
#include "fltx_instances.h"
#undef FLTX_INST
/* the front end of fltx_wlane.h: the token beam of every row of the batch, one wave per row */
__global__ void __launch_bounds__(256) fltx_tokbeam_kernel(DecodeParams P) {
  extern __shared__ __attribute__((aligned(16))) char fltx_tb_smem[];
  wlTokBeamRows(P, fltx_tb_smem);
}
/* streams: LM-state ids nothing can meet again go back to the utterance's free list (compactStates) */
__global__ void __launch_bounds__(1024) fltx_compact_states_kernel(CompactParams Q) {
  extern __shared__ __attribute__((aligned(16))) char fltx_cs_smem[];
  compactStates(Q, fltx_cs_smem);
}
/* host LM: the (LM state, index) questions of the next frame, one workgroup per utterance (hostLmQuestions) */
__global__ void __launch_bounds__(256) fltx_hostlm_questions_kernel(DecodeParams P) {
  extern __shared__ __attribute__((aligned(16))) char fltx_hq_smem[];
  hostLmQuestions(P, fltx_hq_smem);
}
__global__ void __launch_bounds__(512) fltx_backtrace_kernel(BacktraceParams P) {
  extern __shared__ __attribute__((aligned(16))) char fltx_bt_smem[];
  backtraceUtterance(P, fltx_bt_smem);
}
__global__ void __launch_bounds__(64) fltx_snapshot_kernel(SnapParams Q) {
  const int b = Q.map ? Q.map[blockIdx.x] : (int)blockIdx.x;
  snapUtterance(Q, b, (int)threadIdx.x, 64);
}
/* fltx_result_fetch_batch_compact: the rows that exist, tokens narrowed to bytes, packed back to back */
struct PackParams {
  const int32_t* tokens;
  const int32_t* words; /* null: lexicon-free */
  const int64_t* srcOff;
  const int64_t* dstOff;
  const int32_t* count; /* n_hyp[b] * length[b] */
  uint8_t* tok8;
  int32_t* wordsOut;
};
__global__ void __launch_bounds__(256) fltx_pack_results_kernel(PackParams Q) {
  const int b = (int)blockIdx.x;
  const int n = Q.count[b];
  const int32_t* src = Q.tokens + Q.srcOff[b];
  uint8_t* dst = Q.tok8 + Q.dstOff[b];
  for (int i = (int)threadIdx.x; i < n; i += (int)blockDim.x) {
    const int32_t v = src[i];
    dst[i] = v < 0 ? (uint8_t)0xFF : (uint8_t)v;
  }
  if (Q.words) {
    const int32_t* ws = Q.words + Q.srcOff[b];
    int32_t* wd = Q.wordsOut + Q.dstOff[b];
    for (int i = (int)threadIdx.x; i < n; i += (int)blockDim.x) {
      wd[i] = ws[i];
    }
  }
}
__global__ void __launch_bounds__(1024) fltx_streamop_kernel(StreamOpParams Q) {
  __shared__ int32_t sh[kStreamOpLds / 4];
  streamOpUtterance(Q, sh);
}
#endif

/* ------------------------------------------------------------------------ */
/* device back-end                                                           */
/* ------------------------------------------------------------------------ */
namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#ifdef FLTX_EMU
typedef void* Stream;
int devCount() { return 1; }
int devMalloc(void** p, size_t n) {
  *p = calloc(1, n ? n : 1);
  return *p ? 0 : 1;
}
void devFree(void* p) { free(p); }
int devMemset(void* p, int v, size_t n, Stream) {
  memset(p, v, n);
  return 0;
}
int devCopyH2D(void* d, const void* h, size_t n, Stream) {
  memcpy(d, h, n);
  return 0;
}
int devCopyD2H(void* h, const void* d, size_t n, Stream) {
  memcpy(h, d, n);
  return 0;
}
int devSync(Stream) { return 0; }
const char* devErr() { return "emu"; }
constexpr size_t kMaxLds = 160 * 1024;
#else
typedef hipStream_t Stream;
#define HIPCHK(x)                                                              \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess)                                                      \
      return fail(FLTX_ERR_HIP, "%s: %s", #x, hipGetErrorString(e_));          \
  } while (0)
int devMalloc(void** p, size_t n) { return hipMalloc(p, n ? n : 1) == hipSuccess ? 0 : 1; }
void devFree(void* p) { (void)hipFree(p); }
int devMemset(void* p, int v, size_t n, Stream s) { return hipMemsetAsync(p, v, n, s) == hipSuccess ? 0 : 1; }
int devCopyH2D(void* d, const void* h, size_t n, Stream s) {
  return hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s) == hipSuccess ? 0 : 1;
}
int devCopyD2H(void* h, const void* d, size_t n, Stream s) {
  if (hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, s) != hipSuccess) {
    return 1;
  }
  return hipStreamSynchronize(s) == hipSuccess ? 0 : 1;
}
int devSync(Stream s) { return hipStreamSynchronize(s) == hipSuccess ? 0 : 1; }
const char* devErr() { return hipGetErrorString(hipGetLastError()); }
constexpr size_t kMaxLds = 160 * 1024; /* gfx950: 160 KiB LDS per CU */

#endif

/* growable PINNED host buffer: results cross PCIe in one transfer per array */
struct HBuf {
  void* p = nullptr;
  size_t cap = 0;
  int ensure(size_t n) {
    if (n <= cap) {
      return 0;
    }
    release();
#ifdef FLTX_EMU
    p = malloc(n ? n : 1);
    if (!p) {
      return 1;
    }
#else
    if (hipHostMalloc(&p, n ? n : 1, hipHostMallocDefault) != hipSuccess) {
      p = nullptr;
      return 1;
    }
#endif
    cap = n;
    return 0;
  }
  void release() {
    if (p) {
#ifdef FLTX_EMU
      free(p);
#else
      (void)hipHostFree(p);
#endif
    }
    p = nullptr;
    cap = 0;
  }
  ~HBuf() { release(); }
};

/* growable device buffer */
struct DBuf {
  void* p = nullptr;
  size_t cap = 0;
  ~DBuf() {
    if (p) {
      devFree(p);
    }
  }
  /* returns 0 ok; `grew` tells the caller the contents are fresh (zeroed) */
  int ensure(size_t n, Stream s, bool zero, bool* grew = nullptr) {
    if (grew) {
      *grew = false;
    }
    if (n <= cap) {
      return 0;
    }
    if (p) {
      devSync(s);
      devFree(p);
      p = nullptr;
      cap = 0;
    }
    size_t want = n + n / 8;
    if (devMalloc(&p, want)) {
      p = nullptr;
      return 1;
    }
    cap = want;
    if (zero && devMemset(p, 0, want, s)) {
      return 1;
    }
    if (grew) {
      *grew = true;
    }
    return 0;
  }
  template <class T>
  T* as() const {
    return (T*)p;
  }
};

uint32_t nextPow2(uint64_t v) {
  uint64_t p = 1;
  while (p < v) {
    p <<= 1;
  }
  return (uint32_t)p;
}

} // namespace

/* ------------------------------------------------------------------------ */
/* objects                                                                   */
/* ------------------------------------------------------------------------ */
struct fltx_ctx {
  uint64_t uid = 0; /* never reused: per-context tables are keyed by it, not by the object's address */
  int device = 0;
  int numCUs = 256; /* MI355X; read from the device at creation */
  Stream stream = nullptr;
  bool ownStream = false;
};

/* every entry point that allocates, copies or launches selects its context's device first:
 * decoders of several devices are driven from several host threads (fltx_group_*) */
/* ... and puts the calling thread's device back when the entry point returns: the host framework's
 * later allocations and launches in that thread must not move to another GPU because of a call here */
struct DeviceScope {
  int prev = -1;
  bool failed = false;
  explicit DeviceScope(const fltx_ctx* ctx) {
#ifndef FLTX_EMU
    if (ctx) {
      if (hipGetDevice(&prev) != hipSuccess) {
        prev = -1;
      }
      if (prev == ctx->device) {
        prev = -1; /* nothing to restore */
      } else {
        failed = hipSetDevice(ctx->device) != hipSuccess;
      }
    }
#else
    (void)ctx;
#endif
  }
  ~DeviceScope() {
#ifndef FLTX_EMU
    if (prev >= 0) {
      (void)hipSetDevice(prev);
    }
#endif
  }
  DeviceScope(const DeviceScope&) = delete;
  DeviceScope& operator=(const DeviceScope&) = delete;
};

struct fltx_lm {
  fltx_ctx* ctx = nullptr;
  int kind = 0; /* 0 zero, 1 ngram, 2 host callbacks (fltx_lm_host_create) */
  fltx_host_lm host{};
  int order = 0;
  int32_t bos = 0, eos = 0, unk = 0, nUsr = 0;
  uint32_t mask = 0;
  /* device copies of the tables, one set per context (= per device) that has a decoder using this LM */
  struct Dev {
    DBuf tab, backoff, usrToLm;
    std::map<int, std::unique_ptr<DBuf>> tokDense; /* key = token count N: this device's copy of TokDense::tab */
  };
  /* The LM as ONE gather for a decoder over a small token set (fltx_slane.h's TL variant: a token-level n-gram LM on
   * the lexicon-free decoder, LexiconFreeDecoder.cpp:69-85 + KenLM.cpp:63-83): row c = a context the model can be
   * in (the vector of suffix node ids ngScore takes), reached from KenLM::start(false) by any token sequence; column
   * n < N = {float bits of lm.score(c, token n), row of the context after it}; column N = lm.finish(c).  Built on first use
   * per N by a breadth-first walk with the host twin of ngScore (lmStepWord): the scores are the ones the generic engine
   * computes, bit for bit.  nCtx < 0: the model has more contexts than the cap -- such decoders stay on the generic engine. */
  struct TokDense {
    int64_t nCtx = 0;
    std::vector<int2> tab; /* [nCtx][N + 1] */
  };
  std::map<int, std::unique_ptr<TokDense>> tokDense; /* guarded by devMu */
  std::mutex devMu;
  std::unordered_map<uint64_t, std::unique_ptr<Dev>> dev; /* key = fltx_ctx::uid */
  /* host copies for fltx_lm_score_sequence */
  std::vector<NgramSlot> hTab;
  std::vector<float> hBackoff;
  std::vector<int32_t> hUsr;
};

/* live LM objects: fltx_ctx_destroy drops the tables they hold for that context */
static std::mutex g_lmRegMu;
static std::unordered_set<fltx_lm*> g_lmReg;
static std::atomic<uint64_t> g_ctxUid{1};

struct fltx_trie {
  fltx_ctx* ctx = nullptr;
  int64_t nNodes = 0;
  int32_t nTokens = 0;
  DBuf edge, labels, mask;
  /* breadth-first re-layout for the lane = (LM state, trie node) engine (fltx_xlane.h): a node's
   * children are contiguous and in token order, so a child's id is the first child's id plus the
   * number of children with a smaller token; one 32-byte XNode per node, root = 0 */
  DBuf xnode, xdelta, xextra;
  bool xMulti = false;  /* some spelling has several words (Trie.h:19: up to 6 labels per node): fltx_ylane.h's LMK bit 2 */
  std::vector<int32_t> xEndHost; /* XNode::endLabel0 per node (host copy: decoders map it to LM word ids) */
  bool xOk = false;     /* the layout exists and the trie has the shape that engine assumes: */
  int32_t xEndTok = -1; /* every node that carries labels is entered by this one token (the word separator) */
  float xDeltaMin = 0.0f, xDeltaMax = 0.0f; /* range of the smearing differences of the layout */
  bool xZeroSmear = true; /* every maxScore is 0 (a lexicon without LM scores) */
};

struct fltx_decoder {
  fltx_ctx* ctx = nullptr;
  fltx_lm::Dev* lmDev = nullptr; /* this context's copy of the LM tables (null for ZeroLM) */
  int kind = 0;
  fltx_options opt{};
  const fltx_trie* trie = nullptr;
  const fltx_lm* lm = nullptr;
  int sil = 0, blank = 0, unk = 0, isLmToken = 0;
  int nTrans = 0;
  double transMax = 0.0; /* largest transition score, at least 0 */
  DBuf transitions;
  /* tunables */
  int threads = 0; /* 0 = pick per configuration (prepare()) */
  int userThreads = 0;
  int forceGlobalWs = 0;
  /* batch state */
  int B = 0, N = 0;
  bool streaming = false, ended = false, haveResults = false;
  int maxFrames = 0;
  std::vector<int32_t> T;       /* frames given in the last step */
  std::vector<int32_t> frames;  /* host mirror of uttFrame after sync */
  bool framesExact = true;       /* false: frames[] is an upper bound (lexicon prunes not yet asked about) */
  /* an optimistic stream chunk is launched and left alone: whether it has to be decoded again is looked at by the next
   * call that needs the beam (settleStream), so the next chunk's upload runs under this chunk's kernel */
  bool chunkPending = false, settling = false;
  const float* pendEmis = nullptr;
  int pendSlot = 0;
  int pendingPrune = -1;         /* prune(lookBack) asked for while the chunk was pending: runs right after the look */
  bool useLmCache = false;
  int noLmCache = 0;             /* tunable lm_cache = 0 */
  int deferRedo = 1;             /* tunable stream_defer: 0 = look at every chunk before fltx_stream_step returns */
  std::vector<int64_t> histOff; /* records */
  int64_t histRecords = 0;
  uint32_t stateCap = 0;
  uint32_t epoch = 0;
  int CAP = 0, HS = 0, NB = 0, SCAP = 0, dense = 0, noDense = 0;
  int lean = 0, noLean = 0; /* lean: GMAX of the lean lexicon-free kernel, 0 = generic engine */
  int lane = 0, noLane = 0; /* lane: tokens per wave of the lane-per-slot kernel (fltx_lane.h), 0 = off */
  int slaneThreads = 0; /* tuning: workgroup size of the lane = LM state kernel (0 = first that fits) */
  int wlane = 0, noWlane = 0; /* wlane: the lane = LM state kernel is fltx_wlane.h's (token sets beyond 64, token beam <= 64); slane = its list positions per wave */
  int slane = 0, noSlane = 0; /* slane: list positions per wave of the lane = LM state kernel (fltx_slane.h), 0 = off */
  /* ... its variant for a token-level n-gram LM (TL): the LM's dense (context, token) table on this device; noTlane:
   * tunable "tlane" = 0 (tests compare with the generic engine) */
  int tlane = 0, noTlane = 0, tlaneFirst = 0;
  int noTokDense = 0; /* tunable "tok_dense" = 0: no dense table at all (the generic engine probes the n-gram tables) */
  const int2* tokLm = nullptr;
  int64_t tokLmCtx = 0;
  int tlEdgeSlots = 4096, tlMaskSlots = 2048;
  bool tokLmTooBig = false; /* the model has more contexts than a dense table holds (why_not_lane: LM) */
  /* ... with several lane groups (fltx_mlane.h, beams beyond 64): lane groups (0 / 1 = fltx_slane.h), groups per token
   * wave, groups per self wave; userLaneGroups: tuning / tests, 0 = as many as the beam needs, -1 = never */
  int mlaneNG = 0, mlaneGPW = 0, mlaneSPW = 0, userLaneGroups = 0, userMlaneGeo = -1;
  int noYlaneAsg = 0;        /* tests: ASG lexicon decodes stay on the generic engine */
  int userYRankAt = 0;       /* tests: DecodeParams::yRankAt */
  int userTune = 0;          /* development: DecodeParams::tune */
  int userYlaneGroups = 0;   /* tests: at least this many lane groups on fltx_ylane.h (0 = what the beam needs) */
  uint32_t ymemoSlots = 8192; /* slots per utterance of the LM-state memo in HBM (fltx_ylane.h: follows the frames) */
  int64_t fallbackReasons = 0; /* bit r: an utterance of the last batch left fltx_ylane.h for reason r (see YL_WHY there) */
  int64_t whyNotLane = 0; /* FLTX_WHY_* bits: the eligibility terms that kept the last call off the lane engines (0 = it ran there) */
  bool preferYlane = false;
  bool genericAsked = false;  /* fltx_decoder_set touched a tunable of the generic engine */
  int engineFirst = 0;
  /* diagnostics of the engine a call STARTED on (a retry's prepare() must not overwrite them) */
  int64_t whyFirst = 0;
  int wlaneFirst = 0, slaneFirst = 0, laneGroupsFirst = 0, xlaneFirst = 0, ylaneFirst = 0;
  int lastRedo = 0;           /* utterances of the last offline call that had to be decoded again on a general path */
  int packedBits = 8;         /* width of the parent-slot field of those records (fltx_mlane.h: 10) */
  bool batchPacked = false;   /* some utterance of the current results has packed history records (ST_PACKED) */
  bool batchTlane = false;    /* ... by fltx_slane.h's token-LM variant (the back-trace re-accumulates the LM score as well) */
  bool batchWlane = false;    /* ... written by fltx_wlane.h (tokens beyond a byte, emissions gathered by the back-trace) */
  DBuf xlmword;               /* fltx_ylane.h: LM word id of XNode::endLabel0 per node (this decoder's trie x LM) */
  const fltx_trie* xlmwordTrie = nullptr;
  const fltx_lm* xlmwordLm = nullptr;
  int ylane = 0, noYlane = 0, ylaneLm = 0, ylaneRounds = 0, ylaneTpw = 0; /* ylane: lane groups of fltx_ylane.h (0 = not used) */
  int btLdsKb = 0;
  int streamTotalFrames = 0; /* frames a stream will decode in all (sizes its LM-state id tables), 0 = default */
  /* stream chunks of the lexicon-free decoder on the lane = LM state engine (fltx_slane.h, ST): list positions per
   * token wave (0 = not used) and threads; begin / end / prune / best stay the lane-per-slot engine's */
  int sstream = 0, sstreamThreads = 0, noSstream = 0;
  int tstream = 0; /* ... the stream's chunks run on fltx_slane.h's token-LM variant (ids from the generic engine's table) */
  bool sstreamLaunch = false;
  /* streams of the lexicon decoder on the optimistic geometry (LDS workspace, cut-off generation): a chunk that
   * overflows is decoded again from the saved beam on the general path (HBM workspace) */
  int userStreamOpt = 1;
  bool streamOpt = false;
  int streamRedone = 0; /* stream-chunks decoded again since fltx_stream_begin */
  DBuf snap;
  int yshare = 0, userYshare = -1; /* the geometry of fltx_ylane.h that shares a CU (memo in HBM); user: -1 = when the batch exceeds the CUs */
  DBuf ymemo;
  DBuf lmCache;               /* DecodeParams::lmCache */
  int xlane = 0, noXlane = 0; /* xlane: list positions per token wave of the lane = (LM state, trie node) kernel (fltx_xlane.h) */
  bool offlineCall = false;   /* prepare() is sizing an fltx_decode_batch (begin + frames + end in one launch) */
  /* "defer_check": fltx_decode_batch returns with its kernels queued; the look at the utterances' statuses (and the
   * second pass of what a fast path flagged) waits for the first call that reads results */
  int deferCheck = 0; /* 1: an unread batch is settled by the next fltx_decode_batch; 2: dropped there (counted in looksDropped) */
  bool offlinePending = false;
  int64_t looksDropped = 0;  /* batches whose statuses nobody ever looked at ("looks_dropped"; defer_check = 2 only) */
  int64_t unreadRedone = 0;  /* utterances decoded again while settling batches the caller never read ("unread_redone") */
  std::vector<int32_t> offT;
  std::vector<int64_t> offOffsets;
  int offN = 0, offUpSlot = 0;
  const float* offEmis = nullptr;
  const float* lastEmis = nullptr; /* device emissions of the last offline batch (the back-trace re-reads them) */
  size_t hotBytes = 0; /* LDS part of a split (HBM + LDS) workspace */
  int hotLevel = 0;    /* what the LDS part holds (carveWs) */
  size_t ldsBudget = 0; /* tests: smaller LDS than the hardware's */
  int maxHotLevel = 2;
  int itemCap = 0, noItems = 0, itemWide = 0; /* lexicon decoder: list of existing (hypothesis, token) children */
  int CAP2 = 0, cutM = 0, noCut = 0, userCutM = 0;
  int cutRecompute = 0, noSlim = 0; /* cut-off generation without the slim list (beams it does not fit) */ /* lexicon decoder: slim score-pass list + cut-off (runFrame) */
  size_t wsBytes = 0;
  bool wsInLds = true;
  /* device buffers */
  /* what a step uploads, in two slots taking turns: the upload of step k + 1 runs on a copy stream under the kernel of
   * step k (which reads the other slot) */
  DBuf emis[2], emOff[2], stepT[2];
  int upSlot = 0;
  DBuf histOffD, histPT, histW, stateTab, stateCtx;
  DBuf tokRowsBuf; /* fltx_wlane.h: the token beams of all rows (fltx_tokbeam_kernel) */
  int wlMaxT = 0;  /* ... longest utterance of the call (the front-end kernel's grid) */
  DBuf gScore, gAm, gLm, gState, gSPar, gSEdge, gLex, gTokPb;
  DBuf uttNBeam, uttFrame, uttTotal, uttStatus, outN, outScores, gws;
  DBuf childTab, maskTab, uttNextId, gMask, gLexMax;
  /* streams: recycled LM-state ids (DecodeParams::idPar ...) */
  DBuf idPar, idEdge, idBorn, idKeep, idNew, idList, stateVal;
  bool recycle = false;       /* this stream hands ids out from free lists */
  int idFamily = 0;           /* 0: childTab / maskTab engines, 1: generic engine (stateTab + stateVal) */
  int64_t idsUsedBound = 0;   /* upper bound of the ids any stream has taken since its free list was last rebuilt */
  int64_t idsFreeBound = 0;   /* lower bound of what a rebuilt free list holds (beam x max_frames) */
  int compactions = 0;        /* fltx_compact_states_kernel launches since fltx_stream_begin ("compactions") */
  int userCompactAlways = 0;  /* tests ("compact_always"): rebuild the free lists before every chunk */
  DBuf scored; /* n-gram LM queries per utterance (accounting) */
  /* host LM (lm->kind == 2): question lists and beam states in pinned host memory (written by the kernel), the
   * answer tables staged in pinned memory and uploaded once per frame */
  HBuf hlmQCount, hlmQ, hlmBeamN, hlmBeam, hlmStage;
  DBuf hlmTabD;
  int hlmQCap = 0;
  bool hlmAnnounce = false; /* a frame has been decoded since decodeBegin: the beam is announced through update_cache */
  int64_t hlmAsked = 0, hlmDistinct = 0; /* questions listed / answered by the callbacks since decodeBegin ("hlm_asked", "hlm_distinct") */
  bool keepScored = false;
  int64_t idCap = 0;
  DBuf tokens, words, prof, histS, bestLen, bestScores, bestTok, bestWrd;
  HBuf hTokens, hWords, hScores; /* fltx_result_fetch_batch */
  HBuf hTok8, hWordsC, hPackMeta;  /* fltx_result_fetch_batch_compact */
  HBuf hSync;                      /* syncResults */
  HBuf hStat;                      /* DecodeParams::statusHost */
  DBuf dTok8, dWordsC, dPackMeta;
  std::vector<int64_t> packOff;
  bool compactFetched = false, scoresFetched = false;
  DBuf uttMap;                   /* utterances of a partial re-run */
  int nLaunch = 0;               /* workgroups of the next decode launch (0: all B) */
  std::vector<int32_t> hLen, hNHyp;
  bool hostFetched = false;
  int keepScores = 0, userKeepScores = 0; /* (streams force the score history on; offline decodes use the caller's choice) */
  int profile = 0, profWave = 0;
  /* host caches of the last results */
  std::vector<int32_t> hN, hFrame, hStatus;
  bool resultsSynced = false;
  bool backtraced = false;
  int64_t statFrames = 0, statBytes = 0, statDecodeBytes = 0, statEpilogueBytes = 0, statLmBytes = 0;
  bool statsDone = false;
#ifndef FLTX_EMU
  /* HIP events on the launch stream bracketing the two kernels of the last
   * fltx_decode_batch (bench.py's roofline leg reads them) */
  hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
  hipStream_t copyStream = nullptr;
  hipEvent_t evSlotFree[2] = {nullptr, nullptr}; /* recorded on the launch stream once everything that reads a slot is queued */
  bool slotUsed[2] = {false, false};
#endif
  bool timed = false;
};

/* ------------------------------------------------------------------------ */
extern "C" {

const char* fltx_last_error(void) { return g_err.c_str(); }
/* internal: lets the other translation units record an error message */
__attribute__((visibility("hidden"))) int fltx_set_error_(int code, const char* msg) {
  g_err = msg ? msg : "";
  return code;
}
/* internal (fltx_group.cpp): 0 ZeroLM, 1 n-gram tables, 2 host callbacks */
__attribute__((visibility("hidden"))) int fltx_lm_kind_(const fltx_lm* lm) { return lm ? lm->kind : -1; }
const char* fltx_version(void) {
#ifdef FLTX_EMU
  return "fltx 0.1 (host-thread emulation, tests only)";
#else
  return "fltx 0.1 (HIP gfx950)";
#endif
}

int fltx_ctx_create(int device, void* stream, fltx_ctx** out) {
  if (!out) {
    return fail(FLTX_ERR_INVALID, "fltx_ctx_create: out is null");
  }
  auto* c = new fltx_ctx();
  c->uid = g_ctxUid.fetch_add(1);
#ifndef FLTX_EMU
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n == 0) {
    delete c;
    return fail(FLTX_ERR_HIP, "fltx_ctx_create: no HIP device available (this library has no CPU path)");
  }
  if (device < 0) {
    if (hipGetDevice(&device) != hipSuccess) {
      delete c;
      return fail(FLTX_ERR_HIP, "hipGetDevice failed");
    }
  }
  if (device >= n) {
    delete c;
    return fail(FLTX_ERR_INVALID, "fltx_ctx_create: device %d out of range (%d devices)", device, n);
  }
  c->device = device;
  DeviceScope devScope(c);
  if (devScope.failed) {
    delete c;
    return fail(FLTX_ERR_HIP, "hipSetDevice(%d) failed", device);
  }
  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) {
      c->numCUs = cus;
    }
  }
  if (stream) {
    c->stream = (hipStream_t)stream;
  } else {
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
      delete c;
      return fail(FLTX_ERR_HIP, "hipStreamCreate failed");
    }
    c->ownStream = true;
  }
#else
  (void)device;
  (void)stream;
#endif
  *out = c;
  return FLTX_OK;
}

int fltx_ctx_destroy(fltx_ctx* ctx) {
  if (!ctx) {
    return FLTX_OK;
  }
  DeviceScope devScope(ctx);
  { /* the n-gram tables uploaded for this context die with it (a later context may get this address) */
    std::lock_guard<std::mutex> reg(g_lmRegMu);
    for (fltx_lm* lm : g_lmReg) {
      std::lock_guard<std::mutex> lock(lm->devMu);
      lm->dev.erase(ctx->uid);
    }
  }
#ifndef FLTX_EMU
  if (ctx->ownStream) {
    (void)hipStreamDestroy(ctx->stream);
  }
#endif
  delete ctx;
  return FLTX_OK;
}

int fltx_ctx_synchronize(fltx_ctx* ctx) {
  if (!ctx) {
    return fail(FLTX_ERR_INVALID, "null ctx");
  }
  return devSync(ctx->stream) ? fail(FLTX_ERR_HIP, "stream synchronize failed: %s", devErr()) : FLTX_OK;
}

void* fltx_ctx_stream(fltx_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }
uint64_t fltx_ctx_uid(fltx_ctx* ctx) { return ctx ? ctx->uid : 0; }

/* ---- LM ------------------------------------------------------------------ */
int fltx_lm_zero_create(fltx_ctx* ctx, fltx_lm** out) {
  if (!out) {
    return fail(FLTX_ERR_INVALID, "fltx_lm_zero_create: null argument");
  }
  auto* lm = new fltx_lm();
  lm->ctx = ctx;
  lm->kind = 0;
  {
    std::lock_guard<std::mutex> reg(g_lmRegMu);
    g_lmReg.insert(lm);
  }
  *out = lm;
  return FLTX_OK;
}

/* a user subclass of LM behind callbacks (include/fltx.h): nothing to upload */
int fltx_lm_host_create(const fltx_host_lm* cb, fltx_lm** out) {
  if (!cb || !out || !cb->start || !cb->score) {
    return fail(FLTX_ERR_INVALID, "fltx_lm_host_create: null argument (start and score are required)");
  }
  auto* lm = new fltx_lm();
  lm->kind = 2;
  lm->host = *cb;
  {
    std::lock_guard<std::mutex> lock(g_lmRegMu);
    g_lmReg.insert(lm);
  }
  *out = lm;
  return FLTX_OK;
}

int fltx_lm_ngram_create(fltx_ctx* ctx, int32_t order, int64_t nNgrams, const int32_t* ngOrder,
                         const int32_t* ngWords, const float* prob, const float* backoff,
                         const int32_t* usrToLm, int32_t nUsr, int32_t bos, int32_t eos,
                         int32_t unk, fltx_lm** out) {
  if (!out || !ngOrder || !ngWords || !prob || !backoff || nNgrams <= 0) {
    return fail(FLTX_ERR_INVALID, "fltx_lm_ngram_create: null argument");
  }
  if (order < 1 || order > kMaxNgramOrder) {
    return fail(FLTX_ERR_UNSUPPORTED, "n-gram order %d outside 1..%d (FL_TEXT_KENLM_MAX_ORDER)", order,
                kMaxNgramOrder);
  }
  if (nUsr < 0 || (nUsr > 0 && !usrToLm)) {
    return fail(FLTX_ERR_INVALID, "fltx_lm_ngram_create: usr_to_lm is null for n_usr = %d", nUsr);
  }
  for (int32_t u = 0; u < nUsr; ++u) {
    if (usrToLm[u] < 0) {
      return fail(FLTX_ERR_INVALID, "fltx_lm_ngram_create: usr_to_lm[%d] = %d is negative", u, usrToLm[u]);
    }
  }
  /* forward trie of n-grams: node id per n-gram, (context node, word) -> node */
  struct HostNode {
    float prob;
    float backoff;
    bool phantom;
  };
  std::vector<HostNode> nodes(1, HostNode{0, 0, true}); /* node 0 = empty context */
  std::unordered_map<uint64_t, uint32_t> kids;
  kids.reserve((size_t)nNgrams * 2);
  std::vector<uint32_t> entCtx, entWord, entNode;
  auto childOf = [&](uint32_t ctxNode, uint32_t word, bool create) -> uint32_t {
    uint64_t k = ((uint64_t)ctxNode << 32) | word;
    auto it = kids.find(k);
    if (it != kids.end()) {
      return it->second;
    }
    if (!create) {
      return 0;
    }
    uint32_t id = (uint32_t)nodes.size();
    nodes.push_back(HostNode{0, 0, true});
    kids.emplace(k, id);
    entCtx.push_back(ctxNode);
    entWord.push_back(word);
    entNode.push_back(id);
    return id;
  };
  for (int64_t i = 0; i < nNgrams; ++i) {
    int k = ngOrder[i];
    if (k < 1 || k > order) {
      return fail(FLTX_ERR_INVALID, "n-gram %lld has order %d", (long long)i, k);
    }
    const int32_t* wds = ngWords + i * order;
    uint32_t node = 0;
    for (int j = 0; j < k; ++j) {
      if (wds[j] < 0) {
        return fail(FLTX_ERR_INVALID, "n-gram %lld has a negative word id", (long long)i);
      }
      node = childOf(node, (uint32_t)wds[j], true); /* missing prefixes become phantoms */
    }
    nodes[node].prob = prob[i];
    nodes[node].backoff = backoff[i];
    nodes[node].phantom = false;
  }
  if (nodes.size() >= 0x7FFFFFFFull) {
    return fail(FLTX_ERR_UNSUPPORTED, "too many n-grams");
  }
  auto* lm = new fltx_lm();
  lm->ctx = ctx;
  lm->kind = 1;
  lm->order = order;
  lm->bos = bos;
  lm->eos = eos;
  lm->unk = unk;
  lm->nUsr = nUsr;
  /* (4x slots: 2^31 slots of 16 B is what the 32-bit slot index reaches; a model beyond 2^29 n-grams takes 2x) */
  const uint64_t wantSlots = (uint64_t)entNode.size() * ((uint64_t)entNode.size() * 4 + 16 <= (1ull << 31) ? 4 : 2) + 16;
  if (wantSlots > (1ull << 31)) {
    return fail(FLTX_ERR_UNSUPPORTED, "too many n-grams for the 32-bit slot index of the look-up table");
  }
  uint32_t cap = nextPow2(wantSlots); /* at most a quarter full: a look-up is a chain of dependent trips to HBM, one more per occupied slot it meets (C4: 2x -> 4x slots = -4 % on the whole kernel) */
  lm->mask = cap - 1;
  lm->hTab.assign(cap, NgramSlot{0, kEmpty, 0, 0.0f});
  for (size_t e = 0; e < entNode.size(); ++e) {
    uint32_t s = hashKey(entCtx[e], entWord[e], 0x5bd1e995u, 0) & lm->mask;
    while (lm->hTab[s].word != kEmpty) {
      s = (s + 1) & lm->mask;
    }
    const HostNode& nd = nodes[entNode[e]];
    lm->hTab[s] = NgramSlot{entCtx[e], entWord[e], entNode[e] | (nd.phantom ? kPhantomNode : 0u), nd.prob};
  }
  lm->hBackoff.resize(nodes.size());
  for (size_t i = 0; i < nodes.size(); ++i) {
    lm->hBackoff[i] = nodes[i].phantom ? 0.0f : nodes[i].backoff;
  }
  lm->hUsr.assign(usrToLm, usrToLm + (usrToLm ? nUsr : 0));
  /* the tables go to HBM when a decoder is created on a context */
  {
    std::lock_guard<std::mutex> reg(g_lmRegMu);
    g_lmReg.insert(lm);
  }
  *out = lm;
  return FLTX_OK;
}

int fltx_lm_destroy(fltx_lm* lm) {
  if (lm) {
    std::lock_guard<std::mutex> reg(g_lmRegMu);
    g_lmReg.erase(lm);
  }
  delete lm;
  return FLTX_OK;
}

/* host walk over the same flat tables the kernels use (known-answer checks of
 * the table builder; decode-time scoring always happens on the device) */
int fltx_lm_score_sequence(fltx_lm* lm, const int32_t* usrWords, int32_t n, int32_t withFinish,
                           float* perWord, float* total) {
  if (!lm || (!usrWords && n > 0)) {
    return fail(FLTX_ERR_INVALID, "fltx_lm_score_sequence: null argument");
  }
  if (lm->kind == 2) {
    return fail(FLTX_ERR_UNSUPPORTED, "a host LM keeps its own states (call the LM object)");
  }
  float tot = 0;
  if (lm->kind == 0) {
    for (int i = 0; i < n; ++i) {
      if (perWord) {
        perWord[i] = 0.0f;
      }
    }
    if (total) {
      *total = 0;
    }
    return FLTX_OK;
  }
  const int L = lm->order - 1;
  auto find = [&](uint32_t ctx, uint32_t word, uint32_t& node, float& pr) {
    uint32_t s = hashKey(ctx, word, 0x5bd1e995u, 0) & lm->mask;
    for (;;) {
      const NgramSlot& e = lm->hTab[s];
      if (e.word == kEmpty) {
        return false;
      }
      if (e.ctx == ctx && e.word == word) {
        node = e.node;
        pr = e.prob;
        return true;
      }
      s = (s + 1) & lm->mask;
    }
  };
  std::vector<int32_t> ctx(std::max(L, 1), 0), nxt(std::max(L, 1), 0);
  {
    uint32_t nd;
    float pr;
    if (L > 0 && find(0, (uint32_t)lm->bos, nd, pr)) {
      ctx[0] = (int32_t)(nd & ~kPhantomNode);
    }
  }
  auto score = [&](uint32_t word) {
    std::vector<uint32_t> nodes(L + 1, 0);
    std::vector<char> found(L + 1, 0);
    float prob = 0;
    int longest = -1;
    for (int k = 0; k <= L; ++k) {
      uint32_t c = k == 0 ? 0u : (uint32_t)ctx[k - 1];
      if (k > 0 && c == 0) {
        continue;
      }
      float pr;
      if (find(c, word, nodes[k], pr)) {
        found[k] = 1;
        if (!(nodes[k] & kPhantomNode)) {
          longest = k;
          prob = pr;
        }
        nodes[k] &= ~kPhantomNode;
      }
    }
    if (longest < 0) {
      uint32_t nd = 0;
      if (!find(0, (uint32_t)lm->unk, nd, prob)) {
        prob = -100.0f;
      }
      nodes[0] = nd & ~kPhantomNode;
      found[0] = 1;
      longest = 0;
    }
    for (int j = longest + 1; j <= L; ++j) {
      uint32_t c = (uint32_t)ctx[j - 1];
      if (c != 0) {
        prob += lm->hBackoff[c];
      }
    }
    for (int j = 0; j < L; ++j) {
      nxt[j] = (j <= longest && found[j]) ? (int32_t)nodes[j] : 0;
    }
    ctx = nxt;
    return prob;
  };
  for (int i = 0; i < n; ++i) {
    int u = usrWords[i];
    if (u < 0 || u >= lm->nUsr) {
      return fail(FLTX_ERR_RANGE, "[ngram LM] Invalid user token index: %d", u); /* KenLM.cpp:66-69 */
    }
    float s = score((uint32_t)lm->hUsr[u]);
    if (perWord) {
      perWord[i] = s;
    }
    tot += s;
  }
  if (withFinish) {
    tot += score((uint32_t)lm->eos);
  }
  if (total) {
    *total = tot;
  }
  return FLTX_OK;
}

/* LM::start / LM::score / LM::finish on explicit states (decoder/lm/LM.h:61-78),
 * host walk over the flat tables.  A state is lmOrder-1 context node ids. */
static bool lmFind(const fltx_lm* lm, uint32_t ctx, uint32_t word, uint32_t& node, float& pr) {
  uint32_t s = hashKey(ctx, word, 0x5bd1e995u, 0) & lm->mask;
  for (;;) {
    const NgramSlot& e = lm->hTab[s];
    if (e.word == kEmpty) {
      return false;
    }
    if (e.ctx == ctx && e.word == word) {
      node = e.node;
      pr = e.prob;
      return true;
    }
    s = (s + 1) & lm->mask;
  }
}

int fltx_lm_state_size(fltx_lm* lm, int32_t* n) {
  if (!lm || !n) {
    return fail(FLTX_ERR_INVALID, "null argument");
  }
  if (lm->kind == 2) {
    return fail(FLTX_ERR_UNSUPPORTED, "a host LM keeps its own states (call the LM object)");
  }
  *n = lm->kind == 0 ? 0 : std::max(1, lm->order - 1);
  return FLTX_OK;
}

int fltx_lm_start(fltx_lm* lm, int32_t startWithNothing, int32_t* ctxOut) {
  if (!lm) {
    return fail(FLTX_ERR_INVALID, "null lm");
  }
  if (lm->kind == 2) {
    return fail(FLTX_ERR_UNSUPPORTED, "a host LM keeps its own states (call the LM object)");
  }
  if (lm->kind == 0) {
    return FLTX_OK;
  }
  const int L = std::max(1, lm->order - 1);
  for (int j = 0; j < L; ++j) {
    ctxOut[j] = 0;
  }
  uint32_t nd;
  float pr;
  if (!startWithNothing && lm->order > 1 && lmFind(lm, 0, (uint32_t)lm->bos, nd, pr)) {
    ctxOut[0] = (int32_t)(nd & ~kPhantomNode); /* BeginSentenceWrite, KenLM.cpp:57 */
  }
  return FLTX_OK;
}

/* log10 p(word | context) and the context after it: the host twin of ngScore (fltx_kernels.h), same tables, same
 * order of the float additions.  ctxOut may alias ctxIn. */
static float lmStepWord(const fltx_lm* lm, const int32_t* ctxIn, uint32_t word, int32_t* ctxOut) {
  const int L = lm->order - 1;
  uint32_t nodes[kMaxNgramOrder] = {0};
  bool found[kMaxNgramOrder] = {false};
  float prob = 0;
  int longest = -1;
  for (int k = 0; k <= L; ++k) {
    const uint32_t c = k == 0 ? 0u : (uint32_t)ctxIn[k - 1];
    if (k > 0 && c == 0) {
      continue;
    }
    float pr;
    if (lmFind(lm, c, word, nodes[k], pr)) {
      found[k] = true;
      if (!(nodes[k] & kPhantomNode)) {
        longest = k;
        prob = pr;
      }
      nodes[k] &= ~kPhantomNode;
    }
  }
  if (longest < 0) {
    uint32_t nd = 0;
    if (!lmFind(lm, 0, (uint32_t)lm->unk, nd, prob)) {
      prob = -100.0f;
    }
    nodes[0] = nd & ~kPhantomNode;
    found[0] = true;
    longest = 0;
  }
  for (int j = longest + 1; j <= L; ++j) {
    const uint32_t c = (uint32_t)ctxIn[j - 1];
    if (c != 0) {
      prob += lm->hBackoff[c];
    }
  }
  if (ctxOut) {
    int32_t tmp[kMaxNgramOrder];
    for (int j = 0; j < L; ++j) {
      tmp[j] = (j <= longest && found[j]) ? (int32_t)nodes[j] : 0;
    }
    for (int j = 0; j < std::max(1, L); ++j) {
      ctxOut[j] = j < L ? tmp[j] : 0;
    }
  }
  return prob;
}

/* usr_idx >= 0: LM::score; usr_idx == -1: LM::finish (scores </s>) */
int fltx_lm_step(fltx_lm* lm, const int32_t* ctxIn, int32_t usrIdx, int32_t* ctxOut, float* score) {
  if (!lm || !score) {
    return fail(FLTX_ERR_INVALID, "null argument");
  }
  if (lm->kind == 2) {
    return fail(FLTX_ERR_UNSUPPORTED, "a host LM keeps its own states (call the LM object)");
  }
  if (lm->kind == 0) {
    *score = 0.0f;
    return FLTX_OK;
  }
  uint32_t word;
  if (usrIdx == -1) {
    word = (uint32_t)lm->eos;
  } else {
    if (usrIdx < 0 || usrIdx >= lm->nUsr) {
      return fail(FLTX_ERR_RANGE, "[ngram LM] Invalid user token index: %d", usrIdx); /* KenLM.cpp:66-69 */
    }
    word = (uint32_t)lm->hUsr[usrIdx];
  }
  *score = lmStepWord(lm, ctxIn, word, ctxOut);
  return FLTX_OK;
}

/* ---- trie ---------------------------------------------------------------- */
int fltx_trie_create(fltx_ctx* ctx, int64_t nNodes, int32_t nTokens, const int32_t* child,
                     const float* maxScore, const int32_t* labelOff, const int32_t* labels,
                     fltx_trie** out) {
  if (!ctx || !out || !child || !maxScore || !labelOff || nNodes <= 0 || nTokens <= 0) {
    return fail(FLTX_ERR_INVALID, "fltx_trie_create: null or empty argument");
  }
  if (nNodes >= (1ll << 31)) {
    return fail(FLTX_ERR_UNSUPPORTED, "trie too large");
  }
  DeviceScope devScope(ctx);
  if (devScope.failed) {
    return fail(FLTX_ERR_HIP, "hipSetDevice failed");
  }
  if (nNodes >= (1ll << 28) || (int64_t)labelOff[nNodes] >= (1ll << 28)) {
    return fail(FLTX_ERR_UNSUPPORTED, "trie too large for the packed edge records");
  }
  std::vector<int32_t> nKids((size_t)nNodes, 0);
  for (int64_t i = 0; i < nNodes; ++i) {
    for (int t = 0; t < nTokens; ++t) {
      const int32_t c = child[i * nTokens + t];
      if (c >= nNodes) {
        return fail(FLTX_ERR_RANGE, "trie child index %d out of range", c);
      }
      nKids[i] += c >= 0;
    }
    const int nl = labelOff[i + 1] - labelOff[i];
    if (nl < 0 || nl > 6) {
      return fail(FLTX_ERR_INVALID, "trie node %lld has %d labels (kTrieMaxLabel = 6)", (long long)i, nl);
    }
  }
  std::vector<TrieEdge> edge((size_t)nNodes * nTokens);
  for (int64_t i = 0; i < nNodes; ++i) {
    for (int t = 0; t < nTokens; ++t) {
      const int32_t c = child[i * nTokens + t];
      TrieEdge e{-1, 0.0f, -1, 0u};
      if (c >= 0) {
        const int nl = labelOff[c + 1] - labelOff[c];
        e.child = c;
        e.childMax = maxScore[c];
        e.label0 = nl > 0 ? labels[labelOff[c]] : -1;
        e.meta = (uint32_t)nl | (nKids[c] > 0 ? 8u : 0u) | ((uint32_t)labelOff[c] << 4);
      }
      edge[(size_t)i * nTokens + t] = e;
    }
  }
  /* per node: which tokens have a child (the generic engine lists only those) */
  std::vector<unsigned long long> cmask;
  if (nTokens <= 64) {
    cmask.assign((size_t)nNodes, 0ull);
    for (int64_t i = 0; i < nNodes; ++i) {
      for (int t = 0; t < nTokens; ++t) {
        if (child[i * nTokens + t] >= 0) {
          cmask[(size_t)i] |= 1ull << t;
        }
      }
    }
  }
  auto* t = new fltx_trie();
  t->ctx = ctx;
  t->nNodes = nNodes;
  t->nTokens = nTokens;
  Stream st = ctx->stream;
  size_t nLab = (size_t)labelOff[nNodes];
  /* breadth-first layout (see fltx_trie::xnode) */
  std::vector<XNode> xn;
  std::vector<float> xdHost;
  if (nTokens <= 64 && !cmask.empty()) {
    std::vector<int64_t> order; /* new id -> old id */
    std::vector<int32_t> tokOf((size_t)nNodes, -1);
    std::vector<uint32_t> parentOf((size_t)nNodes + 1, 0u); /* by new id */
    order.reserve((size_t)nNodes);
    order.push_back(0);
    xn.resize((size_t)nNodes);
    bool ok = true, multi = false;
    int endTok = -1;
    bool zero = true;
    std::vector<uint32_t> xextraHost((size_t)nNodes, 0u);
    std::vector<uint8_t> seen((size_t)nNodes, 0);
    seen[0] = 1;
    for (size_t q = 0; q < order.size() && ok; ++q) {
      const int64_t o = order[q];
      XNode x;
      memset(&x, 0, sizeof(x));
      x.childMask = cmask[(size_t)o];
      x.firstChild = (uint32_t)order.size();
      x.maxScore = maxScore[o];
      x.endLabel0 = -1;
      x.parent = parentOf[q];
      zero = zero && maxScore[o] == 0.0f;
      for (int tk = 0; tk < nTokens; ++tk) {
        const int32_t c = child[o * nTokens + tk];
        if (c < 0) {
          continue;
        }
        if (seen[(size_t)c] || order.size() >= (size_t)nNodes) { /* not a tree (shared suffix, back edge): */
          ok = false;                                            /* no breadth-first layout, generic engine only */
          break;
        }
        seen[(size_t)c] = 1;
        parentOf[order.size()] = (uint32_t)q;
        order.push_back(c);
        tokOf[(size_t)c] = tk;
        if (nKids[(size_t)c] > 0) {
          x.kidsMask |= 1ull << tk;
        }
        const int nl = labelOff[c + 1] - labelOff[c];
        if (nl > 0) {
          if (endTok < 0) {
            endTok = tk;
          }
          ok = ok && tk == endTok && nl <= 6 && labelOff[c] < (1 << 28); /* one word-ending token; Trie.h:19: at most 6 words per spelling */
          x.endLabel0 = labels[labelOff[c]];
          xextraHost[q] = ((uint32_t)labelOff[c] << 3) | (uint32_t)nl;
          multi = multi || nl > 1;
        }
      }
      xn[q] = x;
    }
    std::vector<float> xd(xn.size(), 0.0f);
    for (size_t q = 1; ok && q < order.size() && q < xd.size(); ++q) { /* lex->maxScore - lexMaxScore, LexiconDecoder.cpp:47,96 */
      const uint32_t pq = xn[q].parent;
      xd[q] = xn[q].maxScore - (pq == 0u ? 0.0f : xn[pq].maxScore);
    }
    ok = ok && order.size() == (size_t)nNodes && (labelOff[1] - labelOff[0]) == 0; /* a tree; no label on the root */
    t->xOk = ok;
    t->xMulti = ok && multi;
    t->xEndTok = endTok;
    if (ok && multi) {
      if (t->xextra.ensure(sizeof(uint32_t) * xextraHost.size(), st, false) ||
          devCopyH2D(t->xextra.p, xextraHost.data(), sizeof(uint32_t) * xextraHost.size(), st) || devSync(st)) {
        delete t;
        return fail(FLTX_ERR_OOM, "trie upload: labels of the spellings");
      }
    }
    t->xZeroSmear = zero;
    if (!ok) {
      xn.clear();
      xd.clear();
    }
    for (float v : xd) {
      t->xDeltaMin = std::min(t->xDeltaMin, v);
      t->xDeltaMax = std::max(t->xDeltaMax, v);
    }
    t->xEndHost.resize(xn.size());
    for (size_t q = 0; q < xn.size(); ++q) {
      t->xEndHost[q] = xn[q].endLabel0;
    }
    xdHost.swap(xd);
  }
  if (!xn.empty()) {
    if (t->xnode.ensure(sizeof(XNode) * xn.size(), st, false) ||
        devCopyH2D(t->xnode.p, xn.data(), sizeof(XNode) * xn.size(), st)) {
      delete t;
      return fail(FLTX_ERR_OOM, "trie: breadth-first layout upload failed");
    }
    if (t->xdelta.ensure(sizeof(float) * xdHost.size(), st, false) ||
        devCopyH2D(t->xdelta.p, xdHost.data(), sizeof(float) * xdHost.size(), st)) {
      delete t;
      return fail(FLTX_ERR_OOM, "trie: breadth-first layout upload failed");
    }
  }
  if (!cmask.empty()) {
    if (t->mask.ensure(8 * cmask.size(), st, false) ||
        devCopyH2D(t->mask.p, cmask.data(), 8 * cmask.size(), st)) {
      delete t;
      return fail(FLTX_ERR_OOM, "trie: child-mask upload failed");
    }
  }
  if (t->edge.ensure(sizeof(TrieEdge) * edge.size(), st, false) ||
      t->labels.ensure(sizeof(int32_t) * std::max<size_t>(1, nLab), st, false)) {
    delete t;
    return fail(FLTX_ERR_OOM, "trie: device allocation failed");
  }
  if (devCopyH2D(t->edge.p, edge.data(), sizeof(TrieEdge) * edge.size(), st) ||
      (nLab && devCopyH2D(t->labels.p, labels, sizeof(int32_t) * nLab, st)) || devSync(st)) {
    delete t;
    return fail(FLTX_ERR_HIP, "trie: upload failed");
  }
  *out = t;
  return FLTX_OK;
}

int fltx_trie_destroy(fltx_trie* t) {
  delete t;
  return FLTX_OK;
}

/* upload the flat n-gram tables to the context's device (once) */
static int lmEnsureUploaded(fltx_lm* lm, fltx_ctx* ctx, fltx_lm::Dev** out) {
  *out = nullptr;
  if (lm->kind != 1) {
    return FLTX_OK; /* (ZeroLM: nothing to compute; host LM: answered on the host) */
  }
  std::lock_guard<std::mutex> lock(lm->devMu);
  auto it = lm->dev.find(ctx->uid);
  if (it != lm->dev.end()) {
    *out = it->second.get();
    return FLTX_OK;
  }
  std::unique_ptr<fltx_lm::Dev> dv(new fltx_lm::Dev());
  Stream st = ctx->stream;
  const size_t cap = lm->hTab.size(), nn = lm->hBackoff.size();
  if (dv->tab.ensure(sizeof(NgramSlot) * cap, st, false) || dv->backoff.ensure(sizeof(float) * nn, st, false) ||
      dv->usrToLm.ensure(sizeof(int32_t) * std::max<size_t>(1, lm->hUsr.size()), st, false)) {
    return fail(FLTX_ERR_OOM, "n-gram tables: device allocation failed");
  }
  if (devCopyH2D(dv->tab.p, lm->hTab.data(), sizeof(NgramSlot) * cap, st) ||
      devCopyH2D(dv->backoff.p, lm->hBackoff.data(), sizeof(float) * nn, st) ||
      (!lm->hUsr.empty() && devCopyH2D(dv->usrToLm.p, lm->hUsr.data(), sizeof(int32_t) * lm->hUsr.size(), st)) ||
      devSync(st)) {
    return fail(FLTX_ERR_HIP, "n-gram tables: upload failed");
  }
  *out = dv.get();
  lm->dev[ctx->uid] = std::move(dv);
  return FLTX_OK;
}

/* The dense (context, token) table of an n-gram LM over N tokens (fltx_lm::TokDense), built once per (LM, N) and
 * uploaded once per context.  *out = null (and FLTX_OK): the model has too many contexts for a dense table. */
constexpr int64_t kTokDenseMaxCtx = 1ll << 24;      /* contexts */
constexpr int64_t kTokDenseMaxBytes = 1ll << 35;    /* 32 GB of the device's 288 (a character LM: 240 B per context) */
static int lmTokDense(fltx_lm* lm, fltx_ctx* ctx, fltx_lm::Dev* dv, int N, const int2** out, int64_t* nCtxOut) {
  *out = nullptr;
  if (lm->kind != 1 || !dv || N <= 0) {
    return FLTX_OK;
  }
  std::lock_guard<std::mutex> lock(lm->devMu);
  auto it = lm->tokDense.find(N);
  if (it == lm->tokDense.end()) {
    std::unique_ptr<fltx_lm::TokDense> td(new fltx_lm::TokDense());
    const int L = std::max(1, lm->order - 1);
    const int stride = N + 1;
    const int64_t capCtx = std::min<int64_t>(kTokDenseMaxCtx, kTokDenseMaxBytes / (8 * (int64_t)stride));
    /* contexts in the order they are first reached: row 0 = KenLM::start(false) (KenLM.cpp:52-61).  Rows live in one
     * flat array, their ids in an open-addressing table keyed by the L node ids; the frontier is scored a block of
     * rows at a time by the host's threads (the LM tables are read-only), the new contexts are numbered by one thread
     * in row order afterwards -- so the numbering does not depend on the number of threads.  (A 5-gram over 29 tokens
     * with 390 k contexts: 8.1 s with a std::map and one thread before; the caps below are what this has to reach.) */
    std::vector<int32_t> rows((size_t)L, 0); /* row r = rows[r * L .. r * L + L) */
    {
      uint32_t nd;
      float pr;
      if (lm->order > 1 && lmFind(lm, 0, (uint32_t)lm->bos, nd, pr)) {
        rows[0] = (int32_t)(nd & ~kPhantomNode);
      }
    }
    int64_t nRows = 1;
    std::vector<int32_t> slots((size_t)1 << 16, -1);
    auto hashOf = [&](const int32_t* c) {
      uint64_t h = 0x9E3779B97F4A7C15ull;
      for (int k = 0; k < L; ++k) {
        h = (h ^ (uint64_t)(uint32_t)c[k]) * 0xBF58476D1CE4E5B9ull;
        h ^= h >> 29;
      }
      return h;
    };
    auto place = [&](int32_t id) { /* (table with room: the caller grows it) */
      const uint64_t mask = slots.size() - 1;
      uint64_t h = hashOf(&rows[(size_t)id * L]) & mask;
      while (slots[h] >= 0) {
        h = (h + 1) & mask;
      }
      slots[h] = id;
    };
    place(0);
    auto findOrAdd = [&](const int32_t* c, bool& full) -> int32_t {
      const uint64_t mask = slots.size() - 1;
      uint64_t h = hashOf(c) & mask;
      while (slots[h] >= 0) {
        if (memcmp(&rows[(size_t)slots[h] * L], c, sizeof(int32_t) * (size_t)L) == 0) {
          return slots[h];
        }
        h = (h + 1) & mask;
      }
      if (nRows >= capCtx) {
        full = true;
        return 0;
      }
      const int32_t id = (int32_t)nRows++;
      rows.insert(rows.end(), c, c + L);
      slots[h] = id;
      if ((uint64_t)nRows * 2 > slots.size()) {
        slots.assign(slots.size() * 4, -1);
        for (int32_t r = 0; r < (int32_t)nRows; ++r) {
          place(r);
        }
      }
      return id;
    };
    bool tooMany = false;
    const int nThr = (int)std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
    constexpr int64_t kBlock = 1 << 14; /* rows scored between two numbering passes */
    std::vector<int32_t> nxt((size_t)kBlock * (size_t)N * (size_t)L);
    for (int64_t r0 = 0, r1 = 0; r0 < nRows && !tooMany; r0 = r1) {
      r1 = std::min(nRows, r0 + kBlock);
      td->tab.resize((size_t)r1 * (size_t)stride);
      auto scoreRows = [&](int64_t a, int64_t b2) {
        for (int64_t r = a; r < b2; ++r) {
          const int32_t* cur = &rows[(size_t)r * L];
          for (int n = 0; n <= N; ++n) {
            /* KenLM::score: usrToLmIdxMap_, an index beyond the dictionary goes to <unk> as on the device (lmScoreDev);
             * column N: KenLM::finish = the score of </s> */
            const uint32_t word = n == N ? (uint32_t)lm->eos : (n < lm->nUsr ? (uint32_t)lm->hUsr[(size_t)n] : (uint32_t)lm->unk);
            int32_t tmp[kMaxNgramOrder];
            const float sc = lmStepWord(lm, cur, word, tmp);
            if (n < N) {
              memcpy(&nxt[((size_t)(r - r0) * (size_t)N + (size_t)n) * (size_t)L], tmp, sizeof(int32_t) * (size_t)L);
            }
            uint32_t bits;
            memcpy(&bits, &sc, 4);
            td->tab[(size_t)r * (size_t)stride + (size_t)n] = make_int2((int)bits, 0);
          }
        }
      };
      const int64_t per = (r1 - r0 + nThr - 1) / nThr;
      if (nThr == 1 || r1 - r0 < 256) {
        scoreRows(r0, r1);
      } else {
        std::vector<std::thread> pool;
        for (int64_t a = r0; a < r1; a += per) {
          pool.emplace_back(scoreRows, a, std::min(r1, a + per));
        }
        for (auto& th : pool) {
          th.join();
        }
      }
      for (int64_t r = r0; r < r1 && !tooMany; ++r) { /* (rows may move: `rows` grows) */
        for (int n = 0; n < N; ++n) {
          const int32_t to = findOrAdd(&nxt[((size_t)(r - r0) * (size_t)N + (size_t)n) * (size_t)L], tooMany);
          if (tooMany) {
            break;
          }
          td->tab[(size_t)r * (size_t)stride + (size_t)n].y = to;
        }
      }
    }
    if (tooMany) {
      td->nCtx = -1;
      td->tab.clear();
      td->tab.shrink_to_fit();
    } else {
      td->nCtx = nRows;
    }
    it = lm->tokDense.emplace(N, std::move(td)).first;
  }
  const fltx_lm::TokDense& td = *it->second;
  if (td.nCtx <= 0) {
    return FLTX_OK;
  }
  auto dit = dv->tokDense.find(N);
  if (dit == dv->tokDense.end()) {
    std::unique_ptr<DBuf> buf(new DBuf());
    Stream st = ctx->stream;
    if (buf->ensure(sizeof(int2) * td.tab.size(), st, false)) {
      return fail(FLTX_ERR_OOM, "dense token-LM table: device allocation failed");
    }
    if (devCopyH2D(buf->p, td.tab.data(), sizeof(int2) * td.tab.size(), st) || devSync(st)) {
      return fail(FLTX_ERR_HIP, "dense token-LM table: upload failed");
    }
    dit = dv->tokDense.emplace(N, std::move(buf)).first;
  }
  *out = dit->second->as<int2>();
  if (nCtxOut) {
    *nCtxOut = td.nCtx;
  }
  return FLTX_OK;
}

/* ---- decoder ------------------------------------------------------------- */
int fltx_decoder_create(fltx_ctx* ctx, int32_t kind, const fltx_options* opt, const fltx_trie* trie,
                        const fltx_lm* lm, int32_t sil, int32_t blank, int32_t unk,
                        const float* transitions, int32_t nTrans, int32_t isLmToken,
                        fltx_decoder** out) {
  if (!ctx || !opt || !lm || !out) {
    return fail(FLTX_ERR_INVALID, "fltx_decoder_create: null argument");
  }
  if (kind != FLTX_DECODER_LEXFREE && kind != FLTX_DECODER_LEXICON) {
    return fail(FLTX_ERR_INVALID, "unknown decoder kind %d", kind);
  }
  if (kind == FLTX_DECODER_LEXICON && !trie) {
    return fail(FLTX_ERR_INVALID, "lexicon decoder needs a trie");
  }
  if (opt->beam_size < 1 || opt->beam_size_token < 1) {
    return fail(FLTX_ERR_INVALID, "beam_size and beam_size_token must be >= 1");
  }
  if (opt->beam_size > 0x7FFFFF) {
    return fail(FLTX_ERR_UNSUPPORTED, "beam_size too large");
  }
  if (opt->criterion != FLTX_CRITERION_ASG && opt->criterion != FLTX_CRITERION_CTC) {
    return fail(FLTX_ERR_UNSUPPORTED, "criterion %d not supported (ASG, CTC only)", opt->criterion);
  }
  if (trie && trie->ctx != ctx) {
    return fail(FLTX_ERR_INVALID, "fltx_decoder_create: the trie was uploaded to another context (device)");
  }
  DeviceScope devScope(ctx);
  if (devScope.failed) {
    return fail(FLTX_ERR_HIP, "hipSetDevice failed");
  }
  fltx_lm::Dev* lmDev = nullptr;
  {
    int rcu = lmEnsureUploaded(const_cast<fltx_lm*>(lm), ctx, &lmDev);
    if (rcu) {
      return rcu;
    }
  }
  auto* d = new fltx_decoder();
  d->lmDev = lmDev;
  d->ctx = ctx;
  d->kind = kind;
  d->opt = *opt;
  d->trie = trie;
  d->lm = lm;
  d->sil = sil;
  d->blank = blank;
  d->unk = unk;
  d->isLmToken = isLmToken ? 1 : 0;
  if (transitions && nTrans > 0) {
    d->nTrans = nTrans;
    d->transMax = 0.0;
    for (int i = 0; i < nTrans; ++i) {
      d->transMax = std::max(d->transMax, (double)transitions[i]);
    }
    if (d->transitions.ensure(sizeof(float) * (size_t)nTrans, ctx->stream, false) ||
        devCopyH2D(d->transitions.p, transitions, sizeof(float) * (size_t)nTrans, ctx->stream) ||
        devSync(ctx->stream)) {
      delete d;
      return fail(FLTX_ERR_HIP, "transitions upload failed");
    }
  }
  *out = d;
  return FLTX_OK;
}

int fltx_decoder_destroy(fltx_decoder* d) {
  DeviceScope devScope(d ? d->ctx : nullptr);
  if (devScope.failed) {
    return fail(FLTX_ERR_HIP, "hipSetDevice failed");
  }
  if (d) {
    devSync(d->ctx->stream);
#ifndef FLTX_EMU
    for (int i = 0; i < 3; ++i) {
      if (d->ev[i]) {
        (void)hipEventDestroy(d->ev[i]);
      }
    }
    if (d->copyStream) {
      (void)hipStreamSynchronize(d->copyStream);
      (void)hipEventDestroy(d->evSlotFree[0]);
      (void)hipEventDestroy(d->evSlotFree[1]);
      (void)hipStreamDestroy(d->copyStream);
    }
#endif
  }
  delete d;
  return FLTX_OK;
}

static int settlePendingLook(fltx_decoder* d); /* (defined after syncResults) */
int fltx_decoder_get(fltx_decoder* d, const char* key, int64_t* value) {
  if (!d || !key || !value) {
    return fail(FLTX_ERR_INVALID, "fltx_decoder_get: null argument");
  }
  if (!strcmp(key, "engine")) { /* the engine the last call started on ("redone" counts what it handed on) */
    *value = d->engineFirst;
  } else if (!strcmp(key, "xlane")) {
    *value = d->xlaneFirst;
  } else if (!strcmp(key, "ylane")) {
    *value = d->ylaneFirst;
  } else if (!strcmp(key, "yshare")) {
    *value = (d->ylane || d->xlane) ? d->yshare : 0;
  } else if (!strcmp(key, "redone")) {
    if (d->offlinePending) { /* ("defer_check": the look has not happened yet) */
      DeviceScope devScope(d->ctx);
      int rc = settlePendingLook(d);
      if (rc) {
        return rc;
      }
    }
    *value = d->lastRedo;
  } else if (!strcmp(key, "sstream")) {
    *value = d->sstream;
  } else if (!strcmp(key, "tstream")) { /* 1: the stream's chunks run on the token-LM variant of fltx_slane.h */
    *value = d->tstream;
  } else if (!strcmp(key, "looks_dropped")) { /* defer_check = 2: batches that were overwritten without a look at their statuses */
    *value = d->looksDropped;
  } else if (!strcmp(key, "unread_redone")) { /* defer_check = 1: utterances decoded again while settling batches nobody read */
    *value = d->unreadRedone;
  } else if (!strcmp(key, "compactions")) { /* streams: times the LM-state free lists were rebuilt since fltx_stream_begin */
    *value = d->compactions;
  } else if (!strcmp(key, "staged_emissions")) { /* address of the library's own device copy of the last offline batch's emissions
                                                     (0 when the caller's device buffer was used): valid until the next call */
    *value = (d->haveResults && d->lastEmis && (d->lastEmis == d->emis[0].as<float>() || d->lastEmis == d->emis[1].as<float>()))
                 ? (int64_t)(uintptr_t)d->lastEmis : 0;
  } else if (!strcmp(key, "id_cap")) {     /* streams: LM-state ids per stream (constant however long the stream runs) */
    *value = d->recycle ? d->idCap : 0;
  } else if (!strcmp(key, "hlm_asked")) { /* host LM: questions the frames listed since decodeBegin ... */
    *value = d->hlmAsked;
  } else if (!strcmp(key, "hlm_distinct")) { /* ... and how many the callbacks were asked (distinct per frame) */
    *value = d->hlmDistinct;
  } else if (!strcmp(key, "stream_redone")) {
    *value = d->streamRedone;
  } else if (!strcmp(key, "wlane")) { /* 1: the last call ran on fltx_wlane.h (token sets beyond 64) */
    *value = d->wlaneFirst;
  } else if (!strcmp(key, "slane")) {
    *value = d->slaneFirst;
  } else if (!strcmp(key, "tlane")) { /* 1: the last call started on fltx_slane.h's token-LM variant */
    *value = d->tlaneFirst;
  } else if (!strcmp(key, "toklm_contexts")) { /* rows of the LM's dense (context, token) table (0: none built) */
    *value = d->tokLmCtx;
  } else if (!strcmp(key, "fallback_reasons")) {
    *value = d->fallbackReasons;
  } else if (!strcmp(key, "why_not_lane")) { /* FLTX_WHY_* (include/fltx.h): why the last call did not start on a lane engine */
    *value = d->whyFirst;
  } else if (!strcmp(key, "lane_groups")) { /* lane groups of the lane = LM state engine: 1 = fltx_slane.h, 2 / 4 / 8 = fltx_mlane.h;
                                                fltx_ylane.h: 1 / 2 / 4 */
    *value = d->laneGroupsFirst;
  } else if (!strcmp(key, "ymemo_slots")) {
    *value = (d->ylane || d->xlane) && d->yshare ? (int64_t)d->ymemoSlots : 0;
  } else if (!strcmp(key, "lane")) {
    *value = d->lane;
  } else if (!strcmp(key, "lean")) {
    *value = d->lean; /* groups per thread kept in registers: 6 / 12, or 255 = streaming */
  } else if (!strcmp(key, "threads")) {
    *value = d->threads;
  } else if (!strcmp(key, "lds")) {
    *value = d->wsInLds ? 1 : 0;
  } else if (!strcmp(key, "cut")) {
    *value = d->cutM;
  } else if (!strcmp(key, "hot_level")) {
    *value = d->wsInLds ? 0 : d->hotLevel;
  } else if (!strcmp(key, "recompute")) {
    *value = d->cutRecompute;
  } else if (!strcmp(key, "cap")) {
    *value = d->CAP;
  } else if (!strcmp(key, "cap2")) {
    *value = d->CAP2;
  } else if (!strcmp(key, "items")) {
    *value = d->itemCap;
  } else {
    return fail(FLTX_ERR_INVALID, "fltx_decoder_get: unknown key '%s'", key);
  }
  return FLTX_OK;
}

int fltx_decoder_set(fltx_decoder* d, const char* key, int64_t value) {
  if (!d || !key) {
    return fail(FLTX_ERR_INVALID, "null argument");
  }
  if (!strcmp(key, "defer_check")) {
    d->deferCheck = value == 2 ? 2 : (value ? 1 : 0);
    return FLTX_OK;
  }
  if (!strcmp(key, "compact_always")) {
    d->userCompactAlways = value ? 1 : 0;
    return FLTX_OK;
  }
  if (!strcmp(key, "threads")) {
    if (value != 64 && value != 128 && value != 256 && value != 512 && value != 1024) {
      return fail(FLTX_ERR_INVALID, "threads must be 64, 128, 256, 512 or 1024");
    }
    d->userThreads = (int)value;
    d->threads = (int)value;
    return FLTX_OK;
  }
  if (!strcmp(key, "keep_scores")) { /* record {score, am, lm} per history slot (getBestHypothesis) */
    d->keepScores = d->userKeepScores = value != 0;
    return FLTX_OK;
  }
  if (!strcmp(key, "profile_wave")) {
    d->profWave = (int)value;
    return FLTX_OK;
  }
  if (!strcmp(key, "profile")) {
    d->profile = value != 0;
    return FLTX_OK;
  }
  if (!strcmp(key, "lean")) { /* 0: lexicon-free + ZeroLM frames use the generic engine */
    d->noLean = value == 0;
    return FLTX_OK;
  }
  if (!strcmp(key, "items")) { /* 0: the lexicon decoder walks the full hypothesis x token grid */
    d->genericAsked = true; /* a knob of the generic lexicon engine: that engine is what the caller wants */
    d->noItems = value ? 0 : 1;
    return FLTX_OK;
  }
  if (!strcmp(key, "slim")) { /* 0: the cut-off generation recomputes instead of keeping slim records */
    d->genericAsked = true; /* a knob of the generic lexicon engine: that engine is what the caller wants */
    d->noSlim = value ? 0 : 1;
    return FLTX_OK;
  }
  if (!strcmp(key, "hot_level")) { /* testing: 1 keeps the candidate records of an HBM workspace in HBM */
    d->genericAsked = true; /* a knob of the generic lexicon engine: that engine is what the caller wants */
    d->maxHotLevel = (int)value;
    return FLTX_OK;
  }
  if (!strcmp(key, "lds_budget")) { /* testing: pretend the CU has this many bytes of LDS (<= 160 KiB) */
    d->genericAsked = true; /* a knob of the generic lexicon engine: that engine is what the caller wants */
    d->ldsBudget = value > 0 && (size_t)value < kMaxLds ? (size_t)value : 0;
    return FLTX_OK;
  }
  if (!strcmp(key, "cut_m")) { /* testing: number of best candidates kept by the cut (default 3K + 64) */
    d->genericAsked = true; /* a knob of the generic lexicon engine: that engine is what the caller wants */
    d->userCutM = (int)value;
    return FLTX_OK;
  }
  if (!strcmp(key, "cut")) { /* 0: the lexicon decoder materialises every candidate (no score-pass cut) */
    d->genericAsked = true; /* a knob of the generic lexicon engine: that engine is what the caller wants */
    d->noCut = value ? 0 : 1;
    return FLTX_OK;
  }
  if (!strcmp(key, "xlane")) { /* 0: do not use the lane = (LM state, trie node) kernel (fltx_xlane.h) */
    d->noXlane = value ? 0 : 1;
    return FLTX_OK;
  }
  if (!strcmp(key, "stream_total_frames")) { /* before fltx_stream_begin: see prepare() */
    d->streamTotalFrames = (int)value;
    return FLTX_OK;
  }
  if (!strcmp(key, "sstream")) { /* 0: stream chunks stay on the lane-per-slot step (fltx_lane.h) */
    d->noSstream = value ? 0 : 1;
    return FLTX_OK;
  }
  if (!strcmp(key, "stream_optimistic")) { /* 0: streams of the lexicon decoder start on the worst-case (HBM) workspace */
    d->userStreamOpt = (int)value;
    return FLTX_OK;
  }
  if (!strcmp(key, "lm_cache")) { /* 0: the generic step asks the n-gram tables every time (DecodeParams::lmCache) */
    d->noLmCache = value ? 0 : 1;
    return FLTX_OK;
  }
  if (!strcmp(key, "stream_defer")) { /* 0: fltx_stream_step waits for its chunk and decodes it again there if it must */
    d->deferRedo = value ? 1 : 0;
    return FLTX_OK;
  }
  if (!strcmp(key, "bt_lds_kb")) { /* LDS the back-trace kernel stages history chunks in (0: 144 KB) */
    d->btLdsKb = (int)value;
    return FLTX_OK;
  }
  if (!strcmp(key, "yshare")) { /* fltx_ylane.h geometry that shares a CU: 1 always, 0 never, -1 when the batch exceeds the CUs */
    d->userYshare = (int)value;
    return FLTX_OK;
  }
  if (!strcmp(key, "ylane")) { /* 0: do not use fltx_ylane.h; 2: prefer it where fltx_xlane.h applies too */
    d->noYlane = value ? 0 : 1;
    d->preferYlane = value == 2;
    return FLTX_OK;
  }
  if (!strcmp(key, "slane_threads")) {
    d->slaneThreads = (int)value;
    return FLTX_OK;
  }
  if (!strcmp(key, "wlane")) { /* 0: large token sets stay off the lane = LM state engine (fltx_wlane.h) */
    d->noWlane = value ? 0 : 1;
    return FLTX_OK;
  }
  if (!strcmp(key, "tok_dense")) { /* 0: a token-level n-gram LM is not flattened to a dense table (tests: the probe chain) */
    d->noTokDense = value ? 0 : 1;
    return FLTX_OK;
  }
  if (!strcmp(key, "tlane")) { /* 0: a token-level n-gram LM on the lexicon-free decoder stays on the generic engine */
    d->noTlane = value ? 0 : 1;
    return FLTX_OK;
  }
  if (!strcmp(key, "slane")) { /* 0: do not use the lane = LM state kernel (fltx_slane.h) */
    d->noSlane = value ? 0 : 1;
    return FLTX_OK;
  }
  if (!strcmp(key, "ylane_asg")) { /* 0: the ASG criterion keeps the lexicon decoder on the generic engine */
    d->noYlaneAsg = value ? 0 : 1;
    return FLTX_OK;
  }
  if (!strcmp(key, "tune")) { /* development: see DecodeParams::tune */
    d->userTune = (int)value;
    return FLTX_OK;
  }
  if (!strcmp(key, "ylane_rank_at")) { /* tests: see DecodeParams::yRankAt */
    d->userYRankAt = (int)value;
    return FLTX_OK;
  }
  if (!strcmp(key, "ylane_groups")) { /* fltx_ylane.h: at least this many lane groups (tests) */
    d->userYlaneGroups = (int)value;
    return FLTX_OK;
  }
  if (!strcmp(key, "lane_groups")) { /* fltx_mlane.h: 0 = as many lane groups as the beam needs, 2 / 4 / 8 = at least that many, -1 = never */
    d->userLaneGroups = (int)value;
    return FLTX_OK;
  }
  if (!strcmp(key, "mlane_geo")) { /* tuning: row of kMlaneGeo to use (-1 = first that fits) */
    d->userMlaneGeo = (int)value;
    return FLTX_OK;
  }
  if (!strcmp(key, "lane")) { /* 0: beams <= 64 use the lean kernel instead of the lane-per-slot kernel */
    d->noLane = value ? 0 : 1;
    return FLTX_OK;
  }
  if (!strcmp(key, "dense")) { /* 0: force the generic hash merge for lexicon-free frames */
    d->noDense = value == 0;
    return FLTX_OK;
  }
  if (!strcmp(key, "force_global_ws")) {
    d->genericAsked = true; /* a knob of the generic lexicon engine: that engine is what the caller wants */
    d->forceGlobalWs = value != 0;
    return FLTX_OK;
  }
  return fail(FLTX_ERR_INVALID, "unknown tunable '%s'", key);
}

} /* extern "C" */

namespace {

/* geometry + buffers for B streams of up to maxFrames frames (plus seed and
 * decodeEnd slots) */
/* fltx_mlane.h geometries that are compiled (fltx_instances.h, FLTX_MLANE_SET); "mlane_geo" = row */
struct MlaneGeo {
  int threads, gt, ng, gpw, spw;
};
static const MlaneGeo kMlaneGeo[] = {{640, 4, 2, 2, 1}, {960, 5, 2, 1, 1}, {640, 10, 2, 2, 1}, {768, 4, 4, 4, 1},
                                     {960, 5, 4, 2, 2}, {960, 11, 4, 2, 2}, {960, 10, 8, 2, 4}};
constexpr int kMlaneGeoCount = (int)(sizeof(kMlaneGeo) / sizeof(kMlaneGeo[0]));

int engineOf(const fltx_decoder* d) {
  return d->ylane ? 6 : d->xlane ? 5 : (d->slane ? 4 : (d->lane ? 3 : (d->lean ? 2 : (d->dense ? 1 : 0))));
}
/* what fltx_decoder_get reports about the engine a call started on */
void latchFirst(fltx_decoder* d) {
  d->engineFirst = engineOf(d);
  d->whyFirst = d->whyNotLane;
  d->wlaneFirst = d->wlane;
  d->slaneFirst = d->slane;
  d->tlaneFirst = d->tlane;
  d->xlaneFirst = d->xlane;
  d->ylaneFirst = d->ylane;
  d->laneGroupsFirst = d->slane ? std::max(1, d->mlaneNG) : d->ylane;
}

int prepare(fltx_decoder* d, int B, int N, const int32_t* Tmax, bool forceWorstCaseCap) {
  /* host LM: the generic engine, a frame per launch, every candidate's record built in one pass (the cut-off
   * generation rebuilds LM-state keys from (state, label) pairs, which a host LM's states are not) */
  const bool hostLm = d->lm->kind == 2;
  forceWorstCaseCap = forceWorstCaseCap || hostLm;
  const size_t kMaxLdsHw = kMaxLds;
  const size_t kMaxLds = d->ldsBudget ? d->ldsBudget : kMaxLdsHw;
  Stream st = d->ctx->stream;
  const int K = d->opt.beam_size;
  if (d->kind == FLTX_DECODER_LEXICON && d->trie->nTokens != N) {
    return fail(FLTX_ERR_INVALID, "N = %d but the trie was built for %d tokens", N, d->trie->nTokens);
  }
  if (d->nTrans && d->nTrans != N * N) {
    return fail(FLTX_ERR_INVALID, "transitions has %d entries, expected N*N = %d", d->nTrans, N * N);
  }
  if (d->opt.criterion == FLTX_CRITERION_ASG && !d->nTrans) {
    return fail(FLTX_ERR_INVALID, "ASG criterion needs N*N transitions");
  }
  if (d->sil < 0 || d->sil >= N) {
    return fail(FLTX_ERR_RANGE, "sil index %d outside [0, %d)", d->sil, N);
  }
  if (d->opt.criterion == FLTX_CRITERION_CTC && (d->blank < 0 || d->blank >= N)) {
    return fail(FLTX_ERR_RANGE, "blank index %d outside [0, %d)", d->blank, N);
  }
  d->B = B;
  d->N = N;
  d->histOff.resize(B + 1);
  int64_t off = 0;
  int maxT = 0;
  for (int b = 0; b < B; ++b) {
    if (Tmax[b] < 0) {
      return fail(FLTX_ERR_INVALID, "T[%d] = %d is negative", b, Tmax[b]);
    }
    d->histOff[b] = off;
    off += (int64_t)(Tmax[b] + 2) * K;
    maxT = std::max(maxT, Tmax[b]);
  }
  d->histOff[B] = off;
  d->histRecords = off;
  /* LM states are created for as long as a stream runs, not for as long as its frames stay buffered: the id
   * tables of a stream are sized for `stream_total_frames` (default: at least 2 048 frames; a longer stream says
   * so before fltx_stream_begin, or its status reports a full table), the history for max_frames */
  /* A stream's ids are recycled (fltx_compact_states_kernel): its tables are sized for what the buffer can hold --
   * beam x max_frames states made between two rebuilds of the free list, plus the states a rebuild keeps -- however
   * long the stream runs ("stream_total_frames" is accepted and ignored).  A host LM numbers its own states. */
  d->recycle = !d->offlineCall && !hostLm;
  const int idT = maxT;
  const int64_t streamIds = (int64_t)K * (idT + 2) + 8 * (int64_t)K + 64;
  uint64_t wantStates = d->recycle ? 2ull * (uint64_t)streamIds : 2ull * ((uint64_t)K * (uint64_t)(idT + 2) + 2);
  uint32_t cap = nextPow2(std::max<uint64_t>(wantStates, 1024));
  if (d->recycle && cap > (1u << 24)) { /* (a stream's ids are bounded below: 2^23 of them fill half of this) */
    cap = 1u << 24;
  }
  if (cap > (1u << 23) && !d->recycle) {
    return fail(FLTX_ERR_UNSUPPORTED, "K * T = %llu exceeds the 2^23 LM states per utterance this build indexes",
                (unsigned long long)K * (maxT + 2));
  }
  /* candidate capacity */
  const int nTok = std::min(d->opt.beam_size_token, N);
  /* lexicon-free frames merge through the dense (hash-free) path: one slot per
   * (LM state, token) group plus one per orphan repeat; the hash is then only
   * used by decodeEnd (<= K candidates) */
  /* (not for a host LM: the dense merge assumes at most two hypotheses per LM state -- true of states that are a trie
   * over their inputs, not of whatever a user's LM::score returns) */
  d->dense = (d->kind == FLTX_DECODER_LEXFREE && !d->noDense && !hostLm) ? 1 : 0;
  /* threads per utterance: the frame step is latency bound, so more waves per
   * utterance win as long as the batch does not fill the CUs on its own
   * (measured on C2: 256 -> 15.3 ms, 512 -> 13.0 ms per 256-utterance batch) */
  if (!d->userThreads) {
    /* one workgroup per CU: give it two waves per SIMD; more utterances than
     * CUs: 256 threads, so that several utterances share a CU (C2 at B=512:
     * 512 threads cannot co-reside (VGPRs), 256 can) */
    d->threads = (d->kind == FLTX_DECODER_LEXICON || B <= d->ctx->numCUs) ? 512 : 256;
    /* (the lexicon decoder's workspace fills a CU's LDS, so one utterance per CU either way:
     * 1024 utterances, 256 -> 512 threads: C3 19.7 -> 27.9 M frames/s, C4 9.8 -> 12.7 M) */
  }
  /* lean frame step (fltx_lean.h): lexicon-free + ZeroLM, groups held in registers */
  d->lean = 0;
  bool leanInHbm = false;
  if (d->dense && !d->noLean && !d->forceGlobalWs && d->lm->kind == 0 && K < 32000 && N <= 64) {
    const int64_t groups = (int64_t)K * (nTok + 1);
    const int64_t per = (groups + d->threads - 1) / d->threads;
    d->lean = per <= 6 ? 6 : (per <= 12 ? 12 : 255); /* 255: streaming lean step (groups evaluated twice) */
  }
  /* lane-per-slot frame step (fltx_lane.h): beam and token set fit one wave's lanes */
  d->lane = 0;
  if (d->lean && !d->noLane && K <= 64 && N <= 64) {
    const int nW = d->threads / 64;
    const int per = (nTok + nW - 1) / nW;
    d->lane = per <= 4 ? 4 : (per <= 8 ? 8 : 0);
  }
  /* lane = LM state decode (fltx_slane.h): offline lexicon-free + ZeroLM max-merge, beam and
   * tokens within one wave's lanes; the history rows are its LM-state memo (23-bit ids) */
  d->slane = 0;
  if (d->lane && !d->noSlane && d->offlineCall && !d->keepScores && !forceWorstCaseCap &&
      d->opt.beam_threshold >= 0.0 && d->sil >= 0 && d->sil < N &&
      (d->opt.criterion != FLTX_CRITERION_CTC || (d->blank >= 0 && d->blank < N)) &&
      (int64_t)K * (maxT + 2) < (1 << 23) - 1) {
    /* (threads, list positions per token wave) pairs that are compiled (fltx_instances.h); two of the
     * waves do not evaluate tokens (own groups of the lanes / row staging and housekeeping) */
    static const int geoOne[][2] = {{576, 4}, {512, 5}, {448, 6}, {384, 7}, {320, 10}, {640, 4}, {512, 12}, {576, 10}}; /* fastest first (C2: 1.95, 2.11, 2.14 ms ...) */
    /* more utterances than CUs: 512 threads (104 VGPRs: four waves per SIMD) let two utterances share a CU --
     * C2 at 512 utterances: 162.7 M frames/s against 119.3 M with nine waves each */
    static const int geoTwo[][2] = {{512, 5}, {448, 6}, {384, 7}, {576, 4}, {320, 10}, {640, 4}, {512, 12}, {576, 10}};
    const auto& geo = B > d->ctx->numCUs ? geoTwo : geoOne;
    const int nList = nTok - ((d->opt.criterion == FLTX_CRITERION_CTC && d->opt.beam_size_token >= N) ? 1 : 0);
    for (const auto& g : geo) {
      if ((d->userThreads && d->threads != g[0]) || (d->slaneThreads && d->slaneThreads != g[0])) {
        continue;
      }
      if (nList <= g[1] * (g[0] / 64 - 2)) {
        d->slane = g[1];
        d->threads = g[0];
        break;
      }
    }
  }
  /* ... with a token-level n-gram LM (slaneUtterance<.., TL = true>; LexiconFreeDecoder.cpp:69-85 + KenLM.cpp:63-83):
   * LM states are still the trie of token histories, so the lanes keep their shape; the LM is one gather from a dense
   * (context, token) table built on the host (lmTokDense) -- when the model's contexts fit one */
  d->tlane = 0;
  d->tokLm = nullptr;
  d->tokLmTooBig = false;
  const int2* tab = nullptr;
  int64_t nCtx = 0;
  const bool tokenLm = d->kind == FLTX_DECODER_LEXFREE || (d->kind == FLTX_DECODER_LEXICON && d->isLmToken);
  if (tokenLm && d->lm->kind == 1 && N <= 64 && !d->noTokDense) { /* (every index the LM is asked about is a token) */
    /* the dense table serves every engine this decoder can run on: the lane engine below gathers from it, and the
     * generic engine (beams beyond 64, streams, fallbacks) keeps a state's context as a row number and replaces its
     * chain of n-gram probes by the same gather (lmScoreDev) */
    int rcd = lmTokDense(const_cast<fltx_lm*>(d->lm), d->ctx, d->lmDev, N, &tab, &nCtx);
    if (rcd) {
      return rcd;
    }
    d->tokLmTooBig = tab == nullptr;
    d->tokLm = tab;
    d->tokLmCtx = tab ? nCtx : 0;
  }
  if (tab && d->kind == FLTX_DECODER_LEXFREE && !d->noSlane && !d->noTlane && !d->genericAsked &&
      !d->noDense && d->offlineCall && !d->keepScores && !forceWorstCaseCap && !d->forceGlobalWs && K <= 64 &&
      d->opt.beam_threshold >= 0.0 && d->sil >= 0 && d->sil < N &&
      (d->opt.criterion != FLTX_CRITERION_CTC || (d->blank >= 0 && d->blank < N)) &&
      (int64_t)K * (maxT + 2) < (1 << 23) - 1) {
    {
      static const int geoOne[][2] = {{576, 4}, {512, 5}, {448, 6}, {384, 7}, {320, 10}, {640, 4}, {512, 12}, {576, 10}};
      static const int geoTwo[][2] = {{512, 5}, {448, 6}, {384, 7}, {576, 4}, {320, 10}, {640, 4}, {512, 12}, {576, 10}};
      const auto& geo = B > d->ctx->numCUs ? geoTwo : geoOne;
      const int nList = nTok - ((d->opt.criterion == FLTX_CRITERION_CTC && d->opt.beam_size_token >= N) ? 1 : 0);
      for (const auto& g : geo) {
        if ((d->userThreads && d->threads != g[0]) || (d->slaneThreads && d->slaneThreads != g[0])) {
          continue;
        }
        if (nList <= g[1] * (g[0] / 64 - 2)) {
          d->slane = g[1];
          d->tlane = 1;
          /* the re-entry memos in LDS: 4 096 edges + 2 048 child masks (75 KB: two workgroups fit a CU) when the batch
           * has more utterances than CUs, twice as many of both (131 KB) when a workgroup has the CU to itself */
          const bool shareCu = B > d->ctx->numCUs || (d->deferCheck && g[0] == 512); /* (defer_check: the caller keeps two batches in flight) */
          d->tlEdgeSlots = shareCu ? 4096 : 8192;
          d->tlMaskSlots = shareCu ? 2048 : 4096;
          d->threads = g[0];
          break;
        }
      }
    }
  }
  /* ... for token sets beyond 64 (word pieces) when the token beam keeps at most 64 of them (fltx_wlane.h): masks and
   * lists by position in the frame's token beam, a front-end wave that cuts the row to it */
  d->wlane = 0;
  if (d->kind == FLTX_DECODER_LEXFREE && !d->noSlane && !d->noWlane && !d->genericAsked && d->offlineCall && !d->keepScores &&
      !forceWorstCaseCap && !d->forceGlobalWs && !d->opt.log_add && d->lm->kind == 0 && !d->isLmToken && N > 64 &&
      N <= kWlMaxN && K <= 64 && nTok <= 64 && d->opt.beam_threshold >= 0.0 && d->sil >= 0 && d->sil < N &&
      (d->opt.criterion != FLTX_CRITERION_CTC || (d->blank >= 0 && d->blank < N)) &&
      (int64_t)K * (maxT + 2) < (1 << 23) - 1) {
    static const int geoW[][2] = {{576, 5}, {576, 8}, {576, 10}};
    for (const auto& g : geoW) {
      if (d->userThreads && d->threads != g[0]) {
        continue;
      }
      if (nTok <= g[1] * (g[0] / 64 - 2)) {
        d->slane = g[1];
        d->wlane = 1;
        d->wlMaxT = maxT;
        d->threads = g[0];
        break;
      }
    }
  }
  /* ... with several groups of 64 lanes for beams beyond one wave's lanes (fltx_mlane.h): same candidates, merges and
   * selection; a wave holds the states of whole lane groups, history records carry 10-bit slots */
  d->mlaneNG = 0;
  if ((!d->slane || d->userLaneGroups > 1) && d->lean && !d->noSlane && d->userLaneGroups >= 0 && d->offlineCall && !d->keepScores &&
      !forceWorstCaseCap && K <= 64 * kMlMaxGroups && (K > 64 || d->userLaneGroups > 1) && N <= 64 &&
      d->opt.beam_threshold >= 0.0 && d->sil >= 0 && d->sil < N &&
      (d->opt.criterion != FLTX_CRITERION_CTC || (d->blank >= 0 && d->blank < N)) &&
      (int64_t)K * (maxT + 2) < (1ll << 31) - 1) {
    const int nList = nTok - ((d->opt.criterion == FLTX_CRITERION_CTC && d->opt.beam_size_token >= N) ? 1 : 0);
    const int needNG = std::max((K + 63) / 64, d->userLaneGroups);
    /* fastest first per group count (C2 shape, profiles/r04: beam 100 3.17 ms on row 1 against 3.23 on row 0, beam 200
     * 5.79 on row 4 against 6.21 on row 3; the wide rows 2 / 5 spill registers and are for token lists beyond 30) */
    static const int order[kMlaneGeoCount] = {1, 0, 2, 4, 3, 5, 6};
    for (int oi = 0; oi < kMlaneGeoCount; ++oi) {
      const int gi = order[oi];
      const MlaneGeo& g = kMlaneGeo[gi];
      if (d->userMlaneGeo >= 0 ? gi != d->userMlaneGeo : (d->userThreads && d->threads != g.threads)) {
        continue;
      }
      const int nBlk = (g.threads / 64 - g.ng / g.spw - 1) / (g.ng / g.gpw);
      if (g.ng >= needNG && nList <= g.gt * nBlk) {
        d->slane = g.gt;
        d->mlaneNG = g.ng;
        d->mlaneGPW = g.gpw;
        d->mlaneSPW = g.spw;
        d->threads = g.threads;
        break;
      }
    }
  }
  /* ... and both: a token-level n-gram LM at beams beyond 64 (mlaneUtterance<.., TL = true>): the dense table's gather per
   * candidate, state ids from a table in HBM (ymemo: a slot per state an utterance can create); token lists
   * of up to 64 at beams up to 256, of up to 30 beyond (the five geometries compiled for it) */
  if (tab && !d->slane && d->kind == FLTX_DECODER_LEXFREE && !d->noSlane && !d->noTlane && !d->genericAsked && !d->noDense &&
      d->userLaneGroups >= 0 && d->offlineCall && !d->keepScores && !forceWorstCaseCap && !d->forceGlobalWs &&
      K > 64 && K <= 64 * kMlMaxGroups && d->opt.beam_threshold >= 0.0 && d->sil >= 0 && d->sil < N &&
      (d->opt.criterion != FLTX_CRITERION_CTC || (d->blank >= 0 && d->blank < N)) &&
      (int64_t)K * (maxT + 2) < (1ll << 27)) {
    const int nList = nTok - ((d->opt.criterion == FLTX_CRITERION_CTC && d->opt.beam_size_token >= N) ? 1 : 0);
    const int needNG = std::max((K + 63) / 64, d->userLaneGroups);
    static const MlaneGeo geoT[] = {{960, 5, 2, 1, 1}, {960, 11, 2, 1, 1}, {960, 5, 4, 2, 2}, {960, 11, 4, 2, 2}, {960, 10, 8, 2, 4}};
    for (const MlaneGeo& g : geoT) {
      const int nBlk = (g.threads / 64 - g.ng / g.spw - 1) / (g.ng / g.gpw);
      if (g.ng >= needNG && nList <= g.gt * nBlk && (!d->userThreads || d->threads == g.threads)) {
        uint32_t slots = 1024;
        while ((int64_t)slots < (int64_t)K * (maxT + 2) && slots < (1u << 27)) {
          slots <<= 1;
        }
        if ((int64_t)slots * 8 * (int64_t)B > (16ll << 30)) {
          break; /* (very long utterances at a large beam: 16 GB of id tables is where this engine stops) */
        }
        d->slane = g.gt;
        d->tlane = 1;
        d->mlaneNG = g.ng;
        d->mlaneGPW = g.gpw;
        d->mlaneSPW = g.spw;
        d->threads = g.threads;
        d->ymemoSlots = slots; /* every state an utterance can create has a slot: the table cannot fill */
        break;
      }
    }
  }
  /* ... and the frames of a stream's decodeStep chunks on the same engine (the parked beam, the (parent, token) ->
   * id tables and the history rows keep the lane-per-slot engine's format) */
  d->sstream = 0;
  if (d->lane && !d->noSlane && !d->noSstream && !d->offlineCall && d->keepScores && !d->opt.log_add &&
      d->opt.beam_threshold >= 0.0 && d->sil >= 0 && d->sil < N &&
      (d->opt.criterion != FLTX_CRITERION_CTC || (d->blank >= 0 && d->blank < N))) {
    static const int geoS[][2] = {{576, 4}, {512, 5}, {576, 10}};
    const int nList = nTok - ((d->opt.criterion == FLTX_CRITERION_CTC && d->opt.beam_size_token >= N) ? 1 : 0);
    for (const auto& g : geoS) {
      if (nList <= g[1] * (g[0] / 64 - 2)) {
        d->sstream = g[1];
        d->sstreamThreads = g[0];
        break;
      }
    }
  }
  /* lane = (LM state, trie node) decode (fltx_xlane.h): offline LexiconDecoder + ZeroLM over a lexicon
   * without LM scores, CTC, max-merge or logAdd (round 5), one word per spelling, every word ending in sil, no <unk> */
  d->xlane = 0;
  d->yshare = 0;
  if (d->kind == FLTX_DECODER_LEXICON && !d->noXlane && !d->genericAsked && d->offlineCall && !d->keepScores &&
      !forceWorstCaseCap && !d->forceGlobalWs && d->lm->kind == 0 && !d->isLmToken && d->trie && d->trie->xOk &&
      !d->trie->xMulti && d->trie->xZeroSmear && d->trie->xEndTok == d->sil && d->sil != d->blank &&
      d->opt.criterion == FLTX_CRITERION_CTC && !(d->opt.unk_score > -std::numeric_limits<double>::infinity()) &&
      K <= 64 && N <= 64 && d->opt.beam_threshold >= 0.0 && d->sil >= 0 && d->sil < N && d->blank >= 0 &&
      d->blank < N && (int64_t)K * (maxT + 2) < (1 << 23) - 1) {
    static const int geo[][2] = {{512, 2}, {512, 3}, {640, 2}, {576, 5}, {640, 10}};
    for (const auto& g : geo) {
      if ((d->userThreads && d->threads != g[0]) || (d->slaneThreads && d->slaneThreads != g[0])) {
        continue;
      }
      if (nTok <= g[1] * (g[0] / 64 - 3)) {
        d->xlane = g[1];
        d->threads = g[0];
        /* more utterances than CUs: the LM-state memo moves to HBM, 28 KB of LDS and 81 VGPRs let several
         * workgroups share a CU (see the lane engine with LM terms below) */
        d->yshare = (d->userYshare >= 0 ? d->userYshare != 0 : B > d->ctx->numCUs) ? 1 : 0;
        d->ymemoSlots = kXlMemoH;
        break;
      }
    }
  }
  /* ... with the LM terms and two lane groups (fltx_ylane.h): n-gram word LM and / or smeared trie,
   * beams up to 128 */
  d->ylane = 0;
  if (d->kind == FLTX_DECODER_LEXICON && !d->noYlane && !d->genericAsked && (!d->xlane || d->preferYlane) &&
      d->offlineCall && !d->keepScores && !forceWorstCaseCap && !d->forceGlobalWs &&
      (d->lm->kind == 0 || d->lm->kind == 1) && !d->isLmToken && d->trie && d->trie->xOk &&
      /* (several words per spelling: with an n-gram LM only -- under ZeroLM the words of a spelling tie in one LM state; one
       * and two lane groups only -- with four, the word wave's twelve candidate slots need twice the 128 registers a
       * 1 024-thread workgroup leaves a wave, and the spills made it 2.3 x slower than the generic engine on the
       * reference's test lexicon at beam 256) */
      (!d->trie->xMulti || (d->lm->kind == 1 && K <= 128 && d->userYlaneGroups <= 2)) && d->trie->xEndTok == d->sil &&
      (d->opt.criterion == FLTX_CRITERION_CTC ? (d->sil != d->blank && d->blank >= 0 && d->blank < N)
                                               : (d->nTrans == N * N && !d->noYlaneAsg)) &&
      !(d->opt.unk_score > -std::numeric_limits<double>::infinity()) && K <= 256 && N <= 64 &&
      d->opt.beam_threshold >= 0.0 && d->sil >= 0 && d->sil < N) {
    const int ng = std::max(K <= 64 ? 1 : (K <= 128 ? 2 : 4), d->userYlaneGroups);
    /* More utterances than CUs: the geometry of which two workgroups fit a CU (512 threads, <= 128
     * VGPRs, 77 KB of LDS: the LM-state memo moves to HBM) -- one utterance's waits are the other's
     * time to run (C5's 1 024 utterances per GPU: 1.6x).  With no more utterances than CUs the second
     * workgroup would not exist and the memo in LDS is the faster one. */
    /* Four lane groups (beams 129 .. 256): sixteen waves, the memo in HBM whatever the batch (the lanes, the merge
     * and orphan tables and ten token waves' pair lists fill the LDS); a long utterance creates more LM states than
     * the memo in LDS numbers (about 2.2 per frame on the C4 shape, 6 144 at most): it takes the HBM memo as well,
     * sized for its frames, instead of leaving the engine half way */
    const bool longUtt = (int64_t)maxT * 5 / 2 + 64 > kYlMemo * 3 / 4;
    /* (several words per spelling: the larger merge table takes the memo's place in LDS -- memo in HBM always) */
    const bool share = (ng == 4 || d->trie->xMulti) ? true : (d->userYshare >= 0 ? d->userYshare != 0 : (B > d->ctx->numCUs || longUtt));
    const bool multi = d->trie->xMulti; /* (one workgroup per CU: two lane groups keep their eight token waves) */
    const int threads = ng == 4 ? 1024 : (ng == 1 ? 512 : ((share && !multi) ? 512 : 768));
    const int nTokWaves = threads / 64 - ng - 2;
    const int tpw = (nTok + nTokWaves - 1) / nTokWaves;
    const int pairCap = ng == 4 ? kYlPairs4 : ((ng == 2 && share && !multi) ? 1024 : 512);
    const int maxTokWaves = ng == 4 ? 10 : 8;
    if ((!d->userThreads || d->threads == threads) && tpw * nTokWaves <= 96 && tpw * 64 * ng <= pairCap &&
        nTokWaves <= maxTokWaves) {
      /* slots of the memo in HBM: four per LM state the utterance can create, a power of two, 16-bit state numbers */
      uint32_t ms = kYlMemo;
      /* (C4 shape: 2.2 new LM states per frame at beam 100, 4 .. 6.5 at beam 256; a memo no bigger than it has to be
       * stays in the L2: 8 192 slots unless the utterance is long or the beam needs four groups) */
      /* (several words per spelling: every further word that stays in the beam is one more LM state -- four times as many) */
      const int perFrame = (ng == 4 ? 8 : 5) * (d->trie->xMulti ? 4 : 1);
      while ((longUtt || ng == 4 || d->trie->xMulti) && ms < 65536u && (int64_t)ms * 3 / 4 < (int64_t)maxT * perFrame + 256) {
        ms *= 2;
      }
      d->ymemoSlots = share ? ms : kYlMemo;
      d->yshare = share ? 1 : 0;
      d->ylane = ng;
      d->ylaneRounds = ng == 1 ? 2 : 4;
      d->ylaneTpw = tpw;
      d->ylaneLm = ((d->lm->kind != 0 || !d->trie->xZeroSmear) ? 1 : 0) | (d->opt.criterion == FLTX_CRITERION_CTC ? 0 : 2) |
                   (d->trie->xMulti ? 4 : 0) | (d->opt.log_add ? 8 : 0);
      d->threads = threads;
      d->xlane = 0;
      if (d->lm->kind == 1 && (d->xlmwordTrie != d->trie || d->xlmwordLm != d->lm)) {
        const std::vector<int32_t>& el = d->trie->xEndHost;
        std::vector<int32_t> w(el.size());
        for (size_t i = 0; i < el.size(); ++i) { /* KenLM::score: usrToLmIdxMap_, unknown words -> <unk> */
          w[i] = el[i] < 0 ? -1 : ((size_t)el[i] < d->lm->hUsr.size() ? d->lm->hUsr[(size_t)el[i]] : d->lm->unk);
        }
        if (d->xlmword.ensure(sizeof(int32_t) * std::max<size_t>(1, w.size()), d->ctx->stream, false) ||
            devCopyH2D(d->xlmword.p, w.data(), sizeof(int32_t) * w.size(), d->ctx->stream)) {
          return fail(FLTX_ERR_OOM, "LM word ids of the lexicon: upload failed");
        }
        devSync(d->ctx->stream); /* (w is a local) */
        d->xlmwordTrie = d->trie;
        d->xlmwordLm = d->lm;
      }
    }
  }
  /* ... and the chunks of a stream with a token-level n-gram LM (slaneUtterance<.., ST, TL>): begin / end / prune /
   * getBestHypothesis stay the generic engine's kernels -- they know n-gram LMs and keep a state's context row in stateCtx --
   * and the chunk's frames take their state ids from that engine's (parent id, edge) -> id table */
  d->tstream = 0;
  if (d->tokLm && d->kind == FLTX_DECODER_LEXFREE && !d->noSlane && !d->noTlane && !d->noSstream && !d->genericAsked &&
      !d->offlineCall && d->keepScores && !d->opt.log_add && K <= 64 && d->recycle && d->opt.beam_threshold >= 0.0 && d->sil >= 0 && d->sil < N &&
      (d->opt.criterion != FLTX_CRITERION_CTC || (d->blank >= 0 && d->blank < N))) {
    static const int geoS[][2] = {{576, 4}, {512, 5}, {576, 10}};
    const int nList = nTok - ((d->opt.criterion == FLTX_CRITERION_CTC && d->opt.beam_size_token >= N) ? 1 : 0);
    for (const auto& g : geoS) {
      if (nList <= g[1] * (g[0] / 64 - 2)) {
        d->sstream = g[1];
        d->sstreamThreads = g[0];
        d->tstream = 1;
        break;
      }
    }
  }
  { /* which eligibility terms kept this call off the lane engines (fltx_decoder_get "why_not_lane") */
    int64_t why = 0;
    if (!(d->slane || d->xlane || d->ylane || d->sstream)) {
      const bool lexi = d->kind == FLTX_DECODER_LEXICON;
      const bool unkOn = d->opt.unk_score > -std::numeric_limits<double>::infinity();
      const int nListAll = nTok - ((d->opt.criterion == FLTX_CRITERION_CTC && d->opt.beam_size_token >= N) ? 1 : 0);
      why |= (N > 64 && (lexi || N > kWlMaxN || nTok > 64)) ? FLTX_WHY_TOKENS : 0; /* (lexicon-free: the token BEAM has to fit, fltx_wlane.h) */
      why |= (lexi ? K > ((d->trie && d->trie->xMulti) ? 128 : 256) : K > 64 * kMlMaxGroups) ? FLTX_WHY_BEAM : 0;
      why |= (!d->offlineCall && (lexi || d->opt.log_add || d->lm->kind == 1)) ? FLTX_WHY_STREAM : 0;
      /* (lexicon-free + n-gram LM: the TL variants of fltx_slane.h / fltx_mlane.h take it at beams up to 512 when the
       * model's contexts fit a dense table -- what is left of the term there: a host LM, a model too large for the
       * table; a token list no compiled geometry covers shows as GEOMETRY below) */
      why |= (d->lm->kind == 2 || (lexi && d->isLmToken) || (!lexi && d->lm->kind == 1 && d->tokLm == nullptr)) ? FLTX_WHY_LM : 0;
      /* (logAdd on the lexicon lane engines: CTC, one word per spelling) */
      /* (logAdd on the lexicon decoder is no reason any more: every configuration the lexicon lane engines take without
       * it, they take with it) */
      why |= (lexi && d->opt.criterion != FLTX_CRITERION_CTC && d->noYlaneAsg) ? FLTX_WHY_ASG : 0;
      why |= (lexi && unkOn) ? FLTX_WHY_UNK : 0;
      why |= (lexi && d->trie && (!d->trie->xOk || (d->trie->xMulti && d->lm->kind == 0))) ? FLTX_WHY_TRIE_SHAPE : 0;
      why |= (lexi && d->trie && d->trie->xOk && (d->trie->xEndTok != d->sil || d->sil == d->blank)) ? FLTX_WHY_WORD_END : 0;
      why |= (!(d->opt.beam_threshold >= 0.0) || d->sil < 0 || d->sil >= N ||
              (d->opt.criterion == FLTX_CRITERION_CTC && (d->blank < 0 || d->blank >= N))) ? FLTX_WHY_OPTIONS : 0;
      why |= ((int64_t)K * (maxT + 2) >= (lexi || K <= 64 ? (1ll << 23) - 1 : (1ll << 31) - 1)) ? FLTX_WHY_LENGTH : 0;
      why |= (d->noSlane || d->noXlane || d->noYlane || d->genericAsked || d->forceGlobalWs || d->noLean || d->noDense ||
              d->userLaneGroups < 0 || forceWorstCaseCap || (d->offlineCall && d->keepScores) ||
              (!lexi && N > 64 && d->noWlane) || (!lexi && d->lm->kind == 1 && d->noTlane)) ? FLTX_WHY_SWITCHED_OFF : 0;
      why |= (!lexi && N > 64 && d->opt.log_add) ? FLTX_WHY_LOGADD : 0; /* (fltx_wlane.h has no logAdd variant) */
      why |= (!lexi && nListAll > 70) ? FLTX_WHY_GEOMETRY : 0;
      if (!why) {
        why = FLTX_WHY_GEOMETRY; /* (no compiled geometry covers this token list / thread count) */
      }
    }
    d->whyNotLane = why;
  }
  if (d->lean && !d->lane && !d->slane) { /* the lean steps keep their (record-free) workspace in LDS or are not used */
    Ws t2;
    const size_t need = carveWs(t2, nullptr, K, 1, 64, 1024, N, K + 256, 2, 0, 0, 0, d->threads / 64);
    if (need > kMaxLds) {
      d->lean = 255; /* too big for LDS: the streaming step over an HBM workspace (leanBarrier) */
      leanInHbm = true;
      if (!d->userThreads) {
        d->threads = 1024; /* (C2 shape, beam 1000: 314 -> 234 ms, beam 2500: 1375 -> 1046) */
      }
    }
  }
  if (d->lean) {
    d->dense = 2; /* lean-only workspace layout (carveWs) */
  }
  int64_t worst = d->kind == FLTX_DECODER_LEXFREE ? (int64_t)K * (nTok + (d->dense ? 1 : 0))
                                                  : (int64_t)K * ((int64_t)nTok * 8 + 2);
  worst = std::max<int64_t>(worst, K);
  if (d->lean) {
    worst = 1; /* no candidate records at all */
  }
  int64_t capC = worst;
  d->NB = 1024;
  d->SCAP = K + 256;
  Ws tmp;
  auto hsFor = [&](int64_t c) {
    if (d->lean) {
      return 64; /* the merge hash is not used */
    }
    int64_t keys = d->dense ? K : c;
    return std::max((int)nextPow2((uint64_t)keys * 2), 64);
  };
  /* lexicon decoder: generate from a list of the (hypothesis, token) pairs that
   * have a child in the trie when the full grid would take several rounds */
  d->itemCap = 0;
  d->itemWide = 0;
  if (d->kind == FLTX_DECODER_LEXICON && !d->noItems && !d->forceGlobalWs && N <= 64 && d->trie &&
      d->trie->mask.p && (int64_t)K * nTok > d->threads && (int64_t)K * nTok < (1 << 29)) {
    /* (an item is hypothesis << 6 | token: 16 bits up to beam 1 024, a 32-bit word -- two of the list's uint16 words --
     * beyond: the reference's own test decodes at beam 2 500, where the full grid is 76 rounds of mostly empty cells) */
    d->itemWide = K > 1024 ? 1 : 0;
    d->itemCap = K * nTok * (d->itemWide ? 2 : 1);
  }
  const int itemCap0 = d->itemCap;
  auto bytesFor = [&](int64_t c) {
    return carveWs(tmp, nullptr, K, (int)c, hsFor(c), d->NB, N, d->SCAP, d->dense, d->lane, 0, d->itemCap,
                   d->threads / 64);
  };
  bool lds = !d->forceGlobalWs && !leanInHbm;
  d->CAP2 = 0;
  d->cutM = 0;
  d->cutRecompute = 0;
  const bool forceCut = d->userCutM > 0 && d->kind == FLTX_DECODER_LEXICON && !forceWorstCaseCap; /* tests */
  if (lds && (bytesFor(capC) > kMaxLds || forceCut)) {
    if (d->kind == FLTX_DECODER_LEXICON && !forceWorstCaseCap) {
      /* trie fan-out is sparse: size for the LDS and retry in HBM on overflow */
      int64_t c = capC;
      while (c > K && bytesFor(c) > kMaxLds) {
        c = c * 3 / 4;
      }
      const bool fits = c >= std::max<int64_t>(K * 4, 64) && bytesFor(c) <= kMaxLds;
      /* About K * (nTok + 2) candidates pass the pre-filter in a frame (measured:
       * C3 mean 280 / max 580 of 600, beam 100 x 29 tokens mean 1300 / max 2900
       * of 3100).  With room for 1.5x that, build every candidate's record in
       * one pass.  Otherwise (max-merge only) score every candidate into a slim
       * {score, order} list first and build records only for the best
       * cutM = 3K + 64 of them (runFrame): the record area then holds a few
       * hundred entries and the slim list takes what is left of the LDS. */
      const int64_t expect = (int64_t)K * (nTok + 2) * 3 / 2;
      bool cut = false;
      if ((!fits || c < expect || forceCut) && !d->noCut && !d->opt.log_add) {
        /* a merge group has up to three members: 3K + 64 of the best candidates hold K groups */
        const int64_t M = d->userCutM > 0 ? std::max<int64_t>(d->userCutM, K) : 3 * (int64_t)K + 64;
        const int64_t capRec = std::max<int64_t>(2 * M, 512);
        auto bytesCut = [&](int64_t c2) {
          return carveWs(tmp, nullptr, K, (int)capRec, hsFor(capRec), d->NB, N, d->SCAP, d->dense, d->lane, (int)c2,
                         d->itemCap, d->threads / 64);
        };
        int64_t c2 = std::min<int64_t>(worst, 16384);
        while (c2 > 4 * M && bytesCut(c2) > kMaxLds) {
          c2 = c2 * 7 / 8;
        }
        /* the slim list should hold what a frame produces (about K * (nTok + 2));
         * otherwise generate twice instead of remembering (runFrame) */
        if (!d->noSlim && bytesCut(c2) <= kMaxLds && c2 >= std::min<int64_t>(std::max<int64_t>(4 * M, expect * 2 / 3), worst)) {
          d->CAP2 = (int)c2;
          d->cutM = (int)M;
          capC = capRec;
          cut = true;
        } else {
          int64_t Mr = M;
          int64_t capRec2 = std::max<int64_t>(Mr * 5 / 4 + 64, 256);
          auto bytesRe = [&]() {
            return carveWs(tmp, nullptr, K, (int)capRec2, hsFor(capRec2), d->NB, N, d->SCAP, d->dense, d->lane, 0,
                           d->itemCap, d->threads / 64);
          };
          if (bytesRe() > kMaxLds && d->itemCap && d->userCutM <= 0) {
            /* The item list is worth more than the third member of every group
             * (groups average ~1.3 members): keep 2K + 64 if that makes it fit.
             * A frame whose merge then yields fewer than K groups although a
             * candidate above the threshold was left out is flagged as always
             * (beam 300 x 29 tokens: 15.6 -> 9.5 ms, none flagged). */
            const int64_t M2 = 2 * (int64_t)K + 64, cap2 = std::max<int64_t>(M2 * 5 / 4 + 64, 256);
            if (carveWs(tmp, nullptr, K, (int)cap2, hsFor(cap2), d->NB, N, d->SCAP, d->dense, d->lane, 0, d->itemCap,
                        d->threads / 64) <= kMaxLds) {
              Mr = M2;
              capRec2 = cap2;
            }
          }
          if (bytesRe() > kMaxLds && d->itemCap) {
            d->itemCap = 0; /* the grid form of the generation needs no list */
          }
          if (bytesRe() <= kMaxLds) {
            d->cutRecompute = 1;
            d->cutM = (int)Mr;
            capC = capRec2;
            cut = true;
          }
        }
      }
      if (!cut) {
        if (fits) {
          capC = c;
        } else {
          lds = false;
        }
      }
    } else {
      lds = false;
    }
  }
  if (!lds) {
    d->lane = 0;
    d->itemCap = 0;
    if (!d->userThreads && d->kind == FLTX_DECODER_LEXICON && !leanInHbm) {
      /* a beam that does not fit the LDS has work for sixteen waves, and the barriers of this path no longer cost
       * per wave (wsNoInv): C4 shape, beam 500 108 -> 86 ms, beam 1000 265 -> 195, beam 2500 961 -> 657 */
      d->threads = 1024;
    }
    /* Lexicon beams too big for the LDS: the recompute form of the cut-off
     * generation still keeps the candidate records few (3K + 64), so those and
     * the merge hash usually fit the LDS while the beam itself, its select
     * lists and the item list live in HBM (carveWs level 2). */
    if (d->kind == FLTX_DECODER_LEXICON && !forceWorstCaseCap && !d->forceGlobalWs && !leanInHbm && !d->noCut &&
        !d->opt.log_add && N <= 64) {
      const int64_t M = d->userCutM > 0 ? std::max<int64_t>(d->userCutM, K) : 3 * (int64_t)K + 64;
      d->cutM = (int)M;
      capC = std::max<int64_t>(M * 5 / 4 + 64, 256);
      d->itemCap = itemCap0;
      if (!d->noSlim) { /* HBM has room for the slim list of everything a frame can produce */
        d->CAP2 = (int)std::min<int64_t>(worst, std::max<int64_t>(4 * M, (int64_t)K * (nTok + 2) * 3));
      } else {
        d->CAP2 = 0;
        d->cutRecompute = 1;
      }
    }
  }
  d->CAP = (int)capC;
  d->HS = hsFor(capC);
  d->hotBytes = 0;
  d->hotLevel = 0;
  if (!lds) {
    d->hotLevel = 1;
    if (!leanInHbm) {
      size_t hb = 0;
      carveWs(tmp, nullptr, K, d->CAP, d->HS, d->NB, N, d->SCAP, d->dense, d->lane, d->CAP2, d->itemCap,
              d->threads / 64, 2, nullptr, &hb);
      if (hb <= kMaxLdsHw && d->maxHotLevel >= 2) {
        d->hotLevel = 2;
      }
    }
    d->wsBytes = carveWs(tmp, nullptr, K, d->CAP, d->HS, d->NB, N, d->SCAP, d->dense, d->lane, d->CAP2, d->itemCap,
                         d->threads / 64, d->hotLevel, nullptr, &d->hotBytes);
  } else {
    d->wsBytes = carveWs(tmp, nullptr, K, d->CAP, d->HS, d->NB, N, d->SCAP, d->dense, d->lane, d->CAP2, d->itemCap,
                         d->threads / 64);
  }
  d->wsInLds = lds;
  if (d->slane) {
    d->wsBytes = d->wlane          ? sizeof(WlaneLds)
                 : d->mlaneNG == 2 ? sizeof(MlaneLds<2>)
                 : d->mlaneNG == 4 ? sizeof(MlaneLds<4>)
                 : d->mlaneNG == 8 ? sizeof(MlaneLds<8>)
                 : d->tlane        ? sizeof(TlaneLds) + 8 * (size_t)d->tlEdgeSlots + 12 * (size_t)d->tlMaskSlots
                                   : offsetof(SlaneLds, amNB); /* (the stream variant's arrays are its last members) */
    d->wsInLds = true;
    lds = true;
  }
  if (d->ylane) {
    /* (shared-CU geometries: the part of YlaneLds::pscore their token waves use -- five waves x 512 floats with one lane
     * group, four with two -- so that two workgroups still fit a CU's 160 KB) */
    const size_t pscorePart = (size_t)(d->ylane == 2 ? 4 : 5) * 512 * sizeof(float);
    if (d->ylaneLm & 4) { /* several words per spelling: a larger merge table and the further words' lists (one workgroup per CU) */
      using YlaneLdsMl = YlaneLdsT<2, true>;
      using YlaneLdsMlLa = YlaneLdsT<2, true, true>;
      static_assert(offsetof(YlaneLdsMl, pscore) + 8 * 512 * sizeof(float) <= 160 * 1024, "one CU's LDS");
      /* (memo in HBM; one and two lane groups only; two groups: 768 threads, eight token waves' kept scores) */
      static_assert(offsetof(YlaneLdsMlLa, pscore) + 8 * 512 * sizeof(float) <= 160 * 1024, "one CU's LDS");
      d->wsBytes = ((d->ylaneLm & 8) ? offsetof(YlaneLdsMlLa, pscore) : offsetof(YlaneLdsMl, pscore)) +
                   (d->ylane == 2 ? (size_t)8 * 512 * sizeof(float) : pscorePart);
    } else if (d->ylaneLm & 8) { /* logAdd: the merge-table slots' sums */
      using YlaneLdsLa = YlaneLdsT<2, false, true>;
      using YlaneLdsLa4 = YlaneLdsT<4, false, true>;
      static_assert(sizeof(YlaneLdsLa) <= 160 * 1024 && offsetof(YlaneLdsLa4, memo) <= 160 * 1024, "one CU's LDS");
      d->wsBytes = d->ylane == 4 ? offsetof(YlaneLdsLa4, memo)
                                 : (d->yshare ? offsetof(YlaneLdsLa, pscore) + pscorePart : sizeof(YlaneLdsLa));
    } else {
      d->wsBytes = d->ylane == 4 ? offsetof(YlaneLdsT<4>, memo)
                                 : (d->yshare ? offsetof(YlaneLds, pscore) + pscorePart : sizeof(YlaneLds));
    }
    d->wsInLds = true;
    lds = true;
    d->itemCap = 0;
    d->CAP2 = 0;
    d->cutM = 0;
    d->cutRecompute = 0;
  }
  if (d->xlane) {
    d->wsBytes = d->yshare ? offsetof(XlaneLds, memo) : sizeof(XlaneLds);
    d->wsInLds = true;
    lds = true;
    d->itemCap = 0;
    d->CAP2 = 0;
    d->cutM = 0;
    d->cutRecompute = 0;
  }
  /* buffers */
  bool grewTab = false;
  int rc = 0;
  rc |= d->histOffD.ensure(sizeof(int64_t) * (B + 1), st, false);
  rc |= d->histPT.ensure(sizeof(int2) * (size_t)off, st, false);
  if (d->wlane) {
    rc |= d->tokRowsBuf.ensure(sizeof(WlTokRow) * ((size_t)off / (size_t)K + 1), st, false);
  }
  if (d->kind == FLTX_DECODER_LEXICON) {
    rc |= d->histW.ensure(sizeof(int32_t) * (size_t)off, st, false);
  }
  if (d->keepScores) {
    rc |= d->histS.ensure(sizeof(double) * 3 * (size_t)off, st, false);
  }
  if (cap != d->stateCap) {
    /* geometry changed: stale keys would hash to other slots; start clean */
    d->stateTab.cap = 0;
    if (d->stateTab.p) {
      devSync(st);
      devFree(d->stateTab.p);
      d->stateTab.p = nullptr;
    }
    d->stateCap = cap;
    d->epoch = 0;
  }
  /* (the lean / lane engines name their states with a counter and childTab: no table) */
  rc |= d->stateTab.ensure(sizeof(unsigned long long) * ((d->lean || d->tlane) ? 1 : (size_t)B * cap), st, true, &grewTab);
  if (grewTab) {
    d->epoch = 0;
  }
  if (d->lm->kind == 1 && !d->tlane) { /* (TL: a state's context is a row number kept with its lane) */
    rc |= d->stateCtx.ensure(sizeof(int32_t) * (size_t)B * cap * std::max(1, d->lm->order - 1), st, false);
  }
  size_t bk = (size_t)B * K;
  rc |= d->gScore.ensure(8 * bk, st, false) | d->gAm.ensure(8 * bk, st, false) | d->gLm.ensure(8 * bk, st, false);
  rc |= d->gState.ensure(4 * bk, st, false) | d->gSPar.ensure(4 * bk, st, false) | d->gSEdge.ensure(4 * bk, st, false);
  rc |= d->gLex.ensure(4 * bk, st, false) | d->gTokPb.ensure(4 * bk, st, false) | d->gLexMax.ensure(4 * bk, st, false);
  rc |= d->uttNBeam.ensure(4 * (size_t)B, st, true) | d->uttFrame.ensure(4 * (size_t)B, st, true);
  rc |= d->uttTotal.ensure(4 * (size_t)B, st, true) | d->uttStatus.ensure(4 * (size_t)B, st, true);
  rc |= d->outN.ensure(4 * (size_t)B, st, true) | d->outScores.ensure(8 * bk * 3, st, false);
  for (int u = 0; u < 2; ++u) {
    rc |= d->emOff[u].ensure(sizeof(int64_t) * (size_t)B, st, false) | d->stepT[u].ensure(4 * (size_t)B, st, false);
  }
  if (!lds) {
    rc |= d->gws.ensure(d->wsBytes * (size_t)B, st, false);
  }
  d->idFamily = d->lean ? 0 : 1;
  if (d->lean && !d->slane) {
    d->idCap = std::min<int64_t>(d->recycle ? streamIds : (int64_t)K * (idT + 2) + 2, (1ll << 23) - 2);
    rc |= d->childTab.ensure(4 * (size_t)B * d->idCap * N, st, false);
    rc |= d->maskTab.ensure(8 * (size_t)B * d->idCap, st, false);
    rc |= d->uttNextId.ensure(4 * (size_t)B, st, true);
    rc |= d->gMask.ensure(8 * bk, st, false);
  } else if (d->recycle) {
    d->idCap = std::min<int64_t>(streamIds, (1ll << 23) - 2);
    rc |= d->uttNextId.ensure(4 * (size_t)B, st, true);
    rc |= d->stateVal.ensure(4 * (size_t)B * cap, st, false);
    rc |= d->idEdge.ensure(4 * (size_t)B * d->idCap, st, false);
  }
  if (d->recycle) {
    if (streamIds > (1ll << 23) - 2) {
      return fail(FLTX_ERR_UNSUPPORTED, "beam %d x max_frames %d exceeds the 2^23 LM-state ids a stream indexes", K, idT);
    }
    rc |= d->idPar.ensure(4 * (size_t)B * d->idCap, st, false) | d->idBorn.ensure(4 * (size_t)B * d->idCap, st, false);
    rc |= d->idKeep.ensure((size_t)B * d->idCap, st, false) | d->idNew.ensure(4 * (size_t)B * d->idCap, st, false);
    rc |= d->idList.ensure(4 * (size_t)B * d->idCap, st, false);
    d->idsFreeBound = (int64_t)K * (idT + 2);
  }
  if ((d->ylane || d->xlane) && d->yshare) {
    rc |= d->ymemo.ensure(sizeof(unsigned long long) * (size_t)d->ymemoSlots * (size_t)B, st, false); /* (wiped by the kernel) */
  }
  if (d->tlane && d->mlaneNG > 1) { /* the token-LM variant of fltx_mlane.h: its state-id table */
    rc |= d->ymemo.ensure(sizeof(unsigned long long) * (size_t)d->ymemoSlots * (size_t)B, st, false); /* (wiped by the kernel) */
  }
  d->useLmCache = d->lm->kind == 1 && !d->ylane && !d->xlane && !d->noLmCache;
  if (d->useLmCache) {
    /* (emptied here and at every new epoch: LM-state ids are table slots, a new batch gives them new meanings) */
    rc |= d->lmCache.ensure(8 * (size_t)kLmCache * (size_t)B, st, false);
    if (!rc) {
      devMemset(d->lmCache.p, 0xFF, 8 * (size_t)kLmCache * (size_t)B, st);
    }
  }
  if (d->lm->kind == 1) {
    rc |= d->scored.ensure(4 * (size_t)B, st, false);
    if (!rc && !d->keepScored) { /* (a partial re-run keeps the counts of the utterances it does not touch) */
      devMemset(d->scored.p, 0, 4 * (size_t)B, st);
    }
  }
  if (rc) {
    return fail(FLTX_ERR_OOM, "device allocation failed (B=%d K=%d T<=%d)", B, K, maxT);
  }
  if (devCopyH2D(d->histOffD.p, d->histOff.data(), sizeof(int64_t) * (B + 1), st)) {
    return fail(FLTX_ERR_HIP, "upload failed");
  }
  return FLTX_OK;
}

void fillParams(fltx_decoder* d, DecodeParams& P) {
  memset(&P, 0, sizeof(P));
  P.K = d->opt.beam_size;
  P.Kt = d->opt.beam_size_token;
  P.beamThreshold = d->opt.beam_threshold;
  P.lmWeight = d->opt.lm_weight;
  P.wordScore = d->opt.word_score;
  P.unkScore = d->opt.unk_score;
  P.silScore = d->opt.sil_score;
  P.logAdd = d->opt.log_add;
  P.criterion = d->opt.criterion;
  P.sil = d->sil;
  P.blank = d->blank;
  P.unk = d->unk;
  P.isLmToken = d->isLmToken;
  P.kind = d->kind;
  P.N = d->N;
  P.transitions = d->nTrans ? d->transitions.as<float>() : nullptr;
  if (d->trie) {
    P.trieEdge = d->trie->edge.as<TrieEdge>();
    P.trieMask = d->trie->mask.p ? d->trie->mask.as<unsigned long long>() : nullptr;
    P.itemCap = P.trieMask ? d->itemCap : 0;
    P.itemWide = d->itemWide;
    P.trieLabels = d->trie->labels.as<int32_t>();
  }
  P.lmKind = d->lm->kind;
  P.lmOrder = d->lm->order;
  P.ngTab = d->lmDev ? d->lmDev->tab.as<NgramSlot>() : nullptr;
  P.ngMask = d->lm->mask;
  P.ngBackoff = d->lmDev ? d->lmDev->backoff.as<float>() : nullptr;
  P.usrToLm = d->lmDev ? d->lmDev->usrToLm.as<int32_t>() : nullptr;
  P.nUsr = d->lm->nUsr;
  P.lmBos = d->lm->bos;
  P.lmEos = d->lm->eos;
  P.lmUnk = d->lm->unk;
  P.stateCtx = d->stateCtx.as<int32_t>();
  P.emOff = d->emOff[d->upSlot].as<int64_t>();
  P.stepT = d->stepT[d->upSlot].as<int32_t>();
  P.uttNBeam = d->uttNBeam.as<int32_t>();
  P.uttFrame = d->uttFrame.as<int32_t>();
  P.uttTotal = d->uttTotal.as<int32_t>();
  P.uttStatus = d->uttStatus.as<int32_t>();
  P.gScore = d->gScore.as<double>();
  P.gAm = d->gAm.as<double>();
  P.gLm = d->gLm.as<double>();
  P.gState = d->gState.as<uint32_t>();
  P.gSPar = d->gSPar.as<uint32_t>();
  P.gSEdge = d->gSEdge.as<int32_t>();
  P.gLex = d->gLex.as<uint32_t>();
  P.gTokPb = d->gTokPb.as<uint32_t>();
  P.histPT = d->histPT.as<int2>();
  P.tokRows = d->wlane ? d->tokRowsBuf.p : nullptr;
  P.tokRowBlocks = d->wlane ? std::max(1, (d->wlMaxT + 3) / 4) : 1;
  P.histW = d->histW.as<int32_t>();
  P.histS = d->keepScores ? d->histS.as<double>() : nullptr;
  P.histOff = d->histOffD.as<int64_t>();
  P.stateTab = d->stateTab.as<unsigned long long>();
  P.stateCap = d->stateCap;
  P.epoch = d->epoch;
  P.CAP = d->CAP;
  P.HS = d->HS;
  P.NB = d->NB;
  P.SCAP = d->SCAP;
  P.dense = d->dense;
  P.lane = d->lane;
  P.CAP2 = d->CAP2;
  P.cutM = d->cutM;
  P.cutRecompute = d->cutRecompute;
  P.hotLevel = d->hotLevel;
  P.gLexMax = d->gLexMax.as<float>();
  P.gws = d->wsInLds ? nullptr : d->gws.as<char>();
  P.gwsStride = (int64_t)d->wsBytes;
  P.outN = d->outN.as<int32_t>();
  P.outScores = d->outScores.as<double>();
  P.childTab = d->childTab.as<uint32_t>();
  P.maskTab = d->maskTab.as<unsigned long long>();
  P.idCap = d->idCap;
  P.uttNextId = d->uttNextId.as<int32_t>();
  P.gMask = d->gMask.as<unsigned long long>();
  P.prof = nullptr;
  P.xnode = (d->trie && d->trie->xnode.p) ? d->trie->xnode.as<XNode>() : nullptr;
  P.xEndTok = d->trie ? d->trie->xEndTok : -1;
  P.xdelta = (d->trie && d->trie->xdelta.p) ? d->trie->xdelta.as<float>() : nullptr;
  P.xextra = (d->trie && d->trie->xextra.p) ? d->trie->xextra.as<uint32_t>() : nullptr;
  P.yTpw = d->ylaneTpw;
  P.xlmword = d->xlmword.p ? d->xlmword.as<int32_t>() : nullptr;
  P.ymemo = (((d->ylane || d->xlane) && d->yshare) || (d->tlane && d->mlaneNG > 1)) ? d->ymemo.as<unsigned long long>() : nullptr;
  P.ymemoSlots = d->ymemoSlots;
  P.yRankAt = d->userYRankAt;
  P.tune = d->userTune;
  P.statusHost = (!d->offlineCall && d->streamOpt) ? (int32_t*)d->hStat.p : nullptr;
  /* (the lean step on an HBM workspace reads its atomically ORed addMask words at L2 as well: wsLoadAtomic64) */
  P.wsNoInv = (!d->wsInLds && d->hotLevel >= 1) ? 1 : 0;
  P.lmCache = d->useLmCache ? d->lmCache.as<unsigned long long>() : nullptr;
  P.tokLm = d->tokLm; /* (null unless this is a lexicon-free decoder over an n-gram LM with a dense table) */
  P.tokLmStride = d->N + 1;
  P.tlEdgeSlots = d->tlEdgeSlots;
  P.tlMaskSlots = d->tlMaskSlots;
  P.yBound = d->trie ? std::max(0.0, std::max(d->opt.lm_weight * (double)d->trie->xDeltaMin,
                                              d->opt.lm_weight * (double)d->trie->xDeltaMax))
                     : 0.0;
  P.yTransMax = d->transMax;
  P.scored = (d->lm->kind == 1 && d->scored.p) ? d->scored.as<uint32_t>() : nullptr;
  if (d->recycle) {
    P.idPar = d->idPar.as<uint32_t>();
    P.idBorn = d->idBorn.as<uint32_t>();
    if (d->idFamily == 1) {
      P.idEdge = d->idEdge.as<int32_t>();
      P.stateVal = d->stateVal.as<uint32_t>();
    }
  }
  if (d->lm->kind == 2) {
    P.hlmQCap = d->hlmQCap;
    P.hlmQCount = (int32_t*)d->hlmQCount.p;
    P.hlmQ = (uint2*)d->hlmQ.p;
    P.hlmBeamN = (int32_t*)d->hlmBeamN.p;
    P.hlmBeam = (uint32_t*)d->hlmBeam.p;
  }
  P.profThread = 64 * d->profWave;
  if (d->profile && !d->prof.ensure(8 * 8 * (size_t)d->B, d->ctx->stream, true)) {
    devMemset(d->prof.p, 0, 8 * 8 * (size_t)d->B, d->ctx->stream);
    P.prof = d->prof.as<unsigned long long>();
  }
}

int launchDecode(fltx_decoder* d, const DecodeParams& P) {
  const int W = d->sstreamLaunch ? d->sstreamThreads : d->threads;
#ifdef FLTX_EMU
#include "fltx_emu_launch.inc" /* tests/emu/: host-thread dispatch over the kernel variants */
#else
  for (int i = 0; i < 3; ++i) {
    if (!d->ev[i]) {
      HIPCHK(hipEventCreate(&d->ev[i]));
    }
  }
  HIPCHK(hipEventRecord(d->ev[0], d->ctx->stream));
  const int nGrid = d->nLaunch > 0 ? d->nLaunch : d->B;
#define FLTX_LAUNCH_LDS(WW, GG)                                                                  \
  do {                                                                                           \
    HIPCHK(hipFuncSetAttribute((const void*)fltx_decode_kernel_lds<WW, GG>,                      \
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)d->wsBytes));    \
    hipLaunchKernelGGL((fltx_decode_kernel_lds<WW, GG>), dim3(nGrid), dim3(WW), d->wsBytes,      \
                       d->ctx->stream, P);                                                       \
  } while (0)
#define FLTX_LAUNCH_SPEC(WW, LX, ZL)                                                             \
  do {                                                                                           \
    HIPCHK(hipFuncSetAttribute((const void*)fltx_decode_kernel_lds_spec<WW, LX, ZL>,             \
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)d->wsBytes));    \
    hipLaunchKernelGGL((fltx_decode_kernel_lds_spec<WW, LX, ZL>), dim3(nGrid), dim3(WW),         \
                       d->wsBytes, d->ctx->stream, P);                                           \
  } while (0)
#define FLTX_LAUNCH_LANE1(WW, GG, LA, FT)                                                        \
  do {                                                                                           \
    HIPCHK(hipFuncSetAttribute((const void*)fltx_decode_kernel_lane<WW, GG, LA, FT>,             \
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)d->wsBytes));    \
    hipLaunchKernelGGL((fltx_decode_kernel_lane<WW, GG, LA, FT>), dim3(nGrid), dim3(WW),         \
                       d->wsBytes, d->ctx->stream, P);                                           \
  } while (0)
#define FLTX_LAUNCH_LANE(WW, GG)                                                                 \
  do {                                                                                           \
    const bool ft_ = d->opt.beam_size_token >= d->N;                                             \
    if (d->opt.log_add && ft_) {                                                                 \
      FLTX_LAUNCH_LANE1(WW, GG, true, true);                                                     \
    } else if (d->opt.log_add) {                                                                 \
      FLTX_LAUNCH_LANE1(WW, GG, true, false);                                                    \
    } else if (ft_) {                                                                            \
      FLTX_LAUNCH_LANE1(WW, GG, false, true);                                                    \
    } else {                                                                                     \
      FLTX_LAUNCH_LANE1(WW, GG, false, false);                                                   \
    }                                                                                            \
  } while (0)
#define FLTX_LAUNCH_SLANE(WW, GG)                                                                \
  do {                                                                                           \
    if (d->tlane && d->opt.log_add) {                                                            \
      HIPCHK(hipFuncSetAttribute((const void*)fltx_decode_kernel_tlane<WW, GG, true>,            \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)d->wsBytes));  \
      hipLaunchKernelGGL((fltx_decode_kernel_tlane<WW, GG, true>), dim3(nGrid), dim3(WW),        \
                         d->wsBytes, d->ctx->stream, P);                                         \
    } else if (d->tlane && d->profile && ((WW == 576 && GG == 4) || (WW == 512 && GG == 5))) {   \
      constexpr int PW_ = (WW == 576 && GG == 4) ? 576 : 512, PG_ = (WW == 576 && GG == 4) ? 4 : 5; \
      HIPCHK(hipFuncSetAttribute((const void*)fltx_decode_kernel_tlane<PW_, PG_, false, true>,   \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)d->wsBytes));  \
      hipLaunchKernelGGL((fltx_decode_kernel_tlane<PW_, PG_, false, true>), dim3(nGrid), dim3(PW_), \
                         d->wsBytes, d->ctx->stream, P);                                         \
    } else if (d->tlane) {                                                                       \
      HIPCHK(hipFuncSetAttribute((const void*)fltx_decode_kernel_tlane<WW, GG, false>,           \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)d->wsBytes));  \
      hipLaunchKernelGGL((fltx_decode_kernel_tlane<WW, GG, false>), dim3(nGrid), dim3(WW),       \
                         d->wsBytes, d->ctx->stream, P);                                         \
    } else if (d->opt.log_add) {                                                                        \
      hipLaunchKernelGGL((fltx_decode_kernel_slane<WW, GG, true, false>), dim3(nGrid), dim3(WW), \
                         d->wsBytes, d->ctx->stream, P);                                         \
    } else if (d->profile) {                                                                     \
      hipLaunchKernelGGL((fltx_decode_kernel_slane<WW, GG, false, true>), dim3(nGrid), dim3(WW), \
                         d->wsBytes, d->ctx->stream, P);                                         \
    } else {                                                                                     \
      hipLaunchKernelGGL((fltx_decode_kernel_slane<WW, GG, false, false>), dim3(nGrid), dim3(WW),\
                         d->wsBytes, d->ctx->stream, P);                                         \
    }                                                                                            \
  } while (0)
#define FLTX_LAUNCH(WW)                                                                          \
  do {                                                                                           \
    if (!d->wsInLds && d->lean) {                                                                \
      hipLaunchKernelGGL(fltx_decode_kernel_gwslean<WW>, dim3(nGrid), dim3(WW), d->hotBytes, d->ctx->stream, P); \
    } else if (!d->wsInLds) {                                                                    \
      HIPCHK(hipFuncSetAttribute((const void*)fltx_decode_kernel_gws<WW>,                        \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)d->hotBytes)); \
      hipLaunchKernelGGL(fltx_decode_kernel_gws<WW>, dim3(nGrid), dim3(WW), d->hotBytes, d->ctx->stream, P); \
    } else if (d->lane == 4) {                                                                   \
      FLTX_LAUNCH_LANE(WW, 4);                                                                   \
    } else if (d->lane == 8) {                                                                   \
      FLTX_LAUNCH_LANE(WW, 8);                                                                   \
    } else if (d->lean == 6) {                                                                   \
      FLTX_LAUNCH_LDS(WW, 6);                                                                    \
    } else if (d->lean == 12) {                                                                  \
      FLTX_LAUNCH_LDS(WW, 12);                                                                   \
    } else if (d->lean == 255) {                                                                 \
      FLTX_LAUNCH_LDS(WW, 255);                                                                  \
    } else if (d->kind == FLTX_DECODER_LEXICON && !d->isLmToken && d->lm->kind == 0) {           \
      FLTX_LAUNCH_SPEC(WW, true, true);                                                          \
    } else if (d->kind == FLTX_DECODER_LEXICON && !d->isLmToken && d->lm->kind == 1) {           \
      FLTX_LAUNCH_SPEC(WW, true, false);                                                         \
    } else if (d->lm->kind == 0 && !d->isLmToken) {                                              \
      FLTX_LAUNCH_SPEC(WW, false, true);                                                         \
    } else {                                                                                     \
      FLTX_LAUNCH_LDS(WW, 0);                                                                    \
    }                                                                                            \
  } while (0)
  if (d->sstreamLaunch) {
#define FLTX_LAUNCH_SSTREAM(WW, GG)                                                              \
  do {                                                                                           \
    if (d->tstream) {                                                                            \
      hipLaunchKernelGGL((fltx_decode_kernel_tlane_stream<WW, GG>), dim3(nGrid), dim3(WW),       \
                         sizeof(TlaneLds) + (size_t)(WW / 64) * kTlGatherBytes, d->ctx->stream, P); \
    } else {                                                                                     \
      hipLaunchKernelGGL((fltx_decode_kernel_slane_stream<WW, GG>), dim3(nGrid), dim3(WW), sizeof(SlaneLds), \
                         d->ctx->stream, P);                                                     \
    }                                                                                            \
  } while (0)
    switch (W * 100 + d->sstream) {
      case 57604: FLTX_LAUNCH_SSTREAM(576, 4); break;
      case 51205: FLTX_LAUNCH_SSTREAM(512, 5); break;
      case 57610: FLTX_LAUNCH_SSTREAM(576, 10); break;
      default: return fail(FLTX_ERR_INVALID, "no stream kernel for %d threads x %d positions", W, d->sstream);
    }
#undef FLTX_LAUNCH_SSTREAM
  } else if (d->ylane) {
#define FLTX_LAUNCH_YLANE(WW, NG, RR, LMK, HM)                                                          \
  do {                                                                                                  \
    HIPCHK(hipFuncSetAttribute((const void*)fltx_decode_kernel_ylane<WW, NG, RR, LMK, HM, false>,       \
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)d->wsBytes));           \
    HIPCHK(hipFuncSetAttribute((const void*)fltx_decode_kernel_ylane<WW, NG, RR, LMK, HM, true>,        \
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)d->wsBytes));           \
    if (d->profile) {                                                                                   \
      hipLaunchKernelGGL((fltx_decode_kernel_ylane<WW, NG, RR, LMK, HM, true>), dim3(nGrid), dim3(WW),  \
                         d->wsBytes, d->ctx->stream, P);                                                \
    } else {                                                                                            \
      hipLaunchKernelGGL((fltx_decode_kernel_ylane<WW, NG, RR, LMK, HM, false>), dim3(nGrid), dim3(WW), \
                         d->wsBytes, d->ctx->stream, P);                                                \
    }                                                                                                   \
  } while (0)
#define FLTX_LAUNCH_YLANE_NP(WW, NG, RR, LMK, HM) /* (no profiling variant) */                         \
  do {                                                                                                  \
    HIPCHK(hipFuncSetAttribute((const void*)fltx_decode_kernel_ylane<WW, NG, RR, LMK, HM, false>,       \
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)d->wsBytes));           \
    hipLaunchKernelGGL((fltx_decode_kernel_ylane<WW, NG, RR, LMK, HM, false>), dim3(nGrid), dim3(WW),   \
                       d->wsBytes, d->ctx->stream, P);                                                  \
  } while (0)
#define FLTX_LAUNCH_YLANE4(LMK)                                                                        \
  do {                                                                                                  \
    HIPCHK(hipFuncSetAttribute((const void*)fltx_decode_kernel_ylane<1024, 4, 4, LMK, 1, false>,        \
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)d->wsBytes));           \
    hipLaunchKernelGGL((fltx_decode_kernel_ylane<1024, 4, 4, LMK, 1, false>), dim3(nGrid), dim3(1024),  \
                       d->wsBytes, d->ctx->stream, P);                                                  \
  } while (0)
    switch (d->ylane * 100 + d->ylaneLm + (d->yshare ? 1000 : 0)) { /* lane groups, LMK, memo in HBM */
      case 100: FLTX_LAUNCH_YLANE(512, 1, 2, 0, 0); break;
      case 101: FLTX_LAUNCH_YLANE(512, 1, 2, 1, 0); break;
      case 200: FLTX_LAUNCH_YLANE(768, 2, 4, 0, 0); break;
      case 201: FLTX_LAUNCH_YLANE(768, 2, 4, 1, 0); break;
      case 1100: FLTX_LAUNCH_YLANE(512, 1, 2, 0, 1); break;
      case 1101: FLTX_LAUNCH_YLANE(512, 1, 2, 1, 1); break;
      case 1200: FLTX_LAUNCH_YLANE(512, 2, 4, 0, 1); break;
      case 1201: FLTX_LAUNCH_YLANE(512, 2, 4, 1, 1); break;
      case 1400: FLTX_LAUNCH_YLANE4(0); break;
      case 1401: FLTX_LAUNCH_YLANE4(1); break;
      case 102: FLTX_LAUNCH_YLANE_NP(512, 1, 2, 2, 0); break; /* ASG (LMK bit 1) */
      case 103: FLTX_LAUNCH_YLANE_NP(512, 1, 2, 3, 0); break;
      case 202: FLTX_LAUNCH_YLANE_NP(768, 2, 4, 2, 0); break;
      case 203: FLTX_LAUNCH_YLANE_NP(768, 2, 4, 3, 0); break;
      case 1102: FLTX_LAUNCH_YLANE_NP(512, 1, 2, 2, 1); break;
      case 1103: FLTX_LAUNCH_YLANE_NP(512, 1, 2, 3, 1); break;
      case 1202: FLTX_LAUNCH_YLANE_NP(512, 2, 4, 2, 1); break;
      case 1203: FLTX_LAUNCH_YLANE_NP(512, 2, 4, 3, 1); break;
      case 1402: FLTX_LAUNCH_YLANE4(2); break;
      case 1403: FLTX_LAUNCH_YLANE4(3); break;
      /* logAdd merges (LMK bit 3; CTC) */
      case 108: FLTX_LAUNCH_YLANE_NP(512, 1, 2, 8, 0); break;
      case 109: FLTX_LAUNCH_YLANE_NP(512, 1, 2, 9, 0); break;
      case 208: FLTX_LAUNCH_YLANE_NP(768, 2, 4, 8, 0); break;
      case 209: FLTX_LAUNCH_YLANE_NP(768, 2, 4, 9, 0); break;
      case 1108: FLTX_LAUNCH_YLANE_NP(512, 1, 2, 8, 1); break;
      case 1109: FLTX_LAUNCH_YLANE_NP(512, 1, 2, 9, 1); break;
      case 1208: FLTX_LAUNCH_YLANE_NP(512, 2, 4, 8, 1); break;
      case 1209: FLTX_LAUNCH_YLANE_NP(512, 2, 4, 9, 1); break;
      case 1408: FLTX_LAUNCH_YLANE4(8); break;
      case 1409: FLTX_LAUNCH_YLANE4(9); break;
      /* several words per spelling (LMK bit 2; with the LM terms) */
      case 1105: FLTX_LAUNCH_YLANE_NP(512, 1, 2, 5, 1); break;
      case 1107: FLTX_LAUNCH_YLANE_NP(512, 1, 2, 7, 1); break;
      case 1205: FLTX_LAUNCH_YLANE_NP(768, 2, 4, 5, 1); break;
      case 1207: FLTX_LAUNCH_YLANE_NP(768, 2, 4, 7, 1); break;
            /* logAdd under ASG (LMK 10 / 11) and over spellings with several words (13 / 15) */
      case 110: FLTX_LAUNCH_YLANE_NP(512, 1, 2, 10, 0); break;
      case 111: FLTX_LAUNCH_YLANE_NP(512, 1, 2, 11, 0); break;
      case 210: FLTX_LAUNCH_YLANE_NP(768, 2, 4, 10, 0); break;
      case 211: FLTX_LAUNCH_YLANE_NP(768, 2, 4, 11, 0); break;
      case 1110: FLTX_LAUNCH_YLANE_NP(512, 1, 2, 10, 1); break;
      case 1111: FLTX_LAUNCH_YLANE_NP(512, 1, 2, 11, 1); break;
      case 1210: FLTX_LAUNCH_YLANE_NP(512, 2, 4, 10, 1); break;
      case 1211: FLTX_LAUNCH_YLANE_NP(512, 2, 4, 11, 1); break;
      case 1410: FLTX_LAUNCH_YLANE4(10); break;
      case 1411: FLTX_LAUNCH_YLANE4(11); break;
      case 1113: FLTX_LAUNCH_YLANE_NP(512, 1, 2, 13, 1); break;
      case 1115: FLTX_LAUNCH_YLANE_NP(512, 1, 2, 15, 1); break;
      case 1213: FLTX_LAUNCH_YLANE_NP(768, 2, 4, 13, 1); break;
      case 1215: FLTX_LAUNCH_YLANE_NP(768, 2, 4, 15, 1); break;
      default: return fail(FLTX_ERR_INVALID, "no fltx_ylane.h kernel for %d lane groups", d->ylane);
    }
#undef FLTX_LAUNCH_YLANE
#undef FLTX_LAUNCH_YLANE4
#undef FLTX_LAUNCH_YLANE_NP
  } else if (d->xlane) {
#define FLTX_LAUNCH_XLANE(WW, GG)                                                                \
  do {                                                                                           \
    HIPCHK(hipFuncSetAttribute((const void*)fltx_decode_kernel_xlane<WW, GG, 0, false>,          \
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(XlaneLds))); \
    HIPCHK(hipFuncSetAttribute((const void*)fltx_decode_kernel_xlane<WW, GG, 0, true>,           \
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(XlaneLds))); \
    if (d->opt.log_add) { /* (logAdd merges: fltx_xlane.h LA) */                                \
      HIPCHK(hipFuncSetAttribute((const void*)fltx_decode_kernel_xlane<WW, GG, 0, false, true>,  \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(XlaneLds))); \
      if (d->yshare) {                                                                           \
        hipLaunchKernelGGL((fltx_decode_kernel_xlane<WW, GG, 1, false, true>), dim3(nGrid), dim3(WW), \
                           d->wsBytes, d->ctx->stream, P);                                       \
      } else {                                                                                   \
        hipLaunchKernelGGL((fltx_decode_kernel_xlane<WW, GG, 0, false, true>), dim3(nGrid), dim3(WW), \
                           d->wsBytes, d->ctx->stream, P);                                       \
      }                                                                                          \
    } else if (d->yshare) {                                                                      \
      hipLaunchKernelGGL((fltx_decode_kernel_xlane<WW, GG, 1, false>), dim3(nGrid), dim3(WW),    \
                         d->wsBytes, d->ctx->stream, P);                                         \
    } else if (d->profile) {                                                                     \
      hipLaunchKernelGGL((fltx_decode_kernel_xlane<WW, GG, 0, true>), dim3(nGrid), dim3(WW),     \
                         d->wsBytes, d->ctx->stream, P);                                         \
    } else {                                                                                     \
      hipLaunchKernelGGL((fltx_decode_kernel_xlane<WW, GG, 0, false>), dim3(nGrid), dim3(WW),    \
                         d->wsBytes, d->ctx->stream, P);                                         \
    }                                                                                            \
  } while (0)
    switch (W * 100 + d->xlane) {
      case 51202: FLTX_LAUNCH_XLANE(512, 2); break;
      case 64002: FLTX_LAUNCH_XLANE(640, 2); break;
      case 51203: FLTX_LAUNCH_XLANE(512, 3); break;
      case 57605: FLTX_LAUNCH_XLANE(576, 5); break;
      case 64010: FLTX_LAUNCH_XLANE(640, 10); break;
      default: return fail(FLTX_ERR_INVALID, "no lane = (LM state, node) kernel for %d threads x %d positions", W, d->xlane);
    }
#undef FLTX_LAUNCH_XLANE
  } else if (d->slane && d->mlaneNG > 1) {
#define FLTX_LAUNCH_MLANE(WW, GG, NG, GPW, SPW)                                                          \
  do {                                                                                                   \
    HIPCHK(hipFuncSetAttribute((const void*)fltx_decode_kernel_mlane<WW, GG, NG, GPW, SPW, false>,       \
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)d->wsBytes));            \
    HIPCHK(hipFuncSetAttribute((const void*)fltx_decode_kernel_mlane<WW, GG, NG, GPW, SPW, true>,        \
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)d->wsBytes));            \
    if (d->opt.log_add) {                                                                                \
      hipLaunchKernelGGL((fltx_decode_kernel_mlane<WW, GG, NG, GPW, SPW, true>), dim3(nGrid), dim3(WW),  \
                         d->wsBytes, d->ctx->stream, P);                                                 \
    } else {                                                                                             \
      hipLaunchKernelGGL((fltx_decode_kernel_mlane<WW, GG, NG, GPW, SPW, false>), dim3(nGrid), dim3(WW), \
                         d->wsBytes, d->ctx->stream, P);                                                 \
    }                                                                                                    \
  } while (0)
#define FLTX_LAUNCH_TMLANE(WW, GG, NG, GPW, SPW)                                                         \
  do {                                                                                                   \
    HIPCHK(hipFuncSetAttribute((const void*)fltx_decode_kernel_tmlane<WW, GG, NG, GPW, SPW, false>,      \
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)d->wsBytes));            \
    HIPCHK(hipFuncSetAttribute((const void*)fltx_decode_kernel_tmlane<WW, GG, NG, GPW, SPW, true>,       \
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)d->wsBytes));            \
    if (d->opt.log_add) {                                                                                \
      hipLaunchKernelGGL((fltx_decode_kernel_tmlane<WW, GG, NG, GPW, SPW, true>), dim3(nGrid), dim3(WW), \
                         d->wsBytes, d->ctx->stream, P);                                                 \
    } else {                                                                                             \
      hipLaunchKernelGGL((fltx_decode_kernel_tmlane<WW, GG, NG, GPW, SPW, false>), dim3(nGrid), dim3(WW), \
                         d->wsBytes, d->ctx->stream, P);                                                 \
    }                                                                                                    \
  } while (0)
    if (d->tlane) {
      switch (d->mlaneNG * 100 + d->slane) {
        case 205: FLTX_LAUNCH_TMLANE(960, 5, 2, 1, 1); break;
        case 405: FLTX_LAUNCH_TMLANE(960, 5, 4, 2, 2); break;
        case 810: FLTX_LAUNCH_TMLANE(960, 10, 8, 2, 4); break;
        case 211: FLTX_LAUNCH_TMLANE(960, 11, 2, 1, 1); break;
        case 411: FLTX_LAUNCH_TMLANE(960, 11, 4, 2, 2); break;
        default:
          return fail(FLTX_ERR_INVALID, "no token-LM fltx_mlane.h kernel for %d lane groups x %d positions", d->mlaneNG, d->slane);
      }
    } else
    switch (((W * 100 + d->slane) * 10 + d->mlaneNG) * 100 + d->mlaneGPW * 10 + d->mlaneSPW) {
      case 64004221: FLTX_LAUNCH_MLANE(640, 4, 2, 2, 1); break;
      case 96005211: FLTX_LAUNCH_MLANE(960, 5, 2, 1, 1); break;
      case 64010221: FLTX_LAUNCH_MLANE(640, 10, 2, 2, 1); break;
      case 76804441: FLTX_LAUNCH_MLANE(768, 4, 4, 4, 1); break;
      case 96005422: FLTX_LAUNCH_MLANE(960, 5, 4, 2, 2); break;
      case 96011422: FLTX_LAUNCH_MLANE(960, 11, 4, 2, 2); break;
      case 96010824: FLTX_LAUNCH_MLANE(960, 10, 8, 2, 4); break;
      default:
        return fail(FLTX_ERR_INVALID, "no fltx_mlane.h kernel for %d threads x %d positions x %d lane groups", W, d->slane,
                    d->mlaneNG);
    }
#undef FLTX_LAUNCH_MLANE
#undef FLTX_LAUNCH_TMLANE
  } else if (d->slane && d->wlane) {
    { /* the token beams of all rows first (same stream) */
      const int maxT = d->wlMaxT;
      if (maxT > 0) {
        hipLaunchKernelGGL(fltx_tokbeam_kernel, dim3((unsigned)(P.tokRowBlocks * nGrid)), dim3(256), 4 * sizeof(WlFrontLds),
                           d->ctx->stream, P);
      }
    }
#define FLTX_LAUNCH_WLANE(WW, GG)                                                                          \
  do {                                                                                                     \
    HIPCHK(hipFuncSetAttribute((const void*)fltx_decode_kernel_wlane<WW, GG>,                              \
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)d->wsBytes));              \
    hipLaunchKernelGGL((fltx_decode_kernel_wlane<WW, GG>), dim3(nGrid), dim3(WW), d->wsBytes, d->ctx->stream, P); \
  } while (0)
    switch (W * 100 + d->slane) {
      case 57605: FLTX_LAUNCH_WLANE(576, 5); break;
      case 57608: FLTX_LAUNCH_WLANE(576, 8); break;
      case 57610: FLTX_LAUNCH_WLANE(576, 10); break;
      default: return fail(FLTX_ERR_INVALID, "no fltx_wlane.h kernel for %d threads x %d positions", W, d->slane);
    }
#undef FLTX_LAUNCH_WLANE
  } else if (d->slane) {
    const int key = W * 100 + d->slane;
    switch (key) {
      case 32010: FLTX_LAUNCH_SLANE(320, 10); break;
      case 38407: FLTX_LAUNCH_SLANE(384, 7); break;
      case 44806: FLTX_LAUNCH_SLANE(448, 6); break;
      case 51205: FLTX_LAUNCH_SLANE(512, 5); break;
      case 57604: FLTX_LAUNCH_SLANE(576, 4); break;
      case 64004: FLTX_LAUNCH_SLANE(640, 4); break;
      case 51212: FLTX_LAUNCH_SLANE(512, 12); break;
      case 57610: FLTX_LAUNCH_SLANE(576, 10); break;
      default: return fail(FLTX_ERR_INVALID, "no lane = LM state kernel for %d threads x %d positions", W, d->slane);
    }
  } else
  switch (W) {
    case 64: FLTX_LAUNCH(64); break;
    case 128: FLTX_LAUNCH(128); break;
    case 256: FLTX_LAUNCH(256); break;
    case 512: FLTX_LAUNCH(512); break;
    case 1024: FLTX_LAUNCH(1024); break;
    default: return fail(FLTX_ERR_INVALID, "threads per utterance must be 64, 128, 256, 512 or 1024");
  }
#undef FLTX_LAUNCH
#undef FLTX_LAUNCH_LDS
#undef FLTX_LAUNCH_LANE
#undef FLTX_LAUNCH_LANE1
#undef FLTX_LAUNCH_SPEC
#undef FLTX_LAUNCH_SLANE
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(d->ev[1], d->ctx->stream));
  d->timed = false;
  return FLTX_OK;
#endif
}

int bumpEpoch(fltx_decoder* d) {
  if (d->epoch >= 65535u) {
    /* the whole allocation, not just this batch's part: a larger earlier batch left keys tagged
     * with epochs that are about to be reused */
    if (devMemset(d->stateTab.p, 0, d->stateTab.cap, d->ctx->stream)) {
      return fail(FLTX_ERR_HIP, "state table reset failed");
    }
    d->epoch = 0;
  }
  d->epoch += 1;
  if (d->useLmCache && d->lmCache.p && devMemset(d->lmCache.p, 0xFF, d->lmCache.cap, d->ctx->stream)) {
    return fail(FLTX_ERR_HIP, "LM score cache reset failed");
  }
  return FLTX_OK;
}

int uploadStep(fltx_decoder* d, const float* emissions, int onDevice, const int64_t* offsets,
               const int32_t* T, DecodeParams& P) {
  Stream st = d->ctx->stream;
  const int B = d->B, N = d->N;
  std::vector<int64_t> offs(B, 0);
  int64_t maxEnd = 0;
  for (int b = 0; b < B; ++b) {
    offs[b] = offsets ? offsets[b] : 0;
    if (offs[b] < 0) {
      return fail(FLTX_ERR_INVALID, "offsets[%d] is negative", b);
    }
    maxEnd = std::max<int64_t>(maxEnd, offs[b] + (int64_t)T[b] * N);
  }
  if (maxEnd > 0 && !emissions) {
    return fail(FLTX_ERR_INVALID, "emissions is null");
  }
#ifdef FLTX_EMU
  const int slot = d->offlineCall ? d->upSlot : d->upSlot ^ 1; /* (streams take turns on the two slots as on the device:
                                                                   a deferred second pass reads the previous chunk's) */
  Stream cs = st;
#else
  /* Stream chunks go to the other slot on a copy stream: it waits for what read that slot two steps ago, not for
   * the kernel of the step before, which is still running (the chunk's H2D under the previous chunk's kernel).
   * Offline batches upload on the launch stream as before (two decoder objects overlap them: bench.py). */
  int slot = d->upSlot;
  Stream cs = st;
  if (!d->offlineCall) {
    if (!d->copyStream) {
      HIPCHK(hipStreamCreateWithFlags(&d->copyStream, hipStreamNonBlocking));
      HIPCHK(hipEventCreateWithFlags(&d->evSlotFree[0], hipEventDisableTiming));
      HIPCHK(hipEventCreateWithFlags(&d->evSlotFree[1], hipEventDisableTiming));
    }
    HIPCHK(hipEventRecord(d->evSlotFree[d->upSlot], st)); /* everything queued so far may read the current slot */
    d->slotUsed[d->upSlot] = true;
    slot = d->upSlot ^ 1;
    cs = d->copyStream;
    if (d->slotUsed[slot]) {
      HIPCHK(hipStreamWaitEvent(cs, d->evSlotFree[slot], 0));
    }
  }
#endif
  if (onDevice) {
    P.emissions = emissions;
  } else {
    if (d->emis[slot].ensure(sizeof(float) * (size_t)std::max<int64_t>(maxEnd, 1), st, false)) {
      return fail(FLTX_ERR_OOM, "emissions staging allocation failed");
    }
    if (maxEnd > 0 && devCopyH2D(d->emis[slot].p, emissions, sizeof(float) * (size_t)maxEnd, cs)) {
      return fail(FLTX_ERR_HIP, "emissions upload failed");
    }
    P.emissions = d->emis[slot].as<float>();
  }
  d->lastEmis = P.emissions;
  if (devCopyH2D(d->emOff[slot].p, offs.data(), sizeof(int64_t) * B, cs) ||
      devCopyH2D(d->stepT[slot].p, T, sizeof(int32_t) * B, cs)) {
    return fail(FLTX_ERR_HIP, "batch descriptor upload failed");
  }
  d->upSlot = slot;
  P.emOff = d->emOff[slot].as<int64_t>();
  P.stepT = d->stepT[slot].as<int32_t>();
#ifndef FLTX_EMU
  /* the H2D copies above read pageable host memory that only lives for this call (offs); when the copy stream is
   * drained they have been consumed -- and are in place for the launch that follows on the other stream */
  if (devSync(cs)) {
    return fail(FLTX_ERR_HIP, "stream synchronize failed");
  }
#endif
  return FLTX_OK;
}

int settleStream(fltx_decoder* d);
} // namespace
static int offlineAttempts(fltx_decoder* d, const float* emissions, int32_t onDevice, int firstAttempt);
namespace {

int syncResults(fltx_decoder* d) {
  if (!d->settling && (d->chunkPending || d->pendingPrune >= 0)) {
    int rc = settleStream(d);
    if (rc) {
      return rc;
    }
  }
  if (d->offlinePending) { /* "defer_check": the look at the batch's statuses, and the second pass of what the fast path flagged */
    d->offlinePending = false;
    d->upSlot = d->offUpSlot;
    int rc = offlineAttempts(d, d->offEmis, 1, 1);
    if (rc) {
      return rc;
    }
  }
  if (d->resultsSynced) {
    return FLTX_OK;
  }
  Stream st = d->ctx->stream;
  const int B = d->B;
  d->hN.resize(B);
  d->hFrame.resize(B);
  d->hStatus.resize(B);
#ifdef FLTX_EMU
  if (devCopyD2H(d->hN.data(), d->uttNBeam.p, 4 * (size_t)B, st) ||
      devCopyD2H(d->hFrame.data(), d->uttFrame.p, 4 * (size_t)B, st) ||
      devCopyD2H(d->hStatus.data(), d->uttStatus.p, 4 * (size_t)B, st)) {
    return fail(FLTX_ERR_HIP, "result copy failed: %s", devErr());
  }
#else
  /* three small arrays, one wait: pinned staging, the copies queued back to back (a stream chunk pays this
   * after every step and every prune) */
  if (d->hSync.ensure(12 * (size_t)B)) {
    return fail(FLTX_ERR_OOM, "pinned staging allocation failed");
  }
  int32_t* h = (int32_t*)d->hSync.p;
  if (hipMemcpyAsync(h, d->uttNBeam.p, 4 * (size_t)B, hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipMemcpyAsync(h + B, d->uttFrame.p, 4 * (size_t)B, hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipMemcpyAsync(h + 2 * B, d->uttStatus.p, 4 * (size_t)B, hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipStreamSynchronize(st) != hipSuccess) {
    return fail(FLTX_ERR_HIP, "result copy failed: %s", devErr());
  }
  memcpy(d->hN.data(), h, 4 * (size_t)B);
  memcpy(d->hFrame.data(), h + B, 4 * (size_t)B);
  memcpy(d->hStatus.data(), h + 2 * B, 4 * (size_t)B);
#endif
  d->resultsSynced = true;
  return FLTX_OK;
}

int launchBacktrace(fltx_decoder* d) {
  Stream st = d->ctx->stream;
  if (d->tokens.ensure(4 * (size_t)std::max<int64_t>(d->histRecords, 1), st, false)) {
    return fail(FLTX_ERR_OOM, "token buffer allocation failed");
  }
  if (d->kind == FLTX_DECODER_LEXICON &&
      d->words.ensure(4 * (size_t)std::max<int64_t>(d->histRecords, 1), st, false)) {
    return fail(FLTX_ERR_OOM, "word buffer allocation failed");
  }
  BacktraceParams Q;
  memset(&Q, 0, sizeof(Q));
  Q.K = d->opt.beam_size;
  Q.kind = d->kind;
  Q.histPT = d->histPT.as<int2>();
  Q.histW = d->histW.as<int32_t>();
  Q.histOff = d->histOffD.as<int64_t>();
  Q.uttFrame = d->uttFrame.as<int32_t>();
  Q.uttNBeam = d->uttNBeam.as<int32_t>();
  Q.tokOff = d->histOffD.as<int64_t>();
  Q.tokens = d->tokens.as<int32_t>();
  Q.words = d->kind == FLTX_DECODER_LEXICON ? d->words.as<int32_t>() : nullptr;
  Q.nbest = 0;
  /* frames per LDS chunk: two record buffers in ({parent, token} 8 B, word 4 B) + token and word tiles out */
#ifdef FLTX_EMU
  const int btThreads = std::min(512, std::max(64, (d->opt.beam_size + 63) / 64 * 64)); /* (the emitting-model pass is one thread per hypothesis) */
#else
  const int btThreads = 512;
#endif
  const size_t perFrame = (size_t)Q.K * (2 * (8 + (d->kind == FLTX_DECODER_LEXICON ? 4 : 0)) + 8);
  /* 140 KB: leaves a CU room for a 16 KB decode workgroup of the next batch beside a back-trace workgroup */
  size_t btBudget = 0;
  int F = 0;
  auto chunkFor = [&](size_t budget) {
    btBudget = budget;
    F = (int)std::min<size_t>(btBudget / perFrame, 512);
    if (d->batchPacked) { /* the emission rows, transitions and addends of a chunk share the same LDS (amLds below) */
      /* (fltx_wlane.h's token sets: the tokens of the paths only, emissions and transitions read where needed) */
      const size_t fixed = d->batchWlane ? 16 : 4 * ((d->opt.criterion == FLTX_CRITERION_ASG && d->nTrans) ? (size_t)d->N * d->N : 0) + 16;
      const size_t perAm = 4 * ((d->batchWlane ? 0 : (size_t)d->N) + (size_t)Q.K);
      const size_t room = btBudget > fixed ? btBudget - fixed : 0;
      F = (int)std::min<size_t>((size_t)F, room / perAm);
    }
    if (F < 8 || Q.K > 4 * btThreads) {
      F = 0;
    }
  };
  chunkFor((size_t)(d->btLdsKb > 0 ? std::min(d->btLdsKb, 144) : 140) * 1024);
  if (F == 0 && d->btLdsKb > 0) {
    chunkFor((size_t)144 * 1024); /* (the caller's budget holds no chunk of this beam: the tunable yields, the decode does not fail) */
  }
  if (d->batchPacked && F == 0) {
    return fail(FLTX_ERR_UNSUPPORTED, "back-trace: packed history records need an LDS chunk (K=%d N=%d)", Q.K, d->N);
  }
  Q.F = F;
  size_t btLds = F > 0 ? (size_t)F * perFrame + 16 : 16;
  if (d->batchPacked && F > 0) { /* packed records; emitting-model scores re-accumulated along the paths */
    Q.packed = d->packedBits;
    Q.packedTokMask = d->batchWlane ? 0x7FFFFFFF : 0xFF;
    Q.amGather = d->batchWlane ? 1 : 0;
    Q.uttStatus = d->uttStatus.as<int32_t>();
    Q.amOut = d->outScores.as<double>();
    Q.emissions = d->lastEmis;
    Q.emOff = d->emOff[d->upSlot].as<int64_t>();
    Q.N = d->N;
    Q.transitions = (d->opt.criterion == FLTX_CRITERION_ASG && d->nTrans) ? d->transitions.as<float>() : nullptr;
    Q.tokLm = d->batchTlane ? d->tokLm : nullptr;
    Q.tokLmStride = d->N + 1;
    Q.blank = d->opt.criterion == FLTX_CRITERION_CTC ? d->blank : -1;
    const size_t amLds = Q.amGather ? 4 * (size_t)Q.K * F + 16
                                    : 4 * ((size_t)F * d->N + (Q.transitions ? (size_t)d->N * d->N : 0) + (size_t)Q.K * F) + 16;
    btLds = std::max(btLds, amLds);
  }
#ifdef FLTX_EMU
  const BacktraceParams* qq = &Q;
  emuLaunch(d->B, btThreads, btLds, [qq](char* smem) { backtraceUtterance(*qq, smem); });
#else
  HIPCHK(hipFuncSetAttribute((const void*)fltx_backtrace_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)btLds));
  hipLaunchKernelGGL(fltx_backtrace_kernel, dim3(d->B), dim3(btThreads), btLds, st, Q);
  HIPCHK(hipGetLastError());
  if (d->ev[2]) {
    HIPCHK(hipEventRecord(d->ev[2], st));
    d->timed = true;
  }
#endif
  d->backtraced = true;
  return FLTX_OK;
}

/* SURVEY.md section 8(d), per utterance: decode phase = T x (4N + 8K (+ 4*K*Kt + 8K + 4K for the
 * lexicon decoder)) + 16 * order per n-gram LM query the kernel issued; epilogue = 16 * H * (T + 2)
 * for the H hypotheses actually returned.  Needs the result counts: evaluated when asked for. */
int accountBytes(fltx_decoder* d) {
  if (d->statsDone) {
    return FLTX_OK;
  }
  const int64_t K = d->opt.beam_size, N = d->N;
  const int64_t Kt = std::min<int64_t>(d->opt.beam_size_token, N);
  int rc = syncResults(d);
  if (rc) {
    return rc;
  }
  std::vector<uint32_t> scored(d->B, 0u);
  if (d->lm->kind == 1 && d->scored.p &&
      devCopyD2H(scored.data(), d->scored.p, 4 * (size_t)d->B, d->ctx->stream)) {
    return fail(FLTX_ERR_HIP, "counter copy failed");
  }
  int64_t frames = 0, dec = 0, epi = 0, lm = 0;
  for (int b = 0; b < d->B; ++b) {
    const int64_t T = d->T[b];
    frames += T;
    int64_t per = 4 * N + 8 * K;
    if (d->kind == FLTX_DECODER_LEXICON) {
      per += 4 * K * Kt + 8 * K + 4 * K;
    }
    dec += per * T;
    lm += 16 * (int64_t)d->lm->order * (int64_t)scored[b];
    epi += 16 * (int64_t)d->hN[b] * (T + 2);
  }
  d->statFrames = frames;
  d->statDecodeBytes = dec + lm;
  d->statLmBytes = lm;
  d->statEpilogueBytes = epi;
  d->statBytes = dec + lm + epi;
  d->statsDone = true;
  return FLTX_OK;
}

/* save (dir 0) / put back (dir 1; `map`: device list of n utterances, null = all) the parked beams */
int streamSnapshot(fltx_decoder* d, int dir, const int32_t* map, int n) {
  SnapParams Q;
  memset(&Q, 0, sizeof(Q));
  Q.K = d->opt.beam_size;
  Q.dir = dir;
  Q.map = map;
  Q.gScore = d->gScore.as<double>();
  Q.gAm = d->gAm.as<double>();
  Q.gLm = d->gLm.as<double>();
  Q.gLexMax = d->gLexMax.as<float>();
  Q.gState = d->gState.as<uint32_t>();
  Q.gSPar = d->gSPar.as<uint32_t>();
  Q.gLex = d->gLex.as<uint32_t>();
  Q.gTokPb = d->gTokPb.as<uint32_t>();
  Q.gSEdge = d->gSEdge.as<int32_t>();
  Q.uttNBeam = d->uttNBeam.as<int32_t>();
  Q.uttFrame = d->uttFrame.as<int32_t>();
  Q.uttTotal = d->uttTotal.as<int32_t>();
  Q.uttStatus = d->uttStatus.as<int32_t>();
  Q.snap = d->snap.as<char>();
  Q.B = d->B;
  if (n <= 0) {
    return FLTX_OK;
  }
#ifdef FLTX_EMU
  for (int i = 0; i < n; ++i) {
    snapUtterance(Q, map ? map[i] : i, 0, 1);
  }
#else
  hipLaunchKernelGGL(fltx_snapshot_kernel, dim3(n), dim3(64), 0, d->ctx->stream, Q);
  HIPCHK(hipGetLastError());
#endif
  return FLTX_OK;
}

/* after an optimistic stream chunk: the streams whose chunk overflowed a candidate list (or whose cut left
 * fewer than K groups) get their beam back and decode the chunk again on the general path */
int streamRedoFlagged(fltx_decoder* d) {
  d->resultsSynced = false;
  int rc = FLTX_OK;
  /* the chunk's kernel left every stream's status in pinned host memory: wait for the kernel, nothing to copy */
  if (devSync(d->ctx->stream)) {
    return fail(FLTX_ERR_HIP, "stream synchronize failed: %s", devErr());
  }
  const int32_t* status = (const int32_t*)d->hStat.p;
  std::vector<int32_t> again;
  for (int b = 0; b < d->B; ++b) {
    if (status[b] & (ST_CAND_OVERFLOW | ST_CUT_RETRY)) {
      again.push_back(b);
    }
  }
  d->resultsSynced = false;
  if (again.empty()) {
    return FLTX_OK;
  }
  Stream st = d->ctx->stream;
  if (d->uttMap.ensure(4 * again.size(), st, false) ||
      devCopyH2D(d->uttMap.p, again.data(), 4 * again.size(), st) || devSync(st)) {
    return fail(FLTX_ERR_OOM, "re-run list upload failed");
  }
  if ((rc = streamSnapshot(d, 1, d->uttMap.as<int32_t>(), (int)again.size()))) {
    return rc;
  }
  const int savedWs = d->forceGlobalWs, savedCut = d->noCut;
  std::vector<int32_t> Tm(d->B, d->maxFrames);
  d->forceGlobalWs = 1;
  d->noCut = 1;
  d->keepScored = true;
  rc = prepare(d, d->B, d->N, Tm.data(), true);
  if (!rc) {
    DecodeParams P;
    fillParams(d, P);
    P.emissions = d->lastEmis;
    P.doBegin = 0;
    P.doEnd = 0;
    P.uttMap = d->uttMap.as<int32_t>();
    d->nLaunch = (int)again.size();
    rc = launchDecode(d, P);
    d->nLaunch = 0;
  }
  d->forceGlobalWs = savedWs;
  d->noCut = savedCut;
  if (!rc) {
    rc = prepare(d, d->B, d->N, Tm.data(), false); /* back to the optimistic geometry for the next chunk */
  }
  d->keepScored = false;
  d->streamRedone += (int)again.size();
  return rc;
}

int launchStreamOp(fltx_decoder* d, int op, int lookBack, int cap) {
  StreamOpParams Q;
  memset(&Q, 0, sizeof(Q));
  Q.K = d->opt.beam_size;
  Q.kind = d->kind;
  Q.op = op;
  Q.lookBack = lookBack;
  Q.histPT = d->histPT.as<int2>();
  Q.histW = d->histW.as<int32_t>();
  Q.histS = d->keepScores ? d->histS.as<double>() : nullptr;
  Q.histOff = d->histOffD.as<int64_t>();
  Q.uttFrame = d->uttFrame.as<int32_t>();
  Q.uttNBeam = d->uttNBeam.as<int32_t>();
  Q.gScore = d->gScore.as<double>();
  Q.outLen = d->bestLen.as<int32_t>();
  Q.outScores = d->bestScores.as<double>();
  Q.outTok = d->bestTok.as<int32_t>();
  Q.outWrd = d->bestWrd.as<int32_t>();
  Q.cap = cap;
#ifdef FLTX_EMU
  const StreamOpParams* qq = &Q;
  emuLaunch(d->B, 256, kStreamOpLds, [qq](char* sm) { streamOpUtterance(*qq, (int32_t*)sm); });
#else
  /* (sixteen waves: they stage the rows the ancestor walk visits, and prune moves up to lookBack + 101 history rows
   * -- a thread's loads and stores follow each other, so the fewer rounds the better) */
  hipLaunchKernelGGL(fltx_streamop_kernel, dim3(d->B), dim3(1024), 0, d->ctx->stream, Q);
  HIPCHK(hipGetLastError());
#endif
  return FLTX_OK;
}

/* the look at the chunk left pending by fltx_stream_step (decode it again where a list overflowed), then the prune that
 * was asked for meanwhile */
int settleStream(fltx_decoder* d) {
  int rc = FLTX_OK;
  struct Scope {
    bool& flag;
    ~Scope() { flag = false; }
  } scope{d->settling};
  d->settling = true;
  if (d->chunkPending) {
    d->chunkPending = false;
    const float* curEmis = d->lastEmis;
    const int curSlot = d->upSlot;
    d->lastEmis = d->pendEmis; /* (the next chunk may be uploaded already: the other slot) */
    d->upSlot = d->pendSlot;
    const int before = d->streamRedone;
    rc = streamRedoFlagged(d);
#ifndef FLTX_EMU
    if (!rc && d->streamRedone != before && d->copyStream) {
      HIPCHK(hipEventRecord(d->evSlotFree[d->pendSlot], d->ctx->stream)); /* the second pass read the slot too */
    }
#endif
    d->lastEmis = curEmis;
    d->upSlot = curSlot;
  }
  if (!rc && d->pendingPrune >= 0) {
    const int lb = d->pendingPrune;
    d->pendingPrune = -1;
    rc = launchStreamOp(d, 1, lb, 0);
    d->resultsSynced = false;
  }
  return rc;
}

/* streams: mode 0 = a new stream, mode 1 = renumber the LM-state ids the beam can still meet to the front.  Runs on the launch
 * stream between two decode launches; nothing is waited for. */
int launchCompact(fltx_decoder* d, int mode) {
  if (!d->recycle) {
    return FLTX_OK;
  }
  int rc;
  if (mode == 1 && d->idFamily == 1 && (rc = bumpEpoch(d))) { /* (the rebuilt table's epoch; empties the LM score cache too) */
    return rc;
  }
  CompactParams Q;
  memset(&Q, 0, sizeof(Q));
  Q.K = d->opt.beam_size;
  Q.N = d->N;
  Q.mode = mode;
  Q.family = d->idFamily;
  Q.idCap = d->idCap;
  Q.uttNBeam = d->uttNBeam.as<int32_t>();
  Q.gState = d->gState.as<uint32_t>();
  Q.gSPar = d->gSPar.as<uint32_t>();
  Q.uttNextId = d->uttNextId.as<int32_t>();
  Q.newId = d->idNew.as<uint32_t>();
  Q.list = d->idList.as<uint32_t>();
  if (d->lm->kind == 1 && d->idFamily == 1) {
    Q.stateCtx = d->stateCtx.as<int32_t>();
    Q.ctxL = std::max(1, d->lm->order - 1);
  }
  Q.idPar = d->idPar.as<uint32_t>();
  Q.idEdge = d->idEdge.as<int32_t>();
  Q.idBorn = d->idBorn.as<uint32_t>();
  Q.keep = d->idKeep.as<uint8_t>();
  Q.childTab = d->childTab.as<uint32_t>();
  Q.maskTab = d->maskTab.as<unsigned long long>();
  Q.gMask = d->gMask.as<unsigned long long>();
  Q.stateTab = d->stateTab.as<unsigned long long>();
  Q.stateVal = d->stateVal.as<uint32_t>();
  Q.stateCap = d->stateCap;
  Q.epoch = d->epoch;
#ifdef FLTX_EMU
  const CompactParams* qq = &Q;
  emuLaunch(d->B, 64, kCompactLds, [qq](char* sm) { compactStates(*qq, sm); });
#else
  HIPCHK(hipFuncSetAttribute((const void*)fltx_compact_states_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                             kCompactLds));
  hipLaunchKernelGGL(fltx_compact_states_kernel, dim3(d->B), dim3(1024), kCompactLds, d->ctx->stream, Q);
  HIPCHK(hipGetLastError());
#endif
  d->idsUsedBound = mode == 0 ? 1 : 0;
  d->compactions += mode;
  return FLTX_OK;
}

/* ---- host LM (fltx_lm_host_create): one frame per launch, the frame's LM questions answered on the host ------- */
int hostLmEnsure(fltx_decoder* d) {
  const int64_t K = d->opt.beam_size, N = d->N, B = d->B;
  const int64_t nTok = std::min<int64_t>(d->opt.beam_size_token, N);
  /* what a frame can ask: one question per (hypothesis, token); the lexicon decoder with a word LM one per word a
   * child ends (<= 6, Trie.h:19) or the unknown word; decodeEnd one per hypothesis */
  int64_t cap = K * nTok * ((d->kind == FLTX_DECODER_LEXICON && !d->isLmToken) ? 6 : 1);
  cap = std::max<int64_t>(cap, K);
  if (cap * B * 8 > (1ll << 32)) {
    return fail(FLTX_ERR_UNSUPPORTED, "host LM: %lld questions per frame x %lld utterances exceed the question buffer",
                (long long)cap, (long long)B);
  }
  d->hlmQCap = (int)cap;
  if (d->hlmQCount.ensure(4 * (size_t)B) || d->hlmQ.ensure(8 * (size_t)B * (size_t)cap) ||
      d->hlmBeamN.ensure(4 * (size_t)B) || d->hlmBeam.ensure(4 * (size_t)B * (size_t)K)) {
    return fail(FLTX_ERR_OOM, "host LM: pinned buffers");
  }
  return FLTX_OK;
}

int hostLmCall(int rc, const char* what) {
  return rc ? fail(FLTX_ERR_CALLBACK, "host LM: the %s callback reported failure (%d)", what, rc) : FLTX_OK;
}

/* list the questions of frame `frame` of the chunk (or of decodeEnd), have the user's LM answer each distinct one
 * and leave the answers where the frame's launch finds them (P.hlmTab / P.hlmDir) */
int hostLmExchange(fltx_decoder* d, DecodeParams& P, int frame, bool end) {
  Stream st = d->ctx->stream;
  const int B = d->B, K = d->opt.beam_size;
  const fltx_host_lm& cb = d->lm->host;
  P.hlmFrame = frame;
  P.hlmEnd = end ? 1 : 0;
  const size_t qLds = 4 * (size_t)(std::min(d->opt.beam_size_token, d->N) + 4);
#ifdef FLTX_EMU
  {
    const DecodeParams* pp = &P;
    emuLaunch(B, 64, qLds, [pp](char* smem) { hostLmQuestions(*pp, smem); }); /* (one wave: host threads are expensive) */
  }
#else
  HIPCHK(hipFuncSetAttribute((const void*)fltx_hostlm_questions_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)qLds));
  hipLaunchKernelGGL(fltx_hostlm_questions_kernel, dim3(B), dim3(256), qLds, st, P);
  HIPCHK(hipGetLastError());
#endif
  if (devSync(st)) { /* (also: the previous frame's table upload has been consumed, the staging buffer is free) */
    return fail(FLTX_ERR_HIP, "host LM: questions kernel failed: %s", devErr());
  }
  const int32_t* qCount = (const int32_t*)d->hlmQCount.p;
  const uint2* q = (const uint2*)d->hlmQ.p;
  const int32_t* beamN = (const int32_t*)d->hlmBeamN.p;
  const int32_t* beam = (const int32_t*)d->hlmBeam.p;
  int rc;
  if (d->hlmAnnounce && cb.update_cache) { /* updateLMCache(lm_, hyp_[t + 1]) of the frame before (Utils.h:346-354) */
    for (int b = 0; b < B; ++b) {
      if (beamN[b] > 0 && (rc = hostLmCall(cb.update_cache(cb.user, b, beamN[b], beam + (size_t)b * K), "update_cache"))) {
        return rc;
      }
    }
  }
  /* one open-addressing table per utterance, at most half full; the tables double as the set of distinct questions */
  std::vector<uint2> dir((size_t)B);
  size_t slots = 0;
  for (int b = 0; b < B; ++b) {
    if (qCount[b] > d->hlmQCap) {
      return fail(FLTX_ERR_UNSUPPORTED, "host LM: utterance %d asked %d questions in one frame (capacity %d)", b, qCount[b],
                  d->hlmQCap);
    }
    const uint32_t n = nextPow2(std::max<uint64_t>(2ull * (uint64_t)qCount[b], 2));
    dir[(size_t)b] = make_uint2((uint32_t)slots, n - 1u);
    slots += n;
  }
  const size_t dirBytes = alignUp(sizeof(uint2) * (size_t)B, 16);
  const size_t bytes = dirBytes + 16 * slots;
  if (d->hlmStage.ensure(bytes) || d->hlmTabD.ensure(bytes, st, false)) {
    return fail(FLTX_ERR_OOM, "host LM: answer table");
  }
  char* stage = (char*)d->hlmStage.p;
  memcpy(stage, dir.data(), sizeof(uint2) * (size_t)B);
  uint4* tab = (uint4*)(stage + dirBytes);
  memset(tab, 0xFF, 16 * slots);
  std::vector<int32_t> qu, qs, qi, os;
  std::vector<float> of;
  std::vector<size_t> where;
  for (int b = 0; b < B; ++b) {
    const uint2 dd = dir[(size_t)b];
    const uint2* qb = q + (size_t)b * d->hlmQCap;
    d->hlmAsked += qCount[b];
    for (int i = 0; i < qCount[b]; ++i) {
      const uint32_t sid = qb[i].x, idx = qb[i].y;
      uint32_t s = hashKey(sid, idx, 0x7f4a7c15u, 0) & dd.y;
      for (;;) {
        uint4& e = tab[(size_t)dd.x + s];
        if (e.x == sid && e.y == idx) {
          break;
        }
        if (e.x == kEmpty && e.y == kEmpty) {
          e.x = sid;
          e.y = idx;
          qu.push_back(b);
          qs.push_back((int32_t)sid);
          qi.push_back((int32_t)idx);
          where.push_back((size_t)dd.x + s);
          break;
        }
        s = (s + 1) & dd.y;
      }
    }
  }
  const size_t nq = qu.size();
  d->hlmDistinct += (int64_t)nq;
  if (nq > 0) {
    os.assign(nq, 0);
    of.assign(nq, 0.0f);
    if ((rc = hostLmCall(cb.score(cb.user, (int32_t)nq, qu.data(), qs.data(), qi.data(), os.data(), of.data()), "score"))) {
      return rc;
    }
    for (size_t i = 0; i < nq; ++i) {
      uint32_t bits;
      memcpy(&bits, &of[i], 4);
      tab[where[i]].z = (uint32_t)os[i];
      tab[where[i]].w = bits;
    }
  }
  if (devCopyH2D(d->hlmTabD.p, stage, bytes, st)) {
    return fail(FLTX_ERR_HIP, "host LM: answer upload failed");
  }
  P.hlmDir = (const uint2*)d->hlmTabD.p;
  P.hlmTab = (const uint4*)((const char*)d->hlmTabD.p + dirBytes);
  return FLTX_OK;
}

/* the frames of one chunk (T[b] frames of utterance b), a launch each */
int hostLmFrames(fltx_decoder* d, DecodeParams& P, const int32_t* T) {
  int maxT = 0;
  for (int b = 0; b < d->B; ++b) {
    maxT = std::max(maxT, T[b]);
  }
  P.doBegin = 0;
  P.doEnd = 0;
  for (int t = 0; t < maxT; ++t) {
    int rc = hostLmExchange(d, P, t, false);
    if (rc || (rc = launchDecode(d, P))) {
      return rc;
    }
    d->hlmAnnounce = true;
  }
  return FLTX_OK;
}

/* decodeBegin on the device (the seed hypothesis in LM state 0) and LM::start on the host */
int hostLmBegin(fltx_decoder* d, DecodeParams& P) {
  int rc = hostLmEnsure(d);
  if (rc) {
    return rc;
  }
  fillParams(d, P); /* (the question buffers exist now) */
  d->hlmAnnounce = false;
  d->hlmAsked = 0;
  d->hlmDistinct = 0;
  return hostLmCall(d->lm->host.start(d->lm->host.user, d->B), "start");
}

/* Decoder::prune: the LM states the beam still holds (the callee may release the others) */
int hostLmRetain(fltx_decoder* d) {
  const fltx_host_lm& cb = d->lm->host;
  if (!cb.retain) {
    return FLTX_OK;
  }
  const int B = d->B, K = d->opt.beam_size;
  std::vector<int32_t> nb((size_t)B), st((size_t)B * K);
  if (devCopyD2H(nb.data(), d->uttNBeam.p, 4 * (size_t)B, d->ctx->stream) ||
      devCopyD2H(st.data(), d->gState.p, 4 * (size_t)B * K, d->ctx->stream)) {
    return fail(FLTX_ERR_HIP, "host LM: beam copy failed");
  }
  for (int b = 0; b < B; ++b) {
    int rc = hostLmCall(cb.retain(cb.user, b, nb[(size_t)b], st.data() + (size_t)b * K), "retain");
    if (rc) {
      return rc;
    }
  }
  return FLTX_OK;
}

/* fltx_decode_batch with a host LM: begin, a launch per frame, end, back-trace */
int hostLmDecodeBatch(fltx_decoder* d, const float* emissions, int32_t onDevice, const int64_t* offsets, const int32_t* T,
                      int32_t B, int32_t N) {
  d->offlineCall = true;
  d->batchPacked = false;
  d->batchWlane = false;
  d->batchTlane = false;
  d->keepScores = d->userKeepScores;
  d->fallbackReasons = 0;
  int rc = prepare(d, B, N, T, true);
  if (rc) {
    return rc;
  }
  latchFirst(d);
  if ((rc = bumpEpoch(d))) {
    return rc;
  }
  DecodeParams P;
  fillParams(d, P);
  if ((rc = hostLmBegin(d, P)) || (rc = uploadStep(d, emissions, onDevice, offsets, T, P))) {
    return rc;
  }
  d->nLaunch = 0;
  P.doBegin = 1;
  P.doEnd = 0;
  P.hlmFrame = 1 << 30; /* (no frame in this launch: decodeBegin only) */
  if ((rc = launchDecode(d, P)) || (rc = hostLmFrames(d, P, T))) {
    return rc;
  }
  if ((rc = hostLmExchange(d, P, 1 << 30, true))) {
    return rc;
  }
  P.doBegin = 0;
  P.doEnd = 1;
  P.hlmFrame = 1 << 30;
  if ((rc = launchDecode(d, P))) {
    return rc;
  }
  d->lastRedo = 0;
  d->resultsSynced = false;
  if ((rc = launchBacktrace(d))) {
    return rc;
  }
  d->T.assign(T, T + B);
  d->statsDone = false;
  d->haveResults = true;
  d->ended = true;
  return FLTX_OK;
}

} // namespace

extern "C" {

int fltx_decode_batch(fltx_decoder* d, const float* emissions, int32_t onDevice, const int64_t* offsets,
                      const int32_t* T, int32_t B, int32_t N) {
  DeviceScope devScope(d ? d->ctx : nullptr);
  if (devScope.failed) {
    return fail(FLTX_ERR_HIP, "hipSetDevice failed");
  }
  if (!d || !T || B <= 0 || N <= 0) {
    return fail(FLTX_ERR_INVALID, "fltx_decode_batch: bad argument");
  }
  if (d->offlinePending) {
    /* "defer_check": the batch before this one was never read.  Its statuses are looked at all the same (and what the
     * fast path flagged is decoded again: that is how a decoder learns that a fallback should stick) -- a caller who
     * only wants kernels queued back to back asks for defer_check = 2 and gets the drops counted */
    if (d->deferCheck == 2) {
      d->offlinePending = false;
      ++d->looksDropped;
    } else {
      int rc = syncResults(d);
      if (rc) {
        return rc;
      }
      d->unreadRedone += d->lastRedo;
    }
  }
  d->streaming = false;
  d->chunkPending = false;
  d->pendingPrune = -1;
  d->haveResults = false;
  d->resultsSynced = false;
  d->backtraced = false;
  d->hostFetched = false;
  d->compactFetched = false;
  d->scoresFetched = false;
  if (d->lm->kind == 2) {
    return hostLmDecodeBatch(d, emissions, onDevice, offsets, T, B, N);
  }
  d->offT.assign(T, T + B);
  d->offOffsets.clear();
  if (offsets) {
    d->offOffsets.assign(offsets, offsets + B);
  }
  d->offN = N;
  d->offlinePending = false;
  return offlineAttempts(d, emissions, onDevice, 0);
}

/* The attempts of one offline batch.  firstAttempt 0: the call itself (the fast path); 1: the look at a batch whose
 * status scan was deferred (fltx_decoder_set "defer_check"): the decode kernel and the back-trace were launched and
 * the call returned -- the scan, and the second pass of whatever the fast path flagged, happen when the results are
 * first asked for (syncResults), on the emissions where the first pass left them in HBM. */
} /* extern "C" */
static int settlePendingLook(fltx_decoder* d) { return syncResults(d); }
static int offlineAttempts(fltx_decoder* d, const float* emissions, int32_t onDevice, int firstAttempt) {
  const int32_t* T = d->offT.data();
  const int64_t* offsets = d->offOffsets.empty() ? nullptr : d->offOffsets.data();
  const int32_t B = (int32_t)d->offT.size(), N = d->offN;
  /* The optimistic fast paths (LDS-sized candidate lists of the lexicon
   * decoder, the score cut, the one-pass histogram select of the lean / lane
   * steps) flag the rare utterance they cannot serve.  Only those utterances
   * are decoded again, on the general path (HBM workspace / no cut / generic
   * engine); the others keep their results.  The fallback sticks to the decoder
   * only when a large part of the batch needed it. */
  std::vector<int32_t> redoList;
  size_t firstRedo = 0;
  if (firstAttempt == 0) {
    d->fallbackReasons = 0;
  }
  bool recomputeRetry = false; /* this attempt is the recompute form of the cut-off generation */
  bool recomputeTried = false;
  const int savedGlobalWs = d->forceGlobalWs, savedNoCut = d->noCut, savedNoLean = d->noLean, savedNoSlim = d->noSlim;
  const int savedNoSlane = d->noSlane, savedNoXlane = d->noXlane, savedNoYlane = d->noYlane, savedNoWlane = d->noWlane;
  d->offlineCall = true;
  if (firstAttempt == 0) {
    d->batchPacked = false;
    d->batchWlane = false;
    d->batchTlane = false;
  }
  d->keepScores = d->userKeepScores; /* a stream on this decoder had switched the score history on */
  for (int attempt = firstAttempt; attempt < 3 + firstAttempt; ++attempt) {
    const bool finalForm = attempt > firstAttempt && !recomputeRetry; /* the general path: nothing left to fall back to */
    const bool look = firstAttempt > 0 && attempt == firstAttempt; /* the deferred look at what is already decoded */
    int rc;
    if (!look) {
      d->keepScored = attempt > 0;
      rc = prepare(d, B, N, T, finalForm);
      d->keepScored = false;
      if (rc) {
        return rc;
      }
    }
    if (!look && attempt > 0 && d->lm->kind == 1 && d->scored.p) {
      for (int32_t b : redoList) { /* only the utterances decoded again start their query count over */
        devMemset(d->scored.as<uint32_t>() + b, 0, 4, d->ctx->stream);
      }
    }
    if (!look) {
    d->batchPacked = d->batchPacked || d->slane || d->xlane || d->ylane;
    d->batchTlane = d->batchTlane || d->tlane != 0;
    if (d->slane || d->xlane || d->ylane) {
      d->packedBits = (d->slane && d->mlaneNG > 1) ? 10 : (d->ylane == 4 ? 13 : 8); /* (a re-run on a general engine leaves plain records) */
      d->batchWlane = d->wlane != 0; /* ... and so is the records' token width (the back-trace reads it, not d->wlane) */
    }
    if (attempt == 0) {
      latchFirst(d);
    }
    if ((rc = bumpEpoch(d))) {
      return rc;
    }
    DecodeParams P;
    fillParams(d, P);
    if ((rc = uploadStep(d, emissions, onDevice, offsets, T, P))) {
      return rc;
    }
    P.doBegin = 1;
    P.doEnd = 1;
    d->nLaunch = 0;
    if (attempt > 0) {
      if (d->uttMap.ensure(4 * redoList.size(), d->ctx->stream, false) ||
          devCopyH2D(d->uttMap.p, redoList.data(), 4 * redoList.size(), d->ctx->stream)) {
        return fail(FLTX_ERR_OOM, "re-run list upload failed");
      }
      P.uttMap = d->uttMap.as<int32_t>();
      d->nLaunch = (int)redoList.size();
    }
    rc = launchDecode(d, P);
    d->nLaunch = 0;
    if (rc) {
      return rc;
    }
    } /* !look */
    const bool cutMode = d->CAP2 > 0 || d->cutRecompute;
    const bool needLook = !finalForm && ((d->kind == FLTX_DECODER_LEXICON && (d->wsInLds || cutMode)) || d->lean || d->wlane || d->xlane || d->ylane || d->tlane);
    if (needLook && attempt == 0 && d->deferCheck) {
      /* the caller keeps the emissions where they are until the results are read: launch the back-trace and return;
       * the look happens in syncResults() */
      d->offlinePending = true;
      d->offEmis = d->lastEmis;
      d->offUpSlot = d->upSlot;
      break;
    }
    if (needLook) {
      d->resultsSynced = false;
      d->offlinePending = false; /* (syncResults below must not come back here) */
      if ((rc = syncResults(d))) {
        return rc;
      }
      bool ws = false, cut = false, lean = false, slaneMiss = false, xlaneMiss = false, wlaneMiss = false;
      const bool slimMode = d->CAP2 > 0;
      std::vector<int32_t> again;
      const bool firstScan = attempt == 0 || look;
      const int nScan = firstScan ? B : (int)redoList.size();
      for (int i = 0; i < nScan; ++i) {
        const int b = firstScan ? i : redoList[i];
        const int st = d->hStatus[b];
        if (firstScan && (st & ST_SELECT_FALLBACK)) {
          d->fallbackReasons |= 1ll << ((st >> 8) & 31); /* (the lexicon lane engine says why: fltx_ylane.h) */
        }
        const bool o = (st & ST_CAND_OVERFLOW) && d->kind == FLTX_DECODER_LEXICON;
        const bool c = (st & ST_CUT_RETRY) && cutMode;
        const bool l = (st & ST_SELECT_FALLBACK) && (d->lean || d->tlane);
        const bool x = (st & ST_SELECT_FALLBACK) && (d->xlane || d->ylane);
        const bool wl = (st & ST_SELECT_FALLBACK) && d->wlane; /* fltx_wlane.h: a row without a defined token beam, a select that did not converge */
        if (o || c || l || x || wl) {
          again.push_back(b);
          ws |= o;
          cut |= c;
          lean |= l;
          slaneMiss |= l && d->slane;
          xlaneMiss |= x;
          wlaneMiss |= wl;
        }
      }
      redoList.swap(again);
      if (firstScan) {
        firstRedo = redoList.size();
      }
      if (!redoList.empty()) {
        recomputeRetry = false;
        if (ws && slimMode && !cut && !recomputeTried) { /* the slim list overflowed: generate twice instead */
          d->noSlim = 1;
          ws = false;
          recomputeRetry = true;
          recomputeTried = true;
        } else if (!d->wsInLds && cutMode) { /* HBM workspace with the cut: the plain HBM path is what is left */
          ws = true;
          cut = true;
        }
        d->forceGlobalWs = ws ? 1 : d->forceGlobalWs;
        d->noCut = cut ? 1 : d->noCut;
        d->noLean = lean ? 1 : d->noLean;
        d->noSlane = slaneMiss ? 1 : d->noSlane; /* (re-run on the generic engine) */
        d->noWlane = wlaneMiss ? 1 : d->noWlane;
        d->noXlane = xlaneMiss ? 1 : d->noXlane;
        d->noYlane = xlaneMiss ? 1 : d->noYlane;
        d->resultsSynced = false;
        continue;
      }
    }
    break;
  }
  if (!d->offlinePending) {
    d->lastRedo = (int)firstRedo;
  }
  if (firstAttempt > 0 && firstRedo == 0) {
    return FLTX_OK; /* the deferred look found nothing to decode again: the back-trace already ran */
  }
  if (firstRedo > 0 && firstRedo * 4 <= (size_t)B) { /* a few outliers: next batch tries the fast path again */
    d->forceGlobalWs = savedGlobalWs;
    d->noCut = savedNoCut;
    d->noLean = savedNoLean;
    d->noSlim = savedNoSlim;
    d->noSlane = savedNoSlane;
    d->noWlane = savedNoWlane;
    d->noXlane = savedNoXlane;
    d->noYlane = savedNoYlane;
  }
  d->resultsSynced = false;
  int rc = launchBacktrace(d);
  if (rc) {
    return rc;
  }
  d->T.assign(T, T + B);
  d->statsDone = false;
  d->haveResults = true;
  d->ended = true;
  return FLTX_OK;
}
extern "C" {

int fltx_stream_begin(fltx_decoder* d, int32_t B, int32_t N, int32_t maxFrames) {
  DeviceScope devScope(d ? d->ctx : nullptr);
  if (devScope.failed) {
    return fail(FLTX_ERR_HIP, "hipSetDevice failed");
  }
  if (!d || B <= 0 || N <= 0 || maxFrames < 0) {
    return fail(FLTX_ERR_INVALID, "fltx_stream_begin: bad argument");
  }
  /* prune(lookBack) of the lexicon decoder walks on from lookBack frames back to the last COMPLETE hypothesis (its parent
   * ended a word), up to kLookBackLimit frames further (Utils.h:28,293-308): that many frames can stay in the buffer
   * whatever the caller prunes.  The reference's buffer grows as needed; here the stream's buffer is sized when it
   * begins, so a lexicon stream gets those kLookBackLimit frames ON TOP of max_frames: a caller whose own count (lookBack
   * frames left after a prune, plus the chunks since) stays within max_frames never fills it (round 5 raised
   * "exceed max_frames" in the middle of such a stream) */
  if (d->kind == FLTX_DECODER_LEXICON) {
    maxFrames += kLookBackLimit;
  }
  std::vector<int32_t> Tm(B, maxFrames);
  d->offlineCall = false;
  d->offlinePending = false; /* (an offline batch whose results were never read) */
  d->batchPacked = false;
  d->keepScores = 1; /* streams serve getBestHypothesis(lookBack) of ancestors */
  /* The lexicon decoder's candidate lists are sized for what a frame usually produces (LDS) when a chunk
   * can be decoded again: the beam a chunk starts from is saved, the state table is idempotent (same
   * (parent, edge) -> same id), history rows are rewritten in place.  (The lexicon-free engines never
   * overflow and number their states with a counter: always one pass.) */
  d->streamOpt = d->kind == FLTX_DECODER_LEXICON && d->userStreamOpt != 0 && !d->forceGlobalWs && d->lm->kind != 2;
  d->streamRedone = 0;
  int rc = prepare(d, B, N, Tm.data(), !d->streamOpt);
  if (rc) {
    return rc;
  }
  if (d->streamOpt && (d->lean || !(d->wsInLds || d->CAP2 > 0 || d->cutRecompute))) {
    d->streamOpt = false; /* nothing optimistic came out of it */
    if ((rc = prepare(d, B, N, Tm.data(), true))) {
      return rc;
    }
  }
  if (d->streamOpt &&
      d->snap.ensure((size_t)B * d->opt.beam_size * (3 * 8 + 6 * 4) + (size_t)B * 16, d->ctx->stream, false)) {
    return fail(FLTX_ERR_OOM, "stream snapshot allocation failed");
  }
  if (d->streamOpt && d->hStat.ensure(4 * (size_t)B)) {
    return fail(FLTX_ERR_OOM, "pinned status allocation failed");
  }
  latchFirst(d);
  if ((rc = bumpEpoch(d))) {
    return rc;
  }
  d->maxFrames = maxFrames;
  d->streaming = true;
  d->ended = false;
  d->haveResults = false;
  d->resultsSynced = false;
  d->backtraced = false;
  d->hostFetched = false;
  d->compactFetched = false;
  d->scoresFetched = false;
  d->frames.assign(B, 0);
  d->framesExact = true;
  d->chunkPending = false;
  d->pendingPrune = -1;
  d->compactions = 0;
  if ((rc = launchCompact(d, 0))) {
    return rc;
  }
  DecodeParams P;
  fillParams(d, P);
  if (d->lm->kind == 2 && (rc = hostLmBegin(d, P))) {
    return rc;
  }
  std::vector<int32_t> zeroT(B, 0);
  if ((rc = uploadStep(d, nullptr, 1, nullptr, zeroT.data(), P))) {
    return rc;
  }
  P.doBegin = 1;
  P.doEnd = 0;
  P.hlmFrame = 1 << 30;
  if ((rc = launchDecode(d, P))) {
    return rc;
  }
  d->haveResults = true;
  return FLTX_OK;
}

int fltx_stream_step(fltx_decoder* d, const float* emissions, int32_t onDevice, const int64_t* offsets,
                     const int32_t* T) {
  DeviceScope devScope(d ? d->ctx : nullptr);
  if (devScope.failed) {
    return fail(FLTX_ERR_HIP, "hipSetDevice failed");
  }
  if (!d || !T) {
    return fail(FLTX_ERR_INVALID, "fltx_stream_step: bad argument");
  }
  if (!d->streaming || d->ended) {
    return fail(FLTX_ERR_STATE, "fltx_stream_step: call fltx_stream_begin first");
  }
  for (int pass = 0; pass < 2; ++pass) {
    int bad = -1;
    for (int b = 0; b < d->B && bad < 0; ++b) {
      if (T[b] < 0 || d->frames[b] + T[b] > d->maxFrames) {
        bad = b;
      }
    }
    if (bad < 0) {
      break;
    }
    if (pass == 0 && !d->framesExact && T[bad] >= 0) {
      int rcs = syncResults(d); /* what the prunes since the last look really left in the buffers */
      if (rcs) {
        return rcs;
      }
      for (int b = 0; b < d->B; ++b) {
        d->frames[b] = d->hFrame[b];
      }
      d->framesExact = true;
      continue;
    }
    return fail(FLTX_ERR_RANGE, "stream %d: %d buffered + %d new frames exceed max_frames %d%s", bad, d->frames[bad],
                T[bad], d->maxFrames,
                d->kind == FLTX_DECODER_LEXICON
                    ? " (a lexicon stream's prune keeps the frames back to the last complete word, up to lookBack + 100: "
                      "size the buffer for 100 + lookBack + the largest chunk, Utils.h:28,293-308)"
                    : "");
  }
  /* the chunk's upload first (copy stream: under the kernel of the chunk before, if that one is still pending), then
   * the look at the chunk before, then this chunk's launch */
  DecodeParams up;
  memset(&up, 0, sizeof(up));
  int rc = uploadStep(d, emissions, onDevice, offsets, T, up);
  if (rc) {
    return rc;
  }
  if ((rc = settleStream(d))) {
    return rc;
  }
  if (d->recycle) {
    /* a frame makes at most beam new LM states per stream: when this chunk could run a stream out of ids, the ids
     * nothing can meet again are given back first (beam x max_frames ids are free after that, or the chunk's status
     * says the table is full) */
    int maxT = 0;
    for (int b = 0; b < d->B; ++b) {
      maxT = std::max(maxT, T[b]);
    }
    const int64_t want = (int64_t)d->opt.beam_size * maxT * (d->streamOpt ? 2 : 1); /* (a chunk decoded again may name states twice) */
    const int64_t room = (d->compactions ? d->idsFreeBound : d->idCap) - d->idsUsedBound;
    if ((want > room || d->userCompactAlways) && (rc = launchCompact(d, 1))) {
      return rc;
    }
    d->idsUsedBound += want;
  }
  DecodeParams P;
  fillParams(d, P); /* (emOff / stepT: the slot uploadStep has just filled) */
  P.emissions = up.emissions;
  P.doBegin = 0;
  P.doEnd = 0;
  if (d->streamOpt && (rc = streamSnapshot(d, 0, nullptr, d->B))) {
    return rc;
  }
  d->sstreamLaunch = d->sstream != 0;
  rc = d->lm->kind == 2 ? hostLmFrames(d, P, T) : launchDecode(d, P);
  d->sstreamLaunch = false;
  if (rc) {
    return rc;
  }
  if (d->streamOpt) {
    d->chunkPending = true;
    d->pendEmis = P.emissions;
    d->pendSlot = d->upSlot;
    /* A chunk handed over as a device pointer is the caller's buffer: a second pass deferred to the next call would
     * read whatever the caller has put there since (one buffer reused chunk after chunk is legal).  Such a chunk is
     * settled before the call returns; host chunks live in this decoder's own staging slots and may wait. */
    if ((!d->deferRedo || onDevice) && (rc = settleStream(d))) {
      return rc;
    }
  }
  for (int b = 0; b < d->B; ++b) {
    d->frames[b] += T[b];
  }
  d->resultsSynced = false;
  d->backtraced = false;
  d->hostFetched = false;
  d->compactFetched = false;
  d->scoresFetched = false;
  return FLTX_OK;
}

int fltx_stream_end(fltx_decoder* d) {
  DeviceScope devScope(d ? d->ctx : nullptr);
  if (devScope.failed) {
    return fail(FLTX_ERR_HIP, "hipSetDevice failed");
  }
  if (!d) {
    return fail(FLTX_ERR_INVALID, "null decoder");
  }
  if (!d->streaming || d->ended) {
    return fail(FLTX_ERR_STATE, "fltx_stream_end: no open stream");
  }
  int rc = settleStream(d);
  if (rc) {
    return rc;
  }
  DecodeParams P;
  fillParams(d, P);
  std::vector<int32_t> zeroT(d->B, 0);
  if ((rc = uploadStep(d, nullptr, 1, nullptr, zeroT.data(), P))) {
    return rc;
  }
  if (d->lm->kind == 2 && (rc = hostLmExchange(d, P, 1 << 30, true))) {
    return rc;
  }
  P.doBegin = 0;
  P.doEnd = 1;
  P.hlmFrame = 1 << 30;
  if ((rc = launchDecode(d, P))) {
    return rc;
  }
  d->ended = true;
  d->resultsSynced = false;
  d->backtraced = false;
  d->hostFetched = false;
  d->compactFetched = false;
  d->scoresFetched = false;
  return FLTX_OK;
}

int fltx_stream_prune(fltx_decoder* d, int32_t lookBack) {
  DeviceScope devScope(d ? d->ctx : nullptr);
  if (devScope.failed) {
    return fail(FLTX_ERR_HIP, "hipSetDevice failed");
  }
  if (!d || lookBack < 0) {
    return fail(FLTX_ERR_INVALID, "fltx_stream_prune: bad argument");
  }
  if (!d->streaming) {
    return fail(FLTX_ERR_STATE, "fltx_stream_prune: call fltx_stream_begin first");
  }
  int rc = FLTX_OK;
  if (d->chunkPending && d->pendingPrune < 0) {
    d->pendingPrune = lookBack; /* runs right after the look at the pending chunk (settleStream) */
  } else if ((rc = settleStream(d)) || (rc = launchStreamOp(d, 1, lookBack, 0))) {
    return rc;
  }
  if (d->lm->kind == 2 && (rc = hostLmRetain(d))) {
    return rc;
  }
  d->resultsSynced = false;
  d->backtraced = false;
  d->hostFetched = false;
  d->compactFetched = false;
  d->scoresFetched = false;
  if (d->kind == FLTX_DECODER_LEXFREE) {
    /* every hypothesis of the lexicon-free decoder is complete (LexiconFreeDecoder.h:84-86): findBestAncestor stops
     * exactly lookBack frames back, so what stays buffered is known without asking the device (no wait per chunk;
     * fltx_stream_frames_in_buffer still reads the device's count) */
    for (int b = 0; b < d->B; ++b) {
      if (d->frames[b] - lookBack >= 1) {
        d->frames[b] = lookBack;
      }
    }
    return FLTX_OK;
  }
  /* the lexicon decoder prunes back to a complete hypothesis (LexiconDecoder.h:97-99), or not at all: what stays buffered
   * is the device's to say.  No wait here: frames[] stays the upper bound "nothing pruned" and fltx_stream_step asks
   * the device only when that bound would not fit max_frames */
  d->framesExact = false;
  return FLTX_OK;
}

static int checkStatus(fltx_decoder* d, int b);

int fltx_stream_frames_in_buffer(fltx_decoder* d, int32_t b, int32_t* n) {
  DeviceScope devScope(d ? d->ctx : nullptr); /* (syncResults may run the deferred second pass of a chunk) */
  if (devScope.failed) {
    return fail(FLTX_ERR_HIP, "hipSetDevice failed");
  }
  if (!d || !n || b < 0 || b >= d->B) {
    return fail(FLTX_ERR_INVALID, "bad argument");
  }
  int rc = syncResults(d);
  if (rc) {
    return rc;
  }
  *n = d->hFrame[b] + 1;
  return FLTX_OK;
}

static int checkStatus(fltx_decoder* d, int b) {
  int s = d->hStatus[b];
  if (s & ST_CAND_OVERFLOW) {
    return fail(FLTX_ERR_UNSUPPORTED, "utterance %d: candidate buffer overflow (CAP=%d)", b, d->CAP);
  }
  if (s & ST_TABLE_FULL) {
    return fail(FLTX_ERR_UNSUPPORTED, "utterance %d: LM-state table full (cap=%u)", b, d->stateCap);
  }
  if (s & ST_SELECT_FALLBACK) {
    return fail(FLTX_ERR_UNSUPPORTED, "utterance %d: top-K select did not converge (non-finite scores?)", b);
  }
  if (s & ST_HLM_MISS) {
    return fail(FLTX_ERR_STATE, "utterance %d: host LM: a frame asked a question that had not been listed (internal error)", b);
  }
  return FLTX_OK;
}

int fltx_result_count(fltx_decoder* d, int32_t b, int32_t* nHyp, int32_t* length) {
  DeviceScope devScope(d ? d->ctx : nullptr);
  if (devScope.failed) {
    return fail(FLTX_ERR_HIP, "hipSetDevice failed");
  }
  if (!d || b < 0 || b >= d->B) {
    return fail(FLTX_ERR_INVALID, "fltx_result_count: bad argument");
  }
  if (!d->haveResults) {
    return fail(FLTX_ERR_STATE, "no decode has been run");
  }
  int rc = syncResults(d);
  if (rc) {
    return rc;
  }
  if ((rc = checkStatus(d, b))) {
    return rc;
  }
  int ff = d->hFrame[b];
  int n = d->hN[b];
  /* LexiconDecoder.cpp:276-280: nothing before the first frame */
  if (d->kind == FLTX_DECODER_LEXICON && ff < 1) {
    n = 0;
  }
  if (nHyp) {
    *nHyp = n;
  }
  if (length) {
    *length = ff + 1;
  }
  return FLTX_OK;
}

int fltx_result_fetch(fltx_decoder* d, int32_t b, int32_t maxHyp, double* scores, int32_t* tokens,
                      int32_t* words, int32_t* nCopied) {
  int32_t n = 0, len = 0;
  int rc = fltx_result_count(d, b, &n, &len);
  if (rc) {
    return rc;
  }
  n = std::min(n, maxHyp);
  if (nCopied) {
    *nCopied = n;
  }
  if (n <= 0) {
    return FLTX_OK;
  }
  Stream st = d->ctx->stream;
  const int K = d->opt.beam_size;
  if (scores) {
    /* beam slots are score-sorted; after decodeEnd they are the final n-best */
    std::vector<double> sc((size_t)n), am((size_t)n), lm((size_t)n);
    if (d->ended) {
      if (devCopyD2H(scores, d->outScores.as<double>() + (size_t)b * K * 3, 8 * 3 * (size_t)n, st)) {
        return fail(FLTX_ERR_HIP, "score copy failed");
      }
    } else {
      if (devCopyD2H(sc.data(), d->gScore.as<double>() + (size_t)b * K, 8 * (size_t)n, st) ||
          devCopyD2H(am.data(), d->gAm.as<double>() + (size_t)b * K, 8 * (size_t)n, st) ||
          devCopyD2H(lm.data(), d->gLm.as<double>() + (size_t)b * K, 8 * (size_t)n, st)) {
        return fail(FLTX_ERR_HIP, "score copy failed");
      }
      for (int i = 0; i < n; ++i) {
        scores[3 * i] = sc[i];
        scores[3 * i + 1] = am[i];
        scores[3 * i + 2] = lm[i];
      }
    }
  }
  if (tokens || words) {
    if (!d->backtraced && (rc = launchBacktrace(d))) {
      return rc;
    }
    const int64_t base = d->histOff[b];
    if (tokens && devCopyD2H(tokens, d->tokens.as<int32_t>() + base, 4 * (size_t)n * len, st)) {
      return fail(FLTX_ERR_HIP, "token copy failed");
    }
    if (words) {
      if (d->kind == FLTX_DECODER_LEXICON) {
        if (devCopyD2H(words, d->words.as<int32_t>() + base, 4 * (size_t)n * len, st)) {
          return fail(FLTX_ERR_HIP, "word copy failed");
        }
      } else {
        std::fill(words, words + (size_t)n * len, -1); /* LexiconFreeDecoder.h:80-82 */
      }
    }
  }
  return FLTX_OK;
}

int fltx_result_fetch_batch(fltx_decoder* d, const int32_t** nHyp, const int32_t** length, const double** scores,
                            const int32_t** tokens, const int32_t** words, const int64_t** offsets) {
  DeviceScope devScope(d ? d->ctx : nullptr);
  if (devScope.failed) {
    return fail(FLTX_ERR_HIP, "hipSetDevice failed");
  }
  if (!d) {
    return fail(FLTX_ERR_INVALID, "fltx_result_fetch_batch: null decoder");
  }
  if (!d->haveResults || !d->ended) {
    return fail(FLTX_ERR_STATE, "fltx_result_fetch_batch: no finished decode (use fltx_result_fetch on a stream)");
  }
  int rc = syncResults(d);
  if (rc) {
    return rc;
  }
  const int B = d->B, K = d->opt.beam_size;
  for (int b = 0; b < B; ++b) {
    if ((rc = checkStatus(d, b))) {
      return rc;
    }
  }
  if (!d->hostFetched) {
    if (!d->backtraced && (rc = launchBacktrace(d))) {
      return rc;
    }
    Stream st = d->ctx->stream;
    const size_t nRec = (size_t)std::max<int64_t>(d->histRecords, 1);
    if (d->hTokens.ensure(4 * nRec) || d->hScores.ensure(8 * 3 * (size_t)B * K) ||
        (d->kind == FLTX_DECODER_LEXICON && d->hWords.ensure(4 * nRec))) {
      return fail(FLTX_ERR_OOM, "pinned result buffers: allocation failed");
    }
    if (devCopyD2H(d->hScores.p, d->outScores.p, 8 * 3 * (size_t)B * K, st) ||
        devCopyD2H(d->hTokens.p, d->tokens.p, 4 * nRec, st) ||
        (d->kind == FLTX_DECODER_LEXICON && devCopyD2H(d->hWords.p, d->words.p, 4 * nRec, st))) {
      return fail(FLTX_ERR_HIP, "result copy failed: %s", devErr());
    }
    d->hLen.resize(B);
    d->hNHyp.resize(B);
    for (int b = 0; b < B; ++b) {
      d->hLen[b] = d->hFrame[b] + 1;
      /* LexiconDecoder.cpp:276-280: nothing before the first frame */
      d->hNHyp[b] = (d->kind == FLTX_DECODER_LEXICON && d->hFrame[b] < 1) ? 0 : d->hN[b];
    }
    d->hostFetched = true;
  }
  if (nHyp) {
    *nHyp = d->hNHyp.data();
  }
  if (length) {
    *length = d->hLen.data();
  }
  if (scores) {
    *scores = (const double*)d->hScores.p;
  }
  if (tokens) {
    *tokens = (const int32_t*)d->hTokens.p;
  }
  if (words) {
    *words = d->kind == FLTX_DECODER_LEXICON ? (const int32_t*)d->hWords.p : nullptr;
  }
  if (offsets) {
    *offsets = d->histOff.data();
  }
  return FLTX_OK;
}

int fltx_result_fetch_batch_compact(fltx_decoder* d, const int32_t** nHyp, const int32_t** length,
                                    const double** scores, const uint8_t** tokensU8, const int32_t** words,
                                    const int64_t** offsets) {
  DeviceScope devScope(d ? d->ctx : nullptr);
  if (devScope.failed) {
    return fail(FLTX_ERR_HIP, "hipSetDevice failed");
  }
  if (!d) {
    return fail(FLTX_ERR_INVALID, "fltx_result_fetch_batch_compact: null decoder");
  }
  if (!d->haveResults || !d->ended) {
    return fail(FLTX_ERR_STATE, "fltx_result_fetch_batch_compact: no finished decode");
  }
  if (d->N >= 255) {
    return fail(FLTX_ERR_UNSUPPORTED, "fltx_result_fetch_batch_compact: %d tokens do not fit a byte", d->N);
  }
  int rc = syncResults(d);
  if (rc) {
    return rc;
  }
  const int B = d->B, K = d->opt.beam_size;
  for (int b = 0; b < B; ++b) {
    if ((rc = checkStatus(d, b))) {
      return rc;
    }
  }
  if (!d->compactFetched) {
    if (!d->backtraced && (rc = launchBacktrace(d))) {
      return rc;
    }
    Stream st = d->ctx->stream;
    const bool lex = d->kind == FLTX_DECODER_LEXICON;
    d->hLen.resize(B);
    d->hNHyp.resize(B);
    d->packOff.resize((size_t)B + 1);
    /* meta (one upload): dstOff[B] int64, count[B] int32 */
    if (d->hPackMeta.ensure(12 * (size_t)B) || d->dPackMeta.ensure(12 * (size_t)B, st, false)) {
      return fail(FLTX_ERR_OOM, "compact results: allocation failed");
    }
    int64_t* hOff = (int64_t*)d->hPackMeta.p;
    int32_t* hCnt = (int32_t*)(hOff + B);
    int64_t total = 0;
    for (int b = 0; b < B; ++b) {
      d->hLen[b] = d->hFrame[b] + 1;
      d->hNHyp[b] = (lex && d->hFrame[b] < 1) ? 0 : d->hN[b]; /* LexiconDecoder.cpp:276-280 */
      d->packOff[b] = total;
      hOff[b] = total;
      hCnt[b] = d->hNHyp[b] * d->hLen[b];
      total += hCnt[b];
    }
    d->packOff[B] = total;
    const size_t tot = (size_t)std::max<int64_t>(total, 1);
    if (d->dTok8.ensure(tot, st, false) || d->hTok8.ensure(tot) || d->hScores.ensure(8 * 3 * (size_t)B * K) ||
        (lex && (d->dWordsC.ensure(4 * tot, st, false) || d->hWordsC.ensure(4 * tot)))) {
      return fail(FLTX_ERR_OOM, "compact results: allocation failed");
    }
    if (devCopyH2D(d->dPackMeta.p, d->hPackMeta.p, 12 * (size_t)B, st)) {
      return fail(FLTX_ERR_HIP, "compact results: upload failed");
    }
#ifdef FLTX_EMU
    for (int b = 0; b < B; ++b) {
      const int32_t* src = d->tokens.as<int32_t>() + d->histOff[b];
      for (int i = 0; i < hCnt[b]; ++i) {
        d->dTok8.as<uint8_t>()[hOff[b] + i] = src[i] < 0 ? (uint8_t)0xFF : (uint8_t)src[i];
        if (lex) {
          d->dWordsC.as<int32_t>()[hOff[b] + i] = d->words.as<int32_t>()[d->histOff[b] + i];
        }
      }
    }
#else
    PackParams Q;
    Q.tokens = d->tokens.as<int32_t>();
    Q.words = lex ? d->words.as<int32_t>() : nullptr;
    Q.srcOff = d->histOffD.as<int64_t>();
    Q.dstOff = d->dPackMeta.as<int64_t>();
    Q.count = (const int32_t*)(d->dPackMeta.as<int64_t>() + B);
    Q.tok8 = d->dTok8.as<uint8_t>();
    Q.wordsOut = lex ? d->dWordsC.as<int32_t>() : nullptr;
    hipLaunchKernelGGL(fltx_pack_results_kernel, dim3(B), dim3(256), 0, st, Q);
    HIPCHK(hipGetLastError());
#endif
    if ((!d->scoresFetched && devCopyD2H(d->hScores.p, d->outScores.p, 8 * 3 * (size_t)B * K, st)) ||
        (total > 0 && devCopyD2H(d->hTok8.p, d->dTok8.p, (size_t)total, st)) ||
        (lex && total > 0 && devCopyD2H(d->hWordsC.p, d->dWordsC.p, 4 * (size_t)total, st))) {
      return fail(FLTX_ERR_HIP, "result copy failed: %s", devErr());
    }
    d->scoresFetched = true;
    d->compactFetched = true;
  }
  if (nHyp) {
    *nHyp = d->hNHyp.data();
  }
  if (length) {
    *length = d->hLen.data();
  }
  if (scores) {
    *scores = (const double*)d->hScores.p;
  }
  if (tokensU8) {
    *tokensU8 = (const uint8_t*)d->hTok8.p;
  }
  if (words) {
    *words = d->kind == FLTX_DECODER_LEXICON ? (const int32_t*)d->hWordsC.p : nullptr;
  }
  if (offsets) {
    *offsets = d->packOff.data();
  }
  return FLTX_OK;
}

int fltx_result_best(fltx_decoder* d, int32_t b, int32_t lookBack, double* scores, int32_t* tokens,
                     int32_t* words, int32_t capacity, int32_t* length) {
  DeviceScope devScope(d ? d->ctx : nullptr);
  if (devScope.failed) {
    return fail(FLTX_ERR_HIP, "hipSetDevice failed");
  }
  if (!d || b < 0 || b >= d->B || lookBack < 0 || !length) {
    return fail(FLTX_ERR_INVALID, "fltx_result_best: bad argument");
  }
  if (!d->haveResults) {
    return fail(FLTX_ERR_STATE, "no decode has been run");
  }
  if (!d->keepScores) {
    return fail(FLTX_ERR_STATE,
                "getBestHypothesis needs the per-frame score history: use the streaming calls or "
                "fltx_decoder_set(dec, \"keep_scores\", 1) before decoding");
  }
  int rc = syncResults(d);
  if (rc) {
    return rc;
  }
  if ((rc = checkStatus(d, b))) {
    return rc;
  }
  int maxLen = 0;
  for (int i = 0; i < d->B; ++i) {
    maxLen = std::max(maxLen, d->hFrame[i] + 1);
  }
  Stream st = d->ctx->stream;
  if (d->bestLen.ensure(4 * (size_t)d->B, st, true) || d->bestScores.ensure(24 * (size_t)d->B, st, true) ||
      d->bestTok.ensure(4 * (size_t)d->B * maxLen, st, false) ||
      d->bestWrd.ensure(4 * (size_t)d->B * maxLen, st, false)) {
    return fail(FLTX_ERR_OOM, "best-hypothesis buffers: allocation failed");
  }
  if ((rc = launchStreamOp(d, 0, lookBack, maxLen))) {
    return rc;
  }
  int32_t len = 0;
  if (devCopyD2H(&len, d->bestLen.as<int32_t>() + b, 4, st)) {
    return fail(FLTX_ERR_HIP, "copy failed");
  }
  *length = len;
  if (len == 0) {
    return FLTX_OK; /* empty DecodeResult */
  }
  if (len > capacity) {
    return fail(FLTX_ERR_RANGE, "fltx_result_best: capacity %d < length %d", capacity, len);
  }
  if (scores && devCopyD2H(scores, d->bestScores.as<double>() + 3 * (size_t)b, 24, st)) {
    return fail(FLTX_ERR_HIP, "copy failed");
  }
  if (tokens && devCopyD2H(tokens, d->bestTok.as<int32_t>() + (size_t)b * maxLen, 4 * (size_t)len, st)) {
    return fail(FLTX_ERR_HIP, "copy failed");
  }
  if (words) {
    if (d->kind == FLTX_DECODER_LEXICON) {
      if (devCopyD2H(words, d->bestWrd.as<int32_t>() + (size_t)b * maxLen, 4 * (size_t)len, st)) {
        return fail(FLTX_ERR_HIP, "copy failed");
      }
    } else {
      std::fill(words, words + len, -1);
    }
  }
  return FLTX_OK;
}

int fltx_result_device(fltx_decoder* d, const int32_t** nHyp, const double** scores,
                       const int32_t** tokens, const int32_t** words, const int64_t** tokOff) {
  DeviceScope devScope(d ? d->ctx : nullptr);
  if (devScope.failed) {
    return fail(FLTX_ERR_HIP, "hipSetDevice failed");
  }
  if (!d || !d->haveResults) {
    return fail(FLTX_ERR_STATE, "no decode has been run");
  }
  if (d->offlinePending) { /* "defer_check": the look at the statuses (and the second pass) before anybody reads the results */
    int rc = syncResults(d);
    if (rc) {
      return rc;
    }
  }
  if (!d->backtraced) {
    int rc = launchBacktrace(d);
    if (rc) {
      return rc;
    }
  }
  if (nHyp) {
    *nHyp = d->outN.as<int32_t>();
  }
  if (scores) {
    *scores = d->outScores.as<double>();
  }
  if (tokens) {
    *tokens = d->tokens.as<int32_t>();
  }
  if (words) {
    *words = d->kind == FLTX_DECODER_LEXICON ? d->words.as<int32_t>() : nullptr;
  }
  if (tokOff) {
    *tokOff = d->histOffD.as<int64_t>();
  }
  return FLTX_OK;
}

/* phase profile of the last launch: out[8] = shader clocks summed over the
 * utterances of the batch (0 prep, 1 generate, 2 fold, 3 select, 4 build,
 * 5 row hand-over) */
int fltx_decoder_profile(fltx_decoder* d, uint64_t* out) {
  DeviceScope devScope(d ? d->ctx : nullptr);
  if (devScope.failed) {
    return fail(FLTX_ERR_HIP, "hipSetDevice failed");
  }
  if (!d || !out) {
    return fail(FLTX_ERR_INVALID, "null argument");
  }
  if (!d->profile || !d->prof.p) {
    return fail(FLTX_ERR_STATE, "profiling is off: fltx_decoder_set(dec, \"profile\", 1)");
  }
  std::vector<unsigned long long> h(8 * (size_t)d->B);
  if (devCopyD2H(h.data(), d->prof.p, 8 * h.size(), d->ctx->stream)) {
    return fail(FLTX_ERR_HIP, "profile copy failed");
  }
  for (int i = 0; i < 8; ++i) {
    out[i] = 0;
  }
  for (size_t i = 0; i < h.size(); ++i) {
    out[i & 7] += h[i];
  }
  return FLTX_OK;
}

int fltx_decoder_timing(fltx_decoder* d, float* decodeMs, float* backtraceMs) {
  DeviceScope devScope(d ? d->ctx : nullptr);
  if (devScope.failed) {
    return fail(FLTX_ERR_HIP, "hipSetDevice failed");
  }
  if (!d) {
    return fail(FLTX_ERR_INVALID, "null decoder");
  }
#ifdef FLTX_EMU
  if (decodeMs) {
    *decodeMs = 0;
  }
  if (backtraceMs) {
    *backtraceMs = 0;
  }
  return FLTX_OK;
#else
  if (!d->timed) {
    return fail(FLTX_ERR_STATE, "no timed decode_batch has been run");
  }
  HIPCHK(hipEventSynchronize(d->ev[2]));
  float a = 0, b = 0;
  HIPCHK(hipEventElapsedTime(&a, d->ev[0], d->ev[1]));
  HIPCHK(hipEventElapsedTime(&b, d->ev[1], d->ev[2]));
  if (decodeMs) {
    *decodeMs = a;
  }
  if (backtraceMs) {
    *backtraceMs = b;
  }
  return FLTX_OK;
#endif
}

int fltx_decoder_stats(fltx_decoder* d, int64_t* frames, int64_t* bytes, int32_t* threads,
                       int32_t* ldsBytes) {
  DeviceScope devScope(d ? d->ctx : nullptr);
  if (devScope.failed) {
    return fail(FLTX_ERR_HIP, "hipSetDevice failed");
  }
  if (!d) {
    return fail(FLTX_ERR_INVALID, "null decoder");
  }
  if (d->haveResults && d->ended && !d->streaming) {
    int rc = accountBytes(d);
    if (rc) {
      return rc;
    }
  }
  if (frames) {
    *frames = d->statFrames;
  }
  if (bytes) {
    *bytes = d->statBytes;
  }
  if (threads) {
    *threads = d->threads;
  }
  if (ldsBytes) {
    *ldsBytes = d->wsInLds ? (int32_t)d->wsBytes : 0;
  }
  return FLTX_OK;
}

int fltx_decoder_bytes(fltx_decoder* d, int64_t* decodeBytes, int64_t* epilogueBytes, int64_t* lmBytes) {
  DeviceScope devScope(d ? d->ctx : nullptr);
  if (devScope.failed) {
    return fail(FLTX_ERR_HIP, "hipSetDevice failed");
  }
  if (!d) {
    return fail(FLTX_ERR_INVALID, "null decoder");
  }
  if (!d->haveResults || !d->ended || d->streaming) {
    return fail(FLTX_ERR_STATE, "fltx_decoder_bytes: no finished offline decode");
  }
  int rc = accountBytes(d);
  if (rc) {
    return rc;
  }
  if (decodeBytes) {
    *decodeBytes = d->statDecodeBytes;
  }
  if (epilogueBytes) {
    *epilogueBytes = d->statEpilogueBytes;
  }
  if (lmBytes) {
    *lmBytes = d->statLmBytes;
  }
  return FLTX_OK;
}

} /* extern "C" */
