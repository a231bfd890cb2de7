/*
 * fltx_kernels.h -- the MI355X beam-search engine (device code).
 *
 * One workgroup (W = 64..1024 threads, wave64) owns one utterance for the
 * whole launch and walks its T frames serially; utterances are independent,
 * so a batch is B workgroups and a node is 8 devices x B/8 (no collective).
 * Per frame the workgroup restates, data-parallel, what the reference does
 * with std::vector / std::sort / shared_ptr tries:
 *
 *   reference (flashlight/lib/text/decoder/)        here
 *   ---------------------------------------------   -------------------------
 *   LexiconFreeDecoder::decodeStep  .cpp:30-125     genLexFree()
 *   LexiconDecoder::decodeStep      .cpp:32-229     genLexicon()
 *   candidatesAdd                   Utils.h:131-144 pushCandidate(): record in
 *                                                   LDS + insert into an LDS
 *                                                   hash keyed by the merge key
 *   candidatesStore step 1 (filter) Utils.h:160-165 foldGroups(): threshold on
 *   candidatesStore step 2 (merge)  Utils.h:167-198   members, max / ordered
 *                                                     log-add per hash chain
 *   candidatesStore step 3 (prune)  Utils.h:200-220 selectTopK(): exact
 *                                                   histogram select on f64
 *   candidatesStore step 4 (move)   Utils.h:222-224 buildBeam(): rank-sort the
 *                                                   survivors, write the next
 *                                                   beam + back-pointers
 *   LMState::child / compare        lm/LM.h:24-49   LM-state id = slot of
 *                                                   (parent id, edge) in a
 *                                                   per-utterance HBM hash
 *   Trie children.find / labels     Trie.h:30-55    flat child table gather
 *   decodeEnd                       .cpp:127-158 /  genEnd() + same machinery
 *                                   .cpp:231-274
 *   getAllHypothesis back-trace     Utils.h:229-266 fltx_backtrace_kernel
 *
 * Identity of LM states (the hard part): the reference merges on
 * the ADDRESS of a memoised LMState trie node.  Here a state is named by the
 * pair (parent state id, edge label); two candidates reach the same state iff
 * the pairs are equal, so merging never needs a table probe.  Only the <= K
 * survivors per frame that entered a new state resolve their id with one
 * lookup-or-insert in the utterance's HBM table, which memoises every state a
 * survivor ever held -- exactly the states whose identity can be observed.
 *
 * Scores are IEEE double, accumulated in the reference's order; the library is
 * built with -ffp-contract=off so `s + w * l` stays mul + add.
 *
 * No MFMA: the path is gather / scan / select.  The bound is HBM (emission
 * rows in, back-pointer records out); see DESIGN.md for the byte accounting.
 */
#pragma once
#include "fltx_rt.h"

namespace fltx {

/* ------------------------------------------------------------------------ */
/* data layout                                                               */
/* ------------------------------------------------------------------------ */
struct TrieNodeInfo { /* 16 B, one gather per visited trie node */
  float maxScore;   /* TrieNode::maxScore after the host smear (Trie.h:54) */
  int32_t labOff;   /* first label in trieLabels */
  int32_t nLabels;  /* TrieNode::labels.size() (<= 6, Trie.h:19) */
  int32_t nChildren;
};

/* One 16-byte record per (trie node, token): everything candidate generation
 * needs about the child reached by that token, so a hypothesis' whole child row
 * is one contiguous N*16-byte read (prefetched a frame ahead into LDS). */
struct TrieEdge {
  int32_t child;   /* child node id, -1 = no such child */
  float childMax;  /* child's TrieNode::maxScore (Trie.h:54) */
  int32_t label0;  /* child's first label, -1 if none */
  uint32_t meta;   /* nLabels:3 | hasChildren:1 | labOff:28 */
};

/* Breadth-first layout of the trie for fltx_xlane.h: the children of a node are contiguous and in
 * token order: child(n) = firstChild + popcount(childMask & ((1 << n) - 1)).  32 bytes. */
struct XNode {
  unsigned long long childMask; /* tokens that have a child */
  unsigned long long kidsMask;  /* ... whose child has children itself (LexiconDecoder.cpp:89-91) */
  uint32_t firstChild;
  int32_t endLabel0;            /* label of the child entered by the word-ending token, -1 = none */
  float maxScore;               /* TrieNode::maxScore (Trie.h:54) */
  uint32_t parent;              /* breadth-first id of the parent node (root: 0) */
};

struct NgramSlot { /* 16 B open-addressing slot: (context node, word) -> n-gram */
  uint32_t ctx;    /* node id of the context n-gram (0 = empty context) */
  uint32_t word;   /* LM word id; 0xFFFFFFFF = empty slot */
  uint32_t node;   /* node id of this n-gram */
  float prob;      /* log10 p */
};

/* histogram arrays are indexed with one pad word per 16 bins so that a lane
 * reading its 16 consecutive bins hits 16 different LDS banks */
#define FLTX_HB(b) ((b) + ((b) >> 4))

constexpr uint32_t kEmpty = 0xFFFFFFFFu;
constexpr uint32_t kNoParent = 0x00FFFFFFu; /* parent id of the root LM state */
constexpr uint32_t kNewState = 0x80000000u; /* cSrc flag: candidate enters a new LM state */
constexpr uint32_t kExtend = 0x40000000u;   /* cSrc flag: cAux holds the child's maxScore bits */
constexpr uint32_t kSrcMask = 0x3FFFFFFFu;
constexpr uint32_t kPrevBlank = 0x80000000u; /* tokPb flag */
constexpr int kFinishEdge = -1;              /* KenLM::finish child key (KenLM.cpp:79) */
constexpr uint32_t kNoParent32 = 0xFFFFFFFFu; /* idPar: no parent on record */
constexpr uint32_t kHostEdge = 0x00FFFFFEu;  /* host LM: a state is named by its id alone, key = (id, kHostEdge) */
constexpr uint32_t kPhantomNode = 0x80000000u; /* NgramSlot.node: navigation-only prefix */
constexpr int kMaxNgramOrder = 6;            /* FL_TEXT_KENLM_MAX_ORDER (lm/CMakeLists.txt:3) */

enum { ST_OK = 0, ST_CAND_OVERFLOW = 1, ST_TABLE_FULL = 2, ST_SELECT_FALLBACK = 4, ST_CUT_RETRY = 8,
       ST_PACKED = 16 /* not an error: the history holds the packed records of fltx_slane.h / fltx_xlane.h */,
       ST_HLM_MISS = 32 /* host LM: a frame asked a question hostLmQuestions() had not listed (a bug, never a user error) */ };

struct DecodeParams {
  /* options (LexiconDecoderOptions, LexiconDecoder.h:21-31) */
  int32_t K, Kt;
  double beamThreshold, lmWeight, wordScore, unkScore, silScore;
  int32_t logAdd, criterion;
  int32_t sil, blank, unk, isLmToken, kind, N;
  const float* transitions; /* [N*N] or null */
  /* flat trie */
  const TrieEdge* trieEdge; /* [nNodes*N] */
  const int32_t* trieLabels;
  const unsigned long long* trieMask; /* [nNodes] bit n set <=> the node has a child for token n (N <= 64), or null */
  int32_t itemCap;                    /* uint16 words of the (hypothesis, token) item list: K * min(Kt, N) items, 0 = unused */
  int32_t itemWide;                   /* 1: beams beyond 1 024 -- an item is a 32-bit word (two uint16 words of itemCap each) */
  /* LM */
  int32_t lmKind; /* 0 ZeroLM, 1 n-gram, 2 host LM (a user subclass of LM answers the frame's questions on the host) */
  int32_t lmOrder;
  const NgramSlot* ngTab;
  uint32_t ngMask;
  const float* ngBackoff;   /* [node] back-off weight, 0 when absent */
  const int32_t* usrToLm;
  int32_t nUsr;
  int32_t lmBos, lmEos, lmUnk;
  int32_t* stateCtx;        /* [B*stateCap*(lmOrder-1)] suffix node ids per LM state */
  /* batch */
  const float* emissions;
  const int64_t* emOff;
  const int32_t* stepT;
  const int32_t* uttMap; /* utterance handled by workgroup i (a partial re-run), or null: workgroup i = utterance i */
  int32_t doBegin, doEnd;
  /* persistent per-utterance state (HBM) */
  int32_t* uttNBeam;
  int32_t* uttFrame;  /* index of the newest frame in the buffer */
  int32_t* uttTotal;  /* nDecodedFrames_ (for the ASG first-frame rule) */
  int32_t* uttStatus;
  double* gScore;
  double* gAm;
  double* gLm;
  float* gLexMax;
  uint32_t* gState;
  uint32_t* gSPar;
  int32_t* gSEdge;
  uint32_t* gLex;
  uint32_t* gTokPb;
  /* history: one {parent, token} (+ word) record per surviving slot per frame */
  int2* histPT;
  int32_t* histW;
  double* histS;      /* optional {score, am, lm} per record (streaming: getBestHypothesis
                         of an ancestor reports the ancestor's scores, Utils.h:236-238) */
  const int64_t* histOff;
  /* LM-state identity table */
  unsigned long long* stateTab;
  uint32_t stateCap; /* per utterance, power of two */
  uint32_t epoch;    /* 1..65535, bumped per decodeBegin */
  /* workspace geometry */
  int32_t CAP, HS, NB, SCAP;
  int32_t CAP2;  /* >0: lexicon decoder scores all candidates into slim {score, order} records first and
                    materialises only the best cutM of them (see runFrame); capacity of that list */
  int32_t cutM;
  int32_t hotLevel;     /* HBM workspace: 1 = histogram / scalars in LDS, 2 = candidate records and merge hash too */
  int32_t cutRecompute; /* 1: cut-off generation without the slim list (count per bin, then generate again) */
  int32_t dense; /* 1: lexicon-free frames use the hash-free dense merge */
  int32_t lane;  /* >0: lane-per-slot frame step (fltx_lane.h), value = tokens per wave */
  char* gws;          /* global workspace (big configurations), or null */
  int64_t gwsStride;
  /* results of decodeEnd */
  int32_t* outN;
  double* outScores;
  /* lean path: LM-state ids are allocated per utterance from a counter; a
   * state's children are remembered in childTab[id*N + token] and the set of
   * tokens that already have a child in maskTab[id], so a state that dropped
   * out of the beam and is re-entered later gets its old id back without any
   * hash (lm/LM.h:24-34 memoisation).  Written fire-and-forget, read only on
   * that rare re-entry. */
  uint32_t* childTab;           /* [B*idCap*N] */
  unsigned long long* maskTab;  /* [B*idCap] */
  int64_t idCap;
  int32_t* uttNextId;
  unsigned long long* gMask;    /* [B*K] parked masks of the beam slots */
  uint32_t* scored;             /* [B] n-gram LM queries issued for utterance b (accounting), or null */
  const XNode* xnode;           /* breadth-first trie layout (fltx_xlane.h), or null */
  int32_t xEndTok;              /* the token every word ends with in that layout */
  const float* xdelta;          /* per node of that layout: maxScore - (parent is the root ? 0 : parent's maxScore) */
  const uint32_t* xextra;       /* ... the words its word-ending child carries: place of the first in trieLabels << 3 | words (fltx_ylane.h, several words per spelling), or null */
  int32_t yTpw;                 /* fltx_ylane.h: list positions per token wave */
  unsigned long long* ymemo;    /* fltx_ylane.h / fltx_xlane.h, memo in HBM: the LM-state memo of every utterance (ymemoSlots each) */
  uint32_t ymemoSlots;          /* power of two; fltx_xlane.h: kXlMemoH */
  int32_t tokRowBlocks;         /* fltx_tokbeam_kernel: workgroups per utterance (four rows each) */
  void* tokRows;                /* fltx_wlane.h: the token beams of all rows (WlTokRow records), row = histOff[b] / K + t */
  int32_t tune;                 /* development: experiment bits of the lane engines (fltx_decoder_set "tune"), 0 in production */
  int32_t yRankAt;              /* tests: (lane, token) pairs beyond which a token wave of fltx_ylane.h ranks its own (0 = its rounds' capacity) */
  unsigned long long* lmCache;  /* generic step with an n-gram LM: (LM state, word) -> score of the last look-ups, kLmCache slots per utterance */
  /* fltx_slane.h with a token-level n-gram LM (TL): the LM flattened to ONE dense table for gather -- row c = an n-gram
   * context the model can be in, column n < N = {float bits of lm.score(c, token n), context after it}, column N = lm.finish(c);
   * built on the host from the same flat tables ngScore walks (buildTokDense, fltx_api.cpp) */
  const int2* tokLm;
  int32_t tokLmStride;          /* N + 1 */
  int32_t tlEdgeSlots, tlMaskSlots; /* ... its LDS memos (TlaneLds::memo): slots of the edge and of the child-mask memo, powers of two */
  int32_t wsNoInv;              /* HBM workspace of the generic step (hot level >= 1): no L1 invalidate after its barriers (wsBarrier) */
  int32_t* statusHost;          /* optimistic stream chunks: uttStatus mirrored in pinned host memory (read after the kernel, no copy) */
  const int32_t* xlmword;       /* ... LM word id of the word a node's separator child carries (n-gram LM), or null */
  double yBound;                /* ... and the largest lmWeight x smearing difference of the lexicon (>= 0) */
  double yTransMax;             /* ASG: the largest transition score, at least 0 (upper bound of what a pair can gain) */
  /* host LM (lmKind == 2, fltx_lm_host_create): the search runs one frame per launch.  Before it, hostLmQuestions()
   * lists every (LM state, index) pair the frame will ask LM::score / LM::finish about into pinned host memory; the
   * host answers them (one call of the user's LM per distinct pair) and uploads the answers as one open-addressing
   * table per utterance, which the frame's lmAsk() reads.  LM states are numbered by the host (the ADDRESS of the
   * LMState object the user's LM returned, lm/LM.h:37-49), so candidates merge exactly where the reference's
   * pointer comparison merges them. */
  int32_t hlmFrame;             /* frame of this launch inside the call's chunk: utterance b takes part iff stepT[b] > hlmFrame */
  int32_t hlmEnd;               /* hostLmQuestions: list the LM::finish questions of decodeEnd instead of a frame's */
  int32_t hlmQCap;              /* questions per utterance the list holds */
  const uint4* hlmTab;          /* answers: {state, index, state the LM returned, score bits} */
  const uint2* hlmDir;          /* [B] {first slot, slots - 1 (a power of two minus one)} of utterance b's table */
  int32_t* hlmQCount;           /* [B] pinned host: questions listed (may exceed hlmQCap: the host then reports the overflow) */
  uint2* hlmQ;                  /* [B*hlmQCap] pinned host: {state, index}; index -1 = LM::finish */
  int32_t* hlmBeamN;            /* [B] pinned host: hypotheses in the beam ... */
  uint32_t* hlmBeam;            /* [B*K] ... and their LM states (LM::updateCache, Utils.h:346-354; which states are live) */
  /* Streams: LM-state ids are recycled (a stream may run forever; LexiconFreeDecoder.cpp:205-227 / Utils.h:312-342 bound
   * the reference's memory the same way).  Every id remembers who made it (idPar / idEdge / idBorn);
   * fltx_compact_states_kernel (compactStates below) renumbers the ids that a hypothesis of the beam can still meet
   * to 1 .. M and the engine's per-utterance id counter goes on from M + 1.  Null for offline decodes. */
  uint32_t* idPar;              /* [B*idCap] parent state of an id (kNoParent32: the root, or cut loose by a compaction) */
  int32_t* idEdge;              /* [B*idCap] generic engine: the edge that leads to it (token / word / -1), for the table's rebuild */
  uint32_t* idBorn;             /* [B*idCap] frame (stream clock: uttTotal) in which it was made */
  uint32_t* stateVal;           /* [B*stateCap] generic engine, streams: the id stored beside a stateTab key (ids are no longer slots) */
  /* optional phase profile: [B*8] accumulated shader clocks (bench/tuning) */
  unsigned long long* prof;
  int32_t profThread; /* the thread whose clock is sampled (lane 0 of the wave under study) */
};

struct Ws {
  /* beam, double buffered */
  /* (each array holds both buffers: index = buffer * K + slot; no arrays of
   * pointers here -- a runtime-indexed pointer array would push this whole
   * struct into scratch memory) */
  double* bScore;
  double* bAm;
  double* bLm;
  uint32_t* bState;
  uint32_t* bSPar;
  int32_t* bSEdge;
  uint32_t* bLex;
  uint32_t* bTokPb;
  float* bLexMax;  /* [2K] maxScore of the slot's trie node, 0 at the root (LexiconDecoder.cpp:58-59) */
  /* candidate records */
  double* cScore;
  uint4* cKey;     /* {state parent, state edge, lex node, token | prevBlank<<31} */
  uint32_t* cSrc;  /* parent slot | kNewState */
  int32_t* cAux;   /* word label or -1 */
  float* cLm;      /* LM score delta (float, as the reference holds it) */
  uint32_t* cOrd;  /* deterministic generation order (tie-break) */
  uint32_t* cNext; /* hash chain */
  unsigned long long* bLexMask; /* [K] child-token mask of the slot's trie node (this frame) */
  uint16_t* itemList; /* [itemCap] existing (hypothesis << 6 | token) children of the beam's trie nodes
                         (N <= 64, K <= 1024) */
  uint8_t* tokPos;    /* [N] position of a token in this frame's short-list */
  double* zScore;  /* [CAP2] score pass of the lexicon decoder: candidate score ... */
  uint32_t* zOrd;  /* [CAP2] ... and generation order = (item, sub-candidate), enough to rebuild it */
  float* zLm;      /* [CAP2] ... its LM score delta (extend candidates of a word LM: the child's maxScore,
                      from which the delta is re-derived), so that the rebuild does not score it again ... */
  int32_t* zAux;   /* [CAP2] ... and the child node (extend) or word label (word end): no second trie gather */
  uint32_t* head;  /* [HS] */
  uint32_t* lead;  /* group leaders (candidate index), later survivors */
  uint8_t* lstat;  /* per leader: 0 dropped, 1 active, 2 taken */
  uint16_t* lbin;
  uint32_t* small; /* boundary-bin members (slow select path) */
  unsigned long long* sKey; /* [SCAP] compact short-list: order-preserving score key */
  uint32_t* sOrd;  /* [SCAP] generation order (tie-break) */
  uint32_t* sIdx;  /* [SCAP] candidate index */
  uint32_t* sSrc;  /* [SCAP] source slot | kNewState (lean path) */
  uint4* sEnt;     /* [SCAP] lean short-list entry {key lo, key hi, order, group} */
  uint32_t* sBin;  /* [SCAP] histogram bin of the entry (lean path) */
  uint32_t* sNext; /* [SCAP] next entry of the same bin (lean path) */
  uint32_t* bhead; /* [NB] first short-list entry of a bin (lean path) */
  uint32_t* hcum;  /* [NB] number of candidates in better bins (lean path) */
  int16_t* dKid;   /* [K*N] lean: any slot whose state is child(state of rep, token) */
  unsigned long long* bMask;   /* [2K] lean: edges of the slot's LM state that already have a child state */
  unsigned long long* addMask; /* [K] lean: edges added this frame, per old-beam representative */
  unsigned long long* eBase;   /* [K] lean: new slot's mask before this frame's additions */
  int32_t* eRep;   /* [K] lean: old-beam representative of the new slot's state, -1 = fresh */
  int32_t* bPar;   /* [K] lean: parent slot of the new slot (for the coalesced history write) */
  unsigned long long* relTab; /* [K*(N+1)] lane path: new slots per LM-state descriptor (bit = slot) */
  unsigned long long* repTab; /* [K] lane path: tokens repeated by children of old slot x's state */
  uint4* bRec;     /* [2K] lane path: {score, token|prevBlank, state descriptor | parent descriptor << 16} */
  uint32_t* wcum;  /* [16*256] lane path: per-wave copy of the histogram prefix (512 x u16 per wave) */
  uint32_t* tick;  /* [512] lane path: scatter tickets per bin */
  int32_t* pMate;  /* [16*64] per-wave partial relation results (lean path) */
  int32_t* pPar;   /* [16*64] */
  uint32_t* surv;  /* [K] candidate index of the survivor with rank r */
  int32_t* dMate;  /* [K] dense merge: the other hypothesis in the same LM state */
  int32_t* dPar;   /* [K] dense merge: representative slot of the parent LM state */
  int16_t* dRep;   /* [K*N] dense merge: slot whose repeat feeds group (state, token) */
  uint8_t* dIn;    /* [N] token is in this frame's short-list */
  uint32_t* hist;  /* [NB] */
  float* erow;     /* [2*N] emission row, double buffered */
  int32_t* tokIdx; /* token short-list */
  uint32_t* wtmp;  /* [32] per-wave scratch for block scans */
  unsigned long long* red; /* [4] block reductions */
  int32_t* sc;     /* [16] block scalars */
};

enum { SC_NCAND = 0, SC_NLEAD = 1, SC_NSURV = 2, SC_BSTAR = 3, SC_CUM = 4, SC_M = 5,
       SC_NSMALL = 6, SC_STATUS = 7, SC_NEED = 8, SC_DONE = 9, SC_NEXTID = 10, SC_RELSLOW = 11, SC_NSLIM = 12, SC_CUT = 13, SC_BM = 14 };

#ifndef FLTX_HD
#ifdef FLTX_EMU
#define FLTX_HD inline
#else
#define FLTX_HD __host__ __device__ inline
#endif
#endif

FLTX_HD size_t alignUp(size_t x, size_t a) { return (x + a - 1) / a * a; }

/* Lane-per-slot path (fltx_lane.h): the per-frame state sits at FIXED offsets
 * from the start of the workgroup's LDS (sized for its limits, beam <= 64), so
 * the frame step addresses it with immediate offsets instead of one base
 * register per array.  The Ws pointers of these fields point into this block,
 * which is how the code shared with the other engines (begin / end / parking)
 * sees the same data.  Three variable-size tables follow it. */
constexpr int kLaneK = 64;      /* largest beam */
constexpr int kLaneSCAP = 320;  /* short-list capacity: beam + 256 */
struct LaneLds {
  uint4 bRec[2 * kLaneK];
  double bScore[2 * kLaneK];
  double bAm[2 * kLaneK];
  unsigned long long bMask[2 * kLaneK];
  unsigned long long addMask[kLaneK];
  unsigned long long eBase[kLaneK];
  unsigned long long repTab[kLaneK];
  uint4 sEnt[kLaneSCAP];
  uint32_t hist[512];
  uint32_t tick[512];
  uint32_t bState[2 * kLaneK];
  uint32_t bSPar[2 * kLaneK];
  uint32_t bTokPb[2 * kLaneK];
  int32_t bSEdge[2 * kLaneK];
  int32_t eRep[kLaneK];
  int32_t bPar[kLaneK];
  int32_t dMate[kLaneK];
  int32_t dPar[kLaneK];
  uint32_t sIdx[kLaneSCAP];
  uint32_t sSrc[kLaneSCAP];
  int32_t sc[16];
  /* token short-list (beamSizeToken < N) of the frame whose emission row sits in
   * erow buffer p: the list for the NEXT frame is made during the build phase */
  unsigned long long tokMask[2];
  uint8_t tokIdx[2][64]; /* position -> token */
  uint8_t tokPos[2][64]; /* token -> position */
};

/* Carve the workspace out of `base` (LDS or HBM); returns bytes used.  With
 * base == nullptr it only computes the size (host side). */
FLTX_HD size_t carveWs(Ws& w, char* base, int K, int CAP, int HS, int NB, int N, int SCAP,
                       int dense, int lane, int CAP2 = 0, int itemCap = 0, int nWaves = 16,
                       int splitHot = 0, char* hotBase = nullptr, size_t* hotBytes = nullptr) {
  size_t off = 0;
  /* splitHot: the workspace proper lives in HBM, but the small arrays every
   * thread hammers with atomics or re-reads all the time (histogram, its
   * prefixes, block scalars, the emission row) are carved from `hotBase` (LDS):
   * a hot L2 atomic costs hundreds of clocks, an LDS one tens */
  size_t offHot = 0;
  /* dense == 2: the lean / lane steps only -- no candidate records, no LM / lexicon
   * fields of the beam, none of the generic select's tables */
  const bool leanOnly = dense == 2; /* (the host then passes CAP = 1, HS = 64) */
  LaneLds* const LL = (lane && base) ? (LaneLds*)base : nullptr;
  if (lane) {
    off = alignUp(sizeof(LaneLds), 16);
  }
#define FLTX_CARVE(field, type, count)                       \
  off = alignUp(off, 16);                                    \
  field = (type*)(base ? base + off : nullptr);              \
  off += sizeof(type) * (size_t)(count);
#define FLTX_CARVE_H(field, type, count)                     \
  if (splitHot) {                                            \
    offHot = alignUp(offHot, 16);                            \
    field = (type*)(hotBase ? hotBase + offHot : nullptr);   \
    offHot += sizeof(type) * (size_t)(count);                \
  } else {                                                   \
    FLTX_CARVE(field, type, count)                           \
  }
  /* level 2: the candidate records and the merge hash go to LDS as well (the
   * beam, its select lists and the item list stay in HBM) */
#define FLTX_CARVE_R(field, type, count)                     \
  if (splitHot >= 2) {                                       \
    FLTX_CARVE_H(field, type, count)                         \
  } else {                                                   \
    FLTX_CARVE(field, type, count)                           \
  }
  /* a field of the fixed lane block in lane mode, carved like the rest otherwise */
#define FLTX_CARVE_L(field, type, count, member)             \
  if (lane) {                                                \
    field = LL ? (type*)LL->member : nullptr;                \
  } else {                                                   \
    FLTX_CARVE(field, type, count)                           \
  }
  FLTX_CARVE_L(w.bScore, double, 2 * K, bScore)
  FLTX_CARVE_L(w.bAm, double, 2 * K, bAm)
  FLTX_CARVE(w.bLm, double, leanOnly ? 0 : 2 * K)
  FLTX_CARVE_L(w.bState, uint32_t, 2 * K, bState)
  FLTX_CARVE_L(w.bSPar, uint32_t, 2 * K, bSPar)
  FLTX_CARVE_L(w.bSEdge, int32_t, 2 * K, bSEdge)
  FLTX_CARVE(w.bLex, uint32_t, leanOnly ? 0 : 2 * K)
  FLTX_CARVE_L(w.bTokPb, uint32_t, 2 * K, bTokPb)
  FLTX_CARVE(w.bLexMax, float, leanOnly ? 0 : 2 * K)
  FLTX_CARVE_H(w.erow, float, 2 * N)
  FLTX_CARVE_R(w.cScore, double, CAP)
  FLTX_CARVE_R(w.cKey, uint4, CAP)
  FLTX_CARVE_R(w.cSrc, uint32_t, CAP)
  FLTX_CARVE_R(w.cAux, int32_t, CAP)
  FLTX_CARVE_R(w.cLm, float, CAP)
  FLTX_CARVE_R(w.cOrd, uint32_t, CAP)
  FLTX_CARVE_R(w.cNext, uint32_t, CAP)
  FLTX_CARVE(w.bLexMask, unsigned long long, itemCap ? K : 0)
  FLTX_CARVE(w.itemList, uint16_t, itemCap)
  FLTX_CARVE(w.tokPos, uint8_t, itemCap ? N : 0)
  FLTX_CARVE(w.zScore, double, CAP2)
  FLTX_CARVE(w.zOrd, uint32_t, CAP2)
  FLTX_CARVE(w.zLm, float, CAP2)
  FLTX_CARVE(w.zAux, int32_t, CAP2)
  FLTX_CARVE_R(w.head, uint32_t, HS)
  FLTX_CARVE_R(w.lead, uint32_t, CAP)
  FLTX_CARVE_R(w.lstat, uint8_t, CAP)
  FLTX_CARVE_R(w.lbin, uint16_t, CAP)
  FLTX_CARVE_R(w.small, uint32_t, CAP)
  FLTX_CARVE(w.sKey, unsigned long long, SCAP)
  FLTX_CARVE(w.sOrd, uint32_t, SCAP)
  FLTX_CARVE_L(w.sIdx, uint32_t, SCAP, sIdx)
  FLTX_CARVE_L(w.sSrc, uint32_t, SCAP, sSrc)
  FLTX_CARVE_L(w.sEnt, uint4, SCAP, sEnt)
  FLTX_CARVE(w.sBin, uint32_t, 0)
  FLTX_CARVE(w.sNext, uint32_t, 0)
  FLTX_CARVE(w.bhead, uint32_t, 0)
  FLTX_CARVE_H(w.hcum, uint32_t, lane ? 0 : NB + NB / 16 + 1)
  FLTX_CARVE(w.dKid, int16_t, dense ? (size_t)K * N : 0)
  FLTX_CARVE_L(w.bMask, unsigned long long, dense ? 2 * K : 0, bMask)
  FLTX_CARVE_L(w.addMask, unsigned long long, dense ? K : 0, addMask)
  FLTX_CARVE_L(w.eBase, unsigned long long, dense ? K : 0, eBase)
  FLTX_CARVE_L(w.eRep, int32_t, dense ? K : 0, eRep)
  FLTX_CARVE_L(w.bPar, int32_t, dense ? K : 0, bPar)
  FLTX_CARVE(w.relTab, unsigned long long, lane ? (size_t)K * (N + 1) : 0)
  FLTX_CARVE_L(w.repTab, unsigned long long, lane ? K : 0, repTab)
  FLTX_CARVE_L(w.bRec, uint4, lane ? 2 * K : 0, bRec)
  FLTX_CARVE_H(w.wcum, uint32_t, (leanOnly && !lane) ? 0 : nWaves * 256)
  if (splitHot) {
    FLTX_CARVE_H(w.tick, uint32_t, (leanOnly && !lane) ? 0 : 512)
  } else {
    FLTX_CARVE_L(w.tick, uint32_t, (leanOnly && !lane) ? 0 : 512, tick)
  }
  FLTX_CARVE(w.pMate, int32_t, (dense && !lane) ? 16 * 64 : 0)
  FLTX_CARVE(w.pPar, int32_t, (dense && !lane) ? 16 * 64 : 0)
  FLTX_CARVE(w.surv, uint32_t, K)
  FLTX_CARVE_L(w.dMate, int32_t, dense ? K : 0, dMate)
  FLTX_CARVE_L(w.dPar, int32_t, dense ? K : 0, dPar)
  FLTX_CARVE(w.dRep, int16_t, (dense && !lane) ? (size_t)K * N : 0)
  FLTX_CARVE_H(w.dIn, uint8_t, N)
  if (splitHot) {
    FLTX_CARVE_H(w.hist, uint32_t, NB + NB / 16 + 1)
  } else {
    FLTX_CARVE_L(w.hist, uint32_t, NB + NB / 16 + 1, hist)
  }
  FLTX_CARVE_H(w.tokIdx, int32_t, N)
  FLTX_CARVE_H(w.wtmp, uint32_t, 32)
  FLTX_CARVE_H(w.red, unsigned long long, 4)
  if (splitHot) {
    FLTX_CARVE_H(w.sc, int32_t, 16)
  } else {
    FLTX_CARVE_L(w.sc, int32_t, 16, sc)
  }
#undef FLTX_CARVE_L
#undef FLTX_CARVE_R
#undef FLTX_CARVE_H
#undef FLTX_CARVE
  if (hotBytes) {
    *hotBytes = alignUp(offHot, 16);
  }
  return alignUp(off, 16);
}

FLTX_HD uint32_t hashKey(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
  uint32_t h = a * 0x9E3779B1u;
  h = (h ^ (h >> 15)) + b * 0x85EBCA77u;
  h = (h ^ (h >> 13)) + c * 0xC2B2AE3Du;
  h = (h ^ (h >> 16)) + d * 0x27D4EB2Fu;
  h ^= h >> 15;
  h *= 0x2C1B3C6Du;
  h ^= h >> 12;
  return h;
}

#ifndef FLTX_HOST_ONLY
/* ------------------------------------------------------------------------ */
/* block-level primitives                                                    */
/* ------------------------------------------------------------------------ */
/* Workgroup barrier for code whose workspace may live in HBM (big beams).
 * There the counters and hash heads are updated with L2 atomics, which do not
 * refresh this CU's vector L1, so a plain load after the barrier could hit a
 * stale line: after the barrier the L1 is invalidated (agent-scope acquire,
 * buffer_inv).  The release side only has to get this wave's stores to the L2
 * (workgroup scope: vmcnt wait; the L1 is write-through) -- every wave of a
 * workgroup sits on the same CU and therefore behind the same L2, so the
 * agent-scope release (buffer_wbl2: write the whole L2's dirty lines back to
 * HBM, tens of microseconds with 32 workgroups dirtying it) is not needed.  With the
 * workspace in LDS it waits for this wave's LDS operations only (ldsBarrier):
 * __syncthreads() would also drain the global loads of the emission-row
 * prefetch (~2k clocks) and the history stores at every barrier.
 * The generic step at hot level >= 1 (histogram and counters in LDS) reads the two kinds of HBM words that are
 * still updated with atomics at L2 (wsLoadAtomic32) and drops the invalidate (DecodeParams::wsNoInv; beam 500 x 29
 * tokens with a 4-gram LM: 215 -> 71 us per frame, beam 1000: 477 -> 177). */
FLTX_DEV void wsBarrier(const DecodeParams& P) {
#ifndef FLTX_EMU
  if (P.gws != nullptr) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); /* my stores are in L2 */
    __syncthreads();
    if (!P.wsNoInv) {     /* (generic step, levels 1, 2: the histogram and the counters are in LDS, the words of the HBM part that
                           * are updated with atomics are read at L2 (wsLoadAtomic32); everything else there is
                           * written with plain stores by waves of this CU, which its write-through L1 sees) */
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    return;
  }
#endif
  ldsBarrier();
}
/* The same, where threads exchange data through GLOBAL memory across it (the
 * n-gram context of a new LM state is written by the thread that builds the
 * slot and read by other threads in later frames): also waits for the wave's
 * outstanding global stores. */
FLTX_DEV void wsBarrierMem(const DecodeParams& P) {
#ifndef FLTX_EMU
  if (P.gws != nullptr) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); /* my stores are in L2 */
    __syncthreads();
    /* invalidate AFTER the barrier: the L1 is shared by the waves of the CU, so
     * a wave that is still loading before its barrier can re-populate lines
     * another wave already dropped */
    if (!P.wsNoInv) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    return;
  }
#endif
  __syncthreads();
}

/* A word of the HBM workspace that other waves update with L2 atomics (hash heads, the rank counts of the select):
 * read at L2, this CU's L1 may hold the line from before the atomics. */
FLTX_DEV uint32_t wsLoadAtomic32(const DecodeParams& P, const uint32_t* p) {
#ifndef FLTX_EMU
  if (P.gws != nullptr && P.hotLevel < 2) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
#endif
  return *p;
}

FLTX_DEV unsigned long long wsLoadAtomic64(const DecodeParams& P, const unsigned long long* p) {
#ifndef FLTX_EMU
  if (P.gws != nullptr) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
#endif
  return *p;
}

/* exclusive prefix sum of v over the workgroup; *total = block sum.
 * Contains two barriers; wtmp has >= nWaves+1 entries. */
FLTX_DEV int blockExclusiveScan(const DecodeParams& P, int v, uint32_t* wtmp, int* total) {
  const int lane = laneId(), wave = waveId();
  const int nW = ((int)blockDim.x + 63) >> 6;
  int inc = waveInclusiveScan(v);
  if (lane == 63) {
    wtmp[wave] = (uint32_t)inc;
  }
  wsBarrier(P);
  int base = 0, tot = 0;
  for (int i = 0; i < nW; ++i) {
    int x = (int)wtmp[i];
    if (i < wave) {
      base += x;
    }
    tot += x;
  }
  wsBarrier(P);
  *total = tot;
  return base + inc - v;
}

/* max / min of an f64 over the workgroup (2 barriers) */
FLTX_DEV double blockMaxF64(const DecodeParams& P, double v, unsigned long long* slot) {
  if (threadIdx.x == 0) {
    *slot = 0ull;
  }
  wsBarrier(P);
  unsigned long long k = waveMax64(f64Key(v));
  if (laneId() == 0) {
    atomMax64(slot, k);
  }
  wsBarrier(P);
  return f64FromKey(*slot);
}
FLTX_DEV double blockMinF64(const DecodeParams& P, double v, unsigned long long* slot) {
  if (threadIdx.x == 0) {
    *slot = ~0ull;
  }
  wsBarrier(P);
  unsigned long long k = waveMin64(f64Key(v));
  if (laneId() == 0) {
    atomMin64(slot, k);
  }
  wsBarrier(P);
  return f64FromKey(*slot);
}

/* ------------------------------------------------------------------------ */
/* candidate records + merge hash (candidatesAdd, Utils.h:131-144)            */
/* ------------------------------------------------------------------------ */

/* Append one candidate and chain it into the merge hash.  Every lane of the
 * wave must call this together (`valid` masks lanes without a candidate): the
 * record index comes from one wave-aggregated LDS atomic. */
FLTX_DEV void pushCandidate(const DecodeParams& P, const Ws& w, bool valid,
                            double score, uint32_t kp, uint32_t ke, uint32_t klex,
                            uint32_t ktp, uint32_t src, int32_t aux, float lm,
                            uint32_t ord, unsigned long long& bestKey, double preThr = -__builtin_huge_val()) {
  /* NaN never enters (Utils.h:138-143); preThr is (a lower bound of the frame's
   * best score) - beamThreshold: what is below can not survive candidatesStore
   * (Utils.h:160-170), so it is not even recorded */
  valid = valid && (score >= preThr);
  const unsigned long long m = waveBallot(valid);
  if (m == 0ull) {
    return;
  }
  const int lane = laneId();
  const int leader = __builtin_ctzll(m);
  uint32_t base = 0;
  if (lane == leader) {
    base = atomAdd32((uint32_t*)&w.sc[SC_NCAND], (uint32_t)popc64(m));
  }
  base = waveShfl32(base, leader);
  if (!valid) {
    return;
  }
  const uint32_t ci = base + (uint32_t)popc64(m & ((1ull << lane) - 1ull));
  if (ci >= (uint32_t)P.CAP) {
    atomOr32((uint32_t*)&w.sc[SC_STATUS], ST_CAND_OVERFLOW);
    return;
  }
  const unsigned long long sk = f64Key(score);
  bestKey = sk > bestKey ? sk : bestKey;
  w.cScore[ci] = score;
  w.cKey[ci] = make_uint4(kp, ke, klex, ktp);
  w.cSrc[ci] = src;
  w.cAux[ci] = aux;
  w.cLm[ci] = lm;
  w.cOrd[ci] = ord;
  compilerFence();
#ifndef FLTX_EMU
  if (P.gws != nullptr && P.hotLevel < 2) {
    /* records in HBM: the record must have reached L2 before the atomic below
     * publishes its index to the other waves (in LDS the pipeline order of one
     * wave's ds_write / ds_cmpst already guarantees this).  The readers sit on
     * the same CU and read at L2 (below), the L1 is write-through: waiting for
     * this wave's stores (workgroup-scope release) is enough -- an agent-scope
     * fence would write the L2 back as well. */
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  }
#endif
  uint32_t s = hashKey(kp, ke, klex, ktp) & (uint32_t)(P.HS - 1);
  for (;;) {
    uint32_t cur = ldsLoad32(&w.head[s]);
    if (cur == kEmpty) {
      cur = atomCas32(&w.head[s], kEmpty, ci);
      if (cur == kEmpty) {
        w.cNext[ci] = kEmpty;
        w.small[ci] = s; /* chain owner: foldGroups starts from head[s] */
        break;
      }
    }
    uint4 k;
#ifndef FLTX_EMU
    if (P.gws != nullptr && P.hotLevel < 2) { /* another wave's record: read it at L2, not from a stale L1 line */
      const uint32_t* kq = (const uint32_t*)&w.cKey[cur];
      k.x = __hip_atomic_load(kq + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      k.y = __hip_atomic_load(kq + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      k.z = __hip_atomic_load(kq + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      k.w = __hip_atomic_load(kq + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else
#endif
    {
      k = w.cKey[cur];
    }
    if (k.x == kp && k.y == ke && k.z == klex && k.w == ktp) {
      const uint32_t old = atomExch32(&w.head[s], ci);
      w.cNext[ci] = old;
      break;
    }
    s = (s + 1) & (uint32_t)(P.HS - 1);
  }
}

/* Score pass of the lexicon decoder: remember only the score and the
 * generation order of a candidate (the order encodes hypothesis, token and
 * which of the item's candidates it is, so the full record can be rebuilt). */
FLTX_DEV void pushSlim(const DecodeParams& P, const Ws& w, bool valid, double score, uint32_t ord, float lm,
                       int32_t aux, unsigned long long& bestKey, double preThr) {
  valid = valid && (score >= preThr);
  const unsigned long long m = waveBallot(valid);
  if (m == 0ull) {
    return;
  }
  const int lane = laneId();
  const int leader = __builtin_ctzll(m);
  uint32_t base = 0;
  if (lane == leader) {
    base = atomAdd32((uint32_t*)&w.sc[SC_NSLIM], (uint32_t)popc64(m));
  }
  base = waveShfl32(base, leader);
  if (!valid) {
    return;
  }
  const uint32_t ci = base + (uint32_t)popc64(m & ((1ull << lane) - 1ull));
  if (ci >= (uint32_t)P.CAP2) {
    atomOr32((uint32_t*)&w.sc[SC_STATUS], ST_CAND_OVERFLOW);
    return;
  }
  const unsigned long long sk = f64Key(score);
  bestKey = sk > bestKey ? sk : bestKey;
  w.zScore[ci] = score;
  w.zOrd[ci] = ord;
  w.zLm[ci] = lm;
  w.zAux[ci] = aux;
}

/* ------------------------------------------------------------------------ */
/* n-gram LM on flat tables (replaces KenLM BaseScore, lm/KenLM.cpp:63-83)    */
/* ------------------------------------------------------------------------ */
FLTX_DEV bool ngFind(const DecodeParams& P, uint32_t ctx, uint32_t word, uint32_t& node,
                     float& prob) {
  uint32_t s = hashKey(ctx, word, 0x5bd1e995u, 0) & P.ngMask;
  for (;;) {
    const NgramSlot e = P.ngTab[s];
    if (e.word == kEmpty) {
      return false;
    }
    if (e.ctx == ctx && e.word == word) {
      node = e.node;
      prob = e.prob;
      return true;
    }
    s = (s + 1) & P.ngMask;
  }
}

/* ctxIn[j] = node id of the suffix n-gram made of the last j+1 context words
 * (0 = absent), j < order-1.  Returns log10 p(word | context) summed in float:
 * longest match first, then the back-off weights of the skipped contexts from
 * the shortest to the longest (KenLM FullScore order, see oracle/arpa_lm.h).
 *
 * The look-ups for the different context lengths do not depend on each other,
 * only the choice among their results does: all first probes and all back-off
 * weights are loaded at once (one HBM/L2 round trip instead of up to
 * 2 * order dependent ones), and everything is indexed statically so that it
 * stays in registers (a dynamically indexed local array lives in scratch
 * memory). */
template <int MO> /* MO: an upper bound of the model's order, to size the unrolled loops */
FLTX_DEV float ngScoreT(const DecodeParams& P, const int32_t* ctxIn, uint32_t word,
                       int32_t* ctxOut) {
  const int L = P.lmOrder - 1;
  uint32_t c[MO];     /* context node for k context words (k = 0: none) */
  uint32_t slot[MO];
  NgramSlot e[MO];
  float bo[MO];
  bool act[MO];
#pragma unroll
  for (int k = 0; k < MO; ++k) {
    c[k] = (k >= 1 && k <= L) ? (uint32_t)ctxIn[k - 1] : 0u;
  }
#pragma unroll
  for (int k = 0; k < MO; ++k) {
    /* k context words; a suffix of the context that is not an n-gram of the model is skipped */
    act[k] = k <= L && (k == 0 || c[k] != 0u);
    slot[k] = hashKey(c[k], word, 0x5bd1e995u, 0) & P.ngMask;
    e[k].ctx = 0u;
    e[k].word = kEmpty;
    e[k].node = 0u;
    e[k].prob = 0.0f;
    bo[k] = 0.0f;
    if (act[k]) {
      e[k] = P.ngTab[slot[k]];
    }
    if (k >= 1 && k <= L && c[k] != 0u) {
      bo[k] = P.ngBackoff[c[k]];
    }
  }
  uint32_t nodes[MO];
  bool found[MO];
  int longest = -1;
  float prob = 0.0f;
#pragma unroll
  for (int k = 0; k < MO; ++k) {
    found[k] = false;
    nodes[k] = 0u;
    if (act[k]) {
      for (;;) { /* open addressing: the first probe almost always decides */
        if (e[k].word == kEmpty) {
          break;
        }
        if (e[k].ctx == c[k] && e[k].word == word) {
          found[k] = true;
          break;
        }
        slot[k] = (slot[k] + 1) & P.ngMask;
        e[k] = P.ngTab[slot[k]];
      }
      if (found[k]) {
        if (!(e[k].node & kPhantomNode)) {
          longest = k;
          prob = e[k].prob;
        }
        nodes[k] = e[k].node & ~kPhantomNode;
      }
    }
  }
  if (longest < 0) { /* not even a unigram: score <unk> */
    uint32_t n0;
    if (!ngFind(P, 0u, (uint32_t)P.lmUnk, n0, prob)) {
      prob = -100.0f;
      n0 = 0;
    }
    nodes[0] = n0;
    found[0] = true;
    longest = 0;
  }
#pragma unroll
  for (int j = 1; j < MO; ++j) {
    if (j > longest && j <= L && c[j] != 0u) {
      prob += bo[j];
    }
  }
  if (ctxOut) {
#pragma unroll
    for (int j = 0; j < MO - 1; ++j) {
      if (j < L) {
        /* suffix of length j+1 of (context, word) = n-gram (last j ctx words, word) */
        ctxOut[j] = (j <= longest && found[j]) ? (int32_t)nodes[j] : 0;
      }
    }
  }
  return prob;
}

FLTX_DEV float ngScore(const DecodeParams& P, const int32_t* ctxIn, uint32_t word, int32_t* ctxOut) {
  return ngScoreT<kMaxNgramOrder>(P, ctxIn, word, ctxOut);
}

/* LM::score for the hypothesis in beam slot h (state id sid). */
/* LM::score of the n-gram LM (lm/KenLM.cpp:63-75) for (LM state id, word).  A hypothesis at a word-end node asks the
 * same question frame after frame until it leaves the beam, and an answer is a chain of dependent table probes
 * (~4 k clocks): the last answers are kept in a direct-mapped table per utterance.  One 64-bit word per entry
 * = tag:32 | score bits:32, written and read whole; slot = (word ^ f(state)) mod kLmCache, tag = state:24 | word >> 12,
 * so slot and tag together determine (state, word): a hit is exact, never a hash coincidence. */
constexpr int kLmCache = 4096;
/* A token-level n-gram LM over a small token set has a dense (context, token) table (DecodeParams::tokLm, built by
 * lmTokDense in fltx_api.cpp with the host twin of ngScore: the same floats).  The generic engine then keeps a state's
 * context as a ROW NUMBER in word 0 of its stateCtx entry and a look-up is one gather instead of a chain of probes. */
FLTX_DEV uint32_t tokLmRow(const DecodeParams& P, int b, uint32_t sid) {
  const int L = P.lmOrder - 1;
  return L > 0 ? (uint32_t)P.stateCtx[((size_t)b * P.stateCap + sid) * L] : 0u;
}
FLTX_DEV float lmScoreDev(const DecodeParams& P, int b, uint32_t sid, int usr) {
  if (P.lmKind == 0) {
    return 0.0f;
  }
  if (P.tokLm != nullptr) {
    const int col = (usr >= 0 && usr < P.N) ? usr : 0;
    return __uint_as_float((uint32_t)P.tokLm[(size_t)tokLmRow(P, b, sid) * (size_t)P.tokLmStride + (size_t)col].x);
  }
  const int L = P.lmOrder - 1;
  const int32_t* ctx = P.stateCtx + ((size_t)b * P.stateCap + sid) * L;
  const uint32_t word = (usr >= 0 && usr < P.nUsr) ? (uint32_t)P.usrToLm[usr] : (uint32_t)P.lmUnk;
  const uint32_t uk = (uint32_t)(usr + 1);
  if (P.lmCache != nullptr && uk < (1u << 20) && sid < 0xFFFFFFu) {
    unsigned long long* c = P.lmCache + (size_t)b * kLmCache;
    const uint32_t slot = (uk ^ sid ^ (sid >> 12)) & (uint32_t)(kLmCache - 1);
    const uint32_t tag = (sid << 8) | (uk >> 12);
    const unsigned long long e = c[slot];
    if ((uint32_t)(e >> 32) == tag) {
      return __uint_as_float((uint32_t)e);
    }
    const float v = ngScore(P, ctx, word, nullptr);
    c[slot] = ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v);
    return v;
  }
  return ngScore(P, ctx, word, nullptr);
}
FLTX_DEV float lmFinishDev(const DecodeParams& P, int b, uint32_t sid) {
  if (P.lmKind == 0) {
    return 0.0f;
  }
  if (P.tokLm != nullptr) {
    return __uint_as_float((uint32_t)P.tokLm[(size_t)tokLmRow(P, b, sid) * (size_t)P.tokLmStride + (size_t)P.N].x);
  }
  const int L = P.lmOrder - 1;
  const int32_t* ctx = P.stateCtx + ((size_t)b * P.stateCap + sid) * L;
  return ngScore(P, ctx, (uint32_t)P.lmEos, nullptr);
}

/* Host LM: the answer to LM::score(state sid, usr) / LM::finish (usr == -1) of this frame, from the table the host
 * uploaded for utterance b (same hash as the host's insert, fltx_api.cpp hostLmAnswer). */
FLTX_DEV float hostLmLookup(const DecodeParams& P, int b, uint32_t sid, int usr, uint32_t& child, uint32_t* status) {
  const uint2 dir = P.hlmDir[b];
  uint32_t s = hashKey(sid, (uint32_t)usr, 0x7f4a7c15u, 0) & dir.y;
  for (uint32_t probes = 0; probes <= dir.y; ++probes) {
    const uint4 e = P.hlmTab[(size_t)dir.x + s];
    if (e.x == sid && e.y == (uint32_t)usr) {
      child = e.z;
      return __uint_as_float(e.w);
    }
    if (e.x == kEmpty && e.y == kEmpty) {
      break;
    }
    s = (s + 1) & dir.y;
  }
  atomOr32(status, ST_HLM_MISS);
  child = sid;
  return 0.0f;
}

/* LM::score(state of a hypothesis, usr) as candidate generation needs it: the score and the merge key (kp, ke) of the
 * state it leads to -- (parent id, edge) for the LMs whose states are a trie over their inputs (ZeroLM, the n-gram
 * tables: lm/LM.h:24-34), (id the host gave the returned state, kHostEdge) for a host LM. */
FLTX_DEV float lmAsk(const DecodeParams& P, const Ws& w, int b, uint32_t sid, int usr, uint32_t& kp, uint32_t& ke) {
  if (P.lmKind == 2) {
    ke = kHostEdge;
    return hostLmLookup(P, b, sid, usr, kp, (uint32_t*)&w.sc[SC_STATUS]);
  }
  kp = sid;
  ke = (uint32_t)usr;
  return lmScoreDev(P, b, sid, usr);
}

/* ------------------------------------------------------------------------ */
/* LM-state identity: lookup-or-insert (parent id, edge) -> id in HBM          */
/* (LMState::child, lm/LM.h:24-34).  Key = epoch:16 | parent:24 | edge+1:24.  */
/* ------------------------------------------------------------------------ */
FLTX_DEV uint32_t stateChild(const DecodeParams& P, int b, uint32_t par, int32_t edge,
                             uint32_t* status, bool& fresh) {
  unsigned long long* tab = P.stateTab + (size_t)b * P.stateCap;
  const unsigned long long key = ((unsigned long long)P.epoch << 48) |
      ((unsigned long long)(par & 0xFFFFFFu) << 24) | (unsigned long long)((uint32_t)(edge + 1) & 0xFFFFFFu);
  const uint32_t mask = P.stateCap - 1;
  uint32_t s = hashKey(par, (uint32_t)edge, 0x9747b28cu, 0) & mask;
  fresh = false;
  for (uint32_t probes = 0; probes < P.stateCap; ++probes) {
    if (s == 0) { /* id 0 is the root state */
      s = 1;
    }
    unsigned long long cur = loadCoherent64(&tab[s]);
    for (;;) {
      if (cur == key) {
        return s;
      }
      if ((cur >> 48) == (unsigned long long)P.epoch) {
        break; /* live entry of another state: next slot */
      }
      const unsigned long long old = atomCas64(&tab[s], cur, key);
      if (old == cur) {
        fresh = true;
        return s;
      }
      cur = old; /* lost the race: re-examine what is there now */
    }
    s = (s + 1) & mask;
  }
  atomOr32(status, ST_TABLE_FULL);
  return 0;
}

/* A fresh LM-state id: k, the value of the engine's per-utterance counter.  Streams remember its parent, edge and
 * frame for compactStates(), which renumbers the ids still needed and puts the counter behind them. */
FLTX_DEV uint32_t allocStateId(const DecodeParams& P, int b, uint32_t k, uint32_t parent, int32_t edge, uint32_t born,
                               uint32_t* status) {
  if ((int64_t)k >= P.idCap) {
    atomOr32(status, ST_TABLE_FULL);
    return 0u;
  }
  if (P.idPar != nullptr) {
    const size_t at = (size_t)b * P.idCap + k;
    P.idPar[at] = parent;
    P.idBorn[at] = born;
    if (P.idEdge) {
      P.idEdge[at] = edge;
    }
  }
  return k;
}

/* stateChild() for streams of the generic engine: the table maps (parent id, edge) to an id kept BESIDE the key
 * (stateVal), taken from the utterance's id counter by whoever inserts the key.  Several survivors of a frame may ask for the same
 * key at once (same LM state, different trie nodes): the one whose compare-and-swap installs the key allocates, the
 * others read the value once it is there.  The three phases are straight-line code for the whole wave -- a lane never
 * waits for a lane of its own wave that has not had its turn (waves of a workgroup all make progress). */
FLTX_DEV uint32_t stateChildIds(const DecodeParams& P, const Ws& w, int b, bool want, uint32_t par, int32_t edge, uint32_t born,
                                bool& fresh) {
  unsigned long long* tab = P.stateTab + (size_t)b * P.stateCap;
  uint32_t* val = P.stateVal + (size_t)b * P.stateCap;
  const unsigned long long key = ((unsigned long long)P.epoch << 48) |
      ((unsigned long long)(par & 0xFFFFFFu) << 24) | (unsigned long long)((uint32_t)(edge + 1) & 0xFFFFFFu);
  const uint32_t mask = P.stateCap - 1;
  uint32_t s = hashKey(par, (uint32_t)edge, 0x9747b28cu, 0) & mask;
  bool won = false, found = false;
  if (want) { /* phase 1: find the key or install it */
    for (uint32_t probes = 0; probes < P.stateCap && !found && !won; ++probes) {
      unsigned long long cur = loadCoherent64(&tab[s]);
      for (;;) {
        if (cur == key) {
          found = true;
          break;
        }
        if ((cur >> 48) == (unsigned long long)P.epoch) {
          break; /* live entry of another state: next slot */
        }
        const unsigned long long old = atomCas64(&tab[s], cur, key);
        if (old == cur) {
          won = true;
          break;
        }
        cur = old;
      }
      if (!found && !won) {
        s = (s + 1) & mask;
      }
    }
    if (!found && !won) {
      atomOr32((uint32_t*)&w.sc[SC_STATUS], ST_TABLE_FULL);
    }
  }
  waveSync();
  uint32_t id = 0u;
  fresh = won;
  if (won) { /* phase 2: the installer names the state */
    const uint32_t k = atomAdd32((uint32_t*)&w.sc[SC_NEXTID], 1u);
    id = allocStateId(P, b, k, par, edge, born, (uint32_t*)&w.sc[SC_STATUS]);
    storeCoherent32(&val[s], id);
  }
  waveSync();
  if (found) { /* phase 3: everybody else reads the name */
    for (;;) {
      id = loadCoherent32(&val[s]);
      if (id != kEmpty) {
        break;
      }
    }
  }
  return id;
}

/* ------------------------------------------------------------------------ */
/* candidate generation                                                      */
/* ------------------------------------------------------------------------ */
#ifdef FLTX_EMU
FLTX_DEV unsigned long long devClock() { return 0ull; }
#else
FLTX_DEV unsigned long long devClock() { return __builtin_readcyclecounter(); }
#endif
/* accumulate the time since the previous mark into profile slot i (thread 0,
 * register accumulators: a global read-modify-write per mark would stall the
 * wave for ~2k clocks and distort what it measures) */
#define FLTX_PROF(i)                                                   \
  do {                                                                 \
    if (P.prof && (int)threadIdx.x == P.profThread) {                  \
      const unsigned long long t_ = devClock();                        \
      f.acc[(i)] += t_ - f.t0;                                         \
      f.t0 = t_;                                                       \
    }                                                                  \
  } while (0)

struct FrameCtx {
  unsigned long long t0;
  unsigned long long acc[8];
  int64_t histBase; /* P.histOff[b], loaded once per launch */
  int b;         /* utterance */
  int cur;       /* beam buffer holding hyp_[frame] */
  int nBeam;
  int nTok;      /* min(beamSizeToken, N) */
  bool useTrans; /* ASG and global frame > 0 */
  uint32_t clock = 0; /* frames decoded since decodeBegin, this one included (a stream's clock: DecodeParams::idBorn) */
  const float* e; /* emission row in LDS */
  mutable uint32_t nScored = 0; /* n-gram LM queries of this thread (roofline accounting) */
};

/* LexiconFreeDecoder::decodeStep inner loops (LexiconFreeDecoder.cpp:54-112) */
FLTX_DEV void genLexFree(const DecodeParams& P, const Ws& w, const FrameCtx& f,
                         unsigned long long& bestKey) {
  const int total = f.nBeam * f.nTok;
  const int W = (int)blockDim.x;
  const bool ctc = P.criterion == 1;
  const int rounds = (total + W - 1) / W;
  for (int it = 0; it < rounds; ++it) {
    const int i = it * W + (int)threadIdx.x;
    const bool valid = i < total;
    double score = 0;
    uint32_t kp = 0, ke = 0, ktp = 0, src = 0;
    float lm = 0.0f;
    if (valid) {
      const int h = i / f.nTok, r = i - h * f.nTok;
      const int n = (f.nTok == P.N) ? r : w.tokIdx[r];
      const uint32_t tp = w.bTokPb[(f.cur) * P.K + h];
      const int prevTok = (int)(tp & 0x7FFFFFFFu);
      const bool prevBlank = (tp & kPrevBlank) != 0;
      score = w.bScore[(f.cur) * P.K + h] + (double)f.e[n]; /* :64, transition excluded */
      if (n == P.sil) {
        score += P.silScore;
      }
      src = (uint32_t)h;
      const bool newTok = ctc ? (n != P.blank && (n != prevTok || prevBlank)) : (n != prevTok);
      if (newTok) { /* :69-85 */
        lm = lmAsk(P, w, f.b, w.bState[(f.cur) * P.K + h], n, kp, ke);
        f.nScored += P.lmKind != 0 ? 1u : 0u;
        score = score + P.lmWeight * (double)lm;
        ktp = (uint32_t)n;
        src |= kNewState;
      } else { /* blank :86-97 / repeat :98-110 keep the LM state */
        kp = w.bSPar[(f.cur) * P.K + h];
        ke = (uint32_t)w.bSEdge[(f.cur) * P.K + h];
        ktp = (uint32_t)n | ((ctc && n == P.blank) ? kPrevBlank : 0u);
      }
    }
    pushCandidate(P, w, valid, score, kp, ke, 0u, ktp, src, -1, lm, (uint32_t)i, bestKey);
  }
}

/* ------------------------------------------------------------------------ */
/* Dense (hash-free) merge for the lexicon-free decoder.                      */
/*                                                                            */
/* With key = (LM state, token, prevBlank) (LexiconFreeDecoder.h:68-78) the    */
/* candidates of a frame fall into groups that can be enumerated without any  */
/* search, because a beam holds at most two hypotheses per LM state S --       */
/* (S, edge(S), pb=0) and, for CTC, (S, blank, pb=1):                          */
/*   G[S][n], n != blank : the "new token n" candidates of the <= 2            */
/*        hypotheses in state S, plus the *repeat* candidate of the hypothesis */
/*        (child(S,n), n, 0) if that one is in the beam (its state's parent is */
/*        S and its edge is n, so the keys coincide);                          */
/*   G[S][blank] (CTC)   : the blank candidates of the <= 2 hypotheses in S;   */
/*   orphan[h]           : the repeat candidate of a hypothesis whose parent   */
/*        state has no hypothesis in the beam (nothing to merge with).         */
/* One thread evaluates one group: <= 3 member scores, max or ordered log-add. */
/* Records are written at the dense index g = rep*nTok + r (orphans after      */
/* them) and only the groups that pass the threshold become leaders, so the    */
/* prune and beam-build stages are shared with the generic hash path.          */
/* Equal to the generic path by construction; tests run both.                  */
/* ------------------------------------------------------------------------ */
struct DenseMember {
  double s;
  uint32_t ord;
  uint32_t src;
  float lm;
};

FLTX_DEV void denseConsider(DenseMember& best, int& nMem, DenseMember* mem, const DenseMember& m) {
  mem[nMem++] = m;
  if (nMem == 1 || m.s > best.s || (m.s == best.s && m.ord < best.ord)) {
    best = m;
  }
}

/* Evaluate dense group g.  Returns false when the group has no member (that
 * passes `thr` when useThr).  score = max, or the ordered log-add when logAdd. */
FLTX_DEV bool denseEval(const DecodeParams& P, const Ws& w, const FrameCtx& f, int g, bool useThr,
                        double thr, double& score, DenseMember& best, uint4& key) {
  const bool ctc = P.criterion == 1;
  const int cur = f.cur;
  const int nGroups = f.nBeam * f.nTok;
  DenseMember mem[3];
  int nMem = 0;
  if (g >= nGroups) { /* orphan repeat of hypothesis h */
    const int h = g - nGroups;
    if (h >= f.nBeam) {
      return false;
    }
    const uint32_t tp = w.bTokPb[(cur) * P.K + h];
    const int t = (int)(tp & 0x7FFFFFFFu);
    if ((tp & kPrevBlank) || (ctc && t == P.blank) || w.dPar[h] >= 0 || !w.dIn[t]) {
      return false;
    }
    int r = t;
    if (f.nTok != P.N) { /* position of t in the short-list (for the generation order) */
      for (r = 0; r < f.nTok && w.tokIdx[r] != t; ++r) {
      }
    }
    double s = w.bScore[(cur) * P.K + h] + (double)f.e[t];
    if (t == P.sil) {
      s += P.silScore;
    }
    if (s != s || (useThr && !(s >= thr))) {
      return false;
    }
    best = DenseMember{s, (uint32_t)(h * f.nTok + r), (uint32_t)h, 0.0f};
    score = s;
    key = make_uint4(w.bSPar[(cur) * P.K + h], (uint32_t)w.bSEdge[(cur) * P.K + h], 0u, (uint32_t)t);
    return true;
  }
  const int rep = g / f.nTok, r = g - rep * f.nTok;
  const int mate = w.dMate[rep];
  if (mate >= 0 && mate < rep) {
    return false; /* the lower slot of the pair evaluates the state's groups */
  }
  const int n = (f.nTok == P.N) ? r : w.tokIdx[r];
  const double en = (double)f.e[n];
  const uint32_t sid = w.bState[(cur) * P.K + rep];
  if (ctc && n == P.blank) { /* LexiconFreeDecoder.cpp:86-97 */
    for (int q = 0; q < 2; ++q) {
      const int h = q == 0 ? rep : mate;
      if (h < 0) {
        continue;
      }
      double s = w.bScore[(cur) * P.K + h] + en;
      if (n == P.sil) {
        s += P.silScore;
      }
      if (s != s || (useThr && !(s >= thr))) {
        continue;
      }
      denseConsider(best, nMem, mem, DenseMember{s, (uint32_t)(h * f.nTok + r), (uint32_t)h, 0.0f});
    }
    key = make_uint4(w.bSPar[(cur) * P.K + rep], (uint32_t)w.bSEdge[(cur) * P.K + rep], 0u, (uint32_t)n | kPrevBlank);
  } else {
    float lm = 0.0f;
    bool haveLm = false;
    for (int q = 0; q < 2; ++q) { /* new token n from the hypotheses in state S, :69-85 */
      const int h = q == 0 ? rep : mate;
      if (h < 0) {
        continue;
      }
      const uint32_t tp = w.bTokPb[(cur) * P.K + h];
      const int prevTok = (int)(tp & 0x7FFFFFFFu);
      const bool prevBlank = (tp & kPrevBlank) != 0;
      const bool newTok = ctc ? (n != prevTok || prevBlank) : (n != prevTok);
      if (!newTok) {
        continue; /* a repeat of h: belongs to h's own state group */
      }
      if (!haveLm) {
        lm = lmScoreDev(P, f.b, sid, n);
        f.nScored += P.lmKind != 0 ? 1u : 0u;
        haveLm = true;
      }
      double s = w.bScore[(cur) * P.K + h] + en;
      if (n == P.sil) {
        s += P.silScore;
      }
      s = s + P.lmWeight * (double)lm;
      if (s != s || (useThr && !(s >= thr))) {
        continue;
      }
      denseConsider(best, nMem, mem, DenseMember{s, (uint32_t)(h * f.nTok + r), (uint32_t)h | kNewState, lm});
    }
    const int hr = (int)w.dRep[rep * P.N + n]; /* repeat of (child(S,n), n, 0), :98-110 */
    if (hr >= 0) {
      double s = w.bScore[(cur) * P.K + hr] + en;
      if (n == P.sil) {
        s += P.silScore;
      }
      if (!(s != s || (useThr && !(s >= thr)))) {
        denseConsider(best, nMem, mem, DenseMember{s, (uint32_t)(hr * f.nTok + r), (uint32_t)hr, 0.0f});
      }
    }
    key = make_uint4(sid, (uint32_t)n, 0u, (uint32_t)n);
  }
  if (nMem == 0) {
    return false;
  }
  score = best.s;
  if (P.logAdd && nMem > 1) { /* Utils.h:186-193: descending order, left to right */
    for (int a = 0; a < nMem; ++a) { /* tiny insertion sort by (score desc, ord asc) */
      for (int c = a + 1; c < nMem; ++c) {
        if (mem[c].s > mem[a].s || (mem[c].s == mem[a].s && mem[c].ord < mem[a].ord)) {
          const DenseMember t = mem[a];
          mem[a] = mem[c];
          mem[c] = t;
        }
      }
    }
    double acc = mem[0].s;
    for (int a = 1; a < nMem; ++a) {
      const double mx = acc > mem[a].s ? acc : mem[a].s;
      const double mn = acc > mem[a].s ? mem[a].s : acc;
      acc = mx + log1p(exp(mn - mx));
    }
    score = acc;
  }
  return true;
}

/* per-frame relation tables: mate (same LM state), parent representative,
 * repeat table; two barriers */
FLTX_DEV void densePrepare(const DecodeParams& P, const Ws& w, const FrameCtx& f) {
  const int W = (int)blockDim.x;
  const int tid = (int)threadIdx.x;
  const int cur = f.cur;
  const bool ctc = P.criterion == 1;
  for (int i = tid; i < f.nBeam * P.N; i += W) {
    w.dRep[i] = (int16_t)-1;
  }
  for (int n = tid; n < P.N; n += W) {
    w.dIn[n] = (f.nTok == P.N) ? 1 : 0;
  }
  for (int h = tid; h < f.nBeam; h += W) {
    const uint32_t sid = w.bState[(cur) * P.K + h];
    const uint32_t sp = w.bSPar[(cur) * P.K + h];
    int mate = -1, par = -1;
#pragma unroll 8
    for (int h2 = 0; h2 < f.nBeam; ++h2) { /* (eight loads in flight: one at a time made this scan the longest phase) */
      const uint32_t s2 = w.bState[(cur) * P.K + h2];
      mate = (s2 == sid && h2 != h) ? h2 : mate;
      par = (s2 == sp && par < 0) ? h2 : par; /* lowest slot in the parent state = its representative */
    }
    w.dMate[h] = mate;
    w.dPar[h] = par;
  }
  wsBarrier(P);
  if (f.nTok != P.N) {
    for (int r = tid; r < f.nTok; r += W) {
      w.dIn[w.tokIdx[r]] = 1;
    }
  }
  for (int h = tid; h < f.nBeam; h += W) {
    const uint32_t tp = w.bTokPb[(cur) * P.K + h];
    const int t = (int)(tp & 0x7FFFFFFFu);
    const int par = w.dPar[h];
    if (!(tp & kPrevBlank) && !(ctc && t == P.blank) && par >= 0 && w.bSEdge[(cur) * P.K + h] == t) {
      w.dRep[par * P.N + t] = (int16_t)h;
    } else if (!(tp & kPrevBlank) && !(ctc && t == P.blank) && par >= 0) {
      w.dPar[h] = -1; /* cannot happen (edge == token when pb == 0); stay correct anyway */
    }
  }
  wsBarrier(P);
}

/* Phase 1: best score over all candidates.  Phase 2 (after the threshold is
 * known): leaders with complete records. */
FLTX_DEV void genLexFreeDense(const DecodeParams& P, const Ws& w, const FrameCtx& f,
                              unsigned long long& bestKey) {
  const int W = (int)blockDim.x;
  const int total = f.nBeam * f.nTok + f.nBeam;
  for (int g = (int)threadIdx.x; g < total; g += W) {
    double sc;
    DenseMember best;
    uint4 key;
    /* the threshold applies per member, but the best candidate of the frame
     * is a single member, so the un-thresholded member maximum is what
     * candidatesAdd tracks (Utils.h:138-140) */
    const bool saveLog = false;
    (void)saveLog;
    if (denseEval(P, w, f, g, false, 0.0, sc, best, key)) {
      const unsigned long long k = f64Key(best.s);
      bestKey = k > bestKey ? k : bestKey;
    }
  }
}

FLTX_DEV void denseLeaders(const DecodeParams& P, const Ws& w, const FrameCtx& f, double thr) {
  const int W = (int)blockDim.x;
  const int total = f.nBeam * f.nTok + f.nBeam;
  const int rounds = (total + W - 1) / W;
  for (int it = 0; it < rounds; ++it) {
    const int g = it * W + (int)threadIdx.x;
    double sc = 0;
    DenseMember best;
    uint4 key;
    const bool isLead = g < total && denseEval(P, w, f, g, true, thr, sc, best, key);
    const unsigned long long m = waveBallot(isLead);
    if (m != 0ull) {
      const int lane = laneId();
      const int leader = __builtin_ctzll(m);
      uint32_t base = 0;
      if (lane == leader) {
        base = atomAdd32((uint32_t*)&w.sc[SC_NLEAD], (uint32_t)popc64(m));
      }
      base = waveShfl32(base, leader);
      if (isLead) {
        const uint32_t li = base + (uint32_t)popc64(m & ((1ull << lane) - 1ull));
        w.lead[li] = (uint32_t)g;
        w.cScore[g] = sc;
        w.cKey[g] = key;
        w.cSrc[g] = best.src;
        w.cAux[g] = -1;
        w.cLm[g] = best.lm;
        w.cOrd[g] = best.ord;
      }
    }
  }
}

/* LexiconDecoder::decodeStep inner loops (LexiconDecoder.cpp:55-215).  Work
 * item = (hypothesis, r): r < nTok tries the r-th short-listed token as a trie
 * child, r == nTok is "same node" (2), r == nTok+1 is CTC blank (3). */
/* linear bins over [preThr, hi] for the recompute form of the cut-off generation */
/* item i of the list of existing (hypothesis, token) children: hypothesis << 6 | token */
FLTX_DEV uint32_t itemCode(const DecodeParams& P, const Ws& w, int i) {
  return P.itemWide ? ((const uint32_t*)w.itemList)[i] : (uint32_t)w.itemList[i];
}

struct CutBins {
  double hi, scale;
  int bM;
};
FLTX_DEV int cutBinOf(const DecodeParams& P, const CutBins& cb, double sc) {
  const double x = (cb.hi - sc) * cb.scale;
  int bin = (x < (double)P.NB) ? (int)x : P.NB - 1;
  return bin < 0 ? 0 : bin;
}

/* MODE 0: every candidate becomes a record.  MODE 1: score pass into slim
 * records (cut-off generation).  MODE 2 / 3: the recompute form of the cut-off
 * generation for beams whose slim list would not fit: pass 2 only counts the
 * candidates per score bin, pass 3 generates everything again and builds
 * records for the candidates at or above the cut bin. */
template <int MODE, bool LISTED>
FLTX_DEV void genLexicon(const DecodeParams& P, const Ws& w, const FrameCtx& f,
                         unsigned long long& bestKey, double preThr, int nItems, const CutBins& cb) {
  constexpr bool SLIM = MODE == 1;
  auto countOnly = [&](bool on, double sc) {
    if (on && sc >= preThr) {
      atomAdd32(&w.hist[FLTX_HB(cutBinOf(P, cb, sc))], 1u);
      const unsigned long long sk = f64Key(sc);
      bestKey = sk > bestKey ? sk : bestKey; /* the frame's best is known after this pass */
    }
  };
  auto aboveCut = [&](bool on, double sc) {
    if (on && sc >= preThr && cutBinOf(P, cb, sc) > cb.bM) {
      w.sc[SC_CUT] = 1; /* left out: noted for the exactness check */
      return false;
    }
    return on;
  };
  const int per = f.nTok + 2;
  /* (hypothesis, token) items: the nItems existing children listed in itemList
   * (nItems >= 0), or the full nBeam x nTok grid; the 2 stay / blank items per
   * hypothesis ride along in the first round(s) */
  constexpr bool listed = LISTED;
  const int total = listed ? nItems : f.nBeam * f.nTok;
  const int W = (int)blockDim.x;
  const bool ctc = P.criterion == 1;
  const bool hasUnk = P.unkScore > -__builtin_huge_val();
  int rounds = (total + W - 1) / W;
  const int stayRounds = (2 * f.nBeam + W - 1) / W;
  rounds = rounds > stayRounds ? rounds : stayRounds;
  /* the trie gather of round it + 1 is issued before round it's candidates are
   * pushed, so its ~2k-clock HBM latency overlaps the appends instead of
   * stalling every round */
  uint4 edPre = make_uint4(0xFFFFFFFFu, 0u, 0u, 0u);
  {
    const int i0 = (int)threadIdx.x;
    if (i0 < total) {
      int h0, n0;
      if (listed) {
        const uint32_t code = itemCode(P, w, i0);
        h0 = (int)(code >> 6);
        n0 = (int)(code & 63u);
      } else {
        h0 = i0 / f.nTok;
        const int r0 = i0 - h0 * f.nTok;
        n0 = (f.nTok == P.N) ? r0 : w.tokIdx[r0];
      }
      edPre = ((const uint4*)P.trieEdge)[(size_t)w.bLex[(f.cur) * P.K + h0] * P.N + n0];
    }
  }
  for (int it = 0; it < rounds; ++it) {
    const int i = it * W + (int)threadIdx.x;
    const bool valid = i < total;
    const uint4 edNow = edPre;
    {
      const int i1 = i + W;
      if (i1 < total) {
        int h1, n1;
        if (listed) {
          const uint32_t code = itemCode(P, w, i1);
          h1 = (int)(code >> 6);
          n1 = (int)(code & 63u);
        } else {
          h1 = i1 / f.nTok;
          const int r1 = i1 - h1 * f.nTok;
          n1 = (f.nTok == P.N) ? r1 : w.tokIdx[r1];
        }
        edPre = ((const uint4*)P.trieEdge)[(size_t)w.bLex[(f.cur) * P.K + h1] * P.N + n1];
      }
    }
    /* up to 1 extend + 6 labels (or 1 unk) candidates per item, pushed in
     * lock-step so the wave-aggregated append stays convergent */
    bool cExt = false, cStay = false;
    int nLab = 0, labOff = 0, lab0 = -1;
    bool cUnk = false;
    int h = 0, n = 0;
    uint32_t childId = 0;
    double base = 0, amDelta = 0;
    float lmTok = 0.0f, lexMax = 0.0f, childMax = 0.0f;
    uint32_t sid = 0, spar = 0, lexId = 0;
    uint32_t tokKp = 0, tokKe = 0; /* token LM: the key of the state lm->score(state, n) leads to (:82-86) */
    int32_t sedge = 0;
    uint32_t ordI = 0;
    if (valid) { /* (1) children, :62-165 */
      int r;
      if (listed) {
        const uint32_t code = itemCode(P, w, i);
        h = (int)(code >> 6);
        n = (int)(code & 63u);
        r = (f.nTok == P.N) ? n : (int)w.tokPos[n];
      } else {
        h = i / f.nTok;
        r = i - h * f.nTok;
        n = (f.nTok == P.N) ? r : w.tokIdx[r];
      }
      ordI = (uint32_t)(h * per + r);
      const uint32_t tp = w.bTokPb[(f.cur) * P.K + h];
      const int prevTok = (int)(tp & 0x7FFFFFFFu);
      const bool prevBlank = (tp & kPrevBlank) != 0;
      lexId = w.bLex[(f.cur) * P.K + h];
      sid = w.bState[(f.cur) * P.K + h];
      spar = w.bSPar[(f.cur) * P.K + h];
      sedge = w.bSEdge[(f.cur) * P.K + h];
      const bool atRoot = lexId == 0u;
      const double hs = w.bScore[(f.cur) * P.K + h];
      TrieEdge ed; /* one 16-byte gather, issued a round ahead */
      ed.child = (int32_t)edNow.x;
      ed.childMax = __uint_as_float(edNow.y);
      ed.label0 = (int32_t)edNow.z;
      ed.meta = edNow.w;
      if (ed.child >= 0) {
        childId = (uint32_t)ed.child;
        lexMax = w.bLexMax[(f.cur) * P.K + h]; /* :58-59 */
        childMax = ed.childMax;
        amDelta = (double)f.e[n];
        if (f.useTrans) {
          amDelta += (double)P.transitions[(size_t)n * P.N + prevTok];
        }
        base = hs + amDelta;
        if (n == P.sil) {
          base += P.silScore;
        }
        if (P.isLmToken) {
          lmTok = lmAsk(P, w, f.b, sid, n, tokKp, tokKe); /* :82-86 */
          f.nScored += (P.lmKind != 0 && MODE != 3) ? 1u : 0u;
        }
        const int nl = (int)(ed.meta & 7u);
        cExt = (!ctc || prevBlank || n != prevTok) && (ed.meta & 8u) != 0; /* :89-91 */
        if (!(atRoot && prevTok == n)) { /* :114-122 */
          nLab = nl;
          labOff = (int)(ed.meta >> 4);
          lab0 = ed.label0;
        }
        cUnk = nl == 0 && hasUnk; /* :145 */
      }
    }
    /* (2) same lexicon node, :168-194, and (3) blank, :197-213: item j of this round */
    bool stayBlank = false;
    int hS = 0, nS = 0;
    double baseS = 0;
    uint32_t sparS = 0, lexS = 0, ordS = 0;
    int32_t sedgeS = 0;
    if (i < 2 * f.nBeam) {
      hS = i >> 1;
      const uint32_t tp = w.bTokPb[(f.cur) * P.K + hS];
      const int prevTok = (int)(tp & 0x7FFFFFFFu);
      const bool prevBlank = (tp & kPrevBlank) != 0;
      lexS = w.bLex[(f.cur) * P.K + hS];
      sparS = w.bSPar[(f.cur) * P.K + hS];
      sedgeS = w.bSEdge[(f.cur) * P.K + hS];
      const bool atRoot = lexS == 0u;
      const double hs = w.bScore[(f.cur) * P.K + hS];
      if ((i & 1) == 0) {
        ordS = (uint32_t)(hS * per + f.nTok);
        if (!ctc || !prevBlank || atRoot) {
          cStay = true;
          nS = atRoot ? P.sil : prevTok;
          double ad = (double)f.e[nS];
          if (f.useTrans) {
            ad += (double)P.transitions[(size_t)nS * P.N + prevTok];
          }
          baseS = hs + ad;
          if (nS == P.sil) {
            baseS += P.silScore;
          }
        }
      } else if (ctc) {
        ordS = (uint32_t)(hS * per + f.nTok + 1);
        cStay = true;
        stayBlank = true;
        nS = P.blank;
        baseS = hs + (double)f.e[nS];
      }
    }
    const uint32_t ordBase = ordI << 3;
    /* (1a) extend into the child node */
    {
      float l = P.isLmToken ? lmTok : (childMax - lexMax); /* float subtraction, :94 */
      double sc = base + P.lmWeight * (double)l;
      uint32_t kp = P.isLmToken ? tokKp : spar;
      uint32_t ke = P.isLmToken ? tokKe : (uint32_t)sedge;
      uint32_t src = (uint32_t)h | (P.isLmToken ? kNewState : 0u) | kExtend;
      if constexpr (SLIM) {
        pushSlim(P, w, cExt, sc, ordBase, P.isLmToken ? l : childMax, (int32_t)childId, bestKey, preThr);
      } else if constexpr (MODE == 2) {
        countOnly(cExt, sc);
      } else {
        pushCandidate(P, w, MODE == 3 ? aboveCut(cExt, sc) : cExt, sc, kp, ke, childId, (uint32_t)n, src,
                      (int32_t)__float_as_uint(childMax), l, ordBase, bestKey, preThr);
      }
    }
    /* (1b) word ends: one candidate per label of the child */
    for (int j = 0; j < 6; ++j) {
      const bool on = j < nLab;
      if (waveBallot(on) == 0ull) {
        break;
      }
      double sc = 0;
      float l = 0.0f;
      int label = -1;
      uint32_t kp = 0, ke = 0;
      if (on) {
        label = j == 0 ? lab0 : P.trieLabels[labOff + j];
        if (!P.isLmToken) {
          l = lmAsk(P, w, f.b, sid, label, kp, ke) - lexMax; /* float subtraction, :127 */
          f.nScored += (P.lmKind != 0 && MODE != 3) ? 1u : 0u;
        } else {
          l = lmTok;
          kp = tokKp;
          ke = tokKe;
        }
        sc = base + P.lmWeight * (double)l + P.wordScore;
      }
      if constexpr (SLIM) {
        pushSlim(P, w, on, sc, ordBase + 1 + (uint32_t)j, l, label, bestKey, preThr);
      } else if constexpr (MODE == 2) {
        countOnly(on, sc);
      } else {
        pushCandidate(P, w, MODE == 3 ? aboveCut(on, sc) : on, sc, kp, ke, 0u, (uint32_t)n, (uint32_t)h | kNewState,
                      label, l, ordBase + 1 + (uint32_t)j, bestKey, preThr);
      }
    }
    /* (1c) unknown word */
    if (waveBallot(cUnk) != 0ull) {
      float l = 0.0f;
      uint32_t kp = tokKp, ke = tokKe;
      if (cUnk) {
        if (!P.isLmToken) {
          l = lmAsk(P, w, f.b, sid, P.unk, kp, ke) - lexMax;
          f.nScored += (P.lmKind != 0 && MODE != 3) ? 1u : 0u;
        } else {
          l = lmTok;
        }
      }
      double sc = base + P.lmWeight * (double)l + P.unkScore;
      if constexpr (SLIM) {
        pushSlim(P, w, cUnk, sc, ordBase + 7, l, P.unk, bestKey, preThr);
      } else if constexpr (MODE == 2) {
        countOnly(cUnk, sc);
      } else {
        pushCandidate(P, w, MODE == 3 ? aboveCut(cUnk, sc) : cUnk, sc, kp, ke, 0u, (uint32_t)n,
                      (uint32_t)h | kNewState, P.unk, l, ordBase + 7, bestKey, preThr);
      }
    }
    /* (2)/(3) stay / blank keep state and node */
    if constexpr (SLIM) {
      pushSlim(P, w, cStay, baseS, ordS << 3, 0.0f, -1, bestKey, preThr);
    } else if constexpr (MODE == 2) {
      countOnly(cStay, baseS);
    } else {
      pushCandidate(P, w, MODE == 3 ? aboveCut(cStay, baseS) : cStay, baseS, sparS, (uint32_t)sedgeS, lexS,
                    (uint32_t)nS | (stayBlank ? kPrevBlank : 0u), (uint32_t)hS, -1, 0.0f, ordS << 3,
                    bestKey, preThr);
    }
  }
}

/* Second pass of the lexicon decoder's cut-off generation: the slim records
 * whose score reaches the cut (bin <= bM of the linear histogram over
 * [thr, best]) are rebuilt into full candidate records and merged.  The order
 * value tells which candidate of which (hypothesis, token) item a record is;
 * its fields are recomputed exactly as genLexicon does, the score is the one
 * the score pass stored. */
FLTX_DEV void genLexiconSelected(const DecodeParams& P, const Ws& w, const FrameCtx& f, int nSlim, int bM,
                                 double best, double thr, double scale) {
  const int W = (int)blockDim.x;
  const int per = f.nTok + 2;
  const bool ctc = P.criterion == 1;
  const int lane = laneId();
  /* compact the indices of the records that make the cut (wave-aggregated
   * append into lead[]), so that rebuilding them takes ceil(#kept / W) rounds
   * of trie gathers, not ceil(nSlim / W) */
  {
    const int rounds = (nSlim + W - 1) / W;
    for (int it = 0; it < rounds; ++it) {
      const int i = it * W + (int)threadIdx.x;
      bool on = i < nSlim;
      if (on) {
        const double sc = w.zScore[i];
        on = sc >= thr;
        if (on) {
          const double x = (best - sc) * scale;
          int bin = (x < (double)P.NB) ? (int)x : P.NB - 1;
          bin = bin < 0 ? 0 : bin;
          if (bin > bM) { /* a candidate that would pass the threshold is left out: noted for the check */
            on = false;
            w.sc[SC_CUT] = 1;
          }
        }
      }
      const unsigned long long m = waveBallot(on);
      if (m != 0ull) {
        const int leader = __builtin_ctzll(m);
        uint32_t base = 0;
        if (lane == leader) {
          base = atomAdd32((uint32_t*)&w.sc[SC_NSMALL], (uint32_t)popc64(m));
        }
        base = waveShfl32(base, leader);
        if (on) {
          const uint32_t pos = base + (uint32_t)popc64(m & ((1ull << lane) - 1ull));
          if (pos < (uint32_t)P.CAP) {
            w.lead[pos] = (uint32_t)i;
          } else {
            atomOr32((uint32_t*)&w.sc[SC_STATUS], ST_CAND_OVERFLOW);
          }
        }
      }
    }
  }
  wsBarrier(P);
  int nSel = w.sc[SC_NSMALL];
  nSel = nSel > P.CAP ? P.CAP : nSel;
  const int rounds = (nSel + W - 1) / W;
  unsigned long long dummy = 0ull;
  for (int it = 0; it < rounds; ++it) {
    const int i = it * W + (int)threadIdx.x;
    const bool on = i < nSel;
    double sc = 0;
    uint32_t ord = 0, kp = 0, ke = 0, klex = 0, ktp = 0, src = 0;
    int32_t aux = -1, zaux = -1;
    float l = 0.0f;
    if (on) {
      const uint32_t zi = w.lead[i];
      sc = w.zScore[zi];
      ord = w.zOrd[zi];
      l = w.zLm[zi]; /* as computed by the score pass */
      zaux = w.zAux[zi];
    }
    if (on) {
      const int item = (int)(ord >> 3), sub = (int)(ord & 7u);
      const int h = item / per, r = item - h * per;
      const int o = f.cur * P.K + h;
      const uint32_t tp = w.bTokPb[o];
      const int prevTok = (int)(tp & 0x7FFFFFFFu);
      const uint32_t lexId = w.bLex[o];
      const uint32_t sid = w.bState[o], spar = w.bSPar[o];
      const int32_t sedge = w.bSEdge[o];
      if (r >= f.nTok) { /* stay (:168-194) or blank (:197-213): keep state and node */
        const bool blank = r > f.nTok;
        const int n = blank ? P.blank : (lexId == 0u ? P.sil : prevTok);
        kp = spar;
        ke = (uint32_t)sedge;
        klex = lexId;
        ktp = (uint32_t)n | (blank ? kPrevBlank : 0u);
        src = (uint32_t)h;
      } else if (!P.isLmToken) { /* word LM: everything needed is in the slim record */
        const int n = (f.nTok == P.N) ? r : w.tokIdx[r];
        ktp = (uint32_t)n;
        if (sub == 0) { /* extend into the child node (:89-112) */
          const float childMax = l;
          l = childMax - w.bLexMax[o]; /* float subtraction, :94, as in the score pass */
          kp = spar;
          ke = (uint32_t)sedge;
          klex = (uint32_t)zaux;
          src = (uint32_t)h | kExtend;
          aux = (int32_t)__float_as_uint(childMax);
        } else { /* word end (:114-143) or unknown word (:145-165): back to the root */
          ke = (uint32_t)zaux;
          kp = sid;
          klex = 0u;
          src = (uint32_t)h | kNewState;
          aux = zaux;
        }
      } else {
        const int n = (f.nTok == P.N) ? r : w.tokIdx[r];
        const TrieEdge ed = P.trieEdge[(size_t)lexId * P.N + n];
        ktp = (uint32_t)n;
        if (sub == 0) { /* extend into the child node (:89-112) */
          kp = P.isLmToken ? sid : spar;
          ke = P.isLmToken ? (uint32_t)n : (uint32_t)sedge;
          klex = (uint32_t)ed.child;
          src = (uint32_t)h | (P.isLmToken ? kNewState : 0u) | kExtend;
          aux = (int32_t)__float_as_uint(ed.childMax);
        } else { /* word end (:114-143) or unknown word (:145-165): back to the root */
          const int j = sub - 1;
          const int label = sub == 7 ? P.unk : (j == 0 ? ed.label0 : P.trieLabels[(int)(ed.meta >> 4) + j]);
          ke = !P.isLmToken ? (uint32_t)label : (uint32_t)n;
          kp = sid;
          klex = 0u;
          src = (uint32_t)h | kNewState;
          aux = label;
        }
      }
    }
    pushCandidate(P, w, on, sc, kp, ke, klex, ktp, src, aux, l, ord, dummy);
  }
  (void)ctc;
}

/* decodeEnd candidates (LexiconFreeDecoder.cpp:127-146, LexiconDecoder.cpp:231-262) */
FLTX_DEV void genEnd(const DecodeParams& P, const Ws& w, const FrameCtx& f,
                     unsigned long long& bestKey) {
  const int W = (int)blockDim.x;
  /* nice ending: only root-node hypotheses finish if any exists */
  bool nice = false;
  if (P.kind == 1) {
    for (int h = 0; h < f.nBeam; ++h) {
      if (w.bLex[(f.cur) * P.K + h] == 0u) {
        nice = true;
        break;
      }
    }
  }
  const bool finishChild = P.lmKind != 0; /* KenLM::finish -> child(-1); ZeroLM -> same */
  const int rounds = (f.nBeam + W - 1) / W;
  for (int it = 0; it < rounds; ++it) {
    const int h = it * W + (int)threadIdx.x;
    bool valid = h < f.nBeam;
    double sc = 0;
    float l = 0.0f;
    uint32_t kp = 0, ke = 0, lex = 0, src = 0;
    if (valid) {
      lex = w.bLex[(f.cur) * P.K + h];
      if (P.kind == 1 && nice && lex != 0u) {
        valid = false;
      }
    }
    if (valid) {
      const uint32_t sid = w.bState[(f.cur) * P.K + h];
      if (P.lmKind == 2) { /* the user's LM::finish: score and the state it returns (LM.h:76) */
        l = lmAsk(P, w, f.b, sid, kFinishEdge, kp, ke);
      } else {
        l = lmFinishDev(P, f.b, sid);
        kp = sid;
        ke = (uint32_t)kFinishEdge;
      }
      sc = w.bScore[(f.cur) * P.K + h] + P.lmWeight * (double)l;
      if (finishChild) {
        src = (uint32_t)h | kNewState;
      } else {
        kp = w.bSPar[(f.cur) * P.K + h];
        ke = (uint32_t)w.bSEdge[(f.cur) * P.K + h];
        src = (uint32_t)h;
      }
    }
    pushCandidate(P, w, valid, sc, kp, ke, lex, (uint32_t)P.sil, src, -1, l, (uint32_t)h, bestKey);
  }
}

/* ------------------------------------------------------------------------ */
/* fold hash chains into groups (candidatesStore steps 1-2, Utils.h:160-198)  */
/* ------------------------------------------------------------------------ */
FLTX_DEV void foldGroups(const DecodeParams& P, const Ws& w, double thr, int nCand) {
  const int W = (int)blockDim.x;
  /* one thread per hash chain: the chain's first inserter (the only member
   * whose next pointer is empty) recorded its slot in small[] */
  const int rounds = (nCand + W - 1) / W;
  for (int it = 0; it < rounds; ++it) {
    const int ci = it * W + (int)threadIdx.x;
    uint32_t bestIdx = kEmpty;
    double acc = 0;
    if (ci < nCand && w.cNext[ci] == kEmpty) {
      const uint32_t hd = wsLoadAtomic32(P, &w.head[w.small[ci]]);
      if (hd != kEmpty) {
        /* best member: highest score, ties to the earliest generated */
        uint32_t bestOrd = 0;
        for (uint32_t m = hd; m != kEmpty; m = w.cNext[m]) {
          const double sc = w.cScore[m];
          if (!(sc >= thr)) {
            continue;
          }
          const uint32_t o = w.cOrd[m];
          if (bestIdx == kEmpty || sc > acc || (sc == acc && o < bestOrd)) {
            bestIdx = m;
            acc = sc;
            bestOrd = o;
          }
        }
        w.head[w.small[ci]] = kEmpty; /* leave the table empty for the next frame */
        if (bestIdx != kEmpty && P.logAdd) {
          /* Utils.h:186-193: members in descending score order, folded
           * left to right: acc = max + log1p(exp(min - max)) */
          double prevS = acc;
          uint32_t prevO = bestOrd;
          for (;;) {
            uint32_t nxt = kEmpty;
            double ns = 0;
            uint32_t no = 0;
            for (uint32_t m = hd; m != kEmpty; m = w.cNext[m]) {
              const double sc = w.cScore[m];
              if (!(sc >= thr)) {
                continue;
              }
              const uint32_t o = w.cOrd[m];
              /* strictly after (prevS, prevO) in (score desc, ord asc) order */
              if (!(sc < prevS || (sc == prevS && o > prevO))) {
                continue;
              }
              if (nxt == kEmpty || sc > ns || (sc == ns && o < no)) {
                nxt = m;
                ns = sc;
                no = o;
              }
            }
            if (nxt == kEmpty) {
              break;
            }
            const double mx = acc > ns ? acc : ns;
            const double mn = acc > ns ? ns : acc;
            acc = mx + log1p(exp(mn - mx));
            prevS = ns;
            prevO = no;
          }
        }
      }
    }
    /* compact the leaders */
    const bool isLead = bestIdx != kEmpty;
    const unsigned long long m = waveBallot(isLead);
    if (m != 0ull) {
      const int lane = laneId();
      const int leader = __builtin_ctzll(m);
      uint32_t base = 0;
      if (lane == leader) {
        base = atomAdd32((uint32_t*)&w.sc[SC_NLEAD], (uint32_t)popc64(m));
      }
      base = waveShfl32(base, leader);
      if (isLead) {
        const uint32_t li = base + (uint32_t)popc64(m & ((1ull << lane) - 1ull));
        w.lead[li] = bestIdx;
        w.cScore[bestIdx] = acc; /* group score lives in the leader's record */
      }
    }
  }
}

/* ------------------------------------------------------------------------ */
/* exact top-K of the group leaders (candidatesStore step 3, Utils.h:200-220) */
/* Order: score descending, ties by generation order.  Histogram select over  */
/* [lo, hi] narrowed until the boundary bin is small, then an exact rank.     */
/* ------------------------------------------------------------------------ */
FLTX_DEV bool precedes(double sa, uint32_t oa, double sb, uint32_t ob) {
  return sa > sb || (sa == sb && oa < ob);
}

FLTX_DEV void selectTopK(const DecodeParams& P, const Ws& w, int nLead, int K) {
  const int W = (int)blockDim.x;
  const int tid = (int)threadIdx.x;
  if (nLead <= K) {
    for (int i = tid; i < nLead; i += W) {
      w.lstat[i] = 2;
    }
    wsBarrier(P);
    return;
  }
  for (int i = tid; i < nLead; i += W) {
    w.lstat[i] = 1;
  }
  if (tid == 0) {
    w.sc[SC_NEED] = K;
    w.sc[SC_DONE] = 0;
  }
  wsBarrier(P);
  const int SMALL = W < 256 ? 256 : W;
  for (int pass = 0; pass < 64; ++pass) {
    /* range of the active set */
    double mx = -__builtin_huge_val(), mn = __builtin_huge_val();
    int cnt = 0;
    for (int i = tid; i < nLead; i += W) {
      if (w.lstat[i] == 1) {
        const double sc = w.cScore[w.lead[i]];
        mx = sc > mx ? sc : mx;
        mn = (sc < mn && sc > -__builtin_huge_val()) ? sc : mn; /* -inf falls in the last bin */
        ++cnt;
      }
    }
    const double hi = blockMaxF64(P, mx, &w.red[0]);
    const double lo = blockMinF64(P, mn, &w.red[1]);
    int active;
    blockExclusiveScan(P, cnt, w.wtmp, &active);
    const int need = w.sc[SC_NEED];
    const double scale = (double)P.NB / (hi - lo);
    if (active <= SMALL || !(hi > lo) || !(scale > 0.0) || !(scale < 1e300)) {
      /* exact rank inside the active set */
      if (tid == 0) {
        w.sc[SC_NSMALL] = 0;
      }
      wsBarrier(P);
      for (int i = tid; i < nLead; i += W) {
        if (w.lstat[i] == 1) {
          const uint32_t p = atomAdd32((uint32_t*)&w.sc[SC_NSMALL], 1u);
          w.small[p] = (uint32_t)i;
        }
      }
      wsBarrier(P);
      for (int j = tid; j < active; j += W) {
        const uint32_t li = w.small[j];
        const uint32_t c = w.lead[li];
        const double sc = w.cScore[c];
        const uint32_t o = w.cOrd[c];
        int rank = 0;
        for (int q = 0; q < active; ++q) {
          const uint32_t c2 = w.lead[w.small[q]];
          rank += precedes(w.cScore[c2], w.cOrd[c2], sc, o) ? 1 : 0;
        }
        w.lstat[li] = rank < need ? 2 : 0;
      }
      wsBarrier(P);
      return;
    }
    /* histogram pass: bin 0 holds the highest scores */
    for (int i = tid; i < P.NB; i += W) {
      w.hist[i] = 0;
    }
    wsBarrier(P);
    for (int i = tid; i < nLead; i += W) {
      if (w.lstat[i] == 1) {
        const double sc = w.cScore[w.lead[i]];
        const double x = (hi - sc) * scale;
        int bin = (x < (double)P.NB) ? (int)x : P.NB - 1; /* also catches inf / NaN */
        bin = bin < 0 ? 0 : bin;
        w.lbin[i] = (uint16_t)bin;
        atomAdd32(&w.hist[bin], 1u);
      }
    }
    wsBarrier(P);
    /* locate the bin where the cumulative count reaches `need` */
    const int per = (P.NB + W - 1) / W;
    int mine = 0;
    for (int q = 0; q < per; ++q) {
      const int bi = tid * per + q;
      if (bi < P.NB) {
        mine += (int)w.hist[bi];
      }
    }
    int tot;
    const int before = blockExclusiveScan(P, mine, w.wtmp, &tot);
    if (before < need && before + mine >= need) {
      int cum = before;
      for (int q = 0; q < per; ++q) {
        const int bi = tid * per + q;
        const int c = (int)w.hist[bi];
        if (cum + c >= need) {
          w.sc[SC_BSTAR] = bi;
          w.sc[SC_CUM] = cum;
          w.sc[SC_M] = c;
          break;
        }
        cum += c;
      }
    }
    wsBarrier(P);
    const int bstar = w.sc[SC_BSTAR], cum = w.sc[SC_CUM], mcnt = w.sc[SC_M];
    for (int i = tid; i < nLead; i += W) {
      if (w.lstat[i] == 1) {
        const int bin = (int)w.lbin[i];
        if (bin < bstar) {
          w.lstat[i] = 2;
        } else if (bin > bstar) {
          w.lstat[i] = 0;
        } else if (cum + mcnt == need) {
          w.lstat[i] = 2;
        }
      }
    }
    wsBarrier(P);
    if (cum + mcnt == need) {
      return;
    }
    if (tid == 0) {
      w.sc[SC_NEED] = need - cum;
    }
    wsBarrier(P);
  }
  /* unreachable for finite inputs: each pass narrows [lo, hi] */
  if (tid == 0) {
    atomOr32((uint32_t*)&w.sc[SC_STATUS], ST_SELECT_FALLBACK);
  }
  wsBarrier(P);
}

/* Fast path of the prune: ONE histogram pass over the leaders locates the bin
 * holding the K-th best score; every leader in that bin or a better one goes
 * to a compact short-list (K + a few entries, contiguous u64 keys) on which an
 * exact rank by (score desc, generation order) is computed.  rank < K survive
 * and the rank IS the slot in the next beam.  Falls back to the iterative
 * selectTopK() when the short-list would not fit (degenerate distributions).
 * Returns the number of survivors; w.surv[r] = candidate index of rank r. */
/* Histogram select with the workspace in LDS, three barriers (the scheme of
 * fltx_lane.h): 512 bins over [thr, best] in two monotone segments; every wave
 * scans the counts itself (8 bins per lane) and keeps its own copy of the
 * prefixes, so there is no "wave 0 works, the others wait" step and no barrier
 * between prefix and scatter; counting sort; rank inside the bin four entries
 * per LDS round trip.  hist[0..512) and tick[] are zero on entry (runFrame)
 * and left dirty.  Returns false (nothing written but hist) when the score
 * range is degenerate or the K-th bin overflows the short-list: the caller
 * then takes the general path. */
FLTX_DEV bool selectFast(const DecodeParams& P, const Ws& w, int nLead, int K, double best, double thr,
                         double spread, int& Lout) {
  constexpr int NB2 = 512;
  const int W = (int)blockDim.x, tid = (int)threadIdx.x;
  const int lane = laneId(), wave = waveId();
  const double range = best - thr;
  const int NF = (NB2 * 3) / 4;
  double cut = spread * 1.25 + 1e-3;
  cut = cut > range * 0.125 ? cut : range * 0.125;
  cut = cut < range * 0.9 ? cut : range * 0.9;
  const double sF = (double)NF / cut, sC = (double)(NB2 - NF) / (range - cut);
  if (!(range > 0.0) || !(range < 1e6) || !(sF > 0.0) || !(sF < 1e300) || !(sC > 0.0) || !(sC < 1e300)) {
    return false; /* uniform: every thread sees the same best / thr / spread */
  }
  for (int i = tid; i < nLead; i += W) {
    const double d = best - w.cScore[w.lead[i]];
    const double x = d < cut ? d * sF : (double)NF + (d - cut) * sC;
    int bin = (x < (double)NB2) ? (int)x : NB2 - 1;
    bin = bin < 0 ? 0 : bin;
    w.lbin[i] = (uint16_t)bin;
    atomAdd32(&w.hist[bin], 1u);
  }
  wsBarrier(P); /* 1 */
  const uint4 cq0 = ((const uint4*)w.hist)[2 * lane], cq1 = ((const uint4*)w.hist)[2 * lane + 1];
  const int mineCnt = (int)(cq0.x + cq0.y + cq0.z + cq0.w + cq1.x + cq1.y + cq1.z + cq1.w);
  const int inc = waveInclusiveScan(mineCnt);
  int pre[9];
  pre[0] = inc - mineCnt;
  pre[1] = pre[0] + (int)cq0.x;
  pre[2] = pre[1] + (int)cq0.y;
  pre[3] = pre[2] + (int)cq0.z;
  pre[4] = pre[3] + (int)cq0.w;
  pre[5] = pre[4] + (int)cq1.x;
  pre[6] = pre[5] + (int)cq1.y;
  pre[7] = pre[6] + (int)cq1.z;
  pre[8] = inc;
  const int total = (int)waveReadLane32((uint32_t)inc, 63);
  const unsigned long long cm = waveBallot(pre[0] < K && inc >= K);
  int bstar = NB2 - 1, L = total;
  {
    int q = 7, cumAt = inc;
#pragma unroll
    for (int i = 6; i >= 0; --i) {
      const bool hit = pre[i + 1] >= K;
      q = hit ? i : q;
      cumAt = hit ? pre[i + 1] : cumAt;
    }
    const int X = cm ? __builtin_ctzll(cm) : 0;
    const uint32_t both = waveReadLane32((uint32_t)(8 * lane + q) | ((uint32_t)cumAt << 16), X);
    if (cm) {
      bstar = (int)(both & 0xFFFFu);
      L = (int)(both >> 16);
    }
  }
  if (L > P.SCAP || total > 65535) {
    return false; /* uniform; the general path recomputes from lead[] */
  }
  ((uint4*)(w.wcum + wave * (NB2 / 2)))[lane] =
      make_uint4((uint32_t)pre[0] | ((uint32_t)pre[1] << 16), (uint32_t)pre[2] | ((uint32_t)pre[3] << 16),
                 (uint32_t)pre[4] | ((uint32_t)pre[5] << 16), (uint32_t)pre[6] | ((uint32_t)pre[7] << 16));
  waveSync();
  for (int i = tid; i < nLead; i += W) {
    const int bin = (int)w.lbin[i];
    if (bin <= bstar) {
      const uint32_t lo = ((const uint16_t*)(w.wcum + wave * (NB2 / 2)))[bin];
      const uint32_t cnt = w.hist[bin];
      const uint32_t p = lo + atomAdd32(&w.tick[bin], 1u);
      const uint32_t c = w.lead[i];
      const unsigned long long key = f64Key(w.cScore[c]);
      w.sEnt[p] = make_uint4((uint32_t)key, (uint32_t)(key >> 32), w.cOrd[c], lo | ((lo + cnt) << 16));
      w.sIdx[p] = c;
    }
  }
  wsBarrier(P); /* 2 */
  for (int p = tid; p < L; p += W) {
    const uint4 me = w.sEnt[p];
    const unsigned long long k = ((unsigned long long)me.y << 32) | me.x;
    const int lo = (int)(me.w & 0xFFFFu), hi = (int)(me.w >> 16);
    int rank = lo;
    for (int q = lo; q < hi; q += 4) {
      uint4 e[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        e[u] = w.sEnt[q + u < L ? q + u : L - 1];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const unsigned long long k2 = ((unsigned long long)e[u].y << 32) | e[u].x;
        rank += (q + u < hi && (k2 > k || (k2 == k && e[u].z < me.z))) ? 1 : 0;
      }
    }
    if (rank < K) {
      w.surv[rank] = w.sIdx[p];
    }
  }
  wsBarrier(P); /* 3 */
  Lout = L;
  return true;
}

FLTX_DEV int selectAndRank(const DecodeParams& P, const Ws& w, int nLead, int K, double best, double thr,
                           double spread) {
  const int W = (int)blockDim.x;
  const int tid = (int)threadIdx.x;
  if (nLead <= 0) {
    return 0;
  }
  const int nS = nLead < K ? nLead : K;
  int L = 0;
  const int direct = P.SCAP < 128 ? P.SCAP : 128;
  bool slow = false;
  if (nLead <= direct || nLead <= K) { /* everything survives or the list is tiny: rank it directly */
    for (int i = tid; i < nLead; i += W) {
      const uint32_t c = w.lead[i];
      w.sKey[i] = f64Key(w.cScore[c]);
      w.sOrd[i] = w.cOrd[c];
      w.sIdx[i] = c;
    }
    L = nLead;
    wsBarrier(P);
  } else if ((P.gws == nullptr || P.hotLevel > 0) && selectFast(P, w, nLead, K, best, thr, spread, L)) {
    return nS;
  } else {
    /* Bins over [best - beamThreshold, best]: the leaders already passed the
     * threshold (foldGroups), so no pass over them is needed to find the range.
     * Two monotone segments as in fltx_lean.h: 3/4 of the bins cover a little
     * more than the current beam's own spread, where the next beam's K-th best
     * is expected; the rest of the range shares the last quarter. */
    if (tid == 0) {
      w.sc[SC_NSMALL] = 0;
      w.sc[SC_BSTAR] = P.NB - 1;
      w.sc[SC_CUM] = 0;
    }
    for (int i = tid; i < P.NB; i += W) {
      w.hist[FLTX_HB(i)] = 0;
    }
    wsBarrier(P);
    const double range = best - thr;
    const int NF = (P.NB * 3) / 4;
    double cut = spread * 1.25 + 1e-3;
    cut = cut > range * 0.125 ? cut : range * 0.125;
    cut = cut < range * 0.9 ? cut : range * 0.9;
    const double sF = (double)NF / cut, sC = (double)(P.NB - NF) / (range - cut);
    if (!(range > 0.0) || !(range < 1e6) || !(sF > 0.0) || !(sF < 1e300) || !(sC > 0.0) || !(sC < 1e300) ||
        P.NB != 1024) {
      slow = true;
    } else {
      for (int i = tid; i < nLead; i += W) {
        const double d = best - w.cScore[w.lead[i]];
        const double x = d < cut ? d * sF : (double)NF + (d - cut) * sC;
        int bin = (x < (double)P.NB) ? (int)x : P.NB - 1;
        bin = bin < 0 ? 0 : bin;
        w.lbin[i] = (uint16_t)bin;
        atomAdd32(&w.hist[FLTX_HB(bin)], 1u);
      }
      wsBarrier(P);
      /* wave 0: counts -> exclusive prefixes up to the bin of the K-th best
       * (16 bins per lane, held in registers; see fltx_lean.h phase C) */
      if (waveId() == 0) {
        constexpr int PER = 16;
        const int lane = laneId();
        int c[PER];
        int mine = 0;
#pragma unroll
        for (int q = 0; q < PER; ++q) {
          c[q] = (int)w.hist[FLTX_HB(lane * PER + q)];
          mine += c[q];
        }
        const int inc = waveInclusiveScan(mine);
        int cum = inc - mine;
        if (cum < K) {
          bool done = false;
#pragma unroll
          for (int q = 0; q < PER; ++q) {
            w.hcum[FLTX_HB(lane * PER + q)] = (uint32_t)cum;
            cum += c[q];
            if (!done && cum >= K) {
              w.sc[SC_BSTAR] = lane * PER + q;
              w.sc[SC_CUM] = cum;
              done = true;
            }
          }
          if (lane < 63) {
            w.hcum[FLTX_HB((lane + 1) * PER)] = (uint32_t)cum;
          }
        }
      }
      wsBarrier(P);
      const int bstar = w.sc[SC_BSTAR];
      L = w.sc[SC_CUM];
      if (L > P.SCAP) {
        slow = true;
      } else {
        /* counting sort by bin: position = better-bin count + ticket in the bin */
        for (int i = tid; i < nLead; i += W) {
          const int bin = (int)w.lbin[i];
          if (bin <= bstar) {
            const int hb = FLTX_HB(bin);
            const uint32_t p = w.hcum[hb] + (atomAdd32(&w.hist[hb], 0xFFFFFFFFu) - 1u);
            const uint32_t c = w.lead[i];
            const unsigned long long key = f64Key(w.cScore[c]);
            w.sEnt[p] = make_uint4((uint32_t)key, (uint32_t)(key >> 32), w.cOrd[c], (uint32_t)bin);
            w.sIdx[p] = c;
          }
        }
        wsBarrier(P);
        for (int p = tid; p < L; p += W) {
          const uint4 me = w.sEnt[p];
          const unsigned long long k = ((unsigned long long)me.y << 32) | me.x;
          const int bin = (int)me.w;
          const int lo2 = (int)w.hcum[FLTX_HB(bin)];
          const int hi2 = bin >= bstar ? L : (int)w.hcum[FLTX_HB(bin + 1)];
          int rank = lo2;
          for (int q = lo2; q < hi2; q += 4) { /* four entries per LDS round trip */
            uint4 e[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              e[u] = w.sEnt[q + u < L ? q + u : L - 1];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const unsigned long long k2 = ((unsigned long long)e[u].y << 32) | e[u].x;
              rank += (q + u < hi2 && (k2 > k || (k2 == k && e[u].z < me.z))) ? 1 : 0;
            }
          }
          if (rank < K) {
            w.surv[rank] = w.sIdx[p];
          }
        }
        wsBarrier(P);
        return nS;
      }
    }
    if (slow) {
      selectTopK(P, w, nLead, K);
      if (tid == 0) {
        w.sc[SC_NSMALL] = 0;
      }
      wsBarrier(P);
      for (int i = tid; i < nLead; i += W) {
        if (w.lstat[i] == 2) {
          const uint32_t p = atomAdd32((uint32_t*)&w.sc[SC_NSMALL], 1u);
          const uint32_t c = w.lead[i];
          w.sKey[p] = f64Key(w.cScore[c]);
          w.sOrd[p] = w.cOrd[c];
          w.sIdx[p] = c;
        }
      }
      wsBarrier(P);
      L = nS;
    }
  }
  /* exact rank on the short-list by all pairs, the pairs spread over the whole
   * workgroup: entry j is compared with a slice of the list by each of
   * W / L threads and the partial counts meet in LDS */
  const int parts = L > 0 && W / L > 1 ? W / L : 1;
  if (parts > 1) {
    for (int j = tid; j < L; j += W) {
      w.small[j] = 0u;
    }
    wsBarrier(P);
    const int per = (L + parts - 1) / parts;
    const int j = tid % L, c = tid / L;
    if (c < parts) {
      const unsigned long long k = w.sKey[j];
      const uint32_t o = w.sOrd[j];
      const int q0 = c * per;
      int q1 = q0 + per;
      q1 = q1 > L ? L : q1;
      uint32_t cnt = 0;
      for (int q = q0; q < q1; ++q) {
        const unsigned long long k2 = w.sKey[q];
        const uint32_t o2 = w.sOrd[q];
        cnt += (k2 > k || (k2 == k && o2 < o)) ? 1u : 0u;
      }
      if (cnt != 0u) {
        atomAdd32(&w.small[j], cnt);
      }
    }
    wsBarrier(P);
    for (int j2 = tid; j2 < L; j2 += W) {
      const int rank = (int)wsLoadAtomic32(P, &w.small[j2]);
      if (rank < K) {
        w.surv[rank] = w.sIdx[j2];
      }
    }
    wsBarrier(P);
    return nS;
  }
  for (int j = tid; j < L; j += W) {
    const unsigned long long k = w.sKey[j];
    const uint32_t o = w.sOrd[j];
    int rank = 0;
    for (int q = 0; q < L; ++q) {
      const unsigned long long k2 = w.sKey[q];
      const uint32_t o2 = w.sOrd[q];
      rank += (k2 > k || (k2 == k && o2 < o)) ? 1 : 0;
    }
    if (rank < K) {
      w.surv[rank] = w.sIdx[j];
    }
  }
  wsBarrier(P);
  return nS;
}

/* ------------------------------------------------------------------------ */
/* survivors -> next beam + history (candidatesStore step 4, Utils.h:222-224) */
/* The beam is kept sorted (score desc, generation order) so the layout is a   */
/* deterministic function of the inputs; decodeEnd needs the sort anyway       */
/* (returnSorted, Utils.h:213-219).                                           */
/* ------------------------------------------------------------------------ */
FLTX_DEV int buildBeam(const DecodeParams& P, const Ws& w, const FrameCtx& f, int nS,
                       int frameOut, bool isEnd) {
  const int W = (int)blockDim.x;
  const int tid = (int)threadIdx.x;
  const int nxt = f.cur ^ 1;
  const int64_t hbase = f.histBase + (int64_t)frameOut * P.K;
  for (int rank0 = 0; rank0 < nS; rank0 += W) { /* (uniform trips: stateChildIds is a wave-wide call) */
    const int rank = rank0 + tid;
    const bool on = rank < nS;
    const uint32_t c = on ? w.surv[rank] : 0u;
    const double sc = on ? w.cScore[c] : 0.0;
    const uint4 key = on ? w.cKey[c] : make_uint4(0u, 0u, 0u, 0u);
    const uint32_t src = on ? w.cSrc[c] : 0u;
    const int h = (int)(src & kSrcMask);
    const int n = (int)(key.w & 0x7FFFFFFFu);
    const float lmd = on ? w.cLm[c] : 0.0f;
    uint32_t sidIds = 0u;
    bool freshIds = false;
    if (P.stateVal != nullptr) {
      sidIds = stateChildIds(P, w, f.b, on && (src & kNewState) != 0u && P.lmKind != 2, key.x, (int32_t)key.y, f.clock,
                             freshIds);
    }
    if (!on) {
      continue;
    }
    /* emitting-model score: recompute the candidate's delta (LexiconFree:
     * transition goes into am only, :59-64; Lexicon: am == score delta) */
    double am = w.bAm[(f.cur) * P.K + h];
    if (!isEnd) {
      double d = (double)f.e[n];
      const bool isBlankCand = (key.w & kPrevBlank) != 0;
      if (f.useTrans && !(P.kind == 1 && isBlankCand)) {
        const int prevTok = (int)(w.bTokPb[(f.cur) * P.K + h] & 0x7FFFFFFFu);
        d += (double)P.transitions[(size_t)n * P.N + prevTok];
      }
      am += d;
    }
    uint32_t sid;
    if ((src & kNewState) && P.lmKind == 2) {
      sid = key.x; /* host LM: the key is the id the host gave the state */
    } else if (src & kNewState) {
      bool fresh;
      if (P.stateVal != nullptr) {
        sid = sidIds;
        fresh = freshIds;
      } else {
        sid = stateChild(P, f.b, key.x, (int32_t)key.y, (uint32_t*)&w.sc[SC_STATUS], fresh);
      }
      if (fresh && P.lmKind == 1) {
        /* materialise the n-gram context of the new state */
        const int L = P.lmOrder - 1;
        const int32_t* cin = P.stateCtx + ((size_t)f.b * P.stateCap + key.x) * L;
        int32_t* cout = P.stateCtx + ((size_t)f.b * P.stateCap + sid) * L;
        const int32_t edge = (int32_t)key.y;
        uint32_t word;
        if (P.tokLm != nullptr) { /* (dense token-LM table: the new state's context is the row the edge leads to) */
          if (L > 0) {
            cout[0] = (edge >= 0 && edge < P.N) ? P.tokLm[(size_t)(uint32_t)cin[0] * (size_t)P.tokLmStride + (size_t)edge].y : 0;
          }
        } else {
        if (edge == kFinishEdge) {
          word = (uint32_t)P.lmEos;
        } else {
          word = (edge >= 0 && edge < P.nUsr) ? (uint32_t)P.usrToLm[edge] : (uint32_t)P.lmUnk;
        }
        ngScore(P, cin, word, cout); /* reads its whole input context before it writes */
        }
      }
    } else {
      sid = w.bState[(f.cur) * P.K + h];
    }
    w.bScore[(nxt) * P.K + rank] = sc;
    w.bAm[(nxt) * P.K + rank] = am;
    w.bLm[(nxt) * P.K + rank] = w.bLm[(f.cur) * P.K + h] + (double)lmd;
    w.bState[(nxt) * P.K + rank] = sid;
    w.bSPar[(nxt) * P.K + rank] = key.x;
    w.bSEdge[(nxt) * P.K + rank] = (int32_t)key.y;
    w.bLex[(nxt) * P.K + rank] = key.z;
    /* maxScore of the slot's node: the child's when the candidate advanced in
     * the trie, the parent's when it stayed, 0 back at the root */
    w.bLexMax[(nxt) * P.K + rank] = (src & kExtend) ? __uint_as_float((uint32_t)w.cAux[c])
                                                    : (key.z == 0u ? 0.0f : w.bLexMax[(f.cur) * P.K + h]);
    w.bTokPb[(nxt) * P.K + rank] = key.w;
    P.histPT[hbase + rank] = make_int2(h, n);
    if (P.kind == 1) {
      P.histW[hbase + rank] = (src & kExtend) ? -1 : w.cAux[c];
    }
    if (P.histS) {
      double* hs = P.histS + 3 * (hbase + rank);
      hs[0] = sc;
      hs[1] = am;
      hs[2] = w.bLm[(f.cur) * P.K + h] + (double)lmd;
    }
  }
  wsBarrier(P);
  return nS;
}

/* top-beamSizeToken tokens of the row (LexiconFreeDecoder.cpp:42-51): rank by
 * emission descending, ties to the lower index. */
FLTX_DEV void tokenShortlist(const DecodeParams& P, const Ws& w, const float* e, int nTok) {
  const int W = (int)blockDim.x;
  if (P.N <= 64) {
    /* lane m holds e[m]; a wave takes token n (its emission is a uniform LDS
     * read) and the number of lanes that beat it is its rank: one ballot per
     * token, the tokens spread over the waves */
    const int m = laneId(), nW = (W + 63) >> 6;
    const float o = m < P.N ? e[m] : 0.0f;
    for (int n = waveId(); n < P.N; n += nW) {
      const float v = e[n];
      const unsigned long long beat = waveBallot(m < P.N && (o > v || (o == v && m < n)));
      const int rank = popc64(beat);
      if (m == 0 && rank < nTok) {
        w.tokIdx[rank] = n;
        if (P.itemCap) {
          w.tokPos[n] = (uint8_t)rank;
          atomOr64(&w.red[3], 1ull << n);
        }
      }
    }
    return;
  }
  for (int n = (int)threadIdx.x; n < P.N; n += W) {
    const float v = e[n];
    int rank = 0;
    for (int m = 0; m < P.N; ++m) {
      const float o = e[m];
      rank += (o > v || (o == v && m < n)) ? 1 : 0;
    }
    if (rank < nTok) {
      w.tokIdx[rank] = n;
    }
  }
}

/* one frame (or decodeEnd when isEnd): returns the new beam size */
FLTX_DEV int runFrame(const DecodeParams& P, const Ws& w, FrameCtx& f, int frameOut, bool isEnd) {
  const int W = (int)blockDim.x;
  const int tid = (int)threadIdx.x;
  /* (the merge hash is empty here: foldGroups resets the slots it used, the
   * kernel prologue cleared it once) */
  if (tid == 0) {
    w.sc[SC_NCAND] = 0;
    w.sc[SC_NLEAD] = 0;
    w.sc[SC_NSLIM] = 0;
    w.sc[SC_NSMALL] = 0;
    w.sc[SC_CUT] = 0;
    w.sc[SC_BM] = P.NB - 1;
    w.red[2] = 0ull;
  }
  if (!isEnd && P.kind == 1 && (P.CAP2 > 0 || P.cutRecompute)) {
    for (int i = tid; i < P.NB; i += W) {
      w.hist[FLTX_HB(i)] = 0;
    }
  }
  if (!isEnd && f.nTok < P.N) {
    tokenShortlist(P, w, f.e, f.nTok);
  }
  const bool listItems = !isEnd && P.kind == 1 && P.itemCap > 0;
  const int itemLimit = P.itemWide ? P.itemCap >> 1 : P.itemCap; /* items the list holds */
  if (listItems) { /* which tokens lead anywhere from each slot's trie node (one 8-byte load per slot) */
    for (int h = tid; h < f.nBeam; h += W) {
      w.bLexMask[h] = P.trieMask[w.bLex[f.cur * P.K + h]];
    }
  }
  wsBarrier(P);
  FLTX_PROF(0);
  const bool dense = !isEnd && P.kind == 0 && P.dense != 0;
  unsigned long long bestKey = 0ull;
  if (isEnd) {
    genEnd(P, w, f, bestKey);
  } else if (dense) {
    densePrepare(P, w, f);
    genLexFreeDense(P, w, f, bestKey);
  } else if (P.kind == 0) {
    genLexFree(P, w, f, bestKey);
  } else {
    /* a candidate that always exists bounds the frame's best score from below:
     * slot 0 (the best hypothesis) taking the blank (CTC, :197-213) or staying
     * in its node (ASG, :168-194) */
    double lb;
    {
      const int o = f.cur * P.K;
      const double hs = w.bScore[o];
      if (P.criterion == 1) {
        lb = hs + (double)f.e[P.blank];
      } else {
        const int prevTok = (int)(w.bTokPb[o] & 0x7FFFFFFFu);
        const int n0 = w.bLex[o] == 0u ? P.sil : prevTok;
        double ad = (double)f.e[n0];
        if (f.useTrans) {
          ad += (double)P.transitions[(size_t)n0 * P.N + prevTok];
        }
        lb = hs + ad;
        if (n0 == P.sil) {
          lb += P.silScore;
        }
      }
    }
    const double preThr = lb == lb ? lb - P.beamThreshold : -__builtin_huge_val();
    int nItems = -1;
    if (listItems) {
      /* A trie node has a child for few of the tokens: list the (hypothesis,
       * token) pairs that exist (bit masks, one prefix sum) and generate from
       * the list -- a round of W items then carries W real candidates' worth of
       * work instead of mostly empty edges. */
      const unsigned long long tokMask =
          f.nTok == P.N ? (P.N >= 64 ? ~0ull : ((1ull << P.N) - 1ull)) : w.red[3];
      int run = 0;
      for (int base = 0; base < f.nBeam; base += W) {
        const int h = base + tid;
        unsigned long long m = h < f.nBeam ? (w.bLexMask[h] & tokMask) : 0ull;
        int tot = 0;
        int pos = run + blockExclusiveScan(P, popc64(m), w.wtmp, &tot);
        while (m != 0ull) { /* slot h lists its own children: a few stores per thread */
          const int n = __builtin_ctzll(m);
          m &= m - 1ull;
          if (pos < itemLimit) {
            if (P.itemWide) {
              ((uint32_t*)w.itemList)[pos] = ((uint32_t)h << 6) | (uint32_t)n;
            } else {
              w.itemList[pos] = (uint16_t)(((uint32_t)h << 6) | (uint32_t)n);
            }
          }
          ++pos;
        }
        run += tot;
      }
      wsBarrier(P);
      nItems = run > itemLimit ? itemLimit : run;
      if (tid == 0) {
        w.red[3] = 0ull; /* next frame's short-list mask starts empty */
      }
    }
    CutBins cb;
    cb.hi = lb + 64.0; /* a candidate that much above the lower bound of the best lands in bin 0 */
    cb.scale = (double)P.NB / (cb.hi - preThr);
    cb.bM = P.NB - 1;
    if (!(cb.scale > 0.0) || !(cb.scale < 1e300)) {
      cb.scale = 0.0; /* unbounded threshold: one bin, nothing is cut */
    }
    if (P.cutRecompute) {
      /* recompute form of the cut-off generation: count per score bin, find the
       * bin of the cutM-th best, generate again and keep what reaches it */
      unsigned long long bk2 = 0ull;
      if (listItems) {
        genLexicon<2, true>(P, w, f, bk2, preThr, nItems, cb);
      } else {
        genLexicon<2, false>(P, w, f, bk2, preThr, nItems, cb);
      }
      bk2 = waveMax64(bk2);
      if (laneId() == 0 && bk2 != 0ull) {
        atomMax64(&w.red[2], bk2);
      }
      wsBarrier(P);
      if (waveId() == 0) {
        constexpr int PER = 16;
        const int lane = laneId();
        int c[PER];
        int mine = 0;
#pragma unroll
        for (int q = 0; q < PER; ++q) {
          c[q] = (int)w.hist[FLTX_HB(lane * PER + q)];
          mine += c[q];
        }
        const int inc = waveInclusiveScan(mine);
        int cum = inc - mine;
        if (cum < P.cutM && inc >= P.cutM) {
          int bq = PER - 1;
          bool done = false;
#pragma unroll
          for (int q = 0; q < PER; ++q) {
            cum += c[q];
            if (!done && cum >= P.cutM) {
              bq = q;
              done = true;
            }
          }
          w.sc[SC_BM] = lane * PER + bq;
        }
      }
      wsBarrier(P);
      cb.bM = w.sc[SC_BM];
      /* the exact threshold is known now: the second generation drops what is
       * below it, and "left out by the cut" means left out although above it */
      double thr2 = preThr;
      if (w.red[2] != 0ull) {
        const double t = f64FromKey(w.red[2]) - P.beamThreshold;
        thr2 = t > thr2 ? t : thr2;
      }
      if (listItems) {
        genLexicon<3, true>(P, w, f, bestKey, thr2, nItems, cb);
      } else {
        genLexicon<3, false>(P, w, f, bestKey, thr2, nItems, cb);
      }
    } else if (P.CAP2 > 0) {
      if (listItems) {
        genLexicon<1, true>(P, w, f, bestKey, preThr, nItems, cb);
      } else {
        genLexicon<1, false>(P, w, f, bestKey, preThr, nItems, cb);
      }
    } else if (listItems) {
      genLexicon<0, true>(P, w, f, bestKey, preThr, nItems, cb);
    } else {
      genLexicon<0, false>(P, w, f, bestKey, preThr, nItems, cb);
    }
  }
  FLTX_PROF(6); /* (lexicon decoder: item list + the score pass / the one-pass generation) */
  bestKey = waveMax64(bestKey);
  if (laneId() == 0 && bestKey != 0ull) {
    atomMax64(&w.red[2], bestKey);
  }
  wsBarrier(P);
  if (w.red[2] == 0ull) {
    return 0; /* no candidate at all */
  }
  const double best = f64FromKey(w.red[2]);
  const double thr = best - P.beamThreshold; /* Utils.h:219 call sites */
  if (!isEnd && P.kind == 1 && P.CAP2 > 0) {
    /* Cut-off generation.  Every candidate has been scored; only the best cutM
     * of them (plus the rest of the last histogram bin) become records and
     * enter the merge.  Exact whenever the merge still yields K groups: a
     * candidate that is left out scores strictly below every one that is kept,
     * so it can neither lead one of the top K groups nor be the best member of
     * a kept group (max-merge).  The check after the fold flags the other case
     * and the host repeats the batch without the cut. */
    int nSlim = w.sc[SC_NSLIM];
    nSlim = nSlim > P.CAP2 ? P.CAP2 : nSlim;
    const double range = best - thr;
    const double scale = (double)P.NB / range;
    int bM = P.NB - 1;
    if (nSlim > P.cutM && range > 0.0 && scale > 0.0 && scale < 1e300) {
      for (int i = tid; i < nSlim; i += W) {
        const double sc = w.zScore[i];
        if (sc >= thr) {
          const double x = (best - sc) * scale;
          int bin = (x < (double)P.NB) ? (int)x : P.NB - 1;
          bin = bin < 0 ? 0 : bin;
          atomAdd32(&w.hist[FLTX_HB(bin)], 1u);
        }
      }
      wsBarrier(P);
      if (waveId() == 0) { /* bin in which the cutM-th best candidate lies */
        constexpr int PER = 16;
        const int lane = laneId();
        int c[PER];
        int mine = 0;
#pragma unroll
        for (int q = 0; q < PER; ++q) {
          c[q] = (int)w.hist[FLTX_HB(lane * PER + q)];
          mine += c[q];
        }
        const int inc = waveInclusiveScan(mine);
        int cum = inc - mine;
        if (cum < P.cutM && inc >= P.cutM) {
          int bq = PER - 1;
          bool done = false;
#pragma unroll
          for (int q = 0; q < PER; ++q) {
            cum += c[q];
            if (!done && cum >= P.cutM) {
              bq = q;
              done = true;
            }
          }
          w.sc[SC_BM] = lane * PER + bq;
        }
      }
      wsBarrier(P);
      bM = w.sc[SC_BM];
    }
    FLTX_PROF(7); /* (the cut: histogram of the slim scores, the bin of the cutM-th best) */
    genLexiconSelected(P, w, f, nSlim, bM, best, thr, scale);
    wsBarrier(P);
  }
  FLTX_PROF(1);
  if (dense) {
    denseLeaders(P, w, f, thr);
  } else {
    int nCand = w.sc[SC_NCAND];
    nCand = nCand > P.CAP ? P.CAP : nCand;
    foldGroups(P, w, thr, nCand);
  }
  if (P.gws == nullptr || P.hotLevel > 0) { /* selectFast expects empty counters (they are in LDS either way) */
    for (int i = tid; i < 512; i += W) {
      w.hist[i] = 0u;
      w.tick[i] = 0u;
    }
  }
  wsBarrier(P);
  FLTX_PROF(2);
  const int nLead = w.sc[SC_NLEAD];
  if (!isEnd && P.kind == 1 && (P.CAP2 > 0 || P.cutRecompute) && w.sc[SC_CUT] != 0 && nLead < P.K && tid == 0) {
    atomOr32((uint32_t*)&w.sc[SC_STATUS], ST_CUT_RETRY); /* the cut left fewer than K groups */
  }
  const double spread = f.nBeam > 0 ? w.bScore[f.cur * P.K] - w.bScore[f.cur * P.K + f.nBeam - 1] : 0.0;
  const int nS = selectAndRank(P, w, nLead, P.K, best, thr, spread);
  FLTX_PROF(3);
  const int nB = buildBeam(P, w, f, nS, frameOut, isEnd);
  FLTX_PROF(4);
  return nB;
}

#include "fltx_lean.h"
#include "fltx_lane.h"
#include "fltx_slane.h"
#include "fltx_mlane.h"
#include "fltx_wlane.h"
#include "fltx_xlane.h"
#include "fltx_ylane.h"

/* ------------------------------------------------------------------------ */
/* the decode kernel: grid = utterances, block = W threads.                  */
/* GMAX == 0: generic engine; GMAX > 0: lean lexicon-free + ZeroLM frame step */
/* with up to GMAX candidate groups per thread (fltx_lean.h).                 */
/* ------------------------------------------------------------------------ */
template <int GMAX, int GT = 0, bool LOGADD = false, bool FULLTOK = true>
FLTX_DEV void decodeUtterance(const DecodeParams& P, char* wsBase, char* hotBase = nullptr) {
  const int b = P.uttMap ? P.uttMap[blockIdx.x] : (int)blockIdx.x;
  const int W = (int)blockDim.x;
  const int tid = (int)threadIdx.x;
  Ws w;
  carveWs(w, wsBase, P.K, P.CAP, P.HS, P.NB, P.N, P.SCAP, P.dense, P.lane, P.CAP2, P.itemCap, (W + 63) >> 6,
          hotBase != nullptr ? (P.hotLevel > 1 ? P.hotLevel : 1) : 0, hotBase);
  int cur = 0;
  int nBeam, frame, total;
  if (tid == 0) {
    w.sc[SC_STATUS] = 0;
  }
  if constexpr (GT == 0) { /* the merge hash starts empty; foldGroups keeps it so */
    for (int i = tid; i < P.HS; i += W) {
      w.head[i] = kEmpty;
    }
    if (tid == 0) {
      w.red[3] = 0ull;
    }
  }
  if (P.doBegin) {
    /* decodeBegin (LexiconFreeDecoder.cpp:20-28, LexiconDecoder.cpp:21-30) */
    if (tid == 0) {
      w.bScore[(0) * P.K + 0] = 0.0;
      w.bAm[(0) * P.K + 0] = 0.0;
      if constexpr (GMAX == 0) {
        w.bLm[(0) * P.K + 0] = 0.0;
      }
      w.bState[(0) * P.K + 0] = 0u;
      w.bSPar[(0) * P.K + 0] = P.lmKind == 2 ? 0u : kNoParent; /* host LM: LM::start's state is id 0, named (0, kHostEdge) */
      w.bSEdge[(0) * P.K + 0] = P.lmKind == 2 ? (int32_t)kHostEdge : 0;
      if constexpr (GMAX == 0) {
        w.bLex[(0) * P.K + 0] = 0u;
      }
      if constexpr (GMAX == 0) {
        w.bLexMax[0] = 0.0f;
      }
      w.bTokPb[(0) * P.K + 0] = (uint32_t)P.sil;
      if constexpr (GMAX > 0) {
        w.bMask[0] = 0ull;
        w.sc[SC_NEXTID] = 1;
        P.maskTab[(size_t)b * P.idCap] = 0ull;
      } else if (P.stateVal != nullptr) {
        w.sc[SC_NEXTID] = 1; /* (streams: ids from a counter, 0 is the root state) */
      }
      const int64_t hb = P.histOff[b];
      P.histPT[hb] = make_int2(-1, P.sil);
      if (P.kind == 1) {
        P.histW[hb] = -1;
      }
      if (P.histS) {
        P.histS[3 * hb] = 0.0;
        P.histS[3 * hb + 1] = 0.0;
        P.histS[3 * hb + 2] = 0.0;
      }
      if (P.lmKind == 1) { /* KenLM::start(false): context = <s> (KenLM.cpp:57) */
        const int L = P.lmOrder - 1;
        int32_t* c0 = P.stateCtx + (size_t)b * P.stateCap * L;
        uint32_t node = 0;
        float pr;
        const bool ok = P.tokLm == nullptr && ngFind(P, 0u, (uint32_t)P.lmBos, node, pr);
        for (int q = 0; q < L; ++q) { /* (dense token-LM table: row 0 is the start context) */
          c0[q] = (q == 0 && ok) ? (int32_t)node : 0;
        }
      }
    }
    nBeam = 1;
    frame = 0;
    total = 0;
  } else {
    nBeam = P.uttNBeam[b];
    frame = P.uttFrame[b];
    total = P.uttTotal[b];
    for (int i = tid; i < nBeam; i += W) {
      const size_t g = (size_t)b * P.K + i;
      w.bScore[(0) * P.K + i] = P.gScore[g];
      w.bAm[(0) * P.K + i] = P.gAm[g];
      if constexpr (GMAX == 0) {
        w.bLm[(0) * P.K + i] = P.gLm[g];
      }
      w.bState[(0) * P.K + i] = P.gState[g];
      w.bSPar[(0) * P.K + i] = P.gSPar[g];
      w.bSEdge[(0) * P.K + i] = P.gSEdge[g];
      if constexpr (GMAX == 0) {
        w.bLex[(0) * P.K + i] = P.gLex[g];
      }
      if constexpr (GMAX == 0) {
        w.bLexMax[i] = P.gLexMax[g];
      }
      w.bTokPb[(0) * P.K + i] = P.gTokPb[g];
      if constexpr (GMAX > 0) {
        w.bMask[i] = P.gMask[g];
      }
    }
    if constexpr (GMAX > 0) {
      if (tid == 0) {
        w.sc[SC_NEXTID] = P.uttNextId[b];
      }
    } else if (P.stateVal != nullptr) {
      if (tid == 0) {
        w.sc[SC_NEXTID] = P.uttNextId[b];
      }
    }
  }
  int T = P.stepT ? P.stepT[b] : 0;
  const float* em = P.emissions ? P.emissions + P.emOff[b] : nullptr;
  const int N = P.N;
  if (P.lmKind == 2) { /* host LM: one frame of the chunk per launch */
    em = em ? em + (size_t)P.hlmFrame * N : em;
    T = T > P.hlmFrame ? 1 : 0;
  }
  LaneCarry lcarry;
  laneCarryInit(lcarry);
  if constexpr (GT > 0) { /* relation tables start empty; the first frame compares state ids */
    for (int i = tid; i < P.K * (N + 1); i += W) {
      w.relTab[i] = 0ull;
    }
    for (int i = tid; i < P.K; i += W) {
      w.repTab[i] = 0ull;
      w.addMask[i] = 0ull;
    }
    for (int i = tid; i < nBeam; i += W) { /* descriptors unknown: the first frame compares ids */
      w.bRec[i] = laneRec(w.bScore[i], w.bTokPb[i], -1, -1);
    }
    for (int i = tid; i < P.K * N; i += W) {
      w.dKid[i] = (int16_t)-1;
    }
    for (int i = tid; i < kLaneNB; i += W) {
      w.hist[i] = 0u;
      w.tick[i] = 0u;
    }
    if (tid == 0) {
      w.sc[SC_RELSLOW] = 1;
      ((LaneLds*)wsBase)->tokMask[0] = 0ull;
      ((LaneLds*)wsBase)->tokMask[1] = 0ull;
    }
  }
  const int nTok = P.Kt < N ? P.Kt : N;
  /* stage row 0; afterwards row t+1 is prefetched into registers while frame
   * t is processed and parked in the other LDS row buffer at its end */
  constexpr int PF = 4;
  float pre[PF];
  const bool regPrefetch = N <= PF * W;
  if (T > 0) {
    for (int n = tid; n < N; n += W) {
      w.erow[n] = em[n];
    }
  }
  wsBarrierMem(P); /* also publishes the root LM state's n-gram context */
  if constexpr (GT > 0) {
    if (!FULLTOK && T > 0) { /* token list of the first frame (later ones are made a frame ahead) */
      laneShortlist(P, w, *(LaneLds*)wsBase, 0, nTok, 0);
      ldsBarrier();
    }
  }
  FrameCtx f;
  f.b = b;
  f.nTok = nTok;
  f.histBase = P.histOff[b];
  f.t0 = devClock();
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    f.acc[q] = 0ull;
  }
  LeanMap<(GMAX > 0 && GMAX < 255 ? GMAX : 1)> lmap; /* GMAX == 255: streaming lean step, no per-thread map */
  if constexpr (GMAX > 0 && GMAX < 255 && GT == 0) {
    leanMapInit(P, nTok, lmap);
  }
  for (int t = 0; t < T; ++t) {
    const int rb = t & 1;
    if (t + 1 < T) {
      const float* nx = em + (size_t)(t + 1) * N;
      if (regPrefetch) {
#pragma unroll
        for (int q = 0; q < PF; ++q) {
          const int n = tid + q * W;
          pre[q] = n < N ? nx[n] : 0.0f;
        }
      }
    }
    f.cur = cur;
    f.nBeam = nBeam;
    f.e = w.erow + rb * P.N;
    f.useTrans = (P.criterion == 0) && (total + t > 0) && P.transitions != nullptr;
    f.clock = (uint32_t)(total + t + 1);
    if constexpr (GT > 0) {
      nBeam = runFrameLane<GT, LOGADD, FULLTOK>(P, w, *(LaneLds*)wsBase, f, lcarry, frame + t + 1, rb, pre[0],
                                                t + 1 < T);
    } else if constexpr (GMAX > 0) {
      nBeam = runFrameLean<GMAX>(P, w, f, lmap, frame + t + 1);
    } else {
      nBeam = runFrame(P, w, f, frame + t + 1, false);
    }
    cur ^= 1; /* with nBeam == 0 either buffer is equally empty */
    if (GT == 0 && t + 1 < T) { /* (the lane step parks the next row itself, before its build phase) */
      if (regPrefetch) {
#pragma unroll
        for (int q = 0; q < PF; ++q) {
          const int n = tid + q * W;
          if (n < N) {
            w.erow[(rb ^ 1) * P.N + n] = pre[q];
          }
        }
      } else {
        const float* nx = em + (size_t)(t + 1) * N;
        for (int n = tid; n < N; n += W) {
          w.erow[(rb ^ 1) * P.N + n] = nx[n];
        }
      }
    }
    if constexpr (GT > 0) {
      ldsBarrier(); /* the row hand-over is LDS only; back-pointer stores stay in flight */
    } else if constexpr (GMAX > 0) {
      leanBarrier(P);
    } else if (P.lmKind != 0) {
      wsBarrierMem(P); /* n-gram contexts of the new LM states, read by next frame's scoring */
    } else {
      wsBarrier(P);
    }
    FLTX_PROF(5);
  }
  frame += T;
  total += T;
  if constexpr (GT > 0) { /* epilogue of the last frame's build */
    f.cur = cur;
    f.nBeam = nBeam;
    laneFlush(P, *(LaneLds*)wsBase, f, lcarry);
    ldsBarrier();
  }
  if (P.doEnd) {
    f.cur = cur;
    f.nBeam = nBeam;
    f.e = w.erow;
    f.useTrans = false;
    if constexpr (GMAX > 0) {
      nBeam = nBeam > 0 ? runEndLean(P, w, f, frame + 1) : 0;
    } else {
      nBeam = nBeam > 0 ? runFrame(P, w, f, frame + 1, true) : 0;
    }
    cur ^= 1;
    frame += 1;
    total += 1;
    for (int i = tid; i < nBeam; i += W) {
      const size_t g = ((size_t)b * P.K + i) * 3;
      P.outScores[g + 0] = w.bScore[(cur) * P.K + i];
      P.outScores[g + 1] = w.bAm[(cur) * P.K + i];
      P.outScores[g + 2] = GMAX > 0 ? 0.0 : w.bLm[(cur) * P.K + i]; /* ZeroLM: lmScore stays 0 */
    }
    if (tid == 0) {
      P.outN[b] = nBeam;
    }
  }
  if (P.prof && tid == P.profThread) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      P.prof[(size_t)b * 8 + q] = f.acc[q];
    }
  }
  if (P.scored && f.nScored != 0u) {
    atomAdd32(&P.scored[b], f.nScored);
  }
  /* park the beam in HBM for the next decodeStep / prune / best */
  for (int i = tid; i < nBeam; i += W) {
    const size_t g = (size_t)b * P.K + i;
    P.gScore[g] = w.bScore[(cur) * P.K + i];
    P.gAm[g] = w.bAm[(cur) * P.K + i];
    P.gLm[g] = GMAX > 0 ? 0.0 : w.bLm[(cur) * P.K + i];
    P.gState[g] = w.bState[(cur) * P.K + i];
    P.gSPar[g] = w.bSPar[(cur) * P.K + i];
    P.gSEdge[g] = w.bSEdge[(cur) * P.K + i];
    P.gLex[g] = GMAX > 0 ? 0u : w.bLex[(cur) * P.K + i];
    P.gLexMax[g] = GMAX > 0 ? 0.0f : w.bLexMax[(cur) * P.K + i];
    P.gTokPb[g] = w.bTokPb[(cur) * P.K + i];
    if constexpr (GMAX > 0) {
      P.gMask[g] = w.bMask[(cur) * P.K + i];
    }
  }
  wsBarrier(P);
  if (tid == 0) {
    P.uttNBeam[b] = nBeam;
    P.uttFrame[b] = frame;
    P.uttTotal[b] = total;
    if constexpr (GMAX > 0) {
      P.uttNextId[b] = w.sc[SC_NEXTID];
    } else if (P.stateVal != nullptr) {
      P.uttNextId[b] = w.sc[SC_NEXTID];
    }
    const int32_t stNow = P.doBegin ? w.sc[SC_STATUS] : (P.uttStatus[b] | w.sc[SC_STATUS]);
    P.uttStatus[b] = stNow;
    if (P.statusHost) {
      P.statusHost[b] = stNow;
    }
  }
}

/* ------------------------------------------------------------------------ */
/* Streams: giving LM-state ids back (fltx_compact_states_kernel, one         */
/* workgroup per stream, between two launches of the decode kernel).          */
/*                                                                            */
/* The reference keeps an LMState alive while a hypothesis in the buffer, or  */
/* the children map of a live ancestor, holds it (lm/LM.h:21-34); prune()     */
/* drops the hypotheses of the frames it cuts (Utils.h:312-342) and with them */
/* every state nothing reaches any more, so a stream's memory is bounded.     */
/* Here a state is an id; what must survive is what can still be OBSERVED:    */
/* identity only shows when two candidates of one frame meet in a state, and  */
/* candidates descend from the states of the current beam.  Needed are        */
/*   (1) the states the beam's hypotheses are in ("live"),                    */
/*   (2) their parents (a hypothesis names its state (parent, edge)),         */
/*   (3) every state on the way from a live state up to a live ancestor: a    */
/*       hypothesis in the ancestor that takes the same edges again must      */
/*       arrive in the live descendant's state, not in a copy of it.          */
/* A state below a live state that is none of these can be entered again, but */
/* only by hypotheses that all get the same fresh id for it -- nobody holds   */
/* the old one.  The needed ids are renumbered 1 .. M in their old order      */
/* (new <= old, so everything moves towards the front in place), a kept state */
/* whose parent went is cut loose (idPar = kNoParent32: a walk up a chain     */
/* ends where the kept states end; it also ends at the first ancestor born    */
/* before the oldest live state -- nothing above it can be live), the beam's  */
/* parked ids follow, and the id counter goes on from M + 1.  The history     */
/* holds no ids.                                                              */
/* ------------------------------------------------------------------------ */
struct CompactParams {
  int32_t K, N;
  int32_t mode;    /* 0: a new stream; 1: compact */
  int32_t family;  /* 0: childTab / maskTab (lean, lane, lane = LM state engines); 1: stateTab + stateVal (generic engine) */
  int64_t idCap;
  const int32_t* uttNBeam;
  uint32_t* gState;
  uint32_t* gSPar;
  int32_t* uttNextId;
  uint32_t* idPar;
  int32_t* idEdge;
  uint32_t* idBorn;
  uint8_t* keep;    /* [B*idCap] scratch: bit 1 = live, bit 0 = needed */
  uint32_t* newId;  /* [B*idCap] scratch: old id -> new id */
  uint32_t* list;   /* [B*idCap] scratch: new id - 1 -> old id */
  uint32_t* childTab;
  unsigned long long* maskTab;
  unsigned long long* gMask;
  unsigned long long* stateTab;
  uint32_t* stateVal;
  uint32_t stateCap;
  uint32_t epoch;   /* the table's new epoch: entries of the old one are free slots */
  int32_t* stateCtx;  /* n-gram LM: [B*stateCap*ctxL] context of a state, moves with its id */
  int32_t ctxL;
};
constexpr int kCompactChunk = 256; /* ids moved per round (their childTab rows are staged in LDS: 256 x 64 x 4 B) */
constexpr int kCompactLds = 16 + 4 * 32 + kCompactChunk * (64 * 4 + 8 + 4 + 4 + 4 + 4 * 8);

FLTX_DEV void compactStates(const CompactParams& Q, char* smem) {
  const int b = (int)blockIdx.x;
  const int W = (int)blockDim.x;
  const int tid = (int)threadIdx.x;
  const size_t at = (size_t)b * Q.idCap;
  uint32_t* idPar = Q.idPar + at;
  const int64_t cap = Q.idCap;
  if (Q.mode == 0) {
    if (tid == 0) {
      idPar[0] = kNoParent32;
    }
    if (Q.family == 1) {
      for (uint32_t s = (uint32_t)tid; s < Q.stateCap; s += (uint32_t)W) {
        Q.stateVal[(size_t)b * Q.stateCap + s] = kEmpty;
      }
    }
    return;
  }
  uint8_t* keep = Q.keep + at;
  uint32_t* newId = Q.newId + at;
  uint32_t* list = Q.list + at;
  uint32_t* idBorn = Q.idBorn + at;
  const int nBeam = Q.uttNBeam[b];
  uint32_t* gState = Q.gState + (size_t)b * Q.K;
  uint32_t* gSPar = Q.gSPar + (size_t)b * Q.K;
  uint32_t* shMin = (uint32_t*)smem;       /* [0] birth frame of the oldest live state */
  uint32_t* shRun = (uint32_t*)smem + 1;   /* [1] kept ids counted so far */
  uint32_t* shWave = (uint32_t*)smem + 4;  /* [32] per-wave counts of a scan round */
  const int used = Q.uttNextId[b];         /* ids handed out so far: [0, used) */
  const int64_t top = used < cap ? used : cap;
  for (int64_t i = tid; i < top; i += W) {
    keep[i] = 0;
  }
  if (tid == 0) {
    *shMin = 0xFFFFFFFFu;
    *shRun = 0u;
  }
  __syncthreads();
  for (int h = tid; h < nBeam; h += W) { /* (1) */
    const uint32_t s = gState[h];
    keep[s] = 3;
    atomMin32(shMin, s == 0u ? 0u : idBorn[s]);
  }
  __syncthreads();
  const uint32_t oldest = *shMin;
  for (int h = tid; h < nBeam; h += W) { /* (2), (3) */
    uint32_t c = gState[h];
    int depth = 0, lastLive = 0;
    for (;;) {
      const uint32_t p = c == 0u ? kNoParent32 : idPar[c];
      if (p == kNoParent32) {
        break;
      }
      ++depth;
      if (keep[p] & 2) {
        lastLive = depth;
      }
      if (p == 0u || idBorn[p] < oldest || depth >= (int)cap) {
        break; /* nothing above was made late enough to be in the beam */
      }
      c = p;
    }
    const int need = lastLive > 1 ? lastLive : 1;
    c = gState[h];
    for (int d = 0; d < need; ++d) {
      const uint32_t p = c == 0u ? kNoParent32 : idPar[c];
      if (p == kNoParent32) {
        break;
      }
      keep[p] = (uint8_t)(keep[p] | 1); /* (racing writers agree: the live bits are final) */
      c = p;
    }
  }
  __syncthreads();
  /* new ids: rank among the kept ids, in id order (id 0, the root state, keeps its name whether needed or not) */
  const int lane = laneId(), wave = waveId(), nWaves = (W + 63) >> 6;
  for (int64_t base = 1; base < top; base += W) {
    const int64_t i = base + tid;
    const bool on = i < top && keep[i] != 0;
    const unsigned long long m = waveBallot(on);
    if (lane == 0) {
      shWave[wave] = (uint32_t)popc64(m);
    }
    __syncthreads();
    uint32_t before = *shRun;
    for (int q = 0; q < wave; ++q) {
      before += shWave[q];
    }
    if (on) {
      const uint32_t r = before + (uint32_t)popc64(m & ((1ull << lane) - 1ull)); /* 0-based rank */
      newId[i] = r + 1u;
      list[r] = (uint32_t)i;
    }
    __syncthreads();
    if (tid == 0) {
      uint32_t tot = *shRun;
      for (int q = 0; q < nWaves; ++q) {
        tot += shWave[q];
      }
      *shRun = tot;
    }
    __syncthreads();
  }
  const int M = (int)*shRun;
  if (tid == 0) {
    newId[0] = 0u;
    keep[0] = (uint8_t)(keep[0] | 1);
  }
  __syncthreads();
  auto renamed = [&](uint32_t old) -> uint32_t { /* kNoParent32 when that id goes */
    return ((int64_t)old < top && keep[old]) ? newId[old] : kNoParent32;
  };
  const int N = Q.N;
  if (Q.family == 0) { /* the beam's parked child masks forget the children that go (old ids, old rows: before the move) */
    for (int h = tid; h < nBeam; h += W) {
      const uint32_t sid = gState[h];
      unsigned long long m = Q.gMask[(size_t)b * Q.K + h], out = m;
      while (m != 0ull) {
        const int n = __builtin_ctzll(m);
        m &= m - 1ull;
        if (renamed(Q.childTab[(at + (size_t)sid) * N + n]) == kNoParent32) {
          out &= ~(1ull << n);
        }
      }
      Q.gMask[(size_t)b * Q.K + h] = out;
    }
    if (tid == 0) { /* ... and so does the root's row, which stays where it is */
      unsigned long long m = Q.maskTab[at], out = m;
      while (m != 0ull) {
        const int n = __builtin_ctzll(m);
        m &= m - 1ull;
        const uint32_t c = renamed(Q.childTab[at * N + n]);
        if (c == kNoParent32) {
          out &= ~(1ull << n);
        } else {
          Q.childTab[at * N + n] = c;
        }
      }
      Q.maskTab[at] = out;
    }
  }
  __syncthreads();
  /* move what an id owns from its old place to its new one, kCompactChunk ids per round in id order: a round reads
   * everything it moves before it writes, and what it writes (new ids of this round) lies in front of every old id a
   * later round reads (new <= old) */
  char* sp = smem + 16 + 4 * 32;
  uint32_t* stRow = (uint32_t*)sp;                                  /* [chunk][64] children, renamed */
  unsigned long long* stMask = (unsigned long long*)(stRow + kCompactChunk * 64);
  uint32_t* stPar = (uint32_t*)(stMask + kCompactChunk);
  uint32_t* stBorn = stPar + kCompactChunk;
  int32_t* stEdge = (int32_t*)(stBorn + kCompactChunk);
  int32_t* stCtx = stEdge + kCompactChunk;                          /* [chunk][8] */
  for (int c0 = 0; c0 < M; c0 += kCompactChunk) {
    const int nc = M - c0 < kCompactChunk ? M - c0 : kCompactChunk;
    for (int q = tid; q < nc; q += W) {
      const uint32_t old = list[c0 + q];
      const uint32_t p = idPar[old];
      stPar[q] = p == kNoParent32 ? kNoParent32 : renamed(p);
      stBorn[q] = idBorn[old];
      if (Q.family == 1) {
        stEdge[q] = Q.idEdge[at + old];
        if (Q.stateCtx) {
          for (int x = 0; x < Q.ctxL; ++x) {
            stCtx[q * 8 + x] = Q.stateCtx[((size_t)b * Q.stateCap + old) * Q.ctxL + x];
          }
        }
      } else {
        stMask[q] = Q.maskTab[at + old];
      }
    }
    __syncthreads();
    if (Q.family == 0) { /* rows: one (id, token) pair per thread */
      for (int e = tid; e < nc * N; e += W) {
        const int q = e / N, n = e - q * N;
        if ((stMask[q] >> n) & 1ull) {
          stRow[q * 64 + n] = renamed(Q.childTab[(at + (size_t)list[c0 + q]) * N + n]);
        }
      }
    }
    __syncthreads();
    for (int q = tid; q < nc; q += W) {
      const uint32_t j = (uint32_t)(c0 + q + 1);
      idPar[j] = stPar[q];
      idBorn[j] = stBorn[q];
      if (Q.family == 1) {
        Q.idEdge[at + j] = stEdge[q];
        if (Q.stateCtx) {
          for (int x = 0; x < Q.ctxL; ++x) {
            Q.stateCtx[((size_t)b * Q.stateCap + j) * Q.ctxL + x] = stCtx[q * 8 + x];
          }
        }
      } else {
        unsigned long long m = stMask[q], out = m;
        while (m != 0ull) {
          const int n = __builtin_ctzll(m);
          m &= m - 1ull;
          const uint32_t c = stRow[q * 64 + n];
          if (c == kNoParent32) {
            out &= ~(1ull << n);
          } else {
            Q.childTab[(at + (size_t)j) * N + n] = c;
          }
        }
        Q.maskTab[at + j] = out;
      }
    }
    __syncthreads();
  }
  for (int h = tid; h < nBeam; h += W) { /* the beam's ids follow (a live state and its parent are kept by construction) */
    const uint32_t sp0 = gSPar[h];
    gState[h] = newId[gState[h]];
    if ((int64_t)sp0 < top && keep[sp0]) {
      gSPar[h] = newId[sp0];
    }
  }
  if (Q.family == 1) {
    /* the table again, under its new epoch, from the ids that stay and still have their parent */
    unsigned long long* tab = Q.stateTab + (size_t)b * Q.stateCap;
    uint32_t* val = Q.stateVal + (size_t)b * Q.stateCap;
    for (uint32_t s = (uint32_t)tid; s < Q.stateCap; s += (uint32_t)W) {
      val[s] = kEmpty;
    }
    __syncthreads();
    const uint32_t mask = Q.stateCap - 1;
    for (int j = 1 + tid; j <= M; j += W) {
      const uint32_t par = idPar[j];
      if (par == kNoParent32) {
        continue;
      }
      const int32_t edge = Q.idEdge[at + j];
      const unsigned long long key = ((unsigned long long)Q.epoch << 48) |
          ((unsigned long long)(par & 0xFFFFFFu) << 24) | (unsigned long long)((uint32_t)(edge + 1) & 0xFFFFFFu);
      uint32_t s = hashKey(par, (uint32_t)edge, 0x9747b28cu, 0) & mask;
      for (uint32_t probes = 0; probes < Q.stateCap; ++probes) {
        unsigned long long cur = loadCoherent64(&tab[s]);
        bool done = false;
        while ((cur >> 48) != (unsigned long long)Q.epoch) {
          const unsigned long long old = atomCas64(&tab[s], cur, key);
          if (old == cur) {
            storeCoherent32(&val[s], (uint32_t)j);
            done = true;
            break;
          }
          cur = old;
        }
        if (done) {
          break;
        }
        s = (s + 1) & mask;
      }
    }
  }
  __syncthreads();
  if (tid == 0) {
    Q.uttNextId[b] = M + 1;
  }
}

/* ------------------------------------------------------------------------ */
/* host LM: the questions of one frame (fltx_hostlm_questions_kernel, one     */
/* workgroup per utterance, before the frame's decode launch).                */
/* Lists, for the parked beam of utterance b and the frame's emission row,    */
/* every (LM state, index) pair candidate generation will pass to lmAsk():    */
/*   lexicon-free  lm->score(state, n) of every new-token extension           */
/*                 (LexiconFreeDecoder.cpp:69-85);                            */
/*   lexicon       lm->score(state, n) per existing trie child with a token   */
/*                 LM (LexiconDecoder.cpp:82-86), else lm->score(state, label)*/
/*                 per word the child ends (:124-127) and lm->score(state,    */
/*                 unk) for a child without words (:145-149);                 */
/*   decodeEnd     lm->finish(state) of the hypotheses that finish            */
/*                 (LexiconFreeDecoder.cpp:131-133, LexiconDecoder.cpp:236-   */
/*                 247: only those at the root if there is one).              */
/* The reference asks per hypothesis; hypotheses that share a state repeat a  */
/* question, the host answers each distinct one once (LMState::child's memo). */
/* Also copies the beam's LM states out (LM::updateCache, Utils.h:346-354).   */
/* LDS: min(Kt, N) + 4 int32.                                                  */
/* ------------------------------------------------------------------------ */
FLTX_DEV void hostLmQuestions(const DecodeParams& P, char* smem) {
  const int b = (int)blockIdx.x;
  const int W = (int)blockDim.x;
  const int tid = (int)threadIdx.x;
  const int N = P.N;
  const int nTok = P.Kt < N ? P.Kt : N;
  int32_t* cnt = (int32_t*)smem; /* [0] questions, [1] tokens listed */
  int32_t* tokIdx = cnt + 4;
  const int nBeam = P.uttNBeam[b];
  const size_t g0 = (size_t)b * P.K;
  for (int i = tid; i < nBeam; i += W) {
    P.hlmBeam[g0 + i] = P.gState[g0 + i];
  }
  if (tid == 0) {
    P.hlmBeamN[b] = nBeam;
    cnt[0] = 0;
    cnt[1] = 0;
  }
  const bool active = P.hlmEnd ? true : (P.stepT[b] > P.hlmFrame);
  if (!active || nBeam <= 0) {
    if (tid == 0) {
      P.hlmQCount[b] = 0;
    }
    return;
  }
  ldsBarrier();
  uint2* out = P.hlmQ + (size_t)b * P.hlmQCap;
  const int lane = laneId();
  auto push = [&](bool on, uint32_t sid, int idx) { /* every lane of the wave calls it together */
    const unsigned long long m = waveBallot(on);
    if (m == 0ull) {
      return;
    }
    const int leader = __builtin_ctzll(m);
    uint32_t base = 0;
    if (lane == leader) {
      base = atomAdd32((uint32_t*)&cnt[0], (uint32_t)popc64(m));
    }
    base = waveShfl32(base, leader);
    if (on) {
      const uint32_t qi = base + (uint32_t)popc64(m & ((1ull << lane) - 1ull));
      if (qi < (uint32_t)P.hlmQCap) {
        out[qi] = make_uint2(sid, (uint32_t)idx);
      }
    }
  };
  if (P.hlmEnd) {
    bool nice = false;
    if (P.kind == 1) {
      for (int h = 0; h < nBeam; ++h) {
        if (P.gLex[g0 + h] == 0u) {
          nice = true;
          break;
        }
      }
    }
    const int rounds = (nBeam + W - 1) / W;
    for (int it = 0; it < rounds; ++it) {
      const int h = it * W + tid;
      bool on = h < nBeam;
      if (on && P.kind == 1 && nice && P.gLex[g0 + h] != 0u) {
        on = false;
      }
      push(on, on ? P.gState[g0 + h] : 0u, kFinishEdge);
    }
  } else {
    const float* e = P.emissions + P.emOff[b] + (size_t)P.hlmFrame * N;
    if (nTok < N) { /* the frame's token beam (LexiconFreeDecoder.cpp:42-51): emission descending, ties to the lower index */
      for (int n = tid; n < N; n += W) {
        const float v = e[n];
        int rank = 0;
        for (int m = 0; m < N; ++m) {
          const float o = e[m];
          rank += (o > v || (o == v && m < n)) ? 1 : 0;
        }
        if (rank < nTok) {
          /* (a NaN emission compares false both ways and ranks 0: more than nTok entries may pass -- the list has
           * room for nTok; the frame's candidates then carry the NaN and the status flag reports it) */
          const uint32_t at = atomAdd32((uint32_t*)&cnt[1], 1u);
          if (at < (uint32_t)nTok) {
            tokIdx[at] = n;
          }
        }
      }
      ldsBarrier();
    }
    const bool ctc = P.criterion == 1;
    const bool hasUnk = P.unkScore > -__builtin_huge_val();
    const long long total = (long long)nBeam * nTok;
    const int rounds = (int)((total + W - 1) / W);
    for (int it = 0; it < rounds; ++it) {
      const long long i = (long long)it * W + tid;
      const bool valid = i < total;
      uint32_t sid = 0;
      int n = 0;
      bool qTok = false, qUnk = false;
      int nLab = 0, labOff = 0, lab0 = -1;
      if (valid) {
        const int h = (int)(i / nTok), r = (int)(i - (long long)h * nTok);
        n = nTok == N ? r : tokIdx[r];
        const uint32_t tp = P.gTokPb[g0 + h];
        const int prevTok = (int)(tp & 0x7FFFFFFFu);
        const bool prevBlank = (tp & kPrevBlank) != 0;
        sid = P.gState[g0 + h];
        if (P.kind == 0) {
          qTok = ctc ? (n != P.blank && (n != prevTok || prevBlank)) : (n != prevTok);
        } else {
          const uint32_t lexId = P.gLex[g0 + h];
          const TrieEdge ed = P.trieEdge[(size_t)lexId * N + n];
          if (ed.child >= 0) {
            if (P.isLmToken) {
              qTok = true;
            } else {
              const int nl = (int)(ed.meta & 7u);
              if (!(lexId == 0u && prevTok == n)) { /* :114-122 */
                nLab = nl;
                labOff = (int)(ed.meta >> 4);
                lab0 = ed.label0;
              }
              qUnk = nl == 0 && hasUnk;
            }
          }
        }
      }
      push(qTok, sid, n);
      for (int j = 0; j < 6; ++j) {
        const bool on = j < nLab;
        if (waveBallot(on) == 0ull) {
          break;
        }
        push(on, sid, on ? (j == 0 ? lab0 : P.trieLabels[labOff + j]) : 0);
      }
      push(qUnk, sid, P.unk);
    }
  }
  ldsBarrier();
  if (tid == 0) {
    P.hlmQCount[b] = cnt[0];
  }
}

/* ------------------------------------------------------------------------ */
/* back-trace (getAllHypothesis, Utils.h:229-266): hypothesis k of utterance  */
/* b walks parent slots from frame `finalFrame` down to 0.                    */
/* ------------------------------------------------------------------------ */
struct BacktraceParams {
  int32_t K, kind;
  const int2* histPT;
  const int32_t* histW;
  const int64_t* histOff;
  const int32_t* uttFrame; /* final frame index per utterance */
  const int32_t* uttNBeam;
  const int64_t* tokOff;   /* output offset of utterance b (int32 elements) */
  int32_t* tokens;
  int32_t* words;
  int32_t nbest;           /* <= 0: all */
  int32_t F;               /* frames per LDS chunk (0: walk straight through HBM) */
  /* records of the lane = LM state engines: parent slot in the low `packed` bits of x (all ones = none): 8 for
   * fltx_slane.h / fltx_xlane.h / fltx_ylane.h, 10 for fltx_mlane.h, 13 for fltx_ylane.h with four lane groups;
   * 0 = plain {parent, token} records */
  int32_t packed;
  int32_t amGather;      /* the emission (and transition) of a path's token is read from HBM where it is needed instead of staging whole rows in LDS (large token sets) */
  int32_t packedTokMask; /* the token in y of those records: 0xFF (fltx_mlane.h keeps state-id bits above it), all bits for fltx_wlane.h */
  const int32_t* uttStatus; /* ST_PACKED per utterance (a re-run on the generic engine leaves plain records) */
  /* that engine does not carry the emitting-model score through the frames; it is
   * re-accumulated here along each returned path, in the reference's order
   * (LexiconFreeDecoder.cpp:58-63,82,95,108: am = prev.am + (e[t][n] (+ transition))) */
  double* amOut;           /* outScores, or null */
  const float* emissions;
  const int64_t* emOff;
  const float* transitions; /* ASG, else null */
  int32_t N;
  /* fltx_slane.h with a token-level n-gram LM (TL) does not carry the LM score either: along a returned path the
   * new-token steps follow from the tokens alone (LexiconFreeDecoder.cpp:69-72: CTC n != blank && n != previous token
   * -- a previous blank is a different token --, ASG n != previous token), each adds lm.score(context, n) from the dense
   * table and moves the context on, decodeEnd adds lm.finish (:131-146): prev.lmScore + lmScore in path order */
  const int2* tokLm;     /* DecodeParams::tokLm, or null */
  int32_t tokLmStride;
  int32_t blank;         /* CTC: the blank token; ASG: -1 */
};

/* A parent-pointer walk is a chain of T dependent loads; straight from HBM that
 * is T x ~1 us per hypothesis and the token rows come out as strided 4-byte
 * stores.  Instead the history is taken F frames at a time, newest first: the
 * chunk's {parent, token} (and word) records are copied to LDS with coalesced
 * loads (double buffered: the next chunk arrives while this one is walked),
 * every hypothesis walks its F steps there (~100 clocks each), drops its
 * tokens into an LDS tile, and the tile leaves as row-contiguous stores. */
FLTX_DEV void backtraceUtterance(const BacktraceParams& P, char* smem) {
  const int b = (int)blockIdx.x;
  const int W = (int)blockDim.x, tid = (int)threadIdx.x;
  const int ff = P.uttFrame[b];
  int nh = P.uttNBeam[b];
  if (P.nbest > 0 && nh > P.nbest) {
    nh = P.nbest;
  }
  const int len = ff + 1;
  const int K = P.K;
  const bool lex = P.kind == 1;
  const bool packed = P.packed && (P.uttStatus[b] & ST_PACKED) != 0;
  const int pmask = (1 << P.packed) - 1;
  const int64_t hb = P.histOff[b], ob = P.tokOff[b];
  if (P.F <= 0) { /* beam too large for a useful chunk: one thread per hypothesis through HBM */
    for (int k = tid; k < nh; k += W) {
      int slot = k;
      int32_t* tk = P.tokens + ob + (int64_t)k * len;
      int32_t* wd = P.words ? P.words + ob + (int64_t)k * len : nullptr;
      for (int fr = ff; fr >= 0; --fr) {
        int tokv = -1, wv = -1;
        if (slot >= 0) {
          const int64_t idx = hb + (int64_t)fr * K + slot;
          const int2 pt = P.histPT[idx];
          tokv = packed ? (pt.y < 0 ? pt.y : (pt.y & P.packedTokMask)) : pt.y;
          wv = lex ? P.histW[idx] : -1;
          slot = packed ? (((pt.x & pmask) == pmask) ? -1 : (pt.x & pmask)) : pt.x;
        }
        tk[fr] = tokv; /* pruned history: -1 below the cut */
        if (wd) {
          wd[fr] = wv;
        }
      }
    }
    return;
  }
  const int F = P.F;
  /* two record buffers: while the walker waves walk chunk c, the other waves
   * copy chunk c + 1 (the copy is most of the kernel's time) */
  int2* cPT0 = (int2*)smem;                                   /* [2][F][K] */
  int32_t* cW0 = (int32_t*)(cPT0 + (size_t)2 * F * K);        /* [2][F][K] (lexicon decoder) */
  int32_t* oT = cW0 + (lex ? (size_t)2 * F * K : 0);          /* [nh][F] */
  int32_t* oW = oT + (size_t)F * K;                           /* [nh][F] */
  int slot[4]; /* hypotheses tid, tid + W, ... (nh <= 4 W is checked by the host) */
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    slot[q] = tid + q * W;
  }
  const int wThreads = nh >= W ? W : ((nh + 63) >> 6) << 6; /* whole waves that walk */
  const int mThreads = W - wThreads;                         /* threads free to copy meanwhile */
  /* copy chunk [lo, hi] into buffer `buf` with threads t0, t0 + step, ...: eight
   * loads in flight per thread before the first LDS store -- the copy is
   * bandwidth work, not a chain of load -> store round trips */
  auto copyChunk = [&](int hi, int buf, int t0, int step) {
    const int lo = hi - F + 1 > 0 ? hi - F + 1 : 0;
    const int n = (hi - lo + 1) * K;
    const int64_t src = hb + (int64_t)lo * K;
    int2* cPT = cPT0 + (size_t)buf * F * K;
    int32_t* cW = cW0 + (size_t)buf * F * K;
    for (int base = t0; base < n; base += 8 * step) {
      int2 v[8];
      int32_t wv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = base + u * step;
        if (i < n) {
          v[u] = P.histPT[src + i];
          wv[u] = lex ? P.histW[src + i] : -1;
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = base + u * step;
        if (i < n) {
          cPT[i] = v[u];
          if (lex) {
            cW[i] = wv[u];
          }
        }
      }
    }
  };
  copyChunk(ff, 0, tid, W);
  __syncthreads();
  int c = 0;
  for (int hi = ff; hi >= 0; hi -= F, ++c) {
    const int lo = hi - F + 1 > 0 ? hi - F + 1 : 0;
    const int nf = hi - lo + 1;
    const int2* cPT = cPT0 + (size_t)(c & 1) * F * K;
    const int32_t* cW = cW0 + (size_t)(c & 1) * F * K;
    const bool more = lo > 0;
    if (tid < wThreads) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int k = tid + q * W;
        if (k < nh) {
          int s = slot[q];
          for (int j = nf - 1; j >= 0; --j) {
            int tokv = -1, wv = -1;
            if (s >= 0) {
              const int2 pt = cPT[j * K + s];
              tokv = packed ? (pt.y < 0 ? pt.y : (pt.y & P.packedTokMask)) : pt.y;
              wv = lex ? cW[j * K + s] : -1;
              s = packed ? (((pt.x & pmask) == pmask) ? -1 : (pt.x & pmask)) : pt.x;
            }
            oT[k * F + j] = tokv;
            if (P.words) {
              oW[k * F + j] = wv;
            }
          }
          slot[q] = s;
        }
      }
    } else if (more) {
      copyChunk(lo - 1, (c + 1) & 1, tid - wThreads, mThreads);
    }
    if (more && mThreads == 0) { /* every thread walks: copy afterwards */
      copyChunk(lo - 1, (c + 1) & 1, tid, W);
    }
    __syncthreads();
    for (int k = tid >> 6; k < nh; k += W >> 6) { /* a wave per row: contiguous stores */
      for (int j = tid & 63; j < nf; j += 64) {
        P.tokens[ob + (int64_t)k * len + lo + j] = oT[k * F + j];
        if (P.words) {
          P.words[ob + (int64_t)k * len + lo + j] = oW[k * F + j];
        }
      }
    }
    __syncthreads();
  }
  if (P.amOut && packed) {
    /* emitting-model scores, oldest frame first: the token rows just written and the
     * emission rows are staged F frames at a time, then hypothesis k (thread k) adds up
     * its path with LDS reads only (the chain is the additions, not the loads) */
    float* eT = (float*)smem;                       /* [F][N] */
    float* trT = eT + (P.amGather ? 0 : (size_t)F * P.N);              /* [N][N] (ASG) */
    int32_t* tT = (int32_t*)(trT + ((P.transitions && !P.amGather) ? (size_t)P.N * P.N : 0)); /* [nh][F] */
    const int N = P.N;
    const float* em = P.emissions + P.emOff[b];
    if (P.transitions && !P.amGather) {
      for (int i = tid; i < N * N; i += W) {
        trT[i] = P.transitions[i];
      }
    }
    const float* const trG = P.amGather ? P.transitions : trT;
    const int Tb = ff - 1; /* frames decoded: history rows 1 .. Tb */
    double am = 0.0;
    int prevTok = (tid < nh && len > 0) ? P.tokens[ob + (int64_t)tid * len] : 0;
    double lmAcc = 0.0; /* token LM: the path's LM score ... */
    int lmCtx = 0;      /* ... and context row (row 0 = lm.start) */
    int lmPrev = prevTok;
    for (int lo = 1; lo <= Tb; lo += F) {
      const int hi = lo + F - 1 < Tb ? lo + F - 1 : Tb;
      const int nf = hi - lo + 1;
      __syncthreads();
      { /* eight loads in flight per thread before the first LDS store (a load -> store loop would
           pay one memory latency per element), and no integer division in the index arithmetic */
        const int nWv = W >> 6, wv = tid >> 6, ln = tid & 63;
        for (int jb = 0; jb < nf; jb += 64) { /* token rows: wave wv takes rows wv, wv + nWv, ... */
          for (int k0 = wv; k0 < nh; k0 += 8 * nWv) {
            int32_t v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const int k = k0 + u * nWv;
              v[u] = (k < nh && jb + ln < nf) ? P.tokens[ob + (int64_t)k * len + lo + jb + ln] : -1;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const int k = k0 + u * nWv;
              if (k < nh && jb + ln < nf) {
                tT[k * F + jb + ln] = v[u];
              }
            }
          }
        }
        const int nEm = P.amGather ? 0 : nf * N; /* emission rows of the chunk: contiguous */
        for (int base = tid; base < nEm; base += 8 * W) {
          float v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int i = base + u * W;
            v[u] = i < nEm ? em[(size_t)(lo - 1) * N + i] : 0.0f;
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int i = base + u * W;
            if (i < nEm) {
              eT[i] = v[u];
            }
          }
        }
      }
      __syncthreads();
      if (tid < nh) {
        /* the chain is the additions: tokens and emissions of eight steps are read ahead of them
         * (two dependent LDS round trips per step otherwise) */
        constexpr int U = 8;
        const int row = tid * F;
        auto walk = [&](auto withTrans) {
          constexpr bool TR = decltype(withTrans)::value;
          for (int j0 = 0; j0 < nf; j0 += U) {
            int tk[U];
            float ev[U], tr[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
              tk[u] = tT[row + (j0 + u < nf ? j0 + u : nf - 1)];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
              const int jj = j0 + u < nf ? j0 + u : nf - 1;
              ev[u] = P.amGather ? em[(size_t)(lo - 1 + jj) * N + (tk[u] >= 0 ? tk[u] : 0)] : eT[jj * N + (tk[u] >= 0 ? tk[u] : 0)];
              tr[u] = 0.0f;
            }
            if (TR) {
#pragma unroll
              for (int u = 0; u < U; ++u) {
                const int pv = u == 0 ? prevTok : tk[u - 1];
                tr[u] = (tk[u] >= 0 && pv >= 0 && lo - 1 + j0 + u > 0) ? trG[(size_t)tk[u] * N + pv] : 0.0f;
              }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
              if (j0 + u < nf && tk[u] >= 0) {
                double x = (double)ev[u];
                if (TR && lo - 1 + j0 + u > 0) {
                  x += (double)tr[u];
                }
                am += x;
                prevTok = tk[u];
              }
            }
          }
        };
        if (P.transitions) {
          walk(SlParity<1>());
        } else {
          walk(SlParity<0>());
        }
        if (P.tokLm) { /* a chain of dependent gathers, one per new-token step (a few hundred per path; rows shared by
                          the paths of an n-best stay in the L1 / L2) */
          for (int j = 0; j < nf; ++j) {
            const int tk = tT[row + j];
            if (tk >= 0) {
              if (tk != lmPrev && tk != P.blank) {
                const int2 e = P.tokLm[(size_t)lmCtx * (size_t)P.tokLmStride + (size_t)tk];
                lmAcc = lmAcc + (double)__uint_as_float((uint32_t)e.x);
                lmCtx = e.y;
              }
              lmPrev = tk;
            }
          }
        }
      }
    }
    if (tid < nh) {
      P.amOut[((size_t)b * K + tid) * 3 + 1] = am;
      if (P.tokLm) {
        lmAcc = lmAcc + (double)__uint_as_float((uint32_t)P.tokLm[(size_t)lmCtx * (size_t)P.tokLmStride + (size_t)P.N].x);
        P.amOut[((size_t)b * K + tid) * 3 + 2] = lmAcc;
      }
    }
  }
}
/* ------------------------------------------------------------------------ */
/* streaming helpers on the device-resident history:                         */
/*   op 0: getBestHypothesis(lookBack)  (LexiconFreeDecoder.cpp:188-194,       */
/*         LexiconDecoder.cpp:285-293, findBestAncestor Utils.h:268-310)       */
/*   op 1: prune(lookBack)              (LexiconFreeDecoder.cpp:205-227,       */
/*         pruneAndNormalize Utils.h:312-342)                                  */
/* ------------------------------------------------------------------------ */
struct StreamOpParams {
  int32_t K, kind, op, lookBack;
  int2* histPT;
  int32_t* histW;
  double* histS;
  const int64_t* histOff;
  int32_t* uttFrame;
  const int32_t* uttNBeam;
  double* gScore;     /* current beam scores (normalised by prune) */
  /* op 0 outputs */
  int32_t* outLen;    /* [B] length of the result (0 = empty DecodeResult) */
  double* outScores;  /* [B*3] */
  int32_t* outTok;    /* [B*cap] */
  int32_t* outWrd;
  int32_t cap;
};

constexpr int kLookBackLimit = 100; /* Utils.h:28 */

constexpr int kStageInts = 8192;    /* LDS (32 KB) the walk stages history rows in */
constexpr int kStreamOpLds = (8 + kStageInts) * 4;

FLTX_DEV void streamOpUtterance(const StreamOpParams& Q, int32_t* sh) {
  const int b = (int)blockIdx.x;
  const int tid = (int)threadIdx.x, W = (int)blockDim.x;
  const int64_t base = Q.histOff[b];
  const int ff = Q.uttFrame[b];
  const int nB = Q.uttNBeam[b];
  const int maxLookBack = Q.lookBack + kLookBackLimit;
  /* findBestAncestor: lookBack steps back from the best hypothesis, then on to a complete one (LexiconDecoder.h:97-99:
   * no parent, or the parent ended a word; LexiconFreeDecoder.h:84-86: every hypothesis is).  A chain of dependent
   * loads -- so the rows it is about to visit (their parent and word columns) are fetched into LDS by all threads, R
   * rows at a time, and one thread walks them there.
   * sh[0] = frame of the ancestor, sh[1] = slot (-1 none), sh[2] = steps n, sh[3] = go, sh[4] = walk finished */
  int32_t* stPar = sh + 8;
  int R = kStageInts / 2 / Q.K;
  R = R > 64 ? 64 : R;
  int32_t* stWrd = stPar + R * Q.K;
  if (tid == 0) {
    int s = nB > 0 ? 0 : -1; /* the beam is sorted: slot 0 is the best */
    bool go = true;
    if (Q.op == 1 && ff - Q.lookBack < 1) {
      go = false; /* not enough decoded frames to prune */
    }
    if (Q.op == 0 && Q.kind == 1 && ff - Q.lookBack < 1) {
      go = false; /* LexiconDecoder.cpp:286-288 */
      s = -1;
    }
    sh[0] = ff;
    sh[1] = s;
    sh[2] = 0;
    sh[3] = go ? 1 : 0;
    sh[4] = (!go || s < 0) ? 1 : 0;
  }
  __syncthreads();
  if (R >= 2) {
    while (sh[4] == 0) {
      const int fTop = sh[0];
      const int lo = fTop - R + 1 > 0 ? fTop - R + 1 : 0;
      const int cnt = (fTop - lo + 1) * Q.K;
      for (int e = tid; e < cnt; e += W) {
        stPar[e] = Q.histPT[base + (int64_t)lo * Q.K + e].x;
        if (Q.kind == 1) {
          stWrd[e] = Q.histW[base + (int64_t)lo * Q.K + e];
        }
      }
      __syncthreads();
      if (tid == 0) {
        int f = fTop, s = sh[1], n = sh[2];
        bool done = false;
        auto par = [&](int fr, int sl) { return fr == 0 ? -1 : stPar[(fr - lo) * Q.K + sl]; };
        while (true) {
          if (s < 0) {
            done = true;
            break;
          }
          if (f > 0 && f < lo) {
            break; /* the next rows */
          }
          if (n < Q.lookBack) {
            ++n;
            s = par(f, s);
            --f;
            continue;
          }
          if (Q.kind != 1) {
            done = true;
            break;
          }
          const int p = par(f, s);
          if (p < 0) {
            done = true;
            break;
          }
          if (f - 1 < lo) {
            break; /* the parent's row is not here */
          }
          if (stWrd[(f - 1 - lo) * Q.K + p] >= 0) {
            done = true;
            break;
          }
          ++n;
          s = p;
          --f;
          if (n == maxLookBack) {
            done = true;
            break;
          }
        }
        sh[0] = f;
        sh[1] = s;
        sh[2] = n;
        sh[4] = done ? 1 : 0;
      }
      __syncthreads();
    }
  } else if (tid == 0 && sh[4] == 0) { /* (a beam too wide for the staging area: the loads one after the other) */
    int f = ff, s = sh[1], n = 0;
    auto parentOf = [&](int fr, int sl) { return fr == 0 ? -1 : Q.histPT[base + (int64_t)fr * Q.K + sl].x; };
    while (s >= 0 && n < Q.lookBack) {
      ++n;
      s = parentOf(f, s);
      --f;
    }
    if (Q.kind == 1) {
      int p = s >= 0 ? parentOf(f, s) : -1;
      while (s >= 0) {
        int w = -1, pp = -1;
        if (p >= 0) {
          w = Q.histW[base + (int64_t)(f - 1) * Q.K + p];
          pp = parentOf(f - 1, p);
        }
        if (p < 0 || w >= 0) {
          break;
        }
        ++n;
        s = p;
        --f;
        p = pp;
        if (n == maxLookBack) {
          break;
        }
      }
    }
    sh[0] = f;
    sh[1] = s;
    sh[2] = n;
  }
  __syncthreads();
  const int f = sh[0], s0 = sh[1], n = sh[2];
  const bool go = sh[3] != 0;
  if (Q.op == 0) {
    if (!go || s0 < 0) {
      if (tid == 0) {
        Q.outLen[b] = 0;
      }
      return;
    }
    const int len = f + 1;
    if (tid == 0) {
      Q.outLen[b] = len;
      const int64_t r = base + (int64_t)f * Q.K + s0;
      if (Q.histS) {
        Q.outScores[3 * b] = Q.histS[3 * r];
        Q.outScores[3 * b + 1] = Q.histS[3 * r + 1];
        Q.outScores[3 * b + 2] = Q.histS[3 * r + 2];
      }
      if (len <= Q.cap) {
        int sl = s0;
        for (int fr = f; fr >= 0; --fr) {
          const int64_t idx = base + (int64_t)fr * Q.K + (sl < 0 ? 0 : sl);
          const int2 pt = Q.histPT[idx];
          Q.outTok[(int64_t)b * Q.cap + fr] = sl < 0 ? -1 : pt.y;
          Q.outWrd[(int64_t)b * Q.cap + fr] = (sl < 0 || Q.kind != 1) ? -1 : Q.histW[idx];
          sl = sl < 0 ? -1 : (fr == 0 ? -1 : pt.x);
        }
      }
    }
    return;
  }
  /* prune */
  if (!go || s0 < 0) {
    return;
  }
  const int startFrame = ff - n;
  if (startFrame < 1) {
    return;
  }
  /* rows startFrame .. startFrame + n move to 0 .. n.  A group of startFrame rows reads only rows beyond the ones it
   * writes, so it is copied by all threads at once; groups follow each other in order */
  const int rows = n + 1;
  for (int i0 = 0; i0 < rows; i0 += startFrame) {
    const int cnt = (rows - i0 < startFrame ? rows - i0 : startFrame) * Q.K;
    for (int e = tid; e < cnt; e += W) {
      const int64_t dst = base + (int64_t)i0 * Q.K + e;
      const int64_t src = dst + (int64_t)startFrame * Q.K;
      int2 pt = Q.histPT[src];
      if (i0 == 0 && e < Q.K) {
        pt.x = -1; /* avoid further back-tracking (Utils.h:325-327) */
      }
      Q.histPT[dst] = pt;
      if (Q.kind == 1) {
        Q.histW[dst] = Q.histW[src];
      }
      if (Q.histS) {
        Q.histS[3 * dst] = Q.histS[3 * src];
        Q.histS[3 * dst + 1] = Q.histS[3 * src + 1];
        Q.histS[3 * dst + 2] = Q.histS[3 * src + 2];
      }
    }
    __syncthreads(); /* (the rows are this workgroup's alone: its barrier orders them) */
  }
  /* avoid score under/overflow: subtract the largest score of the newest frame
   * (Utils.h:329-341); the beam is sorted, so that is slot 0 */
  const double largest = nB > 0 ? Q.gScore[(size_t)b * Q.K] : 0.0;
  __syncthreads();
  for (int k = tid; k < nB; k += W) {
    Q.gScore[(size_t)b * Q.K + k] -= largest;
    if (Q.histS) {
      Q.histS[3 * (base + (int64_t)n * Q.K + k)] -= largest;
    }
  }
  if (tid == 0) {
    Q.uttFrame[b] = n;
  }
}
#endif /* !FLTX_HOST_ONLY */

} // namespace fltx
