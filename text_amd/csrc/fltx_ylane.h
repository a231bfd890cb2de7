/*
 * fltx_ylane.h -- "lane = (LM state, trie node)" decode of a whole utterance for the
 * lexicon decoder with a word LM: LexiconDecoder + ZeroLM or n-gram LM, smeared trie
 * (TrieNode::maxScore), CTC, max-merge, beam <= 256 (one, two or four groups of 64 lanes), <= 64 tokens, one word per
 * spelling, every word ending in the separator token, no <unk>, offline.  Included by
 * fltx_kernels.h after fltx_xlane.h, whose lane formulation it keeps (read the head of
 * that file first) and whose row staging, token-beam ranking, histogram window and scan
 * it shares.  Same candidates, same merge groups, same selection as
 * LexiconDecoder::decodeStep (LexiconDecoder.cpp:32-229) with candidatesStore
 * (Utils.h:146-225): bit-identical n-best, emitting-model and LM scores.
 *
 * What differs from fltx_xlane.h:
 *   * lanes keep their slot for as long as they live (two groups of 64): nothing is
 *     compacted, the fields that describe a lane's trie node are written once, links
 *     between a lane and its trie parent's lane stay valid until one of them drops out;
 *     a new lane takes a slot from the free list the lanes' own waves publish;
 *   * the token waves do not walk a lane x token grid -- deep in the trie a lane has one
 *     or two children -- but compact the (lane, token) pairs that have a child into a
 *     list first (a ballot and a prefix count per token and group) and then evaluate one
 *     pair per thread and round;
 *   * scores carry the LM terms of the reference: the smearing difference
 *     child.maxScore - parent.maxScore when a token is eaten (:96), lm.score - maxScore
 *     when a word ends (:125), lm.finish at the end (:246); a lane also carries the LM
 *     score of its two hypotheses (the emitting-model score is re-accumulated by the
 *     back-trace);
 *   * LM states are numbered as in fltx_xlane.h (LMState::child is a trie over words, also
 *     for KenLM, lm/KenLM.cpp:66-75); the n-gram context of a state lives in
 *     DecodeParams::stateCtx as for the generic engine, and the n-gram score of the word a
 *     lane can end is looked up once per lane and kept with it.
 * Waves: token waves, one per lane group for the lanes' own groups (blank, stay + parent's
 * extension), one for the word ends and blank-then-own-token, one that stages the rows.
 */
#pragma once

/* Sizes follow the lane groups: LG = 2 serves one and two groups (the C4 / C5 geometries), LG = 4 beams up to 256 */
constexpr int kYlMemo = 8192; /* slots of the LM-state memo in LDS: (LM state + 1) << 40 | (word + 1) << 16 | number of the child state */
static_assert(kYlMemo == kXlMemoH, "DecodeParams::ymemo is sized for either engine");
constexpr int kYlPairs = 512; /* (lane, token) pairs a token wave can list: positions per wave x lanes */
constexpr int kYlPairs4 = 768; /* ... with four lane groups: three positions x 256 lanes */
constexpr uint32_t kYlNoLm = 0x7FC00001u; /* endLm: not looked up yet (a NaN no arithmetic produces) */
constexpr int kYlMaxGroups = 8;
constexpr int kYlExtraPerGroup = 128; /* (lane, further word of its spelling) pairs the word wave takes per frame and lane group: 64
                                       * per extra candidate slot of its threads, two slots per lane group */

template <int LG, bool ML = false>
struct YlLanesT { /* in place: slot = lane for as long as the lane lives */
  static constexpr int kYlLanes = 64 * LG;
  double nb[kYlLanes], b[kYlLanes];
  double lmNB[kYlLanes], lmB[kYlLanes];                 /* LM score of the two hypotheses */
  unsigned long long childMask[kYlLanes], kidsMask[kYlLanes]; /* XNode of the lane's node */
  uint32_t info[kYlLanes];       /* own token | history slot of nb << 16 | of b << 24 */
  uint32_t link[kYlLanes];       /* lane + 1 of the trie parent's lane, 0 = not in the beam (or root) */
  uint32_t lmSid[kYlLanes];      /* LM state */
  uint32_t node[kYlLanes];       /* trie node, breadth-first id, 0 = root */
  uint32_t parent[kYlLanes];     /* its parent node */
  uint32_t firstChild[kYlLanes];
  int32_t endLabel[kYlLanes];    /* word that ends when the separator follows, -1 = none */
  uint32_t dPar[kYlLanes];       /* root lanes: the LM state the word was emitted from ... */
  int32_t dWord[kYlLanes];       /* ... and the word (-1: the start state) */
  float maxScore[kYlLanes];      /* TrieNode::maxScore of the node */
  float delta[kYlLanes];         /* maxScore - (parent is the root ? 0 : parent's maxScore), LexiconDecoder.cpp:47,96 */
  int32_t endWord[kYlLanes];     /* LM word id of endLabel (n-gram LM) */
  uint32_t endLm[kYlLanes];      /* float bits: lm.score(LM state, endLabel), kYlNoLm = not looked up */
  int32_t endCtx[kMaxNgramOrder - 1][kYlLanes]; /* ... and the n-gram context of the LM state that word leads to */
  uint32_t endLmX[ML ? 2 * kYlLanes : 4]; /* (LMK bit 2) float bits: lm.score(LM state, second / third word of the spelling), kYlNoLm = not looked up */
  /* (LMK bit 2) several words per spelling (Trie.h:19: up to 6): place of the first in trieLabels << 3 | words.
   * Without the bit these two members are 16 bytes each: the arrays behind keep their 16-byte alignment */
  uint32_t endExtra[ML ? kYlLanes : 4];
};

template <int kYlRoot>
struct alignas(16) YlRootTabT {
  unsigned long long key[kYlRoot];  /* 0 = free */
  unsigned long long best[kYlRoot]; /* order-preserving score key */
  double winLm[kYlRoot];            /* LM score of the arrival that represents the slot */
  uint32_t lane[kYlRoot];           /* lane + 1 of the root lane that already stands there, 0 = none */
  uint32_t minLane[kYlRoot];        /* lowest arriving lane among those that reach `best` */
  uint32_t winHyp[kYlRoot];         /* its history slot ... */
  int32_t winWord[kYlRoot];         /* ... and word (for the root lane's back-pointer) */
};
template <int LG, int kYlOrph>
struct alignas(16) YlOrphTabT {
  static constexpr int kSlots = kYlOrph;
  unsigned long long key[kYlOrph]; /* 0 = free */
  unsigned long long lanes[LG][kYlOrph];
};

template <int LG, bool ML = false, bool LA = false> /* ML: several words per spelling (LMK bit 2) -- room for their arrivals; LA: logAdd (bit 3) */
struct YlaneLdsT {
  static constexpr int kYlLanes = 64 * LG;
  static constexpr int kYlX = ML ? kYlExtraPerGroup * LG : 0; /* further words of the frame's lanes */
  /* slots of the per-frame (LM state, word) merge table: twice the lanes; with further words (one and two lane groups
   * only), the lanes plus the further words and a third as much again */
  static constexpr int kYlRoot = ML ? 512 : 128 * LG;
  static constexpr int kYlOrph = 128 * LG;  /* slots of the per-frame table of lanes without a parent lane */
  static constexpr int kYlTokWaves = LG <= 2 ? 8 : 10;
  static constexpr int kYlPairs = LG <= 2 ? fltx::kYlPairs : fltx::kYlPairs4;
  using YlRootTab = YlRootTabT<kYlRoot>;
  using YlOrphTab = YlOrphTabT<LG, kYlOrph>;
  YlLanesT<LG, ML> L;
  unsigned long long cmask[2][kYlLanes]; /* tokens whose child node holds a lane that links here */
  uint32_t hist[2][kSlNB];
  double eAll[2][64];
  double eTok[2][kSlList];
  unsigned long long tokBit[2][kSlList];
  SlRow row[2];
  uint8_t tokId[2][kSlList];
  YlRootTab root;
  YlOrphTab orph[2];
  unsigned long long bestKey[2];
  unsigned long long alive[2][LG];   /* lanes of the frame, per group */
  unsigned long long surv[LG];       /* ... that stay for the next frame (alive[next] = these + the new lanes) */
  XNode rootNode;
  int32_t rootWord, pad1;            /* LM word id of the root's endLabel */
  uint32_t off[32];                  /* new lanes of the waves before wave i (token waves, then the word wave); [last + 1] = all */
  uint32_t offH[kYlMaxGroups + 4];   /* surviving hypotheses of the lane groups before group g; [NG] = all */
  uint32_t nFree[LG];
  uint8_t freeList[LG][64];          /* free slots of a group, in slot order */
  uint16_t lmReq[kYlLanes];          /* lanes whose word has no n-gram score yet */
  /* arrivals that become root lanes, and what was found out for them; an arrival = slot * 64 + thread of the word
   * wave: slot g < NG is the word of lane g * 64 + thread, slot NG + r the further word xEmit[r * 64 + thread] */
  uint16_t nrList[kYlLanes + kYlX];
  int16_t nrOrph[kYlLanes + kYlX];
  uint32_t nrSid[kYlLanes + kYlX];
  /* further words of a spelling (LMK bit 2): this frame's (lane | index of the word << 8) pairs, 64 per extra slot of the
   * word wave, with the word (its n-gram score: YlLanesT::endLmX for the second and third word, asked each frame beyond) */
  uint16_t xEmit[ML ? kYlX : 4];
  int32_t xLabel[ML ? kYlX : 2];
  uint32_t rootExtra, padX[3]; /* (these three members: a multiple of 16 bytes with and without ML, see endExtra) */
  uint16_t cand[kYlTokWaves][kYlPairs]; /* (lane | list position << 8) pairs of a token wave */
  uint32_t whist[kYlTokWaves][kSlNB];   /* a wave with more pairs than its rounds take ranks them: counts per bin */
  unsigned long long lb[2];          /* a candidate the frame is known to have (stay / blank of a surviving lane): best >= this */
  uint32_t scal[16];
  unsigned long long bKey[kSlBCap];
  uint32_t bOrd[kSlBCap];
  uint32_t lmNext, pad0;
  /* decodeEnd */
  unsigned long long endKey[kYlLanes];
  double endScore[kYlLanes], endLmS[kYlLanes];
  uint32_t endHyp[kYlLanes];
  /* A token wave that ranks its own pairs prices each once and keeps the score here as a float (the order key of every
   * counting pass is taken from that float): 512 per token wave; the shared-CU geometries are launched with the part
   * of it they use (fltx_api.cpp: ten of the sixteen KB with one lane group, eight -- four waves, the first 512 of
   * a wave's 1 024 pairs -- with two).  Four lane groups have no room for it and price a pair in every pass. */
  /* logAdd: per slot of the merge table, the sum of exp(member - the slot's best member) over the members above the
   * frame's threshold (as fltx_xlane.h; 16 bytes without it: pscore keeps its alignment) */
  double rootAcc[LA ? kYlRoot : 2];
  float pscore[LG <= 2 ? kYlTokWaves * fltx::kYlPairs : 4];
  /* Last member: the shared-CU geometry (HM = 1) keeps the memo in HBM (DecodeParams::ymemo) and is
   * launched with offsetof(YlaneLds, memo) bytes of LDS -- 77 KB, so that two workgroups fit a CU and
   * one utterance's waits (two thirds of its wave cycles) are the other's time to run. */
  unsigned long long memo[kYlMemo];
};
using YlaneLds = YlaneLdsT<2>; /* one and two lane groups */

enum { YL_FLAG = 15, YL_NICE = 14, YL_WHYCODE = 13 };

template <typename LDS>
FLTX_DEV int ylRootFind(LDS& S, unsigned long long key) {
  constexpr int kYlRoot = LDS::kYlRoot;
  static_assert((kYlRoot & (kYlRoot - 1)) == 0, "a power of two");
  uint32_t h = xlHash(key) & (kYlRoot - 1);
  for (int probe = 0; probe < kYlRoot; ++probe) {
    const unsigned long long old = atomCas64(&S.root.key[h], 0ull, key);
    if (old == 0ull || old == key) {
      return (int)h;
    }
    h = (h + 1u) & (kYlRoot - 1);
  }
  return -1;
}
template <typename YlOrphTab>
FLTX_DEV void ylOrphAdd(YlOrphTab& tab, unsigned long long key, int li) {
  constexpr int kYlOrph = YlOrphTab::kSlots;
  uint32_t h = xlHash(key) & (kYlOrph - 1);
  for (;;) {
    const unsigned long long old = atomCas64(&tab.key[h], 0ull, key);
    if (old == 0ull || old == key) {
      atomOr64(&tab.lanes[li >> 6][h], 1ull << (li & 63));
      return;
    }
    h = (h + 1u) & (kYlOrph - 1);
  }
}
template <typename YlOrphTab>
FLTX_DEV int ylOrphFind(const YlOrphTab& tab, unsigned long long key) {
  constexpr int kYlOrph = YlOrphTab::kSlots;
  uint32_t h = xlHash(key) & (kYlOrph - 1);
  for (;;) {
    const unsigned long long k = tab.key[h];
    if (k == key) {
      return (int)h;
    }
    if (k == 0ull) {
      return -1;
    }
    h = (h + 1u) & (kYlOrph - 1);
  }
}

/* n-gram score of LM word `word` after LM state `sid` (KenLM::score, lm/KenLM.cpp:66-86).  The
 * contexts of the states live in HBM (DecodeParams::stateCtx) and are written by other waves of
 * this workgroup in earlier frames: read past the L1.  wantCtx: the context of the resulting
 * state goes to state `outSid`. */
FLTX_DEV float ylNgram(const DecodeParams& P, int b, uint32_t sid, uint32_t word, int32_t* out) {
  const int Lc = P.lmOrder - 1;
  const uint32_t* ctx = (const uint32_t*)(P.stateCtx + ((size_t)b * P.stateCap + sid) * Lc);
  int32_t c[kMaxNgramOrder];
#pragma unroll
  for (int q = 0; q < kMaxNgramOrder; ++q) {
    c[q] = q < Lc ? (int32_t)loadCoherent32(ctx + q) : 0;
  }
  if (Lc <= 3) { /* (models up to order 4: two thirds of the unrolled look-up) */
    return ngScoreT<4>(P, c, word, out);
  }
  return ngScore(P, c, word, out);
}
FLTX_DEV uint32_t ylLmWord(const DecodeParams& P, int usr) {
  return (usr >= 0 && usr < P.nUsr) ? (uint32_t)P.usrToLm[usr] : (uint32_t)P.lmUnk;
}

/* why an utterance leaves this engine for the general one (fltx_decoder_get "fallback_reasons": bit r set):
 * 1 = a token wave's pairs tie beyond its rounds / no bound to rank them against, 2 = merge table full, 4 = the frame's
 * best candidate is not finite, 5 = more exact ties in the K-th best's bin than the pairwise list holds, 6 = LM-state
 * memo nearly full, 7 = more pairs than a token wave can list (3: another wave gave up -- not recorded) */
#ifdef FLTX_EMU
#define YL_WHY(r)                                                                          \
  do {                                                                                     \
    if ((r) != 3 && deadWhy == 0) {                                                        \
      deadWhy = (r);                                                                       \
    }                                                                                      \
    if (getenv("FLTX_YL_WHY") && lane == 0) {                                              \
      fprintf(stderr, "ylane: utterance %d wave %d gives up, reason %d\n", b, wave, (r)); \
    }                                                                                      \
  } while (0)
#else
#define YL_WHY(r)                     \
  do {                                \
    if ((r) != 3 && deadWhy == 0) {   \
      deadWhy = (r);                  \
    }                                 \
  } while (0)
#endif
#define FLTX_YLPROF(i)                                        \
  do {                                                        \
    if (PROF && P.prof && (int)threadIdx.x == P.profThread) { \
      const unsigned long long t_ = devClock();               \
      acc[(i)] += t_ - tPrev;                                 \
      tPrev = t_;                                             \
    }                                                         \
  } while (0)

/* NG lane groups of 64; R candidate pairs per token-wave thread; LMK = 0: ZeroLM over an
 * unsmeared lexicon (every LM term is zero and left out), 1: smeared trie and / or n-gram LM;
 * HM = 1: the LM-state memo lives in HBM and, with two lane groups, four token waves list twice the
 * pairs each (512 threads: two workgroups share a CU) */
template <int NG, int R, int LMK, int HM, bool PROF>
FLTX_DEV void ylaneUtterance(const DecodeParams& P, char* smem) {
  /* LMK: bit 0 = the LM terms (smeared trie and / or n-gram LM), bit 1 = ASG criterion (no blank; transitions[n * N + previous
   * token] enters the emitting-model score from the second frame on, LexiconDecoder.cpp:69-72,172-175) */
  constexpr int LMT = LMK & 1;
  constexpr bool ASG = (LMK & 2) != 0;
  /* bit 2 = several words per spelling (LexiconDecoder.cpp:113-142 loops over lex->labels): the word wave gets XR more
   * candidate slots per thread for the further words of the frame's lanes (short homophones fill a beam: the
   * reference's own test lexicon has 34 of 50 lanes on one three-word spelling in a frame) */
  constexpr bool ML = (LMK & 4) != 0;
  /* bit 3 = logAdd: the members of a merge group above the frame's threshold are summed (Utils.h:160-198), as in
   * fltx_xlane.h -- the candidates and the frame's best as without it, the sums once the threshold is known */
  constexpr bool LA = (LMK & 8) != 0;
  constexpr int XR = ML ? 2 * NG : 0;
  static_assert(!ML || NG <= 2, "several words per spelling: one and two lane groups (fltx_api.cpp prepare() says why)");
  constexpr int NW = NG + XR; /* candidate slots of a word-wave thread */
  constexpr int LG = NG > 2 ? NG : 2;
  using LDS = YlaneLdsT<LG, ML, LA>;
  LDS& S = *(LDS*)smem;
  constexpr int kYlLanes = LDS::kYlLanes, kYlRoot = LDS::kYlRoot, kYlOrph = LDS::kYlOrph;
  static_assert(XR * 64 <= (ML ? LDS::kYlX : 0), "xEmit / xLabel / nrList hold the extra slots' arrivals");
  constexpr int PAIRS = (NG == 2 && HM && !ML) ? 2 * kYlPairs : LDS::kYlPairs; /* pairs a token wave can list (ML: 768 threads, eight token waves) */
  constexpr int NS0 = R > NG ? R : (NG > 2 ? NG : 2);
  static_assert(NG == 1 || NG == 2 || NG == 4, "one, two or four lane groups");
  static_assert(NG <= 2 || HM == 1, "four lane groups: the LM-state memo lives in HBM");
  /* History slots in the lanes' records: 8 bits (0xFF = none) up to two groups, 13 bits beyond -- the back-trace masks
   * the parent-slot field of a record accordingly (BacktraceParams::packed = 8 / 13) */
  constexpr bool WIDE = NG > 2;
  constexpr uint32_t kNoHyp = WIDE ? 0x1FFFu : kSlNoHyp;
  auto infoTok = [](uint32_t i) { return (int)(i & 63u); };
  auto infoNB = [](uint32_t i) { return WIDE ? ((i >> 6) & 0x1FFFu) : ((i >> 16) & 0xFFu); };
  auto infoB = [](uint32_t i) { return WIDE ? (i >> 19) : (i >> 24); };
  auto mkInfo = [](uint32_t tok, uint32_t hNBv, uint32_t hBv) {
    return WIDE ? (tok | (hNBv << 6) | (hBv << 19)) : (tok | (hNBv << 16) | (hBv << 24));
  };
  static_assert(R * 64 <= PAIRS, "cand[] holds PAIRS pairs per token wave");
  const int b = P.uttMap ? P.uttMap[blockIdx.x] : (int)blockIdx.x;
  const int W = (int)blockDim.x, tid = (int)threadIdx.x;
  const int lane = laneId(), wave = waveUniform(waveId());
  const int nW = W >> 6;
  const int nTok = nW - NG - 2;
  const int wordWave = nW - 2, prepWave = nW - 1;
  const bool isTokW = wave < nTok, isSelfW = wave >= nTok && wave < nTok + NG, isWordW = wave == wordWave;
  const bool isSvc = wave == prepWave; /* (roles at run time) */
#ifndef FLTX_EMU
  if (!(P.tune & 1) && (isSelfW || isWordW || isSvc)) { /* the waves the token waves wait for win the issue arbitration of their SIMD
                                                            (C4: 10.9 -> 10.2 ms; tune bit 0 switches it off for measurements) */
    __builtin_amdgcn_s_setprio(3);
  }
#endif
  (void)isTokW;
  const int grp = isSelfW ? wave - nTok : 0;
  const int li = grp * 64 + lane; /* self waves: the lane this thread owns */
  const int K = P.K, N = P.N, TPW = P.yTpw;
  const int T = P.stepT ? P.stepT[b] : 0;
  const float* em = P.emissions ? P.emissions + P.emOff[b] : nullptr;
  const int64_t hbase = P.histOff[b];
  const double NEG = slNegInf();
  const int sil = P.sil, blank = P.blank;
  const int endTok = P.xEndTok;
  const double silScore = P.silScore, wordScore = P.wordScore, beamThreshold = P.beamThreshold;
  const double lmWeight = P.lmWeight;
  const bool ngram = LMT != 0 && P.lmKind != 0;
  const float* const trans = P.transitions;
  /* emitting-model score of token n after a hypothesis whose token is `prev` (the reference adds the two floats as
   * doubles before the hypothesis' score) */
  /* (threads without a live lane or pair come through here with whatever their slot holds -- six-bit token fields, so
   * up to 63 * N + 63: the index is clamped to the table, their result is never used) */
  const uint32_t transLast = (uint32_t)(N * N - 1);
  auto emScore = [&](double e, int n, int prev, int t) {
    if (ASG && t > 0) {
      const uint32_t at = (uint32_t)(n * N + prev);
      e = e + (double)trans[at < transLast ? at : transLast];
    }
    return e;
  };
  /* pairs beyond which a token wave ranks its own (tests lower it, never below the beam: the wave's K best must fit) */
  const int rankAt = (P.yRankAt > 0 && P.yRankAt < R * 64) ? (P.yRankAt > K ? P.yRankAt : K) : R * 64;
  int2* const histPT = P.histPT;
  int32_t* const histW = P.histW;
  const XNode* const xnode = P.xnode;
  const float* const xdelta = P.xdelta;
  auto& L = S.L;
  unsigned long long acc[8] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull};
  unsigned long long tPrev = devClock();
  uint32_t nScored = 0u;

  /* ---- decodeBegin (LexiconDecoder.cpp:21-30): the start state at the root ------------- */
  for (int i = tid; i < 2 * kYlLanes; i += W) {
    ((unsigned long long*)S.cmask)[i] = 0ull;
  }
  for (int i = tid; i < 2 * kSlNB; i += W) {
    ((uint32_t*)S.hist)[i] = 0u;
  }
  for (int i = tid; i < kYlRoot; i += W) {
    if constexpr (LA) {
      S.rootAcc[i] = 0.0;
    }
    S.root.key[i] = 0ull;
    S.root.best[i] = 0ull;
    S.root.lane[i] = 0u;
    S.root.minLane[i] = 0xFFFFFFFFu;
  }
  for (int i = tid; i < kYlOrph; i += W) {
    S.orph[0].key[i] = 0ull;
    S.orph[1].key[i] = 0ull;
#pragma unroll
    for (int g = 0; g < LG; ++g) {
      S.orph[0].lanes[g][i] = 0ull;
      S.orph[1].lanes[g][i] = 0ull;
    }
  }
  /* memo in HBM: as many slots as the host sized it for (long utterances create more LM states than the LDS memo holds) */
  const int memoSlots = HM ? (int)P.ymemoSlots : kYlMemo;
  const uint32_t memoMask = (uint32_t)memoSlots - 1u;
  unsigned long long* const memo = HM ? P.ymemo + (size_t)b * (size_t)memoSlots : S.memo;
  uint16_t* const candW = &S.cand[0][0] + (size_t)wave * PAIRS; /* (token waves: wave < 8, or < 4 with twice the pairs) */
  constexpr int PC = LG > 2 ? 0 : 512; /* pairs of a token wave whose score is kept in LDS while the wave ranks them */
  float* const pcW = S.pscore + (size_t)(LG > 2 ? 0 : wave) * PC;
  for (int i = tid; i < memoSlots; i += W) {
    memo[i] = 0ull; /* (HBM: at L2 before the barrier below, where the word wave's atomics will find it) */
  }
  if (tid < 32) {
    S.off[tid] = 0u;
  }
  if (tid < 16) {
    S.scal[tid] = 0u;
  }
  if (tid < kYlMaxGroups + 4) {
    S.offH[tid] = 0u;
  }
  if (tid == 0) {
    const XNode r0 = xnode[0];
    S.rootNode = r0;
    L.nb[0] = 0.0;
    L.b[0] = NEG;
    L.lmNB[0] = 0.0;
    L.lmB[0] = 0.0;
    L.childMask[0] = r0.childMask;
    L.kidsMask[0] = r0.kidsMask;
    L.info[0] = mkInfo((uint32_t)sil, 0u, kNoHyp);
    L.link[0] = 0u;
    L.lmSid[0] = 0u;
    L.node[0] = 0u;
    L.parent[0] = 0u;
    L.firstChild[0] = r0.firstChild;
    L.endLabel[0] = r0.endLabel0;
    if constexpr (ML) {
      L.endExtra[0] = P.xextra ? P.xextra[0] : 0u;
      L.endLmX[0] = kYlNoLm;
      L.endLmX[kYlLanes] = kYlNoLm;
      S.rootExtra = L.endExtra[0];
    }
    L.dPar[0] = 0x7FFFFFFFu;
    L.dWord[0] = -1;
    L.maxScore[0] = r0.maxScore;
    L.delta[0] = 0.0f;
    L.endLm[0] = kYlNoLm;
    L.endWord[0] = (ngram && r0.endLabel0 >= 0) ? (int32_t)ylLmWord(P, r0.endLabel0) : -1;
    S.rootWord = L.endWord[0];
#pragma unroll
    for (int g = 0; g < LG; ++g) {
      S.alive[0][g] = g == 0 ? 1ull : 0ull;
      S.alive[1][g] = 0ull;
    }
    S.row[0].nev = 0u;
    S.row[1].nev = 0u;
    S.row[0].dead = 0u;
    S.row[1].dead = 0u;
    S.bestKey[0] = 0ull;
    S.bestKey[1] = 0ull;
    S.lb[0] = 0ull;
    S.lb[1] = 0ull;
    S.lmNext = 1u;
    histPT[hbase] = make_int2((int)kNoHyp, sil);
    histW[hbase] = -1;
    if (ngram) { /* KenLM::start(false): context = <s> (KenLM.cpp:57) */
      const int Lc = P.lmOrder - 1;
      int32_t* c0 = P.stateCtx + (size_t)b * P.stateCap * Lc;
      uint32_t nd = 0;
      float pr;
      const bool ok = ngFind(P, 0u, (uint32_t)P.lmBos, nd, pr);
      for (int q = 0; q < Lc; ++q) {
        c0[q] = (q == 0 && ok) ? (int32_t)nd : 0;
      }
    }
  }
  if (tid > 0 && tid < K) {
    histPT[hbase + tid] = make_int2((int)kNoHyp, -1);
  }
  float rowA = 0.0f, rowB = 0.0f;
  if (isSvc) {
    const float v0 = (T > 0 && lane < N) ? em[lane] : 0.0f;
    rowA = (T > 1 && lane < N) ? em[(size_t)1 * N + lane] : 0.0f;
    rowB = (T > 2 && lane < N) ? em[(size_t)2 * N + lane] : 0.0f;
    SlRowRegs r0 = slRowScan(P, v0, false, 0.0);
    slRowStore(P, S, 0, r0, 2);
    slRowStore(P, S, 1, r0, 2);
  }
#ifndef FLTX_EMU
  __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory"); /* the start state's context is read back from HBM */
#endif
  ldsBarrier();

  int winShift = kSlCoarseShift, winBase = kSlCoarseBase;
  bool dead = false;
  int deadWhy = 0; /* this wave's own reason for giving up (YL_WHY), 0 = none / another wave's */

  /* RL = role of the wave, compile time as the parity (0 token, 1 own groups, 2 word ends, 3 staging):
   * see fltx_xlane.h */
  auto frameStep = [&](auto PT, auto RL, float& rowReg, const int t) {
    constexpr int p = decltype(PT)::value, q = p ^ 1;
    constexpr int role = decltype(RL)::value;
    constexpr bool isTok = role == 0, isSelf = role == 1, isWord = role == 2, isSvc = role == 3;
    constexpr int NS = (isWord && NW > NS0) ? NW : NS0; /* candidate slots of a thread (the word wave's further words: its own count) */
    const int frameOut = t + 1;
    const int64_t hrow = hbase + (int64_t)frameOut * K;
    /* ---- phase 1a: candidates, the merge table, the frame's best --------------------------- */
    const int silPos = S.row[p].silPos;
    const unsigned long long allow = S.row[p].allow;
    unsigned long long aliveG[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      aliveG[g] = S.alive[p][g];
    }
    double cs[NS], clm[NS];
    double coth[LA ? NS : 1]; /* logAdd: a candidate's other member (the lane's other hypothesis), -inf = none */
#pragma unroll
    for (int j = 0; j < (LA ? NS : 1); ++j) {
      coth[j] = NEG;
    }
    int cbin[NS];
    bool cok[NS];
    uint32_t cinf[NS]; /* token waves: lane | token << 8 | history slot of the source << 16;  word wave: history slot << 16 */
    uint32_t cnode[NS]; /* token waves, word wave (odd slots): the child node */
    float cdl[NS];      /* ... and its smearing difference */
    uint32_t cpl[NS];   /* LM state of the lane the candidate comes from */
    uint32_t cpn[NS];   /* its node (word wave, even slots: the word) */
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      cs[j] = NEG;
      clm[j] = 0.0;
      cbin[j] = kSlInvalid;
      cok[j] = false;
      cinf[j] = 0u;
      cnode[j] = 0u;
      cdl[j] = 0.0f;
      cpl[j] = 0u;
      cpn[j] = 0u;
    }
    /* own lane (self waves) */
    bool live = false, atRoot = false;
    double nb = NEG, bb = NEG, m = NEG, lmM = 0.0, lmOwnNB = 0.0;
    uint32_t info = 0u, hypNB = kNoHyp, hypB = kNoHyp, hypM = kNoHyp, parR = kNoHyp;
    int last = 0, pl = -1, rootSlot = -1;
    bool whichB = false;
    double lmR = 0.0; /* LM score of the winning member of the stay group */
    /* logAdd, self waves: the members of the lane's blank and stay groups besides the best one */
    double laBlankO = NEG, laR0 = NEG, laR0o = NEG, laR1 = NEG, laR2 = NEG;
    /* word wave, per slot (a lane group's lane, or the thread's further word) */
    int nXtra = 0; /* further words this frame (word wave) */
    int wSlot[NW];
    bool wUseB[NW];
    uint32_t wHypB[NW], wHypM[NW];
#pragma unroll
    for (int g = 0; g < NW; ++g) {
      wSlot[g] = -1;
      wUseB[g] = false;
      wHypB[g] = kNoHyp;
      wHypM[g] = kNoHyp;
    }
    int nCand = 0; /* token waves: pairs in cand[wave] */
    /* staging wave: ranks the token beam of the next row in three pieces (see fltx_xlane.h) */
    SlRowRegs nextRow = {};
    const float rv = rowReg;
    int rk = 0;
    auto rankPart = [&](int m0, int m1) {
      for (int mm = m0; mm < m1; ++mm) {
        const float o = __uint_as_float(waveReadLane32(__float_as_uint(rv), mm));
        rk += (o > rv || (o == rv && mm < lane)) ? 1 : 0;
      }
    };
    const bool needRank = P.Kt < N && t + 1 < T;
    const int rk1 = N / 3, rk2 = 2 * N / 3;
    const bool fastRank = needRank && N <= 32 && isSvc && waveBallot(lane < N && !(rv == rv)) == 0ull;
    XlRank rs = {};
    FLTX_YLPROF(0);
    if (isSvc) {
#pragma unroll
      for (int g = 0; g < LG; ++g) {
        S.cmask[q][g * 64 + lane] = 0ull;
      }
      if (lane < 32) {
        S.off[lane] = 0u;
      }
      if (lane < kYlMaxGroups + 4) {
        S.offH[lane] = 0u;
      }
      if (lane == 0) {
        S.scal[SL_BCNT] = 0u;
        S.bestKey[q] = 0ull;
        S.lb[q] = 0ull;
      }
      if (fastRank) {
        rs = xlRankBegin(rv, N);
        xlRankRange<0, 4>(rs);
      } else if (needRank) {
        rankPart(0, rk1);
      }
      rowReg = (t + 3 < T && lane < N) ? em[(size_t)(t + 3) * N + lane] : 0.0f;
      /* blank, then the node's own token again (:89 with prevBlank): into the child, if it has children
       * and no lane -- this wave has the time */
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const int x = g * 64 + lane;
        const bool lv = ((aliveG[g] >> lane) & 1ull) != 0ull;
        const double xb = lv ? L.b[x] : NEG;
        const uint32_t xi = L.info[x];
        const int xl = infoTok(xi);
        const uint32_t xhB = infoB(xi);
        const unsigned long long cmk = L.childMask[x];
        const bool extLast = ((cmk & L.kidsMask[x]) >> xl) & 1ull;
        const bool allowLast = ((allow >> xl) & 1ull) != 0ull;
        const bool go = lv && xhB != kNoHyp && extLast && allowLast && ((S.cmask[p][x] >> xl) & 1ull) == 0ull;
        double cL = xb + S.eAll[p][xl];
        if (xl == sil) {
          cL = cL + silScore;
        }
        const uint32_t child = L.firstChild[x] + (uint32_t)popc64(cmk & ((1ull << xl) - 1ull));
        cnode[g] = child;
        cpl[g] = L.lmSid[x];
        cpn[g] = L.node[x];
        if (LMT) {
          const float dl = go ? xdelta[child] : 0.0f;
          cdl[g] = dl;
          cL = cL + lmWeight * (double)dl;
          clm[g] = L.lmB[x] + (double)dl;
        }
        cs[g] = cL;
        cok[g] = go && cL == cL;
        cinf[g] = (uint32_t)x | ((uint32_t)xl << 9) | (xhB << 16);
      }
    } else if (isTok) {
      /* the (lane, token) pairs with a child that has children and no lane of its own yet, the
       * node's own token excepted (that one needs the blank in between): LexiconDecoder.cpp:89-110 */
      unsigned long long ext[NG];
      double mg[NG];
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const int x = g * 64 + lane;
        const bool lv = ((aliveG[g] >> lane) & 1ull) != 0ull;
        const unsigned long long own = 1ull << (L.info[x] & 63u);
        ext[g] = lv ? (L.childMask[x] & L.kidsMask[x] & ~S.cmask[p][x] & (ASG ? ~0ull : ~own)) : 0ull;
        const double xnb = L.nb[x], xb = L.b[x];
        mg[g] = xb > xnb ? xb : xnb;
      }
      /* Pairs that cannot reach the threshold are left out at once: the frame's best is at least
       * lb (a stay / blank candidate of a lane that survived the last frame), and a pair scores at
       * most m + e (+ silScore) + the largest smearing term of the lexicon. */
      const unsigned long long lbk = S.lb[p];
      const double lbBest = lbk != 0ull ? f64FromKey(lbk) : NEG;
      const double thrLB = lbBest - beamThreshold;
      const double bterm = (LMT ? P.yBound : 0.0) + (ASG ? P.yTransMax : 0.0);
      for (int j = 0; j < TPW; ++j) {
        const int pos = wave * TPW + j;
        const unsigned long long tbj = S.tokBit[p][pos];
        if (tbj == 0ull) {
          continue;
        }
        double ej = S.eTok[p][pos];
        if (pos == silPos) {
          ej = ej + silScore;
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          const bool has = (ext[g] & tbj) != 0ull && (mg[g] + ej) + bterm >= thrLB;
          const unsigned long long bal = waveBallot(has);
          if (bal != 0ull) {
            const int at = nCand + wavePrefixCount(bal);
            if (has && at < PAIRS) {
              candW[at] = (uint16_t)((g * 64 + lane) | (j << 9));
            }
            nCand += popc64(bal);
          }
        }
      }
      if (nCand > PAIRS) { /* (the host sizes the waves' shares so that this cannot happen) */
        dead = true; YL_WHY(7);
        nCand = PAIRS;
      }
      waveSync();
      if (nCand > rankAt && !dead) {
        /* More pairs than the threads' rounds take (the beam fans out at the start of an utterance): none but the K
         * best of this wave's own pairs can be among the frame's K best, so the wave ranks its pairs by itself and
         * goes on with the best R * 64 >= K of them at most.  The order key is the float bit pattern of the distance to
         * the wave's best pair (monotone in the score); the cut is found by counting per bin -- the coarse window first, then, as long
         * as the bin that holds the K-th best would overfill the rounds, the finest window that spans that bin -- and
         * is a plain threshold on the key, so what is kept is a superset of the wave's K best whatever the ties. */
        uint32_t* wh = S.whist[wave];
        double rankRef = 0.0; /* the distance is taken to the wave's best pair */
        /* the score of pair `id` (NaN: not a candidate) -- LDS reads and, with the LM terms, a load from the trie's
         * smearing table in HBM: done once per pair, the float of it kept in pcW for the passes that follow */
        auto pairScore = [&](int id) -> double {
          const bool valid = id < nCand;
          const uint32_t c16 = valid ? (uint32_t)candW[id] : 0u;
          const int x = (int)(c16 & 0x1FFu), pos = wave * TPW + (int)(c16 >> 9);
          const double xnb = L.nb[x], xb = L.b[x];
          const int n = (int)S.tokId[p][pos];
          double c = (xb > xnb ? xb : xnb) + emScore(S.eTok[p][pos], n, (int)(L.info[x] & 63u), t);
          if (pos == silPos) {
            c = c + silScore;
          }
          if (LMT) {
            const uint32_t child = L.firstChild[x] + (uint32_t)popc64(L.childMask[x] & ((1ull << n) - 1ull));
            const float dl = valid ? xdelta[child] : 0.0f;
            c = c + lmWeight * (double)dl;
          }
          return valid ? c : __builtin_nan("");
        };
        /* order key of a pair: the float bits of the distance from the wave's best pair to the pair's score AS A FLOAT
         * (one function for every pair, kept or priced again: monotone in the score, which is all the cut needs) */
        auto pairKey = [&](int id) -> uint32_t { /* 0xFFFFFFFF: not a candidate */
          float cf;
          if (PC > 0 && id < PC) {
            cf = pcW[id];
          } else {
            cf = (float)pairScore(id);
          }
          if (!(cf == cf) || id >= nCand) {
            return 0xFFFFFFFFu;
          }
          float dd = (float)(rankRef - (double)cf);
          dd = dd > 0.0f ? dd : 0.0f;
          return __float_as_uint(dd);
        };
        unsigned long long bLo = 0ull, bHi = 0x7FFFFFFFull;
        int shift = kSlCoarseShift, base = kSlCoarseBase, before = 0;
        uint32_t hiCut = 0x7FFFFFFFu;
        bool okCut = true;
        /* pairs whose key equals tieVal (a float holds fewer bits than the scores: different scores can share one) are
         * ranked by the scores themselves when they straddle the cut: their order keys and list indices go to the
         * wave's count area, tieRoom of them stay */
        constexpr int kTieCap = 96;
        unsigned long long* const tieKey = (unsigned long long*)wh;         /* [kTieCap] */
        uint16_t* const tieId = (uint16_t*)(wh + 2 * kTieCap);             /* [2 * (kSlNB - 2 * kTieCap)] >= kTieCap */
        static_assert(2 * kTieCap + kTieCap / 2 <= kSlNB, "the tie list fits the wave's count area");
        uint32_t tieVal = 0xFFFFFFFFu;
        int tieRoom = 0, nTie = 0;
        { /* Every pair is priced once.  The listing above left out what an upper bound of the score puts below the
           * threshold; the score itself does so for more (the bound carries the largest smearing term of the lexicon),
           * and often for enough that nothing is left to rank.  The list is compacted in place, in order.
           * (The reference of the distances: lb would do as long as no pair beats it; pairs above it would all share
           * key 0 -- it is the wave's best pair.) */
          unsigned long long mk = 0ull;
          int kept0 = 0;
          for (int c0 = 0; c0 < nCand; c0 += 64) {
            const int id = c0 + lane;
            const uint32_t c16 = id < nCand ? (uint32_t)candW[id] : 0u;
            const double c = pairScore(id);
            const bool k0 = c == c && c >= thrLB;
            const unsigned long long bal = waveBallot(k0);
            waveSync();
            if (k0) {
              const int at = kept0 + wavePrefixCount(bal);
              candW[at] = (uint16_t)c16;
              if (PC > 0 && at < PC) {
                pcW[at] = (float)c;
              }
            }
            const unsigned long long k1 = k0 ? f64Key(c) : 0ull;
            mk = k1 > mk ? k1 : mk;
            kept0 += popc64(bal);
            waveSync();
          }
          nCand = kept0;
          mk = waveMax64(mk);
          okCut = mk != 0ull;
          rankRef = okCut ? f64FromKey(mk) : 0.0;
        }
        while (okCut && nCand > rankAt) {
          ((uint4*)wh)[lane] = make_uint4(0u, 0u, 0u, 0u);
          waveSync();
          for (int c0 = 0; c0 < nCand; c0 += 64) {
            const uint32_t kb = pairKey(c0 + lane);
            if (kb != 0xFFFFFFFFu && (unsigned long long)kb >= bLo && (unsigned long long)kb <= bHi) {
              int q = (int)(kb >> shift) - base;
              q = q < 0 ? 0 : (q > kSlNB - 1 ? kSlNB - 1 : q);
              atomAdd32(&wh[q], 1u);
            }
          }
          waveSync();
          const SlScan ws = slScan(wh, K - before, false);
          if (!ws.crossed) { /* fewer than that in the bracket: all of them stay */
            hiCut = (uint32_t)bHi;
            break;
          }
          const unsigned long long v = (unsigned long long)(ws.bstar + base);
          unsigned long long l2 = bLo, h2 = bHi;
          if (ws.bstar > 0 || base == 0) {
            const unsigned long long e = v << shift;
            l2 = e > l2 ? e : l2;
          }
          if (ws.bstar < kSlNB - 1) {
            const unsigned long long e = ((v + 1ull) << shift) - 1ull;
            h2 = e < h2 ? e : h2;
          }
          if (before + ws.cum + ws.cnt <= rankAt) {
            hiCut = (uint32_t)h2;
            break;
          }
          before += ws.cum;
          bLo = l2;
          bHi = h2;
          if (bLo >= bHi) { /* one key value, and more pairs of it than the rounds take */
            if (ws.cnt > kTieCap) {
              okCut = false;
              break;
            }
            tieVal = (uint32_t)bLo;
            tieRoom = rankAt - before;
            hiCut = tieVal; /* (the members of tieVal among them: decided below) */
            waveSync();
            for (int c0 = 0; c0 < nCand; c0 += 64) {
              const int id = c0 + lane;
              const bool mem = pairKey(id) == tieVal;
              const unsigned long long bal = waveBallot(mem);
              if (mem && nTie + wavePrefixCount(bal) < kTieCap) {
                tieKey[nTie + wavePrefixCount(bal)] = f64Key(pairScore(id));
                tieId[nTie + wavePrefixCount(bal)] = (uint16_t)id;
              }
              nTie += popc64(bal);
            }
            okCut = nTie <= kTieCap;
            waveSync();
            break;
          }
          int ns = 0;
          while (((bHi >> ns) - (bLo >> ns)) > (unsigned long long)(kSlNB - 1)) {
            ++ns;
          }
          shift = ns;
          base = (int)(bLo >> ns);
          waveSync();
        }
        if constexpr (LA) {
          /* The ranking above orders the pairs by their best member; the frame selects on the sums, which lie at most
           * ln 2 above it (two members).  At least K pairs of this wave have a best member within the cut, so a pair
           * whose best member is more than ln 2 beyond it cannot be among the frame's K best either way: the cut moves
           * out by ln 2 (and what the float keys round away); ties at the cut are not resolved here (general path). */
          if (tieVal != 0xFFFFFFFFu) {
            okCut = false;
          } else if (hiCut < 0x7F000000u) {
            hiCut = __float_as_uint(__uint_as_float(hiCut) + 0.70f);
          }
        }
        int kept = 0;
        for (int c0 = 0; c0 < nCand && nCand > rankAt; c0 += 64) {
          const int id = c0 + lane;
          const uint32_t c16 = id < nCand ? (uint32_t)candW[id] : 0u;
          const uint32_t kb = pairKey(id);
          bool keep = kb <= hiCut; /* (0xFFFFFFFF: never) */
          if (kb == tieVal && keep) { /* among equal keys the better scores, then the lower list index */
            const unsigned long long k = f64Key(pairScore(id));
            int rank = 0;
            for (int i = 0; i < nTie && i < kTieCap; ++i) {
              const unsigned long long k2 = tieKey[i];
              rank += (k2 > k || (k2 == k && (int)tieId[i] < id)) ? 1 : 0;
            }
            keep = rank < tieRoom;
          }
          const unsigned long long bal = waveBallot(keep);
          waveSync();
          if (keep) {
            candW[kept + wavePrefixCount(bal)] = (uint16_t)c16;
          }
          kept += popc64(bal);
          waveSync();
        }
        nCand = nCand > rankAt ? kept : nCand;
        if ((!okCut && nCand > 0) || nCand > rankAt) { /* ties by the hundred: general path */
          dead = true; YL_WHY(1);
          nCand = nCand > R * 64 ? R * 64 : nCand;
        }
      }
      if (dead && lane == 0) {
        S.bestKey[p] = ~0ull;
      }
      waveSync();
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int id = r * 64 + lane;
        if (r * 64 < nCand) {
          const bool valid = id < nCand;
          const uint32_t c16 = valid ? (uint32_t)candW[id] : 0u;
          const int x = (int)(c16 & 0x1FFu), pos = wave * TPW + (int)(c16 >> 9);
          const double xnb = L.nb[x], xb = L.b[x];
          const uint32_t xi = L.info[x];
          const double ev = S.eTok[p][pos];
          const int n = (int)S.tokId[p][pos];
          const bool wb = xb > xnb;
          double c = (wb ? xb : xnb) + emScore(ev, n, infoTok(xi), t);
          double co = (wb ? xnb : xb) + emScore(ev, n, infoTok(xi), t); /* (logAdd: the lane's other hypothesis) */
          if (pos == silPos) {
            c = c + silScore;
            co = co + silScore;
          }
          const uint32_t hp = wb ? infoB(xi) : infoNB(xi);
          cinf[r] = (uint32_t)x | ((uint32_t)n << 9) | (hp << 16);
          const uint32_t child = L.firstChild[x] + (uint32_t)popc64(L.childMask[x] & ((1ull << n) - 1ull));
          cnode[r] = child;
          cpl[r] = L.lmSid[x];
          cpn[r] = L.node[x];
          if (LMT) {
            const float dl = valid ? xdelta[child] : 0.0f;
            cdl[r] = dl;
            c = c + lmWeight * (double)dl; /* lmScore = lex->maxScore - lexMaxScore, LexiconDecoder.cpp:96,101 */
            co = co + lmWeight * (double)dl;
            clm[r] = (wb ? L.lmB[x] : L.lmNB[x]) + (double)dl;
          }
          cs[r] = c;
          if constexpr (LA) {
            coth[r] = co;
          }
          cok[r] = valid && c == c;
        }
      }
    } else if (isSelf) {
      unsigned long long aliveMine = aliveG[0];
#pragma unroll
      for (int g = 1; g < NG; ++g) {
        aliveMine = g == grp ? aliveG[g] : aliveMine;
      }
      live = ((aliveMine >> lane) & 1ull) != 0ull;
      nb = live ? L.nb[li] : NEG;
      bb = live ? L.b[li] : NEG;
      info = L.info[li];
      last = infoTok(info);
      hypNB = infoNB(info);
      hypB = infoB(info);
      whichB = bb > nb;
      m = whichB ? bb : nb;
      hypM = whichB ? hypB : hypNB;
      atRoot = L.node[li] == 0u;
      pl = live ? (int)L.link[li] - 1 : -1;
      const int pi = pl >= 0 ? pl : 0;
      const double parNB = L.nb[pi], parB = L.b[pi];
      const uint32_t parInfo = L.info[pi];
      double lmNBv = 0.0, lmBv = 0.0, parLmNB = 0.0, parLmB = 0.0, dl = 0.0;
      if (LMT) {
        lmNBv = L.lmNB[li];
        lmBv = L.lmB[li];
        parLmNB = L.lmNB[pi];
        parLmB = L.lmB[pi];
        dl = (double)L.delta[li];
      }
      lmM = whichB ? lmBv : lmNBv;
      lmOwnNB = lmNBv;
      const double eBlank = S.eAll[p][ASG ? 0 : blank], eLast = S.eAll[p][last], eSil = S.eAll[p][sil];
      /* blank (:197-213): always tried */
      cs[0] = m + eBlank;
      if constexpr (LA) {
        laBlankO = (whichB ? nb : bb) + eBlank;
      }
      clm[0] = lmM;
      cok[0] = live && !ASG;
      /* stay (:168-194) + the trie parent's extension by the node's token (a "(1) try children"
       * candidate: needs the token in the token beam, carries the smearing difference) */
      const int lastP = infoTok(parInfo);
      const uint32_t h1 = infoNB(parInfo), h2 = infoB(parInfo);
      const bool allowLast = ((allow >> last) & 1ull) != 0ull;
      const bool hasNB = hypNB != kNoHyp;
      const bool has0 = atRoot ? true : hasNB;
      const bool has1 = pl >= 0 && allowLast && (ASG || last != lastP) && h1 != kNoHyp;
      const bool has2 = pl >= 0 && allowLast && h2 != kNoHyp;
      double r0 = (atRoot ? m : nb) + emScore(atRoot ? eSil : eLast, atRoot ? sil : last, last, t);
      double r1 = has1 ? parNB + emScore(eLast, last, lastP, t) : NEG;
      double r2 = has2 ? parB + eLast : NEG; /* (from the blank hypothesis: CTC only) */
      double r0o = atRoot ? (whichB ? nb : bb) + emScore(eSil, sil, last, t) : NEG; /* (logAdd: the root's other hypothesis stays too) */
      if (silScore != 0.0) {
        const bool ls = atRoot || last == sil;
        r0 = ls ? r0 + silScore : r0;
        r0o = ls ? r0o + silScore : r0o;
        r1 = ls ? r1 + silScore : r1;
        r2 = ls ? r2 + silScore : r2;
      }
      if (LMT) {
        r1 = r1 + lmWeight * dl;
        r2 = r2 + lmWeight * dl;
      }
      if constexpr (LA) {
        laR0 = has0 ? r0 : NEG;
        laR0o = r0o;
        laR1 = r1;
        laR2 = r2;
      }
      double cR = r0;
      parR = atRoot ? hypM : hypNB;
      lmR = atRoot ? lmM : lmNBv;
      if (has1 && (r1 > cR || (r1 == cR && h1 < parR))) {
        cR = r1;
        parR = h1;
        lmR = parLmNB + dl;
      }
      if (has2 && (r2 > cR || (r2 == cR && h2 < parR))) {
        cR = r2;
        parR = h2;
        lmR = parLmB + dl;
      }
      cs[1] = cR;
      clm[1] = lmR;
      cok[1] = live && (has0 || has1 || has2);
      if (live && atRoot) { /* words ending here this frame join through the merge table */
        rootSlot = ylRootFind(S, xlKey(L.dPar[li], L.dWord[li]));
        if (rootSlot >= 0) {
          S.root.lane[rootSlot] = (uint32_t)li + 1u;
          atomMax64(&S.root.best[rootSlot], f64Key(cR));
        }
      }
    } else if (isWord) {
      if (ngram) {
        /* lm.score(LM state, word) of the lanes that can end a word and have not asked yet: once per
         * lane, all of a frame's look-ups side by side (one trip to the n-gram tables) */
        int nReq = 0;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          const int x = g * 64 + lane;
          const bool lv = ((aliveG[g] >> lane) & 1ull) != 0ull;
          const bool need = lv && L.endLabel[x] >= 0 && L.endLm[x] == kYlNoLm;
          const unsigned long long bal = waveBallot(need);
          if (need) {
            S.lmReq[nReq + wavePrefixCount(bal)] = (uint16_t)x;
          }
          nReq += popc64(bal);
        }
        if (nReq > 0) {
          waveSync();
          for (int i0 = 0; i0 < nReq; i0 += 64) {
            if (i0 + lane < nReq) {
              const int x = (int)S.lmReq[i0 + lane];
              int32_t out[kMaxNgramOrder];
#pragma unroll
              for (int k = 0; k < kMaxNgramOrder; ++k) {
                out[k] = 0;
              }
              const float sc = ylNgram(P, b, L.lmSid[x], (uint32_t)L.endWord[x], out);
              L.endLm[x] = __float_as_uint(sc);
#pragma unroll
              for (int k = 0; k < kMaxNgramOrder - 1; ++k) {
                L.endCtx[k][x] = out[k];
              }
              ++nScored;
            }
          }
          waveSync();
        }
      }
      FLTX_YLPROF(7);
      /* the word-end candidate of lane x for word `el` (n-gram score lmBits when there is an n-gram LM) into slot J */
      auto wordEnd = [&](auto JT, int x, bool lv, int32_t el, uint32_t lmBits) {
        constexpr int g = decltype(JT)::value;
        const double xnb = lv ? L.nb[x] : NEG, xb = lv ? L.b[x] : NEG;
        const uint32_t xi = L.info[x];
        const int xl = infoTok(xi);
        const uint32_t xhNB = infoNB(xi), xhB = infoB(xi);
        const bool wb = xb > xnb;
        const double xm = wb ? xb : xnb;
        const bool xRoot = L.node[x] == 0u;
        const uint32_t xlm = L.lmSid[x];
        const double eEnd = S.eAll[p][endTok], eLast = S.eAll[p][xl];
        /* a word ends (:113-142); on the root the nb hypothesis would repeat its token (:114-122) */
        const bool useB = xRoot && xl == endTok;
        const bool can = lv && el >= 0 && ((allow >> endTok) & 1ull) != 0ull && (useB ? xhB != kNoHyp : true);
        double c = (useB ? xb : xm) + emScore(eEnd, endTok, xl, t);
        if (endTok == sil) {
          c = c + silScore;
        }
        float lmS = 0.0f;
        double srcLm = 0.0;
        if (LMT) {
          float sc = 0.0f;
          if (ngram && can) {
            sc = __uint_as_float(lmBits);
          }
          lmS = sc - (xRoot ? 0.0f : L.maxScore[x]); /* lmScore - lexMaxScore, LexiconDecoder.cpp:47,125 */
          srcLm = useB ? L.lmB[x] : (wb ? L.lmB[x] : L.lmNB[x]);
        }
        c = (c + lmWeight * (double)lmS) + wordScore;
        cs[g] = c;
        if constexpr (LA) { /* the lane's other hypothesis ends the word too (not on the root after its own token, :114-122) */
          double co = useB ? NEG : ((wb ? xnb : xb) + emScore(eEnd, endTok, xl, t));
          if (endTok == sil) {
            co = co + silScore;
          }
          coth[g] = (co + lmWeight * (double)lmS) + wordScore;
        }
        clm[g] = srcLm + (double)lmS;
        cok[g] = can && c == c;
        cinf[g] = (useB ? xhB : (wb ? xhB : xhNB)) << 16;
        cpl[g] = xlm;
        cpn[g] = (uint32_t)el;
        wUseB[g] = useB;
        wHypB[g] = xhB;
        wHypM[g] = wb ? xhB : xhNB;
        if (cok[g]) {
          wSlot[g] = ylRootFind(S, xlKey(xlm, el));
          if (wSlot[g] >= 0) {
            atomMax64(&S.root.best[wSlot[g]], f64Key(c));
          }
        }
        (void)eLast;
      };
      auto wordEndOfGroup = [&](auto GT) {
        constexpr int g = decltype(GT)::value;
        const int x = g * 64 + lane;
        const bool lv = ((aliveG[g] >> lane) & 1ull) != 0ull;
        wordEnd(GT, x, lv, L.endLabel[x], L.endLm[x]);
      };
      wordEndOfGroup(SlParity<0>());
      if constexpr (NG > 1) {
        wordEndOfGroup(SlParity<1>());
      }
      if constexpr (NG > 2) {
        wordEndOfGroup(SlParity<2>());
        wordEndOfGroup(SlParity<3>());
      }
      if constexpr (ML) {
        /* the further words of the spellings the frame's lanes can end (Trie.h:19: up to 6 per node): listed, 64 per
         * extra slot, and priced like the first */
        int nX = 0;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          const int x = g * 64 + lane;
          const bool lv = ((aliveG[g] >> lane) & 1ull) != 0ull;
          const int nl = lv ? (int)(L.endExtra[x] & 7u) : 0;
          for (int l = 1; l < 6; ++l) {
            const bool more = nl > l;
            const unsigned long long bal = waveBallot(more);
            if (bal == 0ull) {
              break;
            }
            const int at = nX + wavePrefixCount(bal);
            if (more && at < XR * 64) {
              S.xEmit[at] = (uint16_t)(x | (l << 8));
            }
            nX += popc64(bal);
          }
        }
        if (nX > XR * 64) {
          dead = true; YL_WHY(8); /* more further words in one frame than the word wave has slots for: general path */
          if (lane == 0) {
            S.bestKey[p] = ~0ull;
          }
          nX = 0;
        }
        nXtra = nX;
        if (nX > 0) {
          waveSync();
          auto extraRound = [&](auto RT) {
            constexpr int r = decltype(RT)::value;
            const int idx = r * 64 + lane;
            if (r * 64 >= nX) {
              return;
            }
            const bool mine = idx < nX;
            const int e = mine ? (int)S.xEmit[idx] : 0;
            const int x = e & 0xFF, l = e >> 8;
            int32_t el = -1;
            uint32_t lmBits = 0u;
            if (mine) {
              el = P.trieLabels[(int)(L.endExtra[x] >> 3) + l];
              S.xLabel[idx] = el;
              if (ngram) { /* the second and third word's score stays with the lane, as the first's does (endLm) */
                const bool keep = l <= 2;
                lmBits = keep ? L.endLmX[(l - 1) * kYlLanes + x] : kYlNoLm;
                if (lmBits == kYlNoLm) {
                  int32_t out[kMaxNgramOrder];
                  lmBits = __float_as_uint(ylNgram(P, b, L.lmSid[x], ylLmWord(P, el), out));
                  ++nScored;
                  if (keep) {
                    L.endLmX[(l - 1) * kYlLanes + x] = lmBits;
                  }
                }
              }
            }
            wordEnd(SlParity<NG + r>(), x, mine, el, lmBits);
          };
          extraRound(SlParity<0>());
          extraRound(SlParity<1>());
          if constexpr (XR > 2) {
            extraRound(SlParity<2>());
            extraRound(SlParity<3>());
          }
        }
      }
    }
    {
      bool full = isSelf && live && atRoot && rootSlot < 0;
#pragma unroll
      for (int g = 0; g < NW; ++g) {
        full = full || (isWord && cok[g] && wSlot[g] < 0);
      }
      if (waveBallot(full) != 0ull) {
        dead = true; YL_WHY(2); /* merge table full: general path (uniform after the barrier below via bestKey = ~0) */
        if (lane == 0) {
          S.bestKey[p] = ~0ull;
        }
      }
    }
    { /* the frame's best candidate (Utils.h:131-137) */
      unsigned long long mx = 0ull;
#pragma unroll
      for (int j = 0; j < NS; ++j) {
        const unsigned long long k = cok[j] ? f64Key(cs[j]) : 0ull;
        mx = k > mx ? k : mx;
      }
      if (waveBallot(mx != 0ull) != 0ull) {
        mx = waveMax64(mx);
        if (lane == 0) {
          atomMax64(&S.bestKey[p], mx);
        }
      }
    }
    FLTX_YLPROF(1);
    ldsBarrier(); /* A */
    /* slots this wave uses (uniform per wave): the loops over the slots stop there */
    const int nUsed = isTok ? (nCand + 63) >> 6 : (isSelf ? 2 : NG + ((nXtra + 63) >> 6));
    /* ---- phase 1b: threshold, merge-table verdicts, histogram ---------------------------- */
    const unsigned long long bk = S.bestKey[p];
    if (bk == 0ull || bk == ~0ull) {
      dead = true; YL_WHY(3);
      return;
    }
    const double best = f64FromKey(bk);
    if (!(best - best == 0.0)) {
      dead = true; YL_WHY(4);
      return;
    }
    const double thr = best - beamThreshold;
    /* logAdd: two members of one group (either may be absent = -inf, or below the threshold) */
    auto la2 = [&](double a, double c) {
      const bool oa = a >= thr, oc = c >= thr;
      if (oa && oc) {
        return a >= c ? slLogAdd(a, c) : slLogAdd(c, a);
      }
      return oa ? a : (oc ? c : NEG);
    };
    /* ... the share of a member in its merge-table slot's sum */
    auto laAdd = [&](int slot, double c) {
      if (c >= thr) {
        atomAddF64(&S.rootAcc[slot], exp(c - f64FromKey(S.root.best[slot])));
      }
    };
    if constexpr (LA) {
      if (isTok) {
#pragma unroll
        for (int j = 0; j < NS; ++j) {
          if (j < nUsed && cok[j]) {
            cs[j] = la2(cs[j], coth[j]);
          }
        }
      } else if (isSelf && live) {
        cs[0] = la2(cs[0], laBlankO);
        if (atRoot && rootSlot >= 0) { /* stay on the root: sil from either hypothesis; the words ending here join through the slot */
          laAdd(rootSlot, laR0);
          laAdd(rootSlot, laR0o);
        }
      } else if (isWord) {
#pragma unroll
        for (int g = 0; g < NW; ++g) {
          if (cok[g] && wSlot[g] >= 0) {
            laAdd(wSlot[g], cs[g]);
            laAdd(wSlot[g], coth[g]);
          }
        }
      }
      ldsBarrier(); /* A2: the slots' sums are complete */
    }
    if (isSelf && rootSlot >= 0) { /* the root lane's stay group takes the best word ending on it */
      const unsigned long long rb = S.root.best[rootSlot];
      if (rb > f64Key(cs[1])) {
        cs[1] = f64FromKey(rb);
        parR = kNoHyp; /* back-pointer and LM score: read from the slot in the build */
      }
    }
    if constexpr (LA) {
      if (isSelf && live) {
        if (atRoot) {
          if (rootSlot >= 0) {
            const double mBest = f64FromKey(S.root.best[rootSlot]);
            cs[1] = mBest >= thr ? mBest + log1p(S.rootAcc[rootSlot] - 1.0) : NEG;
          }
        } else { /* stay + the trie parent's extension: up to three members, folded best first (Utils.h:168-198) */
          double hi = laR0, mid = laR1, lo = laR2, tmp;
          if (mid > hi) { tmp = hi; hi = mid; mid = tmp; }
          if (lo > hi) { tmp = hi; hi = lo; lo = tmp; }
          if (lo > mid) { tmp = mid; mid = lo; lo = tmp; }
          double accv = hi >= thr ? hi : NEG;
          if (hi >= thr && mid >= thr) {
            accv = slLogAdd(accv, mid);
          }
          if (hi >= thr && lo >= thr) {
            accv = slLogAdd(accv, lo);
          }
          cs[1] = accv;
        }
      }
    }
    if (isWord) {
      /* of the arrivals that reach a slot's best the lowest lane represents them: the back-pointer a
       * root lane takes when a word ending on it beats its own stay, and the candidate for a new
       * root lane when nobody stands there.  (One wave handles the arrivals of all groups: the two
       * steps below need no barrier.) */
      bool top[NW];
#pragma unroll
      for (int g = 0; g < NW; ++g) {
        top[g] = cok[g] && S.root.best[wSlot[g] >= 0 ? wSlot[g] : 0] == f64Key(cs[g]);
        if (top[g]) {
          atomMin32(&S.root.minLane[wSlot[g]], (uint32_t)(g * 64 + lane));
        }
      }
      waveSync();
#pragma unroll
      for (int g = 0; g < NW; ++g) {
        const int sl = wSlot[g] >= 0 ? wSlot[g] : 0;
        const bool rep = top[g] && S.root.minLane[sl] == (uint32_t)(g * 64 + lane);
        if (rep) {
          S.root.winHyp[sl] = wUseB[g] ? wHypB[g] : wHypM[g];
          S.root.winWord[sl] = (int32_t)cpn[g];
          S.root.winLm[sl] = clm[g];
        }
        cok[g] = rep && S.root.lane[sl] == 0u;
        if (LA && cok[g]) {
          const double mBest = f64FromKey(S.root.best[sl]);
          cs[g] = mBest >= thr ? mBest + log1p(S.rootAcc[sl] - 1.0) : NEG;
        }
      }
    }
    if (isSvc && fastRank) {
      xlRankRange<4, 10>(rs);
    } else if (isSvc && needRank) {
      rankPart(rk1, rk2);
    }
    if (isSelf && live && !atRoot && pl < 0) { /* no parent lane: whoever creates it this frame finds this lane here */
      ylOrphAdd(S.orph[p], xlKey(L.lmSid[li], (int32_t)L.parent[li]), li);
    }
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      if (j >= nUsed) {
        continue;
      }
      if (cok[j] && cs[j] >= thr) {
        cbin[j] = slBin<LA>(best, cs[j], winShift, winBase);
        if (cbin[j] < kSlFar) {
          atomAdd32(&S.hist[p][cbin[j]], 1u);
        }
      }
    }
    FLTX_YLPROF(2);
    ldsBarrier(); /* 1 */
    /* ---- phase 2: which candidates survive (as fltx_slane.h) ------------------------------ */
    unsigned long long selMask[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      selMask[j] = 0ull;
    }
    /* (the usual frame in a straight line, the rare cases in a loop out of its way: see fltx_slane.h) */
    SlScan sc = slScan(S.hist[p], K, true);
    int shift = winShift, base = winBase;
    int lim = -1;
    uint32_t take = 0u;
    bool usual = false;
    if (sc.crossed) {
      if (sc.total <= K) {
        lim = kSlFar - 1;
        usual = true;
      } else if (sc.cnt == K - sc.cum) {
        lim = sc.bstar;
        usual = true;
      }
    }
    if (__builtin_expect(usual, 1)) {
#pragma unroll
      for (int j = 0; j < NS; ++j) {
        if (j >= nUsed) {
          continue;
        }
        selMask[j] = waveBallot(cbin[j] <= lim);
      }
    } else {
      unsigned long long bLo = 0ull, bHi = 0x7FFFFFFFull;
      bool full = false;
      for (;;) {
        if (!full && !sc.crossed) {
          int nFar = 0;
#pragma unroll
          for (int j = 0; j < NS; ++j) {
            if (j >= nUsed) {
              continue;
            }
            nFar += popc64(waveBallot(cbin[j] == kSlFar));
          }
          if (lane == 0 && nFar > 0) {
            atomAdd32(&S.hist[p][kSlFar], (uint32_t)nFar);
          }
          full = true;
          ldsBarrier();
          sc = slScan(S.hist[p], K, false);
          continue;
        }
        if (sc.total <= K) {
          lim = full ? kSlFar : kSlFar - 1;
          break;
        }
        const int need = K - sc.cum;
        if (sc.cnt == need) {
          lim = sc.bstar;
          break;
        }
        if (sc.cnt <= kSlBCap) {
#pragma unroll
          for (int j = 0; j < NS; ++j) {
            if (j >= nUsed) {
              continue;
            }
            if (cbin[j] == sc.bstar) {
              const uint32_t i = atomAdd32(&S.scal[SL_BCNT], 1u);
              S.bKey[i] = f64Key(cs[j]);
              S.bOrd[i] = ((uint32_t)wave << 16) | ((uint32_t)j << 8) | (uint32_t)lane;
            }
          }
          ldsBarrier();
          take |= slRankBin<NS>(S.bKey, S.bOrd, sc.cnt, need, wave, [&](int j) { return j < nUsed && cbin[j] == sc.bstar; },
                                [&](int j) { return f64Key(cs[j]); }); /* (broadcast + ballot: fltx_slane.h) */
          lim = sc.bstar - 1;
          break;
        }
        {
          const unsigned long long v = (unsigned long long)(sc.bstar + base);
          if (sc.bstar > 0 || base == 0) {
            const unsigned long long l2 = v << shift;
            bLo = l2 > bLo ? l2 : bLo;
          }
          if (sc.bstar < kSlNB - 1) {
            const unsigned long long h2 = ((v + 1ull) << shift) - 1ull;
            bHi = h2 < bHi ? h2 : bHi;
          }
          if (bLo >= bHi) {
            dead = true; YL_WHY(5);
            break;
          }
          int ns = 0;
          while (((bHi >> ns) - (bLo >> ns)) > (unsigned long long)(kSlNB - 1)) {
            ++ns;
          }
          shift = ns;
          base = (int)(bLo >> ns);
        }
        ldsBarrier();
        for (int i = tid; i < kSlNB; i += W) {
          S.hist[p][i] = 0u;
        }
        ldsBarrier();
        full = true;
#pragma unroll
        for (int j = 0; j < NS; ++j) {
          if (j >= nUsed) {
            continue;
          }
          if (cbin[j] != kSlInvalid) {
            cbin[j] = slBin<LA>(best, cs[j], shift, base);
            atomAdd32(&S.hist[p][cbin[j]], 1u);
          }
        }
        ldsBarrier();
        sc = slScan(S.hist[p], K, false);
      }
#pragma unroll
      for (int j = 0; j < NS; ++j) {
        if (j >= nUsed) {
          continue;
        }
        selMask[j] = waveBallot(cbin[j] <= lim || ((take >> j) & 1u) != 0u);
      }
    }
    if (dead) {
      return;
    }
    if (sc.total > K) {
      const int q15 = shift >= kSlFineShift ? (sc.bstar + base) << (shift - kSlFineShift)
                                            : (sc.bstar + base) >> (kSlFineShift - shift);
      winShift = kSlFineShift;
      winBase = q15 > kSlMid ? q15 - kSlMid : 0;
    }
    FLTX_YLPROF(3);
    /* ---- what the build needs and is known already ----------------------------------------- */
    uint32_t pend = 0u;         /* slots of this thread that become new lanes */
    int nNewWave = 0;
    int myNew[NS];
    XNode planX = {};           /* record of the child the first new lane of this thread stands on ... */
    int planOrph = -1;          /* ... and the slot of the orphan table that lists the lanes it adopts */
    int planSlot = -1;
    uint32_t rootSid[NW];
    bool wroteCtx = false; /* a new LM state's context is on its way to HBM */
    int rootOrph[NW];
    bool surv = false;
    uint32_t hNB = kNoHyp, hB = kNoHyp;
    unsigned long long balS = 0ull;
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      if (j >= nUsed) {
        continue;
      }
      myNew[j] = 0;
    }
#pragma unroll
    for (int g = 0; g < NW; ++g) {
      rootSid[g] = 0u;
      rootOrph[g] = -1;
    }
    uint32_t planLm = 0u, planPar = 0u, planExtra = 0u;
    int32_t planWord = -1;
    /* (reads nothing of the lanes: a slot whose lane drops out this frame is taken again in the same build) */
    auto planChild = [&](int j) {
      uint32_t cn = cnode[0];
      planLm = cpl[0];
      planPar = cpn[0];
#pragma unroll
      for (int i = 1; i < NS; ++i) {
        cn = i == j ? cnode[i] : cn;
        planLm = i == j ? cpl[i] : planLm;
        planPar = i == j ? cpn[i] : planPar;
      }
      planSlot = j;
      planX = xnode[cn];
      planExtra = (ML && P.xextra) ? P.xextra[cn] : 0u;
      planWord = ngram ? P.xlmword[cn] : -1;
      planOrph = ylOrphFind(S.orph[p], xlKey(planLm, (int32_t)cn));
      return cn;
    };
    uint32_t planNode = 0u;
    if (isTok || isWord || isSvc) {
#pragma unroll
      for (int j = 0; j < NS; ++j) {
        if (j >= nUsed) {
          continue;
        }
        if (selMask[j] != 0ull) {
          myNew[j] = nNewWave + wavePrefixCount(selMask[j]);
          nNewWave += popc64(selMask[j]);
        }
        pend |= (uint32_t)((selMask[j] >> lane) & 1ull) << j;
      }
      /* order of the new lanes: token waves, the word wave, the staging wave */
      const int slot = isTok ? wave : (isWord ? nTok : nTok + 1);
      if (lane > slot && lane <= nTok + 2 && nNewWave > 0) {
        atomAdd32(&S.off[lane], (uint32_t)nNewWave);
      }
      const uint32_t childPend = isWord ? 0u : pend; /* (the word wave's new lanes stand on the root) */
      if (childPend) {
        planNode = planChild(__builtin_ctz(childPend));
      }
      if (isWord) {
        /* New root lanes: the LM state's number (memo: the same (LM state, word) gives the same state
         * back), a new state's n-gram context, the lanes the root lane adopts.  The arrivals of both
         * groups side by side, one per thread. */
        int nr = 0;
#pragma unroll
        for (int g = 0; g < NW; ++g) {
          const bool want = ((pend >> g) & 1u) != 0u;
          const unsigned long long bal = waveBallot(want);
          if (want) {
            S.nrList[nr + wavePrefixCount(bal)] = (uint16_t)(g * 64 + lane);
          }
          nr += popc64(bal);
        }
        if (nr > 0) {
          waveSync();
          for (int i0 = 0; i0 < nr; i0 += 64) {
            if (i0 + lane < nr) {
              const int ar = (int)S.nrList[i0 + lane]; /* the arrival: slot * 64 + thread */
              const bool further = ML && ar >= NG * 64; /* a further word of its lane's spelling */
              const int x = further ? (int)(S.xEmit[ar - NG * 64] & 0xFFu) : ar;
              const uint32_t xlm = L.lmSid[x];
              const int32_t el = further ? S.xLabel[ar - NG * 64] : L.endLabel[x];
              const unsigned long long mkey = ((unsigned long long)(xlm + 1u) << 24) | (unsigned long long)(uint32_t)(el + 1);
              uint32_t h = xlHash(mkey) & memoMask;
              uint32_t sid = 0u;
              bool have = false; /* a number was taken from the counter for this state */
              for (int probe = 0;; ++probe) {
                unsigned long long cur = HM ? loadCoherent64(&memo[h]) : S.memo[h];
                if (cur == 0ull) {
                  if (!have) {
                    sid = atomAdd32(&S.lmNext, 1u);
                    have = true;
                  }
                  cur = atomCas64(&memo[h], 0ull, (mkey << 16) | (unsigned long long)(sid & 0xFFFFu));
                  if (cur == 0ull) { /* a new LM state */
                    if (sid > (uint32_t)(memoSlots / 4 * 3) || sid + 1u >= P.stateCap || sid >= 0xFFFFu ||
                        (uint32_t)(el + 1) >= (1u << 24)) {
                      atomOr32(&S.scal[YL_FLAG], 1u); /* memo nearly full: general path from the next frame on */
                    } else if (ngram) { /* its n-gram context: kept with the lane since the look-up */
                      const int Lc = P.lmOrder - 1;
                      int32_t* dst = P.stateCtx + ((size_t)b * P.stateCap + sid) * Lc;
                      int32_t fctx[kMaxNgramOrder];
                      if (further) { /* (a further word's was not kept: asked again, for the state's sake this time) */
#pragma unroll
                        for (int k = 0; k < kMaxNgramOrder; ++k) {
                          fctx[k] = 0;
                        }
                        (void)ylNgram(P, b, xlm, ylLmWord(P, el), fctx);
                      }
#pragma unroll
                      for (int k = 0; k < kMaxNgramOrder - 1; ++k) {
                        if (k < Lc) {
                          dst[k] = further ? fctx[k] : L.endCtx[k][x];
                        }
                      }
                      wroteCtx = true;
                    }
                    break;
                  }
                }
                if ((cur >> 16) == mkey) { /* (a number taken in vain stays unused) */
                  sid = (uint32_t)(cur & 0xFFFFull);
                  break;
                }
                h = (h + 1u) & memoMask;
                if (probe > memoSlots) {
                  atomOr32(&S.scal[YL_FLAG], 1u);
                  break;
                }
              }
              S.nrSid[ar] = sid;
              S.nrOrph[ar] = (int16_t)ylOrphFind(S.orph[p], xlKey(sid, 0));
            }
          }
          waveSync();
#pragma unroll
          for (int g = 0; g < NW; ++g) {
            if ((pend >> g) & 1u) {
              rootSid[g] = S.nrSid[g * 64 + lane];
              rootOrph[g] = (int)S.nrOrph[g * 64 + lane];
            }
          }
        }
      }
    } else if (isSelf) {
      const unsigned long long balB = selMask[0], balR = selMask[1];
      balS = balB | balR;
      const bool sR = ((balR >> lane) & 1ull) != 0ull;
      surv = ((balS >> lane) & 1ull) != 0ull;
      hNB = (uint32_t)(wavePrefixCount(balR) + wavePrefixCount(balB));
      hB = hNB + (sR ? 1u : 0u);
      const unsigned long long fr = ~balS;
      if ((fr >> lane) & 1ull) {
        S.freeList[grp][wavePrefixCount(fr)] = (uint8_t)lane;
      }
      if (lane == 0) {
        S.alive[q][grp] = balS;
        S.surv[grp] = balS;
        S.nFree[grp] = (uint32_t)popc64(fr);
      }
      if (lane > grp && lane <= NG) {
        atomAdd32(&S.offH[lane], (uint32_t)(popc64(balR) + popc64(balB)));
      }
    }
    if (isSvc && t + 1 < T) {
      nextRow.v = rv;
      nextRow.allow = N >= 64 ? ~0ull : ((1ull << N) - 1ull);
      if (fastRank) {
        xlRankRange<10, 16>(rs);
        nextRow.allow = xlRankEnd(rs, N, P.Kt);
      } else if (needRank) {
        rankPart(rk2, N);
        nextRow.allow = waveBallot(lane < N && rk < P.Kt);
      }
      nextRow.listMask = nextRow.allow;
      nextRow.nList = popc64(nextRow.allow);
      nextRow.best = 0.0; /* (this engine takes the frame's best from the candidates) */
      nextRow.dead = false;
      slRowStore(P, S, q, nextRow, P.Kt < N ? 1 : 0); /* (here: the lanes' own waves price their next stay / blank in the build) */
    }
    FLTX_YLPROF(4);
    ldsBarrier(); /* 2 */
    /* ---- phase 3: every survivor is written by the thread that evaluated it ---------------- */
    const unsigned long long surv0 = S.surv[0], surv1 = NG > 1 ? S.surv[1] : 0ull;
    const int nHSurv = (int)S.offH[NG];
    const int nNew = (int)S.off[nTok + 2];
    const uint32_t nFree0 = S.nFree[0];
    auto survives = [&](int x) {
      if constexpr (NG <= 2) {
        return (((x < 64 ? surv0 : surv1) >> (x & 63)) & 1ull) != 0ull;
      } else {
        return ((S.surv[x >> 6] >> (x & 63)) & 1ull) != 0ull;
      }
    };
    auto freeSlot = [&](int idx) {
      if constexpr (NG <= 2) {
        return (uint32_t)idx < nFree0 ? (int)S.freeList[0][idx] : 64 + (int)S.freeList[1][(idx - (int)nFree0) & 63];
      } else { /* the idx-th free slot over the groups in order */
        int g = 0, at = idx;
#pragma unroll
        for (int k = 0; k < NG - 1; ++k) {
          const int nf = (int)S.nFree[k];
          const bool past = g == k && at >= nf;
          at = past ? at - nf : at;
          g = past ? k + 1 : g;
        }
        return g * 64 + (int)S.freeList[g][at & 63];
      }
    };
    /* the lanes in the beam whose parent pair is the one in orphan slot `os`: they link to lane nl */
    auto adopt = [&](int os, int nl) {
      if (os < 0) {
        return;
      }
      unsigned long long toks = 0ull;
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        unsigned long long o = S.orph[p].lanes[g][os];
        while (o) {
          const int x = g * 64 + __builtin_ctzll(o);
          o &= o - 1ull;
          if (survives(x)) {
            L.link[x] = (uint32_t)nl + 1u;
            toks |= 1ull << (L.info[x] & 63u);
          }
        }
      }
      if (toks) {
        atomOr64(&S.cmask[q][nl], toks);
      }
    };
    /* a new lane on the planned child */
    auto newChild = [&](int idx, double c, double lmc, uint32_t ci, uint32_t child, float dl) {
      const int nl = freeSlot(idx);
      const uint32_t hyp = (uint32_t)(nHSurv + idx);
      const int x = (int)(ci & 0x1FFu), n = (int)((ci >> 9) & 0x7Fu);
      const uint32_t hp = ci >> 16;
      const XNode cx = planX;
      const bool ps = survives(x);
      L.nb[nl] = c;
      L.b[nl] = NEG;
      L.lmNB[nl] = lmc;
      L.lmB[nl] = 0.0;
      L.childMask[nl] = cx.childMask;
      L.kidsMask[nl] = cx.kidsMask;
      L.info[nl] = mkInfo((uint32_t)n, hyp, kNoHyp);
      L.link[nl] = ps ? (uint32_t)x + 1u : 0u;
      L.lmSid[nl] = planLm;
      L.node[nl] = child;
      L.parent[nl] = planPar;
      L.firstChild[nl] = cx.firstChild;
      L.endLabel[nl] = cx.endLabel0;
      if constexpr (ML) {
        L.endExtra[nl] = planExtra;
        L.endLmX[nl] = kYlNoLm;
        L.endLmX[kYlLanes + nl] = kYlNoLm;
      }
      L.dPar[nl] = 0u;
      L.dWord[nl] = -1;
      L.maxScore[nl] = cx.maxScore;
      L.delta[nl] = dl;
      L.endLm[nl] = kYlNoLm;
      L.endWord[nl] = planWord;
      atomOr64(&S.alive[q][nl >> 6], 1ull << (nl & 63));
      if (ps) {
        atomOr64(&S.cmask[q][x], 1ull << n);
      }
      histPT[hrow + hyp] = make_int2((int)hp, n);
      histW[hrow + hyp] = -1;
      adopt(planOrph, nl);
    };
    if (isSvc) {
      ((uint4*)S.hist[q])[lane] = make_uint4(0u, 0u, 0u, 0u);
      /* the merge table of the next frame starts empty (winHyp / winWord / winLm, which the self
       * waves read now, stay), and so does its orphan table: 16 bytes per lane and store */
      static_assert(kYlRoot % 256 == 0 && kYlOrph % 128 == 0, "the wipes below cover whole rounds of 64 lanes");
      const uint4 z4 = make_uint4(0u, 0u, 0u, 0u), f4 = make_uint4(~0u, ~0u, ~0u, ~0u);
#pragma unroll
      for (int i = 0; i < kYlRoot / 128; ++i) { /* 64-bit words: two per lane and store */
        ((uint4*)S.root.key)[lane + 64 * i] = z4;
        ((uint4*)S.root.best)[lane + 64 * i] = z4;
        if constexpr (LA) {
          ((uint4*)S.rootAcc)[lane + 64 * i] = z4;
        }
      }
#pragma unroll
      for (int i = 0; i < kYlRoot / 256; ++i) { /* 32-bit words: four */
        ((uint4*)S.root.lane)[lane + 64 * i] = z4;
        ((uint4*)S.root.minLane)[lane + 64 * i] = f4;
      }
#pragma unroll
      for (int i = 0; i < kYlOrph / 128; ++i) {
        ((uint4*)S.orph[q].key)[lane + 64 * i] = z4;
#pragma unroll
        for (int g = 0; g < LG; ++g) {
          ((uint4*)S.orph[q].lanes[g])[lane + 64 * i] = z4;
        }
      }
      /* blank-then-own-token lanes */
      const int offS = (int)S.off[nTok + 1];
      uint32_t cp = pend;
      bool first = true;
      while (waveBallot(cp != 0u) != 0ull) {
        if (cp) {
          const int j0 = __builtin_ctz(cp);
          cp &= cp - 1u;
          double c = cs[0], lmc = clm[0];
          int mn = myNew[0];
          uint32_t ci = cinf[0];
          float dl = cdl[0];
#pragma unroll
          for (int j = 1; j < NS; ++j) {
            c = j == j0 ? cs[j] : c;
            lmc = j == j0 ? clm[j] : lmc;
            mn = j == j0 ? myNew[j] : mn;
            ci = j == j0 ? cinf[j] : ci;
            dl = j == j0 ? cdl[j] : dl;
          }
          if (!first) {
            planNode = planChild(j0);
          }
          newChild(offS + mn, c, lmc, ci, planNode, dl);
        }
        first = false;
      }
    } else if (isTok) {
      /* most threads create at most one: every round takes each thread's lowest pending slot */
      bool first = true;
      while (waveBallot(pend != 0u) != 0ull) {
        if (pend) {
          const int j0 = __builtin_ctz(pend);
          pend &= pend - 1u;
          double c = cs[0], lmc = clm[0];
          int mn = myNew[0];
          uint32_t ci = cinf[0];
          float dl = cdl[0];
#pragma unroll
          for (int j = 1; j < NS; ++j) {
            c = j == j0 ? cs[j] : c;
            lmc = j == j0 ? clm[j] : lmc;
            mn = j == j0 ? myNew[j] : mn;
            ci = j == j0 ? cinf[j] : ci;
            dl = j == j0 ? cdl[j] : dl;
          }
          if (!first) {
            planNode = planChild(j0);
          }
          newChild((int)S.off[wave] + mn, c, lmc, ci, planNode, dl);
        }
        first = false;
      }
    } else if (isSelf) {
      if (li >= nHSurv + nNew && li < K) { /* unused slots of the history row */
        histPT[hrow + li] = make_int2((int)kNoHyp, -1);
      }
      if (surv) {
        const bool sB = ((selMask[0] >> lane) & 1ull) != 0ull, sR = ((selMask[1] >> lane) & 1ull) != 0ull;
        const uint32_t hb = S.offH[grp];
        L.nb[li] = sR ? cs[1] : NEG;
        L.b[li] = sB ? cs[0] : NEG;
        L.info[li] = mkInfo((uint32_t)last, sR ? hNB + hb : kNoHyp, sB ? hB + hb : kNoHyp);
        if (pl >= 0) {
          if (survives(pl)) {
            atomOr64(&S.cmask[q][pl], 1ull << last);
          } else {
            L.link[li] = 0u;
          }
        }
        double lmStay = lmR;
        if (sR) {
          uint32_t hp = parR;
          int32_t wd = -1;
          if (hp == kNoHyp) { /* a word ending on this root beat its own stay */
            hp = S.root.winHyp[rootSlot];
            wd = S.root.winWord[rootSlot];
            lmStay = S.root.winLm[rootSlot];
          }
          histPT[hrow + hNB + hb] = make_int2((int)hp, atRoot ? sil : last);
          histW[hrow + hNB + hb] = wd;
        }
        if (sB) {
          histPT[hrow + hB + hb] = make_int2((int)hypM, blank);
          histW[hrow + hB + hb] = -1;
        }
        if (LMT) {
          L.lmNB[li] = lmStay;
          L.lmB[li] = lmM;
        }
      }
      if (t + 1 < T) { /* what the next frame is sure to have: this lane's blank and stay (see the token waves) */
        unsigned long long k = 0ull;
        if (surv) {
          const bool sB = ((selMask[0] >> lane) & 1ull) != 0ull, sR = ((selMask[1] >> lane) & 1ull) != 0ull;
          const double nnb = sR ? cs[1] : NEG, nbb = sB ? cs[0] : NEG;
          const double nm = nbb > nnb ? nbb : nnb;
          const double c0 = ASG ? NEG : nm + S.eAll[q][ASG ? 0 : blank];
          k = (c0 == c0 && !ASG) ? f64Key(c0) : 0ull;
          if (atRoot || sR) {
            double c1 = (atRoot ? nm : nnb) + emScore(S.eAll[q][atRoot ? sil : last], atRoot ? sil : last, last, t + 1);
            if (atRoot || last == sil) {
              c1 = c1 + silScore;
            }
            const unsigned long long k1 = c1 == c1 ? f64Key(c1) : 0ull;
            k = k1 > k ? k1 : k;
          }
        }
        if (waveBallot(k != 0ull) != 0ull) {
          k = waveMax64(k);
          if (lane == 0) {
            atomMax64(&S.lb[q], k);
          }
        }
      }
    } else if (isWord) {
      const int offW = (int)S.off[nTok];
#pragma unroll
      for (int g = 0; g < NW; ++g) {
        const int x = g * 64 + lane;
        if ((pend >> g) & 1u) { /* a word ended and nobody stood on that root: a new root lane */
          const int idx = offW + myNew[g];
          const int nl = freeSlot(idx);
          const uint32_t hyp = (uint32_t)(nHSurv + idx);
          const uint32_t hp = cinf[g] >> 16;
          const uint32_t sid = rootSid[g];
          const XNode r0 = S.rootNode;
          const uint32_t xlm = cpl[g];
          const int32_t el = (int32_t)cpn[g];
          L.nb[nl] = cs[g];
          L.b[nl] = NEG;
          L.lmNB[nl] = clm[g];
          L.lmB[nl] = 0.0;
          L.childMask[nl] = r0.childMask;
          L.kidsMask[nl] = r0.kidsMask;
          L.info[nl] = mkInfo((uint32_t)endTok, hyp, kNoHyp);
          L.link[nl] = 0u;
          L.lmSid[nl] = sid;
          L.node[nl] = 0u;
          L.parent[nl] = 0u;
          L.firstChild[nl] = r0.firstChild;
          L.endLabel[nl] = r0.endLabel0;
          if constexpr (ML) {
            L.endExtra[nl] = S.rootExtra;
            L.endLmX[nl] = kYlNoLm;
            L.endLmX[kYlLanes + nl] = kYlNoLm;
          }
          L.dPar[nl] = xlm;
          L.dWord[nl] = el;
          L.maxScore[nl] = r0.maxScore;
          L.delta[nl] = 0.0f;
          L.endLm[nl] = kYlNoLm;
          L.endWord[nl] = S.rootWord;
          atomOr64(&S.alive[q][nl >> 6], 1ull << (nl & 63));
          histPT[hrow + hyp] = make_int2((int)hp, endTok);
          histW[hrow + hyp] = el;
          adopt(rootOrph[g], nl);
          (void)x;
        }
      }
    }
#ifndef FLTX_EMU
    if (wroteCtx) {
      __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory"); /* ... and has arrived before anybody can ask for it */
    }
#endif
    FLTX_YLPROF(5);
    ldsBarrier(); /* 3 */
    if (S.scal[YL_FLAG] != 0u) {
      dead = true; YL_WHY(6);
    }
    FLTX_YLPROF(6);
  };
  auto frames = [&](auto RL) {
    int t = 0;
    for (; t + 1 < T && !dead; t += 2) {
      frameStep(SlParity<0>(), RL, rowA, t);
      if (dead) {
        break;
      }
      frameStep(SlParity<1>(), RL, rowB, t + 1);
    }
    if (!dead && t < T) {
      frameStep(SlParity<0>(), RL, rowA, t);
    }
  };
  if (isSvc) {
    frames(SlParity<3>());
  } else if (isWordW) {
    frames(SlParity<2>());
  } else if (isSelfW) {
    frames(SlParity<1>());
  } else {
    frames(SlParity<0>());
  }

  /* ---- decodeEnd (LexiconDecoder.cpp:231-274): if any hypothesis stands on the root only those
   * finish; lm.finish (KenLM.cpp:88-104: the score of </s>) is added; the two hypotheses of a lane
   * merge; sorted n-best ------------------------------------------------------------------------ */
  const int pe = T & 1;
  const int ff = T + 1;
  if (!dead) {
    bool liveE = false, onRoot = false;
    double mE = NEG, lmE = 0.0, oE = NEG; /* (oE: logAdd -- the lane's other hypothesis finishes into the same group) */
    uint32_t hpE = kNoHyp, lmSidE = 0u;
    if (wave < NG) {
      const int x = wave * 64 + lane;
      liveE = ((S.alive[pe][wave] >> lane) & 1ull) != 0ull;
      const double xnb = liveE ? L.nb[x] : NEG, xb = liveE ? L.b[x] : NEG;
      const uint32_t xi = L.info[x];
      const bool wb = xb > xnb;
      mE = wb ? xb : xnb;
      oE = wb ? xnb : xb;
      hpE = wb ? infoB(xi) : infoNB(xi);
      lmE = LMT ? (wb ? L.lmB[x] : L.lmNB[x]) : 0.0;
      lmSidE = L.lmSid[x];
      onRoot = L.node[x] == 0u;
      if (liveE && onRoot) {
        S.scal[YL_NICE] = 1u;
      }
    }
    ldsBarrier();
    const bool nice = S.scal[YL_NICE] != 0u;
    if (wave < NG) {
      const int x = wave * 64 + lane;
      const bool cand = liveE && (!nice || onRoot);
      double sc = mE;
      if (ngram && cand) {
        const float fs = ylNgram(P, b, lmSidE, (uint32_t)P.lmEos, nullptr);
        ++nScored;
        sc = mE + lmWeight * (double)fs;
        oE = oE + lmWeight * (double)fs;
        lmE = lmE + (double)fs;
      }
      S.endKey[x] = (cand && sc == sc) ? f64Key(sc) : 0ull;
      S.endScore[x] = sc;
      S.endLmS[x] = lmE;
      S.endHyp[x] = hpE;
    }
    ldsBarrier();
    unsigned long long bk = 0ull;
    if (wave < NG) {
      for (int i = 0; i < 64 * NG; ++i) {
        const unsigned long long k2 = S.endKey[i];
        bk = k2 > bk ? k2 : bk;
      }
    }
    /* candidatesBestScore_ is the best of the candidates that finish */
    const double thr = f64FromKey(bk) - P.beamThreshold;
    if constexpr (LA) { /* both hypotheses of a lane above the threshold: their sum ranks and is reported */
      const int x = (wave < NG ? wave : 0) * 64 + lane;
      const bool okm = wave < NG && S.endKey[x] != 0ull && bk != 0ull && S.endScore[x] >= thr;
      const double merged = (okm && oE >= thr) ? slLogAdd(S.endScore[x], oE) : S.endScore[x];
      ldsBarrier(); /* (every lane has read the best members' keys) */
      if (wave < NG) {
        S.endScore[x] = merged;
        S.endKey[x] = okm ? f64Key(merged) : 0ull;
      }
      ldsBarrier();
    }
    if (wave < NG) {
      const int x = wave * 64 + lane;
      const unsigned long long key = S.endKey[x];
      const bool ok = key != 0ull && bk != 0ull && S.endScore[x] >= thr;
      int rank = 0, nOk = 0;
      for (int i = 0; i < 64 * NG; ++i) {
        const unsigned long long k2 = S.endKey[i];
        const bool ok2 = k2 != 0ull && S.endScore[i] >= thr;
        const uint32_t h2 = S.endHyp[i];
        rank += (ok2 && (k2 > key || (k2 == key && h2 < hpE))) ? 1 : 0;
        nOk += ok2 ? 1 : 0;
      }
      if (ok) {
        const size_t g = ((size_t)b * K + rank) * 3;
        P.outScores[g + 0] = S.endScore[x];
        P.outScores[g + 1] = 0.0; /* emitting-model score: the back-trace kernel fills it in */
        P.outScores[g + 2] = S.endLmS[x];
        histPT[hbase + (int64_t)ff * K + rank] = make_int2((int)hpE, sil);
        histW[hbase + (int64_t)ff * K + rank] = -1;
      }
      if (tid == 0) {
        P.outN[b] = nOk;
        P.uttNBeam[b] = nOk;
        P.uttFrame[b] = ff;
        P.uttTotal[b] = ff;
        P.uttStatus[b] = ST_PACKED;
      }
    }
  }
#ifdef FLTX_EMU
  if (getenv("FLTX_YL_WHY") && tid == 0) {
    fprintf(stderr, "ylane: utterance %d ends dead=%d with %u LM states\n", b, (int)dead, S.lmNext);
  }
#endif
  if (dead) { /* (uniform: every way out of the frame loop passes a barrier that spreads it) */
    if (deadWhy != 0 && lane == 0) {
      atomCas32(&S.scal[YL_WHYCODE], 0u, (uint32_t)deadWhy);
    }
    ldsBarrier();
    if (tid == 0) {
      P.outN[b] = 0;
      P.uttNBeam[b] = 0;
      P.uttFrame[b] = ff;
      P.uttTotal[b] = ff;
      P.uttStatus[b] = ST_SELECT_FALLBACK | (int32_t)((S.scal[YL_WHYCODE] & 31u) << 8);
    }
  }
  if (P.scored && nScored != 0u) {
    atomAdd32(&P.scored[b], nScored);
  }
  if (PROF && P.prof && tid == P.profThread) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      P.prof[(size_t)b * 8 + i] = acc[i];
    }
  }
  (void)li;
}
#undef FLTX_YLPROF
#undef YL_WHY
