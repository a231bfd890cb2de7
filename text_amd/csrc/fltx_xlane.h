/*
 * fltx_xlane.h -- "lane = (LM state, trie node)" decode of a whole utterance for the
 * lexicon decoder: LexiconDecoder + ZeroLM over a lexicon without LM scores (every
 * TrieNode::maxScore zero), CTC, max-merge, beam <= 64, <= 64 tokens, one word per
 * spelling, every word ending in one separator token, no <unk>, offline.  Included by
 * fltx_kernels.h after fltx_slane.h, whose organisation, histogram window, scan and
 * emission-row staging it shares.  Same candidates, same merge groups, same selection as
 * LexiconDecoder::decodeStep (LexiconDecoder.cpp:32-229) with candidatesStore
 * (Utils.h:146-225): bit-identical n-best.
 *
 * A lane holds the two hypotheses of one (lmState, lex) pair: `nb` = (.., token of the
 * node, prevBlank = false) and `b` = (.., blank, true), LexiconDecoder.h:79-91.  What the
 * reference does per hypothesis then has the same fixed shape as in fltx_slane.h:
 *   * "eat a new token" (:89-110) from a lane to the child node reached by token n comes
 *     from max(nb, b) -- from b alone when n is the node's own token (:89) -- and is a new
 *     lane, unless a lane holds that child already: then it joins the child lane's "stay"
 *     candidate (:168-194) through the child's link to its parent lane;
 *   * the child's id needs no gather: the trie is laid out breadth first
 *     (fltx_trie::xnode), child(n) = firstChild + popcount(childMask below n); the child's
 *     own 32-byte record is fetched once, when its lane is created;
 *   * blank (:197-213) and stay are the lane's own, and not subject to the token beam;
 *   * a word ends (:113-142) when the child reached by the separator token carries a
 *     label: the candidate lands on the root with a new LM state.  This is the one
 *     many-to-one merge of the lexicon decoder -- several lanes can emit the same word
 *     from the same LM state, and a lane may already stand on that root -- and it goes
 *     through a small LDS hash keyed by (LM state, word): one atomic max per candidate,
 *     resolved after the barrier that also publishes the frame's best candidate (which,
 *     unlike in the lexicon-free decoder, is not a recurrence: the trie decides what a
 *     lane can take);
 *   * an LM state is a number handed out when a word first ends from a given LM state; an
 *     LDS memo (LM state, word) -> number keeps it stable when the root lane that stood
 *     for it drops out of the beam and the word is emitted again (lm/LM.h:24-34);
 *   * a lane is the pair (LM state number, node id), so a lane that comes back after it
 *     dropped out is recognised by value.  What has to be restored is the link of the
 *     lanes in the beam that are its trie children: every lane without a parent lane
 *     enters (LM state, parent node) in a per-frame LDS table, and whoever creates a lane
 *     looks its pair up there and adopts them.
 * Waves: token waves (GT list positions each), one for the lanes' own groups (blank,
 * stay + parent's extension), one for the word ends and blank-then-own-token, one that
 * stages the emission rows.  Four barriers per frame.
 */
#pragma once

constexpr int kXlRoot = 128;  /* slots of the per-frame (LM state, word) merge table */
constexpr int kXlOrph = 128;  /* slots of the per-frame table of lanes without a parent lane */
constexpr int kXlMemo = 4096; /* slots of the LM-state memo */
constexpr int kXlMemoH = 8192; /* ... of its HBM form: (LM state + 1) << 40 | (word + 1) << 16 | number of the child state */
constexpr int kXlWarmBin = kSlMid + kSlMid / 4; /* candidates below this bin of the window (the last K-th sits at kSlMid) get their child node prefetched */

/* the lanes of one frame, one array per field (conflict-free LDS access, and a wave reads only
 * the fields its role needs) */
struct XlLanes {
  double nb[64], b[64];
  unsigned long long childMask[64], kidsMask[64]; /* XNode of the lane's node */
  unsigned long long cmask[64];                   /* tokens whose child node holds a lane that links here */
  uint32_t info[64];       /* own token | history slot of nb << 16 | of b << 24 */
  uint32_t link[64];       /* lane + 1 of the trie parent's lane, 0 = not in the beam (or root) */
  uint32_t lmSid[64];      /* LM state */
  uint32_t node[64];       /* trie node, breadth-first id, 0 = root */
  uint32_t parent[64];     /* its parent node */
  uint32_t firstChild[64];
  int32_t endLabel[64];    /* word that ends when the separator follows, -1 = none */
  uint32_t dPar[64];       /* root lanes: the LM state the word was emitted from ... */
  int32_t dWord[64];       /* ... and the word (-1: the start state) */
};

/* (LM state, word) -> best candidate landing there this frame; one array per field: the table is
 * wiped every frame with a few wide stores */
struct alignas(16) XlRootTab {
  unsigned long long key[kXlRoot];  /* 0 = free */
  unsigned long long best[kXlRoot]; /* order-preserving score key */
  uint32_t lane[kXlRoot];           /* lane + 1 of the root lane that already stands there, 0 = none */
  uint32_t minLane[kXlRoot];        /* lowest arriving lane among those that reach `best` */
  uint32_t winHyp[kXlRoot];         /* its history slot ... */
  int32_t winWord[kXlRoot];         /* ... and word (for the root lane's back-pointer) */
};
/* (LM state, node) -> the lanes whose parent that pair is and has no lane */
struct alignas(16) XlOrphTab {
  unsigned long long key[kXlOrph]; /* 0 = free */
  unsigned long long lanes[kXlOrph];
};

struct XlMemoSlot {
  unsigned long long key; /* 0 = free */
  uint32_t sid, pad;
};

struct XlaneLds {
  XlLanes L[2];
  uint32_t hist[2][kSlNB];
  double eAll[2][64];
  double eTok[2][kSlList];
  unsigned long long tokBit[2][kSlList];
  SlRow row[2];
  uint8_t tokId[2][kSlList];
  XlRootTab root;
  XlOrphTab orph[2];
  unsigned long long bestKey[2]; /* the frame's best candidate (all waves add their own) */
  XNode rootNode;
  uint32_t off[32];
  int32_t newLane[64];
  uint32_t scal[16];
  unsigned long long bKey[kSlBCap];
  uint32_t bOrd[kSlBCap];
  uint32_t memoUsed, lmNext;
  /* logAdd (LA): per slot of the merge table, the sum of exp(member - the slot's best member) over the members above
   * the frame's threshold (Utils.h:186-193 folds a merge group into max + log1p(exp(min - max)), member by member) */
  double rootAcc[kXlRoot];
  /* Last member: with more utterances than CUs (HM = 1) the memo lives in HBM (DecodeParams::ymemo, kXlMemoH
   * packed slots as in fltx_ylane.h) and the kernel is launched with offsetof(XlaneLds, memo) bytes of LDS --
   * 28 KB and 81 VGPRs: three workgroups of 512 threads share a CU. */
  XlMemoSlot memo[kXlMemo];
};

enum { XL_FLAG = 15 }; /* scal[]: a table ran full -> general path */

FLTX_DEV unsigned long long xlKey(uint32_t a, int32_t b) {
  return ((unsigned long long)(a + 1u) << 32) | (uint32_t)(b + 2);
}
FLTX_DEV uint32_t xlHash(unsigned long long k) {
  return hashKey((uint32_t)k, (uint32_t)(k >> 32), 0x9E3779B9u, 0x7F4A7C15u);
}
/* slot of `key` in the per-frame merge table (claimed if new); -1 when the table is full */
FLTX_DEV int xlRootFind(XlaneLds& S, unsigned long long key) {
  uint32_t h = xlHash(key) & (kXlRoot - 1);
  for (int probe = 0; probe < kXlRoot; ++probe) {
    const unsigned long long old = atomCas64(&S.root.key[h], 0ull, key);
    if (old == 0ull || old == key) {
      return (int)h;
    }
    h = (h + 1u) & (kXlRoot - 1);
  }
  return -1;
}
/* a lane without a parent lane announces itself under its parent's pair (<= 64 lanes, 128 slots) */
FLTX_DEV void xlOrphAdd(XlOrphTab& tab, unsigned long long key, int lane) {
  uint32_t h = xlHash(key) & (kXlOrph - 1);
  for (;;) {
    const unsigned long long old = atomCas64(&tab.key[h], 0ull, key);
    if (old == 0ull || old == key) {
      atomOr64(&tab.lanes[h], 1ull << lane);
      return;
    }
    h = (h + 1u) & (kXlOrph - 1);
  }
}
FLTX_DEV unsigned long long xlOrphGet(const XlOrphTab& tab, unsigned long long key) {
  uint32_t h = xlHash(key) & (kXlOrph - 1);
  for (;;) {
    const unsigned long long k = tab.key[h];
    if (k == key) {
      return tab.lanes[h];
    }
    if (k == 0ull) {
      return 0ull;
    }
    h = (h + 1u) & (kXlOrph - 1);
  }
}

/* (XlRank -- the token-beam ranking by row rotations -- lives in fltx_slane.h, which uses it too) */

#define FLTX_XLPROF(i)                                        \
  do {                                                        \
    if (PROF && P.prof && (int)threadIdx.x == P.profThread) { \
      const unsigned long long t_ = devClock();               \
      acc[(i)] += t_ - tPrev;                                 \
      tPrev = t_;                                             \
    }                                                         \
  } while (0)

/* GT = list positions per token wave (allowed tokens <= GT * (waves - 3)) */
/* LA: logAdd -- the members of a merge group that pass the frame's threshold are summed instead of maximised
 * (Utils.h:160-165 filters, :167-198 merges): the candidates and the frame's best are found as without it, the sums
 * are formed once the threshold is known (one more barrier for the groups that span lanes: a root's arrivals) */
template <int GT, int HM, bool PROF, bool LA = false>
FLTX_DEV void xlaneUtterance(const DecodeParams& P, char* smem) {
  XlaneLds& S = *(XlaneLds*)smem;
  const int b = P.uttMap ? P.uttMap[blockIdx.x] : (int)blockIdx.x;
  const int W = (int)blockDim.x, tid = (int)threadIdx.x;
  const int lane = laneId(), wave = waveUniform(waveId());
  const int nW = W >> 6;
  const int selfWave = nW - 3, wordWave = nW - 2, prepWave = nW - 1;
  const bool isSelfW = wave == selfWave, isWordW = wave == wordWave, isSvc = wave == prepWave; /* (roles at run time) */
#ifndef FLTX_EMU
  if ((P.tune & 1) && (isSelfW || isWordW || isSvc)) { /* measured, not the default here: priority for the waves the token waves wait for
                                                           (C3: 3.98 -> 4.04 ms; it pays on the other lane engines) */
    __builtin_amdgcn_s_setprio(3);
  }
#endif
  const int K = P.K, N = P.N;
  const int T = P.stepT ? P.stepT[b] : 0;
  const float* em = P.emissions ? P.emissions + P.emOff[b] : nullptr;
  const int64_t hbase = P.histOff[b];
  const double NEG = slNegInf();
  const int sil = P.sil, blank = P.blank;
  const int endTok = P.xEndTok; /* the word separator (== sil where this engine is selected) */
  const double silScore = P.silScore, wordScore = P.wordScore, beamThreshold = P.beamThreshold;
  int2* const histPT = P.histPT;
  int32_t* const histW = P.histW;
  const XNode* const xnode = P.xnode;
  unsigned long long acc[8] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull};
  unsigned long long tPrev = devClock();
  static_assert(GT >= 2, "the self and word waves keep their two groups in the slot arrays");

  /* ---- decodeBegin (LexiconDecoder.cpp:21-30): the start state at the root ------------- */
  for (int i = tid; i < 2 * 64; i += W) {
    ((unsigned long long*)S.L[0].cmask)[i & 63] = 0ull;
    ((unsigned long long*)S.L[1].cmask)[i & 63] = 0ull;
  }
  for (int i = tid; i < 2 * kSlNB; i += W) {
    ((uint32_t*)S.hist)[i] = 0u;
  }
  for (int i = tid; i < kXlRoot; i += W) {
    if constexpr (LA) {
      S.rootAcc[i] = 0.0;
    }
    S.root.key[i] = 0ull;
    S.root.best[i] = 0ull;
    S.root.lane[i] = 0u;
    S.root.minLane[i] = 0xFFFFFFFFu;
  }
  for (int i = tid; i < kXlOrph; i += W) {
    S.orph[0].key[i] = 0ull;
    S.orph[0].lanes[i] = 0ull;
    S.orph[1].key[i] = 0ull;
    S.orph[1].lanes[i] = 0ull;
  }
  unsigned long long* const gmemo = HM ? P.ymemo + (size_t)(P.uttMap ? P.uttMap[blockIdx.x] : (int)blockIdx.x) * kXlMemoH : nullptr;
  if (HM) {
    for (int i = tid; i < kXlMemoH; i += W) {
      gmemo[i] = 0ull; /* (at L2 before the barrier below, where the word wave's atomics will find it) */
    }
#ifndef FLTX_EMU
    __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  } else {
    for (int i = tid; i < kXlMemo; i += W) {
      S.memo[i].key = 0ull;
    }
  }
  if (tid < 32) {
    S.off[tid] = 0u;
  }
  if (tid < 16) {
    S.scal[tid] = 0u;
  }
  if (tid == 0) {
    const XNode r0 = xnode[0];
    S.rootNode = r0;
    XlLanes& L0 = S.L[0];
    L0.nb[0] = 0.0;
    L0.b[0] = NEG;
    L0.childMask[0] = r0.childMask;
    L0.kidsMask[0] = r0.kidsMask;
    L0.info[0] = (uint32_t)sil | (0u << 16) | (kSlNoHyp << 24);
    L0.link[0] = 0u;
    L0.lmSid[0] = 0u;
    L0.node[0] = 0u;
    L0.parent[0] = 0u;
    L0.firstChild[0] = r0.firstChild;
    L0.endLabel[0] = r0.endLabel0;
    L0.dPar[0] = 0x7FFFFFFFu;
    L0.dWord[0] = -1;
    S.row[0].nev = 0u;
    S.row[1].nev = 0u;
    S.row[0].dead = 0u;
    S.row[1].dead = 0u;
    S.bestKey[0] = 0ull;
    S.bestKey[1] = 0ull;
    S.memoUsed = 0u;
    S.lmNext = 1u;
    histPT[hbase] = make_int2((int)kSlNoHyp, sil);
    histW[hbase] = -1;
  }
  if (tid > 0 && tid < K) {
    histPT[hbase + tid] = make_int2((int)kSlNoHyp, -1);
  }
  float rowA = 0.0f, rowB = 0.0f;
  if (isSvc) {
    const float v0 = (T > 0 && lane < N) ? em[lane] : 0.0f;
    rowA = (T > 1 && lane < N) ? em[(size_t)1 * N + lane] : 0.0f;
    rowB = (T > 2 && lane < N) ? em[(size_t)2 * N + lane] : 0.0f;
    /* (the row's `best` is not used by this engine: ctc = false keeps blank in the token list,
     * which is harmless -- no trie node has a child for it) */
    SlRowRegs r0 = slRowScan(P, v0, false, 0.0);
    slRowStore(P, S, 0, r0, 2);
    slRowStore(P, S, 1, r0, 2);
  }
  ldsBarrier();

  int nState = 1;
  int winShift = kSlCoarseShift, winBase = kSlCoarseBase;
  bool dead = false;

  /* RL = role of the wave, compile time as the parity: the four kinds of waves (0 token, 1 own groups,
   * 2 word ends, 3 staging) share the barriers and the selection and nothing else -- each gets its own
   * straight-line frame and its own registers (fltx_slane.h: C2 kernel 2.37 -> 1.95 ms) */
  auto frameStep = [&](auto PT, auto RL, float& rowReg, const int t) {
    constexpr int p = decltype(PT)::value, q = p ^ 1;
    constexpr int role = decltype(RL)::value;
    constexpr bool isTok = role == 0, isSelf = role == 1, isWord = role == 2, isSvc = role == 3;
    const int frameOut = t + 1;
    const int64_t hrow = hbase + (int64_t)frameOut * K;
    XlLanes& Lp = S.L[p];
    XlLanes& Lq = S.L[q];
    /* ---- phase 1a: own lane, candidates, the merge and orphan tables, the frame's best ------ */
    const int silPos = S.row[p].silPos;
    const unsigned long long allow = S.row[p].allow;
    const bool live = lane < nState && !isSvc;
    const double nb = live ? Lp.nb[lane] : NEG, bb = live ? Lp.b[lane] : NEG;
    const uint32_t info = Lp.info[lane];
    const unsigned long long cm = Lp.cmask[lane];
    const unsigned long long childMask = Lp.childMask[lane], kidsMask = Lp.kidsMask[lane];
    const uint32_t lmSid = Lp.lmSid[lane], node = Lp.node[lane];
    const int32_t endLabel = Lp.endLabel[lane];
    double ev[GT];
    unsigned long long tb[GT];
#pragma unroll
    for (int j = 0; j < GT; ++j) {
      ev[j] = 0.0;
      tb[j] = 0ull;
    }
    if (isTok) {
#pragma unroll
      for (int j = 0; j < GT; ++j) {
        ev[j] = S.eTok[p][wave * GT + j];
        tb[j] = S.tokBit[p][wave * GT + j];
      }
    }
    const int last = (int)(info & 0xFFu) & 63;
    const uint32_t hypNB = (info >> 16) & 0xFFu, hypB = info >> 24;
    const bool hasNB = hypNB != kSlNoHyp, hasB = hypB != kSlNoHyp;
    const bool whichB = bb > nb;
    const double m = whichB ? bb : nb;
    const uint32_t hypM = whichB ? hypB : hypNB;
    const bool atRoot = node == 0u;
    const bool useB = atRoot && last == endTok; /* word end: the nb hypothesis on the root would repeat its token (:114-122) */
    const double eBlank = S.eAll[p][blank], eLast = S.eAll[p][last], eEnd = S.eAll[p][endTok], eSil = S.eAll[p][sil];
    int pl = -1;
    double parNB = NEG, parB = NEG;
    uint32_t parInfo = 0u;
    if (isSelf) {
      pl = live ? (int)Lp.link[lane] - 1 : -1;
      const int pi = pl >= 0 ? pl : 0;
      parNB = Lp.nb[pi];
      parB = Lp.b[pi];
      parInfo = Lp.info[pi];
    }
    FLTX_XLPROF(0);
    double cs[GT];
    int cbin[GT];
    bool cok[GT];
#pragma unroll
    for (int j = 0; j < GT; ++j) {
      cs[j] = NEG;
      cbin[j] = kSlInvalid;
      cok[j] = false;
    }
    uint32_t parR = kSlNoHyp;
    int rootSlot = -1;    /* self wave: the root lane's slot in the merge table; word wave: the arrival's */
    /* staging wave: the token beam of the next frame's row (LexiconDecoder.cpp:42-51: top beamSizeToken
     * by emission, ties to the lower index) is ranked in three pieces, in the gaps this wave has */
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
    /* (rows of <= 32 finite emissions: all pairs by row rotations; else one source token at a time) */
    const bool fastRank = needRank && N <= 32 && isSvc && waveBallot(lane < N && !(rv == rv)) == 0ull;
    XlRank rs = {};
    if (isSvc) {
      Lq.cmask[lane] = 0ull;
      if (lane < 32) {
        S.off[lane] = 0u;
      }
      if (lane == 0) {
        S.scal[SL_BCNT] = 0u;
        S.bestKey[q] = 0ull;
      }
      if (fastRank) {
        rs = xlRankBegin(rv, N);
        xlRankRange<0, 4>(rs);
      } else if (needRank) {
        rankPart(0, rk1);
      }
      rowReg = (t + 3 < T && lane < N) ? em[(size_t)(t + 3) * N + lane] : 0.0f;
    } else if (isTok) {
      /* "eat a new token" into a child that has children (LexiconDecoder.cpp:89-110): every listed
       * token but the node's own (that one needs the blank in between: the self wave) and those
       * whose child node holds a lane already (the child lane merges it into its stay) */
      const unsigned long long ext = childMask & kidsMask;
      const uint32_t lastLo = last < 32 ? 1u << last : 0u, lastHi = last < 32 ? 0u : 1u << (last - 32);
      const uint32_t skLo = live ? ((uint32_t)cm | lastLo | ~(uint32_t)ext) : 0xFFFFFFFFu;
      const uint32_t skHi = live ? ((uint32_t)(cm >> 32) | lastHi | ~(uint32_t)(ext >> 32)) : 0xFFFFFFFFu;
      const int silJ = silPos - wave * GT;
#pragma unroll
      for (int j = 0; j < GT; ++j) {
        double c = m + ev[j];
        if (j == silJ) {
          c = c + silScore;
        }
        const uint32_t hit = (skLo & (uint32_t)tb[j]) | (skHi & (uint32_t)(tb[j] >> 32));
        cs[j] = c;
        cok[j] = hit == 0u && tb[j] != 0ull && c == c;
      }
    } else if (isSelf) {
      /* blank (:197-213): always tried */
      double cB = m + eBlank;
      cs[0] = cB;
      cok[0] = live;
      /* stay (:168-194): the node's own token again from the nb hypothesis -- at the root: sil from
       * either hypothesis -- plus the trie parent's extension by that token.  The extension is a
       * "(1) try children" candidate and needs the token in the token beam; stay and blank do not
       * (LexiconDecoder.cpp:61,168,197) */
      const int lastP = (int)(parInfo & 0xFFu);
      const uint32_t h1 = (parInfo >> 16) & 0xFFu, h2 = parInfo >> 24;
      const bool allowLast = ((allow >> last) & 1ull) != 0ull;
      const bool has0 = atRoot ? true : hasNB;
      const bool has1 = pl >= 0 && allowLast && last != lastP && h1 != kSlNoHyp;
      const bool has2 = pl >= 0 && allowLast && h2 != kSlNoHyp;
      double r0 = (atRoot ? m : nb) + (atRoot ? eSil : eLast);
      double r1 = has1 ? parNB + eLast : NEG;
      double r2 = has2 ? parB + eLast : NEG;
      if (silScore != 0.0) {
        const bool ls = atRoot || last == sil;
        r0 = ls ? r0 + silScore : r0;
        r1 = ls ? r1 + silScore : r1;
        r2 = ls ? r2 + silScore : r2;
      }
      double cR = r0;
      parR = atRoot ? hypM : hypNB;
      if (has1 && (r1 > cR || (r1 == cR && h1 < parR))) {
        cR = r1;
        parR = h1;
      }
      if (has2 && (r2 > cR || (r2 == cR && h2 < parR))) {
        cR = r2;
        parR = h2;
      }
      cs[1] = cR;
      cok[1] = live && (has0 || has1 || has2);
      if (live && atRoot) { /* words ending here this frame join through the merge table */
        rootSlot = xlRootFind(S, xlKey(Lp.dPar[lane], Lp.dWord[lane]));
        if (rootSlot >= 0) {
          S.root.lane[rootSlot] = (uint32_t)lane + 1u;
          atomMax64(&S.root.best[rootSlot], f64Key(cR));
        }
      }
    } else if (isWord) {
      /* a word ends (:113-142): the child reached by the separator carries a label */
      const bool can = live && endLabel >= 0 && ((allow >> endTok) & 1ull) != 0ull;
      const double src = useB ? bb : m;
      const bool hasSrc = useB ? hasB : true;
      double c = src + eEnd;
      if (endTok == sil) {
        c = c + silScore;
      }
      c = (c + P.lmWeight * 0.0) + wordScore; /* ZeroLM, unsmeared lexicon: lmScore - lexMaxScore = 0 */
      cs[0] = c;
      cok[0] = can && hasSrc && c == c;
      if (cok[0]) {
        rootSlot = xlRootFind(S, xlKey(lmSid, endLabel));
        if (rootSlot >= 0) {
          atomMax64(&S.root.best[rootSlot], f64Key(c));
        }
      }
      /* blank, then the node's own token again (:89 with prevBlank): into the child, if it has children
       * and no lane */
      const bool extLast = ((childMask & kidsMask) >> last) & 1ull;
      const bool allowLast = ((allow >> last) & 1ull) != 0ull;
      double cL = bb + eLast;
      if (last == sil) {
        cL = cL + silScore;
      }
      cs[1] = cL;
      cok[1] = live && hasB && extLast && allowLast && ((cm >> last) & 1ull) == 0ull && cL == cL;
    }
    if (waveBallot(rootSlot < 0 && ((isSelf && live && atRoot) || (isWord && cok[0]))) != 0ull) {
      dead = true; /* merge table full: general path (uniform after the barrier below via bestKey = ~0) */
      if (lane == 0) {
        S.bestKey[p] = ~0ull;
      }
    }
    { /* the frame's best candidate (Utils.h:131-137) */
      unsigned long long mx = 0ull;
#pragma unroll
      for (int j = 0; j < GT; ++j) {
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
    FLTX_XLPROF(1);
    ldsBarrier(); /* A */
    /* ---- phase 1b: threshold, merge-table verdicts, histogram ---------------------------- */
    const unsigned long long bk = S.bestKey[p];
    if (bk == 0ull || bk == ~0ull) {
      dead = true;
      return;
    }
    const double best = f64FromKey(bk);
    if (!(best - best == 0.0)) {
      dead = true;
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
        const int silJ = silPos - wave * GT;
#pragma unroll
        for (int j = 0; j < GT; ++j) {
          if (cok[j]) {
            double a = nb + ev[j], c = bb + ev[j];
            if (j == silJ) {
              a = a + silScore;
              c = c + silScore;
            }
            cs[j] = la2(a, c);
          }
        }
      } else if (isSelf && live) {
        cs[0] = la2(nb + eBlank, bb + eBlank);
        if (atRoot) { /* stay on the root: sil from either hypothesis; the words ending here join through the slot */
          double a = nb + eSil, c = bb + eSil;
          a = a + silScore; /* (silScore 0: the same bits) */
          c = c + silScore;
          if (rootSlot >= 0) {
            laAdd(rootSlot, a);
            laAdd(rootSlot, c);
          }
        }
      } else if (isWord && cok[0]) {
        double a = (useB ? NEG : nb) + eEnd, c = bb + eEnd;
        if (endTok == sil) {
          a = a + silScore;
          c = c + silScore;
        }
        a = (a + P.lmWeight * 0.0) + wordScore;
        c = (c + P.lmWeight * 0.0) + wordScore;
        laAdd(rootSlot, a);
        laAdd(rootSlot, c);
      }
      ldsBarrier(); /* A2: the slots' sums are complete */
    }
    bool rootFromWord = false; /* self wave: a word ending on this root lane beats its own stay */
    if (isSelf && rootSlot >= 0) { /* the root lane's stay group takes the best word ending on it */
      const unsigned long long rb = S.root.best[rootSlot];
      if (rb > f64Key(cs[1])) {
        cs[1] = f64FromKey(rb);
        parR = kSlNoHyp; /* back-pointer: read from the slot in the build */
        rootFromWord = true;
      }
    }
    (void)rootFromWord;
    if constexpr (LA) {
      if (isSelf && live) {
        if (atRoot) {
          if (rootSlot >= 0) {
            const double mBest = f64FromKey(S.root.best[rootSlot]);
            cs[1] = mBest >= thr ? mBest + log1p(S.rootAcc[rootSlot] - 1.0) : NEG;
          }
        } else { /* stay + the trie parent's extension: up to three members, folded best first */
          const int lastP2 = (int)(parInfo & 0xFFu);
          const uint32_t g1 = (parInfo >> 16) & 0xFFu, g2 = parInfo >> 24;
          const bool allowLast2 = ((allow >> last) & 1ull) != 0ull;
          double x0 = hasNB ? nb + eLast : NEG;
          double x1 = (pl >= 0 && allowLast2 && last != lastP2 && g1 != kSlNoHyp) ? parNB + eLast : NEG;
          double x2 = (pl >= 0 && allowLast2 && g2 != kSlNoHyp) ? parB + eLast : NEG;
          if (silScore != 0.0 && last == sil) {
            x0 = x0 + silScore;
            x1 = x1 + silScore;
            x2 = x2 + silScore;
          }
          /* order: best first (Utils.h:168-174 sorts a group by score) */
          double hi = x0, mid = x1, lo = x2, tmp;
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
      /* of the arrivals that reach the slot's best the lowest lane represents them: it is the
       * back-pointer a root lane takes when a word ending on it beats its own stay, and it is the
       * candidate for a new root lane when nobody stands there */
      const bool top = cok[0] && S.root.best[rootSlot] == f64Key(cs[0]);
      if (top) {
        atomMin32(&S.root.minLane[rootSlot], (uint32_t)lane);
      }
      waveSync();
      const bool rep = top && S.root.minLane[rootSlot] == (uint32_t)lane;
      if (rep) {
        S.root.winHyp[rootSlot] = useB ? hypB : hypM;
        S.root.winWord[rootSlot] = endLabel;
      }
      cok[0] = rep && S.root.lane[rootSlot] == 0u;
      if (LA && cok[0]) {
        const double mBest = f64FromKey(S.root.best[rootSlot]);
        cs[0] = mBest >= thr ? mBest + log1p(S.rootAcc[rootSlot] - 1.0) : NEG;
      }
    }
    if (isSvc && fastRank) {
      xlRankRange<4, 10>(rs);
    } else if (isSvc && needRank) {
      rankPart(rk1, rk2);
    }
    if (isSelf && live && !atRoot && pl < 0) { /* no parent lane: whoever creates it this frame finds this lane here */
      xlOrphAdd(S.orph[p], xlKey(lmSid, (int32_t)Lp.parent[lane]), lane);
    }
#pragma unroll
    for (int j = 0; j < GT; ++j) {
      if (cok[j] && cs[j] >= thr) {
        cbin[j] = slBin<LA>(best, cs[j], winShift, winBase);
        if (cbin[j] < kSlFar) {
          atomAdd32(&S.hist[p][cbin[j]], 1u);
        }
      }
    }
    /* the record of a child node is read when its lane is created (phase 3); touching it now, for
     * the candidates that land in the part of the window where survivors are found, brings its
     * cache line into this CU's L1 while the selection runs */
    uint32_t warm = 0u;
    if (isTok || isWord) {
      const uint32_t fc = Lp.firstChild[lane];
#pragma unroll
      for (int j = 0; j < GT; ++j) {
        if (cbin[j] < kXlWarmBin && (isTok || j == 1)) {
          const int n = isTok ? __builtin_ctzll(tb[j] | (1ull << 63)) : last;
          const uint32_t child = fc + (uint32_t)popc64(childMask & ((1ull << n) - 1ull));
          warm += ((const volatile uint32_t*)(xnode + child))[2];
        }
      }
    }
    FLTX_XLPROF(2);
    ldsBarrier(); /* 1 */
    /* ---- phase 2: which candidates survive (as fltx_slane.h) ------------------------------ */
    unsigned long long selMask[GT];
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
      for (int j = 0; j < GT; ++j) {
        selMask[j] = waveBallot(cbin[j] <= lim);
      }
    } else {
      unsigned long long bLo = 0ull, bHi = 0x7FFFFFFFull;
      bool full = false;
      for (;;) {
        if (!full && !sc.crossed) {
          int nFar = 0;
#pragma unroll
          for (int j = 0; j < GT; ++j) {
            nFar += popc64(waveBallot(cbin[j] == kSlFar));
          }
          if (lane == 0 && nFar > 0) {
            atomAdd32(&S.hist[p][kSlFar], (uint32_t)nFar);
          }
          full = true;
          if (PROF) {
            acc[7] += 1000000ull; /* frames that had to count the far candidates */
          }
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
        if (sc.cnt <= kSlBCap) { /* the members of the K-th best's bin compare with each other */
          if (PROF) {
            acc[7] += 1ull; /* frames that rank the members of the K-th best's bin */
          }
#pragma unroll
          for (int j = 0; j < GT; ++j) {
            if (cbin[j] == sc.bstar) {
              const uint32_t i = atomAdd32(&S.scal[SL_BCNT], 1u);
              S.bKey[i] = f64Key(cs[j]);
              S.bOrd[i] = ((uint32_t)wave << 16) | ((uint32_t)j << 8) | (uint32_t)lane;
            }
          }
          ldsBarrier();
          take |= slRankBin<GT>(S.bKey, S.bOrd, sc.cnt, need, wave, [&](int j) { return cbin[j] == sc.bstar; },
                                [&](int j) { return f64Key(cs[j]); }); /* (broadcast + ballot: fltx_slane.h) */
          lim = sc.bstar - 1;
          break;
        }
        { /* too many in one bin: look again through the finest window that spans the bracket */
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
            dead = true;
            break;
          }
          if (PROF) {
            acc[7] += 1000ull; /* histogram passes over a narrowed bracket */
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
        for (int j = 0; j < GT; ++j) {
          if (cbin[j] != kSlInvalid) {
            cbin[j] = slBin<LA>(best, cs[j], shift, base);
            atomAdd32(&S.hist[p][cbin[j]], 1u);
          }
        }
        ldsBarrier();
        sc = slScan(S.hist[p], K, false);
      }
#pragma unroll
      for (int j = 0; j < GT; ++j) {
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
    FLTX_XLPROF(3);
    /* new lanes: survivors first (self wave), then new trie lanes wave by wave, then new roots */
    uint32_t pend = 0u;         /* list positions of this lane that become new lanes */
    int planN = 0;
    uint32_t planNode = 0u;     /* the child the next new lane of this lane stands on, its record ... */
    XNode planX = {};
    unsigned long long planOrph = 0ull; /* ... and the lanes in the beam whose parent pair it is */
    uint32_t rootSid = 0u;
    unsigned long long rootOrph = 0ull;
    auto planChild = [&](int n) {
      planN = n;
      planNode = Lp.firstChild[lane] + (uint32_t)popc64(childMask & ((1ull << n) - 1ull));
      planX = xnode[planNode];
      planOrph = xlOrphGet(S.orph[p], xlKey(lmSid, (int32_t)planNode));
    };
    int nNewWave = 0;
    int myNew[GT];
    int surv = -1;
    uint32_t hNB = kSlNoHyp, hB = kSlNoHyp;
#pragma unroll
    for (int j = 0; j < GT; ++j) {
      myNew[j] = 0;
    }
    if (isTok || isWord) {
#pragma unroll
      for (int j = 0; j < GT; ++j) {
        if (selMask[j] != 0ull) {
          myNew[j] = nNewWave + wavePrefixCount(selMask[j]);
          nNewWave += popc64(selMask[j]);
        }
      }
      /* order of the new lanes: token waves, then the word wave (new roots, blank-then-own-token lanes) */
      const int slot = isTok ? wave : selfWave + 1;
      if (lane > slot && lane <= selfWave + 2 && nNewWave > 0) {
        atomAdd32(&S.off[lane], (uint32_t)nNewWave);
      }
      /* what a new lane needs and is known already: the child's record (one 32-byte gather from HBM,
       * in flight across the barrier), the lanes it adopts, the LM state of a new root */
#pragma unroll
      for (int j = 0; j < GT; ++j) {
        pend |= (isTok || j == 1) ? (uint32_t)((selMask[j] >> lane) & 1ull) << j : 0u;
      }
      if (pend) {
        const int j0 = __builtin_ctz(pend);
        unsigned long long tbj = tb[0];
#pragma unroll
        for (int j = 1; j < GT; ++j) {
          tbj = j == j0 ? tb[j] : tbj;
        }
        planChild(isTok ? __builtin_ctzll(tbj | (1ull << 63)) : last);
      }
      if (isWord && ((selMask[0] >> lane) & 1ull)) {
        /* the LM state's number (memo: the same (LM state, word) gives the same state back) */
        if (HM) {
          const unsigned long long mkey = ((unsigned long long)(lmSid + 1u) << 24) | (unsigned long long)(uint32_t)(endLabel + 1);
          uint32_t h = xlHash(mkey) & (kXlMemoH - 1);
          bool have = false; /* a number was taken from the counter for this state */
          for (int probe = 0;; ++probe) {
            unsigned long long cur = loadCoherent64(&gmemo[h]);
            if (cur == 0ull) {
              if (!have) {
                rootSid = atomAdd32(&S.lmNext, 1u);
                have = true;
              }
              cur = atomCas64(&gmemo[h], 0ull, (mkey << 16) | (unsigned long long)(rootSid & 0xFFFFu));
              if (cur == 0ull) { /* a new LM state */
                if (rootSid > (uint32_t)(kXlMemoH * 3 / 4) || rootSid >= 0xFFFFu || (uint32_t)(endLabel + 1) >= (1u << 24)) {
                  atomOr32(&S.scal[XL_FLAG], 1u);
                }
                break;
              }
            }
            if ((cur >> 16) == mkey) { /* (a number taken in vain stays unused) */
              rootSid = (uint32_t)(cur & 0xFFFFull);
              break;
            }
            h = (h + 1u) & (kXlMemoH - 1);
            if (probe > kXlMemoH) {
              atomOr32(&S.scal[XL_FLAG], 1u);
              break;
            }
          }
        } else {
        const unsigned long long mkey = xlKey(lmSid, endLabel);
        uint32_t h = xlHash(mkey) & (kXlMemo - 1);
        for (int probe = 0;; ++probe) {
          const unsigned long long old = atomCas64(&S.memo[h].key, 0ull, mkey);
          if (old == 0ull) {
            rootSid = atomAdd32(&S.lmNext, 1u);
            S.memo[h].sid = rootSid;
            if (atomAdd32(&S.memoUsed, 1u) > (uint32_t)(kXlMemo * 3 / 4)) {
              atomOr32(&S.scal[XL_FLAG], 1u); /* memo nearly full: general path from the next frame on */
            }
            break;
          }
          if (old == mkey) {
            rootSid = S.memo[h].sid;
            break;
          }
          h = (h + 1u) & (kXlMemo - 1);
          if (probe > kXlMemo) {
            atomOr32(&S.scal[XL_FLAG], 1u);
            break;
          }
        }
        }
        rootOrph = xlOrphGet(S.orph[p], xlKey(rootSid, 0));
      }
    } else if (isSelf) {
      const unsigned long long balB = selMask[0], balR = selMask[1];
      const unsigned long long balS = balB | balR;
      const bool sR = ((balR >> lane) & 1ull) != 0ull;
      surv = ((balS >> lane) & 1ull) ? wavePrefixCount(balS) : -1;
      hNB = (uint32_t)(wavePrefixCount(balR) + wavePrefixCount(balB));
      hB = hNB + (sR ? 1u : 0u);
      S.newLane[lane] = surv;
      if (lane == 0) {
        S.scal[SL_NSURV] = (uint32_t)popc64(balS);
        S.scal[SL_NHSURV] = (uint32_t)(popc64(balR) + popc64(balB));
      }
      /* the survivors' links, here and not in the build: the builders overwrite the link of the
       * lanes they adopt */
      waveSync();
      const int pln = pl >= 0 ? S.newLane[pl] : -1;
      if (surv >= 0) {
        Lq.link[surv] = (uint32_t)(pln + 1);
      }
      pl = pln; /* from here on: the parent's lane in the next frame */
    } else if (isSvc && t + 1 < T) {
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
    }
    FLTX_XLPROF(4);
    ldsBarrier(); /* 2 */
    /* ---- phase 3: every survivor is written by the lane that evaluated it ------------------ */
    const int nSurv = (int)S.scal[SL_NSURV], nHSurv = (int)S.scal[SL_NHSURV];
    const int offW = (int)S.off[isTok ? wave : (isSelf ? selfWave : selfWave + 1)], nNew = (int)S.off[selfWave + 2];
    const int myNewLane = S.newLane[lane];
    /* the lanes in the beam whose parent pair (lm, nd) is: they link to the new lane nl */
    auto adopt = [&](unsigned long long o, int nl) {
      unsigned long long toks = 0ull;
      while (o) {
        const int x = __builtin_ctzll(o);
        o &= o - 1ull;
        const int nx = S.newLane[x];
        if (nx >= 0) {
          Lq.link[nx] = (uint32_t)nl + 1u;
          toks |= 1ull << (Lp.info[x] & 63u);
        }
      }
      if (toks) {
        atomOr64(&Lq.cmask[nl], toks);
      }
    };
    /* a new lane on the planned child */
    auto newChild = [&](int idx, double c, uint32_t hp) {
      const int nl = nSurv + idx;
      const uint32_t hyp = (uint32_t)(nHSurv + idx);
      const int n = planN;
      const uint32_t child = planNode;
      const XNode cx = planX;
      Lq.nb[nl] = c;
      Lq.b[nl] = NEG;
      Lq.childMask[nl] = cx.childMask;
      Lq.kidsMask[nl] = cx.kidsMask;
      Lq.info[nl] = (uint32_t)n | (hyp << 16) | (kSlNoHyp << 24);
      Lq.link[nl] = (uint32_t)(myNewLane + 1);
      Lq.lmSid[nl] = lmSid;
      Lq.node[nl] = child;
      Lq.parent[nl] = node;
      Lq.firstChild[nl] = cx.firstChild;
      Lq.endLabel[nl] = cx.endLabel0;
      Lq.dPar[nl] = 0u;
      Lq.dWord[nl] = -1;
      if (myNewLane >= 0) {
        atomOr64(&Lq.cmask[myNewLane], 1ull << n);
      }
      histPT[hrow + hyp] = make_int2((int)hp, n);
      histW[hrow + hyp] = -1;
      adopt(planOrph, nl);
    };
    if (isSvc) {
      /* (no stores to HBM from this wave: a wait for its emission-row load would wait for them too) */
      if (t + 1 < T) {
        slRowStore(P, S, q, nextRow, P.Kt < N ? 1 : 0);
      }
      ((uint4*)S.hist[q])[lane] = make_uint4(0u, 0u, 0u, 0u);
      /* the merge table of the next frame starts empty (winHyp / winWord, which the self wave reads
       * now, stay), and so does its orphan table: 16 bytes per lane and store */
      static_assert(kXlRoot == 128 && kXlOrph == 128, "the wipes below cover 128 slots");
      const uint4 z4 = make_uint4(0u, 0u, 0u, 0u), f4 = make_uint4(~0u, ~0u, ~0u, ~0u);
      ((uint4*)S.root.key)[lane] = z4;
      ((uint4*)S.root.best)[lane] = z4;
      if (lane < 32) {
        ((uint4*)S.root.lane)[lane] = z4;
      } else {
        ((uint4*)S.root.minLane)[lane - 32] = f4;
      }
      ((uint4*)S.orph[q].key)[lane] = z4;
      ((uint4*)S.orph[q].lanes)[lane] = z4;
      if constexpr (LA) {
        ((uint4*)S.rootAcc)[lane] = z4;
      }
    } else if (isTok) {
      /* most lanes create at most one: every round takes each lane's lowest pending position */
      bool first = true;
      while (waveBallot(pend != 0u) != 0ull) {
        if (pend) {
          const int j0 = __builtin_ctz(pend);
          pend &= pend - 1u;
          double c = cs[0];
          int mn = myNew[0];
          unsigned long long tbj = tb[0];
#pragma unroll
          for (int j = 1; j < GT; ++j) {
            c = j == j0 ? cs[j] : c;
            mn = j == j0 ? myNew[j] : mn;
            tbj = j == j0 ? tb[j] : tbj;
          }
          if (!first) {
            planChild(__builtin_ctzll(tbj | (1ull << 63)));
          }
          newChild(offW + mn, c, hypM);
        }
        first = false;
      }
    } else if (isSelf) {
      if (lane >= nHSurv + nNew && lane < K) { /* unused slots of the history row */
        histPT[hrow + lane] = make_int2((int)kSlNoHyp, -1);
      }
      if (surv >= 0) {
        const bool sB = ((selMask[0] >> lane) & 1ull) != 0ull, sR = ((selMask[1] >> lane) & 1ull) != 0ull;
        Lq.nb[surv] = sR ? cs[1] : NEG;
        Lq.b[surv] = sB ? cs[0] : NEG;
        Lq.childMask[surv] = childMask;
        Lq.kidsMask[surv] = kidsMask;
        Lq.info[surv] = (uint32_t)last | ((sR ? hNB : kSlNoHyp) << 16) | ((sB ? hB : kSlNoHyp) << 24);
        Lq.lmSid[surv] = lmSid;
        Lq.node[surv] = node;
        Lq.parent[surv] = Lp.parent[lane];
        Lq.firstChild[surv] = Lp.firstChild[lane];
        Lq.endLabel[surv] = endLabel;
        Lq.dPar[surv] = Lp.dPar[lane];
        Lq.dWord[surv] = Lp.dWord[lane];
        if (pl >= 0) {
          atomOr64(&Lq.cmask[pl], 1ull << last);
        }
        if (sR) {
          uint32_t hp = parR;
          int32_t wd = -1;
          if (hp == kSlNoHyp) { /* a word ending on this root beat its own stay */
            hp = S.root.winHyp[rootSlot];
            wd = S.root.winWord[rootSlot];
          }
          histPT[hrow + hNB] = make_int2((int)hp, atRoot ? sil : last);
          histW[hrow + hNB] = wd;
        }
        if (sB) {
          histPT[hrow + hB] = make_int2((int)hypM, blank);
          histW[hrow + hB] = -1;
        }
      }
    } else if (isWord) {
      if ((selMask[1] >> lane) & 1ull) {
        newChild(offW + myNew[1], cs[1], hypB);
      }
      if ((selMask[0] >> lane) & 1ull) { /* a word ended and nobody stood on that root: a new root lane */
        const int nl = nSurv + offW + myNew[0];
        const uint32_t hyp = (uint32_t)(nHSurv + offW + myNew[0]);
        const uint32_t hp = useB ? hypB : hypM;
        const uint32_t sid = rootSid;
        const XNode r0 = S.rootNode;
        Lq.nb[nl] = cs[0];
        Lq.b[nl] = NEG;
        Lq.childMask[nl] = r0.childMask;
        Lq.kidsMask[nl] = r0.kidsMask;
        Lq.info[nl] = (uint32_t)endTok | (hyp << 16) | (kSlNoHyp << 24);
        Lq.link[nl] = 0u;
        Lq.lmSid[nl] = sid;
        Lq.node[nl] = 0u;
        Lq.parent[nl] = 0u;
        Lq.firstChild[nl] = r0.firstChild;
        Lq.endLabel[nl] = r0.endLabel0;
        Lq.dPar[nl] = lmSid;
        Lq.dWord[nl] = endLabel;
        histPT[hrow + hyp] = make_int2((int)hp, endTok);
        histW[hrow + hyp] = endLabel;
        adopt(rootOrph, nl);
      }
    }
    nState = nSurv + nNew;
#ifndef FLTX_EMU
    __asm__ volatile("" ::"v"(warm));
#else
    (void)warm;
#endif
    FLTX_XLPROF(5);
    ldsBarrier(); /* 3 */
    if (S.scal[XL_FLAG] != 0u) {
      dead = true;
    }
    FLTX_XLPROF(6);
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
   * finish; finish() keeps the LM state (ZeroLM), token = sil; the two hypotheses of a lane merge;
   * sorted n-best ---------------------------------------------------------------------------- */
  const int pe = T & 1;
  const int ff = T + 1;
  if (wave == 0 && !dead) {
    const bool live = lane < nState;
    const XlLanes& Le = S.L[pe];
    const int li = live ? lane : 0;
    const double nb = live ? Le.nb[li] : NEG, bb = live ? Le.b[li] : NEG;
    const uint32_t info = Le.info[li];
    const bool onRoot = Le.node[li] == 0u;
    const bool whichB = bb > nb;
    const double m = whichB ? bb : nb;
    const uint32_t hp = whichB ? (info >> 24) : ((info >> 16) & 0xFFu);
    const bool nice = waveBallot(live && onRoot) != 0ull;
    const bool cand = live && (!nice || onRoot);
    /* candidatesBestScore_ is the best of the candidates that finish */
    const unsigned long long bk = waveMax64(cand ? f64Key(m) : 0ull);
    const double thr = f64FromKey(bk) - P.beamThreshold;
    const bool ok = cand && bk != 0ull && m >= thr;
    double mOut = m;
    if (LA && ok) { /* the lane's two hypotheses finish into one group: both above the threshold -> their sum */
      const double lo = whichB ? nb : bb;
      mOut = lo >= thr ? slLogAdd(m, lo) : m;
    }
    const unsigned long long key = ok ? f64Key(mOut) : 0ull;
    int rank = 0;
    for (int i = 0; i < nState; ++i) {
      const uint32_t lo = waveReadLane32((uint32_t)key, i), hi = waveReadLane32((uint32_t)(key >> 32), i);
      const unsigned long long k2 = ((unsigned long long)hi << 32) | lo;
      const uint32_t h2 = waveReadLane32(hp, i);
      rank += (k2 > key || (k2 == key && h2 < hp)) ? 1 : 0;
    }
    const unsigned long long okMask = waveBallot(ok);
    if (ok) {
      const size_t g = ((size_t)b * K + rank) * 3;
      P.outScores[g + 0] = mOut;
      P.outScores[g + 1] = 0.0; /* emitting-model score: the back-trace kernel fills it in */
      P.outScores[g + 2] = 0.0; /* ZeroLM over an unsmeared lexicon: every lmScore term is 0 */
      histPT[hbase + (int64_t)ff * K + rank] = make_int2((int)hp, sil);
      histW[hbase + (int64_t)ff * K + rank] = -1;
    }
    if (lane == 0) {
      P.outN[b] = popc64(okMask);
      P.uttNBeam[b] = popc64(okMask);
      P.uttFrame[b] = ff;
      P.uttTotal[b] = ff;
      P.uttStatus[b] = ST_PACKED;
    }
  }
  if (dead && tid == 0) {
    P.outN[b] = 0;
    P.uttNBeam[b] = 0;
    P.uttFrame[b] = ff;
    P.uttTotal[b] = ff;
    P.uttStatus[b] = ST_SELECT_FALLBACK;
  }
  if (PROF && P.prof && tid == P.profThread) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      P.prof[(size_t)b * 8 + i] = acc[i];
    }
  }
}
#undef FLTX_XLPROF
