/*
 * fltx_mlane.h -- the "lane = LM state" decode of fltx_slane.h with NG groups of 64 lanes: offline
 * LexiconFreeDecoder + ZeroLM (or, TL, a token-level n-gram LM), max-merge or logAdd, beams 65 .. 64 * NG, <= 64 tokens.  Included by
 * fltx_kernels.h after fltx_slane.h, whose row staging (slRowScan / slRowStore), histogram scan (slScan) and
 * binning (slBin) it shares.  Same candidates, same merge groups, same selection as
 * LexiconFreeDecoder::decodeStep (LexiconFreeDecoder.cpp:30-125) with candidatesStore (Utils.h:146-225):
 * bit-identical n-best.
 *
 * What a beam beyond one wave's lanes changes:
 *   * A wave holds the states of whole lane groups in registers: token wave w evaluates the GT list positions of
 *     its block for GPW groups (GT * GPW candidates per lane); a self wave owns the blank / repeat /
 *     blank-then-last groups of SPW lane groups.  Roles are compile-time as in fltx_slane.h.
 *   * Survivors are compacted group by group: a self wave publishes its groups' counts before barrier 2 and every
 *     wave adds up the groups in front of the one it addresses after it (NG <= 8 words, one LDS read).
 *   * Records carry 10-bit history slots and lane numbers; a history record is
 *       x = parent slot (10 bit, 0x3FF none) | entered-a-new-state (1 bit) | parent state id, low 21 bit
 *       y = token (8 bit) | parent state id, high bits << 8
 *     (the back-trace masks the slot field: BacktraceParams::packed = 10).
 *   * decodeEnd ranks up to 64 * NG states through LDS instead of lane broadcasts.
 *
 * Three barriers per frame, as in fltx_slane.h.
 *
 * TL (round 6): a token-level n-gram LM (LexiconFreeDecoder.cpp:69-85 with KenLM::score, lm/KenLM.cpp:63-83) through the
 * dense (context, token) table of fltx_slane.h's token-LM variant -- a state's table row and the LM score it was entered
 * with travel with its lane, every candidate adds lmWeight x one gathered score, the frame's best is a maximum behind
 * one more barrier, decodeEnd adds lmWeight x finish.  LMState::child's memo is a (parent id, token) -> id table in HBM
 * (mlChildId: the compare-and-swap is the look-up); a state entered again takes its children in the beam back
 * (mlRelink); a wave's new states are built in one gathered pass so that their table accesses overlap.
 */
#pragma once

constexpr uint32_t kMlNoHyp = 0x3FFu;   /* no history slot / no parent slot */
constexpr uint32_t kMlNewFlag = 0x400u; /* history record: the hypothesis entered a new LM state */
constexpr int kMlSidShift = 11;         /* x = slot | flag | (sid & 0x1FFFFF) << 11 */
constexpr int kMlSidLowBits = 21;
constexpr int kMlMaxGroups = 8;
constexpr int kMlGather = 32; /* TL: new states a wave builds in one gathered pass (more: position by position) */

struct MlRec { /* one LM state of the beam, 32 B */
  double nb;     /* score of (S, last token, prevBlank = false) */
  double b;      /* score of (S, blank, prevBlank = true) */
  uint32_t info; /* last token | (parent lane + 1) << 8 */
  uint32_t sid;  /* state id = history index (row * K + slot) of the hypothesis that entered it */
  uint32_t spar; /* id of the parent state */
  uint32_t hyps; /* history slot of nb | of b << 16 (kMlNoHyp = absent) */
};

template <int NG>
struct MlaneLds {
  static constexpr int kLanes = 64 * NG;
  MlRec rec[2][kLanes];
  unsigned long long cmask[2][kLanes]; /* tokens whose child state is in the beam and linked to this lane */
  unsigned long long mask[2][kLanes];  /* tokens whose child state was ever materialised */
  uint32_t hist[2][kSlNB];
  double eAll[2][64];
  double eTok[2][kSlList];
  unsigned long long tokBit[2][kSlList];
  SlRow row[2];
  uint8_t tokId[2][kSlList];
  uint32_t off[32];         /* new states of the waves before wave i; [last wave] = all */
  int32_t newLane[kLanes];  /* old lane -> index among its group's survivors, -1 = dropped */
  uint32_t grp[kMlMaxGroups]; /* per lane group: surviving states | surviving hypotheses << 16 */
  uint32_t scal[16];
  unsigned long long bKey[kSlBCap];
  uint32_t bOrd[kSlBCap];
  uint32_t evLane[kLanes], evSpar[kLanes], evTok[kLanes];
  unsigned long long scanMask;
  unsigned long long mmaxKey[2];
  uint32_t scanMin, pad0;
  float raw[3][64];
  /* token-LM variant (TL): a state's row of the dense (context, token) table, the LM score it was entered with, the
   * frame's best candidate (a maximum again), the ids of re-entered states, the utterance's state counter */
  unsigned long long fbest[2];
  uint32_t ctx[2][kLanes];
  float tlIn[2][kLanes];
  uint32_t evSid[kLanes];
  uint32_t idNext, pad1;
  alignas(16) uint32_t tlNew[15][kMlGather][8]; /* per wave: what its survivors' new states are made from (32 B each) */
};
enum { ML_BCNT = 2, ML_NOUT = 6, ML_FULL = 7 };

/* TL: (parent state id, token) -> id of the child state, the memo behind LMState::child for the token-LM variant.  With
 * LM terms the beam turns over fast and states are entered again every few frames; finding them in the history rows
 * (mlReenter) is a scan per event.  So the ids live in a table in HBM (DecodeParams::ymemo, ymemoSlots slots per
 * utterance, wiped by the kernel): one 64-bit word per slot = (parent + 1):28 << 36 | token:8 << 28 | id:28, key and id
 * installed together by one compare-and-swap.  The decoder asks about a (state, token) pair at most once per frame, so
 * nobody waits for anybody.  `fresh` = false says "entered before": its children in the beam get it back as their
 * parent lane (mlRelink).  Returns 0xFFFFFFFF when the table is full (general path). */
FLTX_DEV uint32_t mlChildId(unsigned long long* tab, uint32_t slots, uint32_t par, uint32_t tok, uint32_t* nextId, bool& fresh) {
  const unsigned long long key = ((unsigned long long)(par + 1u) << 36) | ((unsigned long long)(tok & 0xFFu) << 28);
  const uint32_t mask = slots - 1u;
  uint32_t h = hashKey(par, tok, 0x9747b28cu, 0x85EBCA6Bu) & mask;
  /* (most states are new: the name is drawn first and the compare-and-swap IS the look-up -- one round trip to the L2
   * for a new state, one for a known one; a name drawn for a state that turns out to be known is a name nobody bears) */
  const uint32_t id = atomAdd32(nextId, 1u);
  fresh = false;
  if (id >= (1u << 28)) {
    return 0xFFFFFFFFu;
  }
  for (uint32_t probes = 0; probes < slots; ++probes) {
    const unsigned long long cur = atomCas64(&tab[h], 0ull, key | (unsigned long long)id);
    if (cur == 0ull) {
      fresh = true;
      return id;
    }
    if ((cur >> 28) == (key >> 28)) {
      return (uint32_t)cur & 0x0FFFFFFFu;
    }
    h = (h + 1u) & mask;
  }
  return 0xFFFFFFFFu;
}

/* TL: the states entered again in the last build (evLane / evSid) take their children in the beam back: a lane whose
 * parent state is one of them links to its lane.  All waves; two barriers. */
template <typename LDS>
FLTX_DEV void mlRelink(LDS& S, int q, int nState) {
  const int tid = (int)threadIdx.x, W = (int)blockDim.x;
  const int nev = (int)S.row[q].nev;
  for (int l = tid; l < nState; l += W) {
    const uint32_t sp = S.rec[q][l].spar;
    for (int e = 0; e < nev; ++e) {
      const int X = (int)S.evLane[e];
      if (S.evSid[e] == sp && X != l) {
        const uint32_t info = S.rec[q][l].info;
        S.rec[q][l].info = (info & 0xFFu) | ((uint32_t)(X + 1) << 8);
        atomOr64(&S.cmask[q][X], 1ull << (info & 63u));
      }
    }
  }
  ldsBarrier();
  if (tid == 0) {
    S.row[q].nev = 0u;
  }
  ldsBarrier();
}

FLTX_DEV uint32_t mlParentSid(uint32_t x, uint32_t y) { return (x >> kMlSidShift) | ((y >> 8) << kMlSidLowBits); }
FLTX_DEV int2 mlNewRec(uint32_t hp, uint32_t parSid, int n) {
  return make_int2((int)(hp | kMlNewFlag | (parSid << kMlSidShift)), (int)((uint32_t)n | ((parSid >> kMlSidLowBits) << 8)));
}

/* Re-entry of LM states that had dropped out of the beam: fltx_slane.h's slReenter over the wider records.
 * All waves; rare. */
/* (inlined: the kernels with many candidates per lane spill registers, and a call from there would need a stack
 * frame beyond the spill area) */
template <typename LDS>
FLTX_DEV void mlReenter(LDS& S, const int2* histPT, int q, int nState, int64_t hbase,
                                                   int64_t nRec) {
  const int tid = (int)threadIdx.x, W = (int)blockDim.x;
  const int nev = (int)S.row[q].nev;
#ifndef FLTX_EMU
  __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory"); /* this wave's history stores have reached the L2 */
#endif
  ldsBarrier();
  for (int e = 0; e < nev; ++e) {
    const int X = (int)S.evLane[e];
    const uint32_t ps = S.evSpar[e], n = S.evTok[e];
    if (tid == 0) {
      S.scanMin = 0xFFFFFFFFu;
      S.scanMask = 0ull;
    }
    ldsBarrier();
    const unsigned long long* h = (const unsigned long long*)(histPT + hbase);
    uint32_t found = 0xFFFFFFFFu;
    for (int64_t i = tid; i < nRec; i += W) {
      const unsigned long long r = loadCoherent64(h + i);
      const uint32_t x = (uint32_t)r, y = (uint32_t)(r >> 32);
      if ((x & kMlNewFlag) && (r >> 63) == 0ull && (y & 0xFFu) == n && mlParentSid(x, y) == ps) {
        found = found < (uint32_t)i ? found : (uint32_t)i;
      }
    }
    if (found != 0xFFFFFFFFu) {
      atomMin32(&S.scanMin, found);
    }
    ldsBarrier();
    const uint32_t sid = S.scanMin;
    if (sid != 0xFFFFFFFFu) {
      unsigned long long kids = 0ull;
      for (int64_t i = tid; i < nRec; i += W) {
        const unsigned long long r = loadCoherent64(h + i);
        const uint32_t x = (uint32_t)r, y = (uint32_t)(r >> 32);
        if ((x & kMlNewFlag) && (r >> 63) == 0ull && mlParentSid(x, y) == sid) {
          kids |= 1ull << (y & 63u);
        }
      }
      if (kids) {
        atomOr64(&S.scanMask, kids);
      }
      ldsBarrier();
      if (tid == 0) {
        S.rec[q][X].sid = sid;
        S.mask[q][X] |= S.scanMask;
      }
      for (int l = tid; l < nState; l += W) { /* orphans get their parent back */
        if (l != X && S.rec[q][l].spar == sid) {
          const uint32_t info = S.rec[q][l].info;
          S.rec[q][l].info = (info & 0xFFu) | ((uint32_t)(X + 1) << 8);
          atomOr64(&S.cmask[q][X], 1ull << (info & 63u));
        }
      }
    }
    ldsBarrier();
  }
  if (tid == 0) {
    S.row[q].nev = 0u;
  }
  ldsBarrier();
}

/* GT = list positions per token wave and group, NG = lane groups, GPW = groups a token wave evaluates, SPW = groups
 * a self wave owns, LA = logAdd.  Waves: (NG / GPW) x nBlk token waves (block = wave % nBlk, group set = wave / nBlk),
 * NG / SPW self waves, one wave that stages the emission rows. */
/* TL: a token-level n-gram LM through its dense (context, token) table (DecodeParams::tokLm, fltx_slane.h's token-LM
 * variant says what that changes: the LM term is one gather per candidate, the frame's best a maximum behind one more
 * barrier, decodeEnd adds lmWeight x finish) */
template <int GT, int NG, int GPW, int SPW, bool LA, bool TL = false>
FLTX_DEV void mlaneUtterance(const DecodeParams& P, char* smem) {
  static_assert(NG >= 1 && NG <= kMlMaxGroups && NG % GPW == 0 && NG % SPW == 0, "lane groups per wave");
  constexpr int NC = GT * GPW; /* candidates per lane of a wave */
  static_assert(3 * SPW <= NC, "a self wave keeps three groups per lane group in the slot arrays");
  constexpr int NU = GPW > SPW ? GPW : SPW;
  constexpr int LN = 64 * NG;
  constexpr int nSW = NG / SPW, nGS = NG / GPW;
  using LDS = MlaneLds<NG>;
  LDS& S = *(LDS*)smem;
  const int b = P.uttMap ? P.uttMap[blockIdx.x] : (int)blockIdx.x;
  const int W = (int)blockDim.x, tid = (int)threadIdx.x;
  const int lane = laneId(), wave = waveUniform(waveId());
  const int nW = W >> 6;
  const int nTW = nW - nSW - 1;
  const int nBlk = nTW / nGS;
  const int prepWave = nW - 1;
  const bool isTokW = wave < nTW, isSvcW = wave == prepWave;
#ifndef FLTX_EMU
  if (!(P.tune & 1) && !isTokW) { /* the waves the token waves wait for win the issue arbitration of their SIMD (C2 shape, beam 100:
                                     3.13 -> 2.96 ms; tune bit 0 switches it off for measurements) */
    __builtin_amdgcn_s_setprio(3);
  }
#endif
  const int blk = isTokW ? wave % nBlk : 0;
  const int g0 = isTokW ? (wave / nBlk) * GPW : (isSvcW ? 0 : (wave - nTW) * SPW); /* first lane group of this wave */
  const int pos0 = blk * GT;
  const int K = P.K, N = P.N;
  const bool ctc = P.criterion == 1;
  const int T = P.stepT ? P.stepT[b] : 0;
  const float* em = P.emissions ? P.emissions + P.emOff[b] : nullptr;
  const int64_t hbase = P.histOff[b];
  const double NEG = slNegInf();
  const int2* const tokLm = TL ? P.tokLm : nullptr;
  const int tokStride = TL ? P.tokLmStride : 0;
  const double lmW = P.lmWeight;
  unsigned long long* const idTab = TL ? P.ymemo + (size_t)b * P.ymemoSlots : nullptr;
  if constexpr (TL) {
    for (uint32_t i = (uint32_t)tid; i < P.ymemoSlots; i += (uint32_t)W) {
      idTab[i] = 0ull; /* (at L2 before the barrier below, where the build's atomics will find it) */
    }
#ifndef FLTX_EMU
    __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    if (tid == 0) {
      S.ctx[0][0] = 0u; /* KenLM::start(false): row 0 of the table */
      S.tlIn[0][0] = 0.0f;
      S.fbest[0] = 0ull;
      S.fbest[1] = 0ull;
      S.idNext = 1u; /* (0 names the root state) */
    }
  }

  /* ---- decodeBegin (LexiconFreeDecoder.cpp:20-28): the root state ------------------ */
  for (int i = tid; i < 2 * LN; i += W) {
    ((unsigned long long*)S.cmask)[i] = 0ull;
    ((unsigned long long*)S.mask)[i] = 0ull;
  }
  for (int i = tid; i < 2 * kSlNB; i += W) {
    ((uint32_t*)S.hist)[i] = 0u;
  }
  if (tid < 32) {
    S.off[tid] = 0u;
  }
  if (tid < 16) {
    S.scal[tid] = 0u;
  }
  if (tid < kMlMaxGroups) {
    S.grp[tid] = 0u;
  }
  if (tid == 0) {
    MlRec r;
    r.nb = 0.0;
    r.b = NEG;
    r.info = (uint32_t)P.sil;
    r.sid = 0u;
    r.spar = 0xFFFFFFFFu;
    r.hyps = 0u | (kMlNoHyp << 16);
    S.rec[0][0] = r;
    S.row[0].nev = 0u;
    S.row[1].nev = 0u;
    S.row[0].dead = 0u;
    S.row[1].dead = 0u;
    S.mmaxKey[0] = f64Key(0.0);
    S.mmaxKey[1] = 0ull;
    P.histPT[hbase] = make_int2((int)kMlNoHyp, P.sil);
  }
  for (int i = tid; i < K; i += W) { /* unused slots of a row never look like the record of a new state (mlReenter) */
    if (i > 0) {
      P.histPT[hbase + i] = make_int2((int)kMlNoHyp, -1);
    }
  }
  double bestChain = 0.0;
  if (wave == prepWave) {
    const float v0 = (T > 0 && lane < N) ? em[lane] : 0.0f;
    ldsRowLoad(S.raw[1], em + (size_t)1 * N + lane, T > 1 && lane < N);
    ldsRowLoad(S.raw[2], em + (size_t)2 * N + lane, T > 2 && lane < N);
    SlRowRegs r0 = slRowScan(P, v0, ctc, 0.0);
    bestChain = r0.best;
    slRowStore(P, S, 0, r0, 2);
    slRowStore(P, S, 1, r0, 2);
  }
  ldsBarrier();

  int nState = 1;
  double endBest = 0.0;
  int winShift = kSlCoarseShift, winBase = kSlCoarseBase;
  bool dead = false;
  const int sil = P.sil, blank = P.blank;
  const double silScore = P.silScore;
  int2* const histPT = P.histPT;

  auto frameStep = [&](auto PT, auto RL, const int t) {
    constexpr int p = decltype(PT)::value, q = p ^ 1;
    constexpr bool isTok = decltype(RL)::value == 0, isSelf = decltype(RL)::value == 1, isSvc = decltype(RL)::value == 2;
    constexpr int U = isTok ? GPW : (isSelf ? SPW : 0); /* lane groups this wave holds */
    const int frameOut = t + 1;
    const int64_t hrow = hbase + (int64_t)frameOut * K;
    /* ---- phase 1: own states, candidates, histogram ------------------------------------ */
    double best = S.row[p].best, thr = S.row[p].thr;
    const int silPos = S.row[p].silPos;
    uint32_t rowDead = S.row[p].dead;
    const uint32_t nev = S.row[p].nev;
    if (LA && !TL) {
      const double mmax = f64FromKey(S.mmaxKey[p]);
      const uint32_t ek = S.row[p].ekey;
      const double sS = (mmax + (double)S.row[p].esil) + silScore;
      bool any = ek != 0u;
      best = any ? mmax + (double)f32FromKey(ek) : 0.0;
      if (((S.row[p].allow >> sil) & 1ull) != 0ull && sS == sS && (!any || sS > best)) {
        best = sS;
        any = true;
      }
      thr = best - P.beamThreshold;
      rowDead = (!any || !(best - best == 0.0)) ? 1u : 0u;
    }
    MlRec me[NU];
    unsigned long long cm[NU], mk[NU];
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      me[i] = MlRec{};
      cm[i] = 0ull;
      mk[i] = 0ull;
      if (i < U) {
        const int L = (g0 + i) * 64 + lane;
        me[i] = S.rec[p][L];
        cm[i] = S.cmask[p][L];
        mk[i] = S.mask[p][L];
      }
    }
    double ev[GT];
    unsigned long long tb[GT];
    double eBlank = 0.0;
    unsigned long long allow = 0ull;
#pragma unroll
    for (int j = 0; j < GT; ++j) {
      ev[j] = 0.0;
      tb[j] = 0ull;
    }
    if constexpr (isSelf) {
      eBlank = S.eAll[p][ctc ? blank : 0];
      allow = S.row[p].allow;
    } else if constexpr (isTok) {
#pragma unroll
      for (int j = 0; j < GT; ++j) {
        ev[j] = S.eTok[p][pos0 + j];
        tb[j] = S.tokBit[p][pos0 + j];
      }
    }
    /* TL: the LM scores (and the contexts behind them) of this lane's states for the tokens of this wave's list
     * positions -- one 8-byte gather per (state, position) from the state's row of the dense table, issued as soon as
     * the rows are known; self waves: the score of last(S) after S (blank-then-last) and the score S was entered with */
    int2 lmv[TL ? NC : 1];
    int2 lmLast[TL ? NU : 1];
    float lIn[TL ? NU : 1];
    uint32_t ctxv[TL ? NU : 1];
    if constexpr (TL) {
#pragma unroll
      for (int i = 0; i < NU; ++i) {
        ctxv[i] = 0u;
        lmLast[i] = make_int2(0, 0);
        lIn[i] = 0.0f;
        if (i < U) {
          const int L = (g0 + i) * 64 + lane;
          const bool lv = L < nState;
          ctxv[i] = lv ? S.ctx[p][L] : 0u;
          const int2* lrow = tokLm + (size_t)ctxv[i] * (size_t)tokStride;
          if constexpr (isSelf) {
            lmLast[i] = lrow[lv ? (int)(me[i].info & 63u) : 0];
            lIn[i] = S.tlIn[p][L];
          } else if constexpr (isTok) {
#pragma unroll
            for (int j = 0; j < GT; ++j) {
              lmv[i * GT + j] = lrow[tb[j] != 0ull ? __builtin_ctzll(tb[j]) : 0];
            }
          }
        }
      }
    }
    if (nev != 0u) { /* states re-entered the beam in the previous build (TL: named by the id table; else rare) */
      if constexpr (TL) {
        mlRelink(S, p, nState);
      } else {
        mlReenter(S, histPT, p, nState, hbase, (int64_t)frameOut * K);
      }
#pragma unroll
      for (int i = 0; i < NU; ++i) {
        if (i < U) {
          const int L = (g0 + i) * 64 + lane;
          me[i] = S.rec[p][L];
          cm[i] = S.cmask[p][L];
          mk[i] = S.mask[p][L];
        }
      }
    }
    if (rowDead || (TL && S.scal[ML_FULL] != 0u)) {
      dead = true;
      return;
    }
    bool live[NU], whichB[NU];
    double nbv[NU], bbv[NU], m[NU];
    int last[NU], pl[NU];
    uint32_t hypNB[NU], hypB[NU], hypM[NU];
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      live[i] = i < U && (g0 + i) * 64 + lane < nState;
      nbv[i] = live[i] ? me[i].nb : NEG;
      bbv[i] = live[i] ? me[i].b : NEG;
      last[i] = (int)(me[i].info & 63u);
      pl[i] = live[i] ? (int)(me[i].info >> 8) - 1 : -1;
      hypNB[i] = me[i].hyps & 0xFFFFu;
      hypB[i] = me[i].hyps >> 16;
      whichB[i] = bbv[i] > nbv[i];
      m[i] = whichB[i] ? bbv[i] : nbv[i];
      hypM[i] = whichB[i] ? hypB[i] : hypNB[i];
    }
    /* self waves: what their groups need beyond the lanes' own records (second LDS round trip) */
    MlRec par[NU];
    double eLast[NU];
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      par[i] = MlRec{};
      eLast[i] = 0.0;
      if (isSelf && i < U) {
        par[i] = S.rec[p][pl[i] >= 0 ? pl[i] : 0];
        eLast[i] = S.eAll[p][last[i]];
      }
    }
    /* candidate scores: token waves without logAdd recompute m + e where they need it (one addition, the same
     * double) instead of keeping GT * GPW of them in registers through the selection */
    constexpr bool kKeep = TL || !(isTok && !LA);
    double cs[kKeep ? NC : 1];
    int cbin[NC];
    uint32_t parR[NU];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      if (kKeep) {
        cs[c] = NEG;
      }
      cbin[c] = kSlInvalid;
    }
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      parR[i] = kMlNoHyp;
    }
    SlRowRegs nextRow = {};
    if constexpr (isSvc) {
      if (t + 1 < T) {
        ldsRowWait();
        const float rv = lane < N ? S.raw[(t + 1) % 3][lane] : 0.0f;
        nextRow = slRowScan(P, rv, ctc, bestChain);
        bestChain = nextRow.best;
      }
      ldsRowLoad(S.raw[t % 3], em + (size_t)(t + 3) * N + lane, t + 3 < T && lane < N);
      if (LA && lane == 0) {
        S.mmaxKey[q] = 0ull;
      }
      if (TL && lane == 0) {
        S.fbest[q] = 0ull; /* (the next frame's; last read a frame ago) */
      }
    } else if constexpr (TL) {
      /* (the candidates of a token-LM frame: below, around the barrier that publishes the frame's best) */
    } else if constexpr (isTok) {
      const int silJ = silPos - pos0;
#pragma unroll
      for (int i = 0; i < U; ++i) {
        /* tokens this lane does not extend with here: its own last token (repeat and blank-then-last belong to the
         * self wave) and those whose child state holds a lane (that lane merges the extension into its repeat) */
        const uint32_t lastLo = last[i] < 32 ? 1u << last[i] : 0u, lastHi = last[i] < 32 ? 0u : 1u << (last[i] - 32);
        const uint32_t skLo = live[i] ? ((uint32_t)cm[i] | lastLo) : 0xFFFFFFFFu;
        const uint32_t skHi = live[i] ? ((uint32_t)(cm[i] >> 32) | lastHi) : 0xFFFFFFFFu;
#pragma unroll
        for (int j = 0; j < GT; ++j) {
          double c = m[i] + ev[j]; /* NaN past the end of the list */
          if (j == silJ) {
            c = c + silScore;
          }
          const uint32_t hit = (skLo & (uint32_t)tb[j]) | (skHi & (uint32_t)(tb[j] >> 32));
          const bool ok = hit == 0u && c >= thr;
          if (LA) {
            double c2 = (whichB[i] ? nbv[i] : bbv[i]) + ev[j];
            if (j == silJ) {
              c2 = c2 + silScore;
            }
            if (ok && (whichB[i] ? hypNB[i] : hypB[i]) != kMlNoHyp && c2 >= thr) {
              c = slLogAdd(c, c2);
            }
          }
          if (kKeep) {
            cs[i * GT + j] = c;
          }
          cbin[i * GT + j] = ok ? slBin<LA>(best, c, winShift, winBase) : kSlInvalid;
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < U; ++i) {
        const bool lastOk = live[i] && ((allow >> last[i]) & 1ull) != 0ull && !(ctc && last[i] == blank);
        const bool lastSil = last[i] == sil;
        /* (S, blank, true): LexiconFreeDecoder.cpp:86-97 */
        double cB = m[i] + eBlank;
        if (blank == sil) {
          cB = cB + silScore;
        }
        const bool okB = ctc && live[i] && ((allow >> (ctc ? blank : 0)) & 1ull) != 0ull && cB >= thr;
        if (LA) {
          double cB2 = (whichB[i] ? nbv[i] : bbv[i]) + eBlank;
          if (blank == sil) {
            cB2 = cB2 + silScore;
          }
          if (okB && (whichB[i] ? hypNB[i] : hypB[i]) != kMlNoHyp && cB2 >= thr) {
            cB = slLogAdd(cB, cB2);
          }
        }
        /* (S, last, false): the repeat (:98-110) and the parent state's extension by last (:69-85) */
        const int lastP = (int)(par[i].info & 63u);
        const uint32_t h1 = par[i].hyps & 0xFFFFu, h2 = par[i].hyps >> 16;
        const bool has0 = hypNB[i] != kMlNoHyp;
        const bool has1 = pl[i] >= 0 && last[i] != lastP && h1 != kMlNoHyp;
        const bool has2 = pl[i] >= 0 && ctc && h2 != kMlNoHyp;
        const bool hasB = hypB[i] != kMlNoHyp;
        double r0 = nbv[i] + eLast[i];
        double r1 = has1 ? par[i].nb + eLast[i] : NEG;
        double r2 = has2 ? par[i].b + eLast[i] : NEG;
        double cL = bbv[i] + eLast[i]; /* (S.last, last, false) from (S, blank, true) when no lane holds S.last */
        if (silScore != 0.0) {
          r0 = lastSil ? r0 + silScore : r0;
          r1 = lastSil ? r1 + silScore : r1;
          r2 = lastSil ? r2 + silScore : r2;
          cL = lastSil ? cL + silScore : cL;
        }
        /* max-merge (Utils.h:194-196); a tie goes to the lower history slot */
        double cR = r0;
        uint32_t pR = hypNB[i];
        if (has1 && (r1 > cR || (r1 == cR && h1 < pR))) {
          cR = r1;
          pR = h1;
        }
        if (has2 && (r2 > cR || (r2 == cR && h2 < pR))) {
          cR = r2;
          pR = h2;
        }
        bool okR = lastOk && (has0 || has1 || has2) && cR >= thr;
        if (LA) {
          const bool v0 = has0 && r0 >= thr, v1 = has1 && r1 >= thr, v2 = has2 && r2 >= thr;
          okR = lastOk && (v0 || v1 || v2);
          double a = v0 ? r0 : NEG, bq = v1 ? r1 : NEG, cq = v2 ? r2 : NEG;
          double t0 = a > bq ? a : bq, t1 = a > bq ? bq : a;
          const double hi = t0 > cq ? t0 : cq;
          const double mid = t0 > cq ? (t1 > cq ? t1 : cq) : t0;
          const double lo = t0 > cq ? (t1 > cq ? cq : t1) : t1;
          double acc = hi;
          if (mid > NEG) {
            acc = slLogAdd(acc, mid);
          }
          if (lo > NEG) {
            acc = slLogAdd(acc, lo);
          }
          cR = okR ? acc : cR;
        }
        const bool okL = ctc && lastOk && hasB && ((cm[i] >> last[i]) & 1ull) == 0ull && cL >= thr;
        parR[i] = pR;
        cs[3 * i + 0] = cB;
        cs[3 * i + 1] = cR;
        cs[3 * i + 2] = cL;
        cbin[3 * i + 0] = okB ? slBin<LA>(best, cB, winShift, winBase) : kSlInvalid;
        cbin[3 * i + 1] = okR ? slBin<LA>(best, cR, winShift, winBase) : kSlInvalid;
        cbin[3 * i + 2] = okL ? slBin<LA>(best, cL, winShift, winBase) : kSlInvalid;
      }
    }
    if constexpr (TL) {
      /* ---- the same candidates with the LM term (LexiconFreeDecoder.cpp:64-67,69-85: score = prev.score + e
       * (+ silScore), candidate = score + lmWeight * lmScore); the frame's best is their maximum (Utils.h:131-137) --- */
      bool pre[NC];
      /* logAdd: the smaller member of a token wave's group / of the blank group; the repeat group's three members and
       * which of them exist (whether a member passes the threshold is known after the barrier) */
      double c2v[LA ? NC : 1];
      bool hasOther[NU];
      double r0a[LA ? NU : 1], r1a[LA ? NU : 1], r2a[LA ? NU : 1];
      bool lastOkA[NU], has0a[NU], has1a[NU], has2a[NU];
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        pre[c] = false;
        if (LA) {
          c2v[LA ? c : 0] = NEG;
        }
      }
#pragma unroll
      for (int i = 0; i < NU; ++i) {
        hasOther[i] = (whichB[i] ? hypNB[i] : hypB[i]) != kMlNoHyp;
        lastOkA[i] = has0a[i] = has1a[i] = has2a[i] = false;
      }
      if constexpr (isTok) {
        const int silJ = silPos - pos0;
#pragma unroll
        for (int i = 0; i < U; ++i) {
          const uint32_t lastLo = last[i] < 32 ? 1u << last[i] : 0u, lastHi = last[i] < 32 ? 0u : 1u << (last[i] - 32);
          const uint32_t skLo = live[i] ? ((uint32_t)cm[i] | lastLo) : 0xFFFFFFFFu;
          const uint32_t skHi = live[i] ? ((uint32_t)(cm[i] >> 32) | lastHi) : 0xFFFFFFFFu;
#pragma unroll
          for (int j = 0; j < GT; ++j) {
            const double wl = lmW * (double)__uint_as_float((uint32_t)lmv[i * GT + j].x);
            double c = m[i] + ev[j]; /* NaN past the end of the list */
            if (j == silJ) {
              c = c + silScore;
            }
            c = c + wl;
            const uint32_t hit = (skLo & (uint32_t)tb[j]) | (skHi & (uint32_t)(tb[j] >> 32));
            pre[i * GT + j] = hit == 0u && c == c;
            cs[i * GT + j] = c;
            if (LA) { /* the state's other hypothesis reaches the same child state */
              double c2 = (whichB[i] ? nbv[i] : bbv[i]) + ev[j];
              if (j == silJ) {
                c2 = c2 + silScore;
              }
              c2v[LA ? i * GT + j : 0] = c2 + wl;
            }
          }
        }
      } else if constexpr (isSelf) {
#pragma unroll
        for (int i = 0; i < U; ++i) {
          const bool lastOk = live[i] && ((allow >> last[i]) & 1ull) != 0ull && !(ctc && last[i] == blank);
          const bool lastSil = last[i] == sil;
          /* (S, blank, true): :86-97, no LM term */
          double cB = m[i] + eBlank;
          if (blank == sil) {
            cB = cB + silScore;
          }
          pre[3 * i] = ctc && live[i] && ((allow >> (ctc ? blank : 0)) & 1ull) != 0ull && cB == cB;
          if (LA) {
            double cB2 = (whichB[i] ? nbv[i] : bbv[i]) + eBlank;
            if (blank == sil) {
              cB2 = cB2 + silScore;
            }
            c2v[LA ? 3 * i : 0] = cB2;
          }
          /* (S, last, false): the repeat (:98-110, no LM term) and the parent state's extension by last (:69-85: the LM
           * score S was entered with) */
          const int lastP = (int)(par[i].info & 63u);
          const uint32_t h1 = par[i].hyps & 0xFFFFu, h2 = par[i].hyps >> 16;
          const bool has0 = hypNB[i] != kMlNoHyp;
          const bool has1 = pl[i] >= 0 && last[i] != lastP && h1 != kMlNoHyp;
          const bool has2 = pl[i] >= 0 && ctc && h2 != kMlNoHyp;
          const bool hasB = hypB[i] != kMlNoHyp;
          const double wIn = lmW * (double)lIn[i], wL = lmW * (double)__uint_as_float((uint32_t)lmLast[i].x);
          double r0 = nbv[i] + eLast[i];
          double r1 = par[i].nb + eLast[i];
          double r2 = par[i].b + eLast[i];
          double cL = bbv[i] + eLast[i]; /* (S.last, last, false) from (S, blank, true) when no lane holds S.last */
          if (silScore != 0.0) {
            r0 = lastSil ? r0 + silScore : r0;
            r1 = lastSil ? r1 + silScore : r1;
            r2 = lastSil ? r2 + silScore : r2;
            cL = lastSil ? cL + silScore : cL;
          }
          r1 = has1 ? r1 + wIn : NEG;
          r2 = has2 ? r2 + wIn : NEG;
          cL = cL + wL;
          double cR = r0;
          uint32_t pR = hypNB[i];
          if (has1 && (r1 > cR || (r1 == cR && h1 < pR))) {
            cR = r1;
            pR = h1;
          }
          if (has2 && (r2 > cR || (r2 == cR && h2 < pR))) {
            cR = r2;
            pR = h2;
          }
          parR[i] = pR;
          lastOkA[i] = lastOk;
          has0a[i] = has0;
          has1a[i] = has1;
          has2a[i] = has2;
          if (LA) {
            r0a[LA ? i : 0] = r0;
            r1a[LA ? i : 0] = r1;
            r2a[LA ? i : 0] = r2;
          }
          pre[3 * i + 1] = lastOk && (has0 || has1 || has2) && cR == cR;
          pre[3 * i + 2] = ctc && lastOk && hasB && ((cm[i] >> last[i]) & 1ull) == 0ull && cL == cL;
          cs[3 * i + 0] = cB;
          cs[3 * i + 1] = cR;
          cs[3 * i + 2] = cL;
        }
      }
      if constexpr (!isSvc) {
        unsigned long long k = 0ull;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const unsigned long long kc = pre[c] ? f64Key(cs[c]) : 0ull;
          k = kc > k ? kc : k;
        }
        if (waveBallot(k != 0ull) != 0ull) {
          k = waveMax64(k);
          if (lane == 0) {
            atomMax64(&S.fbest[p], k);
          }
        }
      }
      ldsBarrier(); /* 0: the frame's best candidate */
      {
        const unsigned long long bk = S.fbest[p];
        best = f64FromKey(bk);
        thr = best - P.beamThreshold;
        if (bk == 0ull || !(best - best == 0.0)) { /* no candidate at all, or not finite: the general engines */
          dead = true;
          return;
        }
      }
      if constexpr (isTok) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const bool ok = pre[c] && cs[c] >= thr;
          if (LA && ok && hasOther[c / GT] && c2v[LA ? c : 0] >= thr) {
            cs[c] = slLogAdd(cs[c], c2v[LA ? c : 0]);
          }
          cbin[c] = ok ? slBin<LA>(best, cs[c], winShift, winBase) : kSlInvalid;
        }
      } else if constexpr (isSelf) {
#pragma unroll
        for (int i = 0; i < U; ++i) {
          const bool okB = pre[3 * i] && cs[3 * i] >= thr;
          if (LA && okB && hasOther[i] && c2v[LA ? 3 * i : 0] >= thr) {
            cs[3 * i] = slLogAdd(cs[3 * i], c2v[LA ? 3 * i : 0]);
          }
          bool okR = pre[3 * i + 1] && cs[3 * i + 1] >= thr;
          if (LA) { /* the members that pass the threshold, best first (Utils.h:186-193) */
            const double r0 = r0a[LA ? i : 0], r1 = r1a[LA ? i : 0], r2 = r2a[LA ? i : 0];
            const bool v0 = has0a[i] && r0 >= thr, v1 = has1a[i] && r1 >= thr, v2 = has2a[i] && r2 >= thr;
            okR = lastOkA[i] && (v0 || v1 || v2);
            const double a = v0 ? r0 : NEG, bq = v1 ? r1 : NEG, cq = v2 ? r2 : NEG;
            const double t0 = a > bq ? a : bq, t1 = a > bq ? bq : a;
            const double hi = t0 > cq ? t0 : cq;
            const double mid = t0 > cq ? (t1 > cq ? t1 : cq) : t0;
            const double lo = t0 > cq ? (t1 > cq ? cq : t1) : t1;
            double accv = hi;
            if (mid > NEG) {
              accv = slLogAdd(accv, mid);
            }
            if (lo > NEG) {
              accv = slLogAdd(accv, lo);
            }
            cs[3 * i + 1] = okR ? accv : cs[3 * i + 1];
          }
          const bool okL = pre[3 * i + 2] && cs[3 * i + 2] >= thr;
          cbin[3 * i] = okB ? slBin<LA>(best, cs[3 * i], winShift, winBase) : kSlInvalid;
          cbin[3 * i + 1] = okR ? slBin<LA>(best, cs[3 * i + 1], winShift, winBase) : kSlInvalid;
          cbin[3 * i + 2] = okL ? slBin<LA>(best, cs[3 * i + 2], winShift, winBase) : kSlInvalid;
        }
      }
    }
    const int silJ0 = silPos - pos0;
    auto csAt = [&](int c) -> double {
      if constexpr (kKeep) {
        return cs[c];
      } else {
        double v = m[c / GT] + ev[c % GT];
        if (c % GT == silJ0) {
          v = v + silScore;
        }
        return v;
      }
    };
    /* housekeeping: what this frame's build adds to (the first block's waves wipe their groups' masks) */
    if (isTok && blk == 0) {
#pragma unroll
      for (int i = 0; i < U; ++i) {
        S.cmask[q][(g0 + i) * 64 + lane] = 0ull;
        S.mask[q][(g0 + i) * 64 + lane] = 0ull;
      }
    }
    if (wave == 0) {
      if (lane < 32) {
        S.off[lane] = 0u;
      }
      if (lane == 0) {
        S.scal[ML_BCNT] = 0u;
      }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      if (cbin[c] < kSlFar) {
        atomAdd32(&S.hist[p][cbin[c]], 1u);
      }
    }
    ldsBarrier(); /* 1 */
    /* ---- phase 2: which candidates survive (Utils.h:200-220) ---------------------------- */
    unsigned long long selMask[NC];
    /* (the loop form: the straight-line form of fltx_slane.h costs the eight-group geometry a fifth of its frame --
     * beam 500 on the C2 shape 22.7 -> 27.6 ms -- and gains the others 2 %) */
    SlScan sc;
    int shift = winShift, base = winBase;
    unsigned long long bLo = 0ull, bHi = 0x7FFFFFFFull;
    bool full = false;
    for (;;) {
      sc = slScan(S.hist[p], K, !full);
      if (!full && !sc.crossed) {
        int nFar = 0;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          nFar += popc64(waveBallot(cbin[c] == kSlFar));
        }
        if (lane == 0 && nFar > 0) {
          atomAdd32(&S.hist[p][kSlFar], (uint32_t)nFar);
        }
        full = true;
        ldsBarrier();
        continue;
      }
      if (sc.total <= K) {
        const int lim = full ? kSlFar : kSlFar - 1;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          selMask[c] = waveBallot(cbin[c] <= lim);
        }
        break;
      }
      const int need = K - sc.cum;
      if (sc.cnt == need) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          selMask[c] = waveBallot(cbin[c] <= sc.bstar);
        }
        break;
      }
      if (sc.cnt <= kSlBCap) { /* the members of the K-th best's bin compare with each other */
        uint32_t take = 0u;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          if (cbin[c] == sc.bstar) {
            const uint32_t i = atomAdd32(&S.scal[ML_BCNT], 1u);
            S.bKey[i] = f64Key(csAt(c));
            S.bOrd[i] = ((uint32_t)wave << 16) | ((uint32_t)c << 8) | (uint32_t)lane;
          }
        }
        ldsBarrier();
        take |= slRankBin<NC>(S.bKey, S.bOrd, sc.cnt, need, wave, [&](int c) { return cbin[c] == sc.bstar; },
                              [&](int c) { return f64Key(csAt(c)); }); /* (broadcast + ballot: fltx_slane.h) */
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          selMask[c] = waveBallot(cbin[c] < sc.bstar || ((take >> c) & 1u) != 0u);
        }
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
      for (int c = 0; c < NC; ++c) {
        if (cbin[c] != kSlInvalid) {
          cbin[c] = slBin<LA>(best, csAt(c), shift, base);
          atomAdd32(&S.hist[p][cbin[c]], 1u);
        }
      }
      ldsBarrier();
    }
    if (dead) {
      return;
    }
    if (sc.total > K) { /* next frame's window: the K-th best in the middle, 128 bins per octave */
      const int q15 = shift >= kSlFineShift ? (sc.bstar + base) << (shift - kSlFineShift)
                                            : (sc.bstar + base) >> (kSlFineShift - shift);
      winShift = kSlFineShift;
      winBase = q15 > kSlMid ? q15 - kSlMid : 0;
    }
    /* new lanes: survivors first (group by group), then the new states wave by wave */
    int nNewWave = 0;
    int surv[NU];
    uint32_t hNB[NU], hB[NU];
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      surv[i] = -1;
      hNB[i] = 0u;
      hB[i] = 0u;
    }
    if constexpr (isTok) {
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        nNewWave += popc64(selMask[c]);
      }
    } else if constexpr (isSelf) {
#pragma unroll
      for (int i = 0; i < U; ++i) {
        const unsigned long long balB = selMask[3 * i], balR = selMask[3 * i + 1], balL = selMask[3 * i + 2];
        const unsigned long long balS = balB | balR;
        const bool sR = ((balR >> lane) & 1ull) != 0ull;
        surv[i] = ((balS >> lane) & 1ull) ? wavePrefixCount(balS) : -1;
        hNB[i] = (uint32_t)(wavePrefixCount(balR) + wavePrefixCount(balB));
        hB[i] = hNB[i] + (sR ? 1u : 0u);
        nNewWave += popc64(balL);
        S.newLane[(g0 + i) * 64 + lane] = surv[i];
        if (lane == 0) {
          S.grp[g0 + i] = (uint32_t)popc64(balS) | ((uint32_t)(popc64(balR) + popc64(balB)) << 16);
        }
      }
    }
    if (!isSvc && lane > wave && lane < nW && nNewWave > 0) {
      atomAdd32(&S.off[lane], (uint32_t)nNewWave);
    }
    if (LA && !TL && !isSvc) { /* the best hypothesis of the next beam */
      unsigned long long k = 0ull;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const unsigned long long kj = ((selMask[c] >> lane) & 1ull) ? f64Key(csAt(c)) : 0ull;
        k = kj > k ? kj : k;
      }
      if (waveBallot(k != 0ull) != 0ull) {
        k = waveMax64(k);
        if (lane == 0) {
          atomMax64(&S.mmaxKey[q], k);
        }
      }
    }
    ldsBarrier(); /* 2 */
    /* ---- phase 3: every survivor is written by the lane that evaluated it ---------------- */
    uint32_t gcnt[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      gcnt[g] = S.grp[g];
    }
    const int offW = (int)S.off[wave], nNew = (int)S.off[nW - 1];
    int mySlot[NU], plSlot[NU];
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      mySlot[i] = -1;
      plSlot[i] = -1;
      if (i < U) {
        mySlot[i] = S.newLane[(g0 + i) * 64 + lane];
        if (isSelf) {
          plSlot[i] = S.newLane[pl[i] >= 0 ? pl[i] : 0];
        }
      }
    }
    int nSurv = 0, nHSurv = 0;
    int baseS[NU], baseH[NU], basePl[NU];
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      baseS[i] = 0;
      baseH[i] = 0;
      basePl[i] = 0;
    }
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int cS = (int)(gcnt[g] & 0xFFFFu), cH = (int)(gcnt[g] >> 16);
#pragma unroll
      for (int i = 0; i < NU; ++i) {
        if (i < U) {
          baseS[i] += g < g0 + i ? cS : 0;
          baseH[i] += g < g0 + i ? cH : 0;
          if (isSelf) {
            basePl[i] += g < (pl[i] >> 6) ? cS : 0;
          }
        }
      }
      nSurv += cS;
      nHSurv += cH;
    }
    auto newState = [&](int idx, double c, int n, uint32_t hp, uint32_t srcSid, unsigned long long srcMask, int srcNew,
                        uint32_t ctxNew, float lNew) {
      const int nl = nSurv + idx;
      const uint32_t hyp = (uint32_t)(nHSurv + idx);
      MlRec r;
      r.nb = c;
      r.b = NEG;
      r.info = (uint32_t)n | ((uint32_t)(srcNew + 1) << 8);
      r.hyps = hyp | (kMlNoHyp << 16);
      r.sid = (uint32_t)frameOut * (uint32_t)K + hyp;
      r.spar = srcSid;
      bool again = ((srcMask >> n) & 1ull) != 0ull; /* this edge had a child before */
      if constexpr (TL) {
        /* the id table says whether it had, and who the child is */
        bool fresh = true;
        r.sid = mlChildId(idTab, P.ymemoSlots, srcSid, (uint32_t)n, &S.idNext, fresh);
        again = !fresh;
        if (r.sid == 0xFFFFFFFFu) {
          S.scal[ML_FULL] = 1u; /* table full: general path */
          again = false;
        }
        S.ctx[q][nl] = ctxNew;
        S.tlIn[q][nl] = lNew;
      }
      S.rec[q][nl] = r;
      if (srcNew >= 0) {
        atomOr64(&S.cmask[q][srcNew], 1ull << n);
        if constexpr (!TL) {
          atomOr64(&S.mask[q][srcNew], 1ull << n);
        }
      }
      histPT[hrow + hyp] = mlNewRec(hp, srcSid, n);
      if (again) { /* it may have descendants in the beam */
        const uint32_t e = atomAdd32(&S.row[q].nev, 1u);
        S.evLane[e] = (uint32_t)nl;
        if constexpr (TL) {
          S.evSid[e] = r.sid;
        } else {
          S.evSpar[e] = srcSid;
          S.evTok[e] = (uint32_t)n;
        }
      }
    };
    /* TL: a new state costs a look-up (and, for a fresh one, a compare-and-swap) in the id table in HBM -- a microsecond
     * or two, paid once per list position with a survivor if the positions are built one after the other.  Instead the
     * wave's survivors drop what a new state is made from into the wave's scratch and the wave's first lanes build one
     * state each: their table accesses are in flight together. */
    auto gatherPut = [&](int r, double c, int n, uint32_t hp, uint32_t srcSid, int srcNew, uint32_t ctxNew, float lNew) {
      uint4* sc4 = (uint4*)S.tlNew[wave];
      const unsigned long long cb = (unsigned long long)__double_as_longlong(c);
      sc4[2 * r] = make_uint4((uint32_t)cb, (uint32_t)(cb >> 32), (uint32_t)n | (hp << 8), srcSid);
      sc4[2 * r + 1] = make_uint4((uint32_t)(srcNew + 1), ctxNew, __float_as_uint(lNew), 0u);
    };
    auto gatherBuild = [&](int first, int count) {
      waveSync();
      if (lane < count) {
        const uint4* sc4 = (const uint4*)S.tlNew[wave];
        const uint4 a = sc4[2 * lane], bq = sc4[2 * lane + 1];
        newState(first + lane, __longlong_as_double((long long)(((unsigned long long)a.y << 32) | a.x)), (int)(a.z & 0xFFu), a.z >> 8,
                 a.w, 0ull, (int)bq.x - 1, bq.y, __uint_as_float(bq.z));
      }
    };
    if constexpr (isSvc) {
      if (t + 1 < T) {
        slRowStore(P, S, q, nextRow, P.Kt < N ? 1 : 0);
      }
      ((uint4*)S.hist[q])[lane] = make_uint4(0u, 0u, 0u, 0u);
    } else if constexpr (isTok) {
      int before = offW; /* new states of the waves before this one and of this wave's earlier slots */
      const bool gathered = TL && nNewWave > 1 && nNewWave <= kMlGather;
#pragma unroll
      for (int i = 0; i < U; ++i) {
        const int srcNew = mySlot[i] >= 0 ? baseS[i] + mySlot[i] : -1;
#pragma unroll
        for (int j = 0; j < GT; ++j) {
          const int c = i * GT + j;
          if (selMask[c] != 0ull) { /* (most positions of most frames have no survivor at all) */
            if ((selMask[c] >> lane) & 1ull) {
              const int nTok = (int)S.tokId[p][pos0 + j];
              if constexpr (TL) {
                if (gathered) {
                  gatherPut(before - offW + wavePrefixCount(selMask[c]), csAt(c), nTok, hypM[i], me[i].sid, srcNew,
                            (uint32_t)lmv[TL ? c : 0].y, __uint_as_float((uint32_t)lmv[TL ? c : 0].x));
                } else {
                  newState(before + wavePrefixCount(selMask[c]), csAt(c), nTok, hypM[i], me[i].sid, 0ull, srcNew,
                           (uint32_t)lmv[TL ? c : 0].y, __uint_as_float((uint32_t)lmv[TL ? c : 0].x));
                }
              } else {
                newState(before + wavePrefixCount(selMask[c]), csAt(c), nTok, hypM[i], me[i].sid, mk[i], srcNew, 0u, 0.0f);
              }
            }
            before += popc64(selMask[c]);
          }
        }
      }
      if constexpr (TL) {
        if (gathered) {
          gatherBuild(offW, nNewWave);
        }
      }
    } else {
      for (int i = (wave - nTW) * 64 + lane; i < K; i += nSW * 64) { /* unused slots of the history row: see mlReenter */
        if (i >= nHSurv + nNew) {
          histPT[hrow + i] = make_int2((int)kMlNoHyp, -1);
        }
      }
      int before = offW;
      const bool gatheredS = TL && nNewWave > 1 && nNewWave <= kMlGather;
#pragma unroll
      for (int i = 0; i < U; ++i) {
        const int srcNew = mySlot[i] >= 0 ? baseS[i] + mySlot[i] : -1;
        if (surv[i] >= 0) {
          const bool sB = ((selMask[3 * i] >> lane) & 1ull) != 0ull, sR = ((selMask[3 * i + 1] >> lane) & 1ull) != 0ull;
          const int pln = (pl[i] >= 0 && plSlot[i] >= 0) ? basePl[i] + plSlot[i] : -1;
          const uint32_t slotNB = (uint32_t)baseH[i] + hNB[i], slotB = (uint32_t)baseH[i] + hB[i];
          MlRec r;
          r.nb = sR ? cs[3 * i + 1] : NEG;
          r.b = sB ? cs[3 * i] : NEG;
          r.info = (uint32_t)last[i] | ((uint32_t)(pln + 1) << 8);
          r.hyps = (sR ? slotNB : kMlNoHyp) | ((sB ? slotB : kMlNoHyp) << 16);
          r.sid = me[i].sid;
          r.spar = me[i].spar;
          S.rec[q][srcNew] = r;
          if constexpr (TL) {
            S.ctx[q][srcNew] = ctxv[i];
            S.tlIn[q][srcNew] = lIn[i];
          }
          if (!TL && mk[i]) {
            atomOr64(&S.mask[q][srcNew], mk[i]);
          }
          if (pln >= 0) {
            atomOr64(&S.cmask[q][pln], 1ull << last[i]);
          }
          if (sR) {
            histPT[hrow + slotNB] = make_int2((int)parR[i], last[i]);
          }
          if (sB) {
            histPT[hrow + slotB] = make_int2((int)hypM[i], blank);
          }
        }
        if ((selMask[3 * i + 2] >> lane) & 1ull) {
          if constexpr (TL) {
            if (gatheredS) {
              gatherPut(before - offW + wavePrefixCount(selMask[3 * i + 2]), cs[3 * i + 2], last[i], hypB[i], me[i].sid, srcNew,
                        (uint32_t)lmLast[TL ? i : 0].y, __uint_as_float((uint32_t)lmLast[TL ? i : 0].x));
            } else {
              newState(before + wavePrefixCount(selMask[3 * i + 2]), cs[3 * i + 2], last[i], hypB[i], me[i].sid, 0ull, srcNew,
                       (uint32_t)lmLast[TL ? i : 0].y, __uint_as_float((uint32_t)lmLast[TL ? i : 0].x));
            }
          } else {
            newState(before + wavePrefixCount(selMask[3 * i + 2]), cs[3 * i + 2], last[i], hypB[i], me[i].sid, mk[i], srcNew, 0u, 0.0f);
          }
        }
        before += popc64(selMask[3 * i + 2]);
      }
      if constexpr (TL) {
        if (gatheredS) {
          gatherBuild(offW, nNewWave);
        }
      }
    }
    nState = nSurv + nNew;
    endBest = best;
    ldsBarrier(); /* 3 */
  };
  auto frames = [&](auto RL) {
    int t = 0;
    for (; t + 1 < T && !dead; t += 2) {
      frameStep(SlParity<0>(), RL, t);
      if (dead) {
        break;
      }
      frameStep(SlParity<1>(), RL, t + 1);
    }
    if (!dead && t < T) {
      frameStep(SlParity<0>(), RL, t);
    }
  };
  if (isSvcW) {
    frames(SlParity<2>());
  } else if (!isTokW) {
    frames(SlParity<1>());
  } else {
    frames(SlParity<0>());
  }

  /* ---- decodeEnd (LexiconFreeDecoder.cpp:127-158): finish() keeps the state, token = sil; the two hypotheses of
   * a state merge; sorted n-best (candidatesStore returnSorted).  Up to 64 * NG states: keys through LDS. ---- */
  const int pe = T & 1;
  const int ff = T + 1;
  if (!dead) {
    unsigned long long* keyTab = S.cmask[pe ^ 1]; /* (free: the next build never runs) */
    uint32_t* hpTab = (uint32_t*)S.newLane;
    if (LA && !TL) {
      endBest = f64FromKey(S.mmaxKey[pe]);
    }
    /* TL: lm->finish(state) (KenLM.cpp:77-83: the score of </s> in the state's context): both hypotheses of a state add
     * the same term; the best candidate of decodeEnd is a maximum again */
    auto finishOf = [&](int l, bool lv) {
      return lmW * (double)__uint_as_float((uint32_t)tokLm[(size_t)(lv ? S.ctx[pe][l] : 0u) * (size_t)tokStride + (size_t)N].x);
    };
    if constexpr (TL) {
      if (tid == 0) {
        S.fbest[0] = 0ull;
      }
      ldsBarrier();
      for (int l = tid; l < nState; l += W) {
        const MlRec me = S.rec[pe][l];
        const bool wB = me.b > me.nb;
        const double mm = (wB ? me.b : me.nb) + finishOf(l, true);
        const uint32_t hp = wB ? (me.hyps >> 16) : (me.hyps & 0xFFFFu);
        if (hp != kMlNoHyp && mm == mm) {
          atomMax64(&S.fbest[0], f64Key(mm));
        }
      }
      ldsBarrier();
      const unsigned long long k = S.fbest[0];
      endBest = k != 0ull ? f64FromKey(k) : 0.0;
    }
    const double thr = endBest - P.beamThreshold;
    for (int base = 0; base < LN; base += W) {
      const int l = base + tid;
      if (l < LN) {
        const bool lv = l < nState;
        const MlRec me = S.rec[pe][lv ? l : 0];
        const double nb = lv ? me.nb : NEG, bb = lv ? me.b : NEG;
        const bool wB = bb > nb;
        double mm = wB ? bb : nb;
        if constexpr (TL) {
          mm = mm + finishOf(l, lv);
        }
        const uint32_t hp = wB ? (me.hyps >> 16) : (me.hyps & 0xFFFFu);
        const bool ok = lv && mm >= thr && (!TL || hp != kMlNoHyp);
        if (LA && ok) {
          const double lo = TL ? (wB ? nb : bb) + finishOf(l, lv) : (wB ? nb : bb);
          const uint32_t hl = wB ? (me.hyps & 0xFFFFu) : (me.hyps >> 16);
          if (hl != kMlNoHyp && lo >= thr) {
            mm = slLogAdd(mm, lo);
          }
        }
        keyTab[l] = ok ? f64Key(mm) : 0ull;
        hpTab[l] = hp;
      }
    }
    ldsBarrier();
    for (int base = 0; base < LN; base += W) {
      const int l = base + tid;
      if (l < nState) {
        const unsigned long long key = keyTab[l];
        const uint32_t hp = hpTab[l];
        if (key != 0ull) {
          int rank = 0;
          for (int i = 0; i < nState; ++i) {
            const unsigned long long k2 = keyTab[i];
            const uint32_t h2 = hpTab[i];
            rank += (k2 > key || (k2 == key && h2 < hp)) ? 1 : 0;
          }
          const size_t g = ((size_t)b * K + rank) * 3;
          P.outScores[g + 0] = f64FromKey(key);
          P.outScores[g + 1] = 0.0; /* emitting-model score: the back-trace kernel fills it in */
          P.outScores[g + 2] = 0.0; /* ZeroLM; a token LM's score is re-accumulated by the back-trace kernel as well */
          P.histPT[hbase + (int64_t)ff * K + rank] = make_int2((int)hp, P.sil);
          atomAdd32(&S.scal[ML_NOUT], 1u);
        }
      }
    }
    ldsBarrier();
    if (tid == 0) {
      const int n = (int)S.scal[ML_NOUT];
      P.outN[b] = n;
      P.uttNBeam[b] = n;
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
}
