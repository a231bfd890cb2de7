/*
 * fltx_slane.h -- "lane = LM state" decode of a whole utterance for the headline
 * configuration: LexiconFreeDecoder + ZeroLM, max-merge, beam <= 64, <= 64
 * tokens, offline (decodeBegin + all frames + decodeEnd in one launch).
 * Included by fltx_kernels.h.  Same candidates, same merge groups, same
 * selection as LexiconFreeDecoder::decodeStep (LexiconFreeDecoder.cpp:30-125)
 * with candidatesStore (Utils.h:146-225): bit-identical n-best.
 *
 * A wave of this chip issues one vector instruction every ~7 clocks (10 with a
 * second wave on its SIMD; tools/microbench2.hip), a barrier of eight waves
 * costs ~140: the frame step is priced in instructions per wave, and this file
 * is organised around issuing few of them.
 *
 *   * A lane holds an LM STATE, not a hypothesis: the two hypotheses a state
 *     can have in the beam -- (S, last token, prevBlank = false) and
 *     (S, blank, prevBlank = true), LexiconFreeDecoder.h:68-78 -- are the two
 *     scores `nb` / `b` of one 32-byte record.  The merge of Utils.h:167-198
 *     then has a fixed shape: every extension of S by a new token n reaches
 *     state S.n from max(nb, b) (one candidate per (lane, token), no partner
 *     look-up); S's blank extension and S's repeat are the lane's own; the
 *     only cross-lane member of any group is the parent state's extension by
 *     last(S), which joins S's repeat -- one parent link per lane, kept up to
 *     date by the build step.
 *   * With ZeroLM the frame's best candidate is best hypothesis + best token,
 *     and the best hypothesis of frame t + 1 IS the best candidate of frame t:
 *     best(t) = fl(best(t-1) + emax(t)) is a scalar recurrence, computed by the
 *     wave that stages the emission rows a frame ahead.  The beam therefore
 *     need not be sorted, and no frame ranks its survivors: a candidate
 *     survives iff its histogram bin is better than the bin of the K-th best;
 *     only the members of that one bin are compared with each other (and only
 *     when not all of them survive).
 *   * The histogram is a window of 512 bins over the float bits of
 *     (best - score), 128 bins per octave, centred on where the K-th best was
 *     a frame ago.  Candidates beyond the window are not even counted unless
 *     the window turns out to hold fewer than K.
 *   * No short-list, no scatter: after the histogram scan every lane knows
 *     which of its own candidates survived; new lanes are handed out by a
 *     per-wave count (one LDS add per wave) and every survivor's record,
 *     history entry and parent-mask update is written by the lane that
 *     evaluated it.
 *   * The emitting-model score is not carried through the frames at all (it
 *     takes part in no decision): the back-trace kernel re-accumulates it
 *     along each returned path in the reference's order.
 *   * An LM state's identity is the index of the history record of the
 *     hypothesis that first entered it; that record also stores the parent
 *     state's id and the token, so the history rows double as the memo of
 *     LMState::child (lm/LM.h:24-34): nothing else is written per frame.  The
 *     rare re-entry of a state that had dropped out of the beam (about one
 *     frame in a hundred on the benchmark inputs) scans the rows.
 *
 * Three barriers per frame.
 */
#pragma once
#include <type_traits>

constexpr int kSlNB = 256;          /* histogram bins: 4 per lane of the scan */
constexpr int kSlMid = kSlNB / 2;   /* where the window puts the last frame's K-th best */
constexpr int kSlFar = kSlNB - 1;   /* beyond the window: counted only on demand */
constexpr int kSlInvalid = 1023;    /* not a candidate */
constexpr int kSlList = 96;         /* list positions addressable by the token waves (GT x waves) */
constexpr int kSlBCap = 128;        /* boundary-bin members compared pairwise */
constexpr int kSlFineShift = 16;    /* 128 bins per octave of (best - score): the window spans two octaves */
constexpr int kSlCoarseShift = 19;  /* 16 bins per octave: 16 octaves in 256 bins */
constexpr int kSlCoarseBase = 120 << 4; /* float bits of 2^-7, >> kSlCoarseShift */
constexpr uint32_t kSlNoHyp = 0xFFu;
constexpr uint32_t kSlNewFlag = 0x100u; /* history record: the hypothesis entered a new LM state */

struct SlRec { /* one LM state of the beam, 32 B */
  double nb;     /* score of (S, last, prevBlank = false), -inf if absent */
  double b;      /* score of (S, blank, prevBlank = true), -inf if absent */
  uint32_t info; /* last token | (parent lane + 1) << 8 | history slot of nb << 16 | of b << 24 */
  uint32_t sid;  /* state id = history index (row * K + slot) of the hypothesis that entered it */
  uint32_t spar; /* id of the parent state */
  uint32_t pad;  /* token-level n-gram LM (TL): the state's n-gram context = its row of DecodeParams::tokLm */
};

struct SlRow { /* what a frame needs to know about its emission row, 48 B */
  double best;    /* the frame's best candidate (Utils.h:131-137) */
  double thr;     /* best - beamThreshold */
  int32_t nList;  /* tokens the normal waves evaluate (allowed, not blank) */
  int32_t silPos; /* list position of sil, or a value no wave matches */
  uint32_t dead;  /* no candidate at all / not finite: the utterance goes to the general engines */
  uint32_t nev;   /* re-entry events recorded by the build of the previous frame */
  unsigned long long allow; /* token beam (LexiconFreeDecoder.cpp:42-51): bit n = token n is evaluated */
  uint32_t ekey;  /* order key of the largest allowed emission other than sil's, 0 = none (logAdd: best = beam's best + this) */
  float esil;     /* sil's emission */
};

template <int V>
struct SlParity {
  static constexpr int value = V;
};

struct SlaneLds {
  SlRec rec[2][64];
  unsigned long long cmask[2][64]; /* tokens whose child state is in the beam and linked to this lane */
  unsigned long long mask[2][64];  /* tokens whose child state was ever materialised */
  uint32_t hist[2][kSlNB];
  double eAll[2][64];              /* emission row, widened */
  double eTok[2][kSlList];         /* ... of the listed tokens, by list position; NaN past the list */
  unsigned long long tokBit[2][kSlList]; /* 1 << token of the list position, 0 past the list */
  SlRow row[2];
  uint8_t tokId[2][kSlList];       /* list position -> token */
  uint32_t off[32];                /* new states of the waves before wave i; [self wave + 1] = all */
  int32_t newLane[64];             /* old lane -> lane in the next beam, -1 = dropped */
  uint32_t scal[16];
  unsigned long long bKey[kSlBCap];
  uint32_t bOrd[kSlBCap];
  uint32_t evLane[64], evSpar[64], evTok[64];
  unsigned long long scanMask;
  unsigned long long mmaxKey[2]; /* logAdd: order key of the best hypothesis of the beam a frame starts from */
  uint32_t scanMin, pad0;
  float raw[3][64]; /* emission rows on their way in: row r lands in raw[r % 3] two frames before it is staged */
  /* token-level n-gram LM (TL): the frame's best candidate is a maximum over the candidates (no recurrence: the LM term
   * differs per (state, token)), published per parity; and per lane the LM score its state was ENTERED with -- the
   * parent state's extension by last(S) joins S's repeat group with exactly that term */
  unsigned long long fbest[2];
  float tlIn[2][64];
  /* Last members, streams only (ST): an offline launch takes offsetof(SlaneLds, amNB) bytes -- 16 KB, so that its
   * workgroup and a back-trace workgroup of the batch before (140 KB) fit a CU together */
  double amNB[2][64], amB[2][64]; /* emitting-model score of a state's two hypotheses */
  uint8_t rsNB[64], rsB[64];      /* restore: the parked slot of a lane's two hypotheses */
};
enum { SL_NSURV = 0, SL_NHSURV = 1, SL_BCNT = 2, SL_NEXTID = 3, SL_STATUS = 4, SL_NSTATE = 5 };

/* Token-LM variant (TL): with LM terms the beam turns over faster -- on the benchmark's `ctc` inputs a state that had
 * dropped out is entered again in three frames of ten (one in a hundred under ZeroLM), and a scan of the history rows
 * per re-entry was 90 % of the kernel (24.9 ms per 256 x 1 000 frames).  LMState::child's memo (lm/LM.h:24-34) is
 * therefore ALSO kept where it is cheap to ask, exact or absent, the rows staying the authority behind it:
 *   edge[]  : the newest (parent state id, token) -> state id edges, direct-mapped (the slot holds the whole key: a hit
 *             is exact); written at every creation of a state;
 *   mmTag / mmMask : the child mask a state had when it last dropped out of the beam (nothing can add an edge to a state
 *             that is not in the beam), direct-mapped by state id; written one frame after the drop by the first token
 *             wave, together with the edges the state's last extension created (`dmask`).
 * A miss in either sends that one event to the rows (tlReenter). */
constexpr uint32_t kTlNoSid = 0xFFFFFFFFu;
constexpr int kTlGather = 12; /* survivors a token wave of the stream variant builds in one pass (more: position by position) */
struct TlaneLds : SlaneLds {
  unsigned long long dmask[64];          /* old lane -> tokens whose child state this frame's build created from it */
  uint32_t evSid[64];                    /* re-entry event: the state's id when the edge memo had it, else kTlNoSid */
  uint32_t evNeed[64];                   /* ... bit 0: look the id up in the rows, bit 1: the child mask */
  double lmNB[2][64], lmB[2][64];        /* streams (ST): the LM score of a state's two hypotheses (getBestHypothesis reports an ancestor's) */
  /* the memos, sized by the host (DecodeParams::tlEdgeSlots / tlMaskSlots, powers of two: larger when one workgroup
   * has the CU's LDS to itself): edge[E] (bit 63 | parent id:23 << 37 | token:14 << 23 | state id:23), mmMask[M],
   * mmTag[M] (state id, kTlNoSid = empty) -- or, in the stream variant (whose ids come from the table in HBM and which
   * has no memo), the token waves' scratch for a gathered build: kTlGatherBytes per wave */
  alignas(16) unsigned long long memo[1];
};
constexpr int kTlGatherBytes = kTlGather * 48; /* c, am, lm (doubles), token | hyp | parent's new lane, parent id, ctx row, lm step */
struct TlMemo {
  unsigned long long* edge;
  unsigned long long* mmMask;
  uint32_t* mmTag;
  uint32_t eMask, mMask;
};
FLTX_DEV TlMemo tlMemoOf(TlaneLds& S, int edgeSlots, int maskSlots) {
  TlMemo m;
  m.edge = S.memo;
  m.mmMask = S.memo + edgeSlots;
  m.mmTag = (uint32_t*)(S.memo + edgeSlots + maskSlots);
  m.eMask = (uint32_t)(edgeSlots - 1);
  m.mMask = (uint32_t)(maskSlots - 1);
  return m;
}
FLTX_DEV uint32_t tlHash(uint32_t a, uint32_t b) {
  uint32_t h = (a * 0x9E3779B1u) ^ (b * 0x85EBCA77u) ^ 0x5bd1e995u;
  h ^= h >> 15;
  h *= 0x2C1B3C6Du;
  h ^= h >> 12;
  return h;
}
FLTX_DEV unsigned long long tlEdgePack(uint32_t psid, uint32_t tok, uint32_t csid) {
  return (1ull << 63) | ((unsigned long long)(psid & 0x7FFFFFu) << 37) | ((unsigned long long)(tok & 0x3FFFu) << 23) |
         (unsigned long long)(csid & 0x7FFFFFu);
}
FLTX_DEV uint32_t tlEdgeSlot(const TlMemo& M, uint32_t psid, uint32_t tok) { return tlHash(psid, tok) & M.eMask; }
FLTX_DEV uint32_t tlMaskSlot(const TlMemo& M, uint32_t sid) { return tlHash(sid, 0x51u) & M.mMask; }

FLTX_DEV double slNegInf() { return -__builtin_huge_val(); }

/* bin of a candidate `c` below the frame's best: a window of 512 bins over the
 * float bit pattern of best - c (monotone in the score, and any monotone binning
 * gives exact selection); everything nearer than the window shares bin 0,
 * everything farther bin 511 */
/* above: with logAdd a merged candidate can score above the frame's best raw candidate; it belongs to the
 * nearest bin (the float bits of a negative difference would put it beyond the window) */
template <bool ABOVE = false>
FLTX_DEV int slBin(double best, double c, int shift, int base) {
  float d = (float)(best - c);
  if (ABOVE) {
    d = d > 0.0f ? d : 0.0f;
  }
  int q = (int)(__float_as_uint(d) >> shift) - base;
  q = q < 0 ? 0 : q;
  return q > kSlNB - 1 ? kSlNB - 1 : q;
}

struct SlScan {
  int bstar; /* bin holding the K-th best candidate */
  int cum;   /* candidates in better bins */
  int cnt;   /* candidates in bin bstar */
  int total;
  bool crossed; /* the counted bins hold at least K */
};
/* every wave scans the counts itself (4 bins per lane, DPP prefix) */
/* noFar: leave the last bin out.  Nobody counts there before the first scan, but a wave that finds
 * fewer than K inside the window adds its far candidates right after its own scan: a wave that
 * scans later must not see them, or the two would disagree about which barriers follow. */
FLTX_DEV SlScan slScan(const uint32_t* hist, int K, bool noFar) {
  const int lane = laneId();
  uint4 c0 = ((const uint4*)hist)[lane];
  if (noFar && lane == 63) {
    c0.w = 0u;
  }
  const int mine = (int)(c0.x + c0.y + c0.z + c0.w);
  const int inc = waveInclusiveScan(mine);
  int pre[5];
  pre[0] = inc - mine;
  pre[1] = pre[0] + (int)c0.x;
  pre[2] = pre[1] + (int)c0.y;
  pre[3] = pre[2] + (int)c0.z;
  pre[4] = inc;
  SlScan r;
  r.total = (int)waveReadLane32((uint32_t)inc, 63);
  const unsigned long long cm = waveBallot(pre[0] < K && inc >= K);
  int q = 3, before = pre[3], cq = pre[4] - pre[3];
#pragma unroll
  for (int i = 2; i >= 0; --i) {
    const bool hit = pre[i + 1] >= K;
    q = hit ? i : q;
    before = hit ? pre[i] : before;
    cq = hit ? pre[i + 1] - pre[i] : cq;
  }
  const int X = cm ? __builtin_ctzll(cm) : 0;
  const uint32_t a = waveReadLane32((uint32_t)(4 * lane + q) | ((uint32_t)cq << 16), X);
  r.cum = (int)waveReadLane32((uint32_t)before, X);
  r.bstar = (int)(a & 0xFFFFu);
  r.cnt = (int)(a >> 16);
  r.crossed = cm != 0ull;
  if (!r.crossed) { /* fewer than K counted */
    r.bstar = kSlNB - 1;
    r.cum = 0;
    r.cnt = r.total;
  }
  return r;
}

/* The members of the K-th best's bin ranked against each other (S.bKey / S.bOrd hold them, `cnt` <= kSlBCap = 128 of
 * them, published before the barrier that precedes this): every lane holds one (two) entries of the list, a wave's
 * members are ranked one after the other by broadcasting the member and counting the lanes whose entry beats it -- two
 * ballots per member and no LDS round trip.  (Until round 6 each member looped over the list, one dependent LDS read per
 * entry; with LM terms the bin holds 13 members on average and the wave that owns most of them kept everybody waiting:
 * the token-LM kernel went from 4.15 to 3.38 ms with this.)  member(j) / keyOf(j): is this lane's candidate j in the
 * bin, its order key.  Returns this lane's `take` bits: candidate j is among the `need` best of the bin. */
template <int NJ, typename MemberFn, typename KeyFn>
FLTX_DEV uint32_t slRankBin(const unsigned long long* bKey, const uint32_t* bOrd, int cnt, int need, int wave,
                            MemberFn member, KeyFn keyOf) {
  const int lane = laneId();
  const unsigned long long e0 = lane < cnt ? bKey[lane] : 0ull;
  const uint32_t o0 = lane < cnt ? bOrd[lane] : 0xFFFFFFFFu;
  const unsigned long long e1 = lane + 64 < cnt ? bKey[lane + 64] : 0ull;
  const uint32_t o1 = lane + 64 < cnt ? bOrd[lane + 64] : 0xFFFFFFFFu;
  uint32_t take = 0u;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    unsigned long long mem = waveBallot(member(j));
    if (mem == 0ull) {
      continue;
    }
    const unsigned long long kMine = keyOf(j);
    while (mem) {
      /* (said to be uniform in so many words: the loop must be a scalar one, with every lane in it) */
      const int L = waveUniform(__builtin_ctzll(mem));
      mem &= ~(1ull << L);
      const unsigned long long k = ((unsigned long long)waveReadLane32((uint32_t)(kMine >> 32), L) << 32) |
                                   waveReadLane32((uint32_t)kMine, L);
      const uint32_t o = ((uint32_t)wave << 16) | ((uint32_t)j << 8) | (uint32_t)L;
      const int rank = popc64(waveBallot(lane < cnt && (e0 > k || (e0 == k && o0 < o)))) +
                       popc64(waveBallot(lane + 64 < cnt && (e1 > k || (e1 == k && o1 < o))));
      if (lane == L && rank < need) {
        take |= 1u << j;
      }
    }
  }
  return take;
}

/* The token beam of a row of N <= 32 emissions (LexiconDecoder.cpp:42-51: the beamSizeToken largest,
 * ties to the lower index), all pairs compared with 16 in-row rotations: the four rows of 16
 * lanes take (tokens 0-15 among themselves), (16-31 among themselves), (0-15 against 16-31) and
 * (16-31 against 0-15).  Keys are made unique by the index, so one 64-bit compare orders a pair. */
struct XlRank {
  unsigned long long mine, src;
  int part;
};
FLTX_DEV XlRank xlRankBegin(float v, int N) {
  const int lane = laneId();
  const int tok = lane & 31;
  const float vt = __uint_as_float(waveGather32(__float_as_uint(v + 0.0f), tok)); /* (-0 -> +0: equal as floats) */
  XlRank r;
  r.mine = tok < N ? (((unsigned long long)f32Key(vt) << 6) | (unsigned long long)(63 - tok)) : 0ull;
  const unsigned long long other = waveShfl64(r.mine, lane ^ 16);
  r.src = lane < 32 ? r.mine : other;
  r.part = 0;
  return r;
}
template <int R0, int R1>
FLTX_DEV void xlRankRange(XlRank& r) {
  if constexpr (R0 < R1) {
    r.part += waveRowRor64<R0>(r.src) > r.mine ? 1 : 0;
    xlRankRange<R0 + 1, R1>(r);
  }
}
FLTX_DEV unsigned long long xlRankEnd(const XlRank& r, int N, int Kt) {
  const int lane = laneId();
  const int tot = r.part + (int)waveGather32((uint32_t)r.part, lane ^ 32);
  return waveBallot(lane < N && lane < 32 && tot < Kt);
}

/* Emission row -> what the frame step reads: the widened row, the token beam, the
 * list of tokens the token waves evaluate, and the frame's best candidate: best
 * hypothesis (= `mmax`, the previous frame's best candidate) + best token.  One
 * wave; lane n holds e[n].  In two halves so that the staging wave can spread the
 * work over a frame: slRowScan (registers only) and slRowStore (LDS writes). */
struct SlRowRegs {
  unsigned long long allow, listMask;
  double best;
  float v, esil;
  uint32_t ekey;
  int nList;
  bool dead;
};
/* (silScore: handed over by callers that keep the option in a register -- read through P it is a scalar load from the
 * kernel arguments, and a full wait, in every frame) */
FLTX_DEV SlRowRegs slRowScan(const DecodeParams& P, float v, bool ctc, double mmax, double silScore) {
  const int lane = laneId();
  const int N = P.N;
  const bool inRow = lane < N;
  SlRowRegs r;
  r.v = v;
  r.allow = N >= 64 ? ~0ull : ((1ull << N) - 1ull);
  if (P.Kt < N) { /* LexiconFreeDecoder.cpp:42-51: top beamSizeToken by emission, ties to the lower index */
    if (N <= 32 && waveBallot(inRow && !(v == v)) == 0ull) { /* all pairs by 16 row rotations (C2 with a token beam of 10: 3.19 -> ms) */
      XlRank rs = xlRankBegin(v, N);
      xlRankRange<0, 16>(rs);
      r.allow = xlRankEnd(rs, N, P.Kt);
    } else {
      int rank = 0;
      for (int m = 0; m < N; ++m) {
        const float o = __uint_as_float(waveReadLane32(__float_as_uint(v), m));
        rank += (o > v || (o == v && m < lane)) ? 1 : 0;
      }
      r.allow = waveBallot(inRow && rank < P.Kt);
    }
  }
  const bool mine = inRow && ((r.allow >> lane) & 1ull) != 0ull;
  const uint32_t ek = waveMax32((mine && lane != P.sil && v == v) ? f32Key(v) : 0u);
  r.listMask = r.allow;
  if (ctc) {
    r.listMask &= ~(1ull << P.blank);
  }
  r.nList = popc64(r.listMask);
  const float eSil = __uint_as_float(waveReadLane32(__float_as_uint(v), P.sil));
  /* best candidate of the frame: fl(a + e) is monotone in e, so the best hypothesis with the
   * largest emission -- sil is scored separately because of silScore */
  double best = 0.0;
  bool any = false;
  if (ek != 0u) {
    best = mmax + (double)f32FromKey(ek);
    any = true;
  }
  if ((r.allow >> P.sil) & 1ull) {
    const double sS = (mmax + (double)eSil) + silScore;
    if (sS == sS && (!any || sS > best)) {
      best = sS;
      any = true;
    }
  }
  r.best = best;
  r.dead = !any || !(best - best == 0.0);
  r.ekey = ek;
  r.esil = eSil;
  return r;
}
FLTX_DEV SlRowRegs slRowScan(const DecodeParams& P, float v, bool ctc, double mmax) {
  return slRowScan(P, v, ctc, mmax, P.silScore);
}
/* padPast: also (re)write the positions past the end of the list (NaN emission = no candidate): 2 = all of them
 * (the prologue), 1 = those a list can reach (a token beam keeps min(Kt, N) tokens, of which blank is not listed
 * by the lexicon-free engine: the list is one shorter in the frames whose token beam holds blank), 0 = none (the
 * list has the same length in every frame) */
template <typename LDS>
FLTX_DEV void slRowStore(const DecodeParams& P, LDS& S, int q, const SlRowRegs& r, int padPast) {
  const int lane = laneId();
  const bool inRow = lane < P.N;
  const int pos = wavePrefixCount(r.listMask);
  const bool listed = inRow && ((r.listMask >> lane) & 1ull) != 0ull;
  if (inRow) {
    S.eAll[q][lane] = (double)r.v;
  }
  if (padPast == 2) {
    for (int i = lane; i < kSlList; i += 64) {
      if (i >= r.nList) {
        S.eTok[q][i] = __builtin_nan("");
        S.tokBit[q][i] = 0ull;
        S.tokId[q][i] = (uint8_t)0;
      }
    }
  } else if (padPast == 1) {
    const int reach = P.Kt < P.N ? P.Kt : P.N;
    if (lane >= r.nList && lane < reach) {
      S.eTok[q][lane] = __builtin_nan("");
      S.tokBit[q][lane] = 0ull;
      S.tokId[q][lane] = (uint8_t)0;
    }
  }
  if (listed) {
    S.tokId[q][pos] = (uint8_t)lane;
    S.eTok[q][pos] = (double)r.v;
    S.tokBit[q][pos] = 1ull << lane;
  }
  if (lane == 0) {
    S.row[q].best = r.best;
    S.row[q].thr = r.best - P.beamThreshold;
    S.row[q].nList = r.nList;
    S.row[q].silPos = ((r.listMask >> P.sil) & 1ull) ? popc64(r.listMask & ((1ull << P.sil) - 1ull)) : -4096;
    S.row[q].dead = r.dead ? 1u : 0u;
    S.row[q].allow = r.allow;
    S.row[q].ekey = r.ekey;
    S.row[q].esil = r.esil;
  }
}

/* Re-entry of LM states that had dropped out of the beam (recorded by the
 * build step: a new state whose (parent, token) edge had been materialised
 * before).  The history rows are the memo of LMState::child (lm/LM.h:24-34):
 * the earliest record {new state, parent id, token} names the state; records
 * whose parent id is that state give back its child mask; lanes whose parent id
 * it is get their link back.  All waves; rare. */
FLTX_DEV __attribute__((noinline)) void slReenter(SlaneLds& S, const int2* histPT, int q, int nState, int64_t hbase,
                                                   int64_t nRec) {
  /* (takes no DecodeParams: passing its address to a call would move the kernel's copy into
   * scratch memory, and the frame loop would read every option from there) */
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
      if ((x & kSlNewFlag) && y == n && (x >> 9) == ps) {
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
        if ((x & kSlNewFlag) && (x >> 9) == sid) {
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
      if (tid < nState && tid != X && S.rec[q][tid].spar == sid) { /* orphans get their parent back */
        const uint32_t info = S.rec[q][tid].info;
        S.rec[q][tid].info = (info & ~0xFF00u) | ((uint32_t)(X + 1) << 8);
        atomOr64(&S.cmask[q][X], 1ull << (info & 0xFFu));
      }
    }
    ldsBarrier();
  }
  if (tid == 0) {
    S.row[q].nev = 0u;
  }
  ldsBarrier();
}

/* Streams (ST): a state's id comes from the (parent id, token) -> id table in HBM (DecodeParams::childTab, the
 * lane-per-slot engine's: the parked beam, the tables and the history rows keep that engine's format, so begin,
 * end, prune and getBestHypothesis are its code).  A state that is entered again (its parent's mask had the
 * token) gets its child mask back from maskTab and the lanes whose parent it is get their link back. */
FLTX_DEV __attribute__((noinline)) void slRelink(SlaneLds& S, const unsigned long long* maskTab, int q, int nState) {
  const int tid = (int)threadIdx.x;
  const int nev = (int)S.row[q].nev;
#ifndef FLTX_EMU
  __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  ldsBarrier();
  for (int e = 0; e < nev; ++e) {
    const int X = (int)S.evLane[e];
    const uint32_t sid = S.rec[q][X].sid;
    if (tid == 0) {
      S.mask[q][X] |= loadCoherent64(&maskTab[sid]);
    }
    if (tid < nState && tid != X && S.rec[q][tid].spar == sid) { /* orphans get their parent back */
      const uint32_t info = S.rec[q][tid].info;
      S.rec[q][tid].info = (info & ~0xFF00u) | ((uint32_t)(X + 1) << 8);
      atomOr64(&S.cmask[q][X], 1ull << (info & 0xFFu));
    }
    ldsBarrier();
  }
  if (tid == 0) {
    S.row[q].nev = 0u;
  }
  ldsBarrier();
}

/* Stream chunks of the token-LM variant (ST + TL): a stream's begin / prune / getBestHypothesis / end are the generic
 * engine's kernels (they know n-gram LMs), so state ids come from ITS table -- open addressing over (epoch, parent id,
 * edge) in HBM with the id beside the key (stateVal; ids are recycled by fltx_compact_states_kernel) -- as
 * stateChildIds() hands them out.  The lexicon-free decoder asks about a (state, token) pair at most once per frame
 * (one candidate per lane and token), so nobody waits for anybody: found -> read the id, else install the key and
 * name the state from the stream's id counter.  `fresh` = false is LMState::child's memo saying "entered before". */
FLTX_DEV uint32_t tlStreamChild(const DecodeParams& P, int b, uint32_t par, int32_t edge, uint32_t born, uint32_t* nextId,
                                uint32_t* status, bool& fresh) {
  unsigned long long* tab = P.stateTab + (size_t)b * P.stateCap;
  uint32_t* val = P.stateVal + (size_t)b * P.stateCap;
  const unsigned long long key = ((unsigned long long)P.epoch << 48) |
      ((unsigned long long)(par & 0xFFFFFFu) << 24) | (unsigned long long)((uint32_t)(edge + 1) & 0xFFFFFFu);
  const uint32_t mask = P.stateCap - 1;
  uint32_t s = hashKey(par, (uint32_t)edge, 0x9747b28cu, 0) & mask;
  fresh = false;
  for (uint32_t probes = 0; probes < P.stateCap; ++probes) {
    unsigned long long cur = loadCoherent64(&tab[s]);
    for (;;) {
      if (cur == key) {
        return loadCoherent32(&val[s]);
      }
      if ((cur >> 48) == (unsigned long long)P.epoch) {
        break; /* live entry of another state: next slot */
      }
      const unsigned long long old = atomCas64(&tab[s], cur, key);
      if (old == cur) {
        fresh = true;
        const uint32_t id = allocStateId(P, b, atomAdd32(nextId, 1u), par, edge, born, status);
        storeCoherent32(&val[s], id);
        return id;
      }
      cur = old;
    }
    s = (s + 1) & mask;
  }
  atomOr32(status, ST_TABLE_FULL);
  return 0u;
}

/* Re-entry with the memos of TlaneLds: what they know is applied by one thread per event, what they do not know is
 * looked up in the history rows as slReenter does (per event, rare), then every lane looks for its parent among the
 * re-entered states at once. */
FLTX_DEV __attribute__((noinline)) void tlReenter(TlaneLds& S, const int2* histPT, int q, int nState, int64_t hbase,
                                                   int64_t nRec, int edgeSlots, int maskSlots) {
  const TlMemo M = tlMemoOf(S, edgeSlots, maskSlots);
  const int tid = (int)threadIdx.x, W = (int)blockDim.x;
  const int nev = (int)(S.row[q].nev & 0xFFFFu); /* (the upper half: how many of them the memos do not answer) */
  ldsBarrier(); /* (everybody has the count; the masks saved at the head of this frame are visible) */
  if (tid < nev) {
    const int X = (int)S.evLane[tid];
    const uint32_t sid = S.evSid[tid];
    uint32_t need = 3u;
    if (sid != kTlNoSid) {
      const uint32_t ms = tlMaskSlot(M, sid);
      need = 2u;
      if (M.mmTag[ms] == sid) {
        S.mask[q][X] |= M.mmMask[ms]; /* (one event per lane: nobody else writes this word here) */
        need = 0u;
      }
    }
    S.evNeed[tid] = need;
  }
  ldsBarrier();
  bool settled = false; /* this wave's history stores of the last build have reached the L2 (the look-ups read them) */
  for (int e = 0; e < nev; ++e) {
    const uint32_t need = S.evNeed[e]; /* (uniform) */
    if (need == 0u) {
      continue;
    }
    if (!settled) {
      settled = true;
#ifndef FLTX_EMU
      __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
      ldsBarrier();
    }
    const int X = (int)S.evLane[e];
    const uint32_t ps = S.evSpar[e], n = S.evTok[e];
    const unsigned long long* h = (const unsigned long long*)(histPT + hbase);
    uint32_t sid = S.evSid[e];
    bool wantKids = true;
    if (need & 1u) {
      if (tid == 0) {
        S.scanMin = 0xFFFFFFFFu;
      }
      ldsBarrier();
      uint32_t found = 0xFFFFFFFFu;
      for (int64_t i0 = tid; i0 < nRec; i0 += 8 * (int64_t)W) { /* eight loads in flight */
        unsigned long long rr[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int64_t i = i0 + (int64_t)u * W;
          rr[u] = i < nRec ? loadCoherent64(h + i) : 0ull;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int64_t i = i0 + (int64_t)u * W;
          const uint32_t x = (uint32_t)rr[u], y = (uint32_t)(rr[u] >> 32);
          if (i < nRec && (x & kSlNewFlag) && y == n && (x >> 9) == ps) {
            found = found < (uint32_t)i ? found : (uint32_t)i;
          }
        }
      }
      if (found != 0xFFFFFFFFu) {
        atomMin32(&S.scanMin, found);
      }
      ldsBarrier();
      sid = S.scanMin; /* (at least the record the last build wrote) */
      const uint32_t ms = tlMaskSlot(M, sid);
      wantKids = M.mmTag[ms] != sid;
      ldsBarrier(); /* (everybody has read scanMin and the tag before anything below changes) */
      if (tid == 0) {
        S.rec[q][X].sid = sid;
        S.evSid[e] = sid;
        M.edge[tlEdgeSlot(M, ps, n)] = tlEdgePack(ps, n, sid); /* (the memo learns the edge again) */
        if (!wantKids) {
          S.mask[q][X] |= M.mmMask[ms];
        }
      }
    }
    if (wantKids) {
      if (tid == 0) {
        S.scanMask = 0ull;
      }
      ldsBarrier();
      unsigned long long kids = 0ull;
      for (int64_t i0 = tid; i0 < nRec; i0 += 8 * (int64_t)W) {
        unsigned long long rr[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int64_t i = i0 + (int64_t)u * W;
          rr[u] = i < nRec ? loadCoherent64(h + i) : 0ull;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int64_t i = i0 + (int64_t)u * W;
          const uint32_t x = (uint32_t)rr[u], y = (uint32_t)(rr[u] >> 32);
          if (i < nRec && (x & kSlNewFlag) && (x >> 9) == sid) {
            kids |= 1ull << (y & 63u);
          }
        }
      }
      if (kids) {
        atomOr64(&S.scanMask, kids);
      }
      ldsBarrier();
      if (tid == 0) {
        S.mask[q][X] |= S.scanMask;
      }
    }
    ldsBarrier();
  }
  ldsBarrier();
  if (tid < nState) { /* orphans get their parent back */
    const uint32_t spar = S.rec[q][tid].spar;
    for (int e = 0; e < nev; ++e) {
      const int X = (int)S.evLane[e];
      if (tid != X && spar == S.evSid[e]) {
        const uint32_t info = S.rec[q][tid].info;
        S.rec[q][tid].info = (info & ~0xFF00u) | ((uint32_t)(X + 1) << 8);
        atomOr64(&S.cmask[q][X], 1ull << (info & 0xFFu));
      }
    }
  }
  if (tid == 0) {
    S.row[q].nev = 0u;
  }
  ldsBarrier();
}

#define FLTX_SLPROF(i)                                        \
  do {                                                        \
    if (PROF && P.prof && (int)threadIdx.x == P.profThread) { \
      const unsigned long long t_ = devClock();               \
      acc[(i)] += t_ - tPrev;                                 \
      tPrev = t_;                                             \
    }                                                         \
  } while (0)

/* logAdd merge of two members (Utils.h:186-193): hi is the larger */
FLTX_DEV double slLogAdd(double hi, double lo) { return hi + log1p(exp(lo - hi)); }

/* GT = list positions per normal wave (nList <= GT * (waves - 2)); LA: the members of a merge group are
 * log-added (DecoderOptions::logAdd) -- the groups have the same fixed shape, the members that pass the
 * threshold are added in descending order as Utils.h:167-198 does, the back-pointer stays the best member's;
 * what changes is the frame's best candidate: a merged hypothesis can score above every candidate of its
 * frame, so the best hypothesis is not the last frame's best candidate any more -- the build publishes the
 * best surviving score and every wave prices the row with it at the head of the frame. */
/* TL: a token-level n-gram LM (LexiconFreeDecoder.cpp:69-85 with KenLM::score, lm/KenLM.cpp:63-75).  LM states are still
 * the trie of token histories (LMState::child, lm/LM.h:24-34), so lanes, merge groups and history records keep their
 * shape; what changes: (1) every new-token candidate adds lmWeight x score(context of S, n), gathered from the dense
 * table DecodeParams::tokLm by the lane's context row (SlRec::pad) -- the parent state's extension by last(S) adds the
 * score S was entered with (SlaneLds::tlIn); (2) the frame's best candidate is no recurrence any more: every wave
 * reduces its candidates to a maximum, one LDS atomic max per wave, one more barrier (the frame shape of fltx_ylane.h);
 * (3) decodeEnd adds lmWeight x finish(context) (:127-158; column N of the row); (4) the LM score of a returned path is
 * re-accumulated by the back-trace like the emitting-model score (new-token steps follow from the tokens alone). */
template <int GT, bool LA, bool ST, bool PROF, bool TL = false>
FLTX_DEV void slaneUtterance(const DecodeParams& P, char* smem) {
  static_assert(!(LA && ST), "streams with logAdd stay on the lane-per-slot step");
  using LdsT = typename std::conditional<TL, TlaneLds, SlaneLds>::type;
  LdsT& S = *(LdsT*)smem;
  TlMemo M = {};
  if constexpr (TL && !ST) { /* (a stream's memo is the generic engine's id table in HBM: tlStreamChild) */
    M = tlMemoOf(S, P.tlEdgeSlots, P.tlMaskSlots);
  }
  const int b = P.uttMap ? P.uttMap[blockIdx.x] : (int)blockIdx.x;
  const int W = (int)blockDim.x, tid = (int)threadIdx.x;
  const int lane = laneId(), wave = waveUniform(waveId());
  const int nW = W >> 6;
  /* waves 0 .. nW - 3 evaluate the listed tokens (GT list positions each); the next one owns the
   * blank / repeat / blank-then-last groups of every lane; the last one stages the next emission
   * row and does the per-frame housekeeping, off everybody else's path */
  const int selfWave = nW - 2;
  const int prepWave = nW - 1;
  const bool isSelfW = wave == selfWave, isSvcW = wave == prepWave;
#ifndef FLTX_EMU
  /* The two waves everybody waits for at the barriers win the issue arbitration of their SIMD (C2: 1.86 -> 1.79 ms;
   * tune bit 0 switches it off for measurements) */
  if (!(P.tune & 1) && (isSelfW || isSvcW)) {
    __builtin_amdgcn_s_setprio(3);
  }
#endif
  const int K = P.K, N = P.N;
  const bool ctc = P.criterion == 1;
  const int T = P.stepT ? P.stepT[b] : 0;
  const float* em = P.emissions ? P.emissions + P.emOff[b] : nullptr;
  const int64_t hbase = P.histOff[b];
  const double NEG = slNegInf();
  unsigned long long acc[8] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull};
  unsigned long long tPrev = devClock();
  static_assert(GT >= 3, "the self wave keeps its three groups in the slot arrays");

  /* ---- decodeBegin (LexiconFreeDecoder.cpp:20-28): the root state ------------------ */
  for (int i = tid; i < 2 * 64; i += W) {
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
  if constexpr (TL) {
    if (tid < 2) {
      S.fbest[tid] = 0ull;
    }
    if (tid == 0) {
      S.tlIn[0][0] = 0.0f; /* (the root state was not entered by a token) */
    }
    if constexpr (!ST) {
      for (int i = tid; i < P.tlEdgeSlots; i += W) {
        M.edge[i] = 0ull;
      }
      for (int i = tid; i < P.tlMaskSlots; i += W) {
        M.mmTag[i] = kTlNoSid;
      }
    }
    if (tid < 64) {
      S.dmask[tid] = 0ull;
    }
  }
  const int frame0 = ST ? P.uttFrame[b] : 0;   /* streams: rows already in the buffer */
  const int total0 = ST ? P.uttTotal[b] : 0;   /* ... frames decoded since decodeBegin */
  if (ST) {
    /* the beam parked by the last launch (sorted slots, one hypothesis each) -> lanes (one LM state each) */
    ldsBarrier(); /* (the wipes above) */
    const int nB = P.uttNBeam[b];
    if (tid == 0) {
      S.row[0].nev = 0u;
      S.row[1].nev = 0u;
      S.row[0].dead = 0u;
      S.row[1].dead = 0u;
      S.scal[SL_NEXTID] = (uint32_t)P.uttNextId[b];
    }
    if (wave == 0) {
      const bool valid = lane < nB;
      const size_t g = (size_t)b * K + (valid ? lane : 0);
      const double sc = P.gScore[g], amv = P.gAm[g];
      const double lmv0 = TL ? P.gLm[g] : 0.0;
      const uint32_t sid = valid ? P.gState[g] : 0xFFFFFFFFu, spar = P.gSPar[g], tp = P.gTokPb[g];
      const int32_t edge = P.gSEdge[g];
      const unsigned long long mkv = TL ? 0ull : P.gMask[g];
      /* TL: a state's n-gram context is a row of the dense table, kept by the generic engine's kernels (begin / end) in
       * word 0 of its stateCtx entry; the LM score it was entered with is the parent state's row at its last token */
      uint32_t ctxRow = 0u;
      float linRow = 0.0f;
      if constexpr (TL) {
        if (valid && sid != 0u) {
          ctxRow = tokLmRow(P, b, sid);
          linRow = __uint_as_float((uint32_t)P.tokLm[(size_t)tokLmRow(P, b, spar) * (size_t)P.tokLmStride + (size_t)(edge & 63)].x);
        }
      }
      const bool isB = (tp & kPrevBlank) != 0u;
      int leader = lane;
      for (int i = 0; i < nB; ++i) {
        const uint32_t si = waveReadLane32(sid, i);
        leader = (valid && si == sid && i < leader) ? i : leader;
      }
      const unsigned long long leaders = waveBallot(valid && leader == lane);
      const int L = popc64(leaders & ((1ull << leader) - 1ull));
      if (valid && leader == lane) {
        SlRec r;
        r.nb = NEG;
        r.b = NEG;
        r.info = (uint32_t)(sid == 0u ? P.sil : (edge & 63));
        r.sid = sid;
        r.spar = sid == 0u ? 0x7FFFFFu : spar;
        r.pad = TL ? ctxRow : 0u;
        S.rec[0][L] = r;
        S.mask[0][L] = mkv;
        if constexpr (TL) {
          S.tlIn[0][L] = linRow;
        }
        S.rsNB[L] = (uint8_t)kSlNoHyp;
        S.rsB[L] = (uint8_t)kSlNoHyp;
      }
      waveSync();
      if (valid) {
        if (isB) {
          S.rec[0][L].b = sc;
          S.amB[0][L] = amv;
          S.rsB[L] = (uint8_t)lane;
          if constexpr (TL) {
            S.lmB[0][L] = lmv0;
          }
        } else {
          S.rec[0][L].nb = sc;
          S.amNB[0][L] = amv;
          S.rsNB[L] = (uint8_t)lane;
          if constexpr (TL) {
            S.lmNB[0][L] = lmv0;
          }
        }
      }
      waveSync();
      const int nSt = popc64(leaders);
      /* lane j: history slots of its hypotheses, the lane of its parent state */
      const bool lv = lane < nSt;
      const SlRec mine = S.rec[0][lv ? lane : 0];
      int plr = -1;
      for (int i = 0; i < nSt; ++i) {
        const uint32_t si = waveReadLane32(mine.sid, i);
        plr = (lv && si == mine.spar && i != lane) ? i : plr;
      }
      if (lv) {
        S.rec[0][lane].info = (mine.info & 0xFFu) | ((uint32_t)(plr + 1) << 8) | ((uint32_t)S.rsNB[lane] << 16) |
                              ((uint32_t)S.rsB[lane] << 24);
        if (plr >= 0) {
          atomOr64(&S.cmask[0][plr], 1ull << (mine.info & 63u));
        }
      }
      if (lane == 0) {
        S.scal[SL_NSTATE] = (uint32_t)nSt;
      }
    }
  } else {
  if (tid == 0) {
    SlRec r;
    r.nb = 0.0;
    r.b = NEG;
    r.info = (uint32_t)P.sil | (0u << 8) | (0u << 16) | (kSlNoHyp << 24);
    r.sid = 0u;
    r.spar = 0x7FFFFFu;
    r.pad = 0u;
    S.rec[0][0] = r;
    S.row[0].nev = 0u;
    S.row[1].nev = 0u;
    S.row[0].dead = 0u;
    S.row[1].dead = 0u;
    S.mmaxKey[0] = f64Key(0.0); /* decodeBegin: the root hypothesis, score 0 */
    S.mmaxKey[1] = 0ull;
    P.histPT[hbase] = make_int2((int)kSlNoHyp, P.sil);
  }
  if (tid > 0 && tid < K) { /* unused slots of a row never look like the record of a new state (slReenter) */
    P.histPT[hbase + tid] = make_int2((int)kSlNoHyp, -1);
  }
  }
  /* emission rows: lane n of the prep wave holds e[t + 1][n] (used by the build of frame t) in one of
   * two registers, alternating with the frame parity; the register is refilled with row t + 3 right
   * after its use, so a load has two frames to arrive and is never moved between registers */
  float rowA = 0.0f, rowB = 0.0f;
  double bestChain = 0.0; /* prep wave: best candidate of the newest prepared frame (decodeBegin: score 0) */
  if (wave == prepWave) {
    const float v0 = (T > 0 && lane < N) ? em[lane] : 0.0f;
    ldsRowLoad(S.raw[1], em + (size_t)1 * N + lane, T > 1 && lane < N);
    ldsRowLoad(S.raw[2], em + (size_t)2 * N + lane, T > 2 && lane < N);
    /* (streams: the parked beam is sorted, slot 0 is the best hypothesis; prune has normalised it) */
    SlRowRegs r0 = slRowScan(P, v0, ctc, ST ? P.gScore[(size_t)b * K] : 0.0);
    bestChain = r0.best;
    slRowStore(P, S, 0, r0, 2);
    slRowStore(P, S, 1, r0, 2); /* (the positions past the list; the rest is rewritten by frame 0) */
  }
  ldsBarrier();

  int nState = ST ? (int)S.scal[SL_NSTATE] : 1;
  int nStatePrev = 0; /* TL: lanes the previous frame started from (0: there was none) */
  double endBest = 0.0; /* best hypothesis of the final beam (decodeEnd's threshold) */
  int winShift = kSlCoarseShift, winBase = kSlCoarseBase;
  bool dead = false; /* this utterance goes to the general engines */
  const int sil = P.sil, blank = P.blank;
  double silScore = P.silScore;
  double lmW = TL ? P.lmWeight : 0.0;
#ifndef FLTX_EMU
  /* (kept in a vector register: the frame loop is short of scalar ones, and the compiler would rather load the option
   * from the kernel arguments again at each of its uses -- a scalar load and a full wait per candidate) */
  __asm__ volatile("" : "+v"(silScore));
  if (TL) {
    __asm__ volatile("" : "+v"(lmW));
  }
#endif
  const int2* const tokLm = TL ? P.tokLm : nullptr;
  const int tokStride = TL ? P.tokLmStride : 0;
  int2* const histPT = P.histPT;

  /* one frame; PT = parity of the frame (compile time: every LDS address is an immediate) */
  /* RL = role of the wave (compile time as well: the three kinds of waves share the barriers and the
   * selection, and nothing else -- each gets its own straight-line frame and its own registers) */
  auto frameStep = [&](auto PT, auto RL, float& rowReg, const int t) {
    constexpr int p = decltype(PT)::value, q = p ^ 1;
    constexpr bool isSelf = decltype(RL)::value == 1, isSvc = decltype(RL)::value == 2;
    const int frameOut = frame0 + t + 1;
    const int64_t hrow = hbase + (int64_t)frameOut * K;
    /* ---- phase 1: own state, candidates, histogram ------------------------------------- */
    /* every LDS read of the phase is issued here, before anything waits for one */
    double best = S.row[p].best, thr = S.row[p].thr;
    const int nList = S.row[p].nList, silPos = S.row[p].silPos;
    uint32_t rowDead = S.row[p].dead;
    const uint32_t nev = S.row[p].nev;
    if (LA && !TL) { /* best candidate = best hypothesis of the beam + best token (sil priced apart: silScore) */
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
    SlRec me = {};
    unsigned long long cm = 0ull, mk = 0ull;
    if (!isSvc) { /* (the staging wave has the longest way to the first barrier: it skips what it does not use) */
      me = S.rec[p][lane];
      cm = S.cmask[p][lane];
      mk = S.mask[p][lane];
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
    if (isSelf) {
      eBlank = S.eAll[p][ctc ? blank : 0];
      allow = S.row[p].allow;
    } else if (!isSvc) {
#pragma unroll
      for (int j = 0; j < GT; ++j) {
        ev[j] = S.eTok[p][wave * GT + j];
        tb[j] = S.tokBit[p][wave * GT + j];
      }
    }
    (void)nList;
#ifndef FLTX_EMU
    /* The compiler otherwise sinks the record loads below the two rare branches that follow and predicates the
     * score loads on `live`: three LDS round trips one after the other.  An empty statement that "uses" every value
     * loaded above keeps the loads where they are written -- all issued, one wait. */
    if (isSelf) {
      __asm__ volatile("" : "+v"(me.nb), "+v"(me.b), "+v"(me.info), "+v"(me.sid), "+v"(me.spar), "+v"(cm), "+v"(mk),
                       "+v"(eBlank), "+v"(allow));
    } else if (!isSvc) {
      __asm__ volatile("" : "+v"(me.nb), "+v"(me.b), "+v"(me.info), "+v"(me.sid), "+v"(me.spar), "+v"(cm), "+v"(mk));
    }
#endif
    /* token-level n-gram LM: the LM scores (and the contexts behind them) of this lane's state for the tokens of this
     * wave's list positions -- one 8-byte gather per (lane, position) from the state's row of the dense table (rows of
     * the states in the beam stay in the L1 / L2 from frame to frame); self wave: the score of last(S) after S (the
     * blank-then-last candidate) and the score S was entered with (the parent state's extension by last(S)) */
    int2 lmv[GT];
    int2 lmLast = make_int2(0, 0);
    float lIn = 0.0f;
#pragma unroll
    for (int j = 0; j < GT; ++j) {
      lmv[j] = make_int2(0, 0);
    }
    if (TL && !isSvc) { /* (issued as soon as the lane's record is here: the re-entry work and the second LDS round trip
                           run under their latency; a state's context and last token do not change when it is re-linked) */
      const bool liveE = lane < nState;
      const int2* lrow = tokLm + (size_t)(liveE ? me.pad : 0u) * (size_t)tokStride;
      if (isSelf) {
        lmLast = lrow[liveE ? (int)(me.info & 0xFFu) & 63 : 0];
        lIn = S.tlIn[p][lane];
      } else {
#pragma unroll
        for (int j = 0; j < GT; ++j) {
          lmv[j] = lrow[tb[j] != 0ull ? __builtin_ctzll(tb[j]) : 0];
        }
      }
    }
    if constexpr (TL) {
      /* the lanes the last frame dropped: their child masks go to the memo (the last frame's records and masks are
       * still in the other parity's arrays; `dmask` has the edges their last extension created) -- the staging wave,
       * which waits longest at the barrier behind the frame's best (wave 0 wipes these masks after that barrier) */
      if (isSvc) {
        if constexpr (!ST) {
        const bool dropped = lane < nStatePrev && S.newLane[lane] < 0;
        const uint32_t dsid = S.rec[q][lane].sid;
        const uint32_t ms = tlMaskSlot(M, dsid);
        if (dropped) {
          M.mmTag[ms] = dsid;
        }
        waveSync(); /* (two dropped states of one frame may share a slot: the one whose id stays writes the mask) */
        if (dropped && M.mmTag[ms] == dsid) {
          M.mmMask[ms] = S.mask[q][lane] | S.dmask[lane];
        }
        S.dmask[lane] = 0ull;
        }
        /* ... and the housekeeping for everybody (what this frame's build adds to): nobody else reads the other parity's
         * masks in this frame, and this wave has the time -- the first token wave is on the way to the barrier */
        S.cmask[q][lane] = 0ull;
        S.mask[q][lane] = 0ull;
        if (lane < 32) {
          S.off[lane] = 0u;
        }
        if (lane == 0) {
          S.scal[SL_BCNT] = 0u;
          S.fbest[q] = 0ull; /* (the next frame's; last read a frame ago) */
        }
      }
    }
    if (TL) {
      FLTX_SLPROF(0);
    }
    bool reentryDone = false;
    if constexpr (TL) {
    if (nev != 0u && (nev >> 16) == 0u) {
      reentryDone = true;
      /* states entered again in the last build, all known to the memos (ids and child masks are in place): their
       * orphans -- lanes whose parent state it is -- get their parent lane back, in this wave's registers: lane l holds
       * lane l's record in every wave, and the records the next build writes are made from these registers */
      if (!isSvc) {
        const int ne = (int)nev;
        for (int e0 = 0; e0 < ne; e0 += 4) {
          uint32_t xs[4], ss[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) { /* (four events per LDS round trip) */
            xs[u] = e0 + u < ne ? S.evLane[e0 + u] : 0u;
            ss[u] = e0 + u < ne ? S.evSid[e0 + u] : kTlNoSid;
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            if (e0 + u < ne) {
              const int X = (int)xs[u];
              unsigned long long orphans = waveBallot(lane < nState && lane != X && me.spar == ss[u]);
              unsigned long long bits = 0ull;
              if ((orphans >> lane) & 1ull) {
                me.info = (me.info & ~0xFF00u) | ((uint32_t)(X + 1) << 8);
              }
              while (orphans) {
                const int o = __builtin_ctzll(orphans);
                orphans &= orphans - 1ull;
                bits |= 1ull << (waveReadLane32(me.info, o) & 63u);
              }
              cm |= lane == X ? bits : 0ull;
            }
          }
        }
      }
    }
    }
    if (!reentryDone && nev != 0u) { /* rare: states re-entered the beam in the previous build */
      if constexpr (TL && ST) {
        /* (never: a stream's states are named by the generic engine's table, every event knows its id) */
      } else if constexpr (TL) {
        tlReenter(S, histPT, p, nState, hbase, (int64_t)frameOut * K, P.tlEdgeSlots, P.tlMaskSlots);
      } else if (ST) {
        slRelink(S, P.maskTab + (size_t)b * P.idCap, p, nState);
      } else {
        slReenter(S, histPT, p, nState, hbase, (int64_t)frameOut * K);
      }
      me = S.rec[p][lane];
      cm = S.cmask[p][lane];
      mk = S.mask[p][lane];
    }
    if (TL) {
      FLTX_SLPROF(6); /* (TL: re-entry) */
    }
    if (rowDead) { /* nothing to extend with, or not finite: general path */
      dead = true;
      return;
    }
    const bool live = lane < nState && !isSvc;
    const double nb = live ? me.nb : NEG, bb = live ? me.b : NEG;
    const int last = (int)(me.info & 0xFFu) & 63;
    const int pl = live ? (int)((me.info >> 8) & 0xFFu) - 1 : -1;
    const uint32_t hypNB = (me.info >> 16) & 0xFFu, hypB = me.info >> 24;
    const bool whichB = bb > nb;
    const double m = whichB ? bb : nb;
    const uint32_t hypM = whichB ? hypB : hypNB;
    /* streams carry the emitting-model score of every hypothesis (getBestHypothesis returns an ancestor's) */
    double amNBv = 0.0, amBv = 0.0, parAmNB = 0.0, parAmB = 0.0;
    double lmNBv = 0.0, lmBv = 0.0, parLmNB = 0.0, parLmB = 0.0; /* ... and, with a token LM, its LM score */
    if (ST && !isSvc) {
      amNBv = S.amNB[p][lane];
      amBv = S.amB[p][lane];
      if constexpr (TL) {
        lmNBv = S.lmNB[p][lane];
        lmBv = S.lmB[p][lane];
      }
    }
    const double amM = whichB ? amBv : amNBv;
    const double lmM = whichB ? lmBv : lmNBv;
    auto amStep = [&](double amPrev, double e, int n, int prevTok) {
      double x = e; /* LexiconFreeDecoder.cpp:58-64: the ASG transition enters the emitting-model score only */
      if (!ctc && P.transitions && total0 + t > 0) {
        x = x + (double)P.transitions[(size_t)n * N + prevTok];
      }
      return amPrev + x;
    };
    /* self wave: what its groups need beyond the lane's own record (second LDS round trip) */
    SlRec par;
    double eLast = 0.0;
    if (isSelf) {
      par = S.rec[p][pl >= 0 ? pl : 0];
      eLast = S.eAll[p][last];
      if (ST) {
        parAmNB = S.amNB[p][pl >= 0 ? pl : 0];
        parAmB = S.amB[p][pl >= 0 ? pl : 0];
        if constexpr (TL) {
          parLmNB = S.lmNB[p][pl >= 0 ? pl : 0];
          parLmB = S.lmB[p][pl >= 0 ? pl : 0];
        }
      }
    }
    FLTX_SLPROF(0);
    double cs[GT];
    int cbin[GT];
    uint32_t parR = kSlNoHyp;
    double amR = 0.0; /* streams: emitting-model score of the repeat group's winning member ... */
    int prevR = 0;    /* ... and its token (ASG transition) */
    double lmR = 0.0; /* ... and (token LM) its LM score, the entering score included for the parent's extension */
    SlRowRegs nextRow = {};
    if (isSvc) {
      /* the next frame's emission row (the masks and counters this frame's build adds to are wiped
       * by the first token wave, which reaches the barrier earlier) */
#pragma unroll
      for (int j = 0; j < GT; ++j) {
        cs[j] = NEG;
        cbin[j] = kSlInvalid;
      }
      if (t + 1 < T) {
        ldsRowWait(); /* (issued two frames ago) */
        const float rv = lane < N ? S.raw[(t + 1) % 3][lane] : 0.0f;
        nextRow = slRowScan(P, rv, ctc, TL ? 0.0 : bestChain, silScore); /* (TL: only whether the row is usable) */
        bestChain = nextRow.best;
      }
      ldsRowLoad(S.raw[t % 3], em + (size_t)(t + 3) * N + lane, t + 3 < T && lane < N); /* (row t's slot: read last frame) */
      if (LA && !TL && lane == 0) {
        S.mmaxKey[q] = 0ull; /* (this frame's build raises it) */
      }
      (void)rowReg;
    } else if (TL) {
      /* (the candidates of a token-LM frame: below, around the barrier that publishes the frame's best) */
    } else if (!isSelf) {
      /* tokens this lane does not extend with here: its own last token (the repeat and the
       * blank-then-last case belong to the self wave) and those whose child state holds a lane
       * (that lane merges the extension into its repeat).  A lane without a state skips all. */
      const uint32_t lastLo = last < 32 ? 1u << last : 0u, lastHi = last < 32 ? 0u : 1u << (last - 32);
      const uint32_t skLo = live ? ((uint32_t)cm | lastLo) : 0xFFFFFFFFu;
      const uint32_t skHi = live ? ((uint32_t)(cm >> 32) | lastHi) : 0xFFFFFFFFu;
      const int silJ = silPos - wave * GT; /* list position of sil relative to this wave's first */
#pragma unroll
      for (int j = 0; j < GT; ++j) {
        double c = m + ev[j]; /* NaN past the end of the list */
        if (j == silJ) {
          c = c + silScore;
        }
        const uint32_t hit = (skLo & (uint32_t)tb[j]) | (skHi & (uint32_t)(tb[j] >> 32));
        const bool ok = hit == 0u && c >= thr;
        if (LA) { /* the state's other hypothesis reaches the same child state: fl(a + e) is monotone, so it
                     is the smaller member */
          double c2 = (whichB ? nb : bb) + ev[j];
          if (j == silJ) {
            c2 = c2 + silScore;
          }
          if (ok && (whichB ? hypNB : hypB) != kSlNoHyp && c2 >= thr) {
            c = slLogAdd(c, c2);
          }
        }
        cs[j] = c;
        cbin[j] = ok ? slBin<LA>(best, c, winShift, winBase) : kSlInvalid;
      }
    } else {
      const bool lastOk = live && ((allow >> last) & 1ull) != 0ull && !(ctc && last == blank);
      const bool lastSil = last == sil;
      /* (S, blank, true): LexiconFreeDecoder.cpp:86-97 */
      double cB = m + eBlank;
      if (blank == sil) {
        cB = cB + silScore;
      }
      const bool okB = ctc && live && ((allow >> (ctc ? blank : 0)) & 1ull) != 0ull && cB >= thr;
      if (LA) {
        double cB2 = (whichB ? nb : bb) + eBlank;
        if (blank == sil) {
          cB2 = cB2 + silScore;
        }
        if (okB && (whichB ? hypNB : hypB) != kSlNoHyp && cB2 >= thr) {
          cB = slLogAdd(cB, cB2);
        }
      }
      /* (S, last, false): the repeat (:98-110) and the parent state's extension by last (:69-85) */
      const int lastP = (int)(par.info & 0xFFu);
      const uint32_t h1 = (par.info >> 16) & 0xFFu, h2 = par.info >> 24;
      /* which members exist is told by the history slots, not by the scores: with an unbounded
       * threshold a score of -inf passes every comparison */
      const bool has0 = hypNB != kSlNoHyp;
      const bool has1 = pl >= 0 && last != lastP && h1 != kSlNoHyp;
      const bool has2 = pl >= 0 && ctc && h2 != kSlNoHyp;
      const bool hasB = hypB != kSlNoHyp;
      double r0 = nb + eLast;
      double r1 = has1 ? par.nb + eLast : NEG;
      double r2 = has2 ? par.b + eLast : NEG;
      /* (S.last, last, false) from (S, blank, true) when no lane holds S.last */
      double cL = bb + eLast;
      if (silScore != 0.0) {
        r0 = lastSil ? r0 + silScore : r0;
        r1 = lastSil ? r1 + silScore : r1;
        r2 = lastSil ? r2 + silScore : r2;
        cL = lastSil ? cL + silScore : cL;
      }
      /* max-merge (Utils.h:194-196); a tie goes to the lower history slot (a member that does not
       * exist has slot 255 and score -inf: it never wins against one that does) */
      double cR = r0;
      parR = hypNB;
      amR = amNBv;
      prevR = last;
      /* (selects, not branches: the lanes disagree about who wins, and a divergent branch costs more than both sides) */
      const bool t1 = has1 & ((r1 > cR) | ((r1 == cR) & (h1 < parR)));
      cR = t1 ? r1 : cR;
      parR = t1 ? h1 : parR;
      amR = t1 ? parAmNB : amR;
      prevR = t1 ? lastP : prevR;
      const bool t2 = has2 & ((r2 > cR) | ((r2 == cR) & (h2 < parR)));
      cR = t2 ? r2 : cR;
      parR = t2 ? h2 : parR;
      amR = t2 ? parAmB : amR;
      prevR = t2 ? blank : prevR;
      bool okR = lastOk && (has0 || has1 || has2) && cR >= thr;
      if (LA) {
        /* the members that pass the threshold, best first (the best is cR, its slot parR) */
        const bool v0 = has0 && r0 >= thr, v1 = has1 && r1 >= thr, v2 = has2 && r2 >= thr;
        okR = lastOk && (v0 || v1 || v2);
        double a = v0 ? r0 : NEG, bq = v1 ? r1 : NEG, cq = v2 ? r2 : NEG;
        /* three-element sort, descending */
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
      const bool okL = ctc && lastOk && hasB && ((cm >> last) & 1ull) == 0ull && cL >= thr;
      cs[0] = cB;
      cs[1] = cR;
      cs[2] = cL;
      cbin[0] = okB ? slBin<LA>(best, cB, winShift, winBase) : kSlInvalid;
      cbin[1] = okR ? slBin<LA>(best, cR, winShift, winBase) : kSlInvalid;
      cbin[2] = okL ? slBin<LA>(best, cL, winShift, winBase) : kSlInvalid;
#pragma unroll
      for (int j = 3; j < GT; ++j) {
        cs[j] = NEG;
        cbin[j] = kSlInvalid;
      }
    }
    if constexpr (TL) {
      /* ---- the same candidates with the LM term; the frame's best is their maximum (Utils.h:131-137) ---------- */
      bool pre[GT];    /* the candidate exists (whether it passes the threshold is known after the barrier) */
      double c2v[GT];  /* logAdd: the smaller member of the group */
      bool hasOther = false;
      bool lastOk = false, has0 = false, has1 = false, has2 = false;
      double r0 = NEG, r1 = NEG, r2 = NEG;
#pragma unroll
      for (int j = 0; j < GT; ++j) {
        pre[j] = false;
        c2v[j] = NEG;
        if (!isSvc) {
          cs[j] = NEG;
          cbin[j] = kSlInvalid;
        }
      }
      if (isSvc) {
      } else if (!isSelf) {
        const uint32_t lastLo = last < 32 ? 1u << last : 0u, lastHi = last < 32 ? 0u : 1u << (last - 32);
        const uint32_t skLo = live ? ((uint32_t)cm | lastLo) : 0xFFFFFFFFu;
        const uint32_t skHi = live ? ((uint32_t)(cm >> 32) | lastHi) : 0xFFFFFFFFu;
        const int silJ = silPos - wave * GT;
        hasOther = (whichB ? hypNB : hypB) != kSlNoHyp;
#pragma unroll
        for (int j = 0; j < GT; ++j) {
          /* LexiconFreeDecoder.cpp:64-67,69-85: score = prev.score + e (+ silScore), candidate = score + lmWeight * lmScore */
          const double wl = lmW * (double)__uint_as_float((uint32_t)lmv[j].x);
          double c = m + ev[j]; /* NaN past the end of the list */
          if (j == silJ) {
            c = c + silScore;
          }
          c = c + wl;
          const uint32_t hit = (skLo & (uint32_t)tb[j]) | (skHi & (uint32_t)(tb[j] >> 32));
          pre[j] = hit == 0u && c == c;
          if (LA) {
            double c2 = (whichB ? nb : bb) + ev[j];
            if (j == silJ) {
              c2 = c2 + silScore;
            }
            c2v[j] = c2 + wl;
          }
          cs[j] = c;
        }
      } else {
        lastOk = live && ((allow >> last) & 1ull) != 0ull && !(ctc && last == blank);
        const bool lastSil = last == sil;
        hasOther = (whichB ? hypNB : hypB) != kSlNoHyp;
        /* (S, blank, true): :86-97, no LM term */
        double cB = m + eBlank;
        if (blank == sil) {
          cB = cB + silScore;
        }
        pre[0] = ctc && live && ((allow >> (ctc ? blank : 0)) & 1ull) != 0ull && cB == cB;
        if (LA) {
          double cB2 = (whichB ? nb : bb) + eBlank;
          if (blank == sil) {
            cB2 = cB2 + silScore;
          }
          c2v[0] = cB2;
        }
        /* (S, last, false): the repeat (:98-110, no LM term) and the parent state's extension by last (:69-85: the LM
         * score S was entered with) */
        const int lastP = (int)(par.info & 0xFFu);
        const uint32_t h1 = (par.info >> 16) & 0xFFu, h2 = par.info >> 24;
        has0 = hypNB != kSlNoHyp;
        has1 = pl >= 0 && last != lastP && h1 != kSlNoHyp;
        has2 = pl >= 0 && ctc && h2 != kSlNoHyp;
        const bool hasB = hypB != kSlNoHyp;
        const double wIn = lmW * (double)lIn, wL = lmW * (double)__uint_as_float((uint32_t)lmLast.x);
        r0 = nb + eLast;
        r1 = par.nb + eLast;
        r2 = par.b + eLast;
        double cL = bb + eLast; /* (S.last, last, false) from (S, blank, true) when no lane holds S.last */
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
        parR = hypNB;
        amR = amNBv;
        prevR = last;
        lmR = lmNBv;
        const bool t1 = has1 & ((r1 > cR) | ((r1 == cR) & (h1 < parR)));
        cR = t1 ? r1 : cR;
        parR = t1 ? h1 : parR;
        amR = t1 ? parAmNB : amR;
        prevR = t1 ? lastP : prevR;
        lmR = t1 ? parLmNB + (double)lIn : lmR;
        const bool t2 = has2 & ((r2 > cR) | ((r2 == cR) & (h2 < parR)));
        cR = t2 ? r2 : cR;
        parR = t2 ? h2 : parR;
        amR = t2 ? parAmB : amR;
        prevR = t2 ? blank : prevR;
        lmR = t2 ? parLmB + (double)lIn : lmR;
        pre[1] = lastOk && (has0 || has1 || has2) && cR == cR;
        pre[2] = ctc && lastOk && hasB && ((cm >> last) & 1ull) == 0ull && cL == cL;
        cs[0] = cB;
        cs[1] = cR;
        cs[2] = cL;
      }
      {
        unsigned long long k = 0ull;
#pragma unroll
        for (int j = 0; j < GT; ++j) {
          const unsigned long long kj = pre[j] ? f64Key(cs[j]) : 0ull;
          k = kj > k ? kj : k;
        }
        if (waveBallot(k != 0ull) != 0ull) {
          /* (two 32-bit scans -- the upper halves, then the lower halves of the lanes that hold the largest upper
           * half -- issue half the instructions of one 64-bit scan) */
          const uint32_t hiMax = waveMax32((uint32_t)(k >> 32));
          const uint32_t loMax = waveMax32((uint32_t)(k >> 32) == hiMax ? (uint32_t)k : 0u);
          if (lane == 0) {
            atomMax64(&S.fbest[p], ((unsigned long long)hiMax << 32) | loMax);
          }
        }
      }
      FLTX_SLPROF(7); /* (TL: candidates + maximum) */
      ldsBarrier(); /* 0: the frame's best candidate */
      if (wave == 0 && lane == 0) {
        S.row[p].nev = 0u; /* (every wave has read it; the next build counts here again) */
      }
      {
        const unsigned long long bk = S.fbest[p];
        best = f64FromKey(bk);
        thr = best - P.beamThreshold;
        if (bk == 0ull || !(best - best == 0.0)) { /* no candidate at all, or not finite: the general engines */
          dead = true;
          return;
        }
      }
      if (isSvc) {
      } else if (!isSelf) {
#pragma unroll
        for (int j = 0; j < GT; ++j) {
          const bool ok = pre[j] && cs[j] >= thr;
          if (LA && ok && hasOther && c2v[j] >= thr) {
            cs[j] = slLogAdd(cs[j], c2v[j]);
          }
          cbin[j] = ok ? slBin<LA>(best, cs[j], winShift, winBase) : kSlInvalid;
        }
      } else {
        const bool okB = pre[0] && cs[0] >= thr;
        if (LA && okB && hasOther && c2v[0] >= thr) {
          cs[0] = slLogAdd(cs[0], c2v[0]);
        }
        bool okR = pre[1] && cs[1] >= thr;
        if (LA) {
          const bool v0 = has0 && r0 >= thr, v1 = has1 && r1 >= thr, v2 = has2 && r2 >= thr;
          okR = lastOk && (v0 || v1 || v2);
          double a = v0 ? r0 : NEG, bq = v1 ? r1 : NEG, cq = v2 ? r2 : NEG;
          double t0 = a > bq ? a : bq, t1 = a > bq ? bq : a;
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
          cs[1] = okR ? accv : cs[1];
        }
        const bool okL = pre[2] && cs[2] >= thr;
        cbin[0] = okB ? slBin<LA>(best, cs[0], winShift, winBase) : kSlInvalid;
        cbin[1] = okR ? slBin<LA>(best, cs[1], winShift, winBase) : kSlInvalid;
        cbin[2] = okL ? slBin<LA>(best, cs[2], winShift, winBase) : kSlInvalid;
      }
    }
    if (!TL && wave == 0) { /* housekeeping for everybody: what this frame's build adds to */
      S.cmask[q][lane] = 0ull;
      S.mask[q][lane] = 0ull;
      if (lane < 32) {
        S.off[lane] = 0u;
      }
      if (lane == 0) {
        S.scal[SL_BCNT] = 0u;
      }
    }
    /* one LDS atomic per candidate inside the window (bin 0 = nearer than the window included);
     * what lies beyond the window (hundreds per frame, they would all hit one address) is not
     * counted unless the window turns out to hold fewer than K */
#pragma unroll
    for (int j = 0; j < GT; ++j) {
      if (cbin[j] < kSlFar) {
        atomAdd32(&S.hist[p][cbin[j]], 1u);
      }
    }
    FLTX_SLPROF(1);
    ldsBarrier(); /* 1 */
    /* ---- phase 2: which candidates survive (Utils.h:200-220) ---------------------------- */
    unsigned long long selMask[GT]; /* per slot: the lanes whose candidate survives */
    /* The usual frame: the window holds at least K and the bins up to the K-th best's hold exactly K (or everything
     * counted survives) -- one scan, a bin limit, four ballots, in a straight line.  Everything else (fewer than K
     * inside the window, a boundary bin of which only some survive, a bin too crowded to rank) is the loop below,
     * out of the usual frame's way: the selection is (bin <= lim) or one of the `take` bits either way. */
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
      bool full = false; /* the counts include what lies beyond the window */
      for (;;) {
        if (!full && !sc.crossed) {
          /* fewer than K inside the window: count the far ones too (one add per wave) */
          int nFar = 0;
#pragma unroll
          for (int j = 0; j < GT; ++j) {
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
        if (sc.total <= K) { /* everything counted survives (with the far ones: all above the threshold) */
          lim = full ? kSlFar : kSlFar - 1;
          break;
        }
        const int need = K - sc.cum;
        if (sc.cnt == need) {
          lim = sc.bstar;
          break;
        }
        if (sc.cnt <= kSlBCap) { /* the members of the K-th best's bin compare with each other */
          if (PROF && !TL) {
            acc[6] += 1ull;
            acc[7] += (unsigned long long)sc.cnt;
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
                                [&](int j) { return f64Key(cs[j]); });
          lim = sc.bstar - 1;
          break;
        }
        /* too many in one bin: the K-th best's float bits lie in [lo, hi]; look again
         * through the finest window that spans that bracket (<= 4 rounds: 32 bits, 9 per round) */
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
          if (bLo >= bHi) { /* equal to the last bit and more of them than the pairwise list holds */
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
    /* next frame's window: the K-th best in the middle, 128 bins per octave */
    if (sc.total > K) {
      /* (a window of one octave at 256 bins per octave was tried for the token-LM variant, whose frames hold three times
       * the candidates inside the window: the K-th best then leaves the window every few frames -- 4.2 -> 7.3 ms) */
      const int q15 = shift >= kSlFineShift ? (sc.bstar + base) << (shift - kSlFineShift)
                                            : (sc.bstar + base) >> (kSlFineShift - shift);
      winShift = kSlFineShift;
      winBase = q15 > kSlMid ? q15 - kSlMid : 0;
    }
    FLTX_SLPROF(2);
    /* new lanes: survivors first (self wave), then the new states wave by wave */
    int nNewWave = 0;
    int myNew[GT];
    int surv = -1;
    uint32_t hNB = kSlNoHyp, hB = kSlNoHyp;
    if (isSvc) {
#pragma unroll
      for (int j = 0; j < GT; ++j) {
        myNew[j] = 0;
      }
      /* The next frame's row and the other parity's histogram (last read a frame ago) are written here, while the
       * other waves count their new lanes: this wave then has nothing left between the second and the third barrier.
       * (It issues no stores to HBM: it loads an emission row per frame, and a wait for that load would wait for
       * every store issued since as well.) */
      if (t + 1 < T) {
        slRowStore(P, S, q, nextRow, P.Kt < N ? 1 : 0);
      }
      ((uint4*)S.hist[q])[lane] = make_uint4(0u, 0u, 0u, 0u);
    } else if (!isSelf) {
#pragma unroll
      for (int j = 0; j < GT; ++j) { /* (no branch around an empty position: it costs what it would skip) */
        myNew[j] = nNewWave + wavePrefixCount(selMask[j]);
        nNewWave += popc64(selMask[j]);
      }
      if (lane > wave && lane <= selfWave + 1 && nNewWave > 0) {
        atomAdd32(&S.off[lane], (uint32_t)nNewWave);
      }
    } else {
      const unsigned long long balB = selMask[0], balR = selMask[1], balL = selMask[2];
      const unsigned long long balS = balB | balR;
      const bool sR = ((balR >> lane) & 1ull) != 0ull;
      surv = ((balS >> lane) & 1ull) ? wavePrefixCount(balS) : -1;
      hNB = (uint32_t)(wavePrefixCount(balR) + wavePrefixCount(balB));
      hB = hNB + (sR ? 1u : 0u);
#pragma unroll
      for (int j = 0; j < GT; ++j) {
        myNew[j] = 0;
      }
      myNew[2] = wavePrefixCount(balL);
      nNewWave = popc64(balL);
      S.newLane[lane] = surv;
      if (lane == 0) {
        S.scal[SL_NSURV] = (uint32_t)popc64(balS);
        S.scal[SL_NHSURV] = (uint32_t)(popc64(balR) + popc64(balB));
      }
      if (lane == selfWave + 1 && nNewWave > 0) {
        atomAdd32(&S.off[lane], (uint32_t)nNewWave);
      }
    }
    if (LA && !TL && !isSvc) { /* the best hypothesis of the next beam */
      unsigned long long k = 0ull;
#pragma unroll
      for (int j = 0; j < GT; ++j) {
        const unsigned long long kj = ((selMask[j] >> lane) & 1ull) ? f64Key(cs[j]) : 0ull;
        k = kj > k ? kj : k;
      }
      if (waveBallot(k != 0ull) != 0ull) {
        k = waveMax64(k);
        if (lane == 0) {
          atomMax64(&S.mmaxKey[q], k);
        }
      }
    }
    FLTX_SLPROF(3);
    ldsBarrier(); /* 2 */
    /* ---- phase 3: every survivor is written by the lane that evaluated it ---------------- */
    const int nSurv = (int)S.scal[SL_NSURV], nHSurv = (int)S.scal[SL_NHSURV];
    const int offW = (int)S.off[wave], nNew = (int)S.off[selfWave + 1];
    const int myNewLane = S.newLane[lane];
    const int plNew = S.newLane[pl >= 0 ? pl : 0];
    auto plainRec = [&](uint32_t hp, int n) { return make_int2(hp == kSlNoHyp ? -1 : (int)hp, n); };
    auto scoreRec = [&](int64_t at, double c, double am, double lmv) {
      if (P.histS) {
        double* hs = P.histS + 3 * at;
        hs[0] = c;
        hs[1] = am;
        hs[2] = lmv;
      }
    };
    /* (the parent lane's data as arguments: a token wave of the stream variant with a token LM gathers its survivors
     * into its first lanes and builds them in one pass, see below -- there the parent is another lane's) */
    auto newStateOf = [&](int idx, double c, int n, uint32_t hp, double amNew, uint32_t ctxNew, float lNew, double lmNew,
                          uint32_t parSid, bool againBit, int parNewLane, int parOldLane) {
      const int nl = nSurv + idx;
      const uint32_t hyp = (uint32_t)(nHSurv + idx);
      SlRec r;
      r.nb = c;
      r.b = NEG;
      r.info = (uint32_t)n | ((uint32_t)(parNewLane + 1) << 8) | (hyp << 16) | (kSlNoHyp << 24);
      r.sid = (uint32_t)frameOut * (uint32_t)K + hyp;
      r.spar = parSid;
      r.pad = TL ? ctxNew : 0u; /* (a function of the token history: a state entered again gets the context it had) */
      if (TL) {
        S.tlIn[q][nl] = lNew;
      }
      bool again = againBit; /* this edge had a child before */
      uint32_t known = kTlNoSid;
      bool tlSlow = true; /* a re-entry the memos do not answer: the next frame looks it up in the rows (tlReenter) */
      if constexpr (TL && ST) {
        /* a stream: the generic engine's (parent id, edge) -> id table names the state and remembers that it did */
        bool fresh = false;
        r.sid = tlStreamChild(P, b, parSid, n, (uint32_t)(total0 + t + 1), &S.scal[SL_NEXTID], &S.scal[SL_STATUS], fresh);
        again = !fresh;
        known = r.sid;
        tlSlow = false;
        if (fresh && P.lmOrder > 1) {
          P.stateCtx[((size_t)b * P.stateCap + r.sid) * (size_t)(P.lmOrder - 1)] = (int32_t)ctxNew; /* (its context: a row number) */
        }
        S.amNB[q][nl] = amNew;
        S.lmNB[q][nl] = lmNew;
      } else if constexpr (TL) {
        const uint32_t es = tlEdgeSlot(M, parSid, (uint32_t)n);
        if (again) { /* ... and the edge memo may still know which: the state keeps its id, the rows are not searched */
          const unsigned long long cur = M.edge[es];
          if ((cur >> 23) == (tlEdgePack(parSid, (uint32_t)n, 0u) >> 23)) {
            known = (uint32_t)cur & 0x7FFFFFu;
            r.sid = known;
            /* ... and the mask memo the child mask it had when it dropped out: then nothing is left for the next frame
             * but to give its orphans their parent back, which every wave does in its own registers */
            const uint32_t ms = tlMaskSlot(M, known);
            if (M.mmTag[ms] == known) {
              atomOr64(&S.mask[q][nl], M.mmMask[ms]);
              tlSlow = false;
            }
          }
        } else {
          M.edge[es] = tlEdgePack(parSid, (uint32_t)n, r.sid);
        }
        if (parNewLane < 0) { /* the parent drops out with this frame: its saved child mask must hold this edge */
          atomOr64(&S.dmask[parOldLane], 1ull << n);
        }
      }
      if (ST && !TL) {
        uint32_t* slot = &P.childTab[((size_t)b * P.idCap + parSid) * N + n];
        if (again) {
          r.sid = loadCoherent32(slot);
        } else {
          r.sid = allocStateId(P, b, atomAdd32(&S.scal[SL_NEXTID], 1u), parSid, n, (uint32_t)(total0 + t + 1),
                               &S.scal[SL_STATUS]);
          *slot = r.sid;
          P.maskTab[(size_t)b * P.idCap + r.sid] = 0ull;
          atomOr64(&P.maskTab[(size_t)b * P.idCap + parSid], 1ull << n); /* (the parent may leave the beam) */
        }
        S.amNB[q][nl] = amNew;
      }
      S.rec[q][nl] = r;
      if (parNewLane >= 0) {
        atomOr64(&S.cmask[q][parNewLane], 1ull << n);
        atomOr64(&S.mask[q][parNewLane], 1ull << n);
      }
      if (ST) {
        histPT[hrow + hyp] = plainRec(hp, n);
        scoreRec(hrow + hyp, c, amNew, lmNew);
      } else {
        histPT[hrow + hyp] = make_int2((int)(hp | kSlNewFlag | (parSid << 9)), n);
      }
      if (again) { /* it may have descendants in the beam */
        /* (TL: the upper half counts the events that need the rows) */
        const uint32_t e = atomAdd32(&S.row[q].nev, (TL && tlSlow) ? 0x10001u : 1u) & 0xFFFFu;
        S.evLane[e] = (uint32_t)nl;
        S.evSpar[e] = parSid;
        S.evTok[e] = (uint32_t)n;
        if constexpr (TL) {
          S.evSid[e] = known;
          S.evNeed[e] = known == kTlNoSid ? 3u : (tlSlow ? 2u : 0u);
        }
      }
    };
    auto newState = [&](int idx, double c, int n, uint32_t hp, double amNew, uint32_t ctxNew, float lNew, double lmNew) {
      newStateOf(idx, c, n, hp, amNew, ctxNew, lNew, lmNew, me.sid, ((mk >> n) & 1ull) != 0ull, myNewLane, lane);
    };
    if (isSvc) {
      /* (its part of the build went ahead of the second barrier) */
    } else if (!isSelf) {
      if (!ST && wave == 0 && lane >= nHSurv + nNew && lane < K) { /* unused slots of the history row: see slReenter
                                                                       (a token wave: they wait at the third barrier) */
        histPT[hrow + lane] = make_int2((int)kSlNoHyp, -1);
      }
      bool gathered = false;
      if constexpr (TL && ST) {
        /* A stream's new state costs a look-up (and, for a fresh state, a compare-and-swap) in the id table in HBM: a
         * microsecond or two each.  The loop below would pay that once per list position that has a survivor, one after
         * the other; instead the wave's survivors drop what a new state is made from into the wave's scratch and the
         * wave's first lanes build one state each -- their table accesses are in flight together.  (Offline, where a new
         * state touches LDS only, this was tried and lost: DESIGN 4.1.) */
        if (nNewWave > 1 && nNewWave <= kTlGather) {
          gathered = true;
          uint4* sc4 = (uint4*)((char*)S.memo + wave * kTlGatherBytes);
#pragma unroll
          for (int j = 0; j < GT; ++j) {
            if (selMask[j] != 0ull) {
              if ((selMask[j] >> lane) & 1ull) {
                const uint32_t nTok = tb[j] != 0ull ? (uint32_t)__builtin_ctzll(tb[j]) : 0u;
                const float lj = __uint_as_float((uint32_t)lmv[j].x);
                const unsigned long long cb = (unsigned long long)__double_as_longlong(cs[j]);
                const unsigned long long ab =
                    (unsigned long long)__double_as_longlong(amStep(amM, ev[j], (int)nTok, whichB ? blank : last));
                const unsigned long long lb = (unsigned long long)__double_as_longlong(lmM + (double)lj);
                const uint32_t w0 = nTok | (hypM << 8) | ((uint32_t)(myNewLane + 1) << 16);
                sc4[3 * myNew[j]] = make_uint4((uint32_t)cb, (uint32_t)(cb >> 32), (uint32_t)ab, (uint32_t)(ab >> 32));
                sc4[3 * myNew[j] + 1] = make_uint4((uint32_t)lb, (uint32_t)(lb >> 32), w0, me.sid);
                sc4[3 * myNew[j] + 2] = make_uint4((uint32_t)lmv[j].y, (uint32_t)lmv[j].x, 0u, 0u);
              }
            }
          }
          waveSync();
          if (lane < nNewWave) {
            const uint4 a = sc4[3 * lane], bq = sc4[3 * lane + 1], cq = sc4[3 * lane + 2];
            newStateOf(offW + lane, __longlong_as_double((long long)(((unsigned long long)a.y << 32) | a.x)), (int)(bq.z & 0xFFu),
                       (bq.z >> 8) & 0xFFu, __longlong_as_double((long long)(((unsigned long long)a.w << 32) | a.z)), cq.x,
                       __uint_as_float(cq.y), __longlong_as_double((long long)(((unsigned long long)bq.y << 32) | bq.x)), bq.w,
                       false, (int)((bq.z >> 16) & 0x7Fu) - 1, 0);
          }
        }
      }
      if (!gathered) {
#pragma unroll
      for (int j = 0; j < GT; ++j) {
        if (selMask[j] != 0ull) { /* (most positions of most frames have no survivor at all) */
          if ((selMask[j] >> lane) & 1ull) {
            /* (the position's token from its bit: reading tokId here is an LDS round trip per surviving position) */
            const int nTok = tb[j] != 0ull ? __builtin_ctzll(tb[j]) : 0;
            newState(offW + myNew[j], cs[j], nTok, hypM, ST ? amStep(amM, ev[j], nTok, whichB ? blank : last) : 0.0,
                     (uint32_t)lmv[j].y, __uint_as_float((uint32_t)lmv[j].x),
                     (ST && TL) ? lmM + (double)__uint_as_float((uint32_t)lmv[j].x) : 0.0);
          }
        }
      }
      }
    } else {
      if (surv >= 0) {
        const bool sB = ((selMask[0] >> lane) & 1ull) != 0ull, sR = ((selMask[1] >> lane) & 1ull) != 0ull;
        const int pln = pl >= 0 ? plNew : -1;
        SlRec r;
        r.nb = sR ? cs[1] : NEG;
        r.b = sB ? cs[0] : NEG;
        r.info = (uint32_t)last | ((uint32_t)(pln + 1) << 8) | ((sR ? hNB : kSlNoHyp) << 16) |
                 ((sB ? hB : kSlNoHyp) << 24);
        r.sid = me.sid;
        r.spar = me.spar;
        r.pad = TL ? me.pad : 0u;
        S.rec[q][surv] = r;
        if (TL) {
          S.tlIn[q][surv] = lIn;
        }
        if (mk) {
          atomOr64(&S.mask[q][surv], mk);
        }
        if (pln >= 0) {
          atomOr64(&S.cmask[q][pln], 1ull << last);
        }
        if (ST) {
          if (sR) {
            const double a = amStep(amR, eLast, last, prevR);
            S.amNB[q][surv] = a;
            histPT[hrow + hNB] = plainRec(parR, last);
            scoreRec(hrow + hNB, cs[1], a, lmR);
            if constexpr (TL) {
              S.lmNB[q][surv] = lmR;
            }
          }
          if (sB) {
            const double a = amM + eBlank;
            S.amB[q][surv] = a;
            histPT[hrow + hB] = plainRec(hypM, blank);
            scoreRec(hrow + hB, cs[0], a, lmM);
            if constexpr (TL) {
              S.lmB[q][surv] = lmM;
            }
          }
        } else {
          if (sR) {
            histPT[hrow + hNB] = make_int2((int)parR, last);
          }
          if (sB) {
            histPT[hrow + hB] = make_int2((int)hypM, blank);
          }
        }
      }
      if ((selMask[2] >> lane) & 1ull) {
        newState(offW + myNew[2], cs[2], last, hypB, ST ? amStep(amBv, eLast, last, blank) : 0.0, (uint32_t)lmLast.y,
                 __uint_as_float((uint32_t)lmLast.x), (ST && TL) ? lmBv + (double)__uint_as_float((uint32_t)lmLast.x) : 0.0);
      }
    }
    nStatePrev = nState;
    nState = nSurv + nNew;
    endBest = best;
    FLTX_SLPROF(4);
    ldsBarrier(); /* 3 */
    FLTX_SLPROF(5);
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
  if (isSvcW) {
    frames(SlParity<2>());
  } else if (isSelfW) {
    frames(SlParity<1>());
  } else {
    frames(SlParity<0>());
  }

  if (ST) {
    /* ---- park the beam for the next launch / prune / getBestHypothesis / decodeEnd: slots sorted by score (the
     * lane-per-slot engine's format), the last history row rewritten in that order ------------------------------- */
#ifndef FLTX_EMU
    __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory"); /* this wave's history records are in L2 */
#endif
    ldsBarrier();
    const int pe = T & 1;
    if (wave == 0 && T > 0) {
      /* states entered again in the last frame: the relink of the next frame never runs -- their child masks come
       * from maskTab now (the links are found again by the restore) */
      const int nevEnd = TL ? 0 : (int)S.row[pe].nev; /* (TL: no child masks -- the generic engine's table is the memo) */
      for (int e = 0; e < nevEnd; ++e) {
        if (lane == 0) {
          const int X = (int)S.evLane[e];
          S.mask[pe][X] |= loadCoherent64(&P.maskTab[(size_t)b * P.idCap + S.rec[pe][X].sid]);
        }
      }
      waveSync();
      const bool live = lane < nState && !dead;
      const SlRec me = S.rec[pe][live ? lane : 0];
      const uint32_t sNB = live ? (me.info >> 16) & 0xFFu : kSlNoHyp, sB = live ? me.info >> 24 : kSlNoHyp;
      const bool hasNB = sNB != kSlNoHyp, hasB = sB != kSlNoHyp;
      const unsigned long long kNB = hasNB ? f64Key(me.nb) : 0ull, kB = hasB ? f64Key(me.b) : 0ull;
      int rNB = 0, rB = 0;
      for (int i = 0; i < nState; ++i) {
        const unsigned long long k1 = ((unsigned long long)waveReadLane32((uint32_t)(kNB >> 32), i) << 32) |
                                      waveReadLane32((uint32_t)kNB, i);
        const unsigned long long k2 = ((unsigned long long)waveReadLane32((uint32_t)(kB >> 32), i) << 32) |
                                      waveReadLane32((uint32_t)kB, i);
        const uint32_t s1 = waveReadLane32(sNB, i), s2 = waveReadLane32(sB, i);
        const bool e1 = s1 != kSlNoHyp, e2 = s2 != kSlNoHyp;
        rNB += (e1 && (k1 > kNB || (k1 == kNB && s1 < sNB))) ? 1 : 0;
        rNB += (e2 && (k2 > kNB || (k2 == kNB && s2 < sNB))) ? 1 : 0;
        rB += (e1 && (k1 > kB || (k1 == kB && s1 < sB))) ? 1 : 0;
        rB += (e2 && (k2 > kB || (k2 == kB && s2 < sB))) ? 1 : 0;
      }
      const int nHyp = popc64(waveBallot(hasNB)) + popc64(waveBallot(hasB));
      const int64_t hlast = hbase + (int64_t)(frame0 + T) * K;
      const unsigned long long* hrec = (const unsigned long long*)(P.histPT + hlast);
      const unsigned long long recNB = hasNB ? loadCoherent64(hrec + sNB) : 0ull;
      const unsigned long long recB = hasB ? loadCoherent64(hrec + sB) : 0ull;
      const double aNB = S.amNB[pe][live ? lane : 0], aB = S.amB[pe][live ? lane : 0];
      double lNBo = 0.0, lBo = 0.0;
      if constexpr (TL) {
        lNBo = S.lmNB[pe][live ? lane : 0];
        lBo = S.lmB[pe][live ? lane : 0];
      }
      const unsigned long long mkv = S.mask[pe][live ? lane : 0];
      const uint32_t lastTok = me.info & 0xFFu;
      const uint32_t sparOut = me.sid == 0u ? kNoParent : me.spar;
      const int32_t edgeOut = me.sid == 0u ? 0 : (int32_t)lastTok;
#ifndef FLTX_EMU
      __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory"); /* every lane has its old records before any is rewritten */
#endif
      waveSync();
      auto put = [&](int r, double sc, double am, uint32_t tokpb, unsigned long long rec, double lmv) {
        const size_t g = (size_t)b * K + r;
        P.gScore[g] = sc;
        P.gAm[g] = am;
        P.gLm[g] = lmv;
        P.gState[g] = me.sid;
        P.gSPar[g] = sparOut;
        P.gSEdge[g] = edgeOut;
        P.gLex[g] = 0u;
        P.gLexMax[g] = 0.0f;
        P.gTokPb[g] = tokpb;
        if (!TL) {
          P.gMask[g] = mkv;
        }
        ((unsigned long long*)(P.histPT + hlast))[r] = rec;
        if (P.histS) {
          double* hs = P.histS + 3 * (hlast + r);
          hs[0] = sc;
          hs[1] = am;
          hs[2] = lmv;
        }
      };
      if (hasNB) {
        put(rNB, me.nb, aNB, lastTok, recNB, lNBo);
      }
      if (hasB) {
        put(rB, me.b, aB, (uint32_t)blank | kPrevBlank, recB, lBo);
      }
      if (lane == 0) {
        P.uttNBeam[b] = dead ? 0 : nHyp;
        P.uttFrame[b] = frame0 + T;
        P.uttTotal[b] = total0 + T;
        P.uttNextId[b] = (int32_t)S.scal[SL_NEXTID];
        P.uttStatus[b] = P.uttStatus[b] | (int32_t)S.scal[SL_STATUS] | (dead ? ST_SELECT_FALLBACK : 0);
      }
    }
  } else {
  /* ---- decodeEnd (LexiconFreeDecoder.cpp:127-158): finish() keeps the state, token = sil; the two
   * hypotheses of a state merge; sorted n-best (candidatesStore returnSorted) ------------------ */
  const int pe = T & 1;
  const int ff = T + 1;
  if (wave == 0 && !dead) {
    const bool live = lane < nState;
    const SlRec me = S.rec[pe][live ? lane : 0];
    const double nb = live ? me.nb : NEG, bb = live ? me.b : NEG;
    const bool whichB = bb > nb;
    double m = whichB ? bb : nb;
    const uint32_t hp = whichB ? (me.info >> 24) : ((me.info >> 16) & 0xFFu);
    double wFin = 0.0;
    if (TL) {
      /* lm->finish(state) (KenLM.cpp:77-83: the score of </s> in the state's context, a new state per state): both
       * hypotheses of a state add the same term; the best candidate of decodeEnd is a maximum again */
      wFin = lmW * (double)__uint_as_float((uint32_t)tokLm[(size_t)(live ? me.pad : 0u) * (size_t)tokStride + (size_t)N].x);
      m = m + wFin;
      const bool has = live && hp != kSlNoHyp && m == m;
      const unsigned long long k = waveMax64(has ? f64Key(m) : 0ull);
      endBest = k != 0ull ? f64FromKey(k) : 0.0;
    } else if (LA) { /* the best candidate of decodeEnd is the best hypothesis of the final beam */
      endBest = f64FromKey(S.mmaxKey[pe]);
    }
    const double thr = endBest - P.beamThreshold;
    const bool ok = live && m >= thr && (!TL || hp != kSlNoHyp);
    if (LA && ok) { /* the state's two hypotheses finish into one (LexiconFreeDecoder.cpp:127-158) */
      const double lo = TL ? (whichB ? nb : bb) + wFin : (whichB ? nb : bb);
      const uint32_t hl = whichB ? ((me.info >> 16) & 0xFFu) : (me.info >> 24);
      if (hl != kSlNoHyp && lo >= thr) {
        m = slLogAdd(m, lo);
      }
    }
    const unsigned long long key = ok ? f64Key(m) : 0ull;
    int rank = 0;
    for (int i = 0; i < nState; ++i) {
      const uint32_t lo = waveReadLane32((uint32_t)key, i), hi = waveReadLane32((uint32_t)(key >> 32), i);
      const unsigned long long k2 = ((unsigned long long)hi << 32) | lo;
      const uint32_t h2 = waveReadLane32(hp, i);
      rank += (k2 > key || (k2 == key && h2 < hp)) ? 1 : 0;
    }
    const unsigned long long okMask = waveBallot(ok);
    if (ok) { /* at most K states hold a hypothesis, so every candidate above the threshold stays */
      const size_t g = ((size_t)b * K + rank) * 3;
      P.outScores[g + 0] = m;
      P.outScores[g + 1] = 0.0; /* emitting-model score: the back-trace kernel fills it in */
      P.outScores[g + 2] = 0.0; /* ZeroLM; a token LM's score is re-accumulated by the back-trace kernel as well */
      P.histPT[hbase + (int64_t)ff * K + rank] = make_int2((int)hp, P.sil);
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
  }
  if (PROF && P.prof && tid == P.profThread) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      P.prof[(size_t)b * 8 + i] = acc[i];
    }
  }
}
#undef FLTX_SLPROF
