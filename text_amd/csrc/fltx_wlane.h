/*
 * fltx_wlane.h -- "lane = LM state" decode of a whole utterance for LARGE token sets (word pieces: 65 .. 16 384
 * tokens) with a token beam of at most 64: LexiconFreeDecoder + ZeroLM, max-merge, beam <= 64, offline.  Included
 * by fltx_kernels.h after fltx_slane.h, whose lane formulation, selection and history records it keeps (read the
 * head of that file first).  Same candidates, same merge groups, same selection as
 * LexiconFreeDecoder::decodeStep (LexiconFreeDecoder.cpp:30-125) with candidatesStore (Utils.h:146-225):
 * bit-identical n-best.
 *
 * What differs from fltx_slane.h, whose per-lane masks have one bit per TOKEN (<= 64 tokens):
 *   * the token beam (LexiconFreeDecoder.cpp:42-51: the beamSizeToken largest emissions of the row) is what the
 *     frame works with, and everything token-shaped is keyed by the POSITION of a token in the frame's list of
 *     at most 64: a lane's record carries its last token's position in the list of the frame that reads the
 *     record (0xFF: not in that frame's token beam -- the state can be neither repeated nor extended by it) next
 *     to the token itself; cmask[lane] has a bit per list position; the token waves' skip test is a bit test
 *     with their own position numbers.  The staging wave keeps a token -> position table (posOf, one byte per
 *     token) of the newest list, which the build step reads for the records it writes;
 *   * the token beams are found before the decode kernel starts, by a kernel of their own (fltx_tokbeam_kernel: one
 *     wave per row of N emissions, all rows of the batch in parallel): the histogram selection the frames use (a
 *     window over the float bits of the distance to the row's largest, the members of the one boundary bin ranked
 *     pairwise, ties to the lower token), the list written in token order, one 416-byte record per row; the staging
 *     wave of the decode kernel takes a record per frame, a frame ahead;
 *   * "this (LM state, token) edge had a child before" (LMState::child's memo, lm/LM.h:24-34) is a Bloom filter
 *     over (state id, token) in LDS -- two bits per edge in 512 Kbit -- instead of a 64-bit mask per lane: a
 *     hit, true or false, makes the next frame look the edge up -- first in an exact direct-mapped memo of the newest
 *     4 096 edges in LDS, then in the history rows (slReenter's scan, which finds the new record itself when the hit
 *     was false) --, so the filter only has to be free of false negatives.
 *
 * Roles and barriers as in fltx_slane.h: token waves (GT list positions each), one wave for the lanes' own groups
 * (blank, repeat + the parent state's extension, blank-then-last), the staging wave; three barriers per frame.
 */
#pragma once

constexpr int kWlMaxN = 16384;        /* tokens the position table covers */
constexpr int kWlBloomWords = 16384;  /* 64 KB = 2^19 bits */
constexpr int kWlRowRegs = 16;        /* row values a lane of the front end holds: one chunk of 1 024 tokens */
constexpr uint32_t kWlNoPos = 0xFFu;
constexpr int kWlEdgeSlots = 4096;    /* direct-mapped memo of the newest (state id, token) -> child state id edges, 32 KB */
constexpr uint32_t kWlNoSid = 0xFFFFFFFFu;

struct WlRow { /* what a frame needs to know about its emission row */
  double best;    /* the frame's best candidate (Utils.h:131-137) */
  double thr;     /* best - beamThreshold */
  double eBlank;  /* blank's emission, NaN when blank is not in the token beam (or the criterion has none) */
  int32_t nList;  /* tokens the token waves evaluate (in the token beam, not blank) */
  int32_t silPos; /* list position of sil, or a value no wave matches */
  uint32_t dead;  /* no candidate at all / not finite: the utterance goes to the general engines */
  uint32_t nev;   /* re-entry events recorded by the build of the previous frame */
};

struct WlaneLds {
  SlRec rec[2][64];                /* info: position of the last token in the reading frame's list | (parent lane + 1) << 8 |
                                      history slot of nb << 16 | of b << 24; pad: the last token */
  unsigned long long cmask[2][64]; /* list positions whose child state is in the beam and linked to this lane */
  uint32_t hist[2][kSlNB];
  double eTok[2][64];              /* emission of the listed tokens by list position, NaN past the list */
  uint32_t tokTok[2][64];          /* list position -> token */
  WlRow row[2];
  uint32_t off[32];
  int32_t newLane[64];
  uint32_t scal[16];
  unsigned long long bKey[kSlBCap];
  uint32_t bOrd[kSlBCap];
  uint32_t evLane[64], evSpar[64], evTok[64];
  uint32_t evSid[64];              /* the state's id when the edge memo still had it (no look-up in the history rows), else kWlNoSid */
  uint32_t scanMin, pad0;
  unsigned long long edge[kWlEdgeSlots]; /* bit 63 | parent id:23 << 37 | token:14 << 23 | child id:23 */
  uint32_t bloom[kWlBloomWords];
  uint8_t posOf[kWlMaxN];          /* token -> position in the newest list (the frame after the current one), 0xFF = not listed */
};

FLTX_DEV uint32_t wlHash(uint32_t sid, uint32_t tok, uint32_t seed) {
  uint32_t h = (sid * 0x9E3779B1u) ^ (tok * 0x85EBCA77u) ^ seed;
  h ^= h >> 15;
  h *= 0x2C1B3C6Du;
  h ^= h >> 12;
  return h;
}
/* marks the edge (state id, token) and tells whether it may have been marked before (no false negatives) */
FLTX_DEV bool wlEdgeSeen(WlaneLds& S, uint32_t sid, uint32_t tok) {
  const uint32_t h1 = wlHash(sid, tok, 0x1234567u), h2 = wlHash(tok, sid, 0x89ABCDEu);
  const uint32_t b1 = 1u << (h1 & 31u), b2 = 1u << (h2 & 31u);
  const uint32_t o1 = atomOr32(&S.bloom[(h1 >> 5) & (kWlBloomWords - 1)], b1);
  const uint32_t o2 = atomOr32(&S.bloom[(h2 >> 5) & (kWlBloomWords - 1)], b2);
  return (o1 & b1) != 0u && (o2 & b2) != 0u;
}

FLTX_DEV unsigned long long wlEdgePack(uint32_t psid, uint32_t tok, uint32_t csid) {
  return (1ull << 63) | ((unsigned long long)(psid & 0x7FFFFFu) << 37) | ((unsigned long long)(tok & 0x3FFFu) << 23) |
         (unsigned long long)(csid & 0x7FFFFFu);
}
FLTX_DEV uint32_t wlEdgeSlot(uint32_t psid, uint32_t tok) { return wlHash(psid, tok, 0x5bd1e995u) & (uint32_t)(kWlEdgeSlots - 1); }

/* Re-entry of LM states (see slReenter): the earliest history record {new state, parent id, token} names the
 * state of a lane whose edge the filter had seen; lanes whose parent id it is get their link back.  All waves. */
FLTX_DEV __attribute__((noinline)) void wlReenter(WlaneLds& S, const int2* histPT, int q, int nState, int64_t hbase,
                                                   int64_t nRec) {
  const int tid = (int)threadIdx.x, W = (int)blockDim.x;
  const int nev = (int)S.row[q].nev;
  ldsBarrier(); /* (everybody has read the count before the first one through resets it) */
  bool settled = false; /* the history stores of the last build have reached the L2 (needed by the look-ups only) */
  for (int e = 0; e < nev; ++e) {
    const int X = (int)S.evLane[e];
    const uint32_t ps = S.evSpar[e], n = S.evTok[e];
    const uint32_t known = S.evSid[e]; /* (uniform) */
    if (known != kWlNoSid) { /* the edge memo had the state's id: the record carries it already; its orphans: */
      if (tid < nState && tid != X && S.rec[q][tid].spar == known) {
        const uint32_t info = S.rec[q][tid].info;
        S.rec[q][tid].info = (info & ~0xFF00u) | ((uint32_t)(X + 1) << 8);
        if ((info & 0xFFu) < 64u) {
          atomOr64(&S.cmask[q][X], 1ull << (info & 0xFFu));
        }
      }
      continue;
    }
    if (!settled) { /* (waiting for a wave's stores costs 2 - 4 k clocks: only when the rows are read) */
      settled = true;
#ifndef FLTX_EMU
      __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
      ldsBarrier();
    }
    if (tid == 0) {
      S.scanMin = 0xFFFFFFFFu;
    }
    ldsBarrier();
    const unsigned long long* h = (const unsigned long long*)(histPT + hbase);
    uint32_t found = 0xFFFFFFFFu;
    for (int64_t i0 = tid; i0 < nRec; i0 += 8 * (int64_t)W) { /* eight loads in flight (one at a time, a scan of the
                                                                   rows of a long utterance is tens of microseconds) */
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
    const uint32_t sid = S.scanMin;
    if (sid != 0xFFFFFFFFu && tid == 0) {
      S.edge[wlEdgeSlot(ps, n)] = wlEdgePack(ps, n, sid); /* (the memo learns the edge, again or for the first time) */
    }
    if (sid != 0xFFFFFFFFu && sid != S.rec[q][X].sid) { /* (its own record: the filter's hit was a false one) */
      ldsBarrier();
      if (tid == 0) {
        S.rec[q][X].sid = sid;
      }
      if (tid < nState && tid != X && S.rec[q][tid].spar == sid) { /* orphans get their parent back */
        const uint32_t info = S.rec[q][tid].info;
        S.rec[q][tid].info = (info & ~0xFF00u) | ((uint32_t)(X + 1) << 8);
        if ((info & 0xFFu) < 64u) {
          atomOr64(&S.cmask[q][X], 1ull << (info & 0xFFu));
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

/* ---- front end: the token beams of all rows, a kernel of its own ------------------------------------------------ */
/* Which tokens a frame works with does not depend on the beam: the beamSizeToken largest emissions of every row of the
 * batch are found by fltx_tokbeam_kernel before the decode kernel starts -- one wave per row, all rows of all
 * utterances in parallel over the chip (inside the decode kernel the same selection was one wave's serial work per
 * frame: 15 k .. 23 k clocks against the 4 k of the frame's own) -- and left in HBM as one 416-byte record per row. */
struct WlTokRow {
  float e[64];       /* emissions of the listed tokens (in the token beam, not blank), in token order */
  uint16_t tok[64];  /* ... the tokens */
  float eBlank;      /* blank's emission, NaN when blank is not in the token beam (or the criterion has none) */
  float eSil;        /* sil's emission when it is in the token beam */
  uint32_t ek;       /* order key of the largest emission in the token beam other than sil's (blank's included), 0 = none */
  int32_t nList;
  int32_t silPos;    /* list position of sil, -4096 = not listed */
  uint32_t flags;    /* bit 0: sil is in the token beam; bit 1: the row cannot be cut (NaN, or more equal values at the
                        cut than the pairwise list holds): the utterance goes to the general engines */
  uint32_t pad[2];
};
static_assert(sizeof(WlTokRow) == 416, "one record per row");
constexpr int kWlShift = 18;    /* the counting pass after the coarse one: 32 bins per octave */
constexpr int kWlPairwise = 16; /* members of the boundary bin ranked against each other; beyond that the bin is looked at again */

struct WlFrontLds { /* per wave of the front-end kernel */
  uint32_t fhist[kSlNB];            /* counts of the row's values per bin */
  unsigned long long fKey[kSlBCap]; /* members of the boundary bin: value key << 32 | ~token */
  unsigned long long fSel[kWlMaxN / 64]; /* those of them the token beam takes: a bit per value of the row */
  uint32_t fScal[4];
};

FLTX_DEV void wlLoadChunk(const float* row, int N, int chunk, float (&v)[kWlRowRegs]) {
  const int lane = laneId();
#pragma unroll
  for (int k = 0; k < kWlRowRegs; ++k) {
    const int idx = chunk * (64 * kWlRowRegs) + k * 64 + lane;
    v[k] = idx < N ? row[idx] : -__builtin_huge_valf();
  }
}

/* The token beam of one row = its Kt largest values (LexiconFreeDecoder.cpp:42-51).  One wave.  A counting pass over
 * a window of the float bits of the distance to the row's largest (16 bins per octave, 2^-7 .. 2^9; the two ends of a
 * window would take hundreds of counts on one LDS address and are counted with ballots instead), one scan; a bin
 * with more than a handful of members is looked at again through the finest window that spans it; the members of
 * the bin that holds the Kt-th largest are then ranked against each other (value, then the lower token) by one lane
 * each, and the winners marked in a bitmap over the row.  The list is written in token order. */
FLTX_DEV void wlTokBeamRow(const DecodeParams& P, WlFrontLds& S, const float* row, WlTokRow* out, bool ctc) {
  const int lane = laneId();
  const int N = P.N, Kt = P.Kt < 64 ? P.Kt : 64;
  constexpr int CH = 64 * kWlRowRegs;
  const int nChunk = (N + CH - 1) / CH;
  const float NEGF = -__builtin_huge_valf();
  float v[kWlRowRegs];
  int lastChunk = -1;
  auto chunk = [&](int c) { /* -> v: this lane's sixteen values of chunk c (rows of up to 1 024 tokens: read once) */
    if (c != lastChunk) {
      lastChunk = c;
      wlLoadChunk(row, N, c, v);
    }
  };
  /* pass A: the row's largest value */
  bool anyNan = false;
  float mx = NEGF;
  for (int c = 0; c < nChunk; ++c) {
    chunk(c);
#pragma unroll
    for (int k = 0; k < kWlRowRegs; ++k) {
      anyNan = anyNan || !(v[k] == v[k]);
      mx = v[k] > mx ? v[k] : mx;
    }
  }
  const uint32_t mxKey = waveMax32((mx == mx && mx > NEGF) ? f32Key(mx) : 0u);
  const bool rowBad = waveBallot(anyNan) != 0ull || mxKey == 0u;
  const float rowMax = mxKey != 0u ? f32FromKey(mxKey) : 0.0f;
  auto binOf = [&](float x, int shift, int base) -> int { /* -inf (and NaN): not a candidate */
    const float d = rowMax - x;
    if (!(d == d) || !(x > NEGF)) {
      return kSlInvalid;
    }
    int qn = (int)(__float_as_uint(d > 0.0f ? d : 0.0f) >> shift) - base;
    qn = qn < 0 ? 0 : qn;
    return qn > kSlNB - 1 ? kSlNB - 1 : qn;
  };
  /* pass B: counts per bin and the bin of the Kt-th largest */
  int shift = kSlCoarseShift, base = kSlCoarseBase;
  int lim = -1, need = 0, nBnd = 0, bstar = -1;
  unsigned long long bLo = 0ull, bHi = 0x7FFFFFFFull;
  bool giveUp = rowBad;
  SlScan sc = {};
  while (!giveUp) {
    ((uint4*)S.fhist)[lane] = make_uint4(0u, 0u, 0u, 0u);
    waveSync();
    int nNear = 0, nFarV = 0;
    for (int c = 0; c < nChunk; ++c) {
      chunk(c);
#pragma unroll
      for (int k = 0; k < kWlRowRegs; ++k) {
        const int b = binOf(v[k], shift, base);
        nNear += b == 0 ? 1 : 0;
        nFarV += b == kSlNB - 1 ? 1 : 0;
        if (b > 0 && b < kSlNB - 1) {
          atomAdd32(&S.fhist[b], 1u);
        }
      }
    }
    nNear = (int)waveReadLane32((uint32_t)waveInclusiveScan(nNear), 63);
    nFarV = (int)waveReadLane32((uint32_t)waveInclusiveScan(nFarV), 63);
    if (lane == 0) {
      S.fhist[0] = (uint32_t)nNear;
      S.fhist[kSlNB - 1] = (uint32_t)nFarV;
    }
    waveSync();
    sc = slScan(S.fhist, Kt, false);
    if (sc.total <= Kt) { /* fewer values than the token beam takes: all of them */
      lim = kSlNB - 1;
      break;
    }
    need = Kt - sc.cum;
    if (sc.cnt == need) {
      lim = sc.bstar;
      break;
    }
    if (sc.cnt <= kWlPairwise || (bLo >= bHi && sc.cnt <= kSlBCap)) { /* the members of the boundary bin: the larger values, ties to the lower token */
      bstar = sc.bstar;
      lim = sc.bstar - 1;
      nBnd = sc.cnt;
      if (lane == 0) {
        S.fScal[0] = 0u;
      }
      for (int i = lane; i < 8 * nChunk; i += 64) { /* (the winners' bitmap: a bit per value of the row) */
        ((uint4*)S.fSel)[i] = make_uint4(0u, 0u, 0u, 0u);
      }
      waveSync();
      for (int c = 0; c < nChunk; ++c) {
        chunk(c);
#pragma unroll
        for (int k = 0; k < kWlRowRegs; ++k) {
          if (binOf(v[k], shift, base) == bstar) {
            const uint32_t tok = (uint32_t)(c * CH + k * 64 + lane);
            const uint32_t i = atomAdd32(&S.fScal[0], 1u);
            S.fKey[i] = ((unsigned long long)f32Key(v[k] + 0.0f) << 32) | (unsigned long long)(~tok);
          }
        }
      }
      waveSync();
      for (int m0 = 0; m0 < nBnd; m0 += 64) { /* member m0 + lane: its rank among the members */
        const int m = m0 + lane;
        const unsigned long long mine = S.fKey[m < nBnd ? m : 0];
        int rank = 0;
#pragma unroll 4
        for (int i = 0; i < nBnd; ++i) {
          rank += S.fKey[i] > mine ? 1 : 0;
        }
        if (m < nBnd && rank < need) {
          const uint32_t tok = ~(uint32_t)mine;
          atomOr64(&S.fSel[tok >> 6], 1ull << (tok & 63u));
        }
      }
      waveSync();
      break;
    }
    { /* too many in one bin: the finest window that spans its bracket */
      const unsigned long long vv = (unsigned long long)(sc.bstar + base);
      if (sc.bstar > 0 || base == 0) {
        const unsigned long long l2 = vv << shift;
        bLo = l2 > bLo ? l2 : bLo;
      }
      if (sc.bstar < kSlNB - 1) {
        const unsigned long long h2 = ((vv + 1ull) << shift) - 1ull;
        bHi = h2 < bHi ? h2 : bHi;
      }
      if (bLo >= bHi) { /* one value of the distance, and many of it: the next pass ranks them pairwise, or gives up */
        if (sc.cnt > kSlBCap) {
          giveUp = true;
          break;
        }
        continue;
      }
      int ns = 0;
      while (((bHi >> ns) - (bLo >> ns)) > (unsigned long long)(kSlNB - 1)) {
        ++ns;
      }
      shift = ns;
      base = (int)(bLo >> ns);
    }
  }
  /* pass C: the list, in token order */
  int nList = 0;
  float eBlankF = __builtin_nanf(""), eSilF = 0.0f;
  int silPos = -4096;
  bool silSel = false; /* sil is in the token beam (listed, unless it is the blank) */
  uint32_t ek = 0u;
  for (int c = 0; c < nChunk && !giveUp; ++c) {
    chunk(c);
    /* (the winners' words of this chunk: lane k holds word k) */
    const unsigned long long fsMine = nBnd > 0 ? S.fSel[c * kWlRowRegs + (lane & (kWlRowRegs - 1))] : 0ull;
#pragma unroll
    for (int k = 0; k < kWlRowRegs; ++k) {
      const int tok = c * CH + k * 64 + lane;
      const float x = v[k];
      const int b = binOf(x, shift, base);
      bool sel = b <= lim;
      if (nBnd > 0) {
        const unsigned long long w = ((unsigned long long)waveReadLane32((uint32_t)(fsMine >> 32), k) << 32) |
                                     (unsigned long long)waveReadLane32((uint32_t)fsMine, k);
        sel = sel || ((w >> lane) & 1ull) != 0ull;
      }
      const bool isBlank = ctc && tok == P.blank;
      const unsigned long long selAll = waveBallot(sel);
      if (selAll == 0ull) {
        continue;
      }
      const unsigned long long bal = waveBallot(sel && !isBlank);
      if (sel && isBlank) {
        eBlankF = x;
      }
      if (sel && tok != P.sil) {
        const uint32_t kk = f32Key(x);
        ek = kk > ek ? kk : ek;
      }
      if (sel && tok == P.sil) {
        eSilF = x;
        silSel = true;
      }
      if (sel && !isBlank) {
        const int pos = nList + wavePrefixCount(bal);
        out->e[pos] = x;
        out->tok[pos] = (uint16_t)tok;
        if (tok == P.sil) {
          silPos = pos;
        }
      }
      nList += popc64(bal);
    }
  }
  /* what one lane found, to all */
  ek = waveMax32(ek);
  const unsigned long long blankAt = waveBallot(eBlankF == eBlankF);
  const unsigned long long silAt = waveBallot(silSel);
  if (blankAt != 0ull) {
    eBlankF = __uint_as_float(waveReadLane32(__float_as_uint(eBlankF), __builtin_ctzll(blankAt)));
  }
  if (silAt != 0ull) {
    const int sl = __builtin_ctzll(silAt);
    eSilF = __uint_as_float(waveReadLane32(__float_as_uint(eSilF), sl));
    silPos = (int)waveReadLane32((uint32_t)silPos, sl);
  }
  if (lane == 0) {
    out->eBlank = eBlankF;
    out->eSil = eSilF;
    out->ek = ek;
    out->nList = nList;
    out->silPos = silPos;
    out->flags = (silAt != 0ull ? 1u : 0u) | (giveUp ? 2u : 0u);
  }
}

/* the front-end kernel: workgroup = utterance * tokRowBlocks + row block, (row block) * (waves per workgroup) + wave = row */
FLTX_DEV void wlTokBeamRows(const DecodeParams& P, char* smem) {
  const int wave = waveUniform(waveId());
  const int ub = (int)blockIdx.x / P.tokRowBlocks, rb = (int)blockIdx.x % P.tokRowBlocks;
  const int b = P.uttMap ? P.uttMap[ub] : ub;
  const int t = rb * ((int)blockDim.x >> 6) + wave;
  const int T = P.stepT ? P.stepT[b] : 0;
  if (t >= T) {
    return;
  }
  WlFrontLds& S = ((WlFrontLds*)smem)[wave];
  const float* row = P.emissions + P.emOff[b] + (size_t)t * P.N;
  WlTokRow* out = (WlTokRow*)P.tokRows + (P.histOff[b] / P.K + t);
  wlTokBeamRow(P, S, row, out, P.criterion == 1);
}

struct WlFront {
  double bestChain; /* best candidate of the newest staged frame */
  /* the record of the row the next call stages, loaded a frame ahead: lane l holds list position l */
  float pe;
  uint32_t ptok;
  float pBlank, pSil;
  uint32_t pEk, pFlags;
  int32_t pNList, pSilPos;
};
FLTX_DEV void wlRowPrefetch(const DecodeParams& P, WlFront& F, int64_t rowIdx, bool any) {
  const int lane = laneId();
  const WlTokRow* rec = (const WlTokRow*)P.tokRows + rowIdx;
  if (any) {
    F.pe = rec->e[lane];
    F.ptok = (uint32_t)rec->tok[lane];
    F.pBlank = rec->eBlank;
    F.pSil = rec->eSil;
    F.pEk = rec->ek;
    F.pFlags = rec->flags;
    F.pNList = rec->nList;
    F.pSilPos = rec->silPos;
  }
}
/* Stages the prefetched row into parity q: list (eTok / tokTok / posOf), blank's emission, the frame's best candidate.
 * One wave.  nListOld: the positions of parity q ^ 1's list are taken out of posOf first. */
FLTX_DEV void wlStageRow(const DecodeParams& P, WlaneLds& S, WlFront& F, int64_t rowIdxNext, bool anyNext, int q,
                         double silScore, int nListOld) {
  const int lane = laneId();
  if (lane < nListOld) {
    S.posOf[S.tokTok[q ^ 1][lane]] = (uint8_t)kWlNoPos;
  }
  waveSync();
  const int nList = F.pNList;
  const bool listed = lane < nList;
  S.eTok[q][lane] = listed ? (double)F.pe : __builtin_nan("");
  S.tokTok[q][lane] = listed ? F.ptok : 0u;
  if (listed) {
    S.posOf[F.ptok] = (uint8_t)lane;
  }
  /* best candidate of the frame: best hypothesis (= the last frame's best candidate) + best token, sil priced apart */
  const double mmax = F.bestChain;
  double best = 0.0;
  bool any = false;
  if (F.pEk != 0u) {
    best = mmax + (double)f32FromKey(F.pEk);
    any = true;
  }
  if (F.pFlags & 1u) {
    const double sS = (mmax + (double)F.pSil) + silScore;
    if (sS == sS && (!any || sS > best)) {
      best = sS;
      any = true;
    }
  }
  F.bestChain = best;
  if (lane == 0) {
    S.row[q].best = best;
    S.row[q].thr = best - P.beamThreshold;
    S.row[q].eBlank = (double)F.pBlank;
    S.row[q].nList = nList;
    S.row[q].silPos = F.pSilPos;
    S.row[q].dead = ((F.pFlags & 2u) || !any || !(best - best == 0.0)) ? 1u : 0u;
  }
  wlRowPrefetch(P, F, rowIdxNext, anyNext); /* the row after this one on its way into the registers */
}

template <int GT>
FLTX_DEV void wlaneUtterance(const DecodeParams& P, char* smem) {
  WlaneLds& S = *(WlaneLds*)smem;
  const int b = P.uttMap ? P.uttMap[blockIdx.x] : (int)blockIdx.x;
  const int W = (int)blockDim.x, tid = (int)threadIdx.x;
  const int lane = laneId(), wave = waveUniform(waveId());
  const int nW = W >> 6;
  const int selfWave = nW - 2;
  const int prepWave = nW - 1;
  const bool isSelfW = wave == selfWave, isSvcW = wave == prepWave;
#ifndef FLTX_EMU
  if (!(P.tune & 1) && (isSelfW || isSvcW)) { /* (see fltx_slane.h) */
    __builtin_amdgcn_s_setprio(3);
  }
#endif
  const int K = P.K, N = P.N;
  const bool ctc = P.criterion == 1;
  const int T = P.stepT ? P.stepT[b] : 0;
  const float* em = P.emissions ? P.emissions + P.emOff[b] : nullptr;
  const int64_t hbase = P.histOff[b];
  const double NEG = slNegInf();
  static_assert(GT >= 3, "the self wave keeps its three groups in the slot arrays");
  double silScore = P.silScore;
#ifndef FLTX_EMU
  __asm__ volatile("" : "+v"(silScore)); /* (see fltx_slane.h) */
#endif

  /* ---- decodeBegin (LexiconFreeDecoder.cpp:20-28): the root state ------------------ */
  for (int i = tid; i < 2 * 64; i += W) {
    ((unsigned long long*)S.cmask)[i] = 0ull;
  }
  for (int i = tid; i < 2 * kSlNB; i += W) {
    ((uint32_t*)S.hist)[i] = 0u;
  }
  for (int i = tid; i < kWlBloomWords / 4; i += W) {
    ((uint4*)S.bloom)[i] = make_uint4(0u, 0u, 0u, 0u);
  }
  for (int i = tid; i < kWlEdgeSlots / 2; i += W) {
    ((uint4*)S.edge)[i] = make_uint4(0u, 0u, 0u, 0u);
  }
  for (int i = tid; i < kWlMaxN / 16; i += W) {
    ((uint4*)S.posOf)[i] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
  }
  if (tid < 32) {
    S.off[tid] = 0u;
  }
  if (tid < 16) {
    S.scal[tid] = 0u;
  }
  if (tid == 0) {
    SlRec r;
    r.nb = 0.0;
    r.b = NEG;
    r.info = kWlNoPos | (0u << 8) | (0u << 16) | (kSlNoHyp << 24); /* (the position is filled in below) */
    r.sid = 0u;
    r.spar = 0x7FFFFFu;
    r.pad = (uint32_t)P.sil;
    S.rec[0][0] = r;
    S.row[0].nev = 0u;
    S.row[1].nev = 0u;
    S.row[0].dead = 0u;
    S.row[1].dead = 0u;
    P.histPT[hbase] = make_int2((int)kSlNoHyp, P.sil);
  }
  if (tid > 0 && tid < K) { /* unused slots of a row never look like the record of a new state (wlReenter) */
    P.histPT[hbase + tid] = make_int2((int)kSlNoHyp, -1);
  }
  ldsBarrier();
  WlFront F;
  F.bestChain = 0.0; /* decodeBegin: the root hypothesis, score 0 */
  F.pe = 0.0f;
  F.ptok = 0u;
  F.pBlank = 0.0f;
  F.pSil = 0.0f;
  F.pEk = 0u;
  F.pFlags = 0u;
  F.pNList = 0;
  F.pSilPos = -4096;
  const int64_t rowBase = hbase / K; /* this utterance's records of fltx_tokbeam_kernel */
  if (wave == prepWave) {
    wlRowPrefetch(P, F, rowBase, T > 0);
    if (T > 0) {
      wlStageRow(P, S, F, rowBase + 1, T > 1, 0, silScore, 0);
    } else if (lane == 0) {
      S.row[0].nList = 0;
    }
    waveSync();
    if (lane == 0) { /* the root's token (sil) in the first list */
      S.rec[0][0].info = (uint32_t)S.posOf[P.sil] | (0u << 8) | (0u << 16) | (kSlNoHyp << 24);
    }
  }
  ldsBarrier();

  int nState = 1;
  double endBest = 0.0; /* best hypothesis of the final beam (decodeEnd's threshold) */
  int winShift = kSlCoarseShift, winBase = kSlCoarseBase;
  bool dead = false; /* this utterance goes to the general engines */
  const int blank = P.blank;
  int2* const histPT = P.histPT;
  /* per-phase clocks of one thread (bench.py --profile: the marks of fltx_slane.h) */
  unsigned long long acc[8] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull};
  unsigned long long tPrev = devClock();
  const bool profMe = P.prof != nullptr && tid == P.profThread;
  auto mark = [&](int i) {
    if (profMe) {
      const unsigned long long t_ = devClock();
      acc[i] += t_ - tPrev;
      tPrev = t_;
    }
  };

  /* the first list position of this token wave, kept in a vector register: as a scalar the compiler derives sixteen more
   * scalars per frame from it (position numbers, 64-bit masks), which the frame loop has no scalar registers for --
   * it recomputed and spilled them at the head of every frame */
  int posBase = wave * GT;
#ifndef FLTX_EMU
  __asm__ volatile("" : "+v"(posBase));
#endif

  auto frameStep = [&](auto PT, auto RL, const int t) {
    constexpr int p = decltype(PT)::value, q = p ^ 1;
    constexpr bool isSelf = decltype(RL)::value == 1, isSvc = decltype(RL)::value == 2;
    const int frameOut = t + 1;
    const int64_t hrow = hbase + (int64_t)frameOut * K;
    /* ---- phase 1: own state, candidates, histogram ------------------------------------- */
    const double best = S.row[p].best, thr = S.row[p].thr;
    const int nList = S.row[p].nList, silPos = S.row[p].silPos;
    const uint32_t rowDead = S.row[p].dead;
    const uint32_t nev = S.row[p].nev;
    SlRec me = {};
    unsigned long long cm = 0ull;
    if (!isSvc) {
      me = S.rec[p][lane];
      cm = S.cmask[p][lane];
    }
    double ev[GT];
    uint32_t tk[GT];
    double eBlank = 0.0;
#pragma unroll
    for (int j = 0; j < GT; ++j) {
      ev[j] = 0.0;
      tk[j] = 0u;
    }
    if (isSelf) {
      eBlank = S.row[p].eBlank;
    } else if (!isSvc) {
#pragma unroll
      for (int j = 0; j < GT; ++j) {
        const int pos = posBase + j;
        ev[j] = pos < 64 ? S.eTok[p][pos < 64 ? pos : 0] : __builtin_nan("");
        tk[j] = S.tokTok[p][pos < 64 ? pos : 0];
      }
    }
    (void)nList;
#ifndef FLTX_EMU
    if (!isSvc) { /* (the loads stay above the rare branches that follow: see fltx_slane.h) */
      __asm__ volatile("" : "+v"(me.nb), "+v"(me.b), "+v"(me.info), "+v"(me.sid), "+v"(me.spar), "+v"(me.pad), "+v"(cm));
    }
#endif
    if (nev != 0u) { /* rare: edges the filter had seen were taken in the previous build */
      wlReenter(S, histPT, p, nState, hbase, (int64_t)frameOut * K);
      me = S.rec[p][lane];
      cm = S.cmask[p][lane];
    }
    if (rowDead) { /* nothing to extend with, not finite, or the token beam could not be cut: general path */
      dead = true;
      return;
    }
    const bool live = lane < nState && !isSvc;
    const double nb = live ? me.nb : NEG, bb = live ? me.b : NEG;
    const uint32_t lastPos = me.info & 0xFFu;
    const uint32_t last = me.pad;
    const int pl = live ? (int)((me.info >> 8) & 0xFFu) - 1 : -1;
    const uint32_t hypNB = (me.info >> 16) & 0xFFu, hypB = me.info >> 24;
    const bool whichB = bb > nb;
    const double m = whichB ? bb : nb;
    const uint32_t hypM = whichB ? hypB : hypNB;
    /* self wave: what its groups need beyond the lane's own record (second LDS round trip) */
    SlRec par = {};
    double eLast = 0.0;
    if (isSelf) {
      par = S.rec[p][pl >= 0 ? pl : 0];
      eLast = S.eTok[p][lastPos < 64u ? lastPos : 0u];
    }
    mark(0);
    double cs[GT];
    int cbin[GT];
    uint32_t parR = kSlNoHyp;
    if (isSvc) {
#pragma unroll
      for (int j = 0; j < GT; ++j) {
        cs[j] = NEG;
        cbin[j] = kSlInvalid;
      }
      /* the next frame's list (found by the front-end kernel, on its way since the last frame), position table and
       * best candidate, while the other waves evaluate this frame's candidates (everything written here belongs to
       * the next frame; the position table is read by the build below, after the second barrier, and by nothing
       * before that) */
      if (t + 1 < T) {
        wlStageRow(P, S, F, rowBase + t + 2, t + 2 < T, q, silScore, nList);
      }
    } else if (!isSelf) {
      /* positions this lane does not extend with here: its own last token's (the repeat and the blank-then-last case
       * belong to the self wave) and those whose child state holds a lane (that lane merges the extension into its
       * repeat).  A lane without a state skips all. */
      const unsigned long long skip = live ? (cm | (lastPos < 64u ? 1ull << lastPos : 0ull)) : ~0ull;
#pragma unroll
      for (int j = 0; j < GT; ++j) {
        const int pos = posBase + j;
        double c = m + ev[j]; /* NaN past the end of the list */
        if (pos == silPos) {
          c = c + silScore;
        }
        const bool ok = pos < 64 && ((skip >> (pos & 63)) & 1ull) == 0ull && c >= thr;
        cs[j] = c;
        cbin[j] = ok ? slBin<false>(best, c, winShift, winBase) : kSlInvalid;
      }
    } else {
      const bool lastOk = live && lastPos < 64u; /* the last token is in the frame's token beam (never blank) */
      const bool lastSil = (int)lastPos == silPos;
      /* (S, blank, true): LexiconFreeDecoder.cpp:86-97 -- eBlank is NaN when blank is not in the token beam */
      double cB = m + eBlank;
      if (blank == P.sil) {
        cB = cB + silScore;
      }
      const bool okB = ctc && live && cB >= thr;
      /* (S, last, false): the repeat (:98-110) and the parent state's extension by last (:69-85) */
      const uint32_t lastP = par.pad;
      const uint32_t h1 = (par.info >> 16) & 0xFFu, h2 = par.info >> 24;
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
      /* max-merge (Utils.h:194-196); a tie goes to the lower history slot */
      double cR = r0;
      parR = hypNB;
      const bool t1 = has1 & ((r1 > cR) | ((r1 == cR) & (h1 < parR)));
      cR = t1 ? r1 : cR;
      parR = t1 ? h1 : parR;
      const bool t2 = has2 & ((r2 > cR) | ((r2 == cR) & (h2 < parR)));
      cR = t2 ? r2 : cR;
      parR = t2 ? h2 : parR;
      const bool okR = lastOk && (has0 || has1 || has2) && cR >= thr;
      const bool okL = ctc && lastOk && hasB && ((cm >> (lastPos & 63u)) & 1ull) == 0ull && cL >= thr;
      cs[0] = cB;
      cs[1] = cR;
      cs[2] = cL;
      cbin[0] = okB ? slBin<false>(best, cB, winShift, winBase) : kSlInvalid;
      cbin[1] = okR ? slBin<false>(best, cR, winShift, winBase) : kSlInvalid;
      cbin[2] = okL ? slBin<false>(best, cL, winShift, winBase) : kSlInvalid;
#pragma unroll
      for (int j = 3; j < GT; ++j) {
        cs[j] = NEG;
        cbin[j] = kSlInvalid;
      }
    }
    if (wave == 0) { /* housekeeping for everybody: what this frame's build adds to */
      S.cmask[q][lane] = 0ull;
      if (lane < 32) {
        S.off[lane] = 0u;
      }
      if (lane == 0) {
        S.scal[SL_BCNT] = 0u;
      }
    }
#pragma unroll
    for (int j = 0; j < GT; ++j) {
      if (cbin[j] < kSlFar) {
        atomAdd32(&S.hist[p][cbin[j]], 1u);
      }
    }
    mark(1);
    ldsBarrier(); /* 1 */
    /* ---- phase 2: which candidates survive (Utils.h:200-220; as fltx_slane.h) ------------ */
    unsigned long long selMask[GT];
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
          /* (the members loop over the list here, not slRankBin's broadcast + ballot of the other lane engines: with it
           * this kernel -- which spills 460 scalar registers in its frame loop -- returned wrong survivors on the GPU for
           * token beams of 64 while the emulator, compiling the same source, agreed with the oracle; not understood, so
           * the loop that 600 GPU configurations have checked stays) */
#pragma unroll
          for (int j = 0; j < GT; ++j) {
            if (cbin[j] == sc.bstar) {
              const uint32_t i = atomAdd32(&S.scal[SL_BCNT], 1u);
              S.bKey[i] = f64Key(cs[j]);
              S.bOrd[i] = ((uint32_t)wave << 16) | ((uint32_t)j << 8) | (uint32_t)lane;
            }
          }
          ldsBarrier();
#pragma unroll
          for (int j = 0; j < GT; ++j) {
            if (cbin[j] == sc.bstar) {
              const unsigned long long k = f64Key(cs[j]);
              const uint32_t o = ((uint32_t)wave << 16) | ((uint32_t)j << 8) | (uint32_t)lane;
              int rank = 0;
              for (int i = 0; i < sc.cnt; ++i) {
                const unsigned long long k2 = S.bKey[i];
                rank += (k2 > k || (k2 == k && S.bOrd[i] < o)) ? 1 : 0;
              }
              take |= rank < need ? (1u << j) : 0u;
            }
          }
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
            cbin[j] = slBin<false>(best, cs[j], shift, base);
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
    mark(2);
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
      ((uint4*)S.hist[q])[lane] = make_uint4(0u, 0u, 0u, 0u); /* the other parity's histogram (last read a frame ago) */
    } else if (!isSelf) {
#pragma unroll
      for (int j = 0; j < GT; ++j) {
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
    mark(3);
    ldsBarrier(); /* 2 */
    /* ---- phase 3: every survivor is written by the lane that evaluated it ---------------- */
    const int nSurv = (int)S.scal[SL_NSURV], nHSurv = (int)S.scal[SL_NHSURV];
    const int offW = (int)S.off[wave], nNew = (int)S.off[selfWave + 1];
    const int myNewLane = S.newLane[lane];
    const int plNew = S.newLane[pl >= 0 ? pl : 0];
    /* a token's position in the NEXT frame's list (the staging wave has just written the table) */
    auto nextPosOf = [&](uint32_t tok) -> uint32_t { return (t + 1 < T && tok < (uint32_t)kWlMaxN) ? (uint32_t)S.posOf[tok] : kWlNoPos; };
    auto newState = [&](int idx, double c, uint32_t n, uint32_t np, uint32_t hp) {
      const int nl = nSurv + idx;
      const uint32_t hyp = (uint32_t)(nHSurv + idx);
      SlRec r;
      r.nb = c;
      r.b = NEG;
      r.info = np | ((uint32_t)(myNewLane + 1) << 8) | (hyp << 16) | (kSlNoHyp << 24);
      r.sid = (uint32_t)frameOut * (uint32_t)K + hyp;
      r.spar = me.sid;
      r.pad = n;
      const bool again = wlEdgeSeen(S, me.sid, n); /* this edge may have had a child before */
      const uint32_t eslot = wlEdgeSlot(me.sid, n);
      uint32_t known = kWlNoSid;
      if (again) { /* ... and the memo may still know which: the state keeps its id, the history rows are not searched */
        const unsigned long long cur = S.edge[eslot];
        if ((cur >> 23) == (wlEdgePack(me.sid, n, 0u) >> 23)) {
          known = (uint32_t)cur & 0x7FFFFFu;
          r.sid = known;
        }
      } else {
        S.edge[eslot] = wlEdgePack(me.sid, n, r.sid);
      }
      S.rec[q][nl] = r;
      if (myNewLane >= 0 && np < 64u) {
        atomOr64(&S.cmask[q][myNewLane], 1ull << np);
      }
      histPT[hrow + hyp] = make_int2((int)(hp | kSlNewFlag | (me.sid << 9)), (int)n);
      if (again) { /* it may have descendants in the beam */
        const uint32_t e = atomAdd32(&S.row[q].nev, 1u);
        S.evLane[e] = (uint32_t)nl;
        S.evSpar[e] = me.sid;
        S.evTok[e] = n;
        S.evSid[e] = known;
      }
    };
    if (isSvc) {
      /* (its part of the build went ahead of the second barrier) */
    } else if (!isSelf) {
      if (wave == 0 && lane >= nHSurv + nNew && lane < K) { /* unused slots of the history row: see wlReenter */
        histPT[hrow + lane] = make_int2((int)kSlNoHyp, -1);
      }
#pragma unroll
      for (int j = 0; j < GT; ++j) {
        if (selMask[j] != 0ull) {
          const uint32_t np = nextPosOf(tk[j]);
          if ((selMask[j] >> lane) & 1ull) {
            newState(offW + myNew[j], cs[j], tk[j], np, hypM);
          }
        }
      }
    } else {
      const uint32_t lastNext = nextPosOf(last);
      if (surv >= 0) {
        const bool sB = ((selMask[0] >> lane) & 1ull) != 0ull, sR = ((selMask[1] >> lane) & 1ull) != 0ull;
        const int pln = pl >= 0 ? plNew : -1;
        SlRec r;
        r.nb = sR ? cs[1] : NEG;
        r.b = sB ? cs[0] : NEG;
        r.info = lastNext | ((uint32_t)(pln + 1) << 8) | ((sR ? hNB : kSlNoHyp) << 16) | ((sB ? hB : kSlNoHyp) << 24);
        r.sid = me.sid;
        r.spar = me.spar;
        r.pad = last;
        S.rec[q][surv] = r;
        if (pln >= 0 && lastNext < 64u) {
          atomOr64(&S.cmask[q][pln], 1ull << lastNext);
        }
        if (sR) {
          histPT[hrow + hNB] = make_int2((int)parR, (int)last);
        }
        if (sB) {
          histPT[hrow + hB] = make_int2((int)hypM, blank);
        }
      }
      if ((selMask[2] >> lane) & 1ull) {
        newState(offW + myNew[2], cs[2], last, lastNext, hypB);
      }
    }
    nState = nSurv + nNew;
    endBest = best;
    mark(4);
    ldsBarrier(); /* 3 */
    mark(5);
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
  } else if (isSelfW) {
    frames(SlParity<1>());
  } else {
    frames(SlParity<0>());
  }

  /* ---- decodeEnd (LexiconFreeDecoder.cpp:127-158): finish() keeps the state, token = sil; the two
   * hypotheses of a state merge; sorted n-best (candidatesStore returnSorted) ------------------ */
  const int pe = T & 1;
  const int ff = T + 1;
  if (wave == 0 && !dead) {
    const bool live = lane < nState;
    const SlRec me = S.rec[pe][live ? lane : 0];
    const double nb = live ? me.nb : NEG, bb = live ? me.b : NEG;
    const bool whichB = bb > nb;
    const double m = whichB ? bb : nb;
    const uint32_t hp = whichB ? (me.info >> 24) : ((me.info >> 16) & 0xFFu);
    const double thr = endBest - P.beamThreshold;
    const bool ok = live && m >= thr;
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
      P.outScores[g + 2] = 0.0; /* ZeroLM */
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
  if (profMe) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      P.prof[(size_t)b * 8 + i] = acc[i];
    }
  }
}
