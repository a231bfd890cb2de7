/*
 * fltx_lane.h -- "lane = beam slot" frame step for the headline configuration:
 * LexiconFreeDecoder + ZeroLM, beam <= 64 and <= 64 tokens (C2: beam 50, 29
 * tokens), max-merge or logAdd (template parameter), included by fltx_kernels.h.
 *
 * Same candidates, same merge groups and the same selection as fltx_lean.h
 * (LexiconFreeDecoder.cpp:30-125), bit-identical results; what changes is the
 * shape of the work.  The frame step is a serial chain T long whose cost is the
 * number of instructions a wave issues between barriers, so:
 *   * every wave keeps the WHOLE beam in registers, hypothesis h in lane h
 *     (score, token, and the scores/tokens of the <= 3 related hypotheses it
 *     can merge with), and evaluates the candidates of "its" tokens (the
 *     tokens are dealt to the waves, see laneToken) with no memory access at
 *     all: e[n] is wave-uniform, everything else is lane-local;
 *   * the merge of LexiconFreeDecoder.cpp:101-103 needs no table: a group has
 *     <= 3 members -- the hypothesis of an LM state and its blank/non-blank
 *     twin ("mate"), plus the repeat of the child state's hypothesis -- and a
 *     group with a repeat is evaluated by the repeat's own lane, which carries
 *     the parent's and the parent's mate's scores in registers;
 *   * who is whose mate / parent is NOT searched per frame: the thread that
 *     builds new slot r records a descriptor of r's LM state in terms of the
 *     old beam ("same state as old slot o" or "new child (o, token)") and ORs
 *     bit r into relTab[descriptor]; next frame a lane reads the two masks of
 *     its own and its parent's descriptor and has mate / parent / parent's
 *     mate by ctz.  The one case this cannot see -- an LM state that dropped
 *     out of the beam and re-enters -- raises a flag and the next frame derives
 *     the relations from the state ids by lane broadcast instead;
 *   * histogram prefix, K-th-bin search and scatter positions are computed by
 *     every wave redundantly from the shared counts: no "wave 0 works, the
 *     rest waits" section and no barrier between prefix and scatter;
 *   * the ~K short-listed candidates become the next beam in two waves that
 *     run concurrently (what the next phase 1 reads / everything else), and
 *     the epilogue of the build (state masks, history records) is deferred to
 *     the next frame's first phase and done by the wave with the fewest tokens.
 * Three barriers per frame (the third is the row hand-over of the caller).
 */
#pragma once

constexpr int kLaneNB = 512; /* histogram bins: 8 per lane of the prefix scan */

struct LaneCarry {
  int oldKid;      /* dKid entry this lane set for the current beam, -1 none (kid wave only) */
  int64_t pendHb;  /* history offset of the beam built by the previous frame, -1 = nothing pending */
  int pendOldN;    /* size of the beam it was built from */
  bool repHere;    /* this wave evaluates the repeat group of this lane's slot (fixed for the launch) */
};

/* The repeat groups (one per slot) are spread over the waves other than the
 * last one, which has the deferred epilogue to do. */
FLTX_DEV void laneCarryInit(LaneCarry& c) {
  const int nW = (int)blockDim.x >> 6;
  const int lane = laneId();
  c.oldKid = -1;
  c.pendHb = -1;
  c.pendOldN = 0;
  c.repHere = nW == 1 ? true : (lane % (nW - 1)) == waveId();
}

/* beam record read by every wave at the start of a frame: one 16-byte LDS load
 * brings the score, the token and the two LM-state descriptors of a slot */
FLTX_DEV uint4 laneRec(double s, uint32_t tokpb, int D, int PD) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(s);
  return make_uint4((uint32_t)b, (uint32_t)(b >> 32), tokpb, (uint32_t)(D + 1) | ((uint32_t)(PD + 1) << 16));
}
FLTX_DEV double laneRecScore(const uint4& r) {
  return __longlong_as_double((long long)(((unsigned long long)r.y << 32) | r.x));
}

/* deferred epilogue of the previous frame's build (phase E2 of fltx_lean.h):
 * final child-edge masks of the new slots, the {parent, token} history records
 * as one coalesced store, and the grown masks of the old states.  Done by the
 * last wave, which has the fewest tokens to evaluate. */
FLTX_DEV void laneFlush(const DecodeParams& P, LaneLds& S, const FrameCtx& f, LaneCarry& c) {
  if (c.pendHb < 0) {
    return;
  }
  const int nW = (int)blockDim.x >> 6;
  const int lane = laneId();
  const int K = P.K;
  const int co = f.cur * K, no = (f.cur ^ 1) * K;
  if (waveId() == nW - 1) {
    const int r = lane;
    if (r < f.nBeam) {
      const int rs = S.eRep[r];
      S.bMask[co + r] = S.eBase[r] | (rs >= 0 ? S.addMask[rs] : 0ull);
      const int n = (int)(S.bTokPb[co + r] & 0x7FFFFFFFu);
      P.histPT[c.pendHb + r] = make_int2(S.bPar[r], n);
      if (P.histS) {
        double* hs = P.histS + 3 * (c.pendHb + r);
        hs[0] = S.bScore[co + r];
        hs[1] = S.bAm[co + r];
        hs[2] = 0.0;
      }
    }
    if (r < c.pendOldN) {
      const unsigned long long add = S.addMask[r];
      if (add != 0ull) {
        P.maskTab[(size_t)f.b * P.idCap + S.bState[no + r]] = S.bMask[no + r] | add;
      }
    }
    waveSync();
    if (r < K) {
      S.addMask[r] = 0ull;
    }
  }
  c.pendHb = -1;
}

/* token j of a wave: the waves other than the last share the first
 * (nW - 1) * GT tokens round-robin; the last wave, which also runs the deferred
 * epilogue and the look-up tables of the builders, takes what is left */
template <int GT>
FLTX_DEV int laneToken(int wave, int nW, int j) {
  if (nW == 1) {
    return j;
  }
  return wave < nW - 1 ? wave + j * (nW - 1) : (nW - 1) * GT + j;
}
template <int GT>
FLTX_DEV bool laneTokenValid(int wave, int nW, int n, int N) {
  return n < N && (nW == 1 || wave == nW - 1 || n < (nW - 1) * GT);
}

/* Top-nTok tokens of the emission row in erow buffer `par` (LexiconFreeDecoder.cpp:42-51:
 * by emission descending, ties to the lower index), by the waves first, first + 1, ...:
 * lane m holds e[m]; the number of lanes that beat token n is its position. */
FLTX_DEV void laneShortlist(const DecodeParams& P, const Ws& w, LaneLds& S, int par, int nTok, int first) {
  const int nW = (int)blockDim.x >> 6;
  const int wave = waveId(), m = laneId();
  if (wave < first) {
    return;
  }
  const float* e = w.erow + par * P.N;
  const float o = m < P.N ? e[m] : 0.0f;
  for (int n = wave - first; n < P.N; n += nW - first) {
    const float v = e[n];
    const unsigned long long beat = waveBallot(m < P.N && (o > v || (o == v && m < n)));
    const int rank = popc64(beat);
    if (m == 0 && rank < nTok) {
      S.tokIdx[par][rank] = (uint8_t)n;
      S.tokPos[par][n] = (uint8_t)rank;
      atomOr64(&S.tokMask[par], 1ull << n);
    }
  }
}

FLTX_DEV uint32_t laneBit(unsigned long long m, int n) { /* n is wave-uniform at the call sites */
  return (uint32_t)(m >> n) & 1u;
}

template <int GT, bool LOGADD, bool FULLTOK>
FLTX_DEV int runFrameLane(const DecodeParams& P, const Ws& w, LaneLds& S, FrameCtx& f, LaneCarry& c, int frameOut,
                          int rb, float nextVal, bool haveNext) {
  const int W = (int)blockDim.x;
  const int tid = (int)threadIdx.x;
  const int lane = laneId(), wave = waveId();
  const int nW = W >> 6;
  const int K = P.K, N = P.N, KN = K * N;
  const int co = f.cur * K, no = (f.cur ^ 1) * K;
  const bool ctc = P.criterion == 1;
  const int nBeam = f.nBeam;
  const int kWave = nW - 1;
  const int nTok = f.nTok;
  constexpr bool fullTok = FULLTOK; /* beamSizeToken >= N: position == token (compile time: the
                                       headline configuration pays nothing for token beams) */
  if (tid == 0 && !fullTok) {
    S.tokMask[rb ^ 1] = 0ull; /* the other parity's token list is rebuilt during this frame */
  }
  /* every way out of the frame parks the next emission row in the other erow
   * buffer and, with a token beam, lists its top tokens */
  auto parkRow = [&]() {
    if (haveNext && tid < N) {
      w.erow[(rb ^ 1) * N + tid] = nextVal;
    }
  };
  auto bail = [&]() {
    parkRow();
    if (haveNext && !fullTok) {
      ldsBarrier();
      laneShortlist(P, w, S, rb ^ 1, nTok, 0);
    }
    return 0;
  };
  /* ---- phase 1: beam into lanes, relations, evaluation, histogram ---------------- */
  laneFlush(P, S, f, c);
  if (nBeam == 0) {
    return bail();
  }
  const bool live = lane < nBeam;
  const int hc = live ? lane : 0;
  const uint4 meRec = S.bRec[co + hc];
  const double score = laneRecScore(meRec);
  const uint32_t tokpb = meRec.z;
  const int tok = (int)(tokpb & 0x7FFFFFFFu);
  const bool pb = (tokpb & kPrevBlank) != 0;
  const bool rp = live && !pb && !(ctc && tok == P.blank); /* has a repeat candidate */
  const int D = live ? (int)(meRec.w & 0xFFFFu) - 1 : -1;
  int mate = -1, par = -1, pm = -1;
  unsigned long long repMask = 0ull;
  if (S.sc[SC_RELSLOW] == 0) {
    const int PD = live ? (int)(meRec.w >> 16) - 1 : -1;
    const unsigned long long mD = D >= 0 ? w.relTab[D] : 0ull;
    const unsigned long long mP = PD >= 0 ? w.relTab[PD] : 0ull;
    repMask = D >= KN ? S.repTab[D - KN] : 0ull;
    const unsigned long long oth = mD & ~(1ull << lane);
    mate = oth ? __builtin_ctzll(oth) : -1;
    par = mP ? __builtin_ctzll(mP) : -1;
    const unsigned long long pr = mP & (mP - 1ull);
    pm = pr ? __builtin_ctzll(pr) : -1;
  } else {
    /* first frame of a launch, or an LM state re-entered the beam: compare ids */
    const uint32_t sid = live ? S.bState[co + lane] : 0xFFFFFFFEu;
    const uint32_t sp = live ? S.bSPar[co + lane] : 0xFFFFFFFDu;
    for (int h2 = 0; h2 < nBeam; ++h2) {
      const uint32_t s2 = waveReadLane32(sid, h2);
      mate = (s2 == sid && h2 != lane) ? h2 : mate;
      if (s2 == sp) {
        if (par < 0) {
          par = h2;
        } else if (pm < 0) {
          pm = h2;
        }
      }
    }
    for (int h2 = 0; h2 < nBeam; ++h2) {
      const int p2 = (int)waveReadLane32((uint32_t)par, h2);
      const uint32_t t2 = waveReadLane32(rp ? (uint32_t)tok : 0xFFFFFFFFu, h2);
      if (p2 == lane && t2 != 0xFFFFFFFFu) {
        repMask |= 1ull << t2;
      }
    }
  }
  if (wave == kWave) { /* what the builders of phase 3 look up */
    if (c.oldKid >= 0) {
      w.dKid[c.oldKid] = (int16_t)-1;
    }
    waveSync();
    int kid = -1;
    if (live) {
      S.dMate[lane] = mate;
      S.dPar[lane] = par;
      if (par >= 0) {
        kid = par * N + S.bSEdge[co + lane];
        w.dKid[kid] = (int16_t)lane;
      }
    }
    c.oldKid = kid;
  }
  const int mi = mate >= 0 ? mate : hc, pi = par >= 0 ? par : hc, qi = pm >= 0 ? pm : hc;
  const uint4 mRec = S.bRec[co + mi], pRec = S.bRec[co + pi], qRec = S.bRec[co + qi];
  const unsigned long long tokMaskV = fullTok ? (N >= 64 ? ~0ull : ((1ull << N) - 1ull)) : S.tokMask[rb];
  const float eTok = f.e[tok < N ? tok : 0];
  const int rTok = fullTok ? tok : (int)S.tokPos[rb][tok < N ? tok : 0]; /* position of my token (if listed) */
  const int nLane = fullTok ? lane : (int)S.tokIdx[rb][lane < nTok ? lane : 0];
  const float eLane = f.e[lane < nTok ? nLane : 0];
  const float eSil = f.e[P.sil];
  float eJ[GT]; /* emissions of this wave's tokens (wave-uniform), all in flight with the loads above */
  int nJ[GT];   /* the tokens themselves */
#pragma unroll
  for (int j = 0; j < GT; ++j) {
    const int r = laneToken<GT>(wave, nW, j);
    nJ[j] = fullTok ? r : (int)S.tokIdx[rb][r < nTok ? r : 0];
    eJ[j] = f.e[r < nTok ? nJ[j] : 0];
  }
  const double a0 = S.bScore[co];
  const double aLast = S.bScore[co + nBeam - 1];
  FLTX_PROF(6);
  /* best candidate of the frame: best hypothesis (slot 0, the beam is sorted)
   * with its best token.  fl(a0 + e) is monotone in e for a finite a0, so the
   * maximum over the tokens other than sil is a0 + max e: a 32-bit reduction. */
  double best = 0.0;
  bool any = false;
  if (a0 - a0 == 0.0) {
    uint32_t ek = 0u;
    if (lane < nTok && nLane != P.sil && eLane == eLane) {
      ek = f32Key(eLane);
    }
    ek = waveMax32(ek);
    if (ek != 0u) {
      best = a0 + (double)f32FromKey(ek);
      any = true;
    }
    const double sS = (a0 + (double)eSil) + P.silScore;
    if (laneBit(tokMaskV, P.sil) != 0u && sS == sS && (!any || sS > best)) {
      best = sS;
      any = true;
    }
  } else {
    unsigned long long bk = 0ull;
    if (lane < nTok) {
      const double s = leanScore(P, a0, nLane, (double)eLane);
      if (s == s) {
        bk = f64Key(s);
      }
    }
    bk = waveMax64(bk);
    any = bk != 0ull;
    best = f64FromKey(bk);
  }
  if (!any) {
    return bail();
  }
  const double thr = best - P.beamThreshold;
  /* Two-segment monotone binning of d = best - score over [0, range] (see
   * fltx_lean.h), in float: any function that is monotone in the score and the
   * same in every wave gives exact ranks.  A threshold far wider than the beam
   * (beamThreshold = 1e9, inf) would waste the bins on empty range: the binned
   * range is capped at 8 x the beam's spread + 64 and whatever lies beyond
   * shares the last bin (still monotone; the K-th best is never out there
   * unless fewer than K candidates are closer, and then they all survive). */
  float rangeF = (float)(best - thr);
  const float wideCap = (float)(a0 - aLast) * 8.0f + 64.0f;
  rangeF = rangeF < wideCap ? rangeF : wideCap;
  const float NFf = (float)((kLaneNB * 3) / 4);
  /* fine segment = a little more than the current beam's own spread (best to
   * worst score): the next beam's scores land there unless the frame is
   * unusual, and then the coarse segment still ranks them exactly */
  float cutF = (float)(a0 - aLast) * 1.25f + 1e-3f;
  cutF = cutF > rangeF * 0.125f ? cutF : rangeF * 0.125f;
  cutF = cutF < rangeF * 0.9f ? cutF : rangeF * 0.9f;
  float sF = NFf / cutF, sC = ((float)kLaneNB - NFf) / (rangeF - cutF);
  if (!(rangeF > 0.0f) || !(sF > 0.0f) || !(sF < 1e30f) || !(sC > 0.0f) || !(sC < 1e30f)) {
    cutF = __builtin_huge_valf(); /* degenerate range: everything lands in bin 0 */
    sF = 0.0f;
    sC = 0.0f;
  }
  FLTX_PROF(0);
  const bool owner = live && !(mate >= 0 && mate < lane); /* the lower slot of a pair owns the state's groups */
  const double mScore = laneRecScore(mRec), pScore = laneRecScore(pRec), qScore = laneRecScore(qRec);
  const int mtok = (int)(mRec.z & 0x7FFFFFFFu), ptok = (int)(pRec.z & 0x7FFFFFFFu), qtok = (int)(qRec.z & 0x7FFFFFFFu);
  const bool mpb = (mRec.z & kPrevBlank) != 0, ppb = (pRec.z & kPrevBlank) != 0, qpb = (qRec.z & kPrevBlank) != 0;
  /* tokens for which my / my mate's extension is a candidate of a group I
   * evaluate: not the own repeat (it keeps the LM state and belongs to another
   * group) and not a token repeated by a child state's hypothesis (that group
   * is evaluated by the child's lane below) */
  const unsigned long long allTok = tokMaskV;
  const bool rpM = mate >= 0 && !mpb && !(ctc && mtok == P.blank);
  const unsigned long long maskA = owner ? (allTok & ~repMask & ~(rp ? 1ull << tok : 0ull)) : 0ull;
  const unsigned long long maskB = (owner && mate >= 0) ? (allTok & ~repMask & ~(rpM ? 1ull << mtok : 0ull)) : 0ull;
  double cs[GT + 1];
  int cbin[GT + 1];
  uint32_t validBits = 0u, pickBits = 0u;
#pragma unroll
  for (int j = 0; j < GT; ++j) {
    const int r = laneToken<GT>(wave, nW, j); /* position in the token list, wave-uniform */
    const int n = nJ[j];
    cs[j] = 0.0;
    cbin[j] = 0;
    if (laneTokenValid<GT>(wave, nW, r, nTok)) {
      const double en = (double)eJ[j];
      double sA = score + en, sB = mScore + en;
      if (n == P.sil) {
        sA = sA + P.silScore;
        sB = sB + P.silScore;
      }
      const bool okA = laneBit(maskA, n) != 0u && (sA >= thr);
      const bool okB = laneBit(maskB, n) != 0u && (sB >= thr);
      const bool pickB = okB && (!okA || sB > sA); /* a tie goes to the lower slot: me */
      cs[j] = pickB ? sB : sA;
      if constexpr (LOGADD) { /* Utils.h:186-193: members folded in score-descending order; the
                                 survivor's back-pointer is the best member's either way */
        LeanGroup g;
        leanFold(true, okA, sA, (uint32_t)lane, 0u, okB, sB, (uint32_t)mi, 0u, false, 0.0, 0u, 0u, g);
        cs[j] = g.valid ? g.s : cs[j];
      }
      validBits |= ((okA || okB) ? 1u : 0u) << j;
      pickBits |= (pickB ? 1u : 0u) << j;
    }
  }
  uint32_t who = 0u; /* winner of my repeat's group: 0 parent, 1 parent's mate, 2 my repeat */
  { /* the group of my repeat: parent's new-token candidates + my repeat; or the orphan repeat */
    const int n = tok;
    const double en = (double)eTok;
    const bool here = rp && c.repHere && laneBit(tokMaskV, tok < N ? tok : 0) != 0u; /* (token lane-varying here) */
    double sR = score + en, sA = pScore + en, sB = qScore + en;
    if (n == P.sil) {
      sR = sR + P.silScore;
      sA = sA + P.silScore;
      sB = sB + P.silScore;
    }
    const bool newA = ctc ? (n != ptok || ppb) : (n != ptok);
    const bool newB = ctc ? (n != qtok || qpb) : (n != qtok);
    const bool okR = here && (sR >= thr);
    const bool okA = here && par >= 0 && newA && (sA >= thr);
    const bool okB = here && pm >= 0 && newB && (sB >= thr);
    /* max-merge (Utils.h:194-196): best member,
     * a tie goes to the earlier generated one = the lower slot */
    const bool tB = okB && (!okA || sB > sA); /* par < pm: a tie goes to the parent */
    double s = tB ? sB : sA;
    const int slot = tB ? qi : pi;
    const bool v = okA || okB;
    const bool tR = okR && (!v || sR > s || (sR == s && lane < slot));
    s = tR ? sR : s;
    who = tR ? 2u : (tB ? 1u : 0u);
    if constexpr (LOGADD) {
      LeanGroup g;
      leanFold(true, okA, sA, (uint32_t)pi, 0u, okB, sB, (uint32_t)qi, 0u, okR, sR, (uint32_t)lane, 0u, g);
      s = g.valid ? g.s : s;
    }
    cs[GT] = s;
    cbin[GT] = 0;
    validBits |= ((okA || okB || okR) ? 1u : 0u) << GT;
  }
#pragma unroll
  for (int j = 0; j <= GT; ++j) {
    if ((validBits >> j) & 1u) {
      const float d = (float)(best - cs[j]);
      const float x = d < cutF ? d * sF : NFf + (d - cutF) * sC;
      int bin = (x < (float)kLaneNB) ? (int)x : kLaneNB - 1; /* also catches inf / NaN */
      bin = bin < 0 ? 0 : bin;
      cbin[j] = bin;
      atomAdd32(&S.hist[bin], 1u);
    }
  }
  FLTX_PROF(1);
  ldsBarrier(); /* 1 */
  /* ---- phase 2: every wave: prefix of the counts, K-th best's bin, scatter -------- */
  parkRow(); /* frame t + 1's emissions: visible after barrier 2 */
  if (wave == (nW > 1 ? 1 : 0)) { /* the relation tables have been read by everyone */
    if (D >= 0) {
      w.relTab[D] = 0ull;
    }
    if (lane < K) {
      S.repTab[lane] = 0ull;
    }
    if (lane == 0) {
      S.sc[SC_RELSLOW] = 0;
    }
  }
  static_assert(kLaneNB == 512, "the prefix below handles 8 bins per lane");
  const uint4 cq0 = ((const uint4*)S.hist)[2 * lane], cq1 = ((const uint4*)S.hist)[2 * lane + 1];
  const int mineCnt = (int)(cq0.x + cq0.y + cq0.z + cq0.w + cq1.x + cq1.y + cq1.z + cq1.w);
  const int inc = waveInclusiveScan(mineCnt);
  int pre[9]; /* exclusive prefix of my 8 bins, and the inclusive total after them */
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
  int bstar = kLaneNB - 1, L = total;
  {
    int q = 7, cumAt = inc; /* first of my bins whose inclusive count reaches K */
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
  FLTX_PROF(2);
  /* this wave's copy of the prefixes, 16 bits each (L <= SCAP is checked below) */
  ((uint4*)(w.wcum + wave * (kLaneNB / 2)))[lane] =
      make_uint4((uint32_t)pre[0] | ((uint32_t)pre[1] << 16), (uint32_t)pre[2] | ((uint32_t)pre[3] << 16),
                 (uint32_t)pre[4] | ((uint32_t)pre[5] << 16), (uint32_t)pre[6] | ((uint32_t)pre[7] << 16));
  waveSync();
  if (L > P.SCAP) { /* degenerate score distribution: let the host use the general path */
    if (tid == 0) {
      atomOr32((uint32_t*)&S.sc[SC_STATUS], ST_SELECT_FALLBACK);
    }
    return bail(); /* (parks the row a second time: harmless) */
  }
  /* short-listed candidates of this lane, one per loop trip: about K of the
   * (GT + 1) * W candidate slots are, so the trip count is 1-2, not GT + 1 */
  uint32_t todo = 0u;
#pragma unroll
  for (int j = 0; j <= GT; ++j) {
    todo |= (((validBits >> j) & 1u) && cbin[j] <= bstar) ? (1u << j) : 0u;
  }
  while (waveBallot(todo != 0u) != 0ull) {
    if (todo != 0u) {
      const int j = __builtin_ctz(todo);
      todo &= todo - 1u;
      double sj = cs[GT];
      int bin = cbin[GT];
#pragma unroll
      for (int i = 0; i < GT; ++i) {
        sj = j == i ? cs[i] : sj;
        bin = j == i ? cbin[i] : bin;
      }
      const uint32_t lo = ((const uint16_t*)(w.wcum + wave * (kLaneNB / 2)))[bin];
      const uint32_t cnt = S.hist[bin];
      const uint32_t p = lo + atomAdd32(&S.tick[bin], 1u);
      int n, slot, rep;
      uint32_t flag, orphan = 0u;
      int r;
      if (j < GT) {
        r = laneToken<GT>(wave, nW, j);
        n = fullTok ? r : (int)S.tokIdx[rb][r < nTok ? r : 0];
        slot = ((pickBits >> j) & 1u) ? mi : lane;
        flag = (ctc && n == P.blank) ? 0u : kNewState;
        rep = lane;
      } else {
        r = rTok;
        n = tok;
        slot = who == 0u ? pi : (who == 1u ? qi : lane);
        flag = who < 2u ? kNewState : 0u;
        rep = par >= 0 ? par : lane;
        orphan = par >= 0 ? 0u : 1u;
      }
      const unsigned long long key = f64Key(sj);
      S.sEnt[p] = make_uint4((uint32_t)key, (uint32_t)(key >> 32), (uint32_t)(slot * nTok + r), lo | ((lo + cnt) << 16));
      /* group: representative slot, token, orphan flag; source slot | kNewState */
      S.sIdx[p] = (uint32_t)rep | ((uint32_t)n << 8) | (orphan << 16);
      S.sSrc[p] = (uint32_t)slot | flag;
    }
  }
  FLTX_PROF(3);
  ldsBarrier(); /* 2 */
  /* ---- phase 3: entry p ranks itself; rank < K builds beam slot `rank` ------------- */
  /* Only ~K entries exist, one wave's worth, and what a new slot needs falls in
   * two independent halves, so two waves build concurrently (on different
   * SIMDs): wave 0 writes what the next frame's first phase reads (score,
   * token, LM-state descriptors, relation masks), wave 1 everything the next
   * build and the history need (am, LM-state id, child-edge masks, parent).
   * The other waves reset the histogram. */
  const int nS = L < K ? L : K;
  const int64_t hbase = f.histBase + (int64_t)frameOut * K;
  const int bWave = nW > 1 ? 1 : 0;
  if (nW <= 2 || wave >= 2) {
    const int t0 = nW > 2 ? tid - 128 : tid, st = nW > 2 ? W - 128 : W;
    for (int i = t0; i < kLaneNB; i += st) {
      S.hist[i] = 0u;
      S.tick[i] = 0u;
    }
  }
  if (haveNext && !fullTok) { /* the next frame's token list, by the waves that do not build */
    laneShortlist(P, w, S, rb ^ 1, nTok, nW > 2 ? 2 : 0);
  }
  if (wave == 0 || wave == bWave) {
    for (int p = lane; p < L; p += 64) {
      const uint4 me = S.sEnt[p];
      const uint32_t gi = S.sIdx[p];
      const uint32_t src = S.sSrc[p];
      const unsigned long long k = ((unsigned long long)me.y << 32) | me.x;
      const uint32_t o = me.z;
      const int lo = (int)(me.w & 0xFFFFu), hi = (int)(me.w >> 16);
      /* exact rank = entries in better bins + members of my bin that precede me;
       * a bin's members are contiguous after the counting sort.  Four entries per
       * round trip: the loop length is the largest bin of the wave over four. */
      int rank = lo;
      for (int q = lo; q < hi; q += 4) {
        uint4 e[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          e[i] = S.sEnt[q + i < L ? q + i : L - 1];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const unsigned long long k2 = ((unsigned long long)e[i].y << 32) | e[i].x;
          rank += (q + i < hi && (k2 > k || (k2 == k && e[i].z < o))) ? 1 : 0;
        }
      }
      FLTX_PROF(7);
      if (rank >= K) {
        continue;
      }
      const int h = (int)(src & 0x7FFFFFFFu);
      const int rep = (int)(gi & 0xFFu), n = (int)((gi >> 8) & 0xFFu);
      const bool orphan = (gi >> 16) != 0u;
      const bool newState = (src & kNewState) != 0u;
      const bool blankTok = ctc && n == P.blank;
      const uint32_t ktp = (uint32_t)n | ((!orphan && blankTok) ? kPrevBlank : 0u);
      /* old-beam slot that represents the new slot's LM state, -1 = a state that
       * is not in the old beam */
      const int kid = newState ? (int)w.dKid[rep * N + n] : -1;
      const int same = newState ? kid : h;
      const int samec = same >= 0 ? same : 0;
      const int sm = S.dMate[samec];
      const int repSlot = same < 0 ? -1 : ((sm >= 0 && sm < same) ? sm : same);
      if (wave == 0) {
        /* descriptor of the new slot's LM state and of its parent state, in terms
         * of the old beam: equal descriptors <=> same state (see the header) */
        int Dn, PDn;
        if (repSlot >= 0) {
          const int po = S.dPar[repSlot];
          Dn = KN + repSlot;
          PDn = po >= 0 ? KN + po : -1;
        } else {
          Dn = rep * N + n;
          PDn = KN + rep;
        }
        atomOr64(&w.relTab[Dn], 1ull << rank);
        if (PDn >= 0 && !(ktp & kPrevBlank) && !blankTok) {
          atomOr64(&S.repTab[PDn - KN], 1ull << n);
        }
        const double sc = f64FromKey(k);
        S.bRec[no + rank] = laneRec(sc, ktp, Dn, PDn);
        S.bScore[no + rank] = sc;
        S.bTokPb[no + rank] = ktp;
      }
      if (wave == bWave) {
        const uint32_t sparRep = S.bSPar[co + rep], sidRep = S.bState[co + rep];
        const int32_t sedgeRep = S.bSEdge[co + rep];
        const unsigned long long maskRep = S.bMask[co + rep];
        const double amH = S.bAm[co + h];
        const float eN = f.e[n];
        const uint32_t sidS = S.bState[co + samec];
        const unsigned long long maskS = S.bMask[co + samec];
        const bool keep = orphan || blankTok; /* the new slot stays in rep's LM state */
        const uint32_t kp = keep ? sparRep : sidRep;
        const uint32_t ke = keep ? (uint32_t)sedgeRep : (uint32_t)n;
        double am = amH + (double)eN;
        if (f.useTrans) { /* ASG: transition enters am only (LexiconFreeDecoder.cpp:59-64) */
          const int prevTok = (int)(S.bTokPb[co + h] & 0x7FFFFFFFu);
          am = amH + ((double)eN + (double)P.transitions[(size_t)n * N + prevTok]);
        }
        uint32_t sid;
        unsigned long long base;
        if (same >= 0) { /* the state is in the old beam: take its id from that slot */
          sid = sidS;
          base = maskS;
        } else if ((maskRep >> n) & 1ull) { /* existed, dropped out: rare re-entry */
          sid = loadCoherent32(&P.childTab[((size_t)f.b * P.idCap + kp) * N + n]);
          base = loadCoherent64(&P.maskTab[(size_t)f.b * P.idCap + sid]);
          S.sc[SC_RELSLOW] = 1; /* its children may be in the beam: relations by id next frame */
        } else { /* first time this state is materialised */
          sid = allocStateId(P, f.b, atomAdd32((uint32_t*)&S.sc[SC_NEXTID], 1u), kp, n, f.clock,
                             (uint32_t*)&S.sc[SC_STATUS]);
          base = 0ull;
          atomOr64(&S.addMask[rep], 1ull << n);
          P.childTab[((size_t)f.b * P.idCap + kp) * N + n] = sid;
          P.maskTab[(size_t)f.b * P.idCap + sid] = 0ull;
        }
        S.bAm[no + rank] = am;
        S.bState[no + rank] = sid;
        S.bSPar[no + rank] = kp;
        S.bSEdge[no + rank] = (int32_t)ke;
        S.eBase[rank] = base;
        S.eRep[rank] = repSlot;
        S.bPar[rank] = h;
      }
    }
  }
  c.pendHb = hbase;
  c.pendOldN = nBeam;
  FLTX_PROF(4);
  return nS; /* the caller's row hand-over barrier closes the frame */
}
