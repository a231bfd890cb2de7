/*
 * fltx_lean.h -- the low-latency frame step for the headline configuration:
 * LexiconFreeDecoder + ZeroLM (CTC or ASG), included by fltx_kernels.h.
 *
 * Same algorithm as the dense merge of fltx_kernels.h (see the comment above
 * denseEval): groups of candidates are enumerated instead of hashed.  What
 * changes is where the data lives and how many instructions a wave issues,
 * because the frame step is a serial dependency chain T long and with one
 * wave per SIMD its INSTRUCTION LATENCY, not its byte count, sets the
 * throughput (measured: ~10 clocks per issued instruction):
 *   * the group -> (state slot, token) mapping of a thread is fixed for the
 *     launch (group g = slot * nTok + r, orphan repeats behind them), so the
 *     integer divisions happen once, not per frame;
 *   * a thread keeps the <= GMAX groups it evaluates in REGISTERS through the
 *     whole prune (score, source, order) -- no candidate records in LDS;
 *   * with ZeroLM the frame's best candidate is known before any group is
 *     evaluated (best hypothesis + best token, the beam is kept sorted), so the
 *     threshold filter is applied during the single evaluation pass;
 *   * the "same LM state" / "parent LM state" relations of the <= 64 beam
 *     slots are found by lane broadcast (v_readlane), the slot range split
 *     across the waves;
 *   * one histogram pass over register-resident scores finds the bin of the
 *     K-th best; entries of that bin and better ones (K + a few) form a
 *     short-list in LDS, chained per bin, and the exact rank of an entry is
 *     (entries in better bins, from the histogram prefix) + (entries of its own
 *     bin that precede it, a chain of typically 1-3) -- O(1), not O(K);
 *   * the thread that owns short-list entry of rank r builds beam slot r.
 * Six barriers per frame.  Results are bit-identical to the generic path
 * (tests/test_gpu_parity.py runs both against the reference golden vectors).
 */
#pragma once

/* the lean steps normally run with their workspace in LDS; beams too large for a
 * CU's LDS run the same code over an HBM workspace, where a barrier must also
 * publish / re-read memory at agent scope (wsBarrier) */
FLTX_DEV void leanBarrier(const DecodeParams& P) {
  if (P.gws != nullptr) {
    wsBarrier(P);
  } else {
    ldsBarrier();
  }
}

struct LeanGroup {
  double s;
  uint32_t src; /* parent slot | kNewState */
  uint32_t ord;
  bool valid;
};

/* per-thread mapping of its GMAX groups; fixed for the launch */
template <int GMAX>
struct LeanMap {
  int rep[GMAX]; /* beam slot (state representative, or the orphan's own slot) */
  int r[GMAX];   /* position in the token short-list */
  int kind[GMAX]; /* 0 none, 1 (state, token) group, 2 orphan repeat */
};

template <int GMAX>
FLTX_DEV void leanMapInit(const DecodeParams& P, int nTok, LeanMap<GMAX>& m) {
  const int W = (int)blockDim.x;
  const int nG = P.K * nTok;
#pragma unroll
  for (int j = 0; j < GMAX; ++j) {
    const int g = (int)threadIdx.x + j * W;
    m.kind[j] = 0;
    m.rep[j] = 0;
    m.r[j] = 0;
    if (g < nG) {
      m.kind[j] = 1;
      m.rep[j] = g / nTok;
      m.r[j] = g - m.rep[j] * nTok;
    } else if (g < nG + P.K) {
      m.kind[j] = 2;
      m.rep[j] = g - nG;
    }
  }
}

/* fold up to three members in (score desc, order asc) order: max, or
 * left-to-right log-add (Utils.h:186-193).  Inactive members have ok == false. */
FLTX_DEV void leanFold(bool logAdd, bool ok0, double s0, uint32_t o0, uint32_t c0, bool ok1, double s1,
                       uint32_t o1, uint32_t c1, bool ok2, double s2, uint32_t o2, uint32_t c2,
                       LeanGroup& g) {
  if (!logAdd) { /* max-merge: best member only (Utils.h:194-196) */
    bool ok = ok0;
    double s = s0;
    uint32_t o = o0, c = c0;
    if (ok1 && (!ok || s1 > s || (s1 == s && o1 < o))) {
      ok = true;
      s = s1;
      o = o1;
      c = c1;
    }
    if (ok2 && (!ok || s2 > s || (s2 == s && o2 < o))) {
      ok = true;
      s = s2;
      o = o2;
      c = c2;
    }
    g.valid = ok;
    g.s = s;
    g.src = c;
    g.ord = o;
    return;
  }
  /* order the three by (valid first, score desc, ord asc) with a 3-swap network */
#define FLTX_BEFORE(ka, sa, oa, kb, sb, ob) ((ka) && (!(kb) || (sa) > (sb) || ((sa) == (sb) && (oa) < (ob))))
#define FLTX_CSWAP(ka, sa, oa, ca, kb, sb, ob, cb)            \
  if (!FLTX_BEFORE(ka, sa, oa, kb, sb, ob) && (kb)) {          \
    const bool tk = ka; ka = kb; kb = tk;                      \
    const double ts = sa; sa = sb; sb = ts;                    \
    const uint32_t to = oa; oa = ob; ob = to;                  \
    const uint32_t tc = ca; ca = cb; cb = tc;                  \
  }
  FLTX_CSWAP(ok0, s0, o0, c0, ok1, s1, o1, c1)
  FLTX_CSWAP(ok1, s1, o1, c1, ok2, s2, o2, c2)
  FLTX_CSWAP(ok0, s0, o0, c0, ok1, s1, o1, c1)
#undef FLTX_CSWAP
#undef FLTX_BEFORE
  g.valid = ok0;
  g.s = s0;
  g.src = c0;
  g.ord = o0;
  if (ok0) {
    double acc = s0;
    if (ok1) {
      const double mx = acc > s1 ? acc : s1, mn = acc > s1 ? s1 : acc;
      acc = mx + log1p(exp(mn - mx));
    }
    if (ok2) {
      const double mx = acc > s2 ? acc : s2, mn = acc > s2 ? s2 : acc;
      acc = mx + log1p(exp(mn - mx));
    }
    g.s = acc;
  }
}

/* candidate score of a hypothesis with score a taking token n
 * (LexiconFreeDecoder.cpp:64-67; ZeroLM adds lmWeight * 0.0f, which leaves
 * every finite score unchanged) */
FLTX_DEV double leanScore(const DecodeParams& P, double a, int n, double en) {
  const double s = a + en;
  const double s2 = s + P.silScore;
  return n == P.sil ? s2 : s;
}

/* evaluate group j of this thread entirely in registers (see denseEval) */
FLTX_DEV void leanEval(const DecodeParams& P, const Ws& w, const FrameCtx& f, int kind, int rep, int r,
                       double thr, LeanGroup& out) {
  const bool ctc = P.criterion == 1;
  const int co = f.cur * P.K;
  out.valid = false;
  out.s = 0.0;
  out.src = 0;
  out.ord = 0;
  if (kind == 0 || rep >= f.nBeam) {
    return;
  }
  if (kind == 2) { /* orphan repeat of slot rep */
    const uint32_t tp = w.bTokPb[co + rep];
    const int t = (int)(tp & 0x7FFFFFFFu);
    if ((tp & kPrevBlank) || (ctc && t == P.blank) || w.dPar[rep] >= 0 || !w.dIn[t]) {
      return;
    }
    int rr = t;
    if (f.nTok != P.N) {
      for (rr = 0; rr < f.nTok && w.tokIdx[rr] != t; ++rr) {
      }
    }
    const double s = leanScore(P, w.bScore[co + rep], t, (double)f.e[t]);
    if (s >= thr) { /* false for NaN */
      out.valid = true;
      out.s = s;
      out.src = (uint32_t)rep;
      out.ord = (uint32_t)(rep * f.nTok + rr);
    }
    return;
  }
  if (r >= f.nTok) {
    return;
  }
  const int mate = w.dMate[rep];
  if (mate >= 0 && mate < rep) {
    return; /* the lower slot of a pair owns the state's groups */
  }
  const int n = (f.nTok == P.N) ? r : w.tokIdx[r];
  const double en = (double)f.e[n];
  const bool isBlank = ctc && n == P.blank;
  const uint32_t flag = isBlank ? 0u : kNewState;
  /* loads are unconditional (clamped indices) so they can all be in flight */
  const int mi = mate >= 0 ? mate : rep;
  const int hr = isBlank ? -1 : (int)w.dRep[rep * P.N + n];
  const int ri = hr >= 0 ? hr : rep;
  const uint32_t tpA = w.bTokPb[co + rep], tpB = w.bTokPb[co + mi];
  const double aA = w.bScore[co + rep], aB = w.bScore[co + mi], aR = w.bScore[co + ri];
  const int ptA = (int)(tpA & 0x7FFFFFFFu), ptB = (int)(tpB & 0x7FFFFFFFu);
  const bool pbA = (tpA & kPrevBlank) != 0, pbB = (tpB & kPrevBlank) != 0;
  const double sA = leanScore(P, aA, n, en), sB = leanScore(P, aB, n, en), sR = leanScore(P, aR, n, en);
  const bool newA = isBlank || (ctc ? (n != ptA || pbA) : (n != ptA));
  const bool newB = isBlank || (ctc ? (n != ptB || pbB) : (n != ptB));
  const bool okA = newA && (sA >= thr);
  const bool okB = mate >= 0 && newB && (sB >= thr);
  const bool okR = hr >= 0 && (sR >= thr);
  leanFold(P.logAdd != 0, okA, sA, (uint32_t)(rep * f.nTok + r), (uint32_t)rep | flag, okB, sB,
           (uint32_t)(mi * f.nTok + r), (uint32_t)mi | flag, okR, sR, (uint32_t)(ri * f.nTok + r),
           (uint32_t)ri, out);
}

template <int GMAX>
FLTX_DEV int runFrameLean(const DecodeParams& P, const Ws& w, FrameCtx& f,
                          const LeanMap<(GMAX < 255 ? GMAX : 1)>& map,
                          int frameOut) {
  const int W = (int)blockDim.x;
  const int tid = (int)threadIdx.x;
  const int co = f.cur * P.K, no = (f.cur ^ 1) * P.K;
  const bool ctc = P.criterion == 1;
  const int K = P.K;
  const int lane = laneId(), wave = waveId();
  const int nW = (W + 63) >> 6;
  /* ---- phase A: clears, per-wave partial relations, frame best ---------------- */
  for (int i = tid; i < f.nBeam * P.N; i += W) {
    w.dRep[i] = (int16_t)-1;
    w.dKid[i] = (int16_t)-1;
  }
  for (int i = tid; i < f.nBeam; i += W) {
    w.addMask[i] = 0ull;
  }
  for (int i = tid; i < P.NB; i += W) {
    w.hist[FLTX_HB(i)] = 0;
  }
  for (int n = tid; n < P.N; n += W) {
    w.dIn[n] = (f.nTok == P.N) ? 1 : 0;
  }
  if (f.nTok < P.N) {
    tokenShortlist(P, w, f.e, f.nTok);
  }
  if (tid == 0) {
    w.sc[SC_NSMALL] = 0;
    w.sc[SC_BSTAR] = P.NB - 1;
    w.sc[SC_CUM] = 0;
    w.red[0] = 0ull;
  }
  const bool laneBeam = f.nBeam <= 64;
  const int relStride = (f.nBeam + 63) & ~63;
  int relParts = laneBeam ? 0 : W / relStride; /* (pMate / pPar hold 16 x 64 partial results) */
  relParts = relParts * relStride > 16 * 64 ? (16 * 64) / relStride : relParts;
  if (laneBeam) {
    /* every wave holds the beam's state ids in its lanes and scans its share of
     * the slots by lane broadcast; partial results go to this wave's row */
    const uint32_t sid = lane < f.nBeam ? w.bState[co + lane] : 0xFFFFFFFEu;
    const uint32_t sp = lane < f.nBeam ? w.bSPar[co + lane] : 0xFFFFFFFDu;
    const int per = (f.nBeam + nW - 1) / nW;
    const int lo = wave * per;
    int hi = lo + per;
    hi = hi > f.nBeam ? f.nBeam : hi;
    int mate = -1, par = -1;
    for (int h2 = lo; h2 < hi; ++h2) {
      const uint32_t s2 = waveReadLane32(sid, h2);
      mate = (s2 == sid && h2 != lane) ? h2 : mate;
      par = (s2 == sp && par < 0) ? h2 : par;
    }
    w.pMate[wave * 64 + lane] = mate;
    w.pPar[wave * 64 + lane] = par;
  } else if (relParts >= 2) {
    /* beams of 65 .. W / 2: a slot's scan over the beam is split between relParts threads (rows of relStride partial
     * results, combined after the barrier like the per-wave rows above); the loads of eight steps go out together */
    const int part = tid / relStride, h = tid - part * relStride;
    if (part < relParts && h < f.nBeam) {
      const uint32_t sid = w.bState[co + h];
      const uint32_t sp = w.bSPar[co + h];
      const int per = (f.nBeam + relParts - 1) / relParts;
      const int lo = part * per;
      int hi = lo + per;
      hi = hi > f.nBeam ? f.nBeam : hi;
      int mate = -1, par = -1;
#pragma unroll 8
      for (int h2 = lo; h2 < hi; ++h2) {
        const uint32_t s2 = w.bState[co + h2];
        mate = (s2 == sid && h2 != h) ? h2 : mate;
        par = (s2 == sp && par < 0) ? h2 : par;
      }
      w.pMate[part * relStride + h] = mate;
      w.pPar[part * relStride + h] = par;
    }
  } else {
    for (int h = tid; h < f.nBeam; h += W) {
      const uint32_t sid = w.bState[co + h];
      const uint32_t sp = w.bSPar[co + h];
      int mate = -1, par = -1;
#pragma unroll 8
      for (int h2 = 0; h2 < f.nBeam; ++h2) {
        const uint32_t s2 = w.bState[co + h2];
        mate = (s2 == sid && h2 != h) ? h2 : mate;
        par = (s2 == sp && par < 0) ? h2 : par;
      }
      w.dMate[h] = mate;
      w.dPar[h] = par;
    }
  }
  /* best candidate of the frame: best hypothesis (slot 0, the beam is sorted)
   * with its best token -- fl(fl(a + e) + sil) is monotone in a.  With a token
   * short-list the maximum is taken after barrier 1 (needs tokIdx). */
  if (f.nTok == P.N && f.nBeam > 0 && (wave == 0 || P.N > 64)) {
    unsigned long long bk = 0ull;
    const double a0 = w.bScore[co];
    for (int r = tid; r < P.N; r += W) {
      const double s = leanScore(P, a0, r, (double)f.e[r]);
      if (s == s) {
        const unsigned long long k = f64Key(s);
        bk = k > bk ? k : bk;
      }
    }
    bk = waveMax64(bk);
    if (lane == 0 && bk != 0ull) {
      atomMax64(&w.red[0], bk);
    }
  }
  leanBarrier(P); /* 1 */
  FLTX_PROF(6);
  /* ---- combine relations, repeat table, short-list membership ---------------- */
  for (int h = tid; h < f.nBeam; h += W) {
    int mate, par;
    if (laneBeam) {
      mate = -1;
      par = -1;
      for (int v = 0; v < nW; ++v) {
        const int m = w.pMate[v * 64 + h], p = w.pPar[v * 64 + h];
        mate = m >= 0 ? m : mate;
        par = (par < 0 && p >= 0) ? p : par;
      }
      w.dMate[h] = mate;
      w.dPar[h] = par;
    } else if (relParts >= 2) {
      mate = -1;
      par = -1;
      for (int v = 0; v < relParts; ++v) {
        const int m = w.pMate[v * relStride + h], p = w.pPar[v * relStride + h];
        mate = m >= 0 ? m : mate;
        par = (par < 0 && p >= 0) ? p : par;
      }
      w.dMate[h] = mate;
      w.dPar[h] = par;
    } else {
      par = w.dPar[h];
    }
    const uint32_t tp = w.bTokPb[co + h];
    const int t = (int)(tp & 0x7FFFFFFFu);
    if (!(tp & kPrevBlank) && !(ctc && t == P.blank) && par >= 0) {
      w.dRep[par * P.N + t] = (int16_t)h;
    }
    if (par >= 0) { /* some slot of the child state child(state of par, edge) */
      w.dKid[par * P.N + w.bSEdge[co + h]] = (int16_t)h;
    }
  }
  if (f.nTok != P.N) {
    for (int r = tid; r < f.nTok; r += W) {
      w.dIn[w.tokIdx[r]] = 1;
    }
    if (f.nBeam > 0) {
      unsigned long long bk = 0ull;
      const double a0 = w.bScore[co];
      for (int r = tid; r < f.nTok; r += W) {
        const int n = w.tokIdx[r];
        const double s = leanScore(P, a0, n, (double)f.e[n]);
        if (s == s) {
          const unsigned long long k = f64Key(s);
          bk = k > bk ? k : bk;
        }
      }
      bk = waveMax64(bk);
      if (lane == 0 && bk != 0ull) {
        atomMax64(&w.red[0], bk);
      }
    }
  }
  leanBarrier(P); /* 2 */
  FLTX_PROF(0);
  if (w.red[0] == 0ull) {
    return 0;
  }
  const double best = f64FromKey(w.red[0]);
  const double thr = best - P.beamThreshold;
  /* ---- phase B: evaluate my groups, bin them ---------------------------------- */
  /* GMAX < 255: the <= GMAX groups of a thread stay in registers through the
   * prune.  GMAX == 255 ("streaming", big beams): any number of groups per
   * thread; a group is evaluated here for the histogram and evaluated AGAIN in
   * phase D if its bin made the short-list -- twice the arithmetic, no
   * registers, no candidate records (K = 500 x 29 tokens would otherwise need
   * the 800 KB record workspace in HBM). */
  constexpr bool STREAM = GMAX == 255;
  constexpr int GR = STREAM ? 1 : GMAX;
  LeanGroup grp[GR];
  int bins[GR];
  double lo = thr;
  const bool wide = !(best - thr < 1e6); /* threshold too wide for useful bins */
  const int nGall = K * f.nTok + K; /* (state, token) groups, then one orphan-repeat slot per hypothesis */
  const int dRep_ = W / f.nTok, dR_ = W - dRep_ * f.nTok; /* streaming: (rep, r) advance per W groups */
  if constexpr (!STREAM) {
#pragma unroll
    for (int j = 0; j < GMAX; ++j) {
      leanEval(P, w, f, map.kind[j], map.rep[j], map.r[j], thr, grp[j]);
    }
  }
  if (wide) {
    if constexpr (STREAM) {
      if (tid == 0) { /* the general path handles unbounded thresholds */
        atomOr32((uint32_t*)&w.sc[SC_STATUS], ST_SELECT_FALLBACK);
      }
      return 0;
    } else {
      double mn = __builtin_huge_val();
#pragma unroll
      for (int j = 0; j < GMAX; ++j) {
        if (grp[j].valid && grp[j].s > -__builtin_huge_val()) {
          mn = grp[j].s < mn ? grp[j].s : mn;
        }
      }
      lo = blockMinF64(P, mn, &w.red[1]);
    }
  }
  /* Two-segment monotone binning of d = best - score over [0, range]: the
   * K-th best score is what the histogram has to isolate, so the part of the
   * range where it is expected gets 3/4 of the bins (short short-list) and the
   * rest shares the last quarter.  Each segment is floor(d * const): monotone in d, which is
   * all the exact rank needs.  NB is 1024 on this path (16 bins per lane). */
  const double range = best - lo;
  const int NF = (P.NB * 3) / 4;
  /* fine segment = a little more than the current beam's own spread (best to
   * K-th score): the next beam's K-th best lands there unless the frame is
   * unusual, and then the coarse segment still ranks it exactly */
  double cut = range * (1.0 / 16.0);
  if (f.nBeam >= K) {
    const double spread = (w.bScore[co] - w.bScore[co + f.nBeam - 1]) * 1.25 + 1e-3;
    cut = spread < range * 0.9 ? spread : range * 0.9;
  }
  double sF = (double)NF / cut, sC = (double)(P.NB - NF) / (range - cut);
  if (!(range > 0.0) || !(sF > 0.0) || !(sF < 1e300) || !(sC > 0.0) || !(sC < 1e300)) {
    cut = __builtin_huge_val(); /* degenerate range: everything lands in bin 0 */
    sF = 0.0;
    sC = 0.0;
  }
  auto binOf = [&](double sc) {
    const double d = best - sc;
    const double x = d < cut ? d * sF : (double)NF + (d - cut) * sC;
    int bin = (x < (double)P.NB) ? (int)x : P.NB - 1; /* also catches inf / NaN */
    return bin < 0 ? 0 : bin;
  };
  if constexpr (STREAM) {
    int rep = tid / f.nTok, r = tid - rep * f.nTok;
    for (int g = tid; g < nGall; g += W) {
      const int nG0 = K * f.nTok;
      LeanGroup x;
      leanEval(P, w, f, g < nG0 ? 1 : 2, g < nG0 ? rep : g - nG0, r, thr, x);
      if (x.valid) {
        atomAdd32(&w.hist[FLTX_HB(binOf(x.s))], 1u);
      }
      rep += dRep_;
      r += dR_;
      if (r >= f.nTok) {
        r -= f.nTok;
        rep += 1;
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < GMAX; ++j) {
      bins[j] = 0;
      if (grp[j].valid) {
        bins[j] = binOf(grp[j].s);
        atomAdd32(&w.hist[FLTX_HB(bins[j])], 1u);
      }
    }
  }
  FLTX_PROF(1);
  leanBarrier(P); /* 3 */
  /* ---- phase C: wave 0 turns counts into prefixes up to the K-th best's bin --- */
  if (wave == 0) {
    constexpr int PER = 16; /* NB / 64; the skewed index makes lane*16+q conflict-free */
    int c[PER];
    int mine = 0;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      c[q] = (int)w.hist[FLTX_HB(lane * PER + q)];
      mine += c[q];
    }
    const int inc = waveInclusiveScan(mine);
    int cum = inc - mine;
    if (cum < K) { /* my bins start before the crossing: publish their exclusive prefixes
                      (one past the crossing bin, so phase E knows every bin's extent) */
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
        w.hcum[FLTX_HB((lane + 1) * PER)] = (uint32_t)cum; /* first bin of the next lane */
      }
    }
    if (lane == 63 && inc < K) { /* fewer than K candidates: keep them all */
      w.sc[SC_CUM] = inc;
    }
  }
  leanBarrier(P); /* 4 */
  const int bstar = w.sc[SC_BSTAR];
  const int L = w.sc[SC_CUM];
  FLTX_PROF(2);
  if (L > P.SCAP) { /* degenerate score distribution: let the host use the general path */
    if (tid == 0) {
      atomOr32((uint32_t*)&w.sc[SC_STATUS], ST_SELECT_FALLBACK);
    }
    return 0;
  }
  /* ---- phase D: counting sort of the short-list by bin ---------------------------- */
  /* position = (candidates in better bins) + a ticket inside the bin; the bin's
   * count in hist[] doubles as the ticket counter (counted down) */
  auto scatter = [&](const LeanGroup& x, int bin, int g) {
    const int hb = FLTX_HB(bin);
    const uint32_t p = w.hcum[hb] + (atomAdd32(&w.hist[hb], 0xFFFFFFFFu) - 1u);
    const unsigned long long key = f64Key(x.s);
    w.sEnt[p] = make_uint4((uint32_t)key, (uint32_t)(key >> 32), x.ord, (uint32_t)bin);
    w.sIdx[p] = (uint32_t)g;
    w.sSrc[p] = x.src;
  };
  if constexpr (STREAM) {
    int rep = tid / f.nTok, r = tid - rep * f.nTok;
    for (int g = tid; g < nGall; g += W) {
      const int nG0 = K * f.nTok;
      LeanGroup x;
      leanEval(P, w, f, g < nG0 ? 1 : 2, g < nG0 ? rep : g - nG0, r, thr, x);
      if (x.valid) {
        const int bin = binOf(x.s);
        if (bin <= bstar) {
          scatter(x, bin, g);
        }
      }
      rep += dRep_;
      r += dR_;
      if (r >= f.nTok) {
        r -= f.nTok;
        rep += 1;
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < GMAX; ++j) {
      if (grp[j].valid && bins[j] <= bstar) {
        scatter(grp[j], bins[j], j * W + tid);
      }
    }
  }
  leanBarrier(P); /* 5 */
  FLTX_PROF(3);
  const int nS = L < K ? L : K;
  const int64_t hbase = f.histBase + (int64_t)frameOut * P.K;
  const int nG = P.K * f.nTok;
  /* ---- phase E: entry p ranks itself; rank < K builds beam slot `rank` ---------- */
  for (int p = tid; p < L; p += W) {
    const uint4 me = w.sEnt[p];
    const unsigned long long k = ((unsigned long long)me.y << 32) | me.x;
    const uint32_t o = me.z;
    /* exact rank = candidates in better bins + members of my bin that precede
     * me; a bin's members are contiguous after the counting sort */
    const int bin = (int)me.w;
    const int lo = (int)w.hcum[FLTX_HB(bin)];
    const int hi = bin >= bstar ? L : (int)w.hcum[FLTX_HB(bin + 1)];
    int rank = lo;
    for (int q = lo; q < hi; ++q) {
      const uint4 e = w.sEnt[q];
      const unsigned long long k2 = ((unsigned long long)e.y << 32) | e.x;
      rank += (k2 > k || (k2 == k && e.z < o)) ? 1 : 0;
    }
    FLTX_PROF(7);
    if (rank >= K) {
      continue;
    }
    const int g = (int)w.sIdx[p];
    const uint32_t src = w.sSrc[p];
    const int h = (int)(src & 0x7FFFFFFFu);
    uint32_t kp, ke, ktp;
    int n;
    if (g >= nG) {
      const int ho = g - nG;
      n = (int)(w.bTokPb[co + ho] & 0x7FFFFFFFu);
      kp = w.bSPar[co + ho];
      ke = (uint32_t)w.bSEdge[co + ho];
      ktp = (uint32_t)n;
    } else {
      const int rep = g / f.nTok, r = g - rep * f.nTok;
      n = (f.nTok == P.N) ? r : w.tokIdx[r];
      if (ctc && n == P.blank) {
        kp = w.bSPar[co + rep];
        ke = (uint32_t)w.bSEdge[co + rep];
        ktp = (uint32_t)n | kPrevBlank;
      } else {
        kp = w.bState[co + rep];
        ke = (uint32_t)n;
        ktp = (uint32_t)n;
      }
    }
    double am = w.bAm[co + h] + (double)f.e[n];
    if (f.useTrans) { /* ASG: transition enters am only (LexiconFreeDecoder.cpp:59-64) */
      const int prevTok = (int)(w.bTokPb[co + h] & 0x7FFFFFFFu);
      am = w.bAm[co + h] + ((double)f.e[n] + (double)P.transitions[(size_t)n * P.N + prevTok]);
    }
    /* LM-state id of the new slot, without a global round trip in the common
     * cases (see DecodeParams::childTab) */
    uint32_t sid;
    unsigned long long base;
    int repSlot;
    if (src & kNewState) {
      const int rep = g / f.nTok;
      const int kid = (int)w.dKid[rep * P.N + n];
      if (kid >= 0) { /* the state is in the beam: take its id from that slot */
        const int km = w.dMate[kid];
        sid = w.bState[co + kid];
        base = w.bMask[co + kid];
        repSlot = (km >= 0 && km < kid) ? km : kid;
      } else if ((w.bMask[co + rep] >> n) & 1ull) { /* existed, dropped out: rare re-entry */
        sid = loadCoherent32(&P.childTab[((size_t)f.b * P.idCap + kp) * P.N + n]);
        base = loadCoherent64(&P.maskTab[(size_t)f.b * P.idCap + sid]);
        repSlot = -1;
      } else { /* first time this state is materialised */
        sid = allocStateId(P, f.b, atomAdd32((uint32_t*)&w.sc[SC_NEXTID], 1u), kp, n, f.clock,
                           (uint32_t*)&w.sc[SC_STATUS]);
        base = 0ull;
        repSlot = -1;
        atomOr64(&w.addMask[rep], 1ull << n);
        P.childTab[((size_t)f.b * P.idCap + kp) * P.N + n] = sid;
        P.maskTab[(size_t)f.b * P.idCap + sid] = 0ull;
      }
    } else {
      const int hm = w.dMate[h];
      sid = w.bState[co + h];
      base = w.bMask[co + h];
      repSlot = (hm >= 0 && hm < h) ? hm : h;
    }
    w.bScore[no + rank] = f64FromKey(k);
    w.bAm[no + rank] = am;
    w.bState[no + rank] = sid;
    w.bSPar[no + rank] = kp;
    w.bSEdge[no + rank] = (int32_t)ke;
    w.bTokPb[no + rank] = ktp;
    w.eBase[rank] = base;
    w.eRep[rank] = repSlot;
    w.bPar[rank] = h;
  }
  leanBarrier(P); /* 6 */
  /* ---- phase E2: masks incl. this frame's additions; coalesced history write --- */
  for (int r = tid; r < nS; r += W) {
    const int rs = w.eRep[r];
    w.bMask[no + r] = w.eBase[r] | (rs >= 0 ? wsLoadAtomic64(P, &w.addMask[rs]) : 0ull); /* (ORed with atomics: at L2 on an HBM workspace) */
    const int n = (int)(w.bTokPb[no + r] & 0x7FFFFFFFu);
    P.histPT[hbase + r] = make_int2(w.bPar[r], n);
    if (P.histS) {
      double* hs = P.histS + 3 * (hbase + r);
      hs[0] = w.bScore[no + r];
      hs[1] = w.bAm[no + r];
      hs[2] = 0.0;
    }
  }
  for (int r = tid; r < f.nBeam; r += W) { /* persist the grown masks of the old states */
    const unsigned long long add = wsLoadAtomic64(P, &w.addMask[r]);
    if (add != 0ull) {
      P.maskTab[(size_t)f.b * P.idCap + w.bState[co + r]] = w.bMask[co + r] | add;
    }
  }
  leanBarrier(P); /* 7 */
  FLTX_PROF(4);
  return nS;
}

/* decodeEnd for the lean path (LexiconFreeDecoder.cpp:127-158 with ZeroLM):
 * every hypothesis becomes (state, sil, false); the <= 2 hypotheses of a state
 * merge; the survivors are already in descending order unless logAdd changed a
 * score, so the exact rank is recomputed. */
FLTX_DEV int runEndLean(const DecodeParams& P, const Ws& w, FrameCtx& f, int frameOut) {
  const int W = (int)blockDim.x;
  const int tid = (int)threadIdx.x;
  const int co = f.cur * P.K, no = (f.cur ^ 1) * P.K;
  if (tid == 0) {
    w.sc[SC_NSMALL] = 0;
  }
  for (int h = tid; h < f.nBeam; h += W) {
    const uint32_t sid = w.bState[co + h];
    int mate = -1;
    for (int h2 = 0; h2 < f.nBeam; ++h2) {
      mate = (w.bState[co + h2] == sid && h2 != h) ? h2 : mate;
    }
    w.dMate[h] = mate;
  }
  leanBarrier(P);
  const double best = f.nBeam > 0 ? w.bScore[co] : 0.0;
  const double thr = best - P.beamThreshold;
  const int rounds = (f.nBeam + W - 1) / W;
  for (int it = 0; it < rounds; ++it) {
    const int h = it * W + tid;
    LeanGroup g;
    g.valid = false;
    g.s = 0;
    g.src = 0;
    g.ord = 0;
    if (h < f.nBeam) {
      const int mate = w.dMate[h];
      if (!(mate >= 0 && mate < h)) {
        const double sA = w.bScore[co + h];
        const double sB = mate >= 0 ? w.bScore[co + mate] : 0.0;
        leanFold(P.logAdd != 0, sA >= thr, sA, (uint32_t)h, (uint32_t)h, mate >= 0 && sB >= thr, sB,
                 (uint32_t)(mate >= 0 ? mate : 0), (uint32_t)(mate >= 0 ? mate : 0), false, 0.0, 0u, 0u, g);
      }
    }
    const unsigned long long m = waveBallot(g.valid);
    if (m != 0ull) {
      const int lane = laneId();
      const int leader = __builtin_ctzll(m);
      uint32_t base = 0;
      if (lane == leader) {
        base = atomAdd32((uint32_t*)&w.sc[SC_NSMALL], (uint32_t)popc64(m));
      }
      base = waveShfl32(base, leader);
      if (g.valid) {
        const int p = (int)(base + (uint32_t)popc64(m & ((1ull << lane) - 1ull)));
        w.sKey[p] = f64Key(g.s);
        w.sOrd[p] = g.ord;
        w.sIdx[p] = g.src;
      }
    }
  }
  leanBarrier(P);
  const int L = w.sc[SC_NSMALL];
  const int64_t hbase = f.histBase + (int64_t)frameOut * P.K;
  for (int j = tid; j < L; j += W) {
    const unsigned long long k = w.sKey[j];
    const uint32_t o = w.sOrd[j];
    int rank = 0;
    for (int q = 0; q < L; ++q) {
      const unsigned long long k2 = w.sKey[q];
      const uint32_t o2 = w.sOrd[q];
      rank += (k2 > k || (k2 == k && o2 < o)) ? 1 : 0;
    }
    const int h = (int)w.sIdx[j];
    w.bScore[no + rank] = f64FromKey(k);
    w.bAm[no + rank] = w.bAm[co + h];
    w.bState[no + rank] = w.bState[co + h];
    w.bSPar[no + rank] = w.bSPar[co + h];
    w.bSEdge[no + rank] = w.bSEdge[co + h];
    w.bTokPb[no + rank] = (uint32_t)P.sil;
    w.bMask[no + rank] = w.bMask[co + h];
    P.histPT[hbase + rank] = make_int2(h, P.sil);
    if (P.histS) {
      double* hs = P.histS + 3 * (hbase + rank);
      hs[0] = f64FromKey(k);
      hs[1] = w.bAm[co + h];
      hs[2] = 0.0;
    }
  }
  leanBarrier(P);
  return L;
}
