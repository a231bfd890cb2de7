#!/usr/bin/env python3
"""The secondary C2 runs of SURVEY.md section 8(d): logAdd = true, beamThreshold = 1e9, and a token beam of 10,
each checked against the reference CPU on four utterances."""
import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, cases, helpers
from text_amd import synth
from oracle import orclib
orc = orclib.load("ref" if orclib.have_ref() else "oracle")
s = helpers.FltxSession(None)
for nm, kw in [("logAdd", dict(log_add=True)), ("thr1e9", dict(thr=1e9)), ("kt10", dict(Kt=10))]:
    c = cases.case("sec_" + nm, T=1000, K=50, N=29, **kw)
    d = s.decoder(c, dict(tr=None)); B = 256
    e = synth.batch("ctc", B, 1000, 29, u0=0); Ts = np.full(B, 1000, dtype=np.int32)
    d.decode_batch(e, Ts, 29); d.decode_batch(e, Ts, 29); s.ctx.synchronize()
    k, b = d.timing()
    mism = 0
    for u in range(4):
        want = helpers.run_checker(orc, c, dict(e=e[u], tr=None, lex=None))
        ok, why = helpers.hyps_equal(want, d.results(u), 1e-5 if c["log_add"] else 0.0)
        mism += 0 if ok else 1
    print("C2 %s: engine %d lean %d, kernel %.2f ms, backtrace %.2f ms, %.1f M frames/s (kernel), mismatches vs reference on 4 utterances: %d" % (nm, d.get("engine"), d.get("lean"), k, b, B * 1000 / k / 1e3, mism))
    d.close()
