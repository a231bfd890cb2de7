#!/usr/bin/env python3
"""Edge configurations of the lexicon-free decoders on the GPU against the oracle: tiny token
sets, beam 1, thresholds 0 / inf, token beam 1, silScore of both signs, ASG with transitions,
beam = lanes of a wave.  (Ties in the reference's n-best are skipped, as everywhere.)"""
import itertools
import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import cases, helpers
from oracle import orclib
orc = orclib.load("oracle")
s = helpers.FltxSession(None)
bad = ran = 0
grid = itertools.product([2, 3, 29, 64], [1, 2, 50, 64], [0.0, 1.5, 25.0, float("inf")], [None, 1, 3],
                         [0.0, -0.7, 0.4], ["ctc", "asg"], [1, 2, 17, 120])
for i, (N, K, thr, Kt, sil, crit, T) in enumerate(grid):
    if i % 7 not in (0, 3):
        continue
    if crit == "asg" and N == 2:
        continue
    c = cases.case("edge%d" % i, dist=["ctc", "uniform"][i % 2], u=500 + i, T=T, N=N, K=K,
                   Kt=min(N, Kt) if Kt else None, thr=thr, sil_score=sil, crit=crit,
                   trans_seed=(90 + i) if crit == "asg" else None)
    inp = helpers.case_inputs(c)
    want = helpers.run_checker(orc, c, inp)
    if len({h.score for h in want}) != len(want):
        continue
    d = s.decoder(c, inp)
    d.decode_batch(inp["e"], [c["T"]], c["N"])
    eng = d.get("engine")
    got = d.results(0)
    d.close()
    ok, why = helpers.hyps_equal(want, got)
    ran += 1
    if not ok:
        bad += 1
        print("MISMATCH", {k: c[k] for k in ("N", "K", "Kt", "thr", "sil_score", "crit", "T", "dist")}, "engine", eng, why)
print("ran", ran, "mismatches", bad)
