#!/bin/bash
# tools/soak_r03.sh -- the differential runs behind DESIGN.md section 2's round-3 sentence, on an MI355X (through
# gpurun): every new path of the round against the oracle / the generic engine on more inputs than the -m gpu suite.
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
echo "== fuzz_big, 6000 random configurations, default engines"; FLTX_FUZZ_N=6000 python tools/fuzz_big.py 2>&1 | tail -3
echo "== fuzz_big, 2000 configurations, lane engines on the shared-CU geometry (yshare=1)"; FLTX_FUZZ_N=2000 FLTX_FUZZ_SET="yshare=1" python tools/fuzz_big.py 2>&1 | tail -3
echo "== fuzz_big, 2000 configurations, every workspace in HBM"; FLTX_FUZZ_N=2000 FLTX_FUZZ_SET="lds_budget=2048" python tools/fuzz_big.py 2>&1 | tail -3
echo "== C3 / C4 batches of 256, lane engines (memo in LDS, memo in HBM) vs the generic engine, every utterance"
python tools/cmp_workload.py C3 256 2>&1 | tail -2
python tools/cmp_workload.py C3 256 yshare=1 2>&1 | tail -2
python tools/cmp_workload.py C4 256 2>&1 | tail -2
python tools/cmp_workload.py C4 256 yshare=1 2>&1 | tail -2
echo "== C2 batch, engine 4 (576 threads) vs engine 3, and vs the 512-thread geometry"
python tools/cmp_engines.py 2>&1 | tail -2
python tools/cmp_engines.py ctc 256 1000 50 slane_threads=512 2>&1 | tail -2
python tools/cmp_engines.py ctc 512 1000 50 slane=0 2>&1 | tail -2
python - <<'PY'
import sys, os, itertools
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import cases, helpers, stream_scenarios as ss
from oracle import orclib
orc = orclib.load("oracle")
s = helpers.FltxSession(None)
# every configuration of the stream grid (960), lexicon-free streams on the lane = LM state engine
bad = ran = 0
grid = itertools.product([1, 3, 10, 50, 64], [2.0, 25.0, float("inf")], [None, 5], [12, 29], [7, 60],
                         ["ctc", "uniform"], [0.0, -0.6], ["ctc", "asg"])
for i, (K, thr, Kt, N, T, dist, sil, crit) in enumerate(grid):
    c = cases.case("ss%d" % i, dist=dist, u=1700 + i, T=T, N=N, K=K, Kt=Kt, thr=thr, sil_score=sil, crit=crit,
                   trans_seed=(30 + i) if crit == "asg" else None)
    inp = helpers.case_inputs(c)
    chunks, lbs = [[3, 9, 1, 12, 20, 30], [1] * 60, [25, 25, 25]][i % 3], [[0, 2, 0, 5], [0], [3, 1]][i % 3]
    want = ss.trace_checker(orc, c, inp, chunks, lbs)
    if ss.has_ties(want):
        continue
    got, _ = ss.trace_device(s, c, inp, chunks, lbs)
    ran += 1
    if ss.first_difference(want, got):
        bad += 1
        print("STREAM MISMATCH", {k: c[k] for k in ("K", "thr", "Kt", "N", "T", "dist", "sil_score", "crit")}, ss.first_difference(want, got))
print("== lexicon-free streams on the lane engine: %d configurations, %d mismatches" % (ran, bad))
# long streams: C2-shaped utterances in 50-frame chunks with prune(0) / prune(7), 24 utterances
bad = 0
for u in range(24):
    c = cases.case("long%d" % u, T=1000, K=50, N=29, u=3000 + u, thr=25.0)
    inp = helpers.case_inputs(c)
    chunks, lbs = [50] * 20, [0, 7]
    want = ss.trace_checker(orc, c, inp, chunks, lbs)
    got, _ = ss.trace_device(s, c, inp, chunks, lbs)
    bad += 1 if ss.first_difference(want, got) else 0
print("== 24 streams of 1000 frames in 50-frame chunks: %d mismatches" % bad)
# lexicon streams: optimistic geometry, with and without a forced cut (chunks decoded again), n-gram and ZeroLM
bad = ran = redone = 0
for i in range(120):
    base = cases.BY_NAME[["lx_spell_t60_k12_full", "lx_scores_t50", "ng_word_t60_k16_4g", "ng_word_t40_k10"][i % 4]]
    c = dict(base, name="lxs%d" % i, u=4000 + i, K=[4, 12, 16, 40, 100][i % 5])
    inp = helpers.case_inputs(c)
    chunks, lbs = [[10] * 8, [7, 13, 1, 20, 30], [25, 25, 25]][i % 3], [[0, 3, 1], [0], [2, 0]][i % 3]
    want = ss.trace_checker(orc, c, inp, chunks, lbs)
    tun = [("cut_m", c["K"] + 1)] if i % 2 else []
    got, _ = ss.trace_device(s, c, inp, chunks, lbs, tunables=tun)
    redone += s.last_stream_redone
    ran += 1
    if ss.first_difference(want, got):
        bad += 1
        print("LEXICON STREAM MISMATCH", c["name"], base["name"], c["K"], tun, ss.first_difference(want, got))
print("== lexicon streams on the optimistic geometry: %d configurations (%d chunks decoded again), %d mismatches" % (ran, redone, bad))
# logAdd on the lane = LM state engine: the whole grid
bad = ran = 0
grid = itertools.product([1, 3, 10, 50, 64], [2.0, 25.0, float("inf")], [None, 5], [12, 29], [1, 30, 120],
                         ["ctc", "uniform"], [0.0, -0.6], ["ctc", "asg"])
for i, (K, thr, Kt, N, T, dist, sil, crit) in enumerate(grid):
    c = cases.case("la%d" % i, dist=dist, u=2500 + i, T=T, N=N, K=K, Kt=Kt, thr=thr, sil_score=sil, log_add=True,
                   crit=crit, trans_seed=(30 + i) if crit == "asg" else None)
    inp = helpers.case_inputs(c)
    want = helpers.run_checker(orc, c, inp)
    if any(abs(a.score - b.score) < 1e-4 for a, b in zip(want, want[1:])):
        continue
    got = s.run(c, inp)
    ran += 1
    ok, why = helpers.hyps_equal(want, got, 1e-5)
    if not ok:
        bad += 1
        print("LOGADD MISMATCH", {k: c[k] for k in ("K", "thr", "Kt", "N", "T", "dist", "sil_score", "crit")}, why)
print("== logAdd on the lane = LM state engine: %d configurations, %d mismatches" % (ran, bad))
PY
