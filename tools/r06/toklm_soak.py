#!/usr/bin/env python3
"""Random LexiconFreeDecoder + token-level n-gram LM configurations on the device against the oracle
(tests/test_gpu_batches.py _token_lm_grid: orders 2 - 4, CTC / ASG, token beams, thresholds, silScore, lmWeight of both
signs, max-merge bit-exact, logAdd @1e-5), on the default geometry and on the 512-thread one of which two workgroups
share a CU; then long utterances (T = 600 .. 1500, where re-entries and memo evictions happen).  A mismatch is excused
only when the ORACLE passed a tie on that input (oracle.cpp TieCounts) -- the grid skips inputs whose n-best holds equal
scores, everything else is red.  Prints one summary line.  Test infrastructure: the oracle is the checker."""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases, helpers, test_gpu_batches
from oracle import orclib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
sess = helpers.FltxSession(os.environ.get("EMU_LIB") or None)
orc = orclib.load("oracle")
t0 = time.time()
ran, served, bad = test_gpu_batches._token_lm_grid(sess, orc, n, 606, [1, 2, 7, 20, 45, 90, 200], emu=bool(os.environ.get("EMU_LIB")))
ran2, served2, bad2 = test_gpu_batches._token_lm_grid(sess, orc, n // 4, 607, [5, 33, 120], emu=bool(os.environ.get("EMU_LIB")),
                                                      sets={"slane_threads": 512})
# beams beyond 64: fltx_mlane.h's token-LM variant
ran3, served3, bad3 = test_gpu_batches._token_lm_grid(sess, orc, n // 3, 609, [1, 2, 7, 20, 45, 90, 200], emu=bool(os.environ.get("EMU_LIB")),
                                                      beams=test_gpu_batches.WIDE_BEAMS, tokens=(8, 12, 29, 29, 30), log_add=0.25)
rnd = random.Random(608)
long_ran = long_bad = ties_seen = 0
for i in range(max(4, n // 200)):
    K = rnd.choice([10, 30, 50, 64, 100, 200, 400])
    la = rnd.random() < 0.25
    c = cases.case("tl_long%d" % i, dist=rnd.choice(["ctc", "ctc", "uniform"]), T=rnd.choice([600, 1000, 1500]), N=29,
                   K=K, Kt=rnd.choice([29, 29, 10]), thr=rnd.choice([25.0, 8.0, 100.0]), u=9000 + i,
                   log_add=la, sil_score=rnd.choice([0.0, -0.4]), lm=("ngram", rnd.choice([2, 3, 4]), 50 + i % 4),
                   lm_weight=rnd.choice([0.5, 0.8, 1.5]))
    inp = helpers.case_inputs(c)
    want = helpers.run_checker(orc, c, inp)
    saw_tie = any(orc.last_ties.values())
    d = sess.decoder(c, inp)
    d.decode_batch(inp["e"], [c["T"]], c["N"])
    got = d.results(0)
    on_tl = d.get("tlane") == 1 and d.get("redone") == 0
    d.close()
    ok, why = helpers.hyps_equal(want, got, 1e-5 if la else 0.0)
    long_ran += 1
    if not ok and saw_tie:
        ties_seen += 1
        continue
    if not ok or not on_tl:
        long_bad += 1
        print("LONG MISMATCH", {k: c[k] for k in ("dist", "T", "K", "Kt", "thr", "lm", "lm_weight", "log_add", "u")}, why, on_tl, flush=True)
for b in (bad + bad2 + bad3)[:10]:
    print("MISMATCH", b)
print("token-LM soak: grid %d configurations (%d on the lane engine, %d mismatches), 512-thread geometry %d (%d, %d), "
      "beams 65 .. 512 %d (%d, %d), long utterances %d (%d mismatches, %d excused by ties the oracle saw) in %.0f s" % (
          ran, served, len(bad), ran2, served2, len(bad2), ran3, served3, len(bad3), long_ran, long_bad, ties_seen, time.time() - t0))
print("SOAK", "FAILED" if bad or bad2 or bad3 or long_bad or served != ran or served2 != ran2 or served3 != ran3 else "OK")
