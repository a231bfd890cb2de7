#!/usr/bin/env python3
"""A few LARGE random configurations (long utterances, beams up to 120, the 90k-word trie, 4-gram LM)
on the GPU against the oracle: exercises the cut-off generation, the item list, the lane / lean engines and
the HBM workspace at realistic sizes.  Slow on the oracle side (seconds per case)."""
import random
import sys

sys.path.insert(0, "/root/repo")
sys.path.insert(0, "/root/repo/tests")
import cases  # noqa: E402
import helpers  # noqa: E402
from oracle import orclib  # noqa: E402

rnd = random.Random(7)
cs = []
for i in range(14):
    kind = "lexicon" if i % 2 else "lexfree"
    K = rnd.choice([20, 50, 64, 100, 120])
    if kind == "lexicon":
        lm = "zero" if i % 4 == 1 else ("ngram", 4, 8)
        cs.append(cases.case("L%02d" % i, kind="lexicon", dist="lexspell", u=700 + i, T=rnd.choice([150, 300]), N=29, K=K,
                             Kt=rnd.choice([29, 10]), thr=rnd.choice([10.0, 25.0]), lexicon=cases.FULL_LEX, lm=lm,
                             lm_weight=2.0 if lm != "zero" else 0.0, word_score=2.0 if lm != "zero" else 0.0,
                             sil_score=-1.0 if lm != "zero" else 0.0))
    else:
        cs.append(cases.case("L%02d" % i, dist=rnd.choice(["ctc", "uniform"]), u=700 + i, T=rnd.choice([400, 800]), N=29,
                             K=K, Kt=rnd.choice([29, 29, 12]), thr=rnd.choice([10.0, 25.0]),
                             sil_score=rnd.choice([0.0, -0.4])))
for i, (K, T, Kt, dist, la) in enumerate([(300, 120, 29, "ctc", False), (450, 80, 29, "uniform", False), (300, 60, 12, "ctc", False),
                                          (640, 50, 29, "ctc", False), (900, 30, 29, "ctc", False), (260, 60, 29, "ctc", True),
                                          (700, 30, 29, "ctc", True)]):
    cs.append(cases.case("B%02d" % i, dist=dist, u=900 + i, T=T, N=29, K=K, Kt=Kt, log_add=la))
for i, (K, Kt, lm) in enumerate([(200, 29, "zero"), (300, 29, "zero"), (280, 29, ("ngram", 4, 8)), (240, 10, "zero"),
                                 (500, 29, "zero"), (500, 29, ("ngram", 4, 8)), (800, 29, "zero"), (1200, 10, "zero")]):
    # lexicon beams in the hundreds: recompute form of the cut-off generation, with / without the item list
    cs.append(cases.case("R%02d" % i, kind="lexicon", dist="lexspell", u=950 + i, T=120, N=29, K=K, Kt=Kt, thr=25.0,
                         lexicon=cases.FULL_LEX, lm=lm, lm_weight=2.0 if lm != "zero" else 0.0,
                         word_score=2.0 if lm != "zero" else 0.0, sil_score=-1.0 if lm != "zero" else 0.0))
orc = orclib.load("oracle")
s = helpers.FltxSession(None)
bad = 0
for c in cs:
    inp = helpers.case_inputs(c)
    want = helpers.run_checker(orc, c, inp)
    tie = len({h.score for h in want}) != len(want)
    d = s.decoder(c, inp)
    d.decode_batch(inp["e"], [c["T"]], c["N"])
    got = d.results(0)
    info = "engine %d lean %d lds %d hot %d cut %d re %d items %d" % (d.get("engine"), d.get("lean"), d.get("lds"), d.get("hot_level"),
                                                                     d.get("cut"), d.get("recompute"), d.get("items"))
    d.close()
    ok, why = helpers.hyps_equal(want, got, 1e-5 if c["log_add"] else 0.0)
    print(c["name"], c["kind"], "K=%d Kt=%d T=%d lm=%s" % (c["K"], c["Kt"], c["T"], c["lm"]), info,
          "OK" if ok else ("TIE-" if tie else "") + "MISMATCH " + why, flush=True)
    bad += 0 if ok or tie else 1
print("done,", bad, "mismatches")
