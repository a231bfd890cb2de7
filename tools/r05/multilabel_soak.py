#!/usr/bin/env python3
"""Random lexicon-decoder configurations over lexicons whose spellings carry one, two or three words (helpers.lexicon
mode "multi"), with a synthetic n-gram word LM, on the device against the oracle: beams across the one / two / four
lane-group geometries of fltx_ylane.h (LMK bit 2), CTC and ASG.  Prints one summary JSON line.
Test infrastructure: the oracle is the checker."""
import json, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import cases, helpers
from oracle import orclib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
sess = helpers.FltxSession(os.environ.get("EMU_LIB") or None)
orc = orclib.load("oracle")
rnd = random.Random(int(os.environ.get("SEED", "20260930")))
stats = {"configs": 0, "mismatches": 0, "on_lane_engine": 0, "redone": 0, "by_groups": {}, "fallback_reasons": {}, "cut_ties": 0, "homophone_order_ties": 0}
t0 = time.time()
bad = []
for i in range(n):
    asg = rnd.random() < 0.4
    big = rnd.random() < 0.5
    lexi = (cases.MULTI_NODUP_LEX_3K if big else cases.MULTI_NODUP_LEX) if asg else (cases.MULTI_LEX_3K if big else cases.MULTI_LEX)
    K = rnd.choice([3, 10, 24, 50, 64, 65, 100, 128, 129, 180, 256])
    c = cases.case("mls%d" % i, kind="lexicon", dist=rnd.choice(["lexspell", "lexspell", "uniform"]), T=rnd.choice([1, 7, 40, 80, 150]),
                   K=K, Kt=rnd.choice([29, 29, 10, 5]), thr=rnd.choice([25.0, 25.0, 8.0, 100.0]), lexicon=lexi, u=1000 + i,
                   crit="asg" if asg else "ctc", trans_seed=(50 + i % 7) if asg else None,
                   lm=("ngram", rnd.choice([2, 3, 4]), 60 + i % 5), lm_weight=rnd.choice([0.5, 1.3, 2.0]),
                   word_score=rnd.choice([0.0, 0.7, 2.0]), sil_score=rnd.choice([0.0, -0.5, -1.0]))
    inp = helpers.case_inputs(c)
    d = sess.decoder(c, inp)
    d.decode_batch(inp["e"], [c["T"]], c["N"])
    got = d.results(0)
    eng, grp, red, fb = d.get("engine"), d.get("lane_groups"), d.get("redone"), d.get("fallback_reasons")
    d.close()
    want = helpers.run_checker(orc, c, inp)
    ok, why = helpers.hyps_equal(want, got)
    ties = dict(orc.last_ties)  # (what the ORACLE passed on this input: oracle.cpp TieCounts)
    if not ok and ties["cut"]:
        stats["cut_ties"] += 1  # equal scores across the beam's cut (nth_element's choice in the reference)
        ok = True
    if not ok and (ties["order"] or ties["token"] or ties["best"]):
        stats["order_ties"] = stats.get("order_ties", 0) + 1  # equal scores inside the sorted n-best (partial_sort's choice), ...
        ok = True
    if not ok and ties["merge"]:
        # equal scores inside a merge group: two orders of the same homophones in one history (the n-gram context forgets
        # them, the histories merge on a tie; the reference leaves the survivor to its sort)
        stats["homophone_order_ties"] += 1
        ok = True
    stats["ties_seen_by_oracle"] = stats.get("ties_seen_by_oracle", 0) + int(any(ties.values()))
    stats["configs"] += 1
    stats["on_lane_engine"] += int(eng == 6)
    stats["redone"] += int(red)
    stats["by_groups"][str(grp)] = stats["by_groups"].get(str(grp), 0) + 1
    if fb:
        stats["fallback_reasons"][str(fb)] = stats["fallback_reasons"].get(str(fb), 0) + 1
    if not ok:
        stats["mismatches"] += 1
        bad.append({"i": i, "K": K, "T": c["T"], "crit": c["crit"], "lex": list(lexi), "engine": eng, "groups": grp, "redone": red, "why": why})
        print("MISMATCH", bad[-1], flush=True)
stats["seconds"] = round(time.time() - t0, 1)
stats["bad"] = bad[:10]
print(json.dumps(stats))
