#!/usr/bin/env python3
"""Larger randomized differential run (GPU vs oracle) than the 36 cases of the test-suite.
FLTX_FUZZ_SET="key=value,key=value" applies decoder tunables to every case (e.g.
lds_budget=2048 pushes every lexicon case onto the HBM-workspace paths)."""
import os
import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import cases, helpers
from oracle import orclib
orc = orclib.load("oracle")
s = helpers.FltxSession(os.environ.get("EMU_LIB") or None)  # (EMU_LIB=tests/emu/libfltx_emu.so: the emulated kernels, no GPU)
bad = n_tie = 0
cs = cases.fuzz_cases(int(os.environ.get("FLTX_FUZZ_N", "400")))
for i, c in enumerate(cs):
    inp = helpers.case_inputs(c)
    want = helpers.run_checker(orc, c, inp)
    ties = dict(orc.last_ties)
    if len({h.score for h in want}) != len(want):
        continue  # equal scores: the reference's own result is order dependent
    try:
        d = s.decoder(c, inp)
        for kv in filter(None, os.environ.get("FLTX_FUZZ_SET", "").split(",")):
            d.set(kv.split("=")[0], int(kv.split("=")[1]))
        d.decode_batch(inp["e"], [c["T"]], c["N"])
        got = d.results(0)
        s.last_engine = d.get("engine")
        d.close()
        ok, why = helpers.hyps_equal(want, got, 1e-5 if c["log_add"] else 0.0)
    except Exception as e:
        ok, why = False, "EXC %r" % (e,)
    if not ok and any(ties.values()):
        # the ORACLE passed a tie on this input (oracle.cpp TieCounts: equal scores inside a merge group, across the
        # beam's cut, ...): the reference's own answer depends on addresses there.  A mismatch on a tie-free input
        # is always counted below, whatever the compiled reference says.
        n_tie += 1
        print("TIE (seen by the oracle: %s)" % {k: v for k, v in ties.items() if v}, c["name"])
        continue
    if not ok:
        bad += 1
        print("MISMATCH", c["name"], {k: c[k] for k in ("kind", "dist", "N", "K", "Kt", "thr", "lm", "log_add", "T", "lm_weight", "word_score", "unk_score", "sil_score")}, why, "engine", s.last_engine if hasattr(s, "last_engine") else None)
print("done", len(cs), "cases,", bad, "mismatches, ties_seen_by_oracle", n_tie)
