#!/usr/bin/env python3
"""logAdd on the lexicon lane engines against the oracle, longer than the suite's grids: the generators of
tests/test_gpu_batches.py (_logadd_lexicon_grid: fltx_xlane.h LA; _logadd_lm_lexicon_grid: fltx_ylane.h LMK bit 3) with
other seeds, then ragged batches with `defer_check`.  Prints one JSON line.  Test infrastructure: the oracle checks."""
import json, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import cases, helpers, test_gpu_batches as tgb
from oracle import orclib
from text_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
tol = float(os.environ.get("TOL", "1e-5"))
sess = helpers.FltxSession(os.environ.get("EMU_LIB") or None)
orc = orclib.load("oracle")
t0 = time.time()
out = {}
on5, bad5 = tgb._logadd_lexicon_grid(sess, orc, n, 101, [1, 5, 20, 40, 70, 150], tol)
out["xlane"] = {"configurations": n, "on_engine_5_not_redone": on5, "mismatches": len(bad5)}
on6, red6, bad6 = tgb._logadd_lm_lexicon_grid(sess, orc, n, 102, [1, 5, 20, 40, 70, 150], tol)
out["ylane"] = {"configurations": n, "on_engine_6": on6, "redone": red6, "mismatches": len(bad6)}
# ragged batches, more workgroups than CUs, the look at the statuses deferred
rnd = random.Random(7)
rb = {"batches": 0, "utterances_checked": 0, "mismatches": 0, "redone": 0}
for name in ("ng_word_logadd_t40", "lx_spell_t60_k12_logadd"):
    c = dict(cases.BY_NAME[name]); c["K"] = 40
    inp = helpers.case_inputs(c)
    B = 700
    Ts = [rnd.choice([0, 1, 9, 33, 64, 90]) for _ in range(B)]
    embs = [synth.emissions("lexspell", 9000 + b, T, c["N"], lexicon=inp["lex"]) for b, T in enumerate(Ts)]
    d = sess.decoder(c, inp)
    d.set("defer_check", 1)
    d.decode_batch(np.concatenate([e.reshape(-1) for e in embs]), Ts, c["N"])
    for b in rnd.sample(range(B), 80):
        c1 = dict(c); c1["T"] = Ts[b]
        ok, why = helpers.hyps_equal(helpers.run_checker(orc, c1, dict(inp, e=embs[b])), d.results(b), tol)
        rb["utterances_checked"] += 1
        rb["mismatches"] += 0 if ok else 1
    rb["redone"] += d.get("redone")
    rb["batches"] += 1
    d.close()
out["ragged_deferred_batches"] = rb
out["seconds"] = round(time.time() - t0, 1)
out["bad"] = (bad5 + bad6)[:5]
print(json.dumps(out))
