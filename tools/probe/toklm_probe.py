import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import cases, helpers
from text_amd import synth
sess = helpers.FltxSession(None)
B = 256
for lm, sets in (("zero", {}), (("ngram", 3, 11), {}), (("ngram", 3, 11), {"slane_threads": 512}), (("ngram", 4, 12), {}),
                 (("ngram", 3, 11), {"tlane": 0}), (("ngram", 3, 11), {"tlane": 0, "tok_dense": 0})):
    c = cases.case("probe", dist="ctc", T=1000, N=29, K=50, u=0, lm=lm, lm_weight=0.8 if lm != "zero" else 0.0)
    inp = helpers.case_inputs(c)
    d = sess.decoder(c, inp)
    for k, v in sets.items():
        d.set(k, v)
    e = synth.batch("ctc", B, c["T"], c["N"])
    for _ in range(3):
        t0 = time.perf_counter(); d.decode_batch(e, [c["T"]] * B, c["N"]); sess.ctx.synchronize(); dt = time.perf_counter() - t0
    print(lm, sets, "engine", d.get("engine"), "tlane", d.get("tlane"), "contexts", d.get("toklm_contexts"), "redone", d.get("redone"),
          "why", d.get("why_not_lane"), "wall ms", round(dt * 1e3, 2), "kernel/backtrace ms", d.timing())
    d.close()
# beams beyond the lane engine's 64 (generic engine; with the dense table / with the probe chain), and a stream
for K, sets in ((100, {}), (100, {"tok_dense": 0}), (200, {}), (200, {"tok_dense": 0})):
    c = cases.case("probe", dist="ctc", T=1000, N=29, K=K, u=0, lm=("ngram", 3, 11), lm_weight=0.8)
    inp = helpers.case_inputs(c)
    d = sess.decoder(c, inp)
    for k, v in sets.items():
        d.set(k, v)
    e = synth.batch("ctc", 64, c["T"], c["N"])
    for _ in range(2):
        d.decode_batch(e, [c["T"]] * 64, c["N"]); sess.ctx.synchronize()
    print("beam", K, sets, "engine", d.get("engine"), "why", d.get("why_not_lane"), "64 utterances: kernel/backtrace ms", d.timing())
    d.close()
