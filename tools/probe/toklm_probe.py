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
# a token LM of realistic size: a 6-gram over the 29 tokens with about two million contexts (a 475 MB table; round 6's
# first builder stopped at 2^20 contexts) -- time to build and upload the table, then the C2 shape on the lane-state engine
if "--big" in sys.argv:
    from text_amd import _capi, ngram_synth
    N = 29
    vocab = ngram_synth.words(N, "t")
    path = os.path.join(helpers.NGRAM_DIR, "lm_tok_big_o6.arpa")
    os.makedirs(helpers.NGRAM_DIR, exist_ok=True)
    if not os.path.exists(path):
        t0 = time.perf_counter()
        n = ngram_synth.write_arpa(path, vocab, 6, (0, 900, 25000, 400000, 900000, 600000), 11)
        print("synthetic 6-gram: %d n-grams written in %.0f s" % (n, time.perf_counter() - t0))
    t0 = time.perf_counter(); lm = _capi.ArpaLM(path, vocab, lib=sess.lib); t_load = time.perf_counter() - t0
    c = cases.case("probe_big", dist="ctc", T=1000, N=N, K=50, u=0, lm=("ngram", 6, 11), lm_weight=0.8, is_lm_token=True)
    inp = helpers.case_inputs(c)
    e = synth.batch("ctc", B, c["T"], N)
    for sets in ({}, {"slane_threads": 512}, {"tlane": 0}):
        d = sess.decoder(c, inp, lm=lm)
        for k, v in sets.items():
            d.set(k, v)
        walls = []
        for _ in range(3):
            t0 = time.perf_counter(); d.decode_batch(e, [c["T"]] * B, N); sess.ctx.synchronize(); walls.append(time.perf_counter() - t0)
        print("6-gram", sets, "ARPA load s", round(t_load, 2), "first batch (table build + upload) s", round(walls[0], 2), "engine", d.get("engine"),
              "tlane", d.get("tlane"), "contexts", d.get("toklm_contexts"), "redone", d.get("redone"), "wall ms", round(walls[-1] * 1e3, 2),
              "kernel/backtrace ms", d.timing())
        d.close()
