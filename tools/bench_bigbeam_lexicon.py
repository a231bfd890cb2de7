#!/usr/bin/env python3
"""Lexicon decoder at beams beyond the LDS paths (HBM workspace): kernel time per batch."""
import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, cases, helpers
from text_amd import synth
s = helpers.FltxSession(None)
for K, Kt, lm in [(200, 29, "zero"), (300, 29, "zero"), (300, 29, ("ngram", 4, 8)), (500, 29, "zero"), (500, 10, "zero"), (500, 29, ("ngram", 4, 8))]:
    c = cases.case("x", kind="lexicon", dist="lexspell", T=300, N=29, K=K, Kt=Kt, lexicon=cases.FULL_LEX, lm=lm,
                   lm_weight=2.0 if lm != "zero" else 0.0, word_score=2.0 if lm != "zero" else 0.0)
    inp = helpers.case_inputs(c); d = s.decoder(c, inp)
    B = 64
    e = synth.batch("lexspell", B, c["T"], 29, lexicon=inp["lex"], u0=0); Ts = np.full(B, c["T"], dtype=np.int32)
    d.decode_batch(e, Ts, 29); d.decode_batch(e, Ts, 29); s.ctx.synchronize()
    k, b = d.timing()
    print("lexicon K=%d Kt=%d lm=%s B=%d T=%d: kernel %.1f ms (%.3f ms/frame-batch), lds=%d cut=%d items=%d => %.2f M frames/s" % (K, Kt, lm, B, c["T"], k, k / c["T"], d.get("lds"), d.get("cut"), d.get("items"), B * c["T"] / k / 1e3))
    if os.environ.get("FLTX_PROFILE"):
        d.set("profile", 1); d.decode_batch(e, Ts, 29); s.ctx.synchronize()
        pr = d.profile().astype(np.float64) / (B * c["T"])
        print("  clocks/frame/utt [prep, generate, fold, select, build, row]:", " ".join("%.0f" % v for v in pr), "total %.0f" % pr.sum(), "threads", d.get("threads"), "hot", d.get("hot_level"))
    d.close()
