"""tools/probe/stream_split.py -- where a 50-frame chunk of 256 lexicon-free streams spends its time"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from text_amd import _capi, synth
B, T, N, K, chunk = 256, 1000, 29, 50, 50
e = synth.batch("ctc", B, T, N, u0=0)
ctx = _capi.Context(device=0)
lm = _capi.ZeroLM(ctx)
d = _capi.BatchDecoder(ctx, _capi.LEXFREE, _capi.make_options(K, N, 25.0), lm, 0, N - 1)
Tc = np.full(B, chunk, dtype=np.int32)
chunks = [np.ascontiguousarray(e[:, k * chunk:(k + 1) * chunk, :]) for k in range(T // chunk)]
for mode in ("step+prune+sync",):
    acc = [0.0, 0.0, 0.0]
    for rep in range(2):
        d.stream_begin(B, N, int(os.environ.get("MAXF", T + 8)))
        ctx.synchronize()
        t00 = time.perf_counter()
        for c in chunks:
            t0 = time.perf_counter()
            d.stream_step(c, Tc)
            t1 = time.perf_counter()
            if mode != "step+sync":
                d.stream_prune(0)
            t2 = time.perf_counter()
            if mode != "step,prune no sync":
                ctx.synchronize()
            t3 = time.perf_counter()
            if rep:
                acc[0] += t1 - t0; acc[1] += t2 - t1; acc[2] += t3 - t2
        d.stream_end(); ctx.synchronize()
        tot = time.perf_counter() - t00
    n = len(chunks)
    print("%-22s per chunk: step call %.0f us, prune call %.0f us, sync %.0f us; total %.3f ms/chunk, engine %d" % (
        mode, acc[0] / n * 1e6, acc[1] / n * 1e6, acc[2] / n * 1e6, tot / n * 1e3, d.get("engine")))
