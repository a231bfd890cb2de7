"""tools/probe/pipe2.py -- C2 with two decoders on two streams alternating batches (the back-trace of one
batch under the decode kernel of the next) against one decoder."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from text_amd import _capi, synth
B, T, N, K = 256, 1000, 29, 50
e = torch.from_numpy(synth.batch("ctc", B, T, N, u0=0)).cuda()
torch.cuda.synchronize()
Ts = np.full(B, T, dtype=np.int32)
opt = _capi.make_options(K, N, 25.0)
ctxs = [_capi.Context(device=0) for _ in range(2)]
lms = [_capi.ZeroLM(c) for c in ctxs]
decs = [_capi.BatchDecoder(c, _capi.LEXFREE, opt, lm, 0, N - 1) for c, lm in zip(ctxs, lms)]
for bt in (0, 1):
    if bt:
        for d in decs:
            d.set("bt_lds_kb", 100)
    for n in (1, 2):
        for _ in range(3):
            for d in decs[:n]:
                d.decode_batch(None, Ts, N, device_ptr=e.data_ptr())
        for c in ctxs:
            c.synchronize()
        t0 = time.perf_counter()
        steps = 20
        for i in range(steps):
            decs[i % n].decode_batch(None, Ts, N, device_ptr=e.data_ptr())
        for c in ctxs:
            c.synchronize()
        dt = (time.perf_counter() - t0) / steps
        print("bt_lds %d, %d decoder(s): %.3f ms/batch = %.1f M frames/s; kernel %.3f + %.3f ms" % (
            bt, n, dt * 1e3, B * T / dt / 1e6, *decs[0].timing()))
