"""tools/probe/corun.py -- do the decode kernels of two decoder objects (two HIP streams) run side by side on the CUs?
C2 on the 512-thread geometry (two workgroups fit a CU: 8 waves x 104 VGPRs each) against the 576-thread default,
one / two / three decoder objects taking batches in turn; with and without the back-trace (results fetched or not)."""
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
ND = 4
ctxs = [_capi.Context(device=0) for _ in range(ND)]
lms = [_capi.ZeroLM(c) for c in ctxs]
for th, bt in ((512, 0), (512, 100), (512, 64), (512, 32), (448, 64)):
    decs = [_capi.BatchDecoder(c, _capi.LEXFREE, opt, lm, 0, N - 1) for c, lm in zip(ctxs, lms)]
    for d in decs:
        d.set("slane_threads", th)
        d.set("defer_check", int(os.environ.get("DEFER", "1")))
        if bt:
            d.set("bt_lds_kb", bt)
    for n in (2, 3, 4):
        for _ in range(3):
            for d in decs[:n]:
                d.decode_batch(None, Ts, N, device_ptr=e.data_ptr())
        for c in ctxs:
            c.synchronize()
        t0 = time.perf_counter()
        steps = 30
        for i in range(steps):
            decs[i % n].decode_batch(None, Ts, N, device_ptr=e.data_ptr())
        for c in ctxs:
            c.synchronize()
        dt = (time.perf_counter() - t0) / steps
        print("bt_lds %d threads %d, %d decoder(s): %.3f ms/batch = %.1f M frames/s; kernel %.3f + %.3f ms (events of the last batch)" % (
            bt, th, n, dt * 1e3, B * T / dt / 1e6, *decs[0].timing()), flush=True)
    for d in decs:
        d.close()
