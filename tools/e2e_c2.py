#!/usr/bin/env python3
"""End-to-end timing of C2 at the C ABI: host emissions in (H2D), kernels, the whole n-best back in host
memory (fltx_result_fetch_batch, pinned staging), plus the cost of wrapping it in Python objects."""
import sys, time, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, cases, helpers
from text_amd import synth
c = cases.BY_NAME["C2_ctc_u0"]; B = 256
s = helpers.FltxSession(None); inp = helpers.case_inputs(c); d = s.decoder(c, inp)
e = synth.batch("ctc", B, c["T"], c["N"], u0=0); Ts = np.full(B, c["T"], dtype=np.int32)
for rep in range(3):
    t0 = time.perf_counter(); d.decode_batch(e, Ts, c["N"]); t1 = time.perf_counter()
    raw = d.fetch_batch_raw(); t2 = time.perf_counter()
    allh = d.results_batch(); t3 = time.perf_counter()
    print("rep %d: decode (H2D+kernels) %.2f ms, fetch_batch (D2H into pinned) %.2f ms, python views %.2f ms -> C-level e2e %.2f ms = %.1f M frames/s" % (rep, (t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, (t2-t0)*1e3, B*c["T"]/(t2-t0)/1e6))
d.close()
