"""tools/probe/coreside.py <libfltx.so> -- does a second workgroup on a CU overlap the first one's waits?
The C4 decoder at beam 50 (one lane group, 512 threads) on short utterances, 256 vs 512 utterances."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from text_amd import _capi, synth
import torch

lib = _capi.Lib(sys.argv[1]) if len(sys.argv) > 1 else _capi.default_lib()
_capi._default = lib
class A: tokens = 29; threads = 0; set = []
T, K = int(os.environ.get("T", 300)), int(os.environ.get("K", 50))
cfg = dict(lex=True, batch=512, T=T, K=K, Kt=29, lm=True)
job = bench.Job(A, 0, 0, 512, cfg)
e = torch.from_numpy(job.e_host).cuda()
torch.cuda.synchronize()
for B in (256, 512):
    dec = job.decoder()
    Ts = np.full(B, T, dtype=np.int32)
    for _ in range(2):
        dec.decode_batch(None, Ts, 29, device_ptr=e.data_ptr())
    job.ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        dec.decode_batch(None, Ts, 29, device_ptr=e.data_ptr())
    job.ctx.synchronize()
    dt = (time.perf_counter() - t0) / 5
    st = dec.stats()
    print("B=%d: %.3f ms/batch, kernel %.3f ms, engine %d redone %d threads %d lds %d" % (
        B, dt * 1e3, dec.timing()[0], dec.get("engine"), dec.get("redone"), st["threads_per_utt"], st["lds_bytes"]))
    dec.close()
