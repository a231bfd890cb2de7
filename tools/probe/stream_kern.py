"""tools/probe/stream_kern.py <workload> [key=value ...] -- bench.py's stream loop alone (for rocprofv3 --kernel-trace --stats)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
sys.argv = [sys.argv[0], "--workload", sys.argv[1]] + ["--set=" + kv for kv in sys.argv[2:]]
a = bench.parse()
cfg = bench.WORKLOADS[a.workload]
B = cfg["batch"]
job = bench.Job(a, 0, 0, B, cfg)
T, N, chunk = job.T, job.N, 50
d = job.decoder()
Tc = np.full(B, chunk, dtype=np.int32)
pieces = [np.ascontiguousarray(job.e_host[:, k * chunk:(k + 1) * chunk, :]) for k in range(T // chunk)]
d.set("stream_total_frames", T)
for rep in range(3):
    d.stream_begin(B, N, 4 * chunk + 8)
    job.ctx.synchronize()
    t0 = time.perf_counter()
    acc = [0.0, 0.0, 0.0]
    for p in pieces:
        t1 = time.perf_counter()
        d.stream_step(p, Tc)
        t2 = time.perf_counter()
        d.stream_prune(0)
        t3 = time.perf_counter()
        if rep < 2:
            job.ctx.synchronize()
        t4 = time.perf_counter()
        acc[0] += t2 - t1; acc[1] += t3 - t2; acc[2] += t4 - t3
    d.stream_end()
    job.ctx.synchronize()
    dt = time.perf_counter() - t0
    n = len(pieces)
    print("rep %d: %.3f ms/chunk (step call %.0f us, prune call %.0f us, sync %.0f us) %.1f M frames/s engine %d redone %d" % (
        rep, dt / n * 1e3, acc[0] / n * 1e6, acc[1] / n * 1e6, acc[2] / n * 1e6, B * T / dt / 1e6, d.get("engine"),
        d.get("stream_redone")))
