#!/usr/bin/env python3
"""Differential run of two engine settings on one of bench.py's workloads (GPU).

  python tools/cmp_workload.py C4 [B] [key=value ...]   (key=value: tunables of the 2nd run; default ylane=0 xlane=0)

Decodes the workload's synthetic batch with the default engine choice and with the given
tunables and reports the utterances whose n-best (scores, tokens, words) differ."""
import sys, os, time, types
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench


def main():
    a = sys.argv[1:]
    wl = a[0] if a else "C4"
    B = int(a[1]) if len(a) > 1 else 256
    sets = [x for x in a[2:]] or ["ylane=0", "xlane=0"]
    cfg = dict(bench.WORKLOADS[wl])
    res = []
    for sset in ([], sets):
        args = types.SimpleNamespace(tokens=29, threads=0, set=sset, asg=False, log_add=False, frames=0, beam=0, beam_token=0)
        job = bench.Job(args, 0, 0, B, cfg)
        d = job.decoder()
        d.decode_batch(job.e_host, job.Ts, job.N)
        job.ctx.synchronize()
        print("first call: engine", d.get("engine"), "redone", d.get("redone"))
        t0 = time.perf_counter()
        d.decode_batch(job.e_host, job.Ts, job.N)
        job.ctx.synchronize()
        dt = time.perf_counter() - t0
        res.append([d.results(b) for b in range(B)])
        print("engine", d.get("engine"), "threads", d.get("threads"), "redone", d.get("redone"),
              "%.2f ms incl. upload" % (dt * 1e3))
        d.close()
    bad = []
    eq = lambda g, h: (g.score == h.score and g.am == h.am and g.lm == h.lm and np.array_equal(g.tokens, h.tokens)
                       and np.array_equal(g.words, h.words))
    for b in range(B):
        x, y = res[0][b], res[1][b]
        if not (len(x) == len(y) and all(eq(g, h) for g, h in zip(x, y))):
            bad.append(b)
            if len(bad) <= 3:
                print("utt", b, "n", len(x), len(y))
                for i, (g, h) in enumerate(zip(x, y)):
                    if not eq(g, h):
                        df = np.nonzero(np.asarray(g.tokens) != np.asarray(h.tokens))[0]
                        print("  hyp", i, g.score, h.score, g.am, h.am, g.lm, h.lm, "token diff at", df[:5])
                        break
    print("mismatching utterances:", len(bad), bad[:20])


if __name__ == "__main__":
    main()
