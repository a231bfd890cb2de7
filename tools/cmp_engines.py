#!/usr/bin/env python3
"""Differential run of two engine settings of the product library on one batch (GPU).

  python tools/cmp_engines.py [dist] [B] [T] [K] [key=value ...]   (key=value: tunables of the 2nd run)

Decodes the same synthetic batch with the default engine choice and with the given
tunables (default slane=0: the lane-per-slot step) and reports the utterances whose
n-best differ."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from text_amd import _capi, synth

def main():
    a = sys.argv[1:]
    dist = a[0] if len(a) > 0 else "ctc"
    B = int(a[1]) if len(a) > 1 else 256
    T = int(a[2]) if len(a) > 2 else 1000
    K = int(a[3]) if len(a) > 3 else 50
    sets = [x.split("=") for x in a[4:]] or [["slane", "0"]]
    N = 29
    e = synth.batch(dist, B, T, N)
    ctx = _capi.Context()
    lm = _capi.ZeroLM(ctx)
    opt = _capi.make_options(K, N, 25.0)
    res = []
    for sset in ([], sets):
        d = _capi.BatchDecoder(ctx, _capi.LEXFREE, opt, lm, 0, N - 1)
        for k, v in sset:
            d.set(k, int(v))
        d.decode_batch(e, np.full(B, T, dtype=np.int32), N)
        eng = d.get("engine")
        res.append((eng, [d.results(b) for b in range(B)]))
        d.close()
    print("engines", res[0][0], res[1][0])
    bad = []
    for b in range(B):
        x, y = res[0][1][b], res[1][1][b]
        same = len(x) == len(y) and all(g.score == h.score and g.am == h.am and np.array_equal(g.tokens, h.tokens)
                                        for g, h in zip(x, y))
        if not same:
            bad.append(b)
            if len(bad) <= 3:
                print("utt", b, "n", len(x), len(y))
                for i, (g, h) in enumerate(zip(x, y)):
                    if not (g.score == h.score and g.am == h.am and np.array_equal(g.tokens, h.tokens)):
                        df = np.nonzero(np.asarray(g.tokens) != np.asarray(h.tokens))[0]
                        print("  hyp", i, g.score, h.score, g.am, h.am, "first token diff at", df[:5])
                        break
    print("mismatching utterances:", len(bad), bad[:20])

if __name__ == "__main__":
    main()
