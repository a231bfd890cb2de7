#!/usr/bin/env python3
"""Differential run of two engine settings of the LEXICON decoder on one batch (GPU).

  python tools/cmp_lex_engines.py [dist] [B] [T] [K] [Kt] [key=value ...]   (key=value: tunables of the 2nd run)

ZeroLM over the synthetic lexicon.  Decodes the same batch with
the default engine choice and with the given tunables (default xlane=0: the generic
engine) and reports the utterances whose n-best differ."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from text_amd import _capi, synth


def main():
    a = sys.argv[1:]
    dist = a[0] if len(a) > 0 else "lexspell"
    B = int(a[1]) if len(a) > 1 else 256
    T = int(a[2]) if len(a) > 2 else 1000
    K = int(a[3]) if len(a) > 3 else 50
    Kt = int(a[4]) if len(a) > 4 else 10
    sets = [x.split("=") for x in a[5:]] or [["xlane", "0"], ["ylane", "0"]]
    N = 29
    lex = synth.lexicon()
    e = synth.batch(dist, B, T, N, lexicon=lex if dist == "lexspell" else None)
    ctx = _capi.Context()
    lm = _capi.ZeroLM(ctx)
    W = len(lex[1]) - 1
    ht = _capi.HostTrie(N, 0)
    ht.insert_many(lex[0], lex[1], np.arange(W), np.zeros(W, dtype=np.float32))
    ht.smear(1)
    trie = ht.upload(ctx)
    opt = _capi.make_options(K, Kt, 25.0)
    res = []
    for sset in ([], sets):
        d = _capi.BatchDecoder(ctx, _capi.LEXICON, opt, lm, 0, N - 1, unk=W, trie=trie)
        for k, v in sset:
            d.set(k, int(v))
        Ts = np.full(B, T, dtype=np.int32)
        d.decode_batch(e, Ts, N)
        ctx.synchronize()
        t0 = time.perf_counter()
        d.decode_batch(e, Ts, N)
        ctx.synchronize()
        dt = time.perf_counter() - t0
        res.append((d.get("engine"), [d.results(b) for b in range(B)]))
        print("engine", d.get("engine"), "threads", d.get("threads"), "redone", d.get("redone"),
              "%.2f ms incl. upload" % (dt * 1e3))
        d.close()
    bad = []
    for b in range(B):
        x, y = res[0][1][b], res[1][1][b]
        eq = lambda g, h: (g.score == h.score and g.am == h.am and g.lm == h.lm and np.array_equal(g.tokens, h.tokens)
                           and np.array_equal(g.words, h.words))
        same = len(x) == len(y) and all(eq(g, h) for g, h in zip(x, y))
        if not same:
            bad.append(b)
            if len(bad) <= 3:
                print("utt", b, "n", len(x), len(y))
                for i, (g, h) in enumerate(zip(x, y)):
                    if not eq(g, h):
                        df = np.nonzero(np.asarray(g.tokens) != np.asarray(h.tokens))[0]
                        dw = np.nonzero(np.asarray(g.words) != np.asarray(h.words))[0]
                        print("  hyp", i, g.score, h.score, g.am, h.am, "token diff at", df[:5], "word diff at", dw[:5])
                        break
    print("mismatching utterances:", len(bad), bad[:20])


if __name__ == "__main__":
    main()
