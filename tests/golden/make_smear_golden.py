#!/usr/bin/env python3
"""Golden vector for Trie::smear(SmearingMode::LOGADD) (decoder/Trie.cpp:79-101), generated from
the UNMODIFIED reference (oracle/_ref/libfltref.so; dev container only).

LOGADD folds a node's children in unordered_map iteration order with a float narrowing after
every step (Trie.cpp:84-94), so the result is pinned by the reference build, not by the maths:
300- and 3000-word synthetic lexicons with random label scores, the smeared maxScore of every
word's end node, its parent and the root's children, as float hex.  MAX mode rides along.
Output: tests/golden/trie_smear.json.gz (data only)."""
import gzip
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import helpers  # noqa: E402
from oracle import orclib  # noqa: E402
from text_amd import synth  # noqa: E402

N = 29


def probes(sf, so):
    seen, out = set(), []
    for w in range(len(so) - 1):
        sp = [int(x) for x in sf[so[w]:so[w + 1]]]
        for k in (len(sp), len(sp) - 1, 1, 2):
            key = tuple(sp[:k])
            if k >= 1 and key not in seen:
                seen.add(key)
                out.append(list(key))
    return out


def main():
    ref = orclib.load("ref")
    out = {}
    for W, seed, sseed in ((300, 4242, 77), (3000, 99, 78)):
        sf, so = helpers.lexicon(W, seed)
        scores = synth.floats(sseed, W, -6.0, 0.0)
        pr = probes(sf, so)
        for mode, name in ((2, "logadd"), (1, "max")):
            t = ref.build_trie(N, 0, sf, so, np.arange(W), scores, smear=mode)
            vals = []
            ms = np.zeros(1, dtype=np.float32)
            nl = np.zeros(1, dtype=np.int32)
            for p in pr:
                a = np.asarray(p, dtype=np.int32)
                found = ref.trie_search(t, orclib._ip(a), len(a), orclib._fp(ms), orclib._ip(nl))
                assert found
                vals.append(float(ms[0]).hex())
            ref.trie_destroy(t)
            out["W%d_%s" % (W, name)] = {"W": W, "lex_seed": seed, "score_seed": sseed, "mode": mode,
                                          "probes": pr, "max_score": vals}
            print(W, name, len(pr), vals[:3])
    with gzip.open(os.path.join(HERE, "trie_smear.json.gz"), "wt") as f:
        json.dump(out, f, separators=(",", ":"))


if __name__ == "__main__":
    main()
