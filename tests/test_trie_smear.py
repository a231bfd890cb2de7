"""CPU: Trie::smear (decoder/Trie.cpp:79-101) of the host trie behind the C ABI
(fltx_htrie_smear) and of the oracle's restatement against golden values produced
by the compiled reference (tests/golden/make_smear_golden.py).  SmearingMode::LOGADD
folds the children in unordered_map iteration order with a float narrowing per step,
so only a value taken from the reference build pins it."""
import gzip
import json
import os

import numpy as np
import pytest

import helpers
from oracle import orclib
from text_amd import _capi, synth

GOLD = json.load(gzip.open(os.path.join(helpers.GOLDEN_DIR, "trie_smear.json.gz"), "rt"))
N = 29


@pytest.mark.parametrize("key", sorted(GOLD))
def test_host_trie_smear_matches_reference(key):
    g = GOLD[key]
    sf, so = helpers.lexicon(g["W"], g["lex_seed"])
    scores = synth.floats(g["score_seed"], g["W"], -6.0, 0.0)
    lib = _capi.default_lib()
    t = _capi.HostTrie(N, 0, lib=lib)
    t.insert_many(sf, so, np.arange(g["W"]), scores)
    t.smear(g["mode"])
    for p, want in zip(g["probes"], g["max_score"]):
        got = t.search(p)
        assert got is not None
        assert float(np.float32(got["max_score"])).hex() == want, (key, p)


@pytest.mark.parametrize("key", sorted(GOLD))
def test_oracle_trie_smear_matches_reference(oracle_lib, key):
    g = GOLD[key]
    sf, so = helpers.lexicon(g["W"], g["lex_seed"])
    scores = synth.floats(g["score_seed"], g["W"], -6.0, 0.0)
    t = oracle_lib.build_trie(N, 0, sf, so, np.arange(g["W"]), scores, smear=g["mode"])
    ms = np.zeros(1, dtype=np.float32)
    nl = np.zeros(1, dtype=np.int32)
    for p, want in zip(g["probes"], g["max_score"]):
        a = np.asarray(p, dtype=np.int32)
        assert oracle_lib.trie_search(t, orclib._ip(a), len(a), orclib._fp(ms), orclib._ip(nl))
        assert float(ms[0]).hex() == want, (key, p)
    oracle_lib.trie_destroy(t)
