"""CPU (not gpu): the oracle restatement against the golden vectors produced
by the compiled reference (tests/golden/make_golden.py) and against the
reference's own DecoderTest known answers."""
import gzip
import json
import os

import numpy as np
import pytest

import cases
import helpers
from oracle import orclib

SMALL_MED = [c for c in cases.CASES if c["size"] != "large"]
LARGE = [c for c in cases.CASES if c["size"] == "large"]


@pytest.mark.parametrize("c", SMALL_MED, ids=lambda c: c["name"])
def test_oracle_matches_reference_golden(oracle_lib, golden, c):
    hyps = helpers.run_checker(oracle_lib, c)
    ok, why = helpers.check_against_golden(hyps, golden[c["name"]])
    assert ok, why


@pytest.mark.parametrize("c", LARGE, ids=lambda c: c["name"])
def test_oracle_matches_reference_golden_baseline_shapes(oracle_lib, golden, c):
    hyps = helpers.run_checker(oracle_lib, c)
    ok, why = helpers.check_against_golden(hyps, golden[c["name"]])
    assert ok, why


def test_appendix_b_known_answers(golden):
    """SURVEY.md Appendix B numbers are what the committed fixture holds."""
    for name, (n, hsh, top) in cases.APPENDIX_B.items():
        g = golden[name]
        assert g["n"] == n and int(g["hash"], 16) == hsh
        assert float.fromhex(g["scores"][0][0]) == float.fromhex(top)


def _decodertest_inputs():
    d = os.path.join(helpers.GOLDEN_DIR, "decodertest")
    rd = lambda n: gzip.open(os.path.join(d, n + ".gz"), "rb").read()
    from golden.make_golden import parse_lexicon_dump
    lex = parse_lexicon_dump(rd("lexicon_dump.txt").decode())
    lex["letters"] = rd("letters.lst").decode().split() + ["<1>"]
    TN = np.frombuffer(rd("TN.bin"), dtype=np.int32)
    em = np.frombuffer(rd("emission.bin"), dtype=np.float32).copy()
    tr = np.frombuffer(rd("transition.bin"), dtype=np.float32).copy()
    return lex, TN, em, tr, rd("lm.arpa")


def test_decodertest_known_answers(oracle_lib, tmp_path):
    """flashlight/lib/text/test/decoder/DecoderTest.cpp:107-120,148-155,184-194
    replayed through the oracle (ARPA LM standing in for KenLM)."""
    from golden.make_golden import run_decodertest
    lex, TN, em, tr, arpa = _decodertest_inputs()
    p = tmp_path / "lm.arpa"
    p.write_bytes(arpa)
    got = run_decodertest(oracle_lib, lex, str(p), TN, em, tr)
    tgt_lm = [-1.05971, -4.19448, -3.33383, -2.76726, -1.16237, -4.64589]
    tgt_trie = [-1.05971, -2.87742, -2.64553, -3.05081, -1.05971, -3.08968]
    tgt_hyp = [-284.0998, -284.108, -284.119, -284.127, -284.296]
    assert np.allclose(got["lm_scores"], tgt_lm, atol=1e-5)
    assert abs(got["lm_total"] - (-19.5123)) < 1e-4
    assert np.allclose(got["trie_scores"], tgt_trie, atol=1e-5)
    assert got["nbest"]["n"] == 16
    for s, t in zip(got["nbest"]["scores"], tgt_hyp):
        assert abs(float.fromhex(s[0]) - t) < 1e-3
    # and bit-identical to what the compiled reference produced for this flow
    exp = json.load(open(os.path.join(helpers.GOLDEN_DIR, "decodertest", "expected.json")))
    assert got["nbest"] == exp["nbest"]
    assert got["lm_scores"] == exp["lm_scores"] and got["trie_scores"] == exp["trie_scores"]
