"""Streaming interface (decodeBegin / decodeStep chunks / getBestHypothesis /
prune / decodeEnd: decoder/Decoder.h:20-34, LexiconFreeDecoder.cpp:188-227,
LexiconDecoder.cpp:285-325, Utils.h:268-342) against the oracle, which is
itself pinned to the compiled reference by tests/test_oracle_golden.py.  The
reference has no test for these calls, so parity here is
oracle-only."""
import numpy as np
import pytest

import cases
import helpers
from oracle import orclib


def _oracle_decoder(lib, c, inp):
    opt = orclib.make_options(c["K"], c["Kt"], c["thr"], c["lm_weight"], c["word_score"], c["unk_score"],
                              c["sil_score"], c["log_add"], c["crit"])
    N = c["N"]
    blank = N - 1 if c["crit"] == "ctc" else -1
    lm = helpers.checker_lm(lib, c, inp)
    if c["kind"] == "lexfree":
        return lib.lexfree(opt, lm, 0, blank, inp["tr"]), lm, None
    sf, so = inp["lex"]
    scores = inp["scores"]
    if c["lm"] != "zero" and not c["is_lm_token"]:
        scores = helpers.checker_word_scores(lib, lm, inp["W"])
    trie = lib.build_trie(N, 0, sf, so, inp["labels"], scores, smear=1)
    return lib.lexicon(opt, trie, lm, 0, blank, inp["W"], inp["tr"], c["is_lm_token"]), lm, trie


def _same(a, b):
    return (a.score == b.score and a.am == b.am and a.lm == b.lm and np.array_equal(a.tokens, b.tokens)
            and np.array_equal(a.words, b.words))


def run_stream_scenario(session, oracle_lib, name, chunks, look_backs, threads=None, tunables=()):
    c = cases.BY_NAME[name]
    inp = helpers.case_inputs(c)
    N, T = c["N"], c["T"]
    od, olm, otrie = _oracle_decoder(oracle_lib, c, inp)
    d = session.decoder(c, inp, threads)
    for k, v in tunables:
        d.set(k, v)
    oracle_lib.decoder_begin(od)
    d.stream_begin(1, N, T + 4)
    t = 0
    for i, ch in enumerate(chunks):
        ch = min(ch, T - t)
        row = np.ascontiguousarray(inp["e"][t:t + ch])
        oracle_lib.decoder_step(od, orclib._fp(row), ch, N)
        d.stream_step(row, [ch])
        t += ch
        lb = look_backs[i % len(look_backs)]
        want = oracle_lib.best(od, lb, T + 8)
        got = d.best(0, lb, T + 8)
        assert len(want.tokens) == len(got.tokens), "best(lookBack=%d) length after %d frames" % (lb, t)
        if len(want.tokens):
            assert _same(want, got), "best(lookBack=%d) after %d frames" % (lb, t)
        assert oracle_lib.decoder_n_frames_in_buffer(od) == d.frames_in_buffer(0)
        if i % 2 == 1:
            oracle_lib.decoder_prune(od, lb)
            d.stream_prune(lb)
            assert oracle_lib.decoder_n_frames_in_buffer(od) == d.frames_in_buffer(0), "after prune(%d)" % lb
            # the hypotheses still buffered agree (scores are renormalised by prune)
            a = sorted(oracle_lib.collect(od), key=lambda h: -h.score)
            b = d.results(0)
            ok, why = helpers.hyps_equal(a, b)
            assert ok, "buffer after prune(%d) at %d frames: %s" % (lb, t, why)
        if t >= T:
            break
    oracle_lib.decoder_end(od)
    d.stream_end()
    ok, why = helpers.hyps_equal(oracle_lib.collect(od), d.results(0))
    assert ok, why
    d.close()
    oracle_lib.decoder_destroy(od)


SCENARIOS = [
    ("lf_ctc_t60_k10", [7, 9, 1, 12, 20, 30], [0, 2, 0, 5]),
    ("lx_spell_t60_k12_full", [10, 10, 10, 10, 10, 10, 10], [0, 3, 1]),
    ("lx_scores_t50", [5, 15, 10, 10, 10, 10], [2, 0]),  # (no unk: with ZeroLM an <unk> can be
    # emitted at several frames for the same total score, an exact tie in the merge)
]


@pytest.mark.parametrize("name,chunks,lbs", SCENARIOS[:2], ids=[s[0] for s in SCENARIOS[:2]])
def test_streaming_emulated(emu_session, oracle_lib, name, chunks, lbs):
    run_stream_scenario(emu_session, oracle_lib, name, chunks, lbs, threads=64)


@pytest.mark.gpu
@pytest.mark.parametrize("name,chunks,lbs", SCENARIOS + [
    ("C1_ctc_u0", [50, 50, 30, 20, 50], [0, 10, 3]),
    ("ng_word_t60_k16_4g", [10, 10, 10, 10, 10, 10], [0, 2]),
    ("C3_spell_u0", [200, 100, 300, 150, 250], [0, 20, 5, 50]),
], ids=lambda x: x if isinstance(x, str) else None)
def test_streaming_on_device(gpu_session, oracle_lib, name, chunks, lbs):
    run_stream_scenario(gpu_session, oracle_lib, name, chunks, lbs)


@pytest.mark.parametrize("name,chunks,lbs", SCENARIOS[:2], ids=[s[0] for s in SCENARIOS[:2]])
def test_streaming_emulated_over_hbm_workspace(emu_session, oracle_lib, name, chunks, lbs):
    """The same with the beam in an HBM workspace and only the counters (and
    candidate records) in LDS: nothing kept in LDS may be needed by the next chunk's launch."""
    run_stream_scenario(emu_session, oracle_lib, name, chunks, lbs, threads=64, tunables=[("lds_budget", 2048)])


@pytest.mark.gpu
@pytest.mark.parametrize("hot", [1, 2])
@pytest.mark.parametrize("name,chunks,lbs", SCENARIOS + [("ng_word_t60_k16_4g", [10, 10, 10, 10, 10, 10], [0, 2])],
                         ids=lambda x: x if isinstance(x, str) else None)
def test_streaming_on_device_over_hbm_workspace(gpu_session, oracle_lib, name, chunks, lbs, hot):
    run_stream_scenario(gpu_session, oracle_lib, name, chunks, lbs,
                        tunables=[("lds_budget", 2048), ("hot_level", hot)])
