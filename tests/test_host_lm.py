"""User-defined LM subclasses decode on the device (missing #1; reference extension point:
decoder/lm/LM.h:61-85, bindings/python/flashlight/lib/text/_decoder.cpp:39-56).  The beam search runs in the kernels,
the LM's start / score / finish run on the host once per distinct question per frame (include/fltx.h
fltx_lm_host_create).  CPU: the emulated kernels; -m gpu: the HIP path."""
import numpy as np
import pytest

import cases
import helpers
import host_lms
from text_amd import _capi


def _decoder(sess, c, inp, lm):
    return sess.decoder(c, inp, lm=lm)


def _run(sess, c, inp, lm, sets=None):
    d = _decoder(sess, c, inp, lm)
    for k, v in (sets or {}).items():
        d.set(k, v)
    d.decode_batch(inp["e"], [c["T"]], c["N"])
    out = d.results(0)
    info = dict(engine=d.get("engine"), asked=d.get("hlm_asked"), distinct=d.get("hlm_distinct"))
    d.close()
    return out, info


EMU_SETS = {"threads": 64}  # (the emulator runs a workgroup's threads as host threads, a launch per frame)
ZERO_CASES = ["lf_ctc_t20_k4", "lf_ctc_t60_k10_kt5", "lf_ctc_t60_k10_logadd", "lf_ctc_t0", "lf_ctc_t1",
              "lf_asg_t40_n29_kt7", "lx_spell_t40_k8", "lx_spell_t60_k12_logadd", "lx_spell_unk", "lx_asg_t40",
              "lx_tokenlm_t40", "lx_scores_t50", "lx_t0"]
NGRAM_CASES = ["ng_word_t60_k16_4g", "ng_tok_lexfree_t40", "ng_word_unk_t40", "ng_word_logadd_t40",
               "ng_tok_lexfree_kt8", "ng_tok_lexicon_t40"]


def _zero_clone(sess, golden, name, sets=None):
    c = cases.BY_NAME[name]
    inp = helpers.case_inputs(c)
    lm = _capi.HostLM(host_lms.PyZeroLM(), lib=sess.lib)
    got, info = _run(sess, c, inp, lm, sets)
    assert info["engine"] == 0, info
    tol = 1e-9 if c["log_add"] else 0.0
    ok, why = helpers.check_against_golden(got, golden[name], score_tol=tol)
    assert ok, why
    if c["T"] > 1:
        assert 0 < info["distinct"] <= info["asked"]
    lm.close()


def _ngram_wrapper(sess, golden, oracle_lib, name, gpu, sets=None):
    """A Python LM over the ARPA tables gives the n-best of the device n-gram path, the oracle's and the reference's."""
    c = cases.BY_NAME[name]
    inp = helpers.case_inputs(c)
    arpa = sess.lm_for(c, inp)
    want_dev = sess.run(c, inp)
    user = host_lms.PyNgramLM(arpa)
    lm = _capi.HostLM(user, lib=sess.lib)
    got, info = _run(sess, c, inp, lm, sets)
    assert info["engine"] == 0 and user.calls > 0
    tol = (1e-5 if gpu else 1e-9) if c["log_add"] else 0.0
    ok, why = helpers.hyps_equal(want_dev, got, score_tol=tol)
    assert ok, "vs the device n-gram path: " + why
    ok, why = helpers.check_against_golden(got, golden[name], score_tol=tol)
    assert ok, "vs the reference's golden: " + why
    ok, why = helpers.hyps_equal(helpers.run_checker(oracle_lib, c, inp), got, score_tol=tol)
    assert ok, "vs the oracle: " + why
    lm.close()


@pytest.mark.parametrize("name", ZERO_CASES)
def test_emulated_python_zero_lm_clone_matches_golden(emu_session, golden, name):
    _zero_clone(emu_session, golden, name, EMU_SETS)


@pytest.mark.parametrize("name", NGRAM_CASES)
def test_emulated_python_lm_over_arpa_tables(emu_session, golden, oracle_lib, name):
    _ngram_wrapper(emu_session, golden, oracle_lib, name, gpu=False, sets=EMU_SETS)


def _raises(sess):
    c = cases.BY_NAME["lf_ctc_t20_k4"]
    inp = helpers.case_inputs(c)
    lm = _capi.HostLM(host_lms.FailingLM(after=40), lib=sess.lib)
    d = _decoder(sess, c, inp, lm)
    with pytest.raises(KeyError, match="on purpose"):
        d.decode_batch(inp["e"], [c["T"]], c["N"])
    # the decoder is usable afterwards
    lm2 = _capi.HostLM(host_lms.PyZeroLM(), lib=sess.lib)
    d2 = _decoder(sess, c, inp, lm2)
    d2.decode_batch(inp["e"], [c["T"]], c["N"])
    assert len(d2.results(0)) > 0
    d.close()
    d2.close()


def test_emulated_user_lm_exception_surfaces(emu_session):
    _raises(emu_session)


def _batch_of_ragged_utterances(sess, oracle_lib):
    """Several utterances of different lengths in one call: one exchange per frame serves the whole batch."""
    from text_amd import synth
    c = cases.case("hlm_batch", dist="ctc", T=0, N=12, K=6, Kt=5, u=900, lm=("ngram", 3, 77), lm_weight=0.9,
                   sil_score=-0.2)
    Ts = [17, 0, 30, 1, 9]
    inp0 = dict(e=None, tr=None, lex=None)
    arpa = sess.lm_for(c, inp0)
    es = [synth.emissions("ctc", 900 + i, t, c["N"]) for i, t in enumerate(Ts)]
    flat = np.concatenate([e.reshape(-1) for e in es]).astype(np.float32)
    lm = _capi.HostLM(host_lms.PyNgramLM(arpa), lib=sess.lib)
    d = _decoder(sess, c, inp0, lm)
    d.decode_batch(flat, Ts, c["N"])
    for b, t in enumerate(Ts):
        cb = dict(c, T=t)
        want = helpers.run_checker(oracle_lib, cb, dict(e=es[b], tr=None, lex=None))
        ok, why = helpers.hyps_equal(want, d.results(b))
        assert ok, "utterance %d: %s" % (b, why)
    d.close()
    lm.close()


def test_emulated_ragged_batch_with_a_user_lm(emu_session, oracle_lib):
    _batch_of_ragged_utterances(emu_session, oracle_lib)


# ---- the same on the device --------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("name", ZERO_CASES + ["C1_ctc_u0"])
def test_python_zero_lm_clone_matches_golden(gpu_session, golden, name):
    _zero_clone(gpu_session, golden, name)


@pytest.mark.gpu
@pytest.mark.parametrize("name", NGRAM_CASES)
def test_python_lm_over_arpa_tables(gpu_session, golden, oracle_lib, name):
    _ngram_wrapper(gpu_session, golden, oracle_lib, name, gpu=True)


@pytest.mark.gpu
def test_user_lm_exception_surfaces(gpu_session):
    _raises(gpu_session)


@pytest.mark.gpu
def test_ragged_batch_with_a_user_lm(gpu_session, oracle_lib):
    _batch_of_ragged_utterances(gpu_session, oracle_lib)


# ---- streams: decodeStep chunks, getBestHypothesis(lookBack), prune(lookBack) with a user LM -----------------------
STREAM_CASES = ["hl_lastword_lexfree", "hl_lastword_toklex", "hl_lastword_word", "hl_lastword_asg"]


def _stream(sess, oracle_lib, name, threads=None):
    """Event by event against the oracle's trace (the oracle streams like the compiled reference:
    tests/test_streaming.py, and decodes this LM like it: tests/test_oracle_golden.py)."""
    import stream_scenarios as ss
    c = cases.BY_NAME[name]
    inp = helpers.case_inputs(c)
    chunks, lbs = [7, 9, 1, 12, 20, 30], [0, 2, 0, 5]
    want = ss.trace_checker(oracle_lib, c, inp, chunks, lbs)
    assert not ss.has_ties(want)
    lm = sess.lm_for(c, inp)
    before = lm.released
    got, engine = ss.trace_device(sess, c, inp, chunks, lbs, threads)
    d = ss.first_difference(want, got)
    assert d is None, d
    assert engine == 0
    assert lm.released > before  # prune() released LM states the beam no longer holds


@pytest.mark.parametrize("name", STREAM_CASES[:2])
def test_emulated_stream_with_a_user_lm(emu_session, oracle_lib, name):
    _stream(emu_session, oracle_lib, name, threads=64)


@pytest.mark.gpu
@pytest.mark.parametrize("name", STREAM_CASES)
def test_stream_with_a_user_lm(gpu_session, oracle_lib, name):
    _stream(gpu_session, oracle_lib, name)
