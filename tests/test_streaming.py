"""Streaming interface (decodeBegin / decodeStep chunks / getBestHypothesis /
prune / decodeEnd: decoder/Decoder.h:20-34, LexiconFreeDecoder.cpp:188-227,
LexiconDecoder.cpp:285-325, Utils.h:268-342).  The reference has no test for these
calls: the expectations are traces of the compiled reference
itself (tests/golden/make_stream_golden.py -> streaming_expected.json.gz), and the
chain is  HIP path == trace,  oracle == trace,  emulated kernels == trace."""
import gzip
import json
import os

import pytest

import cases
import helpers
import stream_scenarios as ss


@pytest.fixture(scope="module")
def stream_golden():
    with gzip.open(os.path.join(helpers.GOLDEN_DIR, "streaming_expected.json.gz"), "rt") as f:
        return json.load(f)


ALL = list(ss.SCENARIOS)
SMALL = ["lf_ctc_t60_k10", "lx_spell_t60_k12_full"]


@pytest.mark.parametrize("name", ALL)
def test_oracle_streams_like_the_reference(oracle_lib, stream_golden, name):
    """The last link of the chain: the CPU restatement's chunked decode, getBestHypothesis and prune
    against the compiled reference's trace, bit for bit."""
    c = cases.BY_NAME[name]
    chunks, lbs = ss.SCENARIOS[name]
    got = ss.trace_checker(oracle_lib, c, helpers.case_inputs(c), chunks, lbs)
    assert ss.first_difference(stream_golden[name], got) is None, ss.first_difference(stream_golden[name], got)


def _device(session, stream_golden, name, threads=None, tunables=()):
    c = cases.BY_NAME[name]
    chunks, lbs = ss.SCENARIOS[name]
    got, engine = ss.trace_device(session, c, helpers.case_inputs(c), chunks, lbs, threads, tunables)
    d = ss.first_difference(stream_golden[name], got)
    assert d is None, d
    return engine


@pytest.mark.parametrize("name", SMALL)
def test_streaming_emulated(emu_session, stream_golden, name):
    _device(emu_session, stream_golden, name, threads=64)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ALL)
def test_streaming_on_device(gpu_session, stream_golden, name):
    _device(gpu_session, stream_golden, name)


@pytest.mark.parametrize("name", SMALL)
def test_streaming_emulated_over_hbm_workspace(emu_session, stream_golden, name):
    """The same with the beam in an HBM workspace and only the counters (and
    candidate records) in LDS: nothing kept in LDS may be needed by the next chunk's launch."""
    _device(emu_session, stream_golden, name, threads=64, tunables=[("lds_budget", 2048)])


@pytest.mark.gpu
@pytest.mark.parametrize("hot", [1, 2])
@pytest.mark.parametrize("name", ["lf_ctc_t60_k10", "lx_spell_t60_k12_full", "lx_scores_t50", "ng_word_t60_k16_4g"])
def test_streaming_on_device_over_hbm_workspace(gpu_session, stream_golden, name, hot):
    _device(gpu_session, stream_golden, name, tunables=[("lds_budget", 2048), ("hot_level", hot)])


LEX = ["lx_spell_t60_k12_full", "lx_scores_t50", "ng_word_t60_k16_4g"]


@pytest.mark.parametrize("name", LEX[:2])
def test_stream_chunk_decoded_again_after_an_overflow_emulated(emu_session, stream_golden, name):
    """Lexicon streams run on the optimistic geometry (LDS-sized candidate lists, cut-off generation); with the
    cut forced down to K + 1 candidates chunks get flagged, their streams get the saved beam back and decode
    the chunk again on the general path -- the trace must still be the reference's."""
    c = cases.BY_NAME[name]
    _device(emu_session, stream_golden, name, threads=64, tunables=[("cut_m", c["K"] + 1)])
    assert emu_session.last_stream_redone > 0
    _device(emu_session, stream_golden, name, threads=64, tunables=[("stream_optimistic", 0)])
    assert emu_session.last_stream_redone == 0


@pytest.mark.gpu
@pytest.mark.parametrize("name", LEX + ["C3_spell_u0"])
def test_stream_chunk_decoded_again_after_an_overflow(gpu_session, stream_golden, name):
    c = cases.BY_NAME[name]
    _device(gpu_session, stream_golden, name, tunables=[("cut_m", c["K"] + 1)])
    assert gpu_session.last_stream_redone > 0
    _device(gpu_session, stream_golden, name, tunables=[("stream_optimistic", 0)])
    assert gpu_session.last_stream_redone == 0
