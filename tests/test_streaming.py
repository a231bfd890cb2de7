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


def _random_lexfree_streams(session, oracle_lib, every):
    """Lexicon-free stream chunks run on the lane = LM state engine (fltx_slane.h, ST): beam in and out in the
    lane-per-slot engine's parked format, ids from its (parent, token) -> id table.  Random options (beams 1 .. 64,
    thresholds 2 .. inf, token beams, silScore, CTC and ASG) in chunks with getBestHypothesis / prune between them,
    event by event against the oracle (itself pinned to the reference's traces above)."""
    import itertools
    bad, ran = [], 0
    grid = itertools.product([1, 3, 10, 50, 64], [2.0, 25.0, float("inf")], [None, 5], [12, 29], [7, 60],
                             ["ctc", "uniform"], [0.0, -0.6], ["ctc", "asg"])
    for i, (K, thr, Kt, N, T, dist, sil, crit) in enumerate(grid):
        if i % every:
            continue
        c = cases.case("ss%d" % i, dist=dist, u=700 + i, T=T, N=N, K=K, Kt=Kt, thr=thr, sil_score=sil, crit=crit,
                       trans_seed=(30 + i) if crit == "asg" else None)
        inp = helpers.case_inputs(c)
        # (one-frame chunks: every frame is a launch's last -- a state entered again there is parked before the
        # next frame's relink could run)
        chunks, lbs = [[3, 9, 1, 12, 20, 30], [1] * 60, [25, 25, 25]][i % 3], [[0, 2, 0, 5], [0], [3, 1]][i % 3]
        want = ss.trace_checker(oracle_lib, c, inp, chunks, lbs)
        if ss.has_ties(want):
            continue  # (equal scores in a buffer: the order of the reference's own result is not defined)
        got, _ = ss.trace_device(session, c, inp, chunks, lbs)
        ran += 1
        d = ss.first_difference(want, got)
        if d:
            bad.append(({k: c[k] for k in ("K", "thr", "Kt", "N", "T", "dist", "sil_score", "crit")}, d))
    return ran, bad


def _random_token_lm_streams(session, oracle_lib, every):
    """... and with a token-level n-gram LM (round 6: slaneUtterance<.., ST, TL> -- state ids from the generic engine's
    table, whose kernels do begin / end / prune / getBestHypothesis; a state's context a row of the dense table): the LM
    scores of every event too.  -> (streams compared, of them with the chunks on the token-LM variant, mismatches)"""
    import itertools
    bad, ran, on_t = [], 0, 0
    grid = itertools.product([1, 3, 10, 50, 64], [2.0, 25.0, float("inf")], [None, 5], [12, 29], [7, 60],
                             ["ctc", "uniform"], [0.0, -0.6], ["ctc", "asg"], [2, 3, 4])
    for i, (K, thr, Kt, N, T, dist, sil, crit, order) in enumerate(grid):
        if i % every:
            continue
        c = cases.case("ts%d" % i, dist=dist, u=900 + i, T=T, N=N, K=K, Kt=Kt, thr=thr, sil_score=sil, crit=crit,
                       trans_seed=(30 + i) if crit == "asg" else None, lm=("ngram", order, 80 + i % 3),
                       lm_weight=[0.8, 1.5, 0.4][i % 3])
        inp = helpers.case_inputs(c)
        chunks, lbs = [[3, 9, 1, 12, 20, 30], [1] * 60, [25, 25, 25]][i % 3], [[0, 2, 0, 5], [0], [3, 1]][i % 3]
        want = ss.trace_checker(oracle_lib, c, inp, chunks, lbs)
        if ss.has_ties(want):
            continue
        d = session.decoder(c, inp)
        d.stream_begin(1, N, T + 4)
        on_t += d.get("tstream")
        d.close()
        got, _ = ss.trace_device(session, c, inp, chunks, lbs)
        ran += 1
        diff = ss.first_difference(want, got)
        if diff:
            bad.append(({k: c[k] for k in ("K", "thr", "Kt", "N", "T", "dist", "sil_score", "crit", "lm")}, diff))
    return ran, on_t, bad


def test_token_lm_stream_chunks_on_the_lane_state_engine_emulated(emu_session, oracle_lib):
    ran, on_t, bad = _random_token_lm_streams(emu_session, oracle_lib, 61)
    assert ran > 20 and on_t == ran and not bad, (ran, on_t, bad[:3])
    # ... and with the variant switched off: the generic engine's frames, same traces
    c = cases.case("ts_off", dist="ctc", u=990, T=40, N=29, K=10, lm=("ngram", 3, 81), lm_weight=0.8)
    inp = helpers.case_inputs(c)
    want = ss.trace_checker(oracle_lib, c, inp, [7, 13, 20], [0, 2])
    got, _ = ss.trace_device(emu_session, c, inp, [7, 13, 20], [0, 2], tunables=[("sstream", 0)])
    assert ss.first_difference(want, got) is None


@pytest.mark.gpu
def test_token_lm_stream_chunks_on_the_lane_state_engine(gpu_session, oracle_lib):
    ran, on_t, bad = _random_token_lm_streams(gpu_session, oracle_lib, 3)
    assert ran > 400 and on_t == ran and not bad, (ran, on_t, bad[:3])


def test_lexfree_stream_chunks_on_the_lane_state_engine_emulated(emu_session, oracle_lib, stream_golden):
    c = cases.BY_NAME["lf_ctc_t60_k10"]
    d = emu_session.decoder(c, helpers.case_inputs(c))
    d.stream_begin(1, c["N"], 64)
    assert d.get("sstream") > 0  # the engine is chosen for the stream's chunks ...
    d.set("sstream", 0)
    d.stream_begin(1, c["N"], 64)
    assert d.get("sstream") == 0  # ... and can be switched off
    d.close()
    _device(emu_session, stream_golden, "lf_ctc_t60_k10")
    ran, bad = _random_lexfree_streams(emu_session, oracle_lib, 97)
    assert ran > 8 and not bad, bad[:3]


@pytest.mark.gpu
def test_lexfree_stream_chunks_on_the_lane_state_engine(gpu_session, oracle_lib, stream_golden):
    ran, bad = _random_lexfree_streams(gpu_session, oracle_lib, 3)
    assert ran > 280 and not bad, bad[:3]
    # the lane-per-slot step (sstream = 0) still serves the same streams
    _device(gpu_session, stream_golden, "C1_ctc_u0", tunables=[("sstream", 0)])


@pytest.mark.gpu
@pytest.mark.parametrize("sstream", [1, 0], ids=["lane-state-engine", "lane-per-slot-step"])
def test_long_stream_through_a_small_buffer(gpu_session, oracle_lib, sstream):
    """1 000 frames in 50-frame chunks with prune(0) through a 208-frame buffer (bench.py's streaming leg): LM states
    are created for as long as the stream runs, so the id tables follow `stream_total_frames` (default: at least 2 048
    frames), not max_frames -- round 2 sized them by the buffer and such a stream ended with a full table."""
    import numpy as np
    c = cases.case("cap", T=1000, K=50, N=29, u=5)
    inp = helpers.case_inputs(c)
    want = ss.trace_checker(oracle_lib, c, inp, [50] * 20, [0])
    d = gpu_session.decoder(c, inp)
    d.set("sstream", sstream)
    d.stream_begin(1, 29, 208)
    for k in range(20):
        d.stream_step(np.ascontiguousarray(inp["e"][k * 50:(k + 1) * 50]), [50])
        d.stream_prune(0)
    d.stream_end()
    got = helpers.encode_hyps(d.results(0), True)
    d.close()
    assert got == want[-1]["final"]


def _ragged_parallel_streams(session, oracle_lib, name, sets=()):
    """Several streams of one decoder fed chunks of different lengths (some of them empty) in the same calls: every
    stream's getBestHypothesis / final n-best against the oracle run on that stream alone."""
    import numpy as np
    from text_amd import synth
    c = cases.BY_NAME[name]
    inp = helpers.case_inputs(c)
    N, B = c["N"], 4
    Ts = [c["T"], c["T"] // 2, 7, c["T"] - 3]
    lex = inp["lex"] if c["dist"] == "lexspell" else None
    es = [synth.emissions(c["dist"], 900 + b, Ts[b], N, lexicon=lex) for b in range(B)]
    plan = [[5, 0, 3, 9], [7, 11, 0, 1], [0, 4, 4, 20], [30, 30, 30, 30], [100, 100, 100, 100]]
    ods = [ss.checker_decoder(oracle_lib, c, inp) for _ in range(B)]
    for od, _, _ in ods:
        oracle_lib.decoder_begin(od)
    d = session.decoder(c, inp)
    for k, v in sets:
        d.set(k, v)
    d.stream_begin(B, N, max(Ts) + 4)
    done = [0] * B
    for step, row in enumerate(plan):
        take = [min(row[b], Ts[b] - done[b]) for b in range(B)]
        parts = [es[b][done[b]:done[b] + take[b]] for b in range(B)]
        flat = np.concatenate([p.reshape(-1) for p in parts]) if sum(take) else np.zeros(1, np.float32)
        d.stream_step(flat.astype(np.float32), take)
        for b in range(B):
            if take[b]:
                oracle_lib.decoder_step(ods[b][0], orclib_fp(np.ascontiguousarray(parts[b])), take[b], N)
            done[b] += take[b]
        for b in range(B):
            want = oracle_lib.best(ods[b][0], step % 3, max(Ts) + 8)
            got = d.best(b, step % 3, max(Ts) + 8)
            assert ss._enc_one(want) == ss._enc_one(got), "stream %d after call %d" % (b, step)
        if step == 2:
            d.stream_prune(1)
            for od, _, _ in ods:
                oracle_lib.decoder_prune(od, 1)
    d.stream_end()
    for b in range(B):
        oracle_lib.decoder_end(ods[b][0])
        ok, why = helpers.hyps_equal(oracle_lib.collect(ods[b][0]), d.results(b))
        assert ok, "stream %d: %s" % (b, why)
        oracle_lib.decoder_destroy(ods[b][0])
    d.close()


def orclib_fp(a):
    from oracle import orclib
    return orclib._fp(a)


def test_ragged_parallel_streams_emulated(emu_session, oracle_lib):
    _ragged_parallel_streams(emu_session, oracle_lib, "lf_ctc_t60_k10")


@pytest.mark.parametrize("name", ["lf_ctc_t60_k10", "ng_tok_lexfree_t40", "hl_lastword_lexfree"])
def test_ragged_parallel_streams_with_recycled_ids_emulated(emu_session, oracle_lib, name):
    """... with the LM-state ids renumbered before every chunk (round 5), and with a user-defined LM answering for
    several streams at once (a host LM numbers its own states: nothing to recycle on the device)."""
    _ragged_parallel_streams(emu_session, oracle_lib, name, (("compact_always", 1), ("threads", 64)))


@pytest.mark.gpu
@pytest.mark.parametrize("name,sets", [("lf_ctc_t60_k10", ()), ("lf_ctc_t60_k10", (("sstream", 0),)),
                                       ("C1_ctc_u0", ()), ("lf_asg_t40_n29_kt7", ()), ("lx_spell_t60_k12_full", ()),
                                       ("ng_word_t60_k16_4g", ()), ("ng_word_t60_k16_4g", (("cut_m", 17),)),
                                       ("ng_word_t60_k16_4g", (("cut_m", 17), ("stream_defer", 0))),
                                       ("lf_ctc_t60_k10", (("compact_always", 1),)),
                                       ("lf_ctc_t60_k10", (("compact_always", 1), ("sstream", 0))),
                                       ("lf_asg_t40_n29_kt7", (("compact_always", 1),)),
                                       ("lx_spell_t60_k12_full", (("compact_always", 1),)),
                                       ("ng_word_t60_k16_4g", (("compact_always", 1),)),
                                       ("ng_word_t60_k16_4g", (("compact_always", 1), ("cut_m", 17))),
                                       ("ng_tok_lexfree_t40", (("compact_always", 1),)),
                                       ("hl_lastword_lexfree", ()), ("hl_lastword_word", ()), ("hl_lastword_toklex", ())],
                         ids=lambda x: x if isinstance(x, str) else None)
def test_ragged_parallel_streams(gpu_session, oracle_lib, name, sets):
    _ragged_parallel_streams(gpu_session, oracle_lib, name, sets)


def _device_chunk_reuse(session, golden, name, on_gpu, threads=None):
    """A lexicon stream fed from ONE device buffer that the caller overwrites after every step (having waited for the
    context's stream): a chunk whose candidate list overflowed (cut forced down to K + 1) is decoded again, and that
    second pass must not read the caller's buffer after fltx_stream_step has returned (round-3 advisor finding: it was
    deferred to the next call).  No prune in between: the final n-best is the offline one."""
    import ctypes
    import numpy as np
    c = cases.BY_NAME[name]
    inp = helpers.case_inputs(c)
    N, T = c["N"], c["T"]
    junk = np.full(10 * N, -1.0e3, dtype=np.float32)
    if on_gpu:
        hip = ctypes.CDLL("libamdhip64.so.7")  # (the runtime libfltx.so is linked against: already mapped)
        hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
        hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        hip.hipFree.argtypes = [ctypes.c_void_p]
        buf = ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(buf), 4 * 10 * N) == 0
        ptr = buf.value

        def put(a):
            assert hip.hipMemcpy(buf, a.ctypes.data, 4 * a.size, 1) == 0  # (synchronous H2D)
    else:  # the emulator's "device" memory is host memory
        hbuf = np.zeros(10 * N, dtype=np.float32)
        ptr = hbuf.ctypes.data

        def put(a):
            hbuf[:a.size] = a
    d = session.decoder(c, inp, threads)
    d.set("cut_m", c["K"] + 1)
    d.stream_begin(1, N, T + 4)
    try:
        for t in range(0, T, 10):
            chunk = np.ascontiguousarray(inp["e"][t:t + 10], dtype=np.float32).reshape(-1)
            put(chunk)
            d.stream_step(None, [chunk.size // N], device_ptr=ptr)
            # the chunk is read in stream order on the context's stream (include/fltx.h): a caller writing from
            # elsewhere waits for that stream first -- and after that, nothing may read the buffer any more
            session.ctx.synchronize()
            put(junk)
        d.stream_end()
        got = d.results(0)
        assert d.get("stream_redone") > 0
    finally:
        d.close()
        if on_gpu:
            hip.hipFree(buf)
    ok, why = helpers.check_against_golden(got, golden[name])
    assert ok, why


def test_device_chunk_buffer_reused_between_steps_emulated(emu_session, golden):
    _device_chunk_reuse(emu_session, golden, "lx_spell_t60_k12_full", False, threads=64)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["lx_spell_t60_k12_full", "ng_word_t60_k16_4g"])
def test_device_chunk_buffer_reused_between_steps(gpu_session, golden, name):
    _device_chunk_reuse(gpu_session, golden, name, True)
