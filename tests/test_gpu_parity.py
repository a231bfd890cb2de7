"""GPU (-m gpu): the product path -- text_amd/lib/libfltx.so through the C ABI
on a real MI355X -- against (a) the golden vectors produced by the compiled
reference and (b) the oracle on the same seeded inputs.  Integer outputs
(tokens, words) exact; scores bit-identical for max-merge, 1e-5 for logAdd
(device libm is not glibc: SURVEY.md H3)."""
import numpy as np
import pytest

import cases
import helpers

pytestmark = pytest.mark.gpu

ALL = cases.CASES


@pytest.mark.parametrize("c", ALL, ids=lambda c: c["name"])
def test_hip_matches_reference_golden(gpu_session, golden, c):
    hyps = gpu_session.run(c)
    tol = 1e-5 if c["log_add"] else 0.0
    ok, why = helpers.check_against_golden(hyps, golden[c["name"]], tol)
    assert ok, why


@pytest.mark.parametrize("threads", [64, 128, 256, 512, 1024])
def test_threads_per_utterance_do_not_change_results(gpu_session, golden, threads):
    for name in ("C1_ctc_u0", "lx_spell_t60_k12_full", "lf_asg_t40_n29_kt7"):
        c = cases.BY_NAME[name]
        hyps = gpu_session.run(c, threads=threads)
        ok, why = helpers.check_against_golden(hyps, golden[name])
        assert ok, "%s @%d threads: %s" % (name, threads, why)


def test_batch_of_ragged_utterances_matches_oracle(gpu_session, oracle_lib):
    """B utterances of different length in one launch == B single decodes of
    the oracle (utterances are independent: SURVEY.md section 8e)."""
    from text_amd import synth
    c = dict(cases.BY_NAME["C1_ctc_u0"])
    Ts = [0, 1, 37, 200, 123, 64, 5, 199]
    N = c["N"]
    embs = [synth.emissions("ctc", 100 + i, T, N) for i, T in enumerate(Ts)]
    flat = np.concatenate([e.reshape(-1) for e in embs]) if sum(Ts) else np.zeros(0, np.float32)
    d = gpu_session.decoder(c, dict(tr=None))
    d.decode_batch(flat, Ts, N)
    for b, T in enumerate(Ts):
        cb = dict(c, T=T)
        want = helpers.run_checker(oracle_lib, cb, dict(e=embs[b], tr=None, lex=None))
        ok, why = helpers.hyps_equal(want, d.results(b))
        assert ok, "utterance %d (T=%d): %s" % (b, T, why)
    d.close()


def test_streaming_chunks_equal_offline(gpu_session, golden):
    """decodeBegin + several decodeStep chunks + decodeEnd == decode()."""
    c = cases.BY_NAME["C1_ctc_u0"]
    inp = helpers.case_inputs(c)
    d = gpu_session.decoder(c, inp)
    d.stream_begin(1, c["N"], c["T"])
    t = 0
    for chunk in (1, 7, 64, 100, 28):
        d.stream_step(np.ascontiguousarray(inp["e"][t:t + chunk]), [chunk])
        t += chunk
    assert t == c["T"]
    d.stream_end()
    ok, why = helpers.check_against_golden(d.results(0), golden[c["name"]])
    assert ok, why
    d.close()


def test_c2_batch_property_checks(gpu_session, golden):
    """Full C2 batch shape (B=256, T=1000, K=50): utterances 0 and 255 equal the
    reference golden; every utterance returns K sorted hypotheses whose token
    rows start and end with sil and whose am score equals the sum of the
    emissions along the path (size-independent properties)."""
    from text_amd import synth
    c = cases.BY_NAME["C2_ctc_u0"]
    B, T, N = 256, c["T"], c["N"]
    e = synth.batch("ctc", B, T, N)
    d = gpu_session.decoder(c, dict(tr=None))
    d.decode_batch(e, [T] * B, N)
    for b, name in ((0, "C2_ctc_u0"), (255, "C2_ctc_u255")):
        ok, why = helpers.check_against_golden(d.results(b), golden[name])
        assert ok, "%s: %s" % (name, why)
    for b in range(0, B, 17):
        hyps = d.results(b)
        assert len(hyps) == c["K"]
        sc = [h.score for h in hyps]
        assert all(x > y for x, y in zip(sc, sc[1:]))
        for h in hyps[:5]:
            assert h.tokens[0] == 0 and h.tokens[-1] == 0 and len(h.tokens) == T + 2
            path = e[b][np.arange(T), h.tokens[1:-1]].astype(np.float64)
            acc = 0.0
            for v in path:
                acc += v
            assert acc == h.am == h.score
    d.close()


LEXFREE = [c for c in cases.CASES if c["kind"] == "lexfree"]


@pytest.mark.parametrize("mode", ["hash", "dense", "lean"])
@pytest.mark.parametrize("c", LEXFREE, ids=lambda c: c["name"])
def test_generic_engine_equals_lean_kernel(gpu_session, golden, c, mode):
    """Lexicon-free + ZeroLM frames normally run the lane-per-slot step
    (fltx_lane.h, beam <= 64) or the lean register-resident step
    (fltx_lean.h).  The generic engine -- with its dense merge, and with the
    hash merge the lexicon decoder uses -- and the lean step where the lane step
    is the default must give the same n-best."""
    inp = helpers.case_inputs(c)
    d = gpu_session.decoder(c, inp)
    if mode == "lean":
        d.set("lane", 0)
        d.set("lane_groups", -1)  # (beams 65-512 are fltx_mlane.h's otherwise)
    else:
        d.set("lean", 0)
    if mode == "hash":
        d.set("dense", 0)
    d.decode_batch(inp["e"], [c["T"]], c["N"])
    assert d.get("engine") == {"hash": 0, "dense": 1, "lean": 2}[mode] or c["lm"] != "zero"
    tol = 1e-5 if c["log_add"] else 0.0
    ok, why = helpers.check_against_golden(d.results(0), golden[c["name"]], tol)
    d.close()
    assert ok, why


def test_decodertest_replay_on_device(gpu_session, tmp_path):
    """The reference's only end-to-end decode test
    (flashlight/lib/text/test/decoder/DecoderTest.cpp:57-195: LexiconDecoder +
    KenLM 3-gram + ASG, beam 2500, 26k-word lexicon, T=235) replayed through the
    HIP path: ARPA loader + flat n-gram tables + host trie/smear + kernels.
    Checks the reference's own assertions and bit-equality with what the
    compiled reference (ARPA stand-in for KenLM) produced."""
    import gzip
    import json
    import os
    from golden.make_golden import parse_lexicon_dump
    from text_amd import _capi
    d = os.path.join(helpers.GOLDEN_DIR, "decodertest")
    rd = lambda n: gzip.open(os.path.join(d, n + ".gz"), "rb").read()
    lex = parse_lexicon_dump(rd("lexicon_dump.txt").decode())
    letters = rd("letters.lst").decode().split() + ["<1>"]
    TN = np.frombuffer(rd("TN.bin"), dtype=np.int32)
    T, N = int(TN[0]), int(TN[1])
    em = np.frombuffer(rd("emission.bin"), dtype=np.float32).copy()
    tr = np.frombuffer(rd("transition.bin"), dtype=np.float32).copy()
    arpa = tmp_path / "lm.arpa"
    arpa.write_bytes(rd("lm.arpa"))
    exp = json.load(open(os.path.join(d, "expected.json")))

    lm = _capi.ArpaLM(str(arpa), lex["words"])
    widx = {w: i for i, w in enumerate(lex["words"])}
    per, total = lm.score_sequence([widx[w] for w in "the cat sat on the mat".split()], True)
    assert np.allclose(per, [-1.05971, -4.19448, -3.33383, -2.76726, -1.16237, -4.64589], atol=1e-5)
    assert abs(total - (-19.5123)) < 1e-4
    assert [float(x) for x in per] == exp["lm_scores"]

    ht = _capi.HostTrie(lex["ntok"], lex["sil"])
    cache = {}
    for wi, w, sp in lex["entries"]:
        if wi not in cache:
            cache[wi] = lm.score_sequence([wi], False)[0][0]
        ht.insert(sp, wi, cache[wi])
    ht.smear(1)
    got_trie = [ht.search([letters.index(ch) for ch in w])["max_score"]
                for w in "the cat sat on the mat".split()]
    assert np.allclose(got_trie, [-1.05971, -2.87742, -2.64553, -3.05081, -1.05971, -3.08968], atol=1e-5)
    assert got_trie == exp["trie_scores"]

    ctx = gpu_session.ctx
    opt = _capi.make_options(2500, 25000, 100.0, 2.0, 2.0, -float("inf"), -1.0, False, "asg")
    dec = _capi.BatchDecoder(ctx, _capi.LEXICON, opt, lm, lex["sil"], -1, unk=lex["unk"], trie=ht.upload(ctx),
                             transitions=tr, is_lm_token=False)
    dec.decode_batch(em, [T], N)
    hyps = dec.results(0)
    assert len(hyps) == 16  # DecoderTest.cpp:184
    for h, t in zip(hyps, [-284.0998, -284.108, -284.119, -284.127, -284.296]):
        assert abs(h.score - t) < 1e-3  # DecoderTest.cpp:190-194
    ok, why = helpers.check_against_golden(hyps, exp["nbest"])
    assert ok, why
    dec.close()


LEX_CUT = [c for c in cases.CASES if c["kind"] == "lexicon" and not c["log_add"] and
           (c["lm"] == "zero" or c["lm"][0] != "lastword")]  # (a host LM runs without the score cut: fltx_api.cpp prepare())


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["tight", "off", "recompute", "recompute_tight"])
@pytest.mark.parametrize("c", LEX_CUT, ids=lambda c: c["name"])
def test_lexicon_score_cut(gpu_session, golden, c, mode):
    """The lexicon decoder scores every candidate first and materialises only the
    best 3K + 64 (runFrame, cut-off generation).  Same n-best with the cut
    forced down to K + 1 (exact, or flagged and redone) and with the cut off."""
    inp = helpers.case_inputs(c)
    d = gpu_session.decoder(c, inp)
    if mode == "tight":
        d.set("cut_m", c["K"] + 1)
    elif mode == "recompute":  # no slim records: count per score bin, then generate again
        d.set("cut_m", 3 * c["K"] + 64)
        d.set("slim", 0)
    elif mode == "recompute_tight":
        d.set("cut_m", c["K"] + 1)
        d.set("slim", 0)
    else:
        d.set("cut", 0)
    d.decode_batch(inp["e"], [c["T"]], c["N"])
    ok, why = helpers.check_against_golden(d.results(0), golden[c["name"]], 0.0)
    d.close()
    assert ok, why


@pytest.mark.gpu
@pytest.mark.parametrize("hot,slim,tight", [(2, 1, False), (1, 1, False), (2, 0, False), (1, 0, False), (2, 1, True), (2, 0, True)])
@pytest.mark.parametrize("c", LEX_CUT, ids=lambda c: c["name"])
def test_lexicon_hbm_workspace_with_cut(gpu_session, golden, c, hot, slim, tight):
    """Lexicon beams beyond the LDS (forced with a tiny LDS budget): beam in HBM,
    recompute form of the cut-off generation, records in LDS (level 2) or HBM (1)."""
    inp = helpers.case_inputs(c)
    d = gpu_session.decoder(c, inp)
    d.set("lds_budget", 2048)
    d.set("hot_level", hot)
    d.set("slim", slim)
    if tight:
        d.set("cut_m", c["K"] + 1)
    d.decode_batch(inp["e"], [c["T"]], c["N"])
    if not tight:
        assert d.get("lds") == 0 and d.get("recompute") == 1 - slim and d.get("hot_level") == hot
        assert d.get("cut") > 0 and (d.get("cap2") > 0) == bool(slim)
    ok, why = helpers.check_against_golden(d.results(0), golden[c["name"]], 0.0)
    d.close()
    assert ok, why


@pytest.mark.parametrize("name,engine,sets", [
    ("lf_ctc_t60_k10", 4, {}), ("lf_uni_n64_k64", 4, {}), ("lf_ctc_n29_k64", 4, {}), ("lf_ctc_t60_k10_kt5", 4, {}),
    ("lf_asg_t40_n29_kt7", 4, {}), ("lf_ctc_sil", 4, {}), ("C2_ctc_u0", 4, {}), ("C2_uniform_u0", 4, {}),
    ("lf_ctc_t60_k10", 3, {"slane": 0}), ("lf_uni_n64_k64", 3, {"slane": 0}), ("lf_ctc_n29_k64", 3, {"slane": 0}),
    ("lf_ctc_t60_k10_kt5", 3, {"slane": 0}), ("C2_ctc_u0", 3, {"slane": 0}),
    ("lf_ctc_n29_k65", 4, {}), ("lf_ctc_n29_k65", 2, {"lane_groups": -1}), ("lf_ctc_t60_k10_logadd", 4, {}),
    ("lf_ctc_t60_k10_logadd", 3, {"slane": 0}), ("C2_ctc_u0_logadd", 4, {}), ("lf_ctc_t300_k100", 4, {}),
    ("lf_ctc_t300_k100", 2, {"lane_groups": -1}), ("lf_ctc_t300_k100", 2, {"slane": 0}),
    # every compiled geometry of fltx_mlane.h (rows of kMlaneGeo: 2, 2, 2, 4, 4, 4, 8 lane groups)
    ("lf_ctc_t300_k100", 4, {"mlane_geo": 0}), ("lf_ctc_t300_k100", 4, {"mlane_geo": 1}),
    ("lf_ctc_t300_k100", 4, {"mlane_geo": 2}), ("lf_ctc_t300_k100", 4, {"mlane_geo": 3}),
    ("lf_ctc_t300_k100", 4, {"mlane_geo": 4}), ("lf_ctc_t300_k100", 4, {"mlane_geo": 5}),
    ("lf_ctc_t300_k100", 4, {"mlane_geo": 6}), ("C2_ctc_u0", 4, {"lane_groups": 2}),
    ("C2_ctc_u0_logadd", 4, {"lane_groups": 4}), ("C2_ctc_u0_kt10", 4, {"lane_groups": 8}),
    ("lx_spell_t40_k8", 5, {}), ("lx_spell_t60_k12_full", 5, {}), ("lx_uni_t40_k10", 5, {}), ("lx_t0", 5, {}),
    ("C3_spell_u0", 5, {}), ("C3_spell_u255", 5, {}), ("C3_uniform_u0", 5, {}), ("C3_spell_u0", 5, {"slane_threads": 640}),
    ("lx_spell_t60_k12_full", 5, {"slane_threads": 576}), ("lx_spell_t40_k8", 5, {"slane_threads": 640}),
    ("lx_spell_t40_k8", 6, {"xlane": 0}), ("C3_spell_u0", 0, {"xlane": 0, "ylane": 0}), ("C3_uniform_u0", 0, {"cut": 0}),
    ("lx_spell_unk", 0, {}), ("lx_asg_t40", 6, {}), ("lx_asg_t40", 0, {"ylane_asg": 0}), ("lx_asg_t40", 6, {"ylane_groups": 4}),
    ("lx_asg_t40", 6, {"yshare": 1}), ("lx_spell_t60_k12_logadd", 5, {}), ("lx_spell_t60_k12_logadd", 5, {"yshare": 1}), ("ng_word_unk_t40", 0, {}),
    ("ng_word_logadd_t40", 6, {}), ("ng_word_logadd_t40", 6, {"yshare": 1}), ("ng_tok_lexicon_t40", 0, {}),
    # a token-level n-gram LM on the lexicon-free decoder: fltx_slane.h's token-LM variant (round 6), every geometry
    ("ng_tok_lexfree_t40", 4, {}), ("ng_tok_lexfree_kt8", 4, {}), ("ng_tok_lexfree_t40", 1, {"tlane": 0}),
    ("ng_tok_lexfree_kt8", 1, {"tlane": 0}), ("ng_tok_lexfree_t40", 4, {"slane_threads": 512}),
    ("ng_tok_lexfree_t40", 4, {"slane_threads": 448}), ("ng_tok_lexfree_t40", 4, {"slane_threads": 384}),
    ("ng_tok_lexfree_t40", 4, {"slane_threads": 320}), ("ng_tok_lexfree_t40", 4, {"slane_threads": 640}),
    ("ng_tok_lexfree_kt8", 4, {"slane_threads": 512}), ("ng_tok_lexfree_kt8", 4, {"slane_threads": 320}),
    # ... at beams beyond 64: fltx_mlane.h's token-LM variant (2 / 4 / 8 lane groups); forced onto more groups; switched off
    ("ng_tok_lexfree_k100", 4, {}), ("ng_tok_lexfree_k200_kt8", 4, {}), ("ng_tok_lexfree_asg_k300", 4, {}),
    ("ng_tok_lexfree_k100", 4, {"lane_groups": 4}), ("ng_tok_lexfree_k100", 4, {"lane_groups": 8}),
    ("ng_tok_lexfree_k100", 1, {"tlane": 0}), ("ng_tok_lexfree_k200_kt8", 1, {"tok_dense": 0}),
    ("lx_scores_t50", 6, {}), ("ng_word_t40_k10", 6, {}), ("ng_word_t60_k16_4g", 6, {}), ("C4_spell_u0", 6, {}),
    ("C4_spell_u255", 6, {}), ("C4z_spell_u0", 6, {}), ("lx_spell_t40_k8", 6, {"ylane": 2}),
    ("lx_uni_t40_k10", 6, {"ylane": 2}), ("C3_spell_u0", 6, {"ylane": 2}), ("C3_uniform_u0", 6, {"ylane": 2}),
    ("ng_word_t40_k10", 0, {"ylane": 0}), ("C4_spell_u0", 0, {"ylane": 0}), ("lx_scores_t50", 0, {"slim": 1}),
    # four lane groups of fltx_ylane.h (beams 129 .. 256): forced here on goldens of smaller beams
    ("C4_spell_u0", 6, {"ylane_groups": 4}), ("C4z_spell_u0", 6, {"ylane_groups": 4}),
    ("ng_word_t60_k16_4g", 6, {"ylane_groups": 4}), ("C3_spell_u0", 6, {"ylane": 2, "ylane_groups": 4})])
def test_engine_selection(gpu_session, golden, name, engine, sets):
    """Which engine serves which configuration: the lane = LM state decode (4,
    fltx_slane.h) for offline lexicon-free + ZeroLM max-merge with beam <= 64 and
    <= 64 tokens (also with a token beam, ASG, silScore, logAdd) and, with 2 / 4 / 8 lane groups
    (fltx_mlane.h), for beams up to 512; the lane-per-slot step
    (3) when the former is switched off; the lean step (2) for
    bigger beams and when the lane groups are switched off; the lane = (LM state, trie node) decode (5, fltx_xlane.h) for
    the offline lexicon decoder + ZeroLM over a lexicon without scores (CTC,
    max-merge, no <unk>, beam <= 64); the same with the LM terms (6, fltx_ylane.h)
    for an n-gram word LM and / or a smeared trie, CTC or ASG, and beams up to 256; the
    generic engine (0) for everything else
    and whenever one of its own tunables is touched.  Same n-best either way."""
    c = cases.BY_NAME[name]
    inp = helpers.case_inputs(c)
    d = gpu_session.decoder(c, inp)
    for k, v in sets.items():
        d.set(k, v)
    d.decode_batch(inp["e"], [c["T"]], c["N"])
    got = d.get("engine")
    tol = 1e-5 if c["log_add"] else 0.0
    ok, why = helpers.check_against_golden(d.results(0), golden[c["name"]], tol)
    tl, why_not, redone = d.get("tlane"), d.get("why_not_lane"), d.get("redone")
    d.close()
    assert got == engine
    assert ok, why
    if name.startswith("ng_tok_lexfree"):
        assert tl == (1 if engine == 4 else 0) and redone == 0 and (why_not == 0) == (engine == 4)


LANE_OK = [c for c in cases.CASES if c["kind"] == "lexfree" and c["lm"] == "zero" and c["K"] <= 64 and c["N"] <= 64]


@pytest.mark.parametrize("c", LANE_OK, ids=lambda c: c["name"])
def test_lane_per_slot_step_matches_golden(gpu_session, golden, c):
    """The lane = LM state decode (fltx_slane.h) takes the offline lexicon-free +
    ZeroLM configurations by default; the lane-per-slot step (fltx_lane.h) still
    serves streams and logAdd, and must give the same n-best everywhere."""
    inp = helpers.case_inputs(c)
    d = gpu_session.decoder(c, inp)
    d.set("slane", 0)
    d.decode_batch(inp["e"], [c["T"]], c["N"])
    eng = d.get("engine")
    tol = 1e-5 if c["log_add"] else 0.0
    ok, why = helpers.check_against_golden(d.results(0), golden[c["name"]], tol)
    d.close()
    assert eng == 3
    assert ok, why


HBM_WS = ["lf_ctc_t60_k10", "lf_asg_t30_n8", "lx_scores_t50", "C1_ctc_u0"] + \
    [c["name"] for c in cases.CASES if c["name"].startswith("ng_")][:3]


@pytest.mark.parametrize("name", HBM_WS)
def test_workspace_in_hbm(gpu_session, golden, name):
    """Big beams carve the per-frame workspace from HBM instead of LDS (agent-scope
    barriers, L2 atomics).  Forced here on small cases: same n-best."""
    c = cases.BY_NAME[name]
    inp = helpers.case_inputs(c)
    d = gpu_session.decoder(c, inp)
    d.set("force_global_ws", 1)
    d.decode_batch(inp["e"], [c["T"]], c["N"])
    assert d.get("lds") == 0
    tol = 1e-5 if c["log_add"] else 0.0
    ok, why = helpers.check_against_golden(d.results(0), golden[c["name"]], tol)
    d.close()
    assert ok, why


@pytest.mark.parametrize("N,K,Kt,dist", [(200, 12, 20, "uniform"), (1000, 20, 50, "uniform"), (96, 70, 96, "ctc")])
def test_large_token_sets_match_oracle(gpu_session, oracle_lib, N, K, Kt, dist):
    """Word-piece sized token sets (N > 64: no lane / lean step, token short-list
    by the general path) against the oracle on the fly."""
    from text_amd import synth
    c = cases.case("big_n", dist=dist, T=30, N=N, K=K, Kt=Kt, u=11)
    e = synth.emissions(dist, c["u"], c["T"], N)
    d = gpu_session.decoder(c, dict(tr=None))
    d.decode_batch(e, [c["T"]], N)
    want = helpers.run_checker(oracle_lib, c, dict(e=e, tr=None, lex=None))
    ok, why = helpers.hyps_equal(want, d.results(0))
    d.close()
    assert ok, why


@pytest.mark.parametrize("name", ["lf_ctc_t60_k10", "lx_scores_t50", "ng_word_t40_k10", "lf_ctc_n29_k65"])
def test_decoder_reuse_across_batches(gpu_session, oracle_lib, name):
    """One decoder object, several batches of different content and shape: the
    per-utterance tables that persist across launches (LM-state ids, child
    tables, history) must not leak from one batch into the next."""
    from text_amd import synth
    c = cases.BY_NAME[name]
    inp = helpers.case_inputs(c)
    d = gpu_session.decoder(c, inp)
    lex = inp["lex"] if c["dist"] == "lexspell" else None
    for rnd, (B, T) in enumerate([(3, c["T"]), (5, c["T"] // 2 + 1), (2, c["T"] + 7), (3, c["T"])]):
        embs = [synth.emissions(c["dist"], 500 + 10 * rnd + b, T, c["N"], lexicon=lex) for b in range(B)]
        flat = np.concatenate([e.reshape(-1) for e in embs])
        d.decode_batch(flat, [T] * B, c["N"])
        for b in range(B):
            cb = dict(c, T=T)
            want = helpers.run_checker(oracle_lib, cb, dict(inp, e=embs[b]))
            ok, why = helpers.hyps_equal(want, d.results(b))
            assert ok, "round %d utterance %d: %s" % (rnd, b, why)
    d.close()


@pytest.mark.parametrize("c", cases.fuzz_cases(), ids=lambda c: c["name"])
def test_random_configurations_match_oracle(gpu_session, oracle_lib, c):
    """Differential test over random option combinations (beam, token beam,
    threshold, scores, LM kind and weight, logAdd) against the oracle."""
    inp = helpers.case_inputs(c)
    want = helpers.run_checker(oracle_lib, c, inp)
    if len({h.score for h in want}) != len(want):
        pytest.skip("equal scores in the n-best: the reference's own result is order dependent")
    got = gpu_session.run(c, inp)
    ok, why = helpers.hyps_equal(want, got, 1e-5 if c["log_add"] else 0.0)
    assert ok, "%s: %s" % ({k: c[k] for k in ("kind", "N", "K", "Kt", "thr", "lm", "log_add", "T")}, why)


@pytest.mark.parametrize("name", ["C1_ctc_u0", "lx_scores_t50"])
def test_batched_result_fetch_equals_per_utterance_fetch(gpu_session, name):
    """fltx_result_fetch_batch (one transfer per array through pinned staging)
    returns exactly what the per-utterance fltx_result_fetch returns."""
    from text_amd import synth
    c = cases.BY_NAME[name]
    inp = helpers.case_inputs(c)
    lex = inp["lex"] if c["dist"] == "lexspell" else None
    Ts = [c["T"], 0, 7, c["T"] // 2]
    embs = [synth.emissions(c["dist"], 900 + b, T, c["N"], lexicon=lex) for b, T in enumerate(Ts)]
    flat = np.concatenate([e.reshape(-1) for e in embs])
    d = gpu_session.decoder(c, inp)
    d.decode_batch(flat, Ts, c["N"])
    allh = d.results_batch()
    assert len(allh) == len(Ts)
    for b in range(len(Ts)):
        ok, why = helpers.hyps_equal(d.results(b), allh[b])
        assert ok, "utterance %d: %s" % (b, why)
    # ... and so does the compacted form (rows that exist only, tokens as bytes, packed on the device)
    r = d.results_arrays_compact()
    assert r["tokens_u8"].dtype == np.uint8 and int(r["offsets"][-1]) == int((r["n_hyp"] * r["length"]).sum())
    for b in range(len(Ts)):
        assert int(r["n_hyp"][b]) == len(allh[b])
        for i, h in enumerate(allh[b]):
            assert np.array_equal(d.tokens_of(r, b, i), h.tokens) and np.array_equal(d.words_of(r, b, i), h.words)
            assert tuple(r["scores"][b, i]) == (h.score, h.am, h.lm)
    d.close()


def test_partial_rerun_of_flagged_utterances(gpu_session, oracle_lib):
    """With the score cut forced tight some utterances of a batch flag a frame
    (fewer than K groups survived the cut) and only those are decoded again on
    the general path; every utterance must still equal the oracle."""
    from text_amd import synth
    c = cases.BY_NAME["lx_scores_t50"]
    inp = helpers.case_inputs(c)
    lex = inp["lex"]
    B = 8
    embs = [synth.emissions(c["dist"], 800 + b, c["T"], c["N"], lexicon=lex) for b in range(B)]
    flat = np.concatenate([e.reshape(-1) for e in embs])
    d = gpu_session.decoder(c, inp)
    d.set("cut_m", c["K"] + 1)
    for _ in range(2):  # the second batch starts from the fast path again (or the sticky fallback): same results
        d.decode_batch(flat, [c["T"]] * B, c["N"])
        for b in range(B):
            want = helpers.run_checker(oracle_lib, c, dict(inp, e=embs[b]))
            ok, why = helpers.hyps_equal(want, d.results(b))
            assert ok, "utterance %d: %s" % (b, why)
    d.close()


@pytest.mark.parametrize("K,T,dist", [(700, 40, "ctc"), (1500, 25, "ctc"), (600, 30, "uniform")])
def test_big_lexicon_free_beams(gpu_session, oracle_lib, K, T, dist):
    """Beams beyond what fits a CU's LDS: streaming lean step over the HBM workspace."""
    from text_amd import synth
    c = cases.case("bigbeam", dist=dist, T=T, N=29, K=K, u=23)
    e = synth.emissions(dist, c["u"], T, c["N"])
    d = gpu_session.decoder(c, dict(tr=None))
    d.decode_batch(e, [T], c["N"])
    assert d.get("lean") == 255 and d.get("lds") == 0
    want = helpers.run_checker(oracle_lib, c, dict(e=e, tr=None, lex=None))
    got = d.results(0)
    d.close()
    if len({h.score for h in want}) != len(want):
        pytest.skip("equal scores in the n-best")
    ok, why = helpers.hyps_equal(want, got)
    assert ok, why


@pytest.mark.gpu
@pytest.mark.parametrize("K,hot", [(400, 2), (500, 2), (700, 1)])
def test_big_lexicon_beams_with_ngram_lm(gpu_session, oracle_lib, K, hot):
    """Beams of the lexicon decoder beyond the LDS (C4 shape: 90k-word trie, 4-gram word LM): the generic step over the
    HBM workspace, sixteen waves, no L1 invalidate after its barriers (DecodeParams::wsNoInv) -- level 2 (candidate
    records and merge hash in LDS) and level 1 (in HBM), two utterances each against the oracle."""
    from text_amd import synth
    T = 40
    c = cases.case("bigbeam_lx", kind="lexicon", dist="lexspell", T=T, N=29, K=K, Kt=29, lexicon=cases.FULL_LEX,
                   lm=("ngram", 4, 8), lm_weight=2.0, word_score=2.0, sil_score=-1.0, u=31)
    inp = helpers.case_inputs(c)
    B = 2
    e = synth.batch("lexspell", B, T, 29, lexicon=inp["lex"], u0=c["u"])
    d = gpu_session.decoder(c, inp)
    d.decode_batch(e, [T] * B, 29)
    assert d.get("engine") == 0 and d.get("lds") == 0 and d.get("hot_level") == hot and d.get("threads") == 1024
    for b in range(B):
        want = helpers.run_checker(oracle_lib, c, dict(inp, e=np.ascontiguousarray(e[b])))
        if len({h.score for h in want}) != len(want):
            continue  # equal scores in the n-best
        ok, why = helpers.hyps_equal(want, d.results(b))
        assert ok, "beam %d utterance %d: %s" % (K, b, why)
    d.close()


@pytest.mark.gpu
@pytest.mark.parametrize("K,T,dist,threads", [(100, 60, "ctc", 0), (130, 50, "ctc", 0), (200, 40, "uniform", 0),
                                              (256, 40, "ctc", 0), (300, 40, "ctc", 0), (100, 40, "ctc", 1024),
                                              (70, 40, "uniform", 256), (128, 40, "ctc", 128)])
def test_lexicon_free_beams_above_the_lane_engines(gpu_session, oracle_lib, K, T, dist, threads):
    """Beams 65+ of the lexicon-free decoder (lean step): a slot's scan for the hypotheses of its LM state and of its
    parent state is split between W / stride threads -- every split (4, 2, none; 8 at 1024 threads) against the oracle."""
    from text_amd import synth
    c = cases.case("leanbeam", dist=dist, T=T, N=29, K=K, u=41)
    e = synth.emissions(dist, c["u"], T, c["N"])
    d = gpu_session.decoder(c, dict(tr=None), threads or None)
    d.set("lane_groups", -1)  # (beams up to 512 are the lane = LM state engine's otherwise: fltx_mlane.h)
    d.decode_batch(e, [T], c["N"])
    assert d.get("engine") == 2
    want = helpers.run_checker(oracle_lib, c, dict(e=e, tr=None, lex=None))
    got = d.results(0)
    d.close()
    if len({h.score for h in want}) != len(want):
        pytest.skip("equal scores in the n-best")
    ok, why = helpers.hyps_equal(want, got)
    assert ok, why


def _tied_emissions(T, N, u):
    """`ctc` rows in which the most used token has a twin column: hypotheses that differ only by swapping the two tie
    to the last bit, so the n-best is full of equal scores (the reference's own order is then undefined, SURVEY.md
    section 0)."""
    from text_amd import synth
    import numpy as np
    e = synth.emissions("ctc", u, T, N).copy()
    top = np.bincount(np.argmax(e[:, :N - 1], axis=1), minlength=N - 1)  # the non-blank token the best path uses most
    a = int(np.argmax(top))
    e[:, (a + 1) % (N - 1)] = e[:, a]  # ... gets a twin: every occurrence can be swapped at no cost
    return e


def _nbest_bits(d, b=0):
    return [(float(h.score).hex(), float(h.am).hex(), h.tokens.tobytes()) for h in d.results(b)]


@pytest.mark.gpu
@pytest.mark.parametrize("K,sets_a,sets_b", [
    (20, {"slane_threads": 576}, {"slane_threads": 512}),   # fltx_slane.h: 4 and 5 list positions per wave
    (20, {"slane_threads": 576}, {"slane_threads": 320}),   # ... and 10
    (100, {"mlane_geo": 0}, {"mlane_geo": 0}),              # fltx_mlane.h: the same geometry twice
    (100, {"lane_groups": -1}, {"lane_groups": -1}),        # lean step
])
def test_tied_input_is_decoded_deterministically(gpu_session, K, sets_a, sets_b):
    """The device's tie policy (a tie goes to the earlier generated candidate -- list position,
    then lane, then the lanes' own groups) is a function of the input: the same tied utterance decoded twice, and on
    two geometries of the lane = LM state engine (whose generation order does not depend on how the list positions
    are dealt to the waves), gives the same n-best bit for bit.  Nothing is claimed about the reference's order."""
    T, N = 120, 29
    e = _tied_emissions(T, N, 77)
    c = cases.case("tied", T=T, N=N, K=K)
    runs = []
    for sets in (sets_a, sets_b, sets_a):
        d = gpu_session.decoder(c, dict(tr=None))
        for k, v in sets.items():
            d.set(k, v)
        d.decode_batch(e, [T], N)
        nb = _nbest_bits(d)
        scores = [x[0] for x in nb]
        runs.append((nb, d.get("engine"), d.get("threads")))
        d.close()
    assert len(set(scores)) < len(scores), "the input was meant to produce equal scores"
    assert runs[0][0] == runs[2][0], "two runs of one geometry differ"
    assert runs[0][0] == runs[1][0], "geometries %r and %r differ" % (runs[0][1:], runs[1][1:])
