"""CPU (not gpu): the SAME kernel source as the HIP path, compiled for host
threads (tests/emu), against the reference golden vectors.  This checks kernel
logic only; the parity claim for the product is made by tests/test_gpu_parity.py
on a real MI355X."""
import os

import pytest

import cases
import helpers
from oracle import orclib

SMALL = [c for c in cases.CASES if c["size"] == "small" and c["T"] <= 40]


@pytest.mark.parametrize("c", SMALL, ids=lambda c: c["name"])
def test_emulated_kernels_match_golden(emu_session, golden, c):
    hyps = emu_session.run(c, threads=64)
    tol = 1e-9 if c["log_add"] else 0.0
    ok, why = helpers.check_against_golden(hyps, golden[c["name"]], tol)
    assert ok, why


def test_emulated_two_waves(emu_session, golden):
    c = cases.BY_NAME["lf_ctc_t20_k4"]
    hyps = emu_session.run(c, threads=128)
    ok, why = helpers.check_against_golden(hyps, golden[c["name"]])
    assert ok, why


LANE = ["lf_ctc_t20_k4", "lf_ctc_t60_k10", "lf_uni_t40_k10", "lf_ctc_t60_k10_logadd", "lf_ctc_t1", "lf_ctc_k1",
        "lf_ctc_thr3", "lf_ctc_sil", "lf_asg_t30_n8", "lf_ctc_n4"]


@pytest.mark.parametrize("name", LANE)
def test_emulated_lane_per_slot_kernel(emu_session, golden, name):
    """fltx_lane.h (beam <= 64, all tokens considered): 4 waves cover N = 29 with 8
    tokens per wave; small N runs it with one wave too."""
    c = cases.BY_NAME[name]
    threads = 256 if c["N"] > 8 else 64
    hyps = emu_session.run(c, threads=threads)
    assert emu_session.last_engine == 3
    tol = 1e-9 if c["log_add"] else 0.0
    ok, why = helpers.check_against_golden(hyps, golden[c["name"]], tol)
    assert ok, why


XLANE = [("lx_t0", 0), ("lx_spell_t40_k8", 0), ("lx_spell_t40_k8", 640), ("lx_spell_t60_k12_full", 0),
         ("lx_uni_t40_k10", 0), ("lx_uni_t40_k10", 640)]


@pytest.mark.parametrize("yshare", [0, 1], ids=["memo-in-lds", "memo-in-hbm"])
@pytest.mark.parametrize("name,threads", XLANE)
def test_emulated_lexicon_lane_engine(emu_session, golden, name, threads, yshare):
    """fltx_xlane.h (LexiconDecoder + ZeroLM, beam <= 64): lane = (LM state, trie node); yshare = 1: the geometry
    that shares a CU (LM-state memo in HBM)."""
    c = cases.BY_NAME[name]
    inp = helpers.case_inputs(c)
    d = emu_session.decoder(c, inp)
    d.set("yshare", yshare)
    if threads:
        d.set("slane_threads", threads)
    d.decode_batch(inp["e"], [c["T"]], c["N"])
    assert d.get("engine") == 5 and d.get("redone") == 0 and (not threads or d.get("threads") == threads)
    assert d.get("yshare") == yshare
    ok, why = helpers.check_against_golden(d.results(0), golden[c["name"]])
    d.close()
    assert ok, why


YLANE = [("lx_scores_t50", {}), ("ng_word_t40_k10", {}), ("ng_word_t60_k16_4g", {}), ("lx_spell_t40_k8", {"ylane": 2}),
         ("lx_uni_t40_k10", {"ylane": 2}), ("lx_t0", {"ylane": 2}),
         # the geometry that shares a CU: LM-state memo in HBM (fltx_ylane.h, HM = 1)
         ("ng_word_t40_k10", {"yshare": 1}), ("lx_scores_t50", {"yshare": 1})]


@pytest.mark.parametrize("name,sets", YLANE)
def test_emulated_lexicon_lane_engine_with_lm_terms(emu_session, golden, name, sets):
    """fltx_ylane.h (n-gram word LM / smeared trie / beams up to 128): stable lane slots, compacted
    (lane, token) pairs."""
    c = cases.BY_NAME[name]
    inp = helpers.case_inputs(c)
    d = emu_session.decoder(c, inp)
    for k, v in sets.items():
        d.set(k, v)
    d.decode_batch(inp["e"], [c["T"]], c["N"])
    assert d.get("engine") == 6 and d.get("redone") == 0
    ok, why = helpers.check_against_golden(d.results(0), golden[c["name"]])
    d.close()
    assert ok, why


@pytest.mark.parametrize("yshare", [0, 1])
def test_emulated_two_lane_groups(emu_session, oracle_lib, yshare):
    """beam 90 over an n-gram LM: two groups of 64 lanes, four rounds of pairs per token-wave thread;
    yshare = 1: 512 threads (four token waves with twice the pairs each), LM-state memo in HBM."""
    c = [x for x in cases.fuzz_cases(120) if x["name"] == "fuzz113"][0]
    inp = helpers.case_inputs(c)
    want = helpers.run_checker(oracle_lib, c, inp)
    d = emu_session.decoder(c, inp)
    d.set("yshare", yshare)
    d.decode_batch(inp["e"], [c["T"]], c["N"])
    assert d.get("engine") == 6 and d.get("ylane") == 2 and d.get("redone") == 0
    assert d.get("yshare") == yshare and d.get("threads") == (512 if yshare else 768)
    ok, why = helpers.hyps_equal(want, d.results(0))
    d.close()
    assert ok, why


LEX_SMALL = [c for c in cases.CASES if c["kind"] == "lexicon" and c["size"] == "small" and c["T"] <= 40
             and not c["log_add"]]


@pytest.mark.parametrize("slim", [1, 0])
@pytest.mark.parametrize("c", LEX_SMALL, ids=lambda c: c["name"])
def test_emulated_score_cut_is_exact_or_retried(emu_session, golden, c, slim):
    """Lexicon decoder, cut-off generation (runFrame): with the cut forced down to
    K + 1 candidates either the kept ones still form K groups (exact by
    construction) or the kernel flags the frame and the batch is redone without
    the cut -- the n-best must equal the reference's in both cases."""
    inp = helpers.case_inputs(c)
    d = emu_session.decoder(c, inp, 64)
    d.set("cut_m", c["K"] + 1)
    d.set("slim", slim)  # 0: the recompute form (count per bin, generate again) instead of slim records
    d.decode_batch(inp["e"], [c["T"]], c["N"])
    ok, why = helpers.check_against_golden(d.results(0), golden[c["name"]])
    d.close()
    assert ok, why


@pytest.mark.parametrize("hot,slim,cut_m", [(2, 1, 0), (1, 1, 0), (2, 0, 0), (1, 0, 0), (2, 1, -1), (2, 0, -1)])
@pytest.mark.parametrize("c", LEX_SMALL[:4], ids=lambda c: c["name"])
def test_emulated_hbm_workspace_with_cut(emu_session, golden, c, hot, slim, cut_m):
    """Lexicon beams that do not fit the LDS (forced here with a tiny LDS budget):
    beam in an HBM workspace, recompute form of the cut-off generation, candidate
    records in LDS (level 2) or in HBM (level 1); tight cut -> flagged and redone."""
    inp = helpers.case_inputs(c)
    d = emu_session.decoder(c, inp, 64)
    d.set("lds_budget", 2048)
    d.set("hot_level", hot)
    d.set("slim", slim)
    if cut_m:
        d.set("cut_m", c["K"] + 1)
    d.decode_batch(inp["e"], [c["T"]], c["N"])
    if not cut_m:
        assert d.get("lds") == 0 and d.get("recompute") == 1 - slim and d.get("hot_level") == hot
        assert d.get("cut") > 0 and (d.get("cap2") > 0) == bool(slim)
    ok, why = helpers.check_against_golden(d.results(0), golden[c["name"]])
    d.close()
    assert ok, why


@pytest.mark.parametrize("c", [c for c in cases.fuzz_cases(18) if c["T"] <= 25 and c["K"] <= 33],
                         ids=lambda c: c["name"])
def test_emulated_random_configurations_match_oracle(emu_session, oracle_lib, c):
    """The randomized differential test of tests/test_gpu_parity.py, on the
    emulated kernels (small shapes only)."""
    inp = helpers.case_inputs(c)
    want = helpers.run_checker(oracle_lib, c, inp)
    if len({h.score for h in want}) != len(want):
        pytest.skip("equal scores in the n-best: the reference's own result is order dependent")
    got = emu_session.run(c, inp, threads=128 if c["kind"] == "lexfree" and c["N"] > 16 else 64)
    ok, why = helpers.hyps_equal(want, got, 1e-9 if c["log_add"] else 0.0)
    assert ok, "%s: %s" % ({k: c[k] for k in ("kind", "N", "K", "Kt", "thr", "lm", "log_add", "T")}, why)


@pytest.mark.parametrize("sets", [{}, {"lds_budget": 2048}], ids=["default", "hbm_workspace"])
def test_emulated_fuzz(emu_session, oracle_lib, sets):
    """400 of tools/fuzz_big.py's random configurations at the library's own choice of engine and threads (the
    parametrised test above pins small shapes to one or two waves), once more with the CU's LDS budget cut to 2 KB
    so that every lexicon case takes the HBM-workspace paths.  Cases whose reference n-best holds equal scores are
    left out (order dependent in the reference itself)."""
    bad = []
    ran = 0
    for c in cases.fuzz_cases(440)[40:]:
        inp = helpers.case_inputs(c)
        want = helpers.run_checker(oracle_lib, c, inp)
        if len({h.score for h in want}) != len(want):
            continue
        got = emu_session.run(c, inp, sets=sets)
        ok, why = helpers.hyps_equal(want, got, 1e-9 if c["log_add"] else 0.0)
        ran += 1
        if not ok:
            bad.append((c["name"], why))
    assert ran >= 300 and not bad, (ran, bad[:3])


@pytest.mark.parametrize("name", ["lf_ctc_n29_k65", "lf_ctc_n29_k64", "lf_ctc_t60_k10_logadd"])
def test_emulated_streaming_lean_step(emu_session, golden, name):
    """Big beams: more groups per thread than the register-resident lean step
    holds (one wave, beam 64 / 65 x 29 tokens => 30 groups per thread): groups
    are evaluated twice instead (fltx_lean.h, GMAX == 255)."""
    c = cases.BY_NAME[name]
    inp = helpers.case_inputs(c)
    d = emu_session.decoder(c, inp, 64)
    d.set("lane", 0)
    if c["K"] < 30:
        d.set("threads", 64)
    d.decode_batch(inp["e"], [c["T"]], c["N"])
    lean = d.get("lean")
    tol = 1e-9 if c["log_add"] else 0.0
    ok, why = helpers.check_against_golden(d.results(0), golden[c["name"]], tol)
    d.close()
    assert lean == (255 if c["K"] >= 30 else 6)
    assert ok, why


def test_emulated_streaming_lean_step_over_hbm_workspace(emu_session, oracle_lib):
    """A lexicon-free beam too large for a CU's LDS (700 x 29 tokens): the
    streaming lean step runs over the HBM workspace with agent-scope barriers."""
    from text_amd import synth
    c = cases.case("bigbeam", T=12, N=29, K=700, u=21)
    e = synth.emissions("ctc", c["u"], c["T"], c["N"])
    d = emu_session.decoder(c, dict(tr=None), 64)
    d.decode_batch(e, [c["T"]], c["N"])
    assert d.get("lean") == 255 and d.get("lds") == 0
    want = helpers.run_checker(oracle_lib, c, dict(e=e, tr=None, lex=None))
    got = d.results(0)
    d.close()
    if len({h.score for h in want}) == len(want):
        ok, why = helpers.hyps_equal(want, got)
        assert ok, why


def test_trie_that_is_not_a_tree_is_accepted_without_the_breadth_first_layout(emu_session):
    """fltx_trie_create takes a caller's child table: a shared suffix (DAG), a self loop or a back edge must
    neither write past the breadth-first arrays nor loop for ever -- such a trie simply has no lane-engine
    layout and decodes on the generic engine (round-2 advisor finding)."""
    from text_amd import _capi
    import numpy as np
    N = 5
    for extra in ("dag", "self", "back"):
        child = -np.ones((4, N), dtype=np.int32)
        child[0, 1] = 1
        child[0, 2] = 2
        child[1, 0] = 3
        if extra == "dag":
            child[2, 0] = 3  # node 3 reachable twice
        elif extra == "self":
            child[3, 3] = 3
        else:
            child[3, 1] = 1
        label_off = np.array([0, 0, 0, 0, 1], dtype=np.int32)
        t = _capi.Trie(emu_session.ctx, child, np.zeros(4, np.float32), label_off, np.array([0], np.int32))
        opt = _capi.make_options(4, N, 25.0)
        d = _capi.BatchDecoder(emu_session.ctx, _capi.LEXICON, opt, emu_session.zero, 0, N - 1, unk=1, trie=t)
        e = np.zeros((6, N), dtype=np.float32)
        d.decode_batch(e, [6], N)
        assert d.get("engine") == 0  # generic engine: the lane engines need the layout
        d.close()
        t.close()


def test_compact_result_fetch_on_the_emulator(emu_session):
    """fltx_result_fetch_batch_compact against fltx_result_fetch_batch (host logic; the device packing kernel is
    exercised by tests/test_gpu_parity.py)."""
    import numpy as np
    from text_amd import synth
    c = cases.BY_NAME["lx_spell_t40_k8"]
    inp = helpers.case_inputs(c)
    Ts = [c["T"], 0, 7]
    embs = [synth.emissions(c["dist"], 900 + b, T, c["N"], lexicon=inp["lex"]) for b, T in enumerate(Ts)]
    d = emu_session.decoder(c, inp)
    d.decode_batch(np.concatenate([e.reshape(-1) for e in embs]), Ts, c["N"])
    allh = d.results_batch()
    r = d.results_arrays_compact()
    assert int(r["offsets"][-1]) == int((r["n_hyp"] * r["length"]).sum())
    for b in range(len(Ts)):
        assert int(r["n_hyp"][b]) == len(allh[b])
        for i, h in enumerate(allh[b]):
            assert np.array_equal(d.tokens_of(r, b, i), h.tokens) and np.array_equal(d.words_of(r, b, i), h.words)
    d.close()


@pytest.mark.parametrize("K,T,threads", [(100, 12, 0), (200, 10, 0), (70, 12, 256)])
def test_lean_step_split_relation_scan(emu_session, oracle_lib, K, T, threads):
    """Beams 65+ of the lexicon-free decoder: the split of a slot's relation scan between threads (fltx_lean.h)."""
    from text_amd import synth
    c = cases.case("leanbeam_emu", dist="ctc", T=T, N=29, K=K, u=43)
    e = synth.emissions("ctc", c["u"], T, c["N"])
    d = emu_session.decoder(c, dict(tr=None), threads or None)
    d.set("lane_groups", -1)  # (these beams start on fltx_mlane.h since round 4: the lean step is what is tested here)
    d.decode_batch(e, [T], c["N"])
    assert d.get("engine") == 2
    want = helpers.run_checker(oracle_lib, c, dict(e=e, tr=None, lex=None))
    got = d.results(0)
    d.close()
    if len({h.score for h in want}) != len(want):
        pytest.skip("equal scores in the n-best")
    ok, why = helpers.hyps_equal(want, got)
    assert ok, why


def test_emulated_lane_state_engine_with_lane_groups(emu_session, oracle_lib):
    """fltx_mlane.h on the emulator: a thin slice of the GPU suite's grid (tests/test_gpu_batches.py)."""
    import test_gpu_batches
    ran, served, bad = test_gpu_batches._lane_group_grid(emu_session, oracle_lib, 29, lambda i: [2, 17, 9][i % 3], emu=True)
    assert ran >= 20 and served == ran and not bad, (ran, served, bad[:3])


def test_emulated_four_lane_groups(emu_session, oracle_lib):
    """fltx_ylane.h with four lane groups on the emulator: a thin slice of the GPU suite's grid."""
    import test_gpu_batches
    ran, served, bad = test_gpu_batches._four_lane_group_grid(emu_session, oracle_lib, 89, lambda i: [2, 17, 9][i % 3])
    assert ran >= 16 and served == ran and not bad, (ran, served, bad[:3])


def test_emulated_asg_on_the_lexicon_lane_engine(emu_session, oracle_lib, golden):
    """ASG on fltx_ylane.h: the golden case, then a thin slice of the GPU suite's grid."""
    import test_gpu_batches
    c = cases.BY_NAME["lx_asg_t40"]
    inp = helpers.case_inputs(c)
    d = emu_session.decoder(c, inp)
    d.decode_batch(inp["e"], [c["T"]], c["N"])
    assert d.get("engine") == 6 and d.get("redone") == 0
    ok, why = helpers.check_against_golden(d.results(0), golden[c["name"]])
    d.close()
    assert ok, why
    ran, served, bad = test_gpu_batches._asg_lexicon_grid(emu_session, oracle_lib, 47, lambda i: [2, 17, 9][i % 3])
    assert ran >= 40 and served == ran and not bad, (ran, served, bad[:3])


def test_emulated_logadd_on_the_lexicon_lane_engine(emu_session, oracle_lib, golden):
    """fltx_xlane.h with logAdd merges (LA): the committed vector, then random configurations (the emulator shares
    the host's libm: 1e-9)."""
    import test_gpu_batches
    c = cases.BY_NAME["lx_spell_t60_k12_logadd"]
    inp = helpers.case_inputs(c)
    d = emu_session.decoder(c, inp)
    d.decode_batch(inp["e"], [c["T"]], c["N"])
    assert d.get("engine") == 5 and d.get("redone") == 0 and d.get("why_not_lane") == 0
    ok, why = helpers.check_against_golden(d.results(0), golden[c["name"]], 1e-9)
    d.close()
    assert ok, why
    on5, bad = test_gpu_batches._logadd_lexicon_grid(emu_session, oracle_lib, 200, 4, [1, 5, 20, 40, 70], 1e-9)
    assert on5 == 200 and not bad, (on5, bad[:3])


def test_emulated_logadd_on_the_lexicon_lane_engine_with_lm_terms(emu_session, oracle_lib, golden):
    """fltx_ylane.h with logAdd merges (LMK bit 3): the committed vector, then random configurations."""
    import test_gpu_batches
    c = cases.BY_NAME["ng_word_logadd_t40"]
    inp = helpers.case_inputs(c)
    d = emu_session.decoder(c, inp)
    d.decode_batch(inp["e"], [c["T"]], c["N"])
    assert d.get("engine") == 6 and d.get("redone") == 0 and d.get("why_not_lane") == 0
    ok, why = helpers.check_against_golden(d.results(0), golden[c["name"]], 1e-9)
    d.close()
    assert ok, why
    on6, redone, bad = test_gpu_batches._logadd_lm_lexicon_grid(emu_session, oracle_lib, 200, 6, [1, 5, 20, 40, 70], 1e-9)
    assert on6 == 200 and redone <= 30 and not bad, (on6, redone, bad[:3])


def test_emulated_logadd_under_asg_and_over_homophones(emu_session, oracle_lib):
    import test_gpu_batches
    st, bad = test_gpu_batches._logadd_asg_homophone_grid(emu_session, oracle_lib, 180, 12, [1, 5, 20, 40, 70], 1e-9)
    assert not bad and all(v[1] == v[0] and v[2] <= v[0] // 8 for v in st.values()), (st, bad[:3])


def test_emulated_token_lm_on_the_lane_state_engine(emu_session, oracle_lib, golden):
    """fltx_slane.h's token-LM variant (a token-level n-gram LM on the lexicon-free decoder): the two vectors of the
    compiled reference on engine 4, then a slice of the GPU suite's grid (logAdd @1e-9: the emulator shares the host's libm)."""
    import test_gpu_batches
    for name in ("ng_tok_lexfree_t40", "ng_tok_lexfree_kt8"):
        c = cases.BY_NAME[name]
        inp = helpers.case_inputs(c)
        d = emu_session.decoder(c, inp)
        d.decode_batch(inp["e"], [c["T"]], c["N"])
        assert d.get("engine") == 4 and d.get("tlane") == 1 and d.get("redone") == 0 and d.get("why_not_lane") == 0
        ok, why = helpers.check_against_golden(d.results(0), golden[name])
        d.close()
        assert ok, why
    ran, served, bad = test_gpu_batches._token_lm_grid(emu_session, oracle_lib, 400, 5, [1, 2, 7, 20, 45], emu=True)
    assert ran >= 380 and served == ran and not bad, (ran, served, bad[:3])
    # beams beyond 64 (fltx_mlane.h's token-LM variant)
    ran, served, bad = test_gpu_batches._token_lm_grid(emu_session, oracle_lib, 90, 12, [1, 2, 7, 20, 45], emu=True,
                                                       beams=test_gpu_batches.WIDE_BEAMS, tokens=(8, 12, 29, 29, 30), log_add=0.25)
    assert ran >= 78 and served == ran and not bad, (ran, served, bad[:3])
    ran, served, bad = test_gpu_batches._token_lm_grid(emu_session, oracle_lib, 30, 13, [2, 7, 20], emu=True,
                                                       beams=(65, 100, 128, 129, 200, 256), tokens=(40, 64), log_add=0.2)
    assert ran >= 24 and served == ran and not bad, (ran, served, bad[:3])
    # the generic engine over the same dense table (lane engines switched off, streams, fallbacks), and without it
    ran, served, bad = test_gpu_batches._token_lm_grid(emu_session, oracle_lib, 60, 6, [2, 7, 20], emu=True, sets={"tlane": 0})
    assert ran >= 50 and served == 0 and not bad, (ran, served, bad[:3])
    assert not test_gpu_batches._token_lm_beyond_the_lane_engine(emu_session, oracle_lib, emu=True)
    assert not test_gpu_batches._token_lm_ragged_batch(emu_session, oracle_lib, emu=True)


def _token_lm_with_many_contexts(sess, oracle_lib, T=30):
    """A token-level 4-gram whose dense (context, token) table has more rows than the host builder scores in one block
    (fltx_api.cpp lmTokDense: 16 384 rows between two numbering passes, rows scored by several threads): the lane-state
    engine over it against the oracle, and the number of contexts against the model's own n-gram counts."""
    from text_amd import _capi, ngram_synth
    N = 29
    vocab = ngram_synth.words(N, "t")
    os.makedirs(helpers.NGRAM_DIR, exist_ok=True)
    path = os.path.join(helpers.NGRAM_DIR, "lm_tok_many_ctx_o4.arpa")
    if not os.path.exists(path):
        tmp = "%s.tmp%d" % (path, os.getpid())
        ngram_synth.write_arpa(tmp, vocab, 4, (0, 900, 22000, 30000), 13)
        os.replace(tmp, path)
    with open(path) as f:
        head = [next(f) for _ in range(5)]
    grams = [int(line.split("=")[1]) for line in head[1:5]]
    lm = _capi.ArpaLM(path, vocab, lib=sess.lib)
    olm = oracle_lib.lm_arpa_create(path.encode(), "\n".join(vocab).encode())
    bad = []
    for k, (K, crit, dist) in enumerate(((16, "ctc", "ctc"), (50, "ctc", "uniform"), (33, "asg", "ctc"))):
        c = cases.case(name="many_ctx%d" % k, kind="lexfree", dist=dist, T=T, N=N, K=K, u=930 + k, lm=("ngram", 4, 13),
                       lm_weight=0.9, is_lm_token=True, crit=crit, trans_seed=3 if crit == "asg" else None)
        inp = helpers.case_inputs(c)
        d = sess.decoder(c, inp, lm=lm)
        d.decode_batch(inp["e"], [T], N)
        n_ctx = d.get("toklm_contexts")
        assert d.get("engine") == 4 and d.get("tlane") == 1 and d.get("redone") == 0, (d.get("engine"), d.get("why_not_lane"))
        got = d.results(0)
        d.close()
        opt = orclib.make_options(c["K"], c["Kt"], c["thr"], c["lm_weight"], c["word_score"], c["unk_score"], c["sil_score"],
                                  c["log_add"], c["crit"])
        od = oracle_lib.lexfree(opt, olm, 0, N - 1 if crit == "ctc" else -1, inp["tr"])
        want = oracle_lib.decode(od, inp["e"], T, N)
        oracle_lib.decoder_destroy(od)
        ok, why = helpers.hyps_equal(got, want)
        if not ok:
            bad.append((c["name"], why))
    oracle_lib.lm_destroy(olm)
    # reachable KenLM states are n-grams of order < 4 (plus the empty context): more than one block of them, not more
    # than the model holds
    assert 16384 < n_ctx <= 1 + grams[0] + grams[1] + grams[2], (n_ctx, grams)
    return bad


def test_emulated_token_lm_table_built_in_several_blocks(emu_session, oracle_lib):
    assert not _token_lm_with_many_contexts(emu_session, oracle_lib, T=20)


def test_emulated_word_piece_engine(emu_session, oracle_lib):
    """fltx_wlane.h on the emulator: a thin slice of the GPU suite's grid (tests/test_gpu_batches.py)."""
    import test_gpu_batches
    ran, served, bad = test_gpu_batches._word_piece_grid(emu_session, oracle_lib, 71, lambda i: [2, 11, 7][i % 3], emu=True)
    assert ran >= 32 and served == ran and not bad, (ran, served, bad[:3])


def test_emulated_word_piece_fallback_reaches_the_generic_engine(emu_session):
    """a row without a defined token beam makes fltx_wlane.h flag its utterance; that utterance --
    and only that one -- is decoded again on the generic engine, the others keep their packed (wide-token) records."""
    from text_amd import synth
    N, T, B = 300, 12, 3
    c = cases.case("wp_ties_emu", dist="ctc", T=T, N=N, K=20, Kt=30, u=515)
    e = synth.batch("ctc", B, T, N)
    e[1, 5, :] = -3.0
    d = emu_session.decoder(c, dict(tr=None))
    d.decode_batch(e, [T] * B, N)
    assert (d.get("engine"), d.get("wlane"), d.get("redone")) == (4, 1, 1)
    g = emu_session.decoder(c, dict(tr=None))
    g.set("wlane", 0)
    g.decode_batch(e, [T] * B, N)
    assert g.get("engine") != 4
    for b in range(B):
        ok, why = helpers.hyps_equal(g.results(b), d.results(b))
        assert ok, "utterance %d: %s" % (b, why)
    d.close()
    g.close()


def _deferred_look(sess, golden, T_wp):
    """fltx_decoder_set("defer_check", 1): fltx_decode_batch returns with its kernels queued; the look at the statuses
    and the second pass of what a fast path flagged happen when results are first read -- same results."""
    from text_amd import synth
    # (1) a fast path that flags one utterance (fltx_wlane.h, a row without a defined token beam)
    N, B = 300, 3
    c = cases.case("wp_ties_defer", dist="ctc", T=T_wp, N=N, K=20, Kt=30, u=515)
    e = synth.batch("ctc", B, T_wp, N)
    e[1, 5, :] = -3.0
    want = sess.decoder(c, dict(tr=None))
    want.decode_batch(e, [T_wp] * B, N)
    d = sess.decoder(c, dict(tr=None))
    d.set("defer_check", 1)
    d.decode_batch(e, [T_wp] * B, N)
    assert d.get("redone") == 1 and want.get("redone") == 1
    for b in range(B):
        ok, why = helpers.hyps_equal(want.results(b), d.results(b))
        assert ok, "utterance %d: %s" % (b, why)
    # ... and a second batch on the same decoder object starts clean
    d.decode_batch(e[:1], [T_wp], N)
    ok, why = helpers.hyps_equal(want.results(0), d.results(0))
    assert ok, why
    d.close()
    want.close()
    # (2) the lexicon decoder's score cut forced tight: flagged utterances are decoded again without it
    c = cases.BY_NAME["lx_spell_t40_k8"]
    inp = helpers.case_inputs(c)
    d = sess.decoder(c, inp)
    d.set("defer_check", 1)
    d.set("xlane", 0)
    d.set("ylane", 0)
    d.set("cut_m", c["K"] + 1)
    d.decode_batch(inp["e"], [c["T"]], c["N"])
    ok, why = helpers.check_against_golden(d.results(0), golden[c["name"]])
    d.close()
    assert ok, why
    # (3) the headline shape's engine, nothing flagged
    c = cases.BY_NAME["lf_ctc_t60_k10"]
    inp = helpers.case_inputs(c)
    d = sess.decoder(c, inp)
    d.set("defer_check", 1)
    d.decode_batch(inp["e"], [c["T"]], c["N"])
    ok, why = helpers.check_against_golden(d.results(0), golden[c["name"]])
    assert ok and d.get("engine") == 4 and d.get("redone") == 0, why
    d.close()
    # (4) a batch nobody reads is settled all the same by the next fltx_decode_batch on the object (round 6: it used to be
    # dropped, so a caller that only queues batches never learned about a fallback) -- unless the caller asks for that
    N, B = 300, 3
    c = cases.case("wp_ties_unread", dist="ctc", T=T_wp, N=N, K=20, Kt=30, u=515)
    e = synth.batch("ctc", B, T_wp, N)
    e[1, 5, :] = -3.0  # (a row without a defined token beam: fltx_wlane.h flags the utterance)
    d = sess.decoder(c, dict(tr=None))
    d.set("defer_check", 1)
    d.decode_batch(e, [T_wp] * B, N)
    d.decode_batch(e, [T_wp] * B, N)  # settles the first: its flagged utterance is decoded again, unread
    assert d.get("unread_redone") == 1 and d.get("looks_dropped") == 0
    # (one of three utterances fell back: more than a quarter, so the fallback sticks -- what a settled look is for --
    # and the second batch started on the general engine: nothing of it is decoded again)
    assert d.get("redone") == 0
    ref = sess.decoder(c, dict(tr=None))
    ref.decode_batch(e, [T_wp] * B, N)
    for b in range(B):
        ok, why = helpers.hyps_equal(ref.results(b), d.results(b))
        assert ok, "utterance %d: %s" % (b, why)
    ref.close()
    d.close()
    d = sess.decoder(c, dict(tr=None))
    d.set("defer_check", 2)  # (measurements only: an unread batch's look is dropped, and counted)
    d.decode_batch(e, [T_wp] * B, N)
    d.decode_batch(e, [T_wp] * B, N)
    assert d.get("looks_dropped") == 1 and d.get("unread_redone") == 0
    d.close()


def test_emulated_deferred_status_look(emu_session, golden):
    _deferred_look(emu_session, golden, 12)
