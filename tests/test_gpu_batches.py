"""GPU (-m gpu): the BASELINE.json batch shapes of the lexicon decoder as they are
benchmarked -- C3 (90k-word trie, ZeroLM, B=256), C4 (trie + synthetic 4-gram,
B=256) and C5's per-GPU share (C4 at B=1024: four launch rounds of 256
workgroups) -- through the C ABI.  Utterances with a committed golden equal it;
sampled utterances equal the oracle; every 17th utterance passes size-independent
checks: sorted n-best, emitting-model score = sum of the emissions along the
token path, and the collapsed token path between two word ends spells the word
that was emitted."""
import numpy as np
import pytest

import cases
import helpers
from text_amd import synth

pytestmark = pytest.mark.gpu


def _spelling(lex, w):
    sf, so = lex
    return [int(x) for x in sf[so[w]:so[w + 1]]]


def _check_structure(h, e, lex, T, N, zero_lm, opt):
    """size-independent properties of one hypothesis of the lexicon decoder (CTC)."""
    blank = N - 1
    assert len(h.tokens) == T + 2 and h.tokens[0] == 0 and h.tokens[-1] == 0
    path = e[np.arange(T), h.tokens[1:-1]].astype(np.float64)
    acc = 0.0
    for v in path:
        acc += v
    assert acc == h.am  # LexiconDecoder.cpp:69,105: am = prev.am + emission, in frame order
    letters, prev, n_words, n_sil = [], None, 0, 0
    for f in range(1, T + 1):
        t, w = int(h.tokens[f]), int(h.words[f])
        n_sil += 1 if t == 0 else 0
        if t != blank and t != prev:
            letters.append(t)
        prev = t
        if w >= 0:
            n_words += 1
            while letters and letters[0] == 0:
                letters.pop(0)  # sil emitted while waiting at the root
            assert letters == _spelling(lex, w), (f, w, letters, _spelling(lex, w))
            letters = []
    if zero_lm:
        assert h.lm == 0.0 and h.score == h.am
    else:
        want = h.am + opt["lm_weight"] * h.lm + opt["word_score"] * n_words + opt["sil_score"] * n_sil
        assert abs(want - h.score) < 1e-6 * max(1.0, abs(h.score))
    return n_words


def _run_batch(gpu_session, golden, oracle_lib, name0, name255, B, sample, sets=None, engine=None, expect=None):
    c = cases.BY_NAME[name0]
    inp = helpers.case_inputs(c)
    T, N = c["T"], c["N"]
    e = synth.batch("lexspell", B, T, N, lexicon=inp["lex"])
    d = gpu_session.decoder(c, inp)
    for k, v in (sets or {}).items():
        d.set(k, v)
    d.decode_batch(e, [T] * B, N)
    assert d.get("lds") == 1
    if engine is not None:
        assert d.get("engine") == engine and d.get("redone") == 0
    for k, v in (expect or {}).items():
        assert d.get(k) == v, (k, d.get(k), v)
    for b, name in ((0, name0), (255, name255)):
        if name and b < B:
            ok, why = helpers.check_against_golden(d.results(b), golden[name])
            assert ok, "%s: %s" % (name, why)
    opt = dict(lm_weight=c["lm_weight"], word_score=c["word_score"], sil_score=c["sil_score"])
    words = 0
    for b in range(0, B, 17):
        hyps = d.results(b)
        assert 0 < len(hyps) <= c["K"]
        sc = [h.score for h in hyps]
        assert all(x > y for x, y in zip(sc, sc[1:]))
        for h in hyps[:3]:
            words += _check_structure(h, e[b], inp["lex"], T, N, c["lm"] == "zero", opt)
    assert words > 0
    for b in sample:
        cb = dict(c, u=b)
        want = helpers.run_checker(oracle_lib, cb, dict(inp, e=e[b]))
        ok, why = helpers.hyps_equal(want, d.results(b))
        assert ok, "utterance %d vs oracle: %s" % (b, why)
    d.close()


@pytest.mark.parametrize("sets,engine", [({}, 5), ({"yshare": 1}, 5), ({"xlane": 0}, 6), ({"xlane": 0, "ylane": 0}, 0)],
                         ids=["lane-engine", "lane-engine-sharing-a-cu", "lane-engine-with-lm-terms", "generic-engine"])
def test_c3_batch_of_256(gpu_session, golden, oracle_lib, sets, engine):
    """C3 as benchmarked: served by the lane = (LM state, trie node) engine (fltx_xlane.h); the
    generic engine, which takes over whenever that one does not apply, on the same batch."""
    _run_batch(gpu_session, golden, oracle_lib, "C3_spell_u0", "C3_spell_u255", 256, sample=[97, 201], sets=sets,
               engine=engine)


@pytest.mark.parametrize("sets,engine", [({}, 6), ({"yshare": 1}, 6), ({"ylane": 0}, 0)],
                         ids=["lane-engine", "lane-engine-sharing-a-cu", "generic-engine"])
def test_c4_batch_of_256(gpu_session, golden, oracle_lib, sets, engine):
    """C4 as benchmarked: served by fltx_ylane.h (n-gram word LM, smeared trie, beam 100 = two lane
    groups); the generic engine on the same batch."""
    _run_batch(gpu_session, golden, oracle_lib, "C4_spell_u0", "C4_spell_u255", 256, sample=[131], sets=sets,
               engine=engine)


def test_c5_share_of_one_gpu_1024_utterances(gpu_session, golden, oracle_lib):
    """BASELINE.json configs[4]: 8192 utterances over 8 GPUs = 1024 per GPU, i.e. four launch
    rounds of the 256 workgroups a device runs at a time."""
    _run_batch(gpu_session, golden, oracle_lib, "C4_spell_u0", "C4_spell_u255", 1024, sample=[700, 1023], engine=6,
               expect={"yshare": 1, "threads": 512})  # more utterances than CUs: two workgroups share a CU


def test_random_configurations_slice(gpu_session, oracle_lib):
    """A bounded slice of tools/fuzz_big.py: random option / size / LM / lexicon combinations
    against the oracle (ties in the reference's n-best are skipped: its own result is order
    dependent there)."""
    bad, ran = [], 0
    for c in cases.fuzz_cases(460)[40:]:
        inp = helpers.case_inputs(c)
        want = helpers.run_checker(oracle_lib, c, inp)
        if len({h.score for h in want}) != len(want):
            continue
        got = gpu_session.run(c, inp)
        ok, why = helpers.hyps_equal(want, got, 1e-5 if c["log_add"] else 0.0)
        ran += 1
        if not ok:
            bad.append((c["name"], why))
    assert ran > 200 and not bad, bad[:3]


def test_edge_configurations_of_the_lexicon_free_decoders(gpu_session, oracle_lib):
    """tools/edge_cases.py inside the suite: tiny token sets, beam 1 and beam = a wave's lanes,
    thresholds 0 / inf (with an unbounded threshold a score of -inf passes every comparison:
    what exists must be told by the bookkeeping, not by the scores), token beams of 1 and 3,
    silScore of both signs, ASG with transitions, one-frame utterances."""
    import itertools
    bad, ran = [], 0
    grid = itertools.product([2, 3, 29, 64], [1, 2, 50, 64], [0.0, 1.5, 25.0, float("inf")], [None, 1, 3],
                             [0.0, -0.7, 0.4], ["ctc", "asg"], [1, 2, 17, 120])
    for i, (N, K, thr, Kt, sil, crit, T) in enumerate(grid):
        if i % 7 not in (0, 3) or (crit == "asg" and N == 2):
            continue
        c = cases.case("edge%d" % i, dist=["ctc", "uniform"][i % 2], u=500 + i, T=T, N=N, K=K,
                       Kt=min(N, Kt) if Kt else None, thr=thr, sil_score=sil, crit=crit,
                       trans_seed=(90 + i) if crit == "asg" else None)
        inp = helpers.case_inputs(c)
        want = helpers.run_checker(oracle_lib, c, inp)
        if len({h.score for h in want}) != len(want):
            continue
        ok, why = helpers.hyps_equal(want, gpu_session.run(c, inp))
        ran += 1
        if not ok:
            bad.append(({k: c[k] for k in ("N", "K", "Kt", "thr", "sil_score", "crit", "T", "dist")}, why))
    assert ran > 1000 and not bad, bad[:3]


def test_edge_configurations_of_the_lexicon_lane_engine(gpu_session, oracle_lib):
    """LexiconDecoder + ZeroLM through fltx_xlane.h against the oracle: beam 1 .. 64, thresholds
    0 / inf, token beams of 1 / 3 / 10 (stay and blank are not subject to it, the trie parent's
    extension is), silScore and wordScore of both signs, one-frame utterances, `uniform` rows
    (every token plausible: lanes drop out and come back all the time) and `lexspell` rows."""
    import itertools
    bad, ran, served = [], 0, 0
    grid = itertools.product([1, 2, 7, 50, 64], [0.0, 1.5, 25.0, float("inf")], [None, 1, 3, 10], [0.0, -0.7, 0.4],
                             [0.0, 1.5, -2.0], [1, 2, 17, 120], ["lexspell", "uniform"])
    for i, (K, thr, Kt, sil, ws, T, dist) in enumerate(grid):
        if i % 11 not in (0, 4):
            continue
        c = cases.case("xedge%d" % i, kind="lexicon", dist=dist, u=700 + i, T=T, K=K, Kt=Kt, thr=thr, sil_score=sil,
                       word_score=ws, lexicon=cases.SMALL_LEX)
        inp = helpers.case_inputs(c)
        want = helpers.run_checker(oracle_lib, c, inp)
        if len({h.score for h in want}) != len(want):
            continue
        got = gpu_session.run(c, inp)
        served += gpu_session.last_engine == 5
        ok, why = helpers.hyps_equal(want, got)
        ran += 1
        if not ok:
            bad.append(({k: c[k] for k in ("K", "Kt", "thr", "sil_score", "word_score", "T", "dist")}, why))
    assert ran > 300 and served == ran and not bad, (ran, served, bad[:3])


@pytest.mark.parametrize("kind", ["lexfree", "lexicon", "lexicon-lm"])
def test_utterances_handed_to_the_generic_engine_inside_a_lane_engine_batch(gpu_session, kind):
    """The lane engines write packed history records and flag each utterance they finished
    (ST_PACKED); one whose frame has no finite candidate (a row of -inf) is decoded again on the
    generic engine, which writes plain records, and the back-trace has to read each kind as what
    it is.  Every utterance must equal what the generic engine alone returns for it."""
    c = cases.BY_NAME[{"lexfree": "lf_ctc_t60_k10", "lexicon": "lx_spell_t60_k12_full",
                       "lexicon-lm": "ng_word_t60_k16_4g"}[kind]]
    inp = helpers.case_inputs(c)
    T, N, B = c["T"], c["N"], 6
    e = synth.batch("ctc" if kind == "lexfree" else "lexspell", B, T, N, lexicon=inp["lex"])
    e[2, 17, :] = -np.inf
    e[4, 0, :] = -np.inf
    res = {}
    for name, sets in (("lane", {}), ("generic", {"slane": 0, "xlane": 0, "ylane": 0, "lane": 0, "lean": 0})):
        d = gpu_session.decoder(c, inp)
        for k, v in sets.items():
            d.set(k, v)
        d.decode_batch(e, [T] * B, N)
        res[name] = ([d.results(b) for b in range(B)], d.get("engine"), d.get("redone"))
        d.close()
    assert res["lane"][1] in (4, 5, 6) and res["lane"][2] == 2 and res["generic"][1] in (0, 1)
    for b in range(B):
        ok, why = helpers.hyps_equal(res["generic"][0][b], res["lane"][0][b])
        assert ok, "utterance %d: %s" % (b, why)


@pytest.mark.parametrize("yshare", [-1, 1], ids=["memo-in-lds", "shares-a-cu"])
def test_edge_configurations_of_the_lexicon_lane_engine_with_lm_terms(gpu_session, oracle_lib, yshare):
    """(yshare = 1: the geometry of which two workgroups fit a CU -- 512 threads, LM-state memo in HBM.)
    fltx_ylane.h against the oracle: n-gram word LMs of order 2 .. 4 and label scores without an
    LM (smearing only), beams 1 .. 128 (one and two lane groups), thresholds 0 .. inf, token
    beams, lmWeight / wordScore / silScore of both signs, one-frame utterances, `uniform` rows."""
    import itertools
    bad, ran, served = [], 0, 0
    grid = itertools.product([1, 2, 7, 64, 65, 100, 128], [0.0, 2.0, 25.0, float("inf")], [None, 3, 10],
                             [("ngram", 2, 71), ("ngram", 4, 72), "scores"], [0.7, 2.0, -0.5], [0.0, 1.5, -2.0],
                             [0.0, -0.7], [1, 17, 90], ["lexspell", "uniform"])
    for i, (K, thr, Kt, lm, lw, ws, sil, T, dist) in enumerate(grid):
        if i % 53 not in (0, 19):
            continue
        c = cases.case("yedge%d" % i, kind="lexicon", dist=dist, u=900 + i, T=T, K=K, Kt=Kt, thr=thr, sil_score=sil,
                       word_score=ws, lm_weight=lw, lexicon=cases.SMALL_LEX, lm="zero" if lm == "scores" else lm,
                       label_scores=(50 + i % 7) if lm == "scores" else None)
        inp = helpers.case_inputs(c)
        want = helpers.run_checker(oracle_lib, c, inp)
        if len({h.score for h in want}) != len(want):
            continue
        # (every third: token waves with more than max(8, beam) pairs rank their own -- the fan-out path)
        got = gpu_session.run(c, inp, sets={"yshare": yshare, "ylane_rank_at": 8 if ran % 3 == 0 else 0})
        served += gpu_session.last_engine == 6
        ok, why = helpers.hyps_equal(want, got)
        ran += 1
        if not ok:
            bad.append(({k: c[k] for k in ("K", "Kt", "thr", "lm", "lm_weight", "sil_score", "word_score", "T", "dist",
                                           "label_scores")}, why))
    assert ran > 250 and served == ran and not bad, (ran, served, bad[:3])


def _logadd_lexicon_grid(sess, oracle_lib, n, seed, frames, tol):
    """random LexiconDecoder + ZeroLM configurations with logAdd merges on fltx_xlane.h (LA): every utterance on
    engine 5, none handed back, n-best within `tol` of the oracle (the device's log1p / exp against the host's)"""
    import random
    rnd = random.Random(seed)
    on5 = 0
    bad = []
    for i in range(n):
        c = cases.case("xla%d" % i, kind="lexicon", dist=rnd.choice(["lexspell", "lexspell", "uniform"]), T=rnd.choice(frames),
                       K=rnd.choice([1, 3, 8, 20, 40, 64]), Kt=rnd.choice([29, 29, 10, 4]), thr=rnd.choice([25.0, 8.0, 2.0, 100.0]),
                       lexicon=cases.SMALL_LEX, u=3000 + i, log_add=True, word_score=rnd.choice([0.0, 1.5, -0.5]),
                       sil_score=rnd.choice([0.0, -0.5]))
        inp = helpers.case_inputs(c)
        d = sess.decoder(c, inp)
        if i % 3 == 2:
            d.set("yshare", 1)  # (the geometry with the LM-state memo in HBM)
        d.decode_batch(inp["e"], [c["T"]], c["N"])
        got = d.results(0)
        on5 += int(d.get("engine") == 5 and d.get("redone") == 0)
        d.close()
        ok, why = helpers.hyps_equal(helpers.run_checker(oracle_lib, c, inp), got, tol)
        if not ok:
            bad.append(({k: c[k] for k in ("dist", "T", "K", "Kt", "thr", "word_score", "sil_score")}, why))
    return on5, bad


def _logadd_lm_lexicon_grid(sess, oracle_lib, n, seed, frames, tol):
    """... with the LM terms, on fltx_ylane.h (LMK bit 3): n-gram word LMs, label scores without an LM, ZeroLM beyond
    beam 64; one, two and four lane groups.  An utterance whose token wave would have to rank its own pairs leaves for
    the generic engine (that ranking orders pairs by their best member, the frame selects on the sums): counted."""
    import random
    rnd = random.Random(seed)
    on6 = redone = 0
    bad = []
    for i in range(n):
        kind = rnd.choice(["ngram", "ngram", "scores", "zero"])
        K = rnd.choice([3, 10, 24, 50, 64, 65, 100, 128, 129, 200, 256]) if kind != "zero" else rnd.choice([65, 100, 128, 200, 256])
        kw = dict(lm=("ngram", rnd.choice([2, 3, 4]), 70 + i % 5), lm_weight=rnd.choice([0.5, 1.3, 2.0])) if kind == "ngram" else \
            (dict(label_scores=80 + i % 3, lm_weight=rnd.choice([0.7, 1.5])) if kind == "scores" else {})
        c = cases.case("yla%d" % i, kind="lexicon", dist=rnd.choice(["lexspell", "lexspell", "uniform"]), T=rnd.choice(frames),
                       K=K, Kt=rnd.choice([29, 29, 10, 4]), thr=rnd.choice([25.0, 8.0, 2.0, 100.0]),
                       lexicon=rnd.choice([cases.SMALL_LEX, (3000, 77)]), u=4000 + i, log_add=True,
                       word_score=rnd.choice([0.0, 1.5, -0.5]), sil_score=rnd.choice([0.0, -0.5]), **kw)
        inp = helpers.case_inputs(c)
        d = sess.decoder(c, inp)
        if i % 3 == 2:
            d.set("yshare", 1)
        d.decode_batch(inp["e"], [c["T"]], c["N"])
        got = d.results(0)
        on6 += int(d.get("engine") == 6)
        redone += d.get("redone")
        d.close()
        ok, why = helpers.hyps_equal(helpers.run_checker(oracle_lib, c, inp), got, tol)
        if not ok:
            bad.append((kind, {k: c[k] for k in ("dist", "T", "K", "Kt", "thr", "word_score", "sil_score", "lm_weight")}, why))
    return on6, redone, bad


def _token_lm_grid(sess, oracle_lib, n, seed, frames, emu=False, sets=None, beams=(1, 2, 5, 10, 24, 50, 64),
                   tokens=(8, 12, 29, 29, 29, 40, 64), log_add=0.3):
    """random LexiconFreeDecoder + token-level n-gram LM configurations (LexiconFreeDecoder.cpp:69-85 with KenLM::score,
    lm/KenLM.cpp:63-83) on fltx_slane.h's token-LM variant: orders 2 - 4, CTC / ASG, token beams, thresholds, silScore,
    lmWeight of both signs, max-merge (bit-exact) and logAdd (1e-5 on the device, 1e-9 on the emulator, which shares
    the host's libm) -> (configurations compared, served by engine 4 with nothing redone, mismatches).  beams beyond 64:
    fltx_mlane.h's token-LM variant"""
    import random
    rnd = random.Random(seed)
    ran = served = 0
    bad = []
    for i in range(n):
        N = rnd.choice(list(tokens))
        crit = rnd.choice(["ctc", "ctc", "asg"])
        la = rnd.random() < log_add
        c = cases.case("tlm%d" % i, dist=rnd.choice(["ctc", "ctc", "uniform"]), T=rnd.choice(frames), N=N,
                       K=rnd.choice(list(beams)), Kt=rnd.choice([N, N, max(1, N // 3), 3, 1]),
                       thr=rnd.choice([25.0, 8.0, 2.0, 100.0, float("inf")]), u=7000 + i, log_add=la,
                       sil_score=rnd.choice([0.0, -0.5, 0.4]), crit=crit, trans_seed=(90 + i % 7) if crit == "asg" else None,
                       lm=("ngram", rnd.choice([2, 3, 4]), 50 + i % 4), lm_weight=rnd.choice([0.5, 0.8, 1.5, 2.5, -0.3]))
        inp = helpers.case_inputs(c)
        want = helpers.run_checker(oracle_lib, c, inp)
        if len({h.score for h in want}) != len(want):
            continue  # (equal scores in the n-best: the reference's order is its sort's)
        if la and any(abs(a.score - b.score) < 1e-4 for a, b in zip(want, want[1:])):
            continue  # (near ties: a different libm may order them differently)
        d = sess.decoder(c, inp)
        for k, v in (sets or {}).items():
            d.set(k, v)
        d.decode_batch(inp["e"], [c["T"]], c["N"])
        got = d.results(0)
        served += int(d.get("engine") == 4 and d.get("tlane") == 1 and d.get("redone") == 0)
        d.close()
        ran += 1
        ok, why = helpers.hyps_equal(want, got, (1e-9 if emu else 1e-5) if la else 0.0)
        if not ok:
            bad.append(({k: c[k] for k in ("dist", "T", "N", "K", "Kt", "thr", "sil_score", "crit", "log_add", "lm", "lm_weight", "u")}, why))
    return ran, served, bad


@pytest.mark.gpu
def test_token_level_ngram_lm_on_the_lane_state_engine(gpu_session, oracle_lib):
    """2 000 random configurations of the lexicon-free decoder with a token-level n-gram LM (orders 2 - 4, CTC / ASG,
    token beams, logAdd) against the oracle: all on engine 4 (fltx_slane.h, token-LM variant), nothing handed back,
    max-merge bit-exact (score, emitting-model score, LM score), logAdd @1e-5; a slice of them on the 512-thread
    geometry of which two workgroups share a CU, and the generic engine on the same inputs."""
    ran, served, bad = _token_lm_grid(gpu_session, oracle_lib, 2000, 6, [1, 2, 7, 20, 45, 90])
    assert ran >= 1900 and served == ran and not bad, (ran, served, bad[:3])
    ran, served, bad = _token_lm_grid(gpu_session, oracle_lib, 300, 7, [5, 33, 120], sets={"slane_threads": 512})
    assert ran >= 280 and not bad, (ran, served, bad[:3])
    ran, served, bad = _token_lm_grid(gpu_session, oracle_lib, 200, 8, [5, 33], sets={"tlane": 0})
    assert ran >= 180 and served == 0 and not bad, (ran, served, bad[:3])
    # beams beyond 64: fltx_mlane.h's token-LM variant (two, four, eight lane groups), max-merge and logAdd, up to 30 listed tokens
    ran, served, bad = _token_lm_grid(gpu_session, oracle_lib, 500, 10, [1, 2, 7, 20, 45, 90, 200], beams=WIDE_BEAMS,
                                      tokens=(8, 12, 29, 29, 30), log_add=0.25)
    assert ran >= 470 and served == ran and not bad, (ran, served, bad[:3])
    # ... token lists of up to 64 at beams up to 256 (the wide geometries)
    ran, served, bad = _token_lm_grid(gpu_session, oracle_lib, 150, 11, [2, 7, 20, 45, 90], beams=(65, 100, 128, 129, 200, 256),
                                      tokens=(40, 64), log_add=0.2)
    assert ran >= 130 and served == ran and not bad, (ran, served, bad[:3])
    # ... and the generic engine without the dense table (a chain of n-gram probes per look-up: what token sets beyond 64 get)
    ran, served, bad = _token_lm_grid(gpu_session, oracle_lib, 120, 9, [5, 33], sets={"tlane": 0, "tok_dense": 0})
    assert ran >= 100 and served == 0 and not bad, (ran, served, bad[:3])


WIDE_BEAMS = (65, 80, 100, 128, 129, 200, 256, 257, 300, 512)


def _token_lm_beyond_the_lane_engine(sess, oracle_lib, emu=False):
    """beams 65 .. 300 with the lane engines switched off, or without the dense table: the generic engine, a state's
    context as a row of the dense table / its n-gram probe chain"""
    import random
    rnd = random.Random(31)
    bad = []
    for i, K in enumerate([65, 100, 128, 200, 300, 70, 90]):
        c = cases.case("tlbig%d" % i, dist=rnd.choice(["ctc", "uniform"]), T=rnd.choice([9, 30]) if emu else rnd.choice([30, 80]),
                       N=rnd.choice([12, 29]), K=K, Kt=rnd.choice([29, 6]), thr=rnd.choice([25.0, 8.0]), u=7600 + i,
                       lm=("ngram", rnd.choice([2, 3, 4]), 50 + i % 4), lm_weight=rnd.choice([0.8, 1.5]), sil_score=rnd.choice([0.0, -0.4]))
        inp = helpers.case_inputs(c)
        want = helpers.run_checker(oracle_lib, c, inp)
        if len({h.score for h in want}) != len(want):
            continue
        for sets in ({"tlane": 0}, {"tok_dense": 0}):
            d = sess.decoder(c, inp)
            for k, v in sets.items():
                d.set(k, v)
            d.decode_batch(inp["e"], [c["T"]], c["N"])
            got, eng, ctxs = d.results(0), d.get("engine"), d.get("toklm_contexts")
            d.close()
            ok, why = helpers.hyps_equal(want, got)
            if not ok or eng > 1 or (ctxs > 0) != ("tlane" in sets):
                bad.append((K, sets, eng, ctxs, why))
    return bad


@pytest.mark.gpu
def test_token_lm_beyond_the_lane_engine(gpu_session, oracle_lib):
    assert not _token_lm_beyond_the_lane_engine(gpu_session, oracle_lib)


@pytest.mark.gpu
def test_token_lm_table_built_in_several_blocks(gpu_session, oracle_lib):
    """A token 4-gram with more contexts than one block of the dense table's host builder (tests/test_emu_logic.py)."""
    import test_emu_logic
    assert not test_emu_logic._token_lm_with_many_contexts(gpu_session, oracle_lib, T=300)


@pytest.mark.gpu
def test_token_lm_batch_at_the_c2_shape(gpu_session, oracle_lib):
    """BASELINE configs[1]'s shape (256 utterances, T = 1000, N = 29, beam 50) with a token 3-gram: engine 4, nothing
    redone, sampled utterances equal to the oracle bit for bit, and the generic engine's n-best on every utterance."""
    B, T, N, K = 256, 1000, 29, 50
    c = cases.case("c2tok", dist="ctc", T=T, N=N, K=K, u=0, lm=("ngram", 3, 11), lm_weight=0.8)
    inp = helpers.case_inputs(c)
    e = synth.batch("ctc", B, T, N)
    d = gpu_session.decoder(c, inp)
    d.decode_batch(e, [T] * B, N)
    assert d.get("engine") == 4 and d.get("tlane") == 1 and d.get("redone") == 0
    g = gpu_session.decoder(c, inp)
    g.set("tlane", 0)
    g.decode_batch(e, [T] * B, N)
    assert g.get("engine") == 1  # (the generic engine's dense merge)
    for b in range(B):
        ok, why = helpers.hyps_equal(g.results(b), d.results(b))
        assert ok, (b, why)
    for b in (0, 101, 255):
        cb = dict(c, u=b)
        want = helpers.run_checker(oracle_lib, cb, dict(inp, e=e[b]))
        ok, why = helpers.hyps_equal(want, d.results(b))
        assert ok, (b, why)
    d.close()
    g.close()


def _token_lm_ragged_batch(sess, oracle_lib, emu=False):
    """a batch of utterances of different lengths (an empty one, one frame, ...) with a token LM, at beams on both sides
    of 64: every utterance against the oracle -> mismatches"""
    bad = []
    Ts = [0, 1, 9, 12] if emu else [0, 1, 17, 40, 33, 2, 40]
    for K, groups in ((50, 1), (100, 2)) if emu else ((50, 1), (100, 2), (300, 8)):
        c = cases.case("tlrag%d" % K, dist="ctc", T=40, N=29, K=K, u=0, lm=("ngram", 3, 11), lm_weight=0.9)
        inp = helpers.case_inputs(c)
        e = synth.batch("ctc", len(Ts), 40, 29)
        flat = np.concatenate([e[i, :t].reshape(-1) for i, t in enumerate(Ts)]).astype(np.float32)
        d = sess.decoder(c, inp)
        d.decode_batch(flat, Ts, 29)
        if not (d.get("engine") == 4 and d.get("tlane") == 1 and d.get("lane_groups") == groups and d.get("redone") == 0):
            bad.append((K, "engine", d.get("engine"), d.get("lane_groups"), d.get("redone")))
        for i, t in enumerate(Ts):
            want = helpers.run_checker(oracle_lib, dict(c, T=t, u=i), dict(inp, e=np.ascontiguousarray(e[i, :t]).reshape(-1)))
            ok, why = helpers.hyps_equal(d.results(i), want)
            if not ok:
                bad.append((K, i, t, why))
        d.close()
    return bad


@pytest.mark.gpu
def test_token_lm_ragged_batch(gpu_session, oracle_lib):
    assert not _token_lm_ragged_batch(gpu_session, oracle_lib)


@pytest.mark.gpu
@pytest.mark.parametrize("K,groups", [(100, 2), (200, 4)])
def test_token_lm_batch_at_the_c2_shape_with_lane_groups(gpu_session, oracle_lib, K, groups):
    """... at beams beyond 64: fltx_mlane.h's token-LM variant (state ids from the table in HBM) on 256 utterances of
    1 000 frames, every utterance against the generic engine's n-best, sampled utterances against the oracle."""
    B, T, N = 256, 1000, 29
    c = cases.case("c2tok_k%d" % K, dist="ctc", T=T, N=N, K=K, u=0, lm=("ngram", 3, 11), lm_weight=0.8)
    inp = helpers.case_inputs(c)
    e = synth.batch("ctc", B, T, N)
    d = gpu_session.decoder(c, inp)
    d.decode_batch(e, [T] * B, N)
    assert d.get("engine") == 4 and d.get("tlane") == 1 and d.get("lane_groups") == groups and d.get("redone") == 0
    g = gpu_session.decoder(c, inp)
    g.set("tlane", 0)
    g.decode_batch(e, [T] * B, N)
    assert g.get("engine") == 1
    for b in range(B):
        ok, why = helpers.hyps_equal(g.results(b), d.results(b))
        assert ok, (b, why)
    for b in (0, 77, 255):
        cb = dict(c, u=b)
        want = helpers.run_checker(oracle_lib, cb, dict(inp, e=e[b]))
        ok, why = helpers.hyps_equal(want, d.results(b))
        assert ok, (b, why)
    d.close()
    g.close()


def _logadd_asg_homophone_grid(sess, oracle_lib, n, seed, frames, tol):
    """... under the ASG criterion, over lexicons with several words per spelling (n-gram LM, beams up to 128), and both:
    -> {mode: [configurations, on engine 6, redone, mismatches]}.  Equal-score hypotheses that hold the words of one
    spelling in another order are accepted (tests/test_multilabel.py says why)."""
    import random
    rnd = random.Random(seed)
    st = {"asg": [0, 0, 0, 0], "ml": [0, 0, 0, 0], "mlasg": [0, 0, 0, 0]}
    bad = []
    for i in range(n):
        mode = ["asg", "ml", "mlasg"][i % 3]
        asg, ml = mode != "ml", mode != "asg"
        if ml:
            big = rnd.random() < 0.5
            lexi = (cases.MULTI_NODUP_LEX_3K if big else cases.MULTI_NODUP_LEX) if asg else (cases.MULTI_LEX_3K if big else cases.MULTI_LEX)
            K = rnd.choice([3, 10, 24, 50, 64, 65, 100, 128])
            kw = dict(lm=("ngram", rnd.choice([2, 3, 4]), 60 + i % 5), lm_weight=rnd.choice([0.5, 1.3, 2.0]))
        else:
            lexi = cases.NODUP_LEX
            K = rnd.choice([3, 10, 24, 50, 64, 65, 100, 128, 129, 200, 256])
            kind = rnd.choice(["ngram", "scores", "zero"])
            kw = dict(lm=("ngram", rnd.choice([2, 3, 4]), 70 + i % 5), lm_weight=rnd.choice([0.5, 1.3, 2.0])) if kind == "ngram" else \
                (dict(label_scores=80 + i % 3, lm_weight=rnd.choice([0.7, 1.5])) if kind == "scores" else {})
        c = cases.case("lam%d" % i, kind="lexicon", dist=rnd.choice(["lexspell", "lexspell", "uniform"]), T=rnd.choice(frames), K=K,
                       Kt=rnd.choice([29, 29, 10, 4]), thr=rnd.choice([25.0, 8.0, 2.0, 100.0]), lexicon=lexi, u=6000 + i,
                       log_add=True, crit="asg" if asg else "ctc", trans_seed=(50 + i % 7) if asg else None,
                       word_score=rnd.choice([0.0, 1.5, -0.5]), sil_score=rnd.choice([0.0, -0.5]), **kw)
        inp = helpers.case_inputs(c)
        d = sess.decoder(c, inp)
        if i % 2:
            d.set("yshare", 1)
        d.decode_batch(inp["e"], [c["T"]], c["N"])
        got = d.results(0)
        eng, red = d.get("engine"), d.get("redone")
        d.close()
        want = helpers.run_checker(oracle_lib, c, inp)
        ok, why = helpers.hyps_equal(want, got, tol)
        if not ok and ml and len(want) == len(got):
            sf, so = inp["lex"]
            sp = lambda w: tuple(sf[so[w]:so[w + 1]])
            ok = all(abs(a.score - g.score) <= tol and list(a.tokens) == list(g.tokens) and
                     all(x == y or (x >= 0 and y >= 0 and sp(int(x)) == sp(int(y))) for x, y in zip(a.words, g.words))
                     for a, g in zip(want, got))
        s = st[mode]
        s[0] += 1
        s[1] += int(eng == 6)
        s[2] += red
        s[3] += 0 if ok else 1
        if not ok:
            bad.append((mode, {k: c[k] for k in ("dist", "T", "K", "Kt", "thr")}, why))
    return st, bad


@pytest.mark.gpu
def test_logadd_under_asg_and_over_homophones_on_the_lane_engine(gpu_session, oracle_lib):
    st, bad = _logadd_asg_homophone_grid(gpu_session, oracle_lib, 600, 11, [1, 5, 20, 40, 70, 150], 1e-5)
    assert not bad and all(v[1] == v[0] and v[2] <= v[0] // 10 for v in st.values()), (st, bad[:3])


@pytest.mark.gpu
def test_logadd_on_the_lexicon_lane_engine_with_lm_terms(gpu_session, oracle_lib):
    on6, redone, bad = _logadd_lm_lexicon_grid(gpu_session, oracle_lib, 600, 5, [1, 5, 20, 40, 70, 150], 1e-5)
    assert on6 == 600 and redone <= 90 and not bad, (on6, redone, bad[:3])


@pytest.mark.gpu
def test_logadd_on_the_lexicon_lane_engine(gpu_session, oracle_lib):
    on5, bad = _logadd_lexicon_grid(gpu_session, oracle_lib, 600, 3, [1, 5, 20, 40, 70, 150], 1e-5)
    assert on5 == 600 and not bad, (on5, bad[:3])


@pytest.mark.gpu
def test_logadd_on_the_lane_state_engine(gpu_session, oracle_lib):
    """logAdd merges on fltx_slane.h (engine 4) against the oracle @1e-5 (device libm): beams 1 .. 64,
    thresholds 2 .. inf, token beams, silScore, CTC and ASG, `ctc` and `uniform` rows."""
    import itertools
    bad, ran = [], 0
    grid = itertools.product([1, 3, 10, 50, 64], [2.0, 25.0, float("inf")], [None, 5], [12, 29], [1, 30, 120],
                             ["ctc", "uniform"], [0.0, -0.6], ["ctc", "asg"])
    for i, (K, thr, Kt, N, T, dist, sil, crit) in enumerate(grid):
        if i % 5:
            continue
        c = cases.case("la%d" % i, dist=dist, u=500 + i, T=T, N=N, K=K, Kt=Kt, thr=thr, sil_score=sil, log_add=True,
                       crit=crit, trans_seed=(30 + i) if crit == "asg" else None)
        inp = helpers.case_inputs(c)
        want = helpers.run_checker(oracle_lib, c, inp)
        if any(abs(a.score - b.score) < 1e-4 for a, b in zip(want, want[1:])):
            continue  # (near ties: a different libm may order them differently)
        got = gpu_session.run(c, inp)
        assert gpu_session.last_engine == 4
        ok, why = helpers.hyps_equal(want, got, 1e-5)
        ran += 1
        if not ok:
            bad.append(({k: c[k] for k in ("K", "thr", "Kt", "N", "T", "dist", "sil_score", "crit")}, why))
    assert ran > 150 and not bad, (ran, bad[:3])


def _lane_group_grid(session, oracle_lib, every, T_of, emu=False):
    """fltx_mlane.h (lane = LM state with 2 / 4 / 8 groups of 64 lanes) against the oracle: beams at and around the
    group limits (65, 128, 129, 256, 257, 512), small beams forced onto several groups, thresholds 0.5 .. inf, token
    beams, silScore of both signs, CTC and ASG, max-merge and logAdd, every compiled geometry."""
    import itertools
    bad, ran, served = [], 0, 0
    grid = itertools.product([3, 65, 100, 128, 129, 200, 256, 257, 400, 512], [0.5, 25.0, float("inf")], [None, 5],
                             [12, 29], ["ctc", "uniform"], [0.0, -0.6, 0.4], ["ctc", "asg"], [False, True])
    for i, (K, thr, Kt, N, dist, sil, crit, la) in enumerate(grid):
        if i % every:
            continue
        T = T_of(i)
        c = cases.case("mlg%d" % i, dist=dist, u=1500 + i, T=T, N=N, K=K, Kt=Kt, thr=thr, sil_score=sil, log_add=la,
                       crit=crit, trans_seed=(60 + i) if crit == "asg" else None)
        inp = helpers.case_inputs(c)
        want = helpers.run_checker(oracle_lib, c, inp)
        if la and any(abs(a.score - b.score) < 1e-4 for a, b in zip(want, want[1:])):
            continue  # (near ties: a different libm may order them differently)
        if len({h.score for h in want}) != len(want):
            continue
        need = 2 if K <= 128 else (4 if K <= 256 else 8)
        geos = {2: [0, 1, 2, 3, 4, 5, 6], 4: [3, 4, 5, 6], 8: [6]}[need]
        d = session.decoder(c, inp)
        geo = geos[(i // every) % len(geos)]
        n_list = min(Kt or N, N) - (1 if crit == "ctc" and not Kt else 0)
        if n_list <= [28, 30, 70, 28, 30, 66, 30][geo]:  # (list positions the geometry's token waves cover)
            d.set("mlane_geo", geo)
        if K <= 64:
            d.set("lane_groups", 2)
        d.decode_batch(inp["e"], [T], N)
        got = d.results(0)
        served += 1 if (d.get("engine") == 4 and d.get("lane_groups") >= need and d.get("redone") == 0) else 0
        d.close()
        ok, why = helpers.hyps_equal(want, got, (1e-9 if emu else 1e-5) if la else 0.0)
        ran += 1
        if not ok:
            bad.append(({k: c[k] for k in ("K", "thr", "Kt", "N", "T", "dist", "sil_score", "crit", "log_add")}, why))
    return ran, served, bad


@pytest.mark.gpu
def test_edge_configurations_of_the_lane_state_engine_with_lane_groups(gpu_session, oracle_lib):
    ran, served, bad = _lane_group_grid(gpu_session, oracle_lib, 3, lambda i: [1, 2, 23, 90][i % 4])
    assert ran > 400 and served == ran and not bad, (ran, served, bad[:3])


def _four_lane_group_grid(session, oracle_lib, every, T_of):
    """fltx_ylane.h with four groups of 64 lanes (beams 129 .. 256; smaller beams forced onto four groups): n-gram
    word LMs of order 2 .. 4 and label scores without an LM, thresholds 0 .. inf, token beams, lmWeight / wordScore /
    silScore of both signs, `lexspell` and `uniform` rows, against the oracle."""
    import itertools
    bad, ran, served = [], 0, 0
    grid = itertools.product([5, 100, 129, 160, 200, 256], [0.0, 2.0, 25.0, float("inf")], [None, 3, 10],
                             [("ngram", 2, 71), ("ngram", 4, 72), "scores", "zero"], [0.7, 2.0, -0.5], [0.0, 1.5, -2.0],
                             [0.0, -0.7], ["lexspell", "uniform"])
    for i, (K, thr, Kt, lm, lw, ws, sil, dist) in enumerate(grid):
        if i % every:
            continue
        T = T_of(i)
        plain = lm in ("scores", "zero")
        c = cases.case("y4_%d" % i, kind="lexicon", dist=dist, u=1900 + i, T=T, K=K, Kt=Kt, thr=thr, sil_score=sil,
                       word_score=ws, lm_weight=0.0 if lm == "zero" else lw, lexicon=cases.SMALL_LEX,
                       lm="zero" if plain else lm, label_scores=(50 + i % 7) if lm == "scores" else None)
        inp = helpers.case_inputs(c)
        want = helpers.run_checker(oracle_lib, c, inp)
        if len({h.score for h in want}) != len(want):
            continue
        d = session.decoder(c, inp)
        d.set("ylane", 2)
        d.set("ylane_groups", 4)
        if i % (3 * every) == 0:  # token waves with more than max(8, beam) pairs rank their own: the fan-out path
            d.set("ylane_rank_at", 8)
        d.decode_batch(inp["e"], [T], c["N"])
        got = d.results(0)
        srv = d.get("engine") == 6 and d.get("lane_groups") == 4 and d.get("redone") == 0
        served += 1 if srv else 0
        reasons = d.get("fallback_reasons")
        d.close()
        ok, why = helpers.hyps_equal(want, got)
        ran += 1
        if not ok or not srv:
            bad.append((i, why or "left the engine, reasons %#x" % reasons))
    return ran, served, bad


def test_edge_configurations_of_the_lexicon_lane_engine_with_four_lane_groups(gpu_session, oracle_lib):
    ran, served, bad = _four_lane_group_grid(gpu_session, oracle_lib, 11, lambda i: [1, 17, 90, 40][i % 4])
    assert ran > 150 and served == ran and not bad, (ran, served, bad[:8])


def test_long_utterance_stays_on_the_lexicon_lane_engine(gpu_session, oracle_lib):
    """A C4-shaped utterance of 4 000 frames creates more LM states than the memo in LDS numbers (6 144): it takes the
    memo in HBM, sized for its frames, and stays on engine 6 (round 3 handed it to the generic engine half way)."""
    c = dict(cases.BY_NAME["C4_spell_u0"], T=4000, u=17)
    inp = helpers.case_inputs(c)
    d = gpu_session.decoder(c, inp)
    d.decode_batch(inp["e"], [c["T"]], c["N"])
    got = d.results(0)
    info = (d.get("engine"), d.get("redone"), d.get("yshare"), d.get("ymemo_slots"))
    d.close()
    assert info[0] == 6 and info[1] == 0 and info[2] == 1 and info[3] > 8192, info
    want = helpers.run_checker(oracle_lib, c, inp)
    ok, why = helpers.hyps_equal(want, got)
    assert ok, why


def _asg_lexicon_grid(session, oracle_lib, every, T_of):
    """LexiconDecoder with the ASG criterion on fltx_ylane.h (no blank; transitions enter score and emitting-model
    score from the second frame on, LexiconDecoder.cpp:69-72,172-175): beams over one, two and four lane groups,
    ZeroLM / label scores / n-gram LMs, thresholds, token beams, silScore, a lexicon without doubled letters (what
    replabels guarantee: helpers.lexicon), against the oracle."""
    import itertools
    bad, ran, served = [], 0, 0
    grid = itertools.product([1, 3, 10, 64, 70, 128, 150, 256], [0.0, 2.0, 25.0, float("inf")], [None, 5, 10],
                             ["zero", ("ngram", 3, 91), ("ngram", 4, 92), "scores"], [0.0, -0.6, 0.4], [0.7, -1.0],
                             [0, 1], ["lexspell", "uniform"])
    for i, (K, thr, Kt, lm, sil, ws, share, dist) in enumerate(grid):
        if i % every:
            continue
        T = T_of(i)
        plain = lm in ("scores", "zero")
        c = cases.case("yasg%d" % i, kind="lexicon", dist=dist, u=2500 + i, T=T, K=K, Kt=Kt, thr=thr, sil_score=sil,
                       word_score=ws, lm_weight=0.0 if lm == "zero" else 1.3, lexicon=cases.NODUP_LEX, crit="asg",
                       trans_seed=70 + i, lm="zero" if plain else lm, label_scores=(50 + i % 7) if lm == "scores" else None)
        inp = helpers.case_inputs(c)
        want = helpers.run_checker(oracle_lib, c, inp)
        if len({h.score for h in want}) != len(want):
            continue
        d = session.decoder(c, inp)
        d.set("yshare", share if K <= 128 else -1)
        d.decode_batch(inp["e"], [T], c["N"])
        got = d.results(0)
        srv = d.get("engine") == 6 and d.get("redone") == 0
        reasons = d.get("fallback_reasons")
        d.close()
        served += 1 if srv else 0
        ok, why = helpers.hyps_equal(want, got)
        ran += 1
        if not ok or not srv:
            bad.append((i, why or "left the engine, reasons %#x" % reasons))
    return ran, served, bad


def test_asg_on_the_lexicon_lane_engine(gpu_session, oracle_lib):
    ran, served, bad = _asg_lexicon_grid(gpu_session, oracle_lib, 7, lambda i: [1, 17, 90, 40][i % 4])
    assert ran > 300 and served == ran and not bad, (ran, served, bad[:8])


def _word_piece_grid(session, oracle_lib, every, T_of, emu=False):
    """LexiconFreeDecoder + ZeroLM over word-piece sized token sets on fltx_wlane.h (token beam <= 64, beam <= 64):
    token-set sizes around the front end's chunk of 1 024, beams, token beams, thresholds, silScore, both criteria
    (ASG with transitions: they enter the emitting-model score only, LexiconFreeDecoder.cpp:58-64), against the oracle."""
    import itertools
    bad, ran, served = [], 0, 0
    grid = itertools.product([65, 100, 257, 1024, 1025, 3000, 8192], [1, 7, 50, 64], [1, 5, 30, 50, 64],
                             [0.0, 3.0, 25.0, float("inf")], [0.0, -0.7, 0.4], ["ctc", "asg"], ["ctc", "uniform"])
    for i, (N, K, Kt, thr, sil, crit, dist) in enumerate(grid):
        if i % every:
            continue
        T = T_of(i)
        if emu and N > 1100:  # (host threads: keep the emulator's share of the grid small)
            T = min(T, 6)
        c = cases.case("wp%d" % i, dist=dist, T=T, N=N, K=K, Kt=Kt, u=7000 + i, crit=crit, sil_score=sil, thr=thr,
                       trans_seed=(90 + i % 5) if (crit == "asg" and N <= 1100) else None)
        if crit == "asg" and N > 1100:
            continue  # (an N x N transition table per case: covered at the smaller sizes)
        inp = helpers.case_inputs(c)
        want = helpers.run_checker(oracle_lib, c, inp)
        if len({h.score for h in want}) != len(want):
            continue
        d = session.decoder(c, inp)
        d.decode_batch(inp["e"], [T], N)
        got = d.results(0)
        srv = d.get("engine") == 4 and d.get("wlane") == 1 and d.get("redone") == 0
        d.close()
        served += 1 if srv else 0
        ok, why = helpers.hyps_equal(want, got)
        ran += 1
        if not ok or not srv:
            bad.append((i, (N, K, Kt, thr, sil, crit, dist, T), why or "left the engine"))
    return ran, served, bad


def test_word_piece_token_sets_on_the_lane_state_engine(gpu_session, oracle_lib):
    ran, served, bad = _word_piece_grid(gpu_session, oracle_lib, 11, lambda i: [1, 23, 60, 9][i % 4])
    assert ran > 150 and served == ran and not bad, (ran, served, bad[:6])


@pytest.mark.parametrize("N,Kt", [(1024, 50), (8192, 50)])
def test_word_piece_long_utterance(gpu_session, oracle_lib, N, Kt):
    """T = 400 frames of a 1 024 / 8 192 token set, beam 50: re-entries of LM states, the front end's window following
    the rows, the back-trace reading emissions where it needs them."""
    c = cases.case("wp_long%d" % N, dist="ctc", T=400, N=N, K=50, Kt=Kt, u=4242)
    inp = helpers.case_inputs(c)
    want = helpers.run_checker(oracle_lib, c, inp)
    d = gpu_session.decoder(c, inp)
    d.decode_batch(inp["e"], [c["T"]], N)
    got = d.results(0)
    info = (d.get("engine"), d.get("wlane"), d.get("redone"))
    d.close()
    assert info == (4, 1, 0), info
    if len({h.score for h in want}) != len(want):
        pytest.skip("equal scores in the n-best")
    ok, why = helpers.hyps_equal(want, got)
    assert ok, why


def test_word_piece_row_without_a_defined_token_beam(gpu_session, oracle_lib):
    """A row of 300 equal values has no defined token beam (the reference's partial_sort keeps whichever 30 it meets
    first): the front-end kernel flags the row, fltx_wlane.h hands that utterance -- and only that one -- to the
    generic engine (`redone` 1), whose short-list breaks the tie towards the lower token.  Utterance 1 therefore equals
    what the generic engine returns for the whole batch with fltx_wlane.h switched off; the other utterances equal the
    oracle and keep their fltx_wlane.h results (packed records, tokens beyond a byte)."""
    N, T, B = 300, 40, 5
    c = cases.case("wp_ties", dist="ctc", T=T, N=N, K=20, Kt=30, u=515)
    e = synth.batch("ctc", B, T, N)
    e[1, 9, :] = -3.0
    d = gpu_session.decoder(c, dict(tr=None))
    d.decode_batch(e, [T] * B, N)
    info = (d.get("engine"), d.get("wlane"), d.get("redone"))
    assert info == (4, 1, 1), info
    for b in (0, 2, 3, 4):
        want = helpers.run_checker(oracle_lib, c, dict(e=e[b], tr=None, lex=None))
        if len({h.score for h in want}) != len(want):
            continue
        ok, why = helpers.hyps_equal(want, d.results(b))
        assert ok, "utterance %d: %s" % (b, why)
    got1 = d.results(1)
    assert len(got1) > 0
    g = gpu_session.decoder(c, dict(tr=None))
    g.set("wlane", 0)
    g.decode_batch(e, [T] * B, N)
    assert g.get("engine") != 4 and g.get("wlane") == 0
    for b in range(B):
        ok, why = helpers.hyps_equal(g.results(b), d.results(b))
        assert ok, "utterance %d vs the generic engine: %s" % (b, why)
    d.close()
    g.close()


@pytest.mark.gpu
def test_deferred_status_look(gpu_session, golden):
    import test_emu_logic
    test_emu_logic._deferred_look(gpu_session, golden, 40)
