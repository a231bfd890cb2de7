"""Several words per spelling (the reference's Trie keeps up to six labels on a node, Trie.h:19, and
LexiconDecoder.cpp:113-142 emits one candidate per label; its own test lexicon has 169 such spellings) on the
lexicon lane engine (fltx_ylane.h, LMK bit 2): the committed vectors of the compiled reference, the reference's test
lexicon at the lane engines' beams, the fall-back when a frame has more further words than the word wave lists, and
random configurations against the oracle."""
import gzip
import os
import random

import numpy as np
import pytest

import cases
import helpers

ML_SMALL = ["ml_word_t60_k16", "ml_word_uni_t50_k48", "ml_word_asg_t40_k24"]
ML_ALL = ML_SMALL + ["ml_word_t80_k100", "ml_word_asg_t80_k200"]
GROUPS = {"ml_word_t60_k16": 1, "ml_word_uni_t50_k48": 1, "ml_word_asg_t40_k24": 1, "ml_word_t80_k100": 2,
          "ml_word_asg_t80_k200": 0}  # (beams beyond 128 with such a lexicon: the generic engine, FLTX_WHY_BEAM)


def _run(sess, c, inp=None, sets=None):
    inp = inp or helpers.case_inputs(c)
    d = sess.decoder(c, inp)
    for k, v in (sets or {}).items():
        d.set(k, v)
    d.decode_batch(inp["e"], [c["T"]], c["N"])
    out = d.results(0)
    info = dict(engine=d.get("engine"), groups=d.get("lane_groups"), redone=d.get("redone"), why=d.get("why_not_lane"),
                fallback=d.get("fallback_reasons"))
    d.close()
    return out, info


def _golden_on_the_lane_engine(sess, golden, name):
    from text_amd import _capi
    c = cases.BY_NAME[name]
    got, info = _run(sess, c)
    if GROUPS[name]:
        assert info["engine"] == 6 and info["groups"] == GROUPS[name] and info["redone"] == 0, info
    else:
        assert info["engine"] != 6 and info["why"] & _capi.FLTX_WHY_BEAM, info
    ok, why = helpers.check_against_golden(got, golden[name])
    assert ok, why
    if not GROUPS[name]:
        return
    # ... and the generic engine, which loops over the labels as the reference does
    gen, info = _run(sess, c, sets={"ylane": 0})
    assert info["engine"] != 6
    ok, why = helpers.hyps_equal(gen, got)
    assert ok, why


@pytest.mark.parametrize("name", ML_SMALL)
def test_emulated_spellings_with_several_words(emu_session, golden, name):
    _golden_on_the_lane_engine(emu_session, golden, name)


def _six_word_case(K, T=30):
    return cases.case("ml6_k%d" % K, kind="lexicon", dist="uniform", T=T, K=K, lexicon=(300, 4242, "multi6"), u=77,
                      lm=("ngram", 3, 46), lm_weight=1.0, word_score=1.0)


def _six_words(sess, oracle_lib):
    # six words on every spelling; beam 12: at most 60 further words a frame -- the lane engine keeps the utterance
    c = _six_word_case(12)
    inp = helpers.case_inputs(c)
    got, info = _run(sess, c, inp)
    assert info["engine"] == 6 and info["redone"] == 0, info
    ok, why = helpers.hyps_equal(helpers.run_checker(oracle_lib, c, inp), got)
    assert ok, why
    # beam 64 over uniform emissions: more than the 128 the word wave lists (two extra slots x 64 threads) -- the
    # utterance leaves for the generic engine (fallback reason 8) and the result is still the reference's
    c = _six_word_case(64)
    c["lexicon"] = (300, 4242, "short6")
    inp = helpers.case_inputs(c)
    got, info = _run(sess, c, inp)
    assert info["redone"] == 1 and info["fallback"] & (1 << 8), info
    ok, why = helpers.hyps_equal(helpers.run_checker(oracle_lib, c, inp), got)
    assert ok, why


def test_emulated_six_words_per_spelling_and_the_overflow(emu_session, oracle_lib):
    _six_words(emu_session, oracle_lib)


def _zero_lm_keeps_off(sess):
    """under ZeroLM the words of a spelling tie in one LM state: the reference keeps whichever its sort leaves first;
    such a lexicon stays on the generic engine, which says why"""
    from text_amd import _capi
    c = cases.case("ml_zero", kind="lexicon", dist="lexspell", T=20, K=8, lexicon=cases.MULTI_LEX, u=3)
    got, info = _run(sess, c)
    assert info["engine"] != 6 and info["why"] & _capi.FLTX_WHY_TRIE_SHAPE, info


def test_emulated_zero_lm_stays_on_the_generic_engine(emu_session):
    _zero_lm_keeps_off(emu_session)


def _random_configurations(sess, oracle_lib, n, seed, frames):
    """random (beam, frames, criterion, token beam, threshold, LM order, weights) combinations over the synthetic
    lexicons with one to three words per spelling, against the oracle (tools/r05/multilabel_soak.py runs the same
    generator longer).  Equal-score hypotheses that differ in which word of a spelling they hold are the one accepted
    difference: two orders of the same two words in one history tie once the n-gram context forgets them."""
    rnd = random.Random(seed)
    lane = small = 0
    for i in range(n):
        asg = rnd.random() < 0.4
        big = rnd.random() < 0.5
        lexi = (cases.MULTI_NODUP_LEX_3K if big else cases.MULTI_NODUP_LEX) if asg else \
            (cases.MULTI_LEX_3K if big else cases.MULTI_LEX)
        c = cases.case("mlr%d" % i, kind="lexicon", dist=rnd.choice(["lexspell", "lexspell", "uniform"]),
                       T=rnd.choice(frames), K=rnd.choice([3, 10, 24, 50, 64, 65, 100, 128, 129, 180, 256]),
                       Kt=rnd.choice([29, 29, 10, 5]), thr=rnd.choice([25.0, 25.0, 8.0, 100.0]), lexicon=lexi, u=2000 + i,
                       crit="asg" if asg else "ctc", trans_seed=(50 + i % 7) if asg else None,
                       lm=("ngram", rnd.choice([2, 3, 4]), 60 + i % 5), lm_weight=rnd.choice([0.5, 1.3, 2.0]),
                       word_score=rnd.choice([0.0, 0.7, 2.0]), sil_score=rnd.choice([0.0, -0.5, -1.0]))
        inp = helpers.case_inputs(c)
        got, info = _run(sess, c, inp)
        lane += int(info["engine"] == 6 and info["redone"] == 0)
        small += int(c["K"] <= 128)
        assert (info["engine"] == 6) == (c["K"] <= 128), (i, info)
        want = helpers.run_checker(oracle_lib, c, inp)
        ok, why = helpers.hyps_equal(want, got)
        if not ok:
            sf, so = inp["lex"]
            sp = lambda w: tuple(sf[so[w]:so[w + 1]])
            assert len(want) == len(got), (i, why)
            for a, g in zip(want, got):
                assert a.score == g.score and list(a.tokens) == list(g.tokens), (i, why)
                assert all(x == y or (x >= 0 and y >= 0 and sp(int(x)) == sp(int(y))) for x, y in zip(a.words, g.words)), (i, why)
    assert lane >= small - max(2, n // 60), (lane, small)


def test_emulated_random_configurations_over_lexicons_with_homophones(emu_session, oracle_lib):
    _random_configurations(emu_session, oracle_lib, 60, 78, [1, 7, 24, 40])


# ---- the product path -------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("name", ML_ALL)
def test_spellings_with_several_words_on_the_lane_engine(gpu_session, golden, name):
    _golden_on_the_lane_engine(gpu_session, golden, name)


@pytest.mark.gpu
def test_six_words_per_spelling_and_the_overflow(gpu_session, oracle_lib):
    _six_words(gpu_session, oracle_lib)


@pytest.mark.gpu
def test_zero_lm_stays_on_the_generic_engine(gpu_session):
    _zero_lm_keeps_off(gpu_session)


def _decodertest_inputs(tmp_path):
    from golden.make_golden import parse_lexicon_dump
    d = os.path.join(helpers.GOLDEN_DIR, "decodertest")
    rd = lambda n: gzip.open(os.path.join(d, n + ".gz"), "rb").read()
    lex = parse_lexicon_dump(rd("lexicon_dump.txt").decode())
    TN = np.frombuffer(rd("TN.bin"), dtype=np.int32)
    em = np.frombuffer(rd("emission.bin"), dtype=np.float32).copy()
    tr = np.frombuffer(rd("transition.bin"), dtype=np.float32).copy()
    arpa = tmp_path / "lm.arpa"
    arpa.write_bytes(rd("lm.arpa"))
    return lex, int(TN[0]), int(TN[1]), em, tr, str(arpa)


@pytest.mark.gpu
def test_reference_test_lexicon_at_the_lane_beams(gpu_session, oracle_lib, tmp_path):
    """DecoderTest.cpp:57-195's inputs (26k-word lexicon, 164 spellings with two words and 5 with three; 3-gram ARPA;
    ASG) at beams of one and two lane groups: every utterance stays on fltx_ylane.h (beam 256: the generic engine --
    the word wave of the four-group geometry has no registers for the further words), and the n-best is the
    oracle's -- up to the choice among words of one spelling whose LM scores are equal (unseen words share the
    <unk> probability), which the reference leaves to the order its sort happens to produce (oracle and compiled
    reference differ from each other there)."""
    from oracle import orclib
    from text_amd import _capi
    lex, T, N, em, tr, arpa = _decodertest_inputs(tmp_path)
    spell = {}
    for wi, w, sp in lex["entries"]:
        spell.setdefault(wi, tuple(sp))
    sess = gpu_session
    lm = _capi.ArpaLM(arpa, lex["words"])
    ht = _capi.HostTrie(lex["ntok"], lex["sil"])
    cache = {}
    for wi, w, sp in lex["entries"]:
        if wi not in cache:
            cache[wi] = lm.score_sequence([wi], False)[0][0]
        ht.insert(sp, wi, cache[wi])
    ht.smear(1)
    trie = ht.upload(sess.ctx)
    clm = oracle_lib.lm_arpa_create(arpa.encode(), "\n".join(lex["words"]).encode())
    ctrie = oracle_lib.trie_create(lex["ntok"], lex["sil"])
    for wi, w, sp in lex["entries"]:
        a = np.array(sp, dtype=np.int32)
        oracle_lib.trie_insert(ctrie, orclib._ip(a), len(sp), wi, cache[wi])
    oracle_lib.trie_smear(ctrie, 1)
    for K, groups in ((50, 1), (128, 2), (256, 0)):
        for crit, trv, blank in (("asg", tr, -1), ("ctc", None, N - 1)):
            opt = _capi.make_options(K, 25000, 100.0, 2.0, 2.0, -float("inf"), -1.0, False, crit)
            dec = _capi.BatchDecoder(sess.ctx, _capi.LEXICON, opt, lm, lex["sil"], blank, unk=lex["unk"], trie=trie,
                                     transitions=trv, is_lm_token=False)
            B = 3
            dec.decode_batch(np.tile(em, B), np.full(B, T, dtype=np.int32), N)
            assert (dec.get("engine") == 6 and dec.get("lane_groups") == groups and dec.get("redone") == 0) if groups else \
                dec.get("engine") != 6, \
                (K, crit, dec.get("engine"), dec.get("lane_groups"), dec.get("redone"), dec.get("fallback_reasons"))
            copt = orclib.make_options(K, 25000, 100.0, 2.0, 2.0, -float("inf"), -1.0, False, crit)
            cdec = oracle_lib.lexicon(copt, ctrie, clm, lex["sil"], blank, lex["unk"], trv, False)
            want = oracle_lib.decode(cdec, em, T, N)
            oracle_lib.decoder_destroy(cdec)
            for b in range(B):
                got = dec.results(b)
                assert len(got) == len(want), (K, crit, len(got), len(want))
                for i, (a, g) in enumerate(zip(want, got)):
                    assert a.score == g.score and list(a.tokens) == list(g.tokens), (K, crit, b, i)
                    assert len(a.words) == len(g.words)
                    for x, y in zip(a.words, g.words):
                        assert x == y or (x >= 0 and y >= 0 and spell[int(x)] == spell[int(y)]), (K, crit, b, i, x, y)
            dec.close()


@pytest.mark.gpu
def test_random_configurations_over_lexicons_with_homophones(gpu_session, oracle_lib):
    _random_configurations(gpu_session, oracle_lib, 300, 77, [1, 7, 40, 80, 150])


@pytest.mark.gpu
def test_ragged_batch_larger_than_the_card_with_a_deferred_look(gpu_session, oracle_lib):
    """600 utterances of 0 .. 90 frames over the homophone lexicon in one call with `defer_check` (the call returns with
    its kernels queued): more workgroups than CUs, empty utterances among them; 60 sampled n-bests against the oracle."""
    from text_amd import synth
    c = dict(cases.BY_NAME["ml_word_t80_k100"])
    inp = helpers.case_inputs(c)
    rnd = random.Random(5)
    B = 600
    Ts = [rnd.choice([0, 1, 17, 40, 64, 90]) for _ in range(B)]
    embs = [synth.emissions("lexspell", 5000 + b, T, c["N"], lexicon=inp["lex"]) for b, T in enumerate(Ts)]
    d = gpu_session.decoder(c, inp)
    d.set("defer_check", 1)
    d.decode_batch(np.concatenate([e.reshape(-1) for e in embs]), Ts, c["N"])
    assert d.get("engine") == 6 and d.get("lane_groups") == 2
    for b in rnd.sample(range(B), 60):
        c1 = dict(c)
        c1["T"] = Ts[b]
        want = helpers.run_checker(oracle_lib, c1, dict(inp, e=embs[b]))
        ok, why = helpers.hyps_equal(want, d.results(b))
        assert ok, (b, Ts[b], why)
    assert d.get("redone") == 0
    d.close()
