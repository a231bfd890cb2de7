"""Input classes real callers bring that the synthetic cases do not have (round-4 review, weak #1):
  (i)   sparse -inf emissions (masked vocabulary entries),
  (ii)  token layouts other than sil = 0, blank = N - 1 (torchaudio: blank = 0, sil elsewhere; sil = N - 1; ASG),
  (iii) log-softmax rows (arbitrary floats instead of the 24-bit grid of SURVEY Appendix A) at the C2 size.
Every decode is compared with the oracle on the same inputs, bit for bit; the lane engines (4 / 5 / 6) are the
ones under test, `redone` says how many utterances they handed to the generic engine."""
import numpy as np
import pytest

import cases
import helpers
from oracle import orclib
from text_amd import _capi, synth


def _relabel(N, sil, blank):
    """old index -> new index such that old sil (0) -> sil, old blank (N - 1) -> blank (blank < 0: ASG, no blank)"""
    pi = [-1] * N
    pi[0] = sil
    if blank >= 0:
        pi[N - 1] = blank
    free = [i for i in range(N) if i != sil and i != blank]
    for old in range(N):
        if pi[old] < 0:
            pi[old] = free.pop(0)
    return np.array(pi, dtype=np.int32)


def _oracle(orc, c, e, tr, sil, blank, lex, scores, lm=None):
    opt = orclib.make_options(c["K"], c["Kt"], c["thr"], c["lm_weight"], c["word_score"], c["unk_score"],
                              c["sil_score"], c["log_add"], c["crit"])
    own = lm is None
    lm = lm or orc.lm_zero_create()
    trie = None
    if lex is None:
        dec = orc.lexfree(opt, lm, sil, blank, tr)
    else:
        sf, so = lex
        W = len(so) - 1
        trie = orc.build_trie(c["N"], sil, sf, so, np.arange(W, dtype=np.int32), scores, smear=1)
        dec = orc.lexicon(opt, trie, lm, sil, blank, W, tr, False)
    out = [orc.decode(dec, x, x.shape[0], c["N"]) for x in e]
    orc.decoder_destroy(dec)
    if trie is not None:
        orc.trie_destroy(trie)
    if own:
        orc.lm_destroy(lm)
    return out


def _device(sess, c, e, tr, sil, blank, lex, scores, lm=None, sets=None):
    opt = _capi.make_options(c["K"], c["Kt"], c["thr"], c["lm_weight"], c["word_score"], c["unk_score"],
                             c["sil_score"], c["log_add"], c["crit"])
    lm = lm or sess.zero
    ht = None
    if lex is None:
        d = _capi.BatchDecoder(sess.ctx, _capi.LEXFREE, opt, lm, sil, blank, transitions=tr)
    else:
        sf, so = lex
        W = len(so) - 1
        ht = _capi.HostTrie(c["N"], sil, lib=sess.lib)
        ht.insert_many(sf, so, np.arange(W, dtype=np.int32), scores)
        ht.smear(1)
        d = _capi.BatchDecoder(sess.ctx, _capi.LEXICON, opt, lm, sil, blank, unk=W, trie=ht.upload(sess.ctx),
                               transitions=tr)
    for k, v in (sets or {}).items():
        d.set(k, v)
    T = [x.shape[0] for x in e]
    d.decode_batch(np.concatenate([x.reshape(-1) for x in e]), T, c["N"])
    out = [d.results(b) for b in range(len(e))]
    info = dict(engine=d.get("engine"), redone=d.get("redone"))
    d.close()
    return out, info


def _compare(want, got, what):
    bad = []
    for b, (w, g) in enumerate(zip(want, got)):
        if len({h.score for h in w}) != len(w):
            continue  # equal scores in the n-best: the reference's own order is undefined
        ok, why = helpers.hyps_equal(w, g)
        if not ok:
            bad.append((what, b, why))
    return bad


LAYOUTS = [("blank0_sil4", 4, 0, "ctc"), ("sil_last_blank5", 28, 5, "ctc"), ("asg_sil3", 3, -1, "asg")]


def _layout_grid(sess, orc, B, T, engines, sets=None):
    """lexicon-free (engine 4), lexicon + ZeroLM (5), lexicon with label scores (6) under three token layouts"""
    bad, seen = [], []
    N = 29
    for name, sil, blank, crit in LAYOUTS:
        pi = _relabel(N, sil, blank)
        inv = np.argsort(pi)
        tr = synth.floats(77, N * N, 0.0, 1.0) if crit == "asg" else None
        for kind in engines:
            lexkey = cases.NODUP_LEX if crit == "asg" else cases.SMALL_LEX
            lex = helpers.lexicon(*lexkey) if kind != "lexfree" else None
            c = cases.case("lay_%s_%s" % (name, kind), kind="lexfree" if lex is None else "lexicon", T=T, N=N,
                           K=24 if kind == "lexfree" else 16, Kt=N if kind == "lexfree" else 10, crit=crit,
                           lm_weight=1.5 if kind == "lex_scores" else 0.0, word_score=0.5 if lex is not None else 0.0,
                           sil_score=-0.2)
            dist = "ctc" if lex is None else "lexspell"
            e0 = [synth.emissions(dist, 400 + b, T, N, lexicon=lex) for b in range(B)]
            e = [np.ascontiguousarray(x[:, inv]) for x in e0]  # e'[:, pi[i]] = e[:, i]
            lx = (pi[lex[0]].astype(np.int32), lex[1]) if lex is not None else None
            W = len(lex[1]) - 1 if lex is not None else 0
            scores = synth.floats(55, W, -5.0, 0.0) if kind == "lex_scores" else np.zeros(W, dtype=np.float32)
            want = _oracle(orc, c, e, tr, sil, blank, lx, scores)
            got, info = _device(sess, c, e, tr, sil, blank, lx, scores, sets=sets)
            seen.append((name, kind, info["engine"], info["redone"]))
            bad += _compare(want, got, (name, kind))
    return bad, seen


def _sparse_inf(sess, orc, B, T, engines, sets=None):
    """a fifth of the entries masked to -inf, never a frame's best (a row keeps finite candidates)"""
    bad, seen = [], []
    N = 29
    rng = np.random.RandomState(1234)
    for kind in engines:
        lex = helpers.lexicon(*cases.SMALL_LEX) if kind != "lexfree" else None
        c = cases.case("inf_%s" % kind, kind="lexfree" if lex is None else "lexicon", T=T, N=N,
                       K=24 if kind == "lexfree" else 16, Kt=N if kind == "lexfree" else 12,
                       lm_weight=1.5 if kind == "lex_scores" else 0.0, word_score=0.5 if lex is not None else 0.0)
        dist = "ctc" if lex is None else "lexspell"
        e = []
        for b in range(B):
            x = synth.emissions(dist, 500 + b, T, N, lexicon=lex).copy()
            mask = rng.rand(T, N) < 0.2
            mask[np.arange(T), x.argmax(axis=1)] = False
            mask[:, N - 1] = False  # (the blank stays: a CTC lexicon path needs it between doubled letters)
            x[mask] = -np.inf
            e.append(x)
        W = len(lex[1]) - 1 if lex is not None else 0
        scores = synth.floats(56, W, -5.0, 0.0) if kind == "lex_scores" else np.zeros(W, dtype=np.float32)
        want = _oracle(orc, c, e, None, 0, N - 1, lex, scores)
        got, info = _device(sess, c, e, None, 0, N - 1, lex, scores, sets=sets)
        seen.append((kind, info["engine"], info["redone"]))
        bad += _compare(want, got, kind)
    return bad, seen


ENGINES = ["lexfree", "lex_zero", "lex_scores"]


def test_emulated_token_layouts(emu_session, oracle_lib):
    bad, seen = _layout_grid(emu_session, oracle_lib, B=1, T=24, engines=ENGINES)
    assert not bad, bad[:3]
    assert {s[2] for s in seen} == {4, 5, 6}, seen


def test_emulated_sparse_minus_infinity(emu_session, oracle_lib):
    bad, seen = _sparse_inf(emu_session, oracle_lib, B=2, T=24, engines=ENGINES)
    assert not bad, bad[:3]
    assert {s[1] for s in seen} == {4, 5, 6}, seen


@pytest.mark.gpu
def test_token_layouts_on_the_lane_engines(gpu_session, oracle_lib):
    bad, seen = _layout_grid(gpu_session, oracle_lib, B=6, T=120, engines=ENGINES)
    assert not bad, bad[:3]
    assert {s[2] for s in seen} == {4, 5, 6}, seen
    assert all(s[3] == 0 for s in seen), seen  # nothing handed to the generic engine


@pytest.mark.gpu
def test_sparse_minus_infinity_on_the_lane_engines(gpu_session, oracle_lib):
    bad, seen = _sparse_inf(gpu_session, oracle_lib, B=8, T=150, engines=ENGINES)
    assert not bad, bad[:3]
    assert {s[1] for s in seen} == {4, 5, 6}, seen


@pytest.mark.gpu
def test_log_softmax_inputs_at_the_c2_size(gpu_session, oracle_lib):
    """T = 1000, N = 29, beam 50 on rows that are a log-softmax of random logits (peaky, like a trained model's):
    arbitrary float32 values instead of the synthetic 24-bit grid; blank = 0 as torchaudio lays its tokens out."""
    N, T, B = 29, 1000, 6
    rng = np.random.RandomState(99)
    e = []
    for b in range(B):
        z = rng.randn(T, N).astype(np.float32) * 3.0
        z[np.arange(T), rng.randint(0, N, T)] += 6.0
        z[rng.rand(T) < 0.5, 0] += 8.0  # blank-dominated frames
        z = z - z.max(axis=1, keepdims=True)
        e.append((z - np.log(np.exp(z).sum(axis=1, keepdims=True))).astype(np.float32))
    c = cases.case("lsm_c2", T=T, N=N, K=50)
    want = _oracle(oracle_lib, c, e, None, 4, 0, None, None)
    got, info = _device(gpu_session, c, e, None, 4, 0, None, None)
    assert info["engine"] == 4 and info["redone"] == 0, info
    bad = _compare(want, got, "log-softmax")
    assert not bad, bad[:3]
