"""The reference's Python test-suite for this path
(bindings/python/test/test_decoder.py:92-252 DecoderTestCase, :385-470
DecoderPickleTestCase, test_import.py) re-run against the pybind11 module of
this repo through the `flashlight.lib.text` compat package."""
import gzip
import math
import os
import pickle
import sys

import numpy as np
import pytest

import helpers

COMPAT = os.path.join(helpers.ROOT, "text_amd", "compat")
if COMPAT not in sys.path:
    sys.path.insert(0, COMPAT)


def test_import_and_host_side_objects():
    from flashlight.lib.text.decoder import (CriterionType, LexiconDecoderOptions, LexiconFreeDecoderOptions,
                                             SmearingMode, Trie, ZeroLM)
    from flashlight.lib.text.dictionary import Dictionary
    opts = LexiconDecoderOptions(beam_size=3, beam_size_token=5, beam_threshold=10.0, lm_weight=7.0,
                                 word_score=3.0, unk_score=2.0, sil_score=5.0, log_add=True,
                                 criterion_type=CriterionType.CTC)
    o2 = pickle.loads(pickle.dumps(opts))
    for f in ("beam_size", "beam_size_token", "beam_threshold", "lm_weight", "word_score", "unk_score",
              "sil_score", "log_add", "criterion_type"):
        assert getattr(o2, f) == getattr(opts, f)
    fo = LexiconFreeDecoderOptions(beam_size=3, beam_size_token=5, beam_threshold=10.0, lm_weight=7.0,
                                   sil_score=5.0, log_add=True, criterion_type=CriterionType.CTC)
    f2 = pickle.loads(pickle.dumps(fo))
    assert (f2.beam_size, f2.log_add, f2.criterion_type) == (3, True, CriterionType.CTC)
    trie = Trie(5, 0)
    trie.insert([1, 2], 7, -1.0)
    trie.insert([1, 3], 8, -0.25)
    trie.smear(SmearingMode.MAX)
    assert trie.search([1]).max_score == -0.25 and trie.search([4]) is None
    with pytest.raises(IndexError):
        trie.insert([9], 0, 0.0)
    lm = ZeroLM()
    s = lm.start(False)
    s1, sc = lm.score(s, 3)
    assert sc == 0.0 and s1.compare(lm.score(s, 3)[0]) == 0  # child() is memoised (lm/LM.h:24-34)
    d = Dictionary(["a", "b"])
    assert d.get_index("b") == 1 and d.index_size() == 2


@pytest.mark.gpu
def test_decoder_pickle(gpu_session):
    """DecoderPickleTestCase (test_decoder.py:385-470)."""
    from flashlight.lib.text.decoder import CriterionType, LexiconFreeDecoder, LexiconFreeDecoderOptions, ZeroLM
    opts = LexiconFreeDecoderOptions(beam_size=3, beam_size_token=5, beam_threshold=10.0, lm_weight=7.0,
                                     sil_score=5.0, log_add=True, criterion_type=CriterionType.CTC)
    dec = LexiconFreeDecoder(options=opts, lm=ZeroLM(), sil_token_idx=4, blank_token_idx=22,
                             transitions=[1.0, 4.0, 5.0, 9.0])
    d2 = pickle.loads(pickle.dumps(dec))
    assert d2.get_sil_idx() == 4 and d2.get_blank_idx() == 22
    assert d2.get_transitions() == [1.0, 4.0, 5.0, 9.0]
    assert d2.get_options().lm_weight == 7.0


@pytest.mark.gpu
def test_decoder_test_case(gpu_session, tmp_path):
    """DecoderTestCase (test_decoder.py:92-252): KenLM scores, trie smearing,
    LexiconDecoder on the fixture, same assertions (places=4 / 3)."""
    from flashlight.lib.text.decoder import (CriterionType, KenLM, LexiconDecoder, LexiconDecoderOptions,
                                             SmearingMode, Trie)
    from flashlight.lib.text.dictionary import Dictionary, create_word_dict, load_words
    from flashlight.lib.text.dictionary import tkn_to_idx
    d = os.path.join(helpers.GOLDEN_DIR, "decodertest")
    for name in ("words.lst", "letters.lst", "lm.arpa", "TN.bin", "emission.bin", "transition.bin"):
        (tmp_path / name).write_bytes(gzip.open(os.path.join(d, name + ".gz"), "rb").read())
    T, N = np.fromfile(tmp_path / "TN.bin", dtype=np.int32)
    emissions = np.fromfile(tmp_path / "emission.bin", dtype=np.float32)
    transitions = np.fromfile(tmp_path / "transition.bin", dtype=np.float32)
    lexicon = load_words(str(tmp_path / "words.lst"))
    word_dict = create_word_dict(lexicon)
    token_dict = Dictionary(str(tmp_path / "letters.lst"))
    token_dict.add_entry("<1>")
    lm = KenLM(str(tmp_path / "lm.arpa"), word_dict)

    sentence = ["the", "cat", "sat", "on", "the", "mat"]
    lm_state = lm.start(False)
    total = 0
    for word, target in zip(sentence, [-1.05971, -4.19448, -3.33383, -2.76726, -1.16237, -4.64589]):
        lm_state, sc = lm.score(lm_state, word_dict.get_index(word))
        assert abs(sc - target) < 1e-4
        total += sc
    lm_state, sc = lm.finish(lm_state)
    assert abs(total + sc - (-19.5123)) < 1e-4

    sil_idx = token_dict.get_index("|")
    unk_idx = word_dict.get_index("<unk>")
    trie = Trie(token_dict.index_size(), sil_idx)
    start_state = lm.start(False)
    for word, spellings in lexicon.items():
        usr_idx = word_dict.get_index(word)
        _, score = lm.score(start_state, usr_idx)
        for spelling in spellings:
            trie.insert(tkn_to_idx(spelling, token_dict, 1), usr_idx, score)
    trie.smear(SmearingMode.MAX)
    for word, target in zip(sentence, [-1.05971, -2.87742, -2.64553, -3.05081, -1.05971, -3.08968]):
        node = trie.search([token_dict.get_index(c) for c in word])
        assert abs(node.max_score - target) < 1e-4

    opts = LexiconDecoderOptions(beam_size=2500, beam_size_token=25000, beam_threshold=100.0, lm_weight=2.0,
                                 word_score=2.0, unk_score=-math.inf, sil_score=-1, log_add=False,
                                 criterion_type=CriterionType.ASG)
    decoder = LexiconDecoder(opts, trie, lm, sil_idx, -1, unk_idx, transitions, False)
    results = decoder.decode(emissions.ctypes.data, T, N)
    assert len(results) == 16
    for r, target in zip(results[:5], [-284.0998, -284.108, -284.119, -284.127, -284.296]):
        assert abs(r.score - target) < 1e-3
    assert len(results[0].tokens) == T + 2


@pytest.mark.gpu
def test_decode_batch_binding(gpu_session):
    from flashlight.lib.text.decoder import CriterionType, LexiconFreeDecoder, LexiconFreeDecoderOptions, ZeroLM
    from text_amd import synth
    N, Ts = 29, [40, 7, 25]
    opts = LexiconFreeDecoderOptions(10, N, 25.0, 0.0, 0.0, False, CriterionType.CTC)
    dec = LexiconFreeDecoder(opts, ZeroLM(), 0, N - 1, [])
    embs = [synth.emissions("ctc", 3 + i, T, N) for i, T in enumerate(Ts)]
    flat = np.concatenate([e.reshape(-1) for e in embs])
    batch = dec.decode_batch(flat.ctypes.data, Ts, N)
    for e, T, got in zip(embs, Ts, batch):
        one = dec.decode(e.ctypes.data, T, N)
        assert [r.score for r in one] == [r.score for r in got]
        assert [r.tokens for r in one] == [r.tokens for r in got]


@pytest.mark.gpu
def test_best_hypothesis_with_look_back_after_decode(gpu_session, oracle_lib):
    """decode() runs on the lane engines, which keep no per-frame score history; get_best_hypothesis(look_back) after it
    (Decoder.h:70, Utils.h:229-247: an ancestor of the best final hypothesis, with the ancestor's scores) decodes the
    utterance again with the history from the library's device copy of the emissions: equal to the oracle's answer,
    and the n-best read afterwards is the one decode() returned."""
    from flashlight.lib.text.decoder import CriterionType, LexiconFreeDecoder, LexiconFreeDecoderOptions, ZeroLM
    from oracle import orclib
    from text_amd import synth
    N, T, K = 29, 120, 12
    e = synth.emissions("ctc", 91, T, N)
    dec = LexiconFreeDecoder(LexiconFreeDecoderOptions(K, N, 25.0, 0.0, -0.3, False, CriterionType.CTC), ZeroLM(), 0, N - 1, [])
    nbest = dec.decode(e.ctypes.data, T, N)
    olm = oracle_lib.lm_zero_create()
    od = oracle_lib.lexfree(orclib.make_options(K, N, 25.0, 0.0, 0.0, float("-inf"), -0.3, False, "ctc"), olm, 0, N - 1)
    want = oracle_lib.decode(od, e, T, N)
    assert [r.score for r in nbest] == [h.score for h in want]
    for lb in (5, 0, 30):
        got = dec.get_best_hypothesis(lb)
        ref = oracle_lib.best(od, lb, T + 8)
        assert (got.score, got.emittingModelScore, got.lmScore) == (ref.score, ref.am, ref.lm), lb
        assert list(got.tokens) == [int(x) for x in ref.tokens] and len(got.tokens) == T + 2 - lb
    again = dec.get_all_final_hypothesis()
    assert [(r.score, r.tokens) for r in again] == [(r.score, r.tokens) for r in nbest]
    oracle_lib.decoder_destroy(od)


@pytest.mark.gpu
def test_one_decoder_per_thread(gpu_session):
    """The reference's pattern for parallel decoding -- one decoder object per thread over a shared Trie / LM
    (Utils.h:60-62) -- through the binding: every thread's decoder runs on a stream of its own (a context per thread)
    and decode() releases the GIL while it waits for the device; results equal the sequential ones."""
    import threading
    from flashlight.lib.text.decoder import (CriterionType, KenLM, LexiconDecoder, LexiconDecoderOptions, SmearingMode,
                                             Trie)
    from flashlight.lib.text.dictionary import Dictionary
    from text_amd import synth
    import cases
    c = cases.BY_NAME["ng_word_t60_k16_4g"]
    inp = helpers.case_inputs(c)
    path, vocab = helpers.arpa_path(c, inp)
    wd = Dictionary(vocab[:-1])
    wd.add_entry("<unk>")
    lm = KenLM(path, wd)
    trie = Trie(c["N"], 0)
    sf, so = inp["lex"]
    start = lm.start(False)
    for w in range(inp["W"]):
        trie.insert([int(x) for x in sf[so[w]:so[w + 1]]], w, lm.score(start, w)[1])
    trie.smear(SmearingMode.MAX)
    opts = LexiconDecoderOptions(c["K"], c["Kt"], c["thr"], c["lm_weight"], c["word_score"], c["unk_score"], c["sil_score"],
                                 False, CriterionType.CTC)
    n_thr, per = 4, 6
    embs = [[synth.emissions("lexspell", 300 + 10 * t + i, c["T"], c["N"], lexicon=inp["lex"]) for i in range(per)]
            for t in range(n_thr)]
    seq = LexiconDecoder(opts, trie, lm, 0, c["N"] - 1, inp["W"], [], False)
    want = [[[(r.score, r.tokens, r.words) for r in seq.decode(e.ctypes.data, c["T"], c["N"])] for e in row] for row in embs]
    got = [None] * n_thr
    errs = []

    def work(t):
        try:
            dec = LexiconDecoder(opts, trie, lm, 0, c["N"] - 1, inp["W"], [], False)  # (built in its thread: its context)
            got[t] = [[(r.score, r.tokens, r.words) for r in dec.decode(e.ctypes.data, c["T"], c["N"])] for e in embs[t]]
        except Exception as ex:  # noqa: BLE001
            errs.append(repr(ex))

    th = [threading.Thread(target=work, args=(t,)) for t in range(n_thr)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errs, errs
    assert got == want


def test_trie_node_tree_is_walkable():
    """TrieNode.children through get_root() / search(), as Python code over the reference's
    binding can do (_decoder.cpp:173-186): the tree is materialised from the host trie."""
    from flashlight.lib.text.decoder import SmearingMode, Trie
    trie = Trie(6, 0)
    words = {7: [1, 2], 8: [1, 3], 9: [1, 3, 4], 10: [5]}
    for label, spelling in words.items():
        trie.insert(spelling, label, -float(label))
    trie.smear(SmearingMode.MAX)
    root = trie.get_root()
    assert sorted(root.children.keys()) == [1, 5]
    n1 = root.children[1]
    assert sorted(n1.children.keys()) == [2, 3] and n1.labels == []
    assert n1.children[3].labels == [8] and n1.children[3].children[4].labels == [9]
    assert n1.max_score == -7.0  # smeared maximum of the subtree

    def walk(node, prefix, out):
        for lab in node.labels:
            out[lab] = prefix
        for tok, child in node.children.items():
            assert child.idx == tok
            walk(child, prefix + [tok], out)
        return out
    assert walk(root, [], {}) == words
    assert trie.search([1, 3]).children[4].idx == 4
    trie.insert([5, 5], 11, -0.5)  # a later insert shows up in the next walk
    assert 5 in trie.get_root().children[5].children


@pytest.mark.gpu
def test_decode_batch_arrays_and_devices(gpu_session):
    """decode_batch_arrays: NumPy views over the library's pinned buffers + lazily built
    DecodeResult objects; decode_batch(devices=[0, 0]): two contexts, same results."""
    from flashlight.lib.text.decoder import CriterionType, LexiconFreeDecoder, LexiconFreeDecoderOptions, ZeroLM
    from text_amd import synth
    N, Ts = 29, [40, 7, 25, 0, 33]
    opts = LexiconFreeDecoderOptions(10, N, 25.0, 0.0, 0.0, False, CriterionType.CTC)
    dec = LexiconFreeDecoder(opts, ZeroLM(), 0, N - 1, [])
    embs = [synth.emissions("ctc", 3 + i, T, N) for i, T in enumerate(Ts)]
    flat = np.concatenate([e.reshape(-1) for e in embs])
    ref = dec.decode_batch(flat.ctypes.data, Ts, N)
    two = dec.decode_batch(flat.ctypes.data, Ts, N, devices=[0, 0])
    for a, b in zip(ref, two):
        assert [r.score for r in a] == [r.score for r in b] and [r.tokens for r in a] == [r.tokens for r in b]
        assert [r.emittingModelScore for r in a] == [r.emittingModelScore for r in b]
    res = dec.decode_batch_arrays(flat.ctypes.data, Ts, N)
    assert len(res) == len(Ts) and res.words is None
    assert list(res.n_hyp) == [len(x) for x in ref] and list(res.length) == [T + 2 for T in Ts]
    for b, want in enumerate(ref):
        got = res[b]
        assert [r.score for r in got] == [r.score for r in want]
        for i, r in enumerate(want):
            assert res.scores[b, i, 0] == r.score and res.scores[b, i, 1] == r.emittingModelScore
            assert list(res.tokens_of(b, i)) == r.tokens
            L = res.length[b]
            assert list(res.tokens[res.offsets[b] + i * L: res.offsets[b] + (i + 1) * L]) == r.tokens


@pytest.mark.gpu
def test_streaming_through_the_facade_at_a_big_beam(gpu_session):
    """decode_begin / decode_step / decode_end with beam 500 (the stream's frame budget follows
    the beam: 2^15 frames would not be indexable) equals decode() of the same utterance."""
    from flashlight.lib.text.decoder import CriterionType, LexiconFreeDecoder, LexiconFreeDecoderOptions, ZeroLM
    from text_amd import synth
    N, T = 29, 90
    opts = LexiconFreeDecoderOptions(500, N, 25.0, 0.0, 0.0, False, CriterionType.CTC)
    dec = LexiconFreeDecoder(opts, ZeroLM(), 0, N - 1, [])
    e = synth.emissions("ctc", 11, T, N)
    one = dec.decode(e.ctypes.data, T, N)
    dec.decode_begin()
    for k in range(0, T, 30):
        chunk = np.ascontiguousarray(e[k:k + 30])
        dec.decode_step(chunk.ctypes.data, 30, N)
    dec.decode_end()
    got = dec.get_all_final_hypothesis()
    assert [r.score for r in got] == [r.score for r in one]
    assert [r.tokens for r in got] == [r.tokens for r in one]
    dec.set_max_stream_frames(64)
    dec.decode_begin()
    with pytest.raises(IndexError):
        for k in range(0, T, 30):
            chunk = np.ascontiguousarray(e[k:k + 30])
            dec.decode_step(chunk.ctypes.data, 30, N)


def test_replabels_and_contiguity_known_answers():
    """DictionaryTest.cpp:24-147 known answers (PackReplabels, UnpackReplabels, UnpackReplabelsIgnoresInvalid,
    the contiguity check of Dictionary.cpp:125-137) through the Python names of _dictionary.cpp:45,58-59."""
    from flashlight.lib.text.dictionary import Dictionary, pack_replabels, unpack_replabels
    d = Dictionary()
    for i in (1, 2, 3):
        d.add_entry("<%d>" % i, i)
    labels = [5, 6, 6, 6, 10, 8, 8, 10, 10, 10, 10, 10]
    packed = [labels,
              [5, 6, 1, 6, 10, 8, 1, 10, 1, 10, 1, 10],
              [5, 6, 2, 10, 8, 1, 10, 2, 10, 1],
              [5, 6, 2, 10, 8, 1, 10, 3, 10]]
    for reps in range(4):
        assert pack_replabels(labels, d, reps) == packed[reps]
        assert unpack_replabels(packed[reps], d, reps) == labels
    d = Dictionary()
    for i, e in enumerate(["<1>", "<2>", "<3>", "1", "2", "3"]):
        d.add_entry(e, i + 1)
    labels = [6, 3, 7, 2, 8, 0, 1]
    assert unpack_replabels(labels, d, 1) == [6, 3, 7, 2, 8, 0, 0]
    assert unpack_replabels(labels, d, 2) == [6, 3, 7, 7, 7, 8, 0, 0]
    assert unpack_replabels(labels, d, 3) == [6, 6, 6, 6, 7, 7, 7, 8, 0, 0]
    d = Dictionary()
    for i, e in enumerate(["<1>", "<2>", "1", "2"]):
        d.add_entry(e, i + 1)
    assert unpack_replabels([1, 5, 1, 6], d, 2) == [5, 5, 6]         # leading replabel: nothing to repeat
    assert unpack_replabels([1, 5, 1, 2, 6], d, 2) == [5, 5, 6]      # replabel after a replabel
    assert unpack_replabels([1, 5, 1, 2, 6], d, 1) == [5, 5, 2, 6]   # "<2>" is an ordinary token at maxReps 1
    assert unpack_replabels([5, 1, 2, 1, 2, 6], d, 2) == [5, 5, 6]
    assert unpack_replabels([], d, 2) == [] and unpack_replabels([4, 1], d, 0) == [4, 1]
    # contiguity (DictionaryTest.cpp:24-48: entries added one after the other are contiguous; a hole is not)
    d = Dictionary()
    d.add_entry("a")
    d.add_entry("b")
    assert d.is_contiguous()
    d.add_entry("c", 5)
    assert not d.is_contiguous()
    d2 = Dictionary()
    d2.add_entry("x", 0)
    d2.add_entry("y", 0)   # two entries, one index: still contiguous
    d2.add_entry("z", 1)
    assert d2.is_contiguous() and d2.entry_size() == 3 and d2.index_size() == 2
    assert d2.map_entries_to_indices(["z", "x"]) == [1, 0] and d2.map_indices_to_entries([1, 0]) == ["z", "x"]


@pytest.mark.gpu
def test_python_lm_subclass_decodes_on_the_device(gpu_session, golden, oracle_lib):
    """The reference's PyLM extension point (bindings/python/flashlight/lib/text/_decoder.cpp:39-56): a Python
    subclass of `LM` passed to the decoders.  (i) a ZeroLM clone reproduces the reference's golden n-best for C1;
    (ii) a Python LM over the KenLM object's scores equals the device n-gram path; (iii) a Python exception raised
    inside score() is what decode() raises."""
    import cases
    from flashlight.lib.text.decoder import (LM, CriterionType, KenLM, LexiconDecoder, LexiconDecoderOptions,
                                             LexiconFreeDecoder, LexiconFreeDecoderOptions, LMState, SmearingMode,
                                             Trie, ZeroLM)
    from flashlight.lib.text.dictionary import Dictionary

    class PyZero(LM):
        def __init__(self):
            LM.__init__(self)
            self.calls = 0

        def start(self, start_with_nothing):
            return LMState()

        def score(self, state, idx):
            self.calls += 1
            return state.child(idx), 0.0

        def finish(self, state):
            return state, 0.0

    c = cases.BY_NAME["C1_ctc_u0"]
    inp = helpers.case_inputs(c)
    e = np.ascontiguousarray(inp["e"], dtype=np.float32)
    opts = LexiconFreeDecoderOptions(beam_size=c["K"], beam_size_token=c["Kt"], beam_threshold=c["thr"], lm_weight=0.0,
                                     sil_score=0.0, log_add=False, criterion_type=CriterionType.CTC)
    user = PyZero()
    dec = LexiconFreeDecoder(opts, user, 0, c["N"] - 1, [])
    res = dec.decode(e.ctypes.data, c["T"], c["N"])
    assert user.calls > 0
    exp = golden["C1_ctc_u0"]
    assert len(res) == exp["n"]
    for r, sc in zip(res, exp["scores"]):
        assert r.score == float.fromhex(sc[0]) and r.emittingModelScore == float.fromhex(sc[1])
    assert list(res[0].tokens) == exp["tokens"][0]
    # the same through decode_step chunks
    dec.decode_begin()
    for a, b in ((0, 1), (1, 90), (90, c["T"])):
        chunk = np.ascontiguousarray(e[a:b])
        dec.decode_step(chunk.ctypes.data, b - a, c["N"])
    dec.decode_end()
    res2 = dec.get_all_final_hypothesis()
    assert [r.score for r in res2] == [r.score for r in res] and list(res2[0].tokens) == list(res[0].tokens)

    # (ii) word LM: KenLM wrapped in a Python LM (states are KenLM's own LMState objects)
    cw = cases.BY_NAME["ng_word_t60_k16_4g"]
    inpw = helpers.case_inputs(cw)
    path, vocab = helpers.arpa_path(cw, inpw)
    wd = Dictionary(vocab)
    kenlm = KenLM(path, wd)

    class Wrapped(LM):
        def __init__(self, inner):
            LM.__init__(self)
            self.inner = inner

        def start(self, n):
            return self.inner.start(n)

        def score(self, state, idx):
            return self.inner.score(state, idx)

        def finish(self, state):
            return self.inner.finish(state)

    trie = Trie(cw["N"], 0)
    sf, so = inpw["lex"]
    st0 = kenlm.start(False)
    for w in range(inpw["W"]):
        trie.insert([int(t) for t in sf[so[w]:so[w + 1]]], w, kenlm.score(st0, w)[1])
    trie.smear(SmearingMode.MAX)
    lo = LexiconDecoderOptions(beam_size=cw["K"], beam_size_token=cw["Kt"], beam_threshold=cw["thr"],
                               lm_weight=cw["lm_weight"], word_score=cw["word_score"], unk_score=cw["unk_score"],
                               sil_score=cw["sil_score"], log_add=False, criterion_type=CriterionType.CTC)
    ew = np.ascontiguousarray(inpw["e"], dtype=np.float32)
    d_dev = LexiconDecoder(lo, trie, kenlm, 0, cw["N"] - 1, inpw["W"], [], False)
    d_usr = LexiconDecoder(lo, trie, Wrapped(kenlm), 0, cw["N"] - 1, inpw["W"], [], False)
    r_dev = d_dev.decode(ew.ctypes.data, cw["T"], cw["N"])
    r_usr = d_usr.decode(ew.ctypes.data, cw["T"], cw["N"])
    expw = golden["ng_word_t60_k16_4g"]
    assert len(r_dev) == len(r_usr) == expw["n"]
    for a, b, sc in zip(r_dev, r_usr, expw["scores"]):
        assert (a.score, a.emittingModelScore, a.lmScore) == (b.score, b.emittingModelScore, b.lmScore)
        assert list(a.tokens) == list(b.tokens) and list(a.words) == list(b.words)
        assert b.score == float.fromhex(sc[0])

    # (iii) exceptions
    class Bad(PyZero):
        def score(self, state, idx):
            if self.calls > 50:
                raise KeyError("user LM failed on purpose")
            return PyZero.score(self, state, idx)

    bad = LexiconFreeDecoder(opts, Bad(), 0, c["N"] - 1, [])
    with pytest.raises(KeyError):
        bad.decode(e.ctypes.data, c["T"], c["N"])
