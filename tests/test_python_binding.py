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
