"""One batch over several devices (fltx_group_*): utterances
shard with no exchange, results come back in input order.  The CPU test drives the
host-thread emulation (two contexts on the one emulated device); the GPU test the
HIP library with devices = {0, 0} on the 1-GPU box -- two contexts, two streams,
two host threads, exactly the code path of two devices."""
import numpy as np
import pytest

import cases
import helpers
from text_amd import _capi, synth

TS_FULL = [0, 1, 37, 200, 123, 64, 5, 199, 150, 80, 3]
TS_EMU = [0, 1, 37, 64, 5]  # (the emulator runs a workgroup as hundreds of host threads: short utterances)
TS = TS_FULL


def _ragged_batch(N, TS):
    es = [synth.emissions("ctc", 100 + i, T, N) for i, T in enumerate(TS)]
    flat = np.concatenate([e.reshape(-1) for e in es]) if es else np.zeros(0, dtype=np.float32)
    return es, flat.astype(np.float32)


def _check_group(lib, oracle_lib, devices, kind="lexfree", TS=TS_FULL):
    c = dict(cases.BY_NAME["C1_ctc_u0" if kind == "lexfree" else "lx_spell_t40_k8"])
    N = c["N"]
    opt = _capi.make_options(c["K"], c["Kt"], c["thr"])
    lm = _capi.ZeroLM(lib=lib)
    host_trie = None
    inp = helpers.case_inputs(c)
    if kind == "lexicon":
        host_trie = _capi.HostTrie(N, 0, lib=lib)
        sf, so = inp["lex"]
        host_trie.insert_many(sf, so, inp["labels"], inp["scores"])
        host_trie.smear(1)
    if kind == "lexicon":
        es = [synth.emissions("lexspell", 7 + i, T, N, lexicon=inp["lex"]) for i, T in enumerate(TS)]
        flat = np.concatenate([e.reshape(-1) for e in es]).astype(np.float32)
    else:
        es, flat = _ragged_batch(N, TS)
    g = _capi.DecoderGroup(devices, _capi.LEXFREE if kind == "lexfree" else _capi.LEXICON, opt, lm, 0, N - 1,
                           unk=inp["W"] if kind == "lexicon" else -1, host_trie=host_trie, lib=lib)
    g.decode_batch(flat, TS, N)
    parts = g.parts()
    assert sum(cnt for _, _, cnt in parts) == len(TS)
    assert [f for _, f, _ in parts] == sorted(f for _, f, _ in parts)
    if len(devices) > 1:
        assert sum(1 for _, _, cnt in parts if cnt > 0) > 1  # the batch really was split
    for b, T in enumerate(TS):
        cc = dict(c, T=T)
        want = helpers.run_checker(oracle_lib, cc, dict(inp, e=es[b]))
        ok, why = helpers.hyps_equal(want, g.results(b))
        assert ok, "utterance %d (T=%d): %s" % (b, T, why)
    # a second, differently shaped batch on the same group
    g.decode_batch(flat[: TS[0] * N + TS[1] * N + TS[2] * N], TS[:3], N)
    for b in range(3):
        want = helpers.run_checker(oracle_lib, dict(c, T=TS[b]), dict(inp, e=es[b]))
        ok, why = helpers.hyps_equal(want, g.results(b))
        assert ok, why
    g.close()


@pytest.mark.parametrize("devices", [[0], [0, 0]])
def test_group_on_the_emulated_kernels_matches_oracle(emu_session, oracle_lib, devices):
    _check_group(emu_session.lib, oracle_lib, devices, TS=TS_EMU)


@pytest.mark.gpu
@pytest.mark.parametrize("devices,kind", [([0, 0], "lexfree"), ([0, 0, 0, 0], "lexfree"), ([0, 0], "lexicon")])
def test_group_two_contexts_on_one_gpu_matches_oracle(gpu_session, oracle_lib, devices, kind):
    _check_group(gpu_session.lib, oracle_lib, devices, kind)


def _recreate_groups_with_ngram(sess, oracle_lib, lists=([0], [0, 0], [0], [0, 0, 0], [0, 0])):
    """Groups come and go over one LM object (DeviceDecoder::decodeBatchOn rebuilds its group when the
    device list changes): a context's copy of the n-gram tables dies with the context and a later
    context -- possibly at the same address -- gets its own upload (round-2 advisor finding)."""
    lib = sess.lib
    c = dict(cases.BY_NAME["ng_word_t40_k10"])
    inp = helpers.case_inputs(c)
    N = c["N"]
    lm = sess.lm_for(c, inp)
    host_trie = _capi.HostTrie(N, 0, lib=lib)
    sf, so = inp["lex"]
    scores = np.array([lm.score_sequence([w], False)[0][0] for w in range(inp["W"])], dtype=np.float32)
    host_trie.insert_many(sf, so, inp["labels"], scores)
    host_trie.smear(1)
    opt = _capi.make_options(c["K"], c["Kt"], c["thr"], c["lm_weight"], c["word_score"], c["unk_score"],
                             c["sil_score"], c["log_add"], c["crit"])
    want = helpers.run_checker(oracle_lib, c, inp)
    flat = np.concatenate([inp["e"].reshape(-1)] * 3).astype(np.float32)
    for devices in lists:
        g = _capi.DecoderGroup(devices, _capi.LEXICON, opt, lm, 0, N - 1, unk=inp["W"], host_trie=host_trie,
                               lib=lib)
        g.decode_batch(flat, [c["T"]] * 3, N)
        for b in range(3):
            ok, why = helpers.hyps_equal(want, g.results(b))
            assert ok, "devices %r, utterance %d: %s" % (devices, b, why)
        g.close()


def test_groups_recreated_over_one_ngram_lm_emulated(emu_session, oracle_lib):
    _recreate_groups_with_ngram(emu_session, oracle_lib, lists=([0], [0, 0], [0]))


@pytest.mark.gpu
def test_groups_recreated_over_one_ngram_lm(gpu_session, oracle_lib):
    _recreate_groups_with_ngram(gpu_session, oracle_lib)
