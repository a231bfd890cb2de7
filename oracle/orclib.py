"""ctypes front-end for the two CPU checkers (TEST INFRASTRUCTURE ONLY).

* ``load("oracle")`` -> oracle/liboracle.so, this repo's CPU restatement.
* ``load("ref")``    -> oracle/_ref/libfltref.so, the unmodified reference
  compiled in the dev container by ``make -C oracle ref``.

Only tests/, ``__graft_entry__.smoke()`` and bench.py's ``cpu_baseline`` leg
may import this module; nothing under text_amd/ does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


class Options(C.Structure):
    """Mirror of orc_options (oracle/orc_api.h)."""

    _fields_ = [
        ("beam_size", C.c_int32),
        ("beam_size_token", C.c_int32),
        ("beam_threshold", C.c_double),
        ("lm_weight", C.c_double),
        ("word_score", C.c_double),
        ("unk_score", C.c_double),
        ("sil_score", C.c_double),
        ("log_add", C.c_int32),
        ("criterion", C.c_int32),
    ]


def make_options(beam_size, beam_size_token, beam_threshold=25.0, lm_weight=0.0,
                 word_score=0.0, unk_score=-float("inf"), sil_score=0.0,
                 log_add=False, criterion="ctc"):
    crit = {"asg": 0, "ctc": 1}[criterion] if isinstance(criterion, str) else int(criterion)
    return Options(beam_size, beam_size_token, beam_threshold, lm_weight,
                   word_score, unk_score, sil_score, int(bool(log_add)), crit)


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class Hyp:
    __slots__ = ("score", "am", "lm", "tokens", "words")

    def __init__(self, score, am, lm, tokens, words):
        self.score, self.am, self.lm = score, am, lm
        self.tokens, self.words = tokens, words


class CheckerLib:
    def __init__(self, path, prefix):
        self.lib = C.CDLL(path)
        self.prefix = prefix
        self.path = path
        L = self.lib
        vp, i32, f32 = C.c_void_p, C.c_int32, C.c_float
        pf, pi, pd = C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_double)
        po = C.POINTER(Options)
        sig = {
            "lm_zero_create": (vp, []),
            "lm_arpa_create": (vp, [C.c_char_p, C.c_char_p]),
            "lm_lastword_create": (vp, [i32, i32]),
            "lm_destroy": (None, [vp]),
            "lm_score_sequence": (f32, [vp, pi, i32, i32, pf]),
            "trie_create": (vp, [i32, i32]),
            "trie_insert": (i32, [vp, pi, i32, i32, f32]),
            "trie_smear": (None, [vp, i32]),
            "trie_search": (i32, [vp, pi, i32, pf, pi]),
            "trie_num_nodes": (C.c_int64, [vp]),
            "trie_destroy": (None, [vp]),
            "decoder_create_lexfree": (vp, [po, vp, i32, i32, pf, i32]),
            "decoder_create_lexicon": (vp, [po, vp, vp, i32, i32, i32, pf, i32, i32]),
            "decoder_destroy": (None, [vp]),
            "decoder_begin": (None, [vp]),
            "decoder_step": (None, [vp, pf, i32, i32]),
            "decoder_end": (None, [vp]),
            "decoder_prune": (None, [vp, i32]),
            "decoder_n_frames_in_buffer": (i32, [vp]),
            "decoder_n_final": (i32, [vp, pi]),
            "decoder_get_all": (i32, [vp, i32, pd, pi, pi]),
            "decoder_get_best": (i32, [vp, i32, pd, pi, pi, i32]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, prefix + name)
            fn.restype, fn.argtypes = res, args
            setattr(self, name, fn)
        if prefix == "orc_":
            L.orc_decoder_ties.restype = None
            L.orc_decoder_ties.argtypes = [vp, C.POINTER(C.c_int64), i32]
        if prefix == "ref_":
            L.ref_lexicon_dump.restype = C.c_void_p
            L.ref_lexicon_dump.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, i32]
            L.ref_free.argtypes = [vp]

    # -- convenience --------------------------------------------------------
    def build_trie(self, max_children, root_idx, spell_flat, spell_off, labels,
                   scores, smear=1):
        t = self.trie_create(max_children, root_idx)
        sf = np.ascontiguousarray(spell_flat, dtype=np.int32)
        for w in range(len(spell_off) - 1):
            a, b = int(spell_off[w]), int(spell_off[w + 1])
            seg = sf[a:b]
            rc = self.trie_insert(t, _ip(seg), b - a, int(labels[w]), float(scores[w]))
            if rc != 0:
                raise IndexError("trie_insert: invalid index")
        self.trie_smear(t, smear)
        return t

    def collect(self, dec):
        length = C.c_int32(0)
        n = self.decoder_n_final(dec, C.byref(length))
        L = length.value
        if n == 0:
            return []
        scores = np.zeros(3 * n, dtype=np.float64)
        tokens = np.zeros(n * L, dtype=np.int32)
        words = np.zeros(n * L, dtype=np.int32)
        got = self.decoder_get_all(dec, n, _dp(scores), _ip(tokens), _ip(words))
        assert got == n
        tokens = tokens.reshape(n, L)
        words = words.reshape(n, L)
        return [Hyp(scores[3 * i], scores[3 * i + 1], scores[3 * i + 2],
                    tokens[i].copy(), words[i].copy()) for i in range(n)]

    TIE_KINDS = ("merge", "cut", "order", "token", "best")

    def ties(self, dec, reset=False):
        """{kind: count} of the ties the ORACLE's decoder has passed (oracle.cpp TieCounts): the places where the
        reference's answer depends on addresses.  None for the compiled reference (it cannot know)."""
        if self.prefix != "orc_":
            return None
        out = (C.c_int64 * 5)()
        self.lib.orc_decoder_ties(dec, out, int(reset))
        return dict(zip(self.TIE_KINDS, [int(v) for v in out]))

    def best(self, dec, look_back, capacity):
        scores = np.zeros(3, dtype=np.float64)
        tokens = np.zeros(capacity, dtype=np.int32)
        words = np.zeros(capacity, dtype=np.int32)
        n = self.decoder_get_best(dec, look_back, _dp(scores), _ip(tokens), _ip(words), capacity)
        assert n >= 0
        return Hyp(scores[0], scores[1], scores[2], tokens[:n].copy(), words[:n].copy())

    def decode(self, dec, emissions, T, N):
        """Decoder::decode (decoder/Decoder.h:51-57)."""
        e = np.ascontiguousarray(emissions, dtype=np.float32)
        self.decoder_begin(dec)
        self.decoder_step(dec, _fp(e), T, N)
        self.decoder_end(dec)
        self.last_ties = self.ties(dec)  # (oracle: what the decode passed; the n-best's `order` ties included)
        return self.collect(dec)

    def lexfree(self, opt, lm, sil, blank, transitions=None):
        if transitions is None or len(transitions) == 0:
            return self.decoder_create_lexfree(C.byref(opt), lm, sil, blank, None, 0)
        tr = np.ascontiguousarray(transitions, dtype=np.float32)
        return self.decoder_create_lexfree(C.byref(opt), lm, sil, blank, _fp(tr), tr.size)

    def lexicon(self, opt, trie, lm, sil, blank, unk, transitions=None, is_lm_token=False):
        if transitions is None or len(transitions) == 0:
            return self.decoder_create_lexicon(C.byref(opt), trie, lm, sil, blank, unk,
                                               None, 0, int(is_lm_token))
        tr = np.ascontiguousarray(transitions, dtype=np.float32)
        return self.decoder_create_lexicon(C.byref(opt), trie, lm, sil, blank, unk,
                                           _fp(tr), tr.size, int(is_lm_token))

    def lexicon_dump(self, words_path, tokens_path, extra_token="", max_reps=1):
        assert self.prefix == "ref_"
        p = self.lib.ref_lexicon_dump(words_path.encode(), tokens_path.encode(),
                                      extra_token.encode(), max_reps)
        if not p:
            raise RuntimeError("ref_lexicon_dump failed")
        s = C.string_at(p).decode()
        self.lib.ref_free(p)
        return s


def build(target="oracle"):
    """Compile the checker (g++, seconds).  `ref` needs /root/reference."""
    subprocess.run(["make", "-C", HERE, target], check=True,
                   stdout=subprocess.PIPE, stderr=subprocess.STDOUT)


def load(kind="oracle"):
    if kind == "oracle":
        path = os.path.join(HERE, "liboracle.so")
        if not os.path.exists(path):
            build("oracle")
        return CheckerLib(path, "orc_")
    if kind == "ref":
        path = os.path.join(HERE, "_ref", "libfltref.so")
        if not os.path.exists(path):
            build("ref")
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        return CheckerLib(path, "ref_")
    raise ValueError(kind)


def have_ref():
    return os.path.exists(os.path.join(HERE, "_ref", "libfltref.so"))


def nbest_hash(hyps):
    """n-best hash of SURVEY.md Appendix A (FNV-1a style over score bits,
    tokens and words)."""
    h = 1469598103934665603
    M = (1 << 64) - 1

    def mix(v):
        nonlocal h
        h ^= v
        h = (h * 1099511628211) & M

    for hyp in hyps:
        mix(int(np.float64(hyp.score).view(np.uint64)))
        for t in hyp.tokens:
            mix(int(t) & 0xFFFFFFFF)
        for w in hyp.words:
            mix(int(w) & 0xFFFFFFFF)
    return h
