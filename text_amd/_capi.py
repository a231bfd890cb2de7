"""ctypes binding of the C ABI in include/fltx.h (text_amd/lib/libfltx.so).

This is the only way Python reaches the decoder: every decode call runs the
HIP kernels.  If the shared library is missing or no gfx950 device is usable
the calls raise -- there is no CPU fallback.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "lib", "libfltx.so")

FLTX_OK, ERR_INVALID, ERR_HIP, ERR_OOM, ERR_UNSUPPORTED, ERR_RANGE, ERR_STATE = range(7)
CRITERION = {"asg": 0, "ctc": 1}
LEXFREE, LEXICON = 0, 1
# fltx_decoder_get "why_not_lane" (include/fltx.h FLTX_WHY_*)
(FLTX_WHY_TOKENS, FLTX_WHY_BEAM, FLTX_WHY_STREAM, FLTX_WHY_LM, FLTX_WHY_LOGADD, FLTX_WHY_ASG, FLTX_WHY_UNK,
 FLTX_WHY_TRIE_SHAPE, FLTX_WHY_WORD_END, FLTX_WHY_OPTIONS, FLTX_WHY_LENGTH, FLTX_WHY_SWITCHED_OFF,
 FLTX_WHY_GEOMETRY) = (1 << i for i in range(13))


class Options(C.Structure):
    """fltx_options == LexiconDecoderOptions (decoder/LexiconDecoder.h:21-31)."""

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


class FltxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("fltx error %d: %s" % (code, msg))
        self.code = code


_EXC = {ERR_INVALID: ValueError, ERR_RANGE: IndexError}


class Lib:
    """A loaded libfltx with typed entry points."""

    SYMBOLS = [
        "fltx_last_error", "fltx_version", "fltx_ctx_create", "fltx_ctx_destroy",
        "fltx_ctx_synchronize", "fltx_ctx_stream", "fltx_ctx_uid", "fltx_lm_zero_create",
        "fltx_lm_ngram_create", "fltx_lm_host_create", "fltx_lm_arpa_load", "fltx_lm_state_size", "fltx_lm_start", "fltx_lm_step", "fltx_lm_destroy", "fltx_lm_score_sequence",
        "fltx_trie_create", "fltx_trie_destroy", "fltx_decoder_create",
        "fltx_decoder_destroy", "fltx_decode_batch", "fltx_stream_begin",
        "fltx_stream_step", "fltx_stream_end", "fltx_stream_prune",
        "fltx_stream_frames_in_buffer", "fltx_result_count", "fltx_result_fetch", "fltx_result_fetch_batch", "fltx_result_fetch_batch_compact",
        "fltx_result_best", "fltx_result_device", "fltx_decoder_stats",
        "fltx_decoder_set", "fltx_decoder_get", "fltx_decoder_timing", "fltx_decoder_profile", "fltx_htrie_create", "fltx_htrie_destroy", "fltx_htrie_insert",
        "fltx_htrie_search", "fltx_htrie_smear", "fltx_htrie_num_nodes", "fltx_htrie_upload",
        "fltx_decoder_bytes", "fltx_htrie_node", "fltx_group_create", "fltx_group_destroy", "fltx_group_size", "fltx_group_decoder",
        "fltx_group_decode_batch", "fltx_group_result_count", "fltx_group_result_fetch", "fltx_group_synchronize",
    ]

    def __init__(self, path=None):
        path = path or DEFAULT_LIB
        if not os.path.exists(path):
            raise FileNotFoundError(
                "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback)" % path)
        self.path = path
        L = self.lib = C.CDLL(path)
        vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
        pvp = C.POINTER(C.c_void_p)
        L.fltx_last_error.restype = C.c_char_p
        L.fltx_version.restype = C.c_char_p
        L.fltx_ctx_stream.restype = vp
        L.fltx_ctx_stream.argtypes = [vp]
        L.fltx_ctx_uid.restype = C.c_uint64
        L.fltx_ctx_uid.argtypes = [vp]
        sig = {
            "fltx_ctx_create": [C.c_int, vp, pvp],
            "fltx_ctx_destroy": [vp],
            "fltx_ctx_synchronize": [vp],
            "fltx_lm_zero_create": [vp, pvp],
            "fltx_lm_ngram_create": [vp, i32, i64, vp, vp, vp, vp, vp, i32, i32, i32, i32, pvp],
            "fltx_lm_arpa_load": [C.c_char_p, C.c_char_p, pvp],
            "fltx_lm_host_create": [vp, pvp],
            "fltx_lm_state_size": [vp, vp],
            "fltx_lm_start": [vp, i32, vp],
            "fltx_lm_step": [vp, vp, i32, vp, vp],
            "fltx_lm_destroy": [vp],
            "fltx_lm_score_sequence": [vp, vp, i32, i32, vp, vp],
            "fltx_trie_create": [vp, i64, i32, vp, vp, vp, vp, pvp],
            "fltx_trie_destroy": [vp],
            "fltx_decoder_create": [vp, i32, C.POINTER(Options), vp, vp, i32, i32, i32, vp, i32, i32, pvp],
            "fltx_decoder_destroy": [vp],
            "fltx_decode_batch": [vp, vp, i32, vp, vp, i32, i32],
            "fltx_stream_begin": [vp, i32, i32, i32],
            "fltx_stream_step": [vp, vp, i32, vp, vp],
            "fltx_stream_end": [vp],
            "fltx_stream_prune": [vp, i32],
            "fltx_stream_frames_in_buffer": [vp, i32, vp],
            "fltx_result_count": [vp, i32, vp, vp],
            "fltx_result_fetch": [vp, i32, i32, vp, vp, vp, vp],
            "fltx_result_fetch_batch": [vp, pvp, pvp, pvp, pvp, pvp, pvp],
            "fltx_result_fetch_batch_compact": [vp, pvp, pvp, pvp, pvp, pvp, pvp],
            "fltx_result_best": [vp, i32, i32, vp, vp, vp, i32, vp],
            "fltx_result_device": [vp, pvp, pvp, pvp, pvp, pvp],
            "fltx_decoder_stats": [vp, vp, vp, vp, vp],
            "fltx_decoder_set": [vp, C.c_char_p, i64],
            "fltx_decoder_get": [vp, C.c_char_p, vp],
            "fltx_decoder_timing": [vp, vp, vp],
            "fltx_decoder_profile": [vp, vp],
            "fltx_htrie_create": [i32, i32, pvp],
            "fltx_htrie_destroy": [vp],
            "fltx_htrie_insert": [vp, vp, i32, i32, C.c_float],
            "fltx_htrie_search": [vp, vp, i32, vp, vp, vp, vp, vp],
            "fltx_htrie_smear": [vp, i32],
            "fltx_htrie_num_nodes": [vp, vp],
            "fltx_htrie_upload": [vp, vp, pvp],
            "fltx_decoder_bytes": [vp, vp, vp, vp],
            "fltx_htrie_node": [vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, i32],
            "fltx_group_create": [vp, i32, i32, C.POINTER(Options), vp, vp, i32, i32, i32, vp, i32, i32, pvp],
            "fltx_group_destroy": [vp],
            "fltx_group_size": [vp, vp],
            "fltx_group_decoder": [vp, i32, pvp, vp, vp],
            "fltx_group_decode_batch": [vp, vp, vp, vp, vp, i32, i32],
            "fltx_group_result_count": [vp, i32, vp, vp],
            "fltx_group_result_fetch": [vp, i32, i32, vp, vp, vp, vp],
            "fltx_group_synchronize": [vp],
        }
        for name, args in sig.items():
            fn = getattr(L, name)
            fn.restype = C.c_int
            fn.argtypes = args

    def check(self, rc):
        if rc != FLTX_OK:
            msg = self.lib.fltx_last_error().decode()
            raise _EXC.get(rc, FltxError)(msg) if rc in _EXC else FltxError(rc, msg)

    def version(self):
        return self.lib.fltx_version().decode()


_default = None

# Live handles are destroyed explicitly at interpreter exit, decoders first and
# contexts last, while the HIP runtime is still loaded (garbage-collection order
# at shutdown is arbitrary and the runtime aborts if it is torn down first).
import atexit
import weakref

_live = {"dec": weakref.WeakSet(), "trie": weakref.WeakSet(), "lm": weakref.WeakSet(), "ctx": weakref.WeakSet()}


def _close_all():
    for kind in ("dec", "trie", "lm", "ctx"):
        for obj in list(_live[kind]):
            try:
                obj.close()
            except Exception:
                pass


atexit.register(_close_all)


def default_lib():
    global _default
    if _default is None:
        _default = Lib()
    return _default


def _ptr(a):
    return None if a is None else a.ctypes.data


class Context:
    def __init__(self, device=-1, stream=None, lib=None):
        self.L = lib or default_lib()
        h = C.c_void_p()
        self.L.check(self.L.lib.fltx_ctx_create(device, stream, C.byref(h)))
        self.h = h
        _live["ctx"].add(self)

    def synchronize(self):
        self.L.check(self.L.lib.fltx_ctx_synchronize(self.h))

    @property
    def stream(self):
        return self.L.lib.fltx_ctx_stream(self.h)

    def close(self):
        if self.h:
            self.L.lib.fltx_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ZeroLM:
    def __init__(self, ctx=None, lib=None):
        self.ctx, self.L = ctx, (ctx.L if ctx is not None else (lib or default_lib()))
        h = C.c_void_p()
        self.L.check(self.L.lib.fltx_lm_zero_create(ctx.h if ctx is not None else None, C.byref(h)))
        self.h = h
        _live["lm"].add(self)

    def score_sequence(self, words, with_finish=True):
        return self._score(words, with_finish)

    # explicit-state LM::start / score / finish on the host copy of the tables (decoder/lm/LM.h:61-78)
    def state_size(self):
        n = C.c_int32(0)
        self.L.check(self.L.lib.fltx_lm_state_size(self.h, C.addressof(n)))
        return n.value

    def start(self, start_with_nothing=False):
        ctx = np.zeros(max(1, self.state_size()), dtype=np.int32)
        self.L.check(self.L.lib.fltx_lm_start(self.h, int(start_with_nothing), _ptr(ctx)))
        return ctx

    def step(self, ctx, usr_idx):
        """-> (context of the next state, score); usr_idx == -1: LM::finish"""
        out = np.zeros_like(ctx)
        sc = C.c_float(0)
        self.L.check(self.L.lib.fltx_lm_step(self.h, _ptr(ctx), int(usr_idx), _ptr(out), C.addressof(sc)))
        return out, sc.value

    def _score(self, words, with_finish):
        w = np.ascontiguousarray(words, dtype=np.int32)
        per = np.zeros(len(w), dtype=np.float32)
        tot = C.c_float(0)
        self.L.check(self.L.lib.fltx_lm_score_sequence(self.h, _ptr(w), len(w), int(with_finish),
                                                       _ptr(per), C.addressof(tot)))
        return per, tot.value

    def close(self):
        if self.h:
            self.L.lib.fltx_lm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class NgramLM(ZeroLM):
    """Flat back-off n-gram tables in HBM (replaces lm/KenLM.cpp:32-83)."""

    def __init__(self, ctx, order, ngram_order, ngram_words, prob, backoff, usr_to_lm, bos, eos, unk, lib=None):
        self.ctx, self.L = ctx, (ctx.L if ctx is not None else (lib or default_lib()))
        no = np.ascontiguousarray(ngram_order, dtype=np.int32)
        nw = np.ascontiguousarray(ngram_words, dtype=np.int32).reshape(len(no), order)
        pr = np.ascontiguousarray(prob, dtype=np.float32)
        bo = np.ascontiguousarray(backoff, dtype=np.float32)
        um = np.ascontiguousarray(usr_to_lm, dtype=np.int32)
        h = C.c_void_p()
        self.L.check(self.L.lib.fltx_lm_ngram_create(ctx.h if ctx is not None else None, order, len(no), _ptr(no), _ptr(nw), _ptr(pr),
                                                     _ptr(bo), _ptr(um), len(um), bos, eos, unk,
                                                     C.byref(h)))
        self.h = h
        _live["lm"].add(self)


class ArpaLM(ZeroLM):
    """KenLM(path, usr_token_dict) for ARPA text models (lm/KenLM.cpp:32-50)."""

    def __init__(self, path, usr_words, lib=None):
        self.ctx, self.L = None, lib or default_lib()
        h = C.c_void_p()
        self.L.check(self.L.lib.fltx_lm_arpa_load(path.encode(), "\n".join(usr_words).encode(), C.byref(h)))
        self.h = h
        _live["lm"].add(self)


_HLM_START = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_int32)
_HLM_SCORE = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                         C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_float))
_HLM_STATES = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_int32))


class _HostLmStruct(C.Structure):
    _fields_ = [("user", C.c_void_p), ("start", _HLM_START), ("score", _HLM_SCORE),
                ("update_cache", _HLM_STATES), ("retain", _HLM_STATES)]


class HostLM(ZeroLM):
    """A user-defined LM behind fltx_lm_host_create (include/fltx.h): `lm` is any object with the reference's LM
    methods (decoder/lm/LM.h:61-85) -- start(start_with_nothing) -> state, score(state, idx) -> (state, score),
    finish(state) -> (state, score), optionally update_cache(states).  States are arbitrary Python objects; as in
    the reference two hypotheses share an LM state iff the LM returned the SAME object (`is`).  The beam search runs
    in the HIP kernels; once per frame the distinct (state, idx) questions of the whole batch are answered here."""

    def __init__(self, lm, lib=None):
        self.ctx, self.L = None, lib or default_lib()
        self.lm = lm
        self.states = []     # per utterance: id -> state object
        self.ids = []        # per utterance: id(state object) -> id
        self.error = None    # the exception a callback raised (re-raised by the decoder call)
        self.calls = 0       # LM.score / LM.finish calls made
        self.released = 0    # states dropped after prune

        def guard(fn):
            def run(*a):
                try:
                    return fn(*a)
                except BaseException as e:  # noqa: BLE001 -- must not unwind through the C frames
                    self.error = e
                    return 1
            return run

        def start(_user, n_utt):
            self.states, self.ids = [], []
            for _ in range(n_utt):
                s0 = self.lm.start(False)
                self.states.append({0: s0})
                self.ids.append({id(s0): 0})
            self.next = [1] * n_utt
            return 0

        def score(_user, n, utt, state, idx, out_state, out_score):
            for i in range(n):
                b = utt[i]
                st = self.states[b][state[i]]
                self.calls += 1
                ns, sc = self.lm.finish(st) if idx[i] < 0 else self.lm.score(st, idx[i])
                k = self.ids[b].get(id(ns))
                if k is None or self.states[b][k] is not ns:
                    k = self.next[b]
                    self.next[b] += 1
                    self.ids[b][id(ns)] = k
                    self.states[b][k] = ns
                out_state[i] = k
                out_score[i] = sc
            return 0

        def update_cache(_user, b, n, states):
            f = getattr(self.lm, "update_cache", None)
            if f is not None:
                f([self.states[b][states[i]] for i in range(n)])
            return 0

        def retain(_user, b, n, states):
            keep = {states[i] for i in range(n)}
            for k in [k for k in self.states[b] if k not in keep]:
                self.ids[b].pop(id(self.states[b][k]), None)
                del self.states[b][k]
                self.released += 1
            return 0

        self._cb = _HostLmStruct(None, _HLM_START(guard(start)), _HLM_SCORE(guard(score)),
                                 _HLM_STATES(guard(update_cache)), _HLM_STATES(guard(retain)))
        h = C.c_void_p()
        self.L.check(self.L.lib.fltx_lm_host_create(C.byref(self._cb), C.byref(h)))
        self.h = h
        _live["lm"].add(self)

    def score_sequence(self, words, with_finish=True):
        st = self.lm.start(False)
        per = np.zeros(len(words), dtype=np.float32)
        for i, w in enumerate(words):
            st, per[i] = self.lm.score(st, int(w))
        tot = float(per.sum())
        if with_finish:
            tot += self.lm.finish(st)[1]
        return per, tot


class Trie:
    """Flattened, already smeared lexicon trie in HBM (decoder/Trie.h:64-92)."""

    def __init__(self, ctx, child, max_score, label_off, labels):
        self.ctx, self.L = ctx, ctx.L
        ch = np.ascontiguousarray(child, dtype=np.int32)
        n_nodes, n_tokens = ch.shape
        ms = np.ascontiguousarray(max_score, dtype=np.float32)
        lo = np.ascontiguousarray(label_off, dtype=np.int32)
        lb = np.ascontiguousarray(labels, dtype=np.int32)
        assert len(ms) == n_nodes and len(lo) == n_nodes + 1
        h = C.c_void_p()
        self.L.check(self.L.lib.fltx_trie_create(ctx.h, n_nodes, n_tokens, _ptr(ch), _ptr(ms), _ptr(lo),
                                                 _ptr(lb) if len(lb) else None, C.byref(h)))
        self.h = h
        self.n_nodes, self.n_tokens = n_nodes, n_tokens
        _live["trie"].add(self)

    def close(self):
        if self.h:
            self.L.lib.fltx_trie_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HostTrie:
    """Host-side Trie(maxChildren, rootIdx) with insert/search/smear
    (decoder/Trie.h:64-92); `upload` flattens it into HBM."""

    def __init__(self, max_children, root_idx, lib=None):
        self.L = lib or default_lib()
        h = C.c_void_p()
        self.L.check(self.L.lib.fltx_htrie_create(max_children, root_idx, C.byref(h)))
        self.h = h

    def insert(self, indices, label, score):
        a = np.ascontiguousarray(indices, dtype=np.int32)
        self.L.check(self.L.lib.fltx_htrie_insert(self.h, _ptr(a), len(a), int(label), float(score)))

    def insert_many(self, spell_flat, spell_off, labels, scores):
        sf = np.ascontiguousarray(spell_flat, dtype=np.int32)
        f = self.L.lib.fltx_htrie_insert
        base = sf.ctypes.data
        for w in range(len(spell_off) - 1):
            a, b = int(spell_off[w]), int(spell_off[w + 1])
            self.L.check(f(self.h, base + 4 * a, b - a, int(labels[w]), float(scores[w])))

    def search(self, indices):
        a = np.ascontiguousarray(indices, dtype=np.int32)
        found, nl = C.c_int32(0), C.c_int32(0)
        ms = C.c_float(0)
        labels = np.zeros(6, dtype=np.int32)
        scores = np.zeros(6, dtype=np.float32)
        self.L.check(self.L.lib.fltx_htrie_search(self.h, _ptr(a), len(a), C.addressof(found),
                                                  C.addressof(ms), C.addressof(nl), _ptr(labels),
                                                  _ptr(scores)))
        if not found.value:
            return None
        return {"max_score": ms.value, "labels": labels[:nl.value].tolist(),
                "scores": scores[:nl.value].tolist()}

    def smear(self, mode=1):
        self.L.check(self.L.lib.fltx_htrie_smear(self.h, int(mode)))

    def num_nodes(self):
        n = C.c_int64(0)
        self.L.check(self.L.lib.fltx_htrie_num_nodes(self.h, C.addressof(n)))
        return n.value

    def upload(self, ctx):
        h = C.c_void_p()
        self.L.check(self.L.lib.fltx_htrie_upload(self.h, ctx.h, C.byref(h)))
        t = Trie.__new__(Trie)
        t.ctx, t.L, t.h = ctx, ctx.L, h
        t.n_nodes, t.n_tokens = self.num_nodes(), None
        _live["trie"].add(t)
        return t

    def close(self):
        if self.h:
            self.L.lib.fltx_htrie_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Hyp:
    __slots__ = ("score", "am", "lm", "tokens", "words")

    def __init__(self, score, am, lm, tokens, words):
        self.score, self.am, self.lm = score, am, lm
        self.tokens, self.words = tokens, words


_U8_TO_I32 = np.arange(256, dtype=np.int32)
_U8_TO_I32[255] = -1  # (fltx_result_fetch_batch_compact: 0xFF = -1)


class _CompactHyp(Hyp):
    """A hypothesis of results_batch(): its token row stays the byte row the device packed until somebody reads it
    (an n-best of 50 x 1 002 frames is read in full by few callers; widening 12.8 M bytes per batch up front cost more
    than everything else on the way to Python objects)."""
    __slots__ = ("_t8", "_tok")

    def __init__(self, score, am, lm, t8, words):
        self.score, self.am, self.lm, self.words = score, am, lm, words
        self._t8, self._tok = t8, None

    @property
    def tokens(self):
        t = self._tok
        if t is None:
            t = self._tok = _U8_TO_I32[self._t8]
        return t


class _CompactHypSmall(_CompactHyp):
    """... over at most 128 tokens: 0xFF read as a signed byte IS -1, so widening is one astype."""
    __slots__ = ()

    @property
    def tokens(self):
        t = self._tok
        if t is None:
            t = self._tok = self._t8.view(np.int8).astype(np.int32)
        return t


class BatchDecoder:
    """fltx_decoder: batched LexiconFreeDecoder / LexiconDecoder on the device."""

    def __init__(self, ctx, kind, options, lm, sil, blank, unk=-1, trie=None, transitions=None,
                 is_lm_token=False):
        self.ctx, self.L = ctx, ctx.L
        self.kind, self.options = kind, options
        self._keep = (lm, trie)
        tr = None if transitions is None or len(transitions) == 0 else \
            np.ascontiguousarray(transitions, dtype=np.float32)
        h = C.c_void_p()
        self.L.check(self.L.lib.fltx_decoder_create(
            ctx.h, kind, C.byref(options), trie.h if trie is not None else None, lm.h, sil, blank, unk,
            _ptr(tr), 0 if tr is None else tr.size, int(is_lm_token), C.byref(h)))
        self.h = h
        self.B = 0
        self.N = None  # token-set size of the last offline batch
        _live["dec"].add(self)

    def _chk(self, rc):
        """Like Lib.check; a failure reported by a host-LM callback re-raises what the user's LM raised."""
        lm = self._keep[0]
        if rc == 7 and getattr(lm, "error", None) is not None:
            e, lm.error = lm.error, None
            raise e
        self.L.check(rc)

    def set(self, key, value):
        self.L.check(self.L.lib.fltx_decoder_set(self.h, key.encode(), int(value)))

    def decode_batch(self, emissions, T, N, offsets=None, device_ptr=None):
        """emissions: host float32 array (any shape, flat layout) or None when
        device_ptr (int) addresses HBM-resident emissions."""
        T = np.ascontiguousarray(T, dtype=np.int32)
        B = len(T)
        if offsets is None:
            offsets = np.concatenate([[0], np.cumsum(T.astype(np.int64) * N)[:-1]]).astype(np.int64)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        if device_ptr is not None:
            self._chk(self.L.lib.fltx_decode_batch(self.h, device_ptr, 1, _ptr(offsets), _ptr(T), B, N))
        else:
            e = np.ascontiguousarray(emissions, dtype=np.float32)
            self._chk(self.L.lib.fltx_decode_batch(self.h, _ptr(e), 0, _ptr(offsets), _ptr(T), B, N))
        self.B = B
        self.N = N

    def stream_begin(self, B, N, max_frames):
        self._chk(self.L.lib.fltx_stream_begin(self.h, B, N, max_frames))
        self.B = B
        self.N = None  # (streams: fltx_result_fetch per utterance)
        self._N = N

    def stream_step(self, emissions, T, offsets=None, device_ptr=None):
        """emissions: host float32 array, or None when device_ptr (int) addresses the chunk in HBM (the buffer
        is the caller's again as soon as the call returns)."""
        T = np.ascontiguousarray(T, dtype=np.int32)
        if offsets is None:
            offsets = np.concatenate([[0], np.cumsum(T.astype(np.int64) * self._N)[:-1]]).astype(np.int64)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        if device_ptr is not None:
            self._chk(self.L.lib.fltx_stream_step(self.h, device_ptr, 1, _ptr(offsets), _ptr(T)))
            return
        e = np.ascontiguousarray(emissions, dtype=np.float32)
        self._chk(self.L.lib.fltx_stream_step(self.h, _ptr(e), 0, _ptr(offsets), _ptr(T)))

    def stream_end(self):
        self._chk(self.L.lib.fltx_stream_end(self.h))

    def stream_prune(self, look_back=0):
        self._chk(self.L.lib.fltx_stream_prune(self.h, look_back))

    def frames_in_buffer(self, b):
        n = C.c_int32(0)
        self.L.check(self.L.lib.fltx_stream_frames_in_buffer(self.h, b, C.addressof(n)))
        return n.value

    def count(self, b):
        n, ln = C.c_int32(0), C.c_int32(0)
        self.L.check(self.L.lib.fltx_result_count(self.h, b, C.addressof(n), C.addressof(ln)))
        return n.value, ln.value

    def results(self, b, max_hyp=None):
        n, ln = self.count(b)
        if max_hyp is not None:
            n = min(n, max_hyp)
        if n == 0:
            return []
        scores = np.zeros(3 * n, dtype=np.float64)
        tokens = np.zeros((n, ln), dtype=np.int32)
        words = np.zeros((n, ln), dtype=np.int32)
        got = C.c_int32(0)
        self.L.check(self.L.lib.fltx_result_fetch(self.h, b, n, _ptr(scores), _ptr(tokens), _ptr(words),
                                                  C.addressof(got)))
        assert got.value == n
        return [Hyp(scores[3 * i], scores[3 * i + 1], scores[3 * i + 2], tokens[i].copy(), words[i].copy())
                for i in range(n)]

    def fetch_batch_raw(self):
        """The six pointers of fltx_result_fetch_batch (n_hyp, length, scores, tokens, words, offsets)."""
        ptrs = [C.c_void_p() for _ in range(6)]
        self.L.check(self.L.lib.fltx_result_fetch_batch(self.h, *[C.byref(p) for p in ptrs]))
        return [p.value for p in ptrs]

    def results_arrays(self):
        """n-best of every utterance of the last decode_batch as NumPy arrays over the
        decoder's pinned host buffers (one transfer per array; valid until the next decode):
        n_hyp [B], length [B], scores [B, K, 3] (score, emitting-model score, LM score),
        tokens / words: flat int32 with offsets [B + 1] -- hypothesis i of utterance b is
        tokens[offsets[b] + i * length[b] : offsets[b] + (i + 1) * length[b]]."""
        pn, pl, ps, pt, pw, po = (C.c_void_p() for _ in range(6))
        self.L.check(self.L.lib.fltx_result_fetch_batch(self.h, C.byref(pn), C.byref(pl), C.byref(ps), C.byref(pt),
                                                        C.byref(pw), C.byref(po)))
        B = self.B

        def view(ptr, ctype, n):
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(n,))
        off = view(po, C.c_int64, B + 1)
        total = int(off[B])
        K = int(self.options.beam_size)
        return {"n_hyp": view(pn, C.c_int32, B), "length": view(pl, C.c_int32, B), "offsets": off,
                "scores": view(ps, C.c_double, B * K * 3).reshape(B, K, 3),
                "tokens": view(pt, C.c_int32, max(total, 1)),
                "words": view(pw, C.c_int32, max(total, 1)) if pw.value else None}

    def results_arrays_compact(self):
        """The same through fltx_result_fetch_batch_compact: only the rows of the hypotheses that exist cross
        PCIe, tokens as uint8 (0xFF = -1), words as int32 rows; `offsets` [B + 1] index both flat arrays.
        tokens_of(r, b, i) / words_of(r, b, i) below widen one hypothesis' rows on demand."""
        pn, pl, ps, pt, pw, po = (C.c_void_p() for _ in range(6))
        self.L.check(self.L.lib.fltx_result_fetch_batch_compact(self.h, C.byref(pn), C.byref(pl), C.byref(ps),
                                                                C.byref(pt), C.byref(pw), C.byref(po)))
        B = self.B

        def view(ptr, ctype, n):
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(n,))
        off = view(po, C.c_int64, B + 1)
        total = int(off[B])
        K = int(self.options.beam_size)
        return {"n_hyp": view(pn, C.c_int32, B), "length": view(pl, C.c_int32, B), "offsets": off,
                "scores": view(ps, C.c_double, B * K * 3).reshape(B, K, 3),
                "tokens_u8": view(pt, C.c_uint8, max(total, 1)),
                "words": view(pw, C.c_int32, max(total, 1)) if pw.value else None}

    @staticmethod
    def tokens_of(r, b, i):
        L, o = int(r["length"][b]), int(r["offsets"][b])
        t = r["tokens_u8"][o + i * L:o + (i + 1) * L].astype(np.int32)
        t[t == 255] = -1
        return t

    @staticmethod
    def words_of(r, b, i):
        L, o = int(r["length"][b]), int(r["offsets"][b])
        return r["words"][o + i * L:o + (i + 1) * L] if r["words"] is not None else np.full(L, -1, dtype=np.int32)

    def results_batch(self, max_hyp=None):
        """[[Hyp]] for every utterance: Python objects over the arrays of results_arrays_compact()
        (or results_arrays() for token sets that do not fit a byte)."""
        compact = self.N is not None and self.N < 255
        if compact:
            r = self.results_arrays_compact()
            nh, ln, off, sc, wrd = r["n_hyp"], r["length"], r["offsets"], r["scores"], r["words"]
            total = int(off[self.B])
            tok = r["tokens_u8"][:total].copy()  # (the library's buffer is the next batch's too)
            wrd = wrd[:total].copy() if wrd is not None else None
        else:
            r = self.results_arrays()
            nh, ln, off, sc, tok, wrd = r["n_hyp"], r["length"], r["offsets"], r["scores"], r["tokens"], r["words"]
        make = (_CompactHypSmall if self.N <= 128 else _CompactHyp) if compact else Hyp
        no_words = {}
        out = []
        for b in range(self.B):
            n = int(nh[b]) if max_hyp is None else min(int(nh[b]), max_hyp)
            L = int(ln[b])
            o = int(off[b])
            tb = tok[o:o + n * L].reshape(n, L)
            if wrd is not None:
                wb = wrd[o:o + n * L].reshape(n, L)
            else:  # lexicon-free: every word slot is -1 (LexiconFreeDecoder.h:80-82) -- one read-only row per length
                row = no_words.get(L)
                if row is None:
                    row = no_words[L] = np.full(L, -1, dtype=np.int32)
                    row.flags.writeable = False
                wb = None
            s3 = sc[b, :n].tolist()
            out.append([make(s3[i][0], s3[i][1], s3[i][2], tb[i], wb[i] if wb is not None else row) for i in range(n)])
        return out

    def best(self, b, look_back=0, capacity=1 << 16):
        scores = np.zeros(3, dtype=np.float64)
        tokens = np.zeros(capacity, dtype=np.int32)
        words = np.zeros(capacity, dtype=np.int32)
        ln = C.c_int32(0)
        self.L.check(self.L.lib.fltx_result_best(self.h, b, look_back, _ptr(scores), _ptr(tokens),
                                                 _ptr(words), capacity, C.addressof(ln)))
        n = ln.value
        return Hyp(scores[0], scores[1], scores[2], tokens[:n].copy(), words[:n].copy())

    def get(self, key):
        v = C.c_int64(0)
        self.L.check(self.L.lib.fltx_decoder_get(self.h, key.encode(), C.addressof(v)))
        return v.value

    def stats(self):
        fr, by = C.c_int64(0), C.c_int64(0)
        th, lds = C.c_int32(0), C.c_int32(0)
        self.L.check(self.L.lib.fltx_decoder_stats(self.h, C.addressof(fr), C.addressof(by),
                                                   C.addressof(th), C.addressof(lds)))
        return {"frames": fr.value, "algorithmic_bytes": by.value, "threads_per_utt": th.value,
                "lds_bytes": lds.value}

    def bytes(self):
        """Algorithmic bytes of the last offline decode, split by kernel."""
        a, b, c = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        self.L.check(self.L.lib.fltx_decoder_bytes(self.h, C.addressof(a), C.addressof(b), C.addressof(c)))
        return {"decode": a.value, "epilogue": b.value, "lm": c.value}

    def profile(self):
        out = np.zeros(8, dtype=np.uint64)
        self.L.check(self.L.lib.fltx_decoder_profile(self.h, _ptr(out)))
        return out

    def timing(self):
        a, b = C.c_float(0), C.c_float(0)
        self.L.check(self.L.lib.fltx_decoder_timing(self.h, C.addressof(a), C.addressof(b)))
        return a.value, b.value

    def close(self):
        if self.h:
            self.L.lib.fltx_decoder_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DecoderGroup:
    """fltx_group: one batch sharded over several devices (one context, decoder
    and host thread per entry of `devices`; no inter-device traffic)."""

    def __init__(self, devices, kind, options, lm, sil, blank, unk=-1, host_trie=None, transitions=None,
                 is_lm_token=False, lib=None):
        self.L = lib or lm.L
        self.options = options
        self._keep = (lm, host_trie)
        dv = np.ascontiguousarray(devices, dtype=np.int32)
        tr = None if transitions is None or len(transitions) == 0 else \
            np.ascontiguousarray(transitions, dtype=np.float32)
        h = C.c_void_p()
        self.L.check(self.L.lib.fltx_group_create(
            _ptr(dv), len(dv), kind, C.byref(options), host_trie.h if host_trie is not None else None, lm.h,
            sil, blank, unk, _ptr(tr), 0 if tr is None else tr.size, int(is_lm_token), C.byref(h)))
        self.h = h
        self.n = len(dv)
        self.B = 0

    def decode_batch(self, emissions, T, N, offsets=None, device_ptrs=None):
        """emissions: one host float32 array for the whole batch, or device_ptrs =
        one HBM address per device (the part's shard, addressed by `offsets`)."""
        T = np.ascontiguousarray(T, dtype=np.int32)
        B = len(T)
        off = None if offsets is None else np.ascontiguousarray(offsets, dtype=np.int64)
        ptrs = (C.c_void_p * self.n)()
        ond = np.zeros(self.n, dtype=np.int32)
        if device_ptrs is not None:
            for i, p in enumerate(device_ptrs):
                ptrs[i] = p
            ond[:] = 1
        else:
            self._e = np.ascontiguousarray(emissions, dtype=np.float32)
            for i in range(self.n):
                ptrs[i] = self._e.ctypes.data
        self.L.check(self.L.lib.fltx_group_decode_batch(self.h, ptrs, _ptr(ond), _ptr(off), _ptr(T), B, N))
        self.B = B

    def parts(self):
        """[(BatchDecoder-like handle, first, count)] of the last batch."""
        out = []
        for i in range(self.n):
            d, f, c = C.c_void_p(), C.c_int32(0), C.c_int32(0)
            self.L.check(self.L.lib.fltx_group_decoder(self.h, i, C.byref(d), C.addressof(f), C.addressof(c)))
            out.append((d, f.value, c.value))
        return out

    def results(self, b, max_hyp=None):
        n, ln = C.c_int32(0), C.c_int32(0)
        self.L.check(self.L.lib.fltx_group_result_count(self.h, b, C.addressof(n), C.addressof(ln)))
        n, ln = n.value, ln.value
        if max_hyp is not None:
            n = min(n, max_hyp)
        if n == 0:
            return []
        scores = np.zeros(3 * n, dtype=np.float64)
        tokens = np.zeros((n, ln), dtype=np.int32)
        words = np.zeros((n, ln), dtype=np.int32)
        got = C.c_int32(0)
        self.L.check(self.L.lib.fltx_group_result_fetch(self.h, b, n, _ptr(scores), _ptr(tokens), _ptr(words),
                                                        C.addressof(got)))
        return [Hyp(scores[3 * i], scores[3 * i + 1], scores[3 * i + 2], tokens[i].copy(), words[i].copy())
                for i in range(n)]

    def synchronize(self):
        self.L.check(self.L.lib.fltx_group_synchronize(self.h))

    def close(self):
        if self.h:
            self.L.lib.fltx_group_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def make_options(beam_size, beam_size_token, beam_threshold=25.0, lm_weight=0.0, word_score=0.0,
                 unk_score=-float("inf"), sil_score=0.0, log_add=False, criterion="ctc"):
    crit = CRITERION[criterion] if isinstance(criterion, str) else int(criterion)
    return Options(beam_size, beam_size_token, beam_threshold, lm_weight, word_score, unk_score, sil_score,
                   int(bool(log_add)), crit)
