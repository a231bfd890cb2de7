"""User-defined LMs for the host-LM tests (tests/test_host_lm.py): plain Python classes with the reference's LM
interface (decoder/lm/LM.h:61-85) -- what a user of the reference's PyLM trampoline
(bindings/python/flashlight/lib/text/_decoder.cpp:39-56) writes."""
import numpy as np


class State:
    """LMState with the child memo of decoder/lm/LM.h:24-34."""
    __slots__ = ("children", "data")

    def __init__(self, data=None):
        self.children = {}
        self.data = data

    def child(self, idx):
        c = self.children.get(idx)
        if c is None:
            c = self.children[idx] = State()
        return c


class PyZeroLM:
    """decoder/lm/ZeroLM.cpp:14-26 written against the Python interface."""

    def start(self, start_with_nothing):
        return State()

    def score(self, state, idx):
        return state.child(idx), 0.0

    def finish(self, state):
        return state, 0.0


class PyNgramLM:
    """The KenLM adapter (decoder/lm/KenLM.cpp:52-83) written against the Python interface over the product's host
    copy of the flat n-gram tables (`arpa` = text_amd._capi.ArpaLM: explicit-state start / step)."""

    def __init__(self, arpa):
        self.arpa = arpa
        self.calls = 0

    def start(self, start_with_nothing):
        return State(self.arpa.start(start_with_nothing))

    def score(self, state, idx):
        self.calls += 1
        ctx, sc = self.arpa.step(state.data, idx)
        out = state.child(idx)
        out.data = ctx
        return out, sc

    def finish(self, state):
        ctx, sc = self.arpa.step(state.data, -1)
        out = state.child(-1)
        out.data = ctx
        return out, sc


def pair_score(prev, w, seed):
    """Exact in float32 on every host: a 16-bit integer over 2^13 (no libm)."""
    h = (prev * 1000003 + w * 7919 + seed) & 0xFFFFFFFF
    h ^= h >> 13
    h = (h * 0x5BD1E995) & 0xFFFFFFFF
    h ^= h >> 15
    return -float(h & 0xFFFF) / 8192.0


class LastWordLM:
    """An LM whose state is its last input only: ONE state object per index, shared by every history that ends in
    it (a bigram LM).  Hypotheses with different histories then share an LM state and merge -- by the ADDRESS of the
    state object in the reference (decoder/lm/LM.h:37-49), by the id the host gives that object here.  The
    reference-side twin is oracle/ref_driver.cpp's LastWordRefLM; golden vectors come from it."""

    def __init__(self, n_idx, seed):
        self.seed = seed
        self.begin = State(-1)
        self.states = [State(i) for i in range(n_idx)]

    def start(self, start_with_nothing):
        return self.begin

    def score(self, state, idx):
        return self.states[idx], pair_score(state.data + 2, idx + 2, self.seed)

    def finish(self, state):
        return state, pair_score(state.data + 2, 1, self.seed)


class FailingLM(PyZeroLM):
    def __init__(self, after):
        self.left = after

    def score(self, state, idx):
        self.left -= 1
        if self.left < 0:
            raise KeyError("user LM failed on purpose")
        return state.child(idx), 0.0
