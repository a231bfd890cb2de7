"""Synthetic measurement inputs via libfltx_synth.so.

Host-only; used by tests/, bench.py and tests/golden/make_golden.py.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "lib", "libfltx_synth.so")
_SRC = os.path.join(_HERE, "csrc", "fltx_synth.cpp")
DIST = {"uniform": 0, "ctc": 1, "lexspell": 2}
_lib = None


def build():
    os.makedirs(os.path.dirname(_LIB), exist_ok=True)
    subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", _LIB, _SRC],
                   check=True)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        _lib = C.CDLL(_LIB)
        _lib.fltx_synth_emissions.restype = C.c_int
        _lib.fltx_synth_emissions.argtypes = [
            C.c_int, C.c_uint64, C.c_uint64, C.c_int, C.c_int,
            C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        _lib.fltx_synth_lexicon.restype = C.c_int64
        _lib.fltx_synth_lexicon.argtypes = [C.c_uint64, C.c_int64, C.c_void_p,
                                            C.c_int64, C.c_void_p]
        _lib.fltx_synth_floats.restype = None
        _lib.fltx_synth_floats.argtypes = [C.c_uint64, C.c_int64, C.c_float, C.c_float, C.c_void_p]
    return _lib


def floats(seed, n, lo=0.0, hi=1.0):
    out = np.empty(n, dtype=np.float32)
    lib().fltx_synth_floats(seed, n, lo, hi, out.ctypes.data)
    return out


def emissions(dist, u, T, N, S=1, lexicon=None, out=None):
    """One utterance [T, N] float32, frame-major."""
    if out is None:
        out = np.empty((T, N), dtype=np.float32)
    sf = so = None
    W = 0
    if lexicon is not None:
        sf, so = lexicon
        W = len(so) - 1
    rc = lib().fltx_synth_emissions(
        DIST[dist], S, u, T, N,
        sf.ctypes.data if sf is not None else None,
        so.ctypes.data if so is not None else None, W, out.ctypes.data)
    if rc != 0:
        raise ValueError("fltx_synth_emissions failed")
    return out


def batch(dist, B, T, N, S=1, lexicon=None, u0=0):
    out = np.empty((B, T, N), dtype=np.float32)
    for b in range(B):
        emissions(dist, u0 + b, T, N, S, lexicon, out[b])
    return out


def lexicon(W=90000, seed=4242):
    """(spell_flat int32[], spell_off int64[W+1]); word id = rank."""
    cap = W * 13
    sf = np.empty(cap, dtype=np.int32)
    so = np.empty(W + 1, dtype=np.int64)
    n = lib().fltx_synth_lexicon(seed, W, sf.ctypes.data, cap, so.ctypes.data)
    if n < 0:
        raise RuntimeError("fltx_synth_lexicon overflow")
    return sf[:n].copy(), so
