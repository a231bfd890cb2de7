"""CPU (not gpu): the C-ABI shared library loads and exports every symbol that
include/fltx.h declares; host-only entry points (trie builder, error paths)
behave like the reference's.  No decode is run here -- that needs the GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

import helpers
from text_amd import _capi

HEADER = os.path.join(helpers.ROOT, "include", "fltx.h")


def _declared():
    txt = open(HEADER).read()
    return sorted(set(re.findall(r"FLTX_API\s+[\w\s\*]+?\b(fltx_\w+)\s*\(", txt)))


@pytest.fixture(scope="module")
def lib():
    path = _capi.DEFAULT_LIB
    if not os.path.exists(path):
        import __graft_entry__
        __graft_entry__.build()
    return _capi.Lib(path)


def test_every_declared_symbol_is_exported(lib):
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib.lib, n), "libfltx.so does not export " + n
    assert set(names) == set(_capi.Lib.SYMBOLS)


def test_host_trie_matches_reference_semantics(lib):
    t = _capi.HostTrie(5, 0, lib=lib)
    t.insert([1, 2], 10, -1.0)
    t.insert([1, 2], 11, -2.0)
    t.insert([1, 3, 0], 12, -0.5)
    with pytest.raises(IndexError):  # Trie.cpp:31-34 throws std::out_of_range
        t.insert([1, 7], 13, 0.0)
    for i in range(8):  # kTrieMaxLabel = 6: extra labels are dropped (Trie.cpp:40-46)
        t.insert([4], 100 + i, -3.0)
    assert len(t.search([4])["labels"]) == 6
    assert t.search([2]) is None
    t.smear(1)
    # node [1,2]: own scores are log-added even in MAX mode (Trie.cpp:80-83)
    want = np.float32(np.logaddexp(-1.0, -2.0))
    assert np.float32(t.search([1, 2])["max_score"]) == want
    assert np.float32(t.search([1])["max_score"]) == np.float32(-0.5)
    assert t.num_nodes() == 6


def test_no_device_fails_loudly(lib):
    """Without a GPU the context cannot be created and says so (no CPU path)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(_capi.FltxError) as ei:
        _capi.Context(lib=lib)
    assert "no CPU path" in str(ei.value) or "HIP" in str(ei.value)


def test_bench_finds_the_committed_hbm_traffic_of_its_default_workload():
    """bench.py quotes roofline.traffic from the rocprofv3 --pmc summary committed
    under profiles/ -- keyed by the exact geometry it was measured on."""
    import bench
    tr = bench.pmc_traffic("C2", 576, 256, 1000, 29, 50, engine=4)
    assert tr is not None and tr[0] > 1e8 and tr[1].startswith("profiles/r0")  # (the newest round that holds the key)
    assert bench.pmc_traffic("C2", 256, 256, 1000, 29, 50, engine=4) is None  # other geometry: not quoted
    assert bench.pmc_traffic("C2", 576, 256, 1000, 29, 50, engine=3) is None  # other engine: not quoted
    assert bench.pmc_traffic("C3", 512, 256, 1000, 29, 50, engine=5) is not None  # fltx_xlane.h
    assert bench.pmc_traffic("C4", 768, 256, 1500, 29, 100, engine=6) is not None  # fltx_ylane.h
    assert bench.pmc_traffic("C5", 512, 1024, 1500, 29, 100, engine=6) is not None  # C5's share: two workgroups per CU
    assert bench.pmc_traffic("C4", 512, 256, 1500, 29, 100, engine=0) is None  # the generic engine was not re-measured
