import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import orclib
    return orclib.load("oracle")


@pytest.fixture(scope="session")
def golden():
    import helpers
    return helpers.load_golden()


@pytest.fixture(scope="session")
def emu_session():
    """Host-thread emulation of the kernels (logic check without a GPU)."""
    import subprocess
    import helpers
    src = [os.path.join(ROOT, "text_amd", "csrc", f) for f in
           ("fltx_api.cpp", "fltx_kernels.h", "fltx_lean.h", "fltx_lane.h", "fltx_slane.h", "fltx_rt.h",
            "fltx_host_trie.cpp", "fltx_arpa.cpp", "fltx_group.cpp", "fltx_kernel_entry.h")] + \
          [os.path.join(ROOT, "tests", "emu", f) for f in ("hip_emu.h", "hip_emu.cpp")]
    if (not os.path.exists(helpers.EMU_LIB) or
            os.path.getmtime(helpers.EMU_LIB) < max(os.path.getmtime(s) for s in src)):
        subprocess.run([os.path.join(ROOT, "tests", "emu", "build.sh")], check=True)
    return helpers.FltxSession(helpers.EMU_LIB)


@pytest.fixture(scope="session")
def gpu_session():
    """The product path: text_amd/lib/libfltx.so on cuda:0.  Fails loudly when
    the HIP library is missing -- there is no fallback."""
    import helpers
    return helpers.FltxSession(None)
