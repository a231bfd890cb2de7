import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # pytest.ini's `timeout` is pytest-timeout's option: without the plugin it would be ignored in silence and a
    # kernel that never returns would hold the (GPU) box.  Fall back to an alarm per test.
    config._fltx_alarm = not config.pluginmanager.hasplugin("timeout")


@pytest.hookimpl(hookwrapper=True)
def pytest_runtest_call(item):
    import signal
    use = getattr(item.config, "_fltx_alarm", False) and hasattr(signal, "SIGALRM")
    if use:
        def on_alarm(signum, frame):
            raise TimeoutError("test exceeded 900 s (pytest-timeout is not installed: conftest fallback)")
        old = signal.signal(signal.SIGALRM, on_alarm)
        signal.alarm(900)
    try:
        yield
    finally:
        if use:
            signal.alarm(0)
            signal.signal(signal.SIGALRM, old)


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
    import glob
    # every kernel header and host source the emulator is compiled from (a hand-kept list once
    # missed the lane engines' headers and the CPU suite kept testing a stale library)
    src = glob.glob(os.path.join(ROOT, "text_amd", "csrc", "fltx_*.h")) + \
        glob.glob(os.path.join(ROOT, "text_amd", "csrc", "fltx_*.cpp")) + \
        [os.path.join(ROOT, "include", "fltx.h")] + \
        glob.glob(os.path.join(ROOT, "tests", "emu", "*.h")) + \
        glob.glob(os.path.join(ROOT, "tests", "emu", "*.cpp")) + \
        glob.glob(os.path.join(ROOT, "tests", "emu", "*.inc")) + \
        [os.path.join(ROOT, "tests", "emu", "build.sh")]
    import fcntl
    with open(os.path.join(ROOT, "tests", "emu", ".build.lock"), "w") as lock:  # (pytest-xdist: one worker builds, the others wait)
        fcntl.flock(lock, fcntl.LOCK_EX)
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
