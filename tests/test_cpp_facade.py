"""The fl::lib::text C++ facade (text_amd/csrc/flashlight/lib/text/...): the
reference's DecoderTest.cpp flow re-run in C++ against this repo's classes."""
import gzip
import os
import subprocess

import pytest

import helpers

SRC = os.path.join(helpers.ROOT, "tests", "cpp", "decoder_test.cpp")
EXE = os.path.join(helpers.ROOT, "tests", "cpp", "decoder_test")


def _build():
    import glob
    deps = [SRC, os.path.join(helpers.ROOT, "include", "fltx.h")] + \
        glob.glob(os.path.join(helpers.ROOT, "text_amd", "csrc", "flashlight", "**", "*.h"), recursive=True)
    if os.path.exists(EXE) and os.path.getmtime(EXE) > max(os.path.getmtime(p) for p in deps):  # (the facade is header-only)
        return
    subprocess.run(["g++", "-std=c++17", "-O1", "-I" + os.path.join(helpers.ROOT, "text_amd", "csrc"),
                    "-I" + os.path.join(helpers.ROOT, "include"), SRC,
                    "-L" + os.path.join(helpers.ROOT, "text_amd", "lib"), "-lfltx",
                    "-Wl,-rpath," + os.path.join(helpers.ROOT, "text_amd", "lib"), "-o", EXE], check=True)


def test_facade_compiles_against_reference_call_sequence():
    """CPU: the facade headers compile and link (no device needed for that)."""
    _build()
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_decodertest_in_cpp(tmp_path):
    _build()
    d = os.path.join(helpers.GOLDEN_DIR, "decodertest")
    for name in ("TN.bin", "emission.bin", "transition.bin", "lm.arpa", "lexicon_dump.txt"):
        (tmp_path / name).write_bytes(gzip.open(os.path.join(d, name + ".gz"), "rb").read())
    r = subprocess.run([EXE, str(tmp_path)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                       timeout=600)
    assert r.returncode == 0 and "PASSED" in r.stdout, r.stdout[-3000:]


def test_lexicon_loader_matches_reference_dump(tmp_path):
    """CPU: loadWords / createWordDict / Dictionary(file) / tkn2Idx of the
    facade (dictionary/Utils.h) reproduce, byte for byte, what the reference
    produced for its own words.lst / letters.lst (word ids follow the
    unordered_map iteration order, dictionary/Utils.cpp:19-26)."""
    src = os.path.join(helpers.ROOT, "tests", "cpp", "lexicon_test.cpp")
    exe = os.path.join(helpers.ROOT, "tests", "cpp", "lexicon_test")
    subprocess.run(["g++", "-std=c++17", "-O1", "-I" + os.path.join(helpers.ROOT, "text_amd", "csrc"),
                    "-I" + os.path.join(helpers.ROOT, "include"), src, "-o", exe], check=True)
    d = os.path.join(helpers.GOLDEN_DIR, "decodertest")
    for name in ("words.lst", "letters.lst", "lexicon_dump.txt"):
        (tmp_path / name).write_bytes(gzip.open(os.path.join(d, name + ".gz"), "rb").read())
    r = subprocess.run([exe, str(tmp_path / "words.lst"), str(tmp_path / "letters.lst"),
                        str(tmp_path / "lexicon_dump.txt")], stdout=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stdout
