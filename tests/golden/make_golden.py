#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the UNMODIFIED
reference (oracle/_ref/libfltref.so, built by `make -C oracle ref` from the
sources under /root/reference).  Runs only in the dev container.

Outputs (all data, no reference source text):
  synthetic_expected.json.gz   n-best of every case in tests/cases.py, run
                               TWICE under different heap layouts (must agree:
                               the reference is only deterministic on tie-free
                               inputs)
  decodertest/                 the data files the reference's own DecoderTest
                               reads (emission/transition/TN/letters/words/lm),
                               gzip'd, plus
  decodertest/lexicon_dump.txt.gz   word ids + spellings as produced by the
                               reference's loadWords/createWordDict/tkn2Idx
  decodertest/expected.json    LM / trie known answers and the n-best of the
                               DecoderTest flow as computed by the reference
"""
import ctypes
import gzip
import json
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import cases  # noqa: E402
import helpers  # noqa: E402
from oracle import orclib  # noqa: E402

REF_DATA = "/root/reference/flashlight/lib/text/test/decoder/data"


def perturb_heap(n):
    libc = ctypes.CDLL(None)
    libc.malloc.restype = ctypes.c_void_p
    keep = [libc.malloc(24 + (i * 37) % 200) for i in range(n)]
    return keep  # leaked on purpose for the lifetime of the process


def synthetic(ref):
    out = {}
    only = [x for x in os.environ.get("GOLDEN_ONLY", "").split(",") if x]
    if only:  # add cases without touching the vectors already committed
        with gzip.open(os.path.join(HERE, "synthetic_expected.json.gz"), "rt") as f:
            out = json.load(f)
    for c in cases.CASES:
        if only and c["name"] not in only:
            continue
        inp = helpers.case_inputs(c)
        h1 = helpers.run_checker(ref, c, inp)
        keep = perturb_heap(33333)
        h2 = helpers.run_checker(ref, c, inp)
        ok, why = helpers.hyps_equal(h1, h2)
        if not ok:
            raise SystemExit("reference is heap-layout dependent on %s: %s" % (c["name"], why))
        # no adjacent equal scores in the n-best (tie-free requirement)
        for a, b in zip(h1, h1[1:]):
            if a.score == b.score:
                raise SystemExit("tie in n-best of %s" % c["name"])
        full = c["size"] != "large"
        out[c["name"]] = helpers.encode_hyps(h1, full)
        if c["name"] in cases.APPENDIX_B:
            n, hsh, top = cases.APPENDIX_B[c["name"]]
            assert len(h1) == n and orclib.nbest_hash(h1) == hsh and h1[0].score == float.fromhex(top), \
                "Appendix B mismatch on " + c["name"]
        print("%-28s n=%3d hash=%s top=%s" % (c["name"], len(h1), out[c["name"]]["hash"],
                                              float(h1[0].score).hex() if h1 else "-"))
        del keep
    with gzip.open(os.path.join(HERE, "synthetic_expected.json.gz"), "wt") as f:
        json.dump(out, f, separators=(",", ":"))


def decodertest(ref):
    dst = os.path.join(HERE, "decodertest")
    os.makedirs(dst, exist_ok=True)
    for name in ("TN.bin", "emission.bin", "transition.bin", "letters.lst", "words.lst", "lm.arpa"):
        with open(os.path.join(REF_DATA, name), "rb") as fi, \
                gzip.GzipFile(os.path.join(dst, name + ".gz"), "wb", 9, mtime=0) as fo:
            shutil.copyfileobj(fi, fo)
    dump = ref.lexicon_dump(os.path.join(REF_DATA, "words.lst"), os.path.join(REF_DATA, "letters.lst"),
                            "<1>", 1)
    with gzip.GzipFile(os.path.join(dst, "lexicon_dump.txt.gz"), "wb", 9, mtime=0) as fo:
        fo.write(dump.encode())
    lex = parse_lexicon_dump(dump)
    lex["letters"] = open(os.path.join(REF_DATA, "letters.lst")).read().split() + ["<1>"]
    exp = run_decodertest(ref, lex, REF_DATA + "/lm.arpa",
                          np.fromfile(REF_DATA + "/TN.bin", dtype=np.int32),
                          np.fromfile(REF_DATA + "/emission.bin", dtype=np.float32),
                          np.fromfile(REF_DATA + "/transition.bin", dtype=np.float32))
    # the reference's own assertions (DecoderTest.cpp:110-120,148-155,184-194)
    tgt_lm = [-1.05971, -4.19448, -3.33383, -2.76726, -1.16237, -4.64589]
    tgt_trie = [-1.05971, -2.87742, -2.64553, -3.05081, -1.05971, -3.08968]
    tgt_hyp = [-284.0998, -284.108, -284.119, -284.127, -284.296]
    assert all(abs(a - b) < 1e-5 for a, b in zip(exp["lm_scores"], tgt_lm))
    assert abs(exp["lm_total"] - (-19.5123)) < 1e-4
    assert all(abs(a - b) < 1e-5 for a, b in zip(exp["trie_scores"], tgt_trie))
    assert exp["nbest"]["n"] == 16
    assert all(abs(float.fromhex(s[0]) - t) < 1e-3 for s, t in zip(exp["nbest"]["scores"], tgt_hyp))
    with open(os.path.join(dst, "expected.json"), "w") as f:
        json.dump(exp, f, separators=(",", ":"))
    print("decodertest: n_hyp=%d top=%s" % (exp["nbest"]["n"], exp["nbest"]["scores"][0][0]))


def parse_lexicon_dump(dump):
    head, rest = dump.split("\n", 1)
    ntok, nword, sil, unk = map(int, head.split())
    body, words = rest.split("#words\n")
    words = words.strip("\n").split("\n")
    entries = []
    for line in body.strip("\n").split("\n"):
        wi, w, sp = line.split("\t")
        entries.append((int(wi), w, [int(x) for x in sp.split()]))
    return dict(ntok=ntok, nword=nword, sil=sil, unk=unk, words=words, entries=entries)


def run_decodertest(lib, lex, arpa_path, TN, em, tr):
    """DecoderTest.cpp:57-195 through a checker library."""
    T, N = int(TN[0]), int(TN[1])
    widx = {w: i for i, w in enumerate(lex["words"])}
    lm = lib.lm_arpa_create(arpa_path.encode(), "\n".join(lex["words"]).encode())
    sent = np.array([widx[w] for w in "the cat sat on the mat".split()], dtype=np.int32)
    per = np.zeros(6, dtype=np.float32)
    total = lib.lm_score_sequence(lm, orclib._ip(sent), 6, 1, orclib._fp(per))
    trie = lib.trie_create(lex["ntok"], lex["sil"])
    one = np.zeros(1, dtype=np.float32)
    word_score = {}
    for wi, w, sp in lex["entries"]:
        if wi not in word_score:
            s1 = np.array([wi], dtype=np.int32)
            lib.lm_score_sequence(lm, orclib._ip(s1), 1, 0, orclib._fp(one))
            word_score[wi] = float(one[0])
        a = np.array(sp, dtype=np.int32)
        lib.trie_insert(trie, orclib._ip(a), len(sp), wi, word_score[wi])
    lib.trie_smear(trie, 1)
    # DecoderTest.cpp:37-49 tokens2Tensor: the word's letters, no trailing "|"
    letters = lex["letters"]
    trie_scores = []
    ms = ctypes.c_float(0)
    for w in "the cat sat on the mat".split():
        a = np.array([letters.index(ch) for ch in w], dtype=np.int32)
        assert lib.trie_search(trie, orclib._ip(a), len(a), ctypes.byref(ms), None) == 1
        trie_scores.append(ms.value)
    opt = orclib.make_options(2500, 25000, 100.0, 2.0, 2.0, -float("inf"), -1.0, False, "asg")
    dec = lib.lexicon(opt, trie, lm, lex["sil"], -1, lex["unk"], tr, False)
    hyps = lib.decode(dec, em, T, N)
    lib.decoder_destroy(dec)
    lib.trie_destroy(trie)
    lib.lm_destroy(lm)
    return dict(lm_scores=[float(x) for x in per], lm_total=float(total), trie_scores=trie_scores,
                word_scores_hex={str(k): float(np.float32(v)).hex() for k, v in sorted(word_score.items())
                                 if k < 50},
                nbest=helpers.encode_hyps(hyps, True))


if __name__ == "__main__":
    orclib.build("ref")
    ref_lib = orclib.load("ref")
    synthetic(ref_lib)
    if not os.environ.get("GOLDEN_ONLY"):
        decodertest(ref_lib)
