"""Shared test plumbing: materialise a case from tests/cases.py and run it through
a checker (oracle / compiled reference) or through the C ABI (HIP library, or
the host-thread emulation of the same kernels)."""
import functools
import gzip
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import orclib  # noqa: E402
from text_amd import _capi, synth  # noqa: E402

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
EMU_LIB = os.path.join(ROOT, "tests", "emu", "libfltx_emu.so")


@functools.lru_cache(maxsize=8)
def lexicon(W, seed, mode=False):
    """Synthetic lexicon.  mode True / "nodup" drops spellings with adjacent repeated letters
    (what replabels guarantee for ASG lexicons: with a doubled letter, "stay in
    the node" and "advance to the child" consume the same token at the same
    score, an exact tie whose resolution the reference leaves to nth_element).
    mode "multi" / "multi_nodup": homophones, as the reference's own test lexicon has them -- word w takes the
    spelling of word w - 1 when w % 3 == 1 and of word w - 2 when w % 9 == 2, so spellings carry one, two or three
    words (Trie.h:19 allows six); "multi6": six on every spelling."""
    sf, so = synth.lexicon(W, seed)
    if not mode:
        return sf, so
    spell = [sf[so[w]:so[w + 1]] for w in range(W)]
    if mode in (True, "nodup", "multi_nodup"):
        spell = [sp for sp in spell if not np.any(sp[:-1][1:] == sp[:-1][:-1])]
    if mode in ("multi", "multi_nodup"):
        for w in range(len(spell)):
            if w % 3 == 1:
                spell[w] = spell[w - 1]
            elif w % 9 == 2:
                spell[w] = spell[w - 2]
    if mode == "multi6":  # six words on every spelling: the most a trie node takes
        spell = [spell[w - w % 6] for w in range(len(spell))]
    if mode == "short6":  # ... on spellings of one or two letters: most lanes of a beam can end a word in every frame
        spell = [np.array(([1 + (w // 6)] if w // 6 < 24 else [1 + (w // 6) % 24, 1 + (w // 6) // 24]) + [0], dtype=np.int32)
                 for w in range(W)]
    nsf = np.concatenate(spell).astype(np.int32)
    nso = np.zeros(len(spell) + 1, dtype=np.int64)
    nso[1:] = np.cumsum([len(sp) for sp in spell])
    return nsf, nso


def case_inputs(c):
    """-> dict(emissions [T,N] f32, transitions or None, lex (sf, so) or None,
    labels, label scores)"""
    lex = lexicon(*c["lexicon"]) if c["lexicon"] else None
    e = synth.emissions(c["dist"], c["u"], c["T"], c["N"], lexicon=lex if c["dist"] == "lexspell" else None)
    tr = None
    if c["trans_seed"] is not None:
        tr = synth.floats(c["trans_seed"], c["N"] * c["N"], 0.0, 1.0)
    out = dict(e=e, tr=tr, lex=lex)
    if lex is not None:
        W = len(lex[1]) - 1
        out["W"] = W
        out["labels"] = np.arange(W, dtype=np.int32)
        if c["label_scores"] is not None:
            out["scores"] = synth.floats(c["label_scores"], W, -5.0, 0.0)
        else:
            out["scores"] = np.zeros(W, dtype=np.float32)
    return out


# ---- n-gram LM plumbing -----------------------------------------------------
NGRAM_DIR = os.environ.get("FLTX_NGRAM_CACHE", "/tmp/fltx_ngram_cache")  # synthetic ARPA files, regenerated on demand


def lm_vocab(c, inp):
    """user-index -> word list the LM is built over (KenLM.cpp:44-49)."""
    from text_amd import ngram_synth
    if c["kind"] == "lexicon" and not c["is_lm_token"]:
        return ngram_synth.words(inp["W"]) + ["<unk>"]  # word ids 0..W-1, unk = W
    return ngram_synth.words(c["N"], "t")


def arpa_path(c, inp):
    """Deterministic synthetic ARPA file for a case (cached on disk)."""
    from text_amd import ngram_synth
    _, order, seed = c["lm"]
    vocab = lm_vocab(c, inp)
    big = len(vocab) > 10000
    counts = (0, 200000, 100000, 50000) if big else (0, 3000, 1500, 800)
    os.makedirs(NGRAM_DIR, exist_ok=True)
    path = os.path.join(NGRAM_DIR, "lm_o%d_s%d_v%d.arpa" % (order, seed, len(vocab)))
    if not os.path.exists(path):
        tmp = "%s.tmp%d" % (path, os.getpid())  # (pytest-xdist workers may build the same file at the same time)
        ngram_synth.write_arpa(tmp, [w for w in vocab if w != "<unk>"], order, counts, seed)
        os.replace(tmp, path)
    return path, vocab


def lastword_size(c, inp):
    """indices a ("lastword", seed) LM is asked about: words + <unk> for a word LM, tokens otherwise"""
    return inp["W"] + 1 if (c["kind"] == "lexicon" and not c["is_lm_token"]) else c["N"]


def checker_lm(lib, c, inp):
    if c["lm"] == "zero":
        return lib.lm_zero_create()
    if c["lm"][0] == "lastword":
        return lib.lm_lastword_create(lastword_size(c, inp), c["lm"][1])
    path, vocab = arpa_path(c, inp)
    lm = lib.lm_arpa_create(path.encode(), "\n".join(vocab).encode())
    assert lm, "lm_arpa_create failed"
    return lm


def checker_word_scores(lib, lm, W):
    """trie label scores = lm.score(start, word) (DecoderTest.cpp:137-146)."""
    out = np.zeros(W, dtype=np.float32)
    one = np.zeros(1, dtype=np.float32)
    w1 = np.zeros(1, dtype=np.int32)
    for w in range(W):
        w1[0] = w
        lib.lm_score_sequence(lm, orclib._ip(w1), 1, 0, orclib._fp(one))
        out[w] = one[0]
    return out


def run_checker(lib, c, inp=None):
    """Run a case through oracle/liboracle.so or oracle/_ref/libfltref.so."""
    inp = inp or case_inputs(c)
    opt = orclib.make_options(c["K"], c["Kt"], c["thr"], c["lm_weight"], c["word_score"], c["unk_score"],
                              c["sil_score"], c["log_add"], c["crit"])
    N = c["N"]
    blank = N - 1 if c["crit"] == "ctc" else -1
    lm = checker_lm(lib, c, inp)
    trie = None
    try:
        if c["kind"] == "lexfree":
            dec = lib.lexfree(opt, lm, 0, blank, inp["tr"])
        else:
            sf, so = inp["lex"]
            scores = inp["scores"]
            if c["lm"] != "zero" and not c["is_lm_token"]:
                scores = checker_word_scores(lib, lm, inp["W"])
            trie = lib.build_trie(N, 0, sf, so, inp["labels"], scores, smear=1)
            dec = lib.lexicon(opt, trie, lm, 0, blank, inp["W"], inp["tr"], c["is_lm_token"])
        hyps = lib.decode(dec, inp["e"], c["T"], N)
        lib.decoder_destroy(dec)
    finally:
        if trie is not None:
            lib.trie_destroy(trie)
        lib.lm_destroy(lm)
    return hyps


class FltxSession:
    """Context + cached tries for one loaded libfltx (HIP or emulation)."""

    def __init__(self, lib_path=None, device=-1):
        self.lib = _capi.Lib(lib_path) if lib_path else _capi.default_lib()
        self.ctx = _capi.Context(device=device, lib=self.lib)
        self.zero = _capi.ZeroLM(self.ctx)
        self._tries = {}
        self._lms = {}

    def lm_for(self, c, inp):
        if c["lm"] == "zero":
            return self.zero
        if c["lm"][0] == "lastword":  # a user-defined LM: decoded through the per-frame host exchange
            import host_lms
            key = ("lastword", lastword_size(c, inp), c["lm"][1])
            if key not in self._lms:
                self._lms[key] = _capi.HostLM(host_lms.LastWordLM(key[1], key[2]), lib=self.lib)
            return self._lms[key]
        path, vocab = arpa_path(c, inp)
        key = (path,)
        if key not in self._lms:
            self._lms[key] = _capi.ArpaLM(path, vocab, lib=self.lib)
        return self._lms[key]

    def trie_for(self, c, inp):
        word_lm = c["lm"] != "zero" and not c["is_lm_token"]
        key = (c["lexicon"], c["N"], c["label_scores"], c["lm"] if word_lm else None)
        if key not in self._tries:
            ht = _capi.HostTrie(c["N"], 0, lib=self.lib)
            sf, so = inp["lex"]
            scores = inp["scores"]
            if word_lm:  # lm.score(start, word) through the product's own tables
                lm = self.lm_for(c, inp)
                scores = np.array([lm.score_sequence([w], False)[0][0] for w in range(inp["W"])],
                                  dtype=np.float32)
            ht.insert_many(sf, so, inp["labels"], scores)
            ht.smear(1)
            self._tries[key] = (ht, ht.upload(self.ctx))
        return self._tries[key][1]

    def decoder(self, c, inp, threads=None, lm=None):
        opt = _capi.make_options(c["K"], c["Kt"], c["thr"], c["lm_weight"], c["word_score"], c["unk_score"],
                                 c["sil_score"], c["log_add"], c["crit"])
        N = c["N"]
        blank = N - 1 if c["crit"] == "ctc" else -1
        lm = lm or self.lm_for(c, inp)
        if c["kind"] == "lexfree":
            d = _capi.BatchDecoder(self.ctx, _capi.LEXFREE, opt, lm, 0, blank, transitions=inp["tr"])
        else:
            d = _capi.BatchDecoder(self.ctx, _capi.LEXICON, opt, lm, 0, blank, unk=inp["W"],
                                   trie=self.trie_for(c, inp), transitions=inp["tr"],
                                   is_lm_token=c["is_lm_token"])
        if threads:
            d.set("threads", threads)
        return d

    def run(self, c, inp=None, threads=None, sets=None):
        inp = inp or case_inputs(c)
        d = self.decoder(c, inp, threads)
        for k, v in (sets or {}).items():
            d.set(k, v)
        d.decode_batch(inp["e"], [c["T"]], c["N"])
        out = d.results(0)
        self.last_engine = d.get("engine")
        d.close()
        return out


def hyps_equal(a, b, score_tol=0.0):
    """Exact comparison (score bit patterns, tokens, words) unless score_tol."""
    if len(a) != len(b):
        return False, "n-best size %d != %d" % (len(a), len(b))
    for i, (x, y) in enumerate(zip(a, b)):
        for f in ("score", "am", "lm"):
            xv, yv = getattr(x, f), getattr(y, f)
            if score_tol == 0.0:
                if not (xv == yv or (np.isnan(xv) and np.isnan(yv))):
                    return False, "hyp %d %s %r != %r" % (i, f, float(xv).hex(), float(yv).hex())
            elif abs(xv - yv) > score_tol:
                return False, "hyp %d %s %.9f vs %.9f" % (i, f, xv, yv)
        if len(x.tokens) != len(y.tokens) or not np.array_equal(x.tokens, y.tokens):
            return False, "hyp %d tokens differ" % i
        if not np.array_equal(x.words, y.words):
            return False, "hyp %d words differ" % i
    return True, ""


# ---- golden expectations ---------------------------------------------------
def encode_hyps(hyps, full):
    out = {"n": len(hyps), "hash": "%016x" % orclib.nbest_hash(hyps),
           "scores": [[float(h.score).hex(), float(h.am).hex(), float(h.lm).hex()] for h in hyps]}
    keep = hyps if full else hyps[:2]
    out["tokens"] = [[int(t) for t in h.tokens] for h in keep]
    out["words"] = [[int(t) for t in h.words] for h in keep]
    return out


def load_golden():
    with gzip.open(os.path.join(GOLDEN_DIR, "synthetic_expected.json.gz"), "rt") as f:
        return json.load(f)


def check_against_golden(hyps, exp, score_tol=0.0):
    if len(hyps) != exp["n"]:
        return False, "n-best size %d != golden %d" % (len(hyps), exp["n"])
    for i, h in enumerate(hyps):
        for j, f in enumerate(("score", "am", "lm")):
            want = float.fromhex(exp["scores"][i][j])
            got = getattr(h, f)
            if score_tol == 0.0:
                if got != want:
                    return False, "hyp %d %s %s != golden %s" % (i, f, float(got).hex(), exp["scores"][i][j])
            elif abs(got - want) > score_tol:
                return False, "hyp %d %s %.9f vs golden %.9f" % (i, f, got, want)
    for i, tk in enumerate(exp["tokens"]):
        if not np.array_equal(hyps[i].tokens, np.array(tk, dtype=np.int32)):
            return False, "hyp %d tokens differ from golden" % i
        if not np.array_equal(hyps[i].words, np.array(exp["words"][i], dtype=np.int32)):
            return False, "hyp %d words differ from golden" % i
    if score_tol == 0.0 and "%016x" % orclib.nbest_hash(hyps) != exp["hash"]:
        return False, "n-best hash differs from golden"
    return True, ""
