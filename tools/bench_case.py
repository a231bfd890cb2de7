#!/usr/bin/env python3
"""Time one parity case (tests/cases.py) as a batch of identical-shape utterances on the GPU.

Usage: tools/bench_case.py CASE [batch] [steps]   e.g. tools/bench_case.py C4_spell_u0 256 3
Used for the secondary workloads quoted in DESIGN.md (C3 / C4); bench.py stays the contract bench.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases  # noqa: E402
import helpers  # noqa: E402
from text_amd import synth  # noqa: E402


def main():
    name = sys.argv[1]
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    c = cases.BY_NAME[name]
    s = helpers.FltxSession(None)
    inp = helpers.case_inputs(c)
    d = s.decoder(c, inp, int(os.environ["FLTX_THREADS"]) if os.environ.get("FLTX_THREADS") else None)
    lex = inp.get("lex") if c["dist"] == "lexspell" else None
    e = synth.batch(c["dist"], B, c["T"], c["N"], lexicon=lex, u0=0)
    Ts = np.full(B, c["T"], dtype=np.int32)
    d.decode_batch(e, Ts, c["N"])
    ms = []
    for _ in range(steps):
        t0 = time.perf_counter()
        d.decode_batch(e, Ts, c["N"])
        s.ctx.synchronize()
        ms.append((time.perf_counter() - t0) * 1e3)
    k, b = d.timing()
    if os.environ.get("FLTX_PROFILE"):
        d.set("profile", 1)
        d.decode_batch(e, Ts, c["N"])
        s.ctx.synchronize()
        pr = d.profile().astype(np.float64) / (B * c["T"])
        print("  clocks/frame/utt by phase [prep, generate, fold, select, build, row, -, -]:",
              " ".join("%.0f" % v for v in pr), "total %.0f" % pr.sum())
        d.set("profile", 0)
    print("  geometry: lds=%d cap=%d cap2=%d cut=%d items=%d lds_bytes=%d" % (
        d.get("lds"), d.get("cap"), d.get("cap2"), d.get("cut"), d.get("items"), d.stats()["lds_bytes"]))
    print("%s batch=%d T=%d K=%d: engine %d threads %d, decode kernel %.2f ms, backtrace %.2f ms, "
          "wall/batch %.2f ms (incl. H2D), %.2f M frames/s (kernel)" %
          (name, B, c["T"], c["K"], d.get("engine"), d.get("threads"), k, b, min(ms), B * c["T"] / k / 1e3))
    for rep in range(2 if os.environ.get("FLTX_E2E") else 0):  # host emissions in, every hypothesis out on the host
        # (second repetition = steady state: the pinned staging buffers exist)
        t0 = time.perf_counter()
        d.decode_batch(e, Ts, c["N"])
        t1 = time.perf_counter()
        allh = d.results_batch()
        nh = sum(len(h) for h in allh)
        t2 = time.perf_counter()
        t3 = time.perf_counter()
        raw = d.fetch_batch_raw()
        t4 = time.perf_counter()
        print("  end to end #%d: H2D + kernels %.2f ms, + n-best of all utterances on the host (%d hypotheses as NumPy "
              "views) %.2f ms => %.2f M frames/s; C-ABI fetch alone (already staged) %.3f ms" %
              (rep, (t1 - t0) * 1e3, nh, (t2 - t0) * 1e3, B * c["T"] / (t2 - t0) / 1e6, (t4 - t3) * 1e3))
        del raw
    ncpu = int(os.environ.get("FLTX_CPU", "0"))
    if ncpu > 0:  # reference (oracle/_ref) on the host, one thread, same utterances; n-best compared
        from oracle import orclib
        lib = orclib.load("ref" if orclib.have_ref() else "oracle")
        opt = orclib.make_options(c["K"], c["Kt"], c["thr"], c["lm_weight"], c["word_score"], c["unk_score"],
                                  c["sil_score"], c["log_add"], c["crit"])
        N = c["N"]
        blank = N - 1 if c["crit"] == "ctc" else -1
        lm = helpers.checker_lm(lib, c, inp)
        trie = None
        if c["kind"] == "lexicon":
            sf, so = inp["lex"]
            scores = inp["scores"]
            if c["lm"] != "zero" and not c["is_lm_token"]:
                scores = helpers.checker_word_scores(lib, lm, inp["W"])
            trie = lib.build_trie(N, 0, sf, so, inp["labels"], scores, smear=1)
        tt, mism = 0.0, 0
        for b in range(min(ncpu, B)):
            dec = (lib.lexfree(opt, lm, 0, blank, inp["tr"]) if c["kind"] == "lexfree" else
                   lib.lexicon(opt, trie, lm, 0, blank, inp["W"], inp["tr"], c["is_lm_token"]))
            t0 = time.perf_counter()
            hyps = lib.decode(dec, e[b], c["T"], N)
            tt += time.perf_counter() - t0
            lib.decoder_destroy(dec)
            ok, _ = helpers.hyps_equal(d.results(b), hyps, 1e-5 if c["log_add"] else 0.0)
            mism += 0 if ok else 1
        n = min(ncpu, B)
        print("  reference CPU, 1 thread: %.1f k frames/s over %d utterances (%.1f s); GPU n-best mismatches: %d; "
              "GPU/CPU-thread = %.0fx" % (n * c["T"] / tt / 1e3, n, tt, mism, (B * c["T"] / k * 1e3) / (n * c["T"] / tt)))
    d.close()


if __name__ == "__main__":
    main()
