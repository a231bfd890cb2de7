"""Streaming scenarios shared by tests/golden/make_stream_golden.py (which drives the compiled
REFERENCE through them in the dev container) and the tests (oracle and HIP path against the committed
trace).  A scenario = a case of tests/cases.py fed in chunks through decodeStep with
getBestHypothesis(lookBack) after every chunk and prune(lookBack) after every second one
(decoder/Decoder.h:20-34, LexiconFreeDecoder.cpp:188-227, LexiconDecoder.cpp:285-325, Utils.h:268-342)."""
import numpy as np

import helpers
from oracle import orclib

SCENARIOS = {
    "lf_ctc_t60_k10": ([7, 9, 1, 12, 20, 30], [0, 2, 0, 5]),
    "lx_spell_t60_k12_full": ([10, 10, 10, 10, 10, 10, 10], [0, 3, 1]),
    "lx_scores_t50": ([5, 15, 10, 10, 10, 10], [2, 0]),
    "C1_ctc_u0": ([50, 50, 30, 20, 50], [0, 10, 3]),
    "ng_word_t60_k16_4g": ([10, 10, 10, 10, 10, 10], [0, 2]),
    "C3_spell_u0": ([200, 100, 300, 150, 250], [0, 20, 5, 50]),
    "C2_ctc_u0": ([50] * 20, [0]),  # the shape bench.py streams: 50-frame chunks, prune(0)
}


def checker_decoder(lib, c, inp):
    opt = orclib.make_options(c["K"], c["Kt"], c["thr"], c["lm_weight"], c["word_score"], c["unk_score"],
                              c["sil_score"], c["log_add"], c["crit"])
    N = c["N"]
    blank = N - 1 if c["crit"] == "ctc" else -1
    lm = helpers.checker_lm(lib, c, inp)
    if c["kind"] == "lexfree":
        return lib.lexfree(opt, lm, 0, blank, inp["tr"]), lm, None
    sf, so = inp["lex"]
    scores = inp["scores"]
    if c["lm"] != "zero" and not c["is_lm_token"]:
        scores = helpers.checker_word_scores(lib, lm, inp["W"])
    trie = lib.build_trie(N, 0, sf, so, inp["labels"], scores, smear=1)
    return lib.lexicon(opt, trie, lm, 0, blank, inp["W"], inp["tr"], c["is_lm_token"]), lm, trie


def _enc_one(h):
    if h is None or len(h.tokens) == 0:
        return None
    return {"scores": [float(h.score).hex(), float(h.am).hex(), float(h.lm).hex()],
            "tokens": [int(t) for t in h.tokens], "words": [int(t) for t in h.words]}


def trace_checker(lib, c, inp, chunks, look_backs):
    """The scenario on the CPU checker (oracle or compiled reference) -> list of events."""
    N, T = c["N"], c["T"]
    od, olm, otrie = checker_decoder(lib, c, inp)
    lib.decoder_begin(od)
    ev, t = [], 0
    for i, ch in enumerate(chunks):
        ch = min(ch, T - t)
        row = np.ascontiguousarray(inp["e"][t:t + ch])
        lib.decoder_step(od, orclib._fp(row), ch, N)
        t += ch
        lb = look_backs[i % len(look_backs)]
        e = {"frames_done": t, "look_back": lb, "best": _enc_one(lib.best(od, lb, T + 8)),
             "in_buffer": lib.decoder_n_frames_in_buffer(od)}
        if i % 2 == 1:
            lib.decoder_prune(od, lb)
            e["after_prune"] = {"in_buffer": lib.decoder_n_frames_in_buffer(od),
                                "buffer": helpers.encode_hyps(sorted(lib.collect(od), key=lambda h: -h.score), True)}
        ev.append(e)
        if t >= T:
            break
    lib.decoder_end(od)
    ev.append({"final": helpers.encode_hyps(lib.collect(od), True)})
    lib.decoder_destroy(od)
    return ev


def trace_device(session, c, inp, chunks, look_backs, threads=None, tunables=()):
    """The same on a libfltx session (HIP path or the emulator)."""
    N, T = c["N"], c["T"]
    d = session.decoder(c, inp, threads)
    for k, v in tunables:
        d.set(k, v)
    d.stream_begin(1, N, T + 4)
    ev, t = [], 0
    for i, ch in enumerate(chunks):
        ch = min(ch, T - t)
        d.stream_step(np.ascontiguousarray(inp["e"][t:t + ch]), [ch])
        t += ch
        lb = look_backs[i % len(look_backs)]
        e = {"frames_done": t, "look_back": lb, "best": _enc_one(d.best(0, lb, T + 8)),
             "in_buffer": d.frames_in_buffer(0)}
        if i % 2 == 1:
            d.stream_prune(lb)
            e["after_prune"] = {"in_buffer": d.frames_in_buffer(0), "buffer": helpers.encode_hyps(d.results(0), True)}
        ev.append(e)
        if t >= T:
            break
    d.stream_end()
    ev.append({"final": helpers.encode_hyps(d.results(0), True)})
    engine = d.get("engine")
    session.last_stream_redone = d.get("stream_redone")
    d.close()
    return ev, engine


def first_difference(want, got):
    """None if the traces agree bit for bit, else a description of the first event that differs."""
    if len(want) != len(got):
        return "%d events, expected %d" % (len(got), len(want))
    for i, (a, b) in enumerate(zip(want, got)):
        if a != b:
            keys = [k for k in a if a.get(k) != b.get(k)]
            return "event %d (%s) differs in %s" % (i, a.get("frames_done", "final"), keys)
    return None


def has_ties(trace):
    """True if some buffer of the trace holds two hypotheses of equal score (their order is then not defined)."""
    for e in trace:
        for buf in (e.get("final"), (e.get("after_prune") or {}).get("buffer")):
            if buf:
                sc = [x[0] for x in buf["scores"]]
                if len(set(sc)) != len(sc):
                    return True
    return False
