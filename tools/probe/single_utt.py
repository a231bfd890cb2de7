#!/usr/bin/env python3
"""Where the time of ONE utterance per call goes (the reference's API shape): C-ABI decode_batch with B = 1 + result
fetches, then the pybind facade's decode().  C2 shape (T = 1000, N = 29, beam 50)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "text_amd", "compat"))
import numpy as np
from text_amd import _capi, synth
T, N, K = 1000, 29, 50
e = synth.batch("ctc", 8, T, N).reshape(8, T * N)
ctx = _capi.Context()
opt = _capi.make_options(K, N, 25.0, 0.0, 0.0, float("-inf"), 0.0, False, "ctc")
dec = _capi.BatchDecoder(ctx, _capi.LEXFREE, opt, _capi.ZeroLM(ctx), 0, N - 1)
def t(f, n=20):
    f(); best = 1e9
    for _ in range(n):
        t0 = time.perf_counter(); f(); best = min(best, time.perf_counter() - t0)
    return best * 1e3
def only_decode():
    dec.decode_batch(e[1], [T], N); ctx.synchronize()
print("C ABI decode_batch(B=1) + sync: %.3f ms; kernel %.3f ms backtrace %.3f ms" % ((t(only_decode),) + tuple(dec.timing())))
print("  + results(0) (50 Hyp objects):  %.3f ms" % t(lambda: (dec.decode_batch(e[1], [T], N), dec.results(0))))
print("  + results_arrays_compact:       %.3f ms" % t(lambda: (dec.decode_batch(e[1], [T], N), dec.results_arrays_compact())))
for k in ("engine", "threads"):
    try: print(" ", k, dec.get(k))
    except Exception as ex: print(" ", k, ex)
from flashlight.lib.text.decoder import CriterionType, LexiconFreeDecoder, LexiconFreeDecoderOptions, ZeroLM
opts = LexiconFreeDecoderOptions(beam_size=K, beam_size_token=N, beam_threshold=25.0, lm_weight=0.0, sil_score=0.0, log_add=False,
                                 criterion_type=CriterionType.CTC)
fd = LexiconFreeDecoder(opts, ZeroLM(), 0, N - 1, [])
print("facade decode(): %.3f ms" % t(lambda: fd.decode(e[1].ctypes.data, T, N)))
def steps():
    fd.decode_begin(); fd.decode_step(e[1].ctypes.data, T, N); fd.decode_end()
print("facade decode_begin/step/end: %.3f ms" % t(steps))
print("  + get_all_final_hypothesis: %.3f ms" % t(lambda: (steps(), fd.get_all_final_hypothesis())))
print("  + get_best_hypothesis:      %.3f ms" % t(lambda: (steps(), fd.get_best_hypothesis())))
