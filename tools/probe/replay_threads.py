#!/usr/bin/env python3
"""DecoderTest replay (beam 2 500, one utterance) at different workgroup sizes of the generic engine."""
import gzip, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import helpers
from golden.make_golden import parse_lexicon_dump
from text_amd import _capi
d = os.path.join(helpers.GOLDEN_DIR, "decodertest")
rd = lambda n: gzip.open(os.path.join(d, n + ".gz"), "rb").read()
lex = parse_lexicon_dump(rd("lexicon_dump.txt").decode())
TN = np.frombuffer(rd("TN.bin"), dtype=np.int32); T, N = int(TN[0]), int(TN[1])
em = np.frombuffer(rd("emission.bin"), dtype=np.float32).copy()
tr = np.frombuffer(rd("transition.bin"), dtype=np.float32).copy()
tmp = tempfile.NamedTemporaryFile(suffix=".arpa", delete=False); tmp.write(rd("lm.arpa")); tmp.close()
sess = helpers.FltxSession(None)
lm = _capi.ArpaLM(tmp.name, lex["words"])
ht = _capi.HostTrie(lex["ntok"], lex["sil"]); cache = {}
for wi, w, sp in lex["entries"]:
    if wi not in cache: cache[wi] = lm.score_sequence([wi], False)[0][0]
    ht.insert(sp, wi, cache[wi])
ht.smear(1); trie = ht.upload(sess.ctx)
ref = None
for K in (2500, 500):
    for thr in (0, 256, 512, 1024):
        opt = _capi.make_options(K, 25000, 100.0, 2.0, 2.0, -float("inf"), -1.0, False, "asg")
        dec = _capi.BatchDecoder(sess.ctx, _capi.LEXICON, opt, lm, lex["sil"], -1, unk=lex["unk"], trie=trie, transitions=tr)
        if thr: dec.set("threads", thr)
        try:
            dec.decode_batch(em, [T], N); sess.ctx.synchronize()
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter(); dec.decode_batch(em, [T], N); sess.ctx.synchronize(); best = min(best, time.perf_counter() - t0)
            got = [(h.score, tuple(h.tokens)) for h in dec.results(0)]
            if thr == 0: ref = got
            print("beam", K, "threads asked", thr, "used", dec.get("threads"), "engine", dec.get("engine"), "ms %.2f" % (best * 1e3), "same n-best", got == ref, flush=True)
        except Exception as ex:
            print("beam", K, "threads", thr, "error", str(ex)[:100])
        dec.close()
os.unlink(tmp.name)
