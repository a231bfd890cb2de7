#!/usr/bin/env python3
"""Per-phase shader clocks of the generic engine on the reference's own DecoderTest configuration (see
tools/r04/decodertest_replay_time.py): where a frame's time goes at beam 2 500 / 500 / 256 / 50."""
import gzip, json, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import helpers
from golden.make_golden import parse_lexicon_dump
from text_amd import _capi

d = os.path.join(helpers.GOLDEN_DIR, "decodertest")
rd = lambda n: gzip.open(os.path.join(d, n + ".gz"), "rb").read()
lex = parse_lexicon_dump(rd("lexicon_dump.txt").decode())
TN = np.frombuffer(rd("TN.bin"), dtype=np.int32)
T, N = int(TN[0]), int(TN[1])
em = np.frombuffer(rd("emission.bin"), dtype=np.float32).copy()
tr = np.frombuffer(rd("transition.bin"), dtype=np.float32).copy()
tmp = tempfile.NamedTemporaryFile(suffix=".arpa", delete=False)
tmp.write(rd("lm.arpa")); tmp.close()
sess = helpers.FltxSession(None)
lm = _capi.ArpaLM(tmp.name, lex["words"])
ht = _capi.HostTrie(lex["ntok"], lex["sil"])
cache = {}
for wi, w, sp in lex["entries"]:
    if wi not in cache:
        cache[wi] = lm.score_sequence([wi], False)[0][0]
    ht.insert(sp, wi, cache[wi])
ht.smear(1)
trie = ht.upload(sess.ctx)
names = ["prep", "rebuild+merge-insert", "fold", "select", "build", "row+barrier", "list+score-pass", "cut-histogram"]
for K, B in ((2500, 1), (1000, 1), (500, 1), (256, 1), (50, 1), (500, 64), (50, 256)):
    opt = _capi.make_options(K, 25000, 100.0, 2.0, 2.0, -float("inf"), -1.0, False, "asg")
    dec = _capi.BatchDecoder(sess.ctx, _capi.LEXICON, opt, lm, lex["sil"], -1, unk=lex["unk"], trie=trie,
                             transitions=tr, is_lm_token=False)
    e = np.tile(em, B)
    Ts = np.full(B, T, dtype=np.int32)
    dec.decode_batch(e, Ts, N); sess.ctx.synchronize()
    dec.decode_batch(e, Ts, N); sess.ctx.synchronize()
    k_ms, _ = dec.timing()
    dec.set("profile", 1)
    dec.decode_batch(e, Ts, N); sess.ctx.synchronize()
    pr = dec.profile().astype(np.float64) / (B * T)
    info = {k: dec.get(k) for k in ("engine", "threads", "lds", "hot_level", "cut", "cap2", "items")}
    print("beam %d batch %d: kernel %.2f ms = %.1f us/frame; clocks/frame: %s | %s" % (
        K, B, k_ms, k_ms * 1e3 / T, ", ".join("%s %.0f" % (n, v) for n, v in zip(names, pr)), info), flush=True)
    dec.close()
os.unlink(tmp.name)
