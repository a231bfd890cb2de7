import gzip, json, os, sys, tempfile, time
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import helpers
from golden.make_golden import parse_lexicon_dump
from oracle import orclib
from text_amd import _capi
EMU = int(os.environ.get("EMU", "1"))
d = os.path.join(helpers.GOLDEN_DIR, "decodertest")
rd = lambda n: gzip.open(os.path.join(d, n + ".gz"), "rb").read()
lex = parse_lexicon_dump(rd("lexicon_dump.txt").decode())
TN = np.frombuffer(rd("TN.bin"), dtype=np.int32)
T, N = int(TN[0]), int(TN[1])
em = np.frombuffer(rd("emission.bin"), dtype=np.float32).copy()
tr = np.frombuffer(rd("transition.bin"), dtype=np.float32).copy()
tmp = tempfile.NamedTemporaryFile(suffix=".arpa", delete=False)
tmp.write(rd("lm.arpa")); tmp.close()
sess = helpers.FltxSession(os.environ.get("EMU_LIB", helpers.EMU_LIB) if EMU else None)
lm = _capi.ArpaLM(tmp.name, lex["words"], lib=sess.lib)
ht = _capi.HostTrie(lex["ntok"], lex["sil"], lib=sess.lib)
cache = {}
for wi, w, sp in lex["entries"]:
    if wi not in cache:
        cache[wi] = lm.score_sequence([wi], False)[0][0]
    ht.insert(sp, wi, cache[wi])
ht.smear(1)
trie = ht.upload(sess.ctx)
cpu = orclib.load("ref" if orclib.have_ref() else "oracle")
clm = cpu.lm_arpa_create(tmp.name.encode(), "\n".join(lex["words"]).encode())
ctrie = cpu.trie_create(lex["ntok"], lex["sil"])
for wi, w, sp in lex["entries"]:
    a = np.array(sp, dtype=np.int32)
    cpu.trie_insert(ctrie, orclib._ip(a), len(sp), wi, cache[wi])
cpu.trie_smear(ctrie, 1)
Ks = [int(x) for x in os.environ.get("KS", "50,20,100,128,200,256").split(",")]
rng = np.random.default_rng(5)
for K in Ks:
  for variant in range(int(os.environ.get("NV", "2"))):
    e1 = em if variant == 0 else (em + rng.normal(0, 0.8, em.shape).astype(np.float32))
    Tv = T if variant == 0 else int(rng.integers(20, T))
    for crit, trv in (("asg", tr), ("ctc", None)):
        blank = -1 if crit == "asg" else N - 1
        opt = _capi.make_options(K, 25000, 100.0, 2.0, 2.0, -float("inf"), -1.0, False, crit)
        dec = _capi.BatchDecoder(sess.ctx, _capi.LEXICON, opt, lm, lex["sil"], blank, unk=lex["unk"], trie=trie,
                                 transitions=trv, is_lm_token=False)
        t0 = time.time()
        dec.decode_batch(e1[:Tv * N], np.array([Tv], dtype=np.int32), N); sess.ctx.synchronize()
        dt = time.time() - t0
        got = dec.results(0)
        copt = orclib.make_options(K, 25000, 100.0, 2.0, 2.0, -float("inf"), -1.0, False, crit)
        cdec = cpu.lexicon(copt, ctrie, clm, lex["sil"], blank, lex["unk"], trv, False)
        want = cpu.decode(cdec, e1[:Tv * N], Tv, N)
        cpu.decoder_destroy(cdec)
        ok, why = helpers.hyps_equal(want, got)
        ties = len({h.score for h in want}) != len(want)
        if ties and not ok:
            ok2 = [h.score for h in want] == [h.score for h in got]
            if ok2:
                from collections import defaultdict
                gw, gg = defaultdict(set), defaultdict(set)
                for h in want: gw[h.score].add((tuple(h.words), tuple(h.tokens)))
                for h in got: gg[h.score].add((tuple(h.words), tuple(h.tokens)))
                cut = want[-1].score
                bad = [s for s in gw if gw[s] != gg[s] and s != cut]
                ok2 = not bad
                why = "tie-groups ok" if ok2 else "tie-group differs at %r" % bad[:3]
        if not ok and os.environ.get("GEN", "1") == "1":
            dec.set("ylane", 0)
            dec.decode_batch(e1[:Tv * N], np.array([Tv], dtype=np.int32), N); sess.ctx.synchronize()
            gen = dec.results(0)
            okg, whyg = helpers.hyps_equal(want, gen)
            oky, whyy = helpers.hyps_equal(gen, got)
            why += " | generic(engine %d) vs ref: %s %s | ylane vs generic: %s %s" % (dec.get("engine"), okg, whyg, oky, whyy)
        print(K, variant, crit, Tv, "engine", dec.get("engine"), "groups", dec.get("lane_groups"), "redone", dec.get("redone"),
              "why", dec.get("why_not_lane"), "fb", dec.get("fallback_reasons"), "nhyp", len(got), len(want), "ok", ok, why, "%.1fs" % dt, flush=True)
        dec.close()
os.unlink(tmp.name)
