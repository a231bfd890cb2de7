import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import helpers, cases
from oracle import orclib
lib = os.environ.get("EMU_LIB")
sess = helpers.FltxSession(lib) if lib else helpers.FltxSession(None)
gold = helpers.load_golden()
orc = orclib.load("oracle")
for name in os.environ.get("NAMES", "ml_word_t60_k16,ml_word_uni_t50_k48,ml_word_asg_t40_k24").split(","):
    c = cases.BY_NAME[name]
    inp = helpers.case_inputs(c)
    t0 = time.time()
    d = sess.decoder(c, inp)
    d.decode_batch(inp["e"], [c["T"]], c["N"])
    got = d.results(0)
    info = (d.get("engine"), d.get("lane_groups"), d.get("redone"), d.get("why_not_lane"), d.get("fallback_reasons"))
    d.close()
    want = helpers.run_checker(orc, c, inp)
    ok, why = helpers.hyps_equal(want, got)
    okg = helpers.check_against_golden(got, gold[name]) if name in gold else None
    print(name, "engine/groups/redone/why/fb", info, "n", len(got), len(want), "oracle:", ok, why, "golden:", okg, "%.1fs" % (time.time() - t0), flush=True)
