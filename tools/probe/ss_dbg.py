import sys, itertools
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import cases, helpers, stream_scenarios as ss
from oracle import orclib
orc = orclib.load("oracle")
s = helpers.FltxSession(None)
grid = list(itertools.product([1, 3, 10, 50, 64], [2.0, 25.0, float("inf")], [None, 5], [12, 29], [7, 60], ["ctc", "uniform"], [0.0, -0.6], ["ctc", "asg"]))
for i, (K, thr, Kt, N, T, dist, sil, crit) in enumerate(grid):
    c = cases.case("ss%d" % i, dist=dist, u=1700 + i, T=T, N=N, K=K, Kt=Kt, thr=thr, sil_score=sil, crit=crit, trans_seed=(30 + i) if crit == "asg" else None)
    if not (K >= 50 and T == 60): continue
    inp = helpers.case_inputs(c)
    chunks, lbs = [[3, 9, 1, 12, 20, 30], [1] * 60, [25, 25, 25]][i % 3], [[0, 2, 0, 5], [0], [3, 1]][i % 3]
    want = ss.trace_checker(orc, c, inp, chunks, lbs)
    got, _ = ss.trace_device(s, c, inp, chunks, lbs)
    d = ss.first_difference(want, got)
    if not d: continue
    got3, _ = ss.trace_device(s, c, inp, chunks, lbs, tunables=[("sstream", 0)])
    print(i, {k: c[k] for k in ("K","thr","Kt","N","dist","sil_score","crit")}, d, "| lane-per-slot:", ss.first_difference(want, got3))
    for a, b in zip(want, got):
        if a != b:
            A, B = a["after_prune"]["buffer"], b["after_prune"]["buffer"]
            print("   n", A["n"], B["n"], "in_buffer", a["after_prune"]["in_buffer"], b["after_prune"]["in_buffer"])
            for j,(x,y) in enumerate(zip(A["scores"], B["scores"])):
                if x != y or A["tokens"][j] != B["tokens"][j]:
                    print("   first diff at hyp", j, x, y, A["tokens"][j][-6:], B["tokens"][j][-6:], "prev", A["scores"][j-1] if j else None, "next", A["scores"][j+1] if j+1<len(A["scores"]) else None)
                    break
            break
