import sys, itertools
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import cases, helpers
from oracle import orclib
orc = orclib.load("oracle")
sess = helpers.FltxSession(sys.argv[1] if len(sys.argv) > 1 else None)
grid = list(itertools.product([65, 100, 257, 1024, 1025, 3000, 8192], [1, 7, 50, 64], [1, 5, 30, 50, 64],
                         [0.0, 3.0, 25.0, float("inf")], [0.0, -0.7, 0.4], ["ctc", "asg"], ["ctc", "uniform"]))
T_of = lambda i: [1, 23, 60, 9][i % 4]
for i in (198, 473):
    N, K, Kt, thr, sil, crit, dist = grid[i]
    T = T_of(i)
    c = cases.case("wp%d" % i, dist=dist, T=T, N=N, K=K, Kt=Kt, u=7000 + i, crit=crit, sil_score=sil, thr=thr,
                   trans_seed=(90 + i % 5) if (crit == "asg" and N <= 1100) else None)
    inp = helpers.case_inputs(c)
    want = helpers.run_checker(orc, c, inp)
    d = sess.decoder(c, inp)
    d.decode_batch(inp["e"], [T], N)
    got = d.results(0)
    print(i, grid[i], T, "engine", d.get("engine"), d.get("wlane"), d.get("redone"), helpers.hyps_equal(want, got), orc.last_ties)
    d.close()
