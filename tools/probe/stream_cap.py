"""tools/probe/stream_cap.py -- a 1000-frame stream through a 208-frame buffer (bench.py's streaming leg): status and
result against the oracle, on both lexicon-free stream engines."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases, helpers, stream_scenarios as ss
from oracle import orclib
from text_amd import _capi
orc = orclib.load("oracle")
s = helpers.FltxSession(None)
c = cases.case("cap", T=1000, K=50, N=29, u=5)
inp = helpers.case_inputs(c)
want = ss.trace_checker(orc, c, inp, [50] * 20, [0])
for sstream in (1, 0):
    d = s.decoder(c, inp)
    d.set("sstream", sstream)
    d.set("stream_total_frames", int(os.environ.get("TOTAL", 0))); d.stream_begin(1, 29, 208)
    try:
        for k in range(20):
            d.stream_step(np.ascontiguousarray(inp["e"][k * 50:(k + 1) * 50]), [50])
            d.stream_prune(0)
        d.stream_end()
        got = helpers.encode_hyps(d.results(0), True)
        print("sstream", sstream, "final equals oracle:", got == want[-1]["final"])
    except Exception as e:
        print("sstream", sstream, "ERROR", e)
    d.close()
