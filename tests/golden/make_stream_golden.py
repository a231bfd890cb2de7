#!/usr/bin/env python3
"""Streaming golden vectors from the UNMODIFIED reference (oracle/_ref/libfltref.so): every scenario of
tests/stream_scenarios.py -- decodeStep chunks, getBestHypothesis(lookBack) after each, prune(lookBack)
after every second one, decodeEnd -- run twice under different heap layouts (must agree).  Dev container
only; writes tests/golden/streaming_expected.json.gz (data only)."""
import gzip
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import cases  # noqa: E402
import helpers  # noqa: E402
import stream_scenarios as ss  # noqa: E402
from make_golden import perturb_heap  # noqa: E402
from oracle import orclib  # noqa: E402


def main():
    if not orclib.have_ref():
        raise SystemExit("oracle/_ref/libfltref.so is missing: make -C oracle ref (needs /root/reference)")
    ref = orclib.load("ref")
    out = {}
    for name, (chunks, lbs) in ss.SCENARIOS.items():
        c = cases.BY_NAME[name]
        inp = helpers.case_inputs(c)
        t1 = ss.trace_checker(ref, c, inp, chunks, lbs)
        keep = perturb_heap(33333)
        t2 = ss.trace_checker(ref, c, inp, chunks, lbs)
        del keep
        d = ss.first_difference(t1, t2)
        if d:
            raise SystemExit("reference is heap-layout dependent on %s: %s" % (name, d))
        out[name] = t1
        print("%-26s %d events, final n = %d" % (name, len(t1), t1[-1]["final"]["n"]))
    with gzip.open(os.path.join(HERE, "streaming_expected.json.gz"), "wt") as f:
        json.dump(out, f, separators=(",", ":"))


if __name__ == "__main__":
    main()
