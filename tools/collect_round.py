#!/usr/bin/env python3
"""Copies the summaries tools/profile_round.sh left under gpurun_out/<round>/ into profiles/<round>/ and
rebuilds hbm_traffic_*.json with the workload keys bench.py looks up (geometry and engine are
taken from the bench lines of the same pass).  Run from the repo root after the gpurun call;
start from an empty gpurun_out/<round>/ (older passes would be mixed in)."""
import glob, json, os, shutil, subprocess, sys

R = sys.argv[1] if len(sys.argv) > 1 else "r03"
O, D = "gpurun_out/" + R, "profiles/" + R
os.makedirs(D, exist_ok=True)
SHAPE = {"C2": (256, 1000, 50), "C3": (256, 1000, 50), "C4": (256, 1500, 100), "C5": (1024, 1500, 100)}


def one(pattern):
    hits = glob.glob(pattern)
    assert len(hits) == 1, (pattern, hits)
    return hits[0]


for w in ("C2", "C3", "C4"):
    shutil.copy("%s/bench_%s.json" % (O, w), "%s/bench_%s.json" % (D, w))
    shutil.copy(one("%s/prof_%s/*/*_kernel_stats.csv" % (O, w)), "%s/rocprofv3_kernel_stats_%s.csv" % (D, w))
    sq = glob.glob("%s/sq1_%s/*/*_counter_collection.csv" % (O, w)) + glob.glob("%s/sq2_%s/*/*_counter_collection.csv" % (O, w))
    open("%s/pmc_SQ_%s.txt" % (D, w), "w").write(
        subprocess.run([sys.executable, "tools/pmc_summary.py"] + sq, capture_output=True, text=True).stdout)
shutil.copy("%s/bench_C5_1gpu.json" % O, "%s/bench_C5_one_gpu.json" % D)
for w in ("C2", "C3", "C4", "C5"):
    f = one("%s/pmc_fetch_%s/*/*_counter_collection.csv" % (O, w))
    wr = one("%s/pmc_write_%s/*/*_counter_collection.csv" % (O, w))
    shutil.copy(f, "%s/pmc_FETCH_SIZE_%s.csv" % (D, w))
    shutil.copy(wr, "%s/pmc_WRITE_SIZE_%s.csv" % (D, w))
    b = json.load(open("%s/bench_%s.json" % (D, w if w != "C5" else "C5_one_gpu")))
    B, T, K = SHAPE[w]
    key = "%s threads=%d batch=%d T=%d N=29 beam=%d engine=%d" % (
        w, b["config"]["threads_per_utterance"], B, T, K, b["config"]["engine"])
    subprocess.run([sys.executable, "tools/pmc_traffic.py", f, wr, key, "%s/hbm_traffic_%s.json" % (D, w)],
                   stdout=subprocess.DEVNULL, check=True)
    print(key)
