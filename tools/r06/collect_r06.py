#!/usr/bin/env python3
"""Copies the summaries tools/r06/profile_r06.sh left under gpurun_out/r06p/ into profiles/r06/ and rebuilds
hbm_traffic_*.json with the workload keys bench.py looks up (geometry and engine from the bench lines of the same pass)."""
import glob, json, os, shutil, subprocess, sys

O, D = "gpurun_out/r06p", "profiles/r06"
os.makedirs(D, exist_ok=True)
SHAPE = {"C2": (256, 1000, 50), "C2T": (256, 1000, 50), "C3": (256, 1000, 50), "C4": (256, 1500, 100)}


def one(pattern):
    hits = sorted(glob.glob(pattern), key=os.path.getmtime)  # (gpurun merges into gpurun_out/: an earlier pass may have left its own)
    assert hits, pattern
    return hits[-1]


for f in ("bench_default.json", "bench_C2T.json", "bench_C2T_one_batch_at_a_time.json", "bench_C3.json", "bench_C4.json",
          "toklm_probe.txt", "phase_split_C2T.txt", "bench_C2T_beam100.json", "bench_C2T_beam200.json", "bench_C2T_beam500.json"):
    if os.path.exists(os.path.join(O, f)):
        shutil.copy(os.path.join(O, f), os.path.join(D, f))
for w in ("C2", "C2T", "C3", "C4"):
    shutil.copy(one("%s/prof_%s/*/*_kernel_stats.csv" % (O, w)), "%s/rocprofv3_kernel_stats_%s.csv" % (D, w))
for w in ("C2", "C2T", "C3"):
    hits = sorted(glob.glob("%s/prof_stream_%s/*/*_kernel_stats.csv" % (O, w)), key=os.path.getmtime)
    if hits:
        shutil.copy(hits[-1], "%s/rocprofv3_kernel_stats_streams_%s.csv" % (D, w))
for w in ("C2", "C2T"):
    sq = [one("%s/sq1_%s/*/*_counter_collection.csv" % (O, w)), one("%s/sq2_%s/*/*_counter_collection.csv" % (O, w))]
    open("%s/pmc_SQ_%s.txt" % (D, w), "w").write(
        subprocess.run([sys.executable, "tools/pmc_summary.py"] + sq, capture_output=True, text=True).stdout)
bench = {"C2": json.load(open(D + "/bench_default.json")), "C2T": json.load(open(D + "/bench_C2T.json")),
         "C3": json.load(open(D + "/bench_C3.json")), "C4": json.load(open(D + "/bench_C4.json"))}
for w in ("C2", "C2T", "C3", "C4"):
    f = one("%s/pmc_fetch_%s/*/*_counter_collection.csv" % (O, w))
    wr = one("%s/pmc_write_%s/*/*_counter_collection.csv" % (O, w))
    shutil.copy(f, "%s/pmc_FETCH_SIZE_%s.csv" % (D, w))
    shutil.copy(wr, "%s/pmc_WRITE_SIZE_%s.csv" % (D, w))
    b = bench[w]
    B, T, K = SHAPE[w]
    key = "%s threads=%d batch=%d T=%d N=29 beam=%d engine=%d" % (
        w, b["config"]["threads_per_utterance"], B, T, K, b["config"]["engine"])
    subprocess.run([sys.executable, "tools/pmc_traffic.py", f, wr, key, "%s/hbm_traffic_%s.json" % (D, w)],
                   stdout=subprocess.DEVNULL, check=True)
    print(key)
