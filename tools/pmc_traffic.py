#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into profiles/<round>/hbm_traffic.json.

Usage: tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <workload> <out.json>

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md "HBM [CDNA4]":
FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE tallies 128-byte
requests at 64 bytes, so it is doubled; WRITE_SIZE is used as reported (uncalibrated).
"""
import csv
import json
import sys


def mean_by_kernel(path):
    acc = {}
    for r in csv.DictReader(open(path)):
        acc.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    fetch, write, workload, out = sys.argv[1:5]
    f, w = mean_by_kernel(fetch), mean_by_kernel(write)
    res = {"workload": workload, "kernels": {}}
    for k in f:
        if "fltx" not in k:
            continue
        fb = 2.0 * f[k] * 1024.0
        wb = w.get(k, 0.0) * 1024.0
        res["kernels"][k] = {"FETCH_SIZE_KiB": f[k], "WRITE_SIZE_KiB": w.get(k), "fetch_bytes_corrected": fb,
                             "write_bytes": wb, "hbm_bytes_per_launch": fb + wb}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
