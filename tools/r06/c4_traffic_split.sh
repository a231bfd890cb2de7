#!/bin/bash
# Where C4's HBM traffic comes from: FETCH_SIZE / WRITE_SIZE of the decode kernel with the
# LM-state memo in HBM (the default when two batches are in flight: two workgroups per CU) and in LDS (--no-corun: one
# 768-thread workgroup per CU), and with ZeroLM in place of the 4-gram (no probe chain).  Separate --pmc passes.
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$(pwd)}"; O="$R/gpurun_out/r06t"; rm -rf "$O"; mkdir -p "$O"; cd /tmp
run() { # name, bench flags
  n=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d "$O/${n}_$c" -- python "$R/bench.py" "$@" --steps 3 --warmup 1 --no-cpu --no-secondary --sustained-seconds 0 > "$O/${n}_$c.log" 2>&1
  done
  f=$(ls -t $O/${n}_FETCH_SIZE/*/*_counter_collection.csv | head -1); w=$(ls -t $O/${n}_WRITE_SIZE/*/*_counter_collection.csv | head -1)
  python "$R/tools/pmc_traffic.py" "$f" "$w" "$n" "$O/traffic_$n.json" > /dev/null
  python - "$O/traffic_$n.json" "$n" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d["kernels"].items():
    if "backtrace" not in k:
        print("%-28s %-60s fetch %7.1f MB  write %7.1f MB  total %7.1f MB" % (sys.argv[2], k[:60], v["fetch_bytes_corrected"] / 1e6, v["write_bytes"] / 1e6, v["hbm_bytes_per_launch"] / 1e6))
PY
}
run C4_memo_in_HBM --workload C4
run C4_memo_in_LDS --workload C4 --no-corun
run C3_for_scale --workload C3
find "$O" -name "*.db" -delete 2>/dev/null
