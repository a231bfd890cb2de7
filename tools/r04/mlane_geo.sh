#!/bin/bash
# which fltx_mlane.h geometry is fastest per lane-group count (C2 shape): beam x geometry grid
R="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$1"
: > "$OUT"
for spec in "100 0" "100 1" "100 2" "128 0" "128 1" "200 3" "200 4" "200 5" "256 3" "256 4" "100 3" "100 4" "500 6"; do
  set -- $spec
  python "$R/bench.py" --workload C2 --beam $1 --set mlane_geo=$2 --steps 3 --warmup 2 --no-extras --no-cpu 2>> "$OUT.err" | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); c=j['config']
print('beam $1 geo $2 ms/step %.2f kernel %.2f threads %d engine %d' % (j['ms_per_step'], j['roofline']['kernel_ms'], c['threads_per_utterance'], c['engine']))" >> "$OUT"
done
cat "$OUT"
