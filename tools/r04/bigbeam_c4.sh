#!/bin/bash
# Lexicon + 4-gram beam sweep on the C4 shape (256 x 1500 frames): one bench.py line per beam
R="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$1"; shift
BEAMS="${@:-100 128 129 160 200 256 300}"
: > "$OUT"
for K in $BEAMS; do
  python "$R/bench.py" --workload C4 --beam $K --steps 3 --warmup 2 --no-extras --cpu-sample 2 >> "$OUT" 2>> "$OUT.err" || echo "{\"beam\": $K, \"failed\": true}" >> "$OUT"
done
python - "$OUT" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    try:
        j = json.loads(line)
    except Exception:
        continue
    if "config" not in j:
        print(line.strip()); continue
    c = j["config"]
    print("beam", c["workload"].split("beam=")[1].split(",")[0], "ms/step %.2f" % j["ms_per_step"], "kernel %.2f" % j["roofline"]["kernel_ms"], "engine", c["engine"], "groups", c.get("lane_groups"), "threads", c["threads_per_utterance"], "redone", c["redone"], "mismatch", j.get("cpu_baseline", {}).get("gpu_nbest_mismatches_on_sample"))
PY
