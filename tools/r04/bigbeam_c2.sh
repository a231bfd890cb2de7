#!/bin/bash
# Lexicon-free beam sweep on the C2 shape (256 x 1000 frames, N = 29): one bench.py line per beam (0 n-best mismatches vs the
# compiled reference on the sampled utterances is part of bench.py's exit code).  Usage: bigbeam_c2.sh <out.jsonl> [beams...]
R="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$1"; shift
BEAMS="${@:-64 65 100 128 160 200 256 300 400 500}"
: > "$OUT"
for K in $BEAMS; do
  python "$R/bench.py" --workload C2 --beam $K --steps 3 --warmup 2 --no-extras --cpu-sample 4 >> "$OUT" 2>> "$OUT.err" || echo "{\"beam\": $K, \"failed\": true}" >> "$OUT"
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
    print("beam", c["workload"].split("beam=")[1].split(",")[0], "ms/step %.2f" % j["ms_per_step"], "kernel %.2f" % j["roofline"]["kernel_ms"], "engine", c["engine"], "threads", c["threads_per_utterance"], "redone", c["redone"], "mismatch", j.get("cpu_baseline", {}).get("gpu_nbest_mismatches_on_sample"))
PY
