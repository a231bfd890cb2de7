#!/bin/bash
# Round 5, one gpurun call: (a) multi-GPU plumbing on the one GPU of the box -- eight ranks through torch.distributed
# (gloo: ranks sharing a device measure nothing, the line shows the launcher path end to end) and eight contexts through
# fltx_group_*; (b) C2 on every compiled geometry of fltx_slane.h (threads x list positions per token wave): what a
# barrier costs against what a list position costs.
R="${GRAFT_REPO_ROOT:-$(pwd)}"
O="$R/gpurun_out/r05"
mkdir -p "$O"
cd "$R"
timeout 900 python bench.py --gpus 8 --workload C5 --device 0 --backend gloo --steps 2 --warmup 1 --no-cpu > "$O/bench_process_8x_same_gpu.json" 2> "$O/bench_process_8x_same_gpu.err"
timeout 600 python bench.py --mode group --gpus 8 --device 0 --workload C5 --batch 128 --steps 2 --warmup 1 > "$O/bench_group_8x_same_gpu.json" 2> "$O/bench_group_8x_same_gpu.err"
: > "$O/c2_geometries.jsonl"
for th in 320 384 448 512 576 640; do
  timeout 300 python bench.py --no-extras --no-cpu --steps 20 --warmup 5 --set slane_threads=$th 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(json.dumps({'slane_threads': $th, 'threads': d['config']['threads_per_utterance'], 'kernel_ms': d['roofline']['kernel_ms'], 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'us_per_frame': d['roofline']['us_per_frame_step']}))" >> "$O/c2_geometries.jsonl"
done
cat "$O/c2_geometries.jsonl"
tail -c 400 "$O/bench_process_8x_same_gpu.json"; echo; tail -c 300 "$O/bench_group_8x_same_gpu.json"
