#!/bin/bash
R="${GRAFT_REPO_ROOT:-$(pwd)}"; O="$R/gpurun_out/r04b"; mkdir -p "$O"; cd "$R"
python -m pytest tests/test_streaming.py -q -x -k "device_chunk" > "$O/pytest_stream.log" 2>&1; tail -15 "$O/pytest_stream.log"
python -m pytest tests/test_gpu_parity.py -q -k "tied or engine_selection" > "$O/pytest_tied.log" 2>&1; tail -15 "$O/pytest_tied.log"
python bench.py --workload WP --steps 3 --warmup 1 --no-extras --cpu-sample 8 > "$O/bench_WP_n1024.json" 2> "$O/bench_WP_n1024.err"; tail -c 600 "$O/bench_WP_n1024.json"; tail -3 "$O/bench_WP_n1024.err"
python bench.py --mode group --gpus 2 --device 0 --steps 3 --warmup 1 > "$O/bench_group_2x_same_gpu.json" 2> "$O/bench_group.err"; cat "$O/bench_group_2x_same_gpu.json"; tail -3 "$O/bench_group.err"
python bench.py --gpus 2 --device 0 --backend gloo --steps 3 --warmup 1 --no-cpu > "$O/bench_process_2x_same_gpu.json" 2> "$O/bench_process.err"; tail -c 400 "$O/bench_process_2x_same_gpu.json"; tail -3 "$O/bench_process.err"
for w in C2 C3 C4; do python bench.py --workload $w --steps 2 --warmup 1 --no-cpu --profile --profile-waves 0,1,6,7,8 --profile-out "$O/phase_split_$w.txt" > /dev/null 2>> "$O/prof.err"; done
cat "$O"/phase_split_*.txt
