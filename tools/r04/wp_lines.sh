#!/bin/bash
# The word-piece lines of profiles/r04 on the current code: bench lines (N = 1 024 x 256 utterances, N = 8 192 x 64),
# rocprofv3 stats of the three kernels of a batch (one stream), per-phase clocks.  Outputs under gpurun_out/r04w/.
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$(pwd)}"; O="$R/gpurun_out/r04w"; rm -rf "$O"; mkdir -p "$O"; cd "$R"
timeout 600 python bench.py --workload WP --steps 4 --warmup 2 --no-extras --cpu-sample 16 > "$O/bench_WP_n1024.json" 2> "$O/bench_WP.err"
timeout 900 python bench.py --workload WP --tokens 8192 --batch 64 --steps 2 --warmup 1 --no-extras --cpu-sample 2 > "$O/bench_WP_n8192_b64.json" 2>> "$O/bench_WP.err"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_WP" -- python "$R/bench.py" --workload WP --steps 4 --warmup 2 --no-cpu --no-extras --pipeline 1 > "$O/prof_WP.log" 2>&1 )
timeout 300 python bench.py --workload WP --steps 2 --warmup 1 --no-cpu --no-extras --profile --profile-waves 0,7,8 --profile-out "$O/phase_split_WP.txt" > /dev/null 2>> "$O/bench_WP.err"
find "$O" -name "*.db" -delete 2>/dev/null
