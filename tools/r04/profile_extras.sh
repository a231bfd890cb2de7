#!/bin/bash
# Round-4 extras beside tools/profile_round.sh r04: beam sweeps (the cliffs of round 3), word-piece lines, per-phase clock
# split, group / process plumbing on the one GPU ("same GPU, not a scaling number"), a clean build of every kernel.
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$(pwd)}"; O="$R/gpurun_out/r04x"; rm -rf "$O"; mkdir -p "$O"; cd "$R"
tools/r04/bigbeam_c2.sh "$O/bigbeam_C2.jsonl" 64 65 100 128 160 200 256 300 400 500 > "$O/bigbeam_C2.txt"
tools/r04/bigbeam_c4.sh "$O/bigbeam_C4.jsonl" 100 128 129 160 200 256 300 500 > "$O/bigbeam_C4.txt"
python bench.py --workload WP --steps 3 --warmup 1 --no-extras --cpu-sample 8 > "$O/bench_WP_n1024.json" 2> "$O/bench_WP.err"
python bench.py --workload WP --tokens 8192 --batch 64 --steps 2 --warmup 1 --no-extras --cpu-sample 2 > "$O/bench_WP_n8192_b64.json" 2>> "$O/bench_WP.err"
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_WP" -- python "$R/bench.py" --workload WP --steps 4 --warmup 2 --no-cpu --no-extras --pipeline 1 > "$O/prof_WP.log" 2>&1 )
python bench.py --workload WP --steps 2 --warmup 1 --no-cpu --no-extras --profile --profile-waves 0,7,8 --profile-out "$O/phase_split_WP.txt" > /dev/null 2>> "$O/prof.err"
for w in C2 C3 C4; do python bench.py --workload $w --steps 2 --warmup 1 --no-cpu --profile --profile-waves 0,1,6,7,8 --profile-out "$O/phase_split_$w.txt" > /dev/null 2>> "$O/prof.err"; done
python bench.py --mode group --gpus 2 --device 0 --steps 3 --warmup 1 > "$O/bench_group_2x_same_gpu.json" 2> "$O/bench_group.err"
python bench.py --gpus 2 --device 0 --backend gloo --steps 3 --warmup 1 --no-cpu > "$O/bench_process_2x_same_gpu.json" 2> "$O/bench_process.err"
( time FLTX_BUILD_CLEAN=1 python -c "import __graft_entry__ as g; print(g.build())" ) > "$O/clean_build.log" 2>&1
python -m pytest tests -m gpu -q > "$O/pytest_gpu_after_clean_build.log" 2>&1; tail -3 "$O/pytest_gpu_after_clean_build.log"
cat "$O/bigbeam_C2.txt" "$O/bigbeam_C4.txt"; tail -4 "$O/clean_build.log"
