#!/bin/bash
R="${GRAFT_REPO_ROOT:-$(pwd)}"; O="$R/gpurun_out/r04c"; mkdir -p "$O"; cd "$R"
python -m pytest tests/test_streaming.py -q -x -k "device_chunk" > "$O/pytest_stream.log" 2>&1; tail -3 "$O/pytest_stream.log"
python -m pytest tests/test_gpu_parity.py -q -k "tied or engine_selection" > "$O/pytest_tied.log" 2>&1; tail -8 "$O/pytest_tied.log"
python -m pytest tests/test_gpu_batches.py -q -k "four_lane or long_utterance or with_lm_terms or c4_batch or c5_share" > "$O/pytest_y4.log" 2>&1; tail -8 "$O/pytest_y4.log"
tools/r04/bigbeam_c4.sh "$O/bigbeam_C4.jsonl"
for w in C2 C3 C4; do python bench.py --workload $w --steps 2 --warmup 1 --no-cpu --profile --profile-waves 0,1,6,7,8 --profile-out "$O/phase_split_$w.txt" > /dev/null 2>> "$O/prof.err"; done
cat "$O"/phase_split_*.txt; tail -3 "$O/prof.err"
