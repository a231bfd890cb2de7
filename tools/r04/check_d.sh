#!/bin/bash
R="${GRAFT_REPO_ROOT:-$(pwd)}"; O="$R/gpurun_out/r04d"; mkdir -p "$O"; cd "$R"
python -m pytest tests/test_gpu_batches.py -q -k "four_lane or long_utterance or with_lm_terms or c4_batch or c5_share" > "$O/pytest_y4.log" 2>&1; tail -8 "$O/pytest_y4.log"
tools/r04/bigbeam_c4.sh "$O/bigbeam_C4.jsonl" 100 128 129 160 200 256
python -m pytest tests/test_streaming.py -q -x -k "device_chunk" > "$O/pytest_stream.log" 2>&1; tail -3 "$O/pytest_stream.log"
