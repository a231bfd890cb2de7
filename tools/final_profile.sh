#!/bin/bash
# Regenerates the measurements kept under profiles/<round>/final_* on an MI355X box
# (run through gpurun from the repo root; outputs land in gpurun_out/final/).
# PMC passes are separate runs with --pmc only (no trace domains besides the kernel trace).
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$(pwd)}"
O="$R/gpurun_out/final"
rm -rf "$O"; mkdir -p "$O"
cd "$R"
python -m pytest tests -m gpu -q > "$O/pytest_gpu.log" 2>&1
python bench.py > "$O/bench_C2.json" 2> "$O/bench_C2.err"
python bench.py --workload C3 > "$O/bench_C3.json" 2> "$O/bench_C3.err"
python bench.py --workload C4 --steps 5 > "$O/bench_C4.json" 2> "$O/bench_C4.err"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_C2" -- python "$R/bench.py" --steps 5 --warmup 2 --no-cpu > "$O/prof_C2.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_C3" -- python "$R/bench.py" --workload C3 --steps 3 --warmup 2 --no-cpu > "$O/prof_C3.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch" -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu > "$O/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/pmc_write" -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu > "$O/pmc_write.log" 2>&1
cd "$R"
{
  FLTX_CPU=8 python tools/bench_case.py C3_spell_u0 256 3
  FLTX_CPU=6 python tools/bench_case.py C4z_spell_u0 256 2
  FLTX_CPU=6 python tools/bench_case.py C4_spell_u0 256 2
  python tools/bench_secondary_c2.py
  python tools/e2e_c2.py
  python tools/bench_bigbeam_lexicon.py
} > "$O/secondary.txt" 2>&1
# keep the merge-back small: only the csv summaries
find "$O" -name "*.db" -delete 2>/dev/null
du -sh "$O"
# afterwards, in the container (bench.py looks the traffic up by this exact workload string):
#   python tools/pmc_traffic.py <fetch>_counter_collection.csv <write>_counter_collection.csv \
#       "C2 threads=512 batch=256 T=1000 N=29 beam=50" profiles/rNN/hbm_traffic_C2.json
