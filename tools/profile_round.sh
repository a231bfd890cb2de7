#!/bin/bash
# Regenerates the measurements kept under profiles/<round>/ on an MI355X box: tools/profile_round.sh r03 (through gpurun,
# from the repo root; outputs land in gpurun_out/<round>/, tools/collect_round.py copies the summaries).  PMC passes are
# separate runs with --pmc only (kernel trace is the only trace domain), one counter set per pass.
ROUND="${1:-r03}"
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$(pwd)}"
O="$R/gpurun_out/${ROUND}"
rm -rf "$O"; mkdir -p "$O"
cd "$R"
for w in C2 C3 C4; do
  st=5; [ $w = C4 ] && st=3
  python bench.py --workload $w --steps $st > "$O/bench_$w.json" 2> "$O/bench_$w.err"
done
python bench.py --workload C5 --steps 4 --no-extras > "$O/bench_C5_1gpu.json" 2> "$O/bench_C5_1gpu.err"
cd /tmp
# C5's share of one GPU (1 024 utterances): traffic only (--batch keeps the 8 192-utterance pass out of the PMC runs)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch_C5" -- python "$R/bench.py" --workload C5 --batch 1024 --steps 2 --warmup 1 --no-cpu > "$O/pmc_fetch_C5.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/pmc_write_C5" -- python "$R/bench.py" --workload C5 --batch 1024 --steps 2 --warmup 1 --no-cpu > "$O/pmc_write_C5.log" 2>&1
for w in C2 C3 C4; do
  st=5; [ $w = C4 ] && st=3
  rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_$w" -- python "$R/bench.py" --workload $w --steps $st --warmup 2 --no-cpu > "$O/prof_$w.log" 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch_$w" -- python "$R/bench.py" --workload $w --steps 3 --warmup 1 --no-cpu > "$O/pmc_fetch_$w.log" 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/pmc_write_$w" -- python "$R/bench.py" --workload $w --steps 3 --warmup 1 --no-cpu > "$O/pmc_write_$w.log" 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES --output-format csv -d "$O/sq1_$w" -- python "$R/bench.py" --workload $w --steps 3 --warmup 1 --no-cpu > "$O/sq1_$w.log" 2>&1
  rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_BRANCH --output-format csv -d "$O/sq2_$w" -- python "$R/bench.py" --workload $w --steps 3 --warmup 1 --no-cpu > "$O/sq2_$w.log" 2>&1
done
cd "$R"
find "$O" -name "*.db" -delete 2>/dev/null
du -sh "$O"
