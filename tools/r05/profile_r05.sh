#!/bin/bash
# Regenerates the measurements kept under profiles/r05/ on an MI355X box (through gpurun, from the repo root; outputs
# land in gpurun_out/r05p/, tools/r05/collect_r05.py copies the summaries).  PMC passes are separate runs with --pmc only
# (the kernel trace is the only trace domain), one counter set per pass.
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$(pwd)}"
O="$R/gpurun_out/r05p"
rm -rf "$O"; mkdir -p "$O"
cd "$R"
python bench.py --steps 20 --warmup 5 > "$O/bench_default.json" 2> "$O/bench_default.err"
python bench.py --steps 20 --warmup 5 --no-corun --no-secondary > "$O/bench_C2_one_batch_at_a_time.json" 2> "$O/bench_C2_nocorun.err"
for w in C3 C4; do
  st=10; [ $w = C4 ] && st=6
  python bench.py --workload $w --steps $st > "$O/bench_$w.json" 2> "$O/bench_$w.err"
done
python bench.py --workload C5 --steps 4 --no-extras > "$O/bench_C5_1gpu.json" 2> "$O/bench_C5_1gpu.err"
python bench.py --workload WP --steps 10 --no-extras > "$O/bench_WP_n1024.json" 2> "$O/bench_WP.err"
python tools/probe/corun.py > "$O/corun_C2.txt" 2>&1
tools/r05_plumbing.sh > "$O/plumbing.log" 2>&1
cp "$R"/gpurun_out/r05/bench_process_8x_same_gpu.json "$R"/gpurun_out/r05/bench_group_8x_same_gpu.json "$R"/gpurun_out/r05/c2_geometries.jsonl "$O"/ 2>/dev/null
cd /tmp
for w in C2 C3 C4 WP; do
  st=6; [ $w = C4 ] && st=4
  rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_$w" -- python "$R/bench.py" --workload $w --steps $st --warmup 2 --no-cpu > "$O/prof_$w.log" 2>&1
done
for w in C2 C3 C4; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch_$w" -- python "$R/bench.py" --workload $w --steps 3 --warmup 1 --no-cpu > "$O/pmc_fetch_$w.log" 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/pmc_write_$w" -- python "$R/bench.py" --workload $w --steps 3 --warmup 1 --no-cpu > "$O/pmc_write_$w.log" 2>&1
done
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES --output-format csv -d "$O/sq1_C2" -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu > "$O/sq1_C2.log" 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_BRANCH --output-format csv -d "$O/sq2_C2" -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu > "$O/sq2_C2.log" 2>&1
cd "$R"
find "$O" -name "*.db" -delete 2>/dev/null
du -sh "$O"
