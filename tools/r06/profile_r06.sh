#!/bin/bash
# Regenerates the measurements kept under profiles/r06/ on an MI355X box (through gpurun, from the repo root; outputs land in
# gpurun_out/r06p/, tools/r06/collect_r06.py copies the summaries).  PMC passes are separate runs with --pmc only (the kernel
# trace is the only trace domain), one counter set per pass.  C2T = C2's shape with a token-level 3-gram LM (round 6).
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$(pwd)}"
O="$R/gpurun_out/r06p"
rm -rf "$O"; mkdir -p "$O"
cd "$R"
python bench.py --steps 20 --warmup 5 > "$O/bench_default.json" 2> "$O/bench_default.err"
python bench.py --workload C2T --steps 20 --warmup 5 --no-secondary > "$O/bench_C2T.json" 2> "$O/bench_C2T.err"
python bench.py --workload C2T --steps 20 --warmup 5 --no-corun --no-extras > "$O/bench_C2T_one_batch_at_a_time.json" 2> "$O/bench_C2T_nocorun.err"
for w in C3 C4; do
  st=10; [ $w = C4 ] && st=6
  python bench.py --workload $w --steps $st > "$O/bench_$w.json" 2> "$O/bench_$w.err"
done
python tools/probe/toklm_probe.py --big > "$O/toklm_probe.txt" 2>&1
for k in 100 200 500; do  # fltx_mlane.h's token-LM variant: two / four / eight lane groups
  python bench.py --workload C2T --beam $k --steps 8 --warmup 2 --no-extras --no-cpu > "$O/bench_C2T_beam$k.json" 2> "$O/bench_C2T_beam$k.err"
done
for w in C2 C2T C3; do  # a stream's kernels, chunk by chunk
  rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_stream_$w" -- python bench.py --workload $w --streaming-only > "$O/stream_only_$w.json" 2> "$O/stream_only_$w.err"
done
python bench.py --workload C2T --steps 5 --warmup 2 --no-extras --no-cpu --pipeline 1 --profile --profile-waves 0,3,7,8 --profile-out "$O/phase_split_C2T.txt" > /dev/null 2>&1
cd /tmp
for w in C2 C2T C3 C4; do
  st=6; [ $w = C4 ] && st=4
  rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_$w" -- python "$R/bench.py" --workload $w --steps $st --warmup 2 --no-cpu --no-secondary --sustained-seconds 0 > "$O/prof_$w.log" 2>&1
done
for w in C2 C2T C3 C4; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch_$w" -- python "$R/bench.py" --workload $w --steps 3 --warmup 1 --no-cpu --no-secondary --sustained-seconds 0 > "$O/pmc_fetch_$w.log" 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/pmc_write_$w" -- python "$R/bench.py" --workload $w --steps 3 --warmup 1 --no-cpu --no-secondary --sustained-seconds 0 > "$O/pmc_write_$w.log" 2>&1
done
for w in C2 C2T; do
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES --output-format csv -d "$O/sq1_$w" -- python "$R/bench.py" --workload $w --steps 3 --warmup 1 --no-cpu --no-secondary --sustained-seconds 0 > "$O/sq1_$w.log" 2>&1
  rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_BRANCH --output-format csv -d "$O/sq2_$w" -- python "$R/bench.py" --workload $w --steps 3 --warmup 1 --no-cpu --no-secondary --sustained-seconds 0 > "$O/sq2_$w.log" 2>&1
done
cd "$R"
find "$O" -name "*.db" -delete 2>/dev/null
find "$O" -name "*_kernel_trace.csv" -size +4M -delete 2>/dev/null  # (the per-dispatch trace: the stats file is what is kept)
du -sh "$O"
# the token LM on lane groups (beam 100): kernel trace + HBM traffic, as for the other workloads
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_C2T_beam100" -- python "$R/bench.py" --workload C2T --beam 100 --steps 6 --warmup 2 --no-cpu --no-secondary --sustained-seconds 0 > "$O/prof_C2T_beam100.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch_C2T_beam100" -- python "$R/bench.py" --workload C2T --beam 100 --steps 3 --warmup 1 --no-cpu --no-secondary --sustained-seconds 0 > "$O/pmc_fetch_C2T_beam100.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/pmc_write_C2T_beam100" -- python "$R/bench.py" --workload C2T --beam 100 --steps 3 --warmup 1 --no-cpu --no-secondary --sustained-seconds 0 > "$O/pmc_write_C2T_beam100.log" 2>&1
cd "$R"
find "$O" -name "*.db" -delete 2>/dev/null
find "$O" -name "*_kernel_trace.csv" -size +4M -delete 2>/dev/null
