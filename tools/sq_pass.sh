#!/bin/bash
# SQ instruction / wait counters of the decode kernel of one workload (separate --pmc pass,
# kernel trace only).  usage: tools/sq_pass.sh <outdir under gpurun_out> [bench.py args...]
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$(pwd)}"
O="$R/gpurun_out/$1"; shift
mkdir -p "$O"
cd /tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVES --output-format csv -d "$O/sq1" -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu "$@" > "$O/sq1.log" 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_BRANCH --output-format csv -d "$O/sq2" -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu "$@" > "$O/sq2.log" 2>&1
cd "$R"
python tools/pmc_summary.py $(find "$O" -name "*counter_collection.csv") | tee "$O/sq_summary.txt"
find "$O" -name "*.db" -delete 2>/dev/null
