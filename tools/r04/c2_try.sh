#!/bin/bash
# C2 kernel time for a list of "tune" values (development bits of the lane engines) + the lexicon-free parity slice:
# tools/r04/c2_try.sh 0 1 ...   (one gpurun call)
R="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$R"; mkdir -p gpurun_out
for t in "$@"; do
  python bench.py --steps 10 --warmup 3 --no-extras --cpu-sample 4 --set tune=$t 2> gpurun_out/c2_try.err | python -c "
import json,sys
j=json.loads(sys.stdin.read())
print('tune $t: value %.1f M ms/step %.3f kernel_ms %.3f in-region %.3f mismatches %s' % (j['value']/1e6, j['ms_per_step'], j['roofline']['kernel_ms'], j['roofline']['kernel_ms_in_timed_region'], j['cpu_baseline']['gpu_nbest_mismatches_on_sample']))"
done
python bench.py --workload C2 --steps 2 --warmup 1 --no-cpu --profile --profile-waves 0,1,6,7,8 --profile-out gpurun_out/ps_c2_try.txt --set tune=${1:-0} > /dev/null 2>&1; cut -c60-330 gpurun_out/ps_c2_try.txt
python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -2
