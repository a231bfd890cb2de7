#!/bin/bash
# "tune" bit 0 (priority of the waves the token waves wait for) on every lane engine + the C2 geometries; then the GPU suite
R="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$R"
tools/r04/c2_geo.sh "384 448 512 576 640" "0 1"
line() { python -c "
import json,sys
j=json.loads(sys.stdin.read())
print('$1: ms/step %.3f kernel_ms %.3f engine %s threads %s' % (j['ms_per_step'], j['roofline']['kernel_ms'], j['config']['engine'], j['config']['threads_per_utterance']))"; }
for w in C3 C4; do for t in 0 1; do python bench.py --workload $w --steps 4 --warmup 2 --no-cpu --no-extras --set tune=$t 2>/dev/null | line "$w tune $t"; done; done
for k in 100 200; do for t in 0 1; do python bench.py --workload C2 --beam $k --steps 4 --warmup 2 --no-cpu --no-extras --set tune=$t 2>/dev/null | line "C2 beam $k tune $t"; done; done
for t in 0 1; do python bench.py --workload C5 --batch 1024 --steps 3 --warmup 2 --no-cpu --no-extras --set tune=$t 2>/dev/null | line "C5 share tune $t"; done
python -m pytest tests -m gpu -q -x 2>&1 | tail -3
