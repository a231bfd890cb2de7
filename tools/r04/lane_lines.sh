#!/bin/bash
# One kernel-time line per lane engine workload (C2, C3, C4, C2 beams 100 / 200, C5 share), then the GPU suite
R="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$R"
line() { python -c "
import json,sys
j=json.loads(sys.stdin.read())
print('$1: ms/step %.3f kernel_ms %.3f engine %s threads %s redone %s' % (j['ms_per_step'], j['roofline']['kernel_ms'], j['config']['engine'], j['config']['threads_per_utterance'], j['config']['redone']))"; }
for w in C2 C3 C4; do python bench.py --workload $w --steps 5 --warmup 2 --no-cpu --no-extras $EXTRA 2>/dev/null | line "$w"; done
for k in 100 200; do python bench.py --workload C2 --beam $k --steps 4 --warmup 2 --no-cpu --no-extras $EXTRA 2>/dev/null | line "C2 beam $k"; done
python bench.py --workload C4 --beam 200 --steps 3 --warmup 2 --no-cpu --no-extras $EXTRA 2>/dev/null | line "C4 beam 200"
python bench.py --workload C5 --batch 1024 --steps 3 --warmup 2 --no-cpu --no-extras $EXTRA 2>/dev/null | line "C5 share"
[ -n "$NOTEST" ] || python -m pytest tests -m gpu -q -x 2>&1 | tail -3
