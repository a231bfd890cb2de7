#!/bin/bash
# C2 kernel time per slane geometry and "tune" value: tools/r04/c2_geo.sh "<threads...>" "<tunes...>"
R="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$R"; mkdir -p gpurun_out
for w in $1; do for t in $2; do
  python bench.py --steps 10 --warmup 3 --no-cpu --no-extras --set tune=$t --set slane_threads=$w 2> gpurun_out/c2_geo.err | python -c "
import json,sys
j=json.loads(sys.stdin.read())
print('threads $w tune $t: ms/step %.3f kernel_ms %.3f threads %s redone %s' % (j['ms_per_step'], j['roofline']['kernel_ms'], j['config']['threads_per_utterance'], j['config']['redone']))"
done; done
