for wl in C3 C4; do
for extra in "" "--threads 256 --set lds_budget=81920" "--threads 256 --set lds_budget=53000" "--threads 512 --set lds_budget=81920"; do
  echo "== $wl generic $extra"
  python bench.py --workload $wl --steps 8 --warmup 2 --no-extras --no-cpu --set xlane=0 --set ylane=0 $extra 2>/dev/null | python -c "
import json,sys
o=json.loads(sys.stdin.read()); c=o['config']; r=o['roofline']
print(round(o['value']/1e6,2),'M', round(o['ms_per_step'],2),'ms/step kernel', round(r['kernel_ms'],2), 'alone', round(r['kernel_ms_alone'],2), 'engine', c['engine'], 'threads', c['threads_per_utterance'], 'lds', c['lds_bytes_per_workgroup'], 'redone', c['redone'], c['unread_redone'])"
done; done
