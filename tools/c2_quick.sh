#!/bin/sh
# tools/c2_quick.sh [threads...] -- C2 parity slice + kernel time per slane geometry (one gpurun call)
python -m pytest tests/test_gpu_parity.py tests/test_gpu_batches.py -m gpu -x -q -k "lf_ or C1 or C2 or lexfree or lane" 2>&1 | tail -3
for w in "$@"; do
  echo "threads $w"
  python bench.py --steps 10 --warmup 2 --no-cpu --set slane_threads=$w 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['threads_per_utterance'], d['config']['redone'])"
done
