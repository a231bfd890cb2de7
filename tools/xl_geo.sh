#!/bin/sh
# C3 on the geometries of the lexicon lane engine
for t in 512 640; do
  echo "threads $t"
  python bench.py --workload C3 --no-cpu --no-extras --set slane_threads=$t 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['threads_per_utterance'], d['config']['engine'])"
done
python tools/cmp_lex_engines.py lexspell 256 1000 50 10 xlane=0 2>&1 | tail -3
python tools/cmp_lex_engines.py lexspell 256 1000 50 16 xlane=0 2>&1 | tail -3
