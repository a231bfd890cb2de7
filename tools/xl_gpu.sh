#!/bin/sh
# GPU check of the lexicon lane engine: differential runs against the generic engine + C3 phase profile
python tools/cmp_lex_engines.py lexspell 256 1000 50 10 xlane=0 ylane=0 2>&1 | tail -3
python tools/cmp_lex_engines.py uniform 256 1000 50 10 xlane=0 ylane=0 2>&1 | tail -3
python tools/cmp_lex_engines.py lexspell 256 1000 64 29 xlane=0 ylane=0 2>&1 | tail -3
python bench.py --workload C3 --no-cpu --no-extras --profile --profile-waves 0,5,6,7 2>&1 | cut -c1-420 | tail -6
