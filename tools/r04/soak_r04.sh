#!/bin/bash
# Differential runs on the final code of round 4 (beyond the -m gpu suite): random configurations on the default engines
# and on the shared-CU geometry against the oracle, whole C3 / C4 batches on the lane engines against the generic engine,
# the C2 batch on two geometries, random word-piece configurations.  One gpurun call; output = profiles/r04/soak_r04.log.
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
echo "== fuzz_big, 5000 random configurations, default engines"; FLTX_FUZZ_N=5000 timeout 900 python tools/fuzz_big.py 2>&1 | tail -3
echo "== fuzz_big, 2000 configurations, lane engines on the shared-CU geometry (yshare=1)"; FLTX_FUZZ_N=2000 FLTX_FUZZ_SET="yshare=1" timeout 600 python tools/fuzz_big.py 2>&1 | tail -3
echo "== C3 / C4 batches of 256, lane engines (memo in LDS, memo in HBM) vs the generic engine, every utterance"
timeout 300 python tools/cmp_workload.py C3 256 2>&1 | tail -2
timeout 300 python tools/cmp_workload.py C4 256 2>&1 | tail -2
timeout 300 python tools/cmp_workload.py C4 256 yshare=1 2>&1 | tail -2
echo "== C2 batch, engine 4 (576 threads) vs engine 3, and the 512-thread geometry"
timeout 300 python tools/cmp_engines.py 2>&1 | tail -2
timeout 300 python tools/cmp_engines.py ctc 256 1000 50 slane_threads=512 2>&1 | tail -2
echo "== word-piece soak"; timeout 400 python tools/r04/wp_soak.py 200 2>&1 | tail -3
