#!/bin/bash
# Round-6 soak on the GPU against the oracle (one gpurun call; log -> profiles/r06/soak_r06.log).  Every tool classifies a
# mismatch by the ORACLE's own tie counters (oracle.cpp TieCounts), never by "the compiled reference disagrees too".
R="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$R"
python tools/r06/toklm_soak.py "${SOAK_TOKLM:-6000}"
SOAK_HOST="${SOAK_HOST:-900}" SOAK_DEFER="${SOAK_DEFER:-1500}" SOAK_STREAM="${SOAK_STREAM:-90}" python tools/r05/soak_r05.py
FLTX_FUZZ_N="${SOAK_FUZZ:-3000}" python tools/fuzz_big.py | tail -5
python tools/r05/multilabel_soak.py "${SOAK_ML:-1500}" | tail -3
