#!/bin/sh
# Build the host-thread EMULATION of the kernels (tests only; see hip_emu.h).
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
g++ -std=c++17 -O1 -fPIC -shared -DFLTX_EMU $EMU_EXTRA -ffp-contract=off -Wall -Wno-unused-function -Wno-unknown-pragmas -Wno-unused-variable -Wno-unused-but-set-variable \
    -I"$HERE" -I"$ROOT/include" -I"$ROOT/text_amd/csrc" \
    "$ROOT/text_amd/csrc/fltx_api.cpp" "$ROOT/text_amd/csrc/fltx_host_trie.cpp" "$ROOT/text_amd/csrc/fltx_arpa.cpp" "$ROOT/text_amd/csrc/fltx_group.cpp" "$HERE/hip_emu.cpp" \
    -o "${EMU_OUT:-$HERE/libfltx_emu.so}" -lpthread
