#!/bin/sh
# tools/kstats.sh <object.o> -- registers / LDS / scratch of every kernel in a hipcc object (gfx950 code object notes)
set -e
LLVM=/opt/rocm/lib/llvm/bin
tmp=$(mktemp -d)
$LLVM/llvm-objcopy --dump-section .hip_fatbin=$tmp/fat "$1"
$LLVM/clang-offload-bundler --unbundle --type=o --input=$tmp/fat --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$tmp/co
$LLVM/llvm-readelf --notes $tmp/co | grep -E "^ +\.name:|\.vgpr_count|\.agpr_count|\.sgpr_count|group_segment_fixed|private_segment_fixed|spill_count" |
  awk '/\.name:/{if(n)print n, s; n=$2; s=""} !/\.name:/{s=s" "$1$2} END{print n, s}' | sed 's/_Z[0-9]*//'
rm -rf $tmp
