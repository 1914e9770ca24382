#!/bin/bash
# usage: tools/kernel_regs.sh <object built by hipcc -c> [name filter]: VGPRs, spills and scratch bytes per kernel (gfx950 code object)
set -e
O=$1; F=${2:-k4k_}
T=$(mktemp -d)
/opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=$T/f.bin $O
TG=$(/opt/rocm/lib/llvm/bin/clang-offload-bundler --list --type=o --input=$T/f.bin | grep gfx950)
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$T/f.bin --targets=$TG --output=$T/k.co
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/k.co | grep -E "\.name:|\.vgpr_count|vgpr_spill|private_segment_fixed|\.group_segment_fixed" | paste - - - - - | sed 's/ \+/ /g' | grep -i "$F" | sed 's/_Z[0-9]*//' | cut -c1-260
rm -rf $T
