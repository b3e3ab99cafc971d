#!/bin/bash
# register / spill / scratch figures of every kernel in build/libgroot_hip.so (gfx950 code object metadata)
set -e
T=$(mktemp -d)
objcopy -O binary --only-section=.hip_fatbin "${1:-build/libgroot_hip.so}" $T/fat.bin
B=/opt/rocm/lib/llvm/bin
TG=$($B/clang-offload-bundler --type=o --input=$T/fat.bin --list | grep gfx950)
$B/clang-offload-bundler --type=o --targets=$TG --input=$T/fat.bin --output=$T/k.co --unbundle
$B/llvm-readelf --notes $T/k.co | grep -E "^\s+\.name:|\.vgpr_count|vgpr_spill|sgpr_spill|private_segment_fixed|\.sgpr_count" | paste - - - - - - | sed 's/  */ /g' | cut -c1-300
rm -rf $T
