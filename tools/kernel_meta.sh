#!/bin/bash
# register / spill / scratch figures of every kernel of libgroot_hip.so (gfx950 code object metadata), read from the objects of its
# four translation units (build/obj/*.o; pass other object files to look at those):  name  scratch  sgprs  sgpr-spills  vgprs  vgpr-spills
set -e
B=/opt/rocm/lib/llvm/bin
for O in ${@:-build/obj/*.o}; do
  T=$(mktemp -d)
  objcopy -O binary --only-section=.hip_fatbin "$O" $T/fat.bin
  TG=$($B/clang-offload-bundler --type=o --input=$T/fat.bin --list | grep gfx950)
  $B/clang-offload-bundler --type=o --targets=$TG --input=$T/fat.bin --output=$T/k.co --unbundle
  $B/llvm-readelf --notes $T/k.co | grep -E "^\s+\.name:|\.vgpr_count|vgpr_spill|sgpr_spill|private_segment_fixed|\.sgpr_count" | paste - - - - - - \
    | sed 's/  */ /g; s/\.name: //; s/\.private_segment_fixed_size: /scratch /; s/\.sgpr_count: /sgpr /; s/\.sgpr_spill_count: /sgpr_spill /; s/\.vgpr_count: /vgpr /; s/\.vgpr_spill_count: /vgpr_spill /' \
    | while read -r name rest; do echo "$(echo $name | c++filt | sed 's/groot:://; s/(.*//' | cut -c1-70) | $rest"; done
  rm -rf $T
done
