#!/bin/bash
# a variant of libgroot_hip.so that differs in ONE translation unit's compile-time knobs:  tools/variant.sh NAME TU -DFLAG=...   (TU: align | seed_fast | seed_full | groot_hip)
# -> build/v_NAME/libgroot_hip.so (the other translation units are the product's objects in build/obj); use with GROOT_HIP_LIB, several in one gpurun call
set -e
N=$1; TU=$2; shift 2
mkdir -p build/v_$N
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-pass-failed -Iinclude -Igroot_amd/csrc/hip "$@" -c -o build/v_$N/$TU.o groot_amd/csrc/hip/$TU.hip
OBJS=""
for t in groot_hip seed_full seed_fast align; do if [ $t = $TU ]; then OBJS="$OBJS build/v_$N/$t.o"; else OBJS="$OBJS build/obj/$t.o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/v_$N/libgroot_hip.so $OBJS
echo build/v_$N/libgroot_hip.so
