#!/bin/bash
# a variant of libgroot_hip.so that differs in the align kernel's compile-time knobs only:  tools/variant.sh NAME -DGROOT_STARVE_MAX=4 ...
# -> build/v_NAME/libgroot_hip.so (the other three translation units are the product's objects); use with GROOT_HIP_LIB
set -e
N=$1; shift
mkdir -p build/v_$N
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-pass-failed -Iinclude -Igroot_amd/csrc/hip "$@" -c -o build/v_$N/align.o groot_amd/csrc/hip/align.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/v_$N/libgroot_hip.so build/obj/groot_hip.o build/obj/seed_full.o build/obj/seed_fast.o build/v_$N/align.o
echo build/v_$N/libgroot_hip.so
