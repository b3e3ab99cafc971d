#!/bin/bash
# per-read counts of one large resident batch (8-10 M reads) across seeding paths and align-kernel builds: product | GROOT_NO_SIG=1 (full-width hashing only) |
# build/v_plain (tools/variant.sh plain align -DGROOT_NO_LEVEL4_RULE=1 -DGROOT_FORK_MIN=100000000: no level-4 rule, no fork) -- run on the GPU box from the repo root
for w in mixed99 sub1 mixed90; do
  python tools/cross_check.py $w /tmp/cc_${w}_product.npz 2>&1 | grep -v amdgpu
  GROOT_NO_SIG=1 python tools/cross_check.py $w /tmp/cc_${w}_nosig.npz 2>&1 | grep -v amdgpu
  GROOT_HIP_LIB=build/v_plain/libgroot_hip.so python tools/cross_check.py $w /tmp/cc_${w}_plainalign.npz 2>&1 | grep -v amdgpu
  python tools/cross_check.py --diff /tmp/cc_${w}_product.npz /tmp/cc_${w}_nosig.npz /tmp/cc_${w}_plainalign.npz
done
