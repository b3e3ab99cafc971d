#!/bin/bash
# The guard against round 4's lost store, seen failing (run on the GPU box from the repo root; `python __graft_entry__.py rep` built build/rep_bad/ here):
# tests/test_signature_path.py::test_no_read_is_left_out_by_the_seed_stage against the library with -DGROOT_REP_BAD_LIST (the signature kernel in the
# arrangement of c6ce697: vote before the staging, none behind the bad-base check) must FAIL, and against the product it must PASS.  Then
# tools/first_use_check.py on both.  Output: gpurun_out/rep_bad_list.txt (copy to profiles/ if it shows the failure).
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/rep_bad_list.txt; mkdir -p gpurun_out; : > $OUT
T='tests/test_signature_path.py::test_no_read_is_left_out_by_the_seed_stage'
echo "== product build: the guard must pass" >> $OUT
python -m pytest "$T" -m gpu -q -x 2>&1 | tail -3 >> $OUT
echo "== build/rep_bad (-DGROOT_REP_BAD_LIST): the guard must fail" >> $OUT
GROOT_HIP_LIB=build/rep_bad/libgroot_hip.so python -m pytest "$T" -m gpu -q 2>&1 | tail -12 >> $OUT
echo "== tools/first_use_check.py, product" >> $OUT
python tools/first_use_check.py 2>&1 | tail -4 >> $OUT
echo "== tools/first_use_check.py, build/rep_bad" >> $OUT
GROOT_HIP_LIB=build/rep_bad/libgroot_hip.so python tools/first_use_check.py 2>&1 | tail -4 >> $OUT
cat $OUT
