#!/bin/bash
# SQ counters of the align stage's kernels per library variant (run on the GPU box):  tools/pmc_ab.sh WORKLOAD dir1 dir2 ...  ("." = the product build)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
WL=$1; shift
for L in "$@"; do
  LIB=build/$L/libgroot_hip.so; [ "$L" = "." ] && LIB=build/libgroot_hip.so
  rm -rf /tmp/pab/$L
  GROOT_HIP_LIB=$LIB PROBE_SERIAL=1 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU --output-format csv -d /tmp/pab/$L -o c -- python tools/kernel_path_probe.py $WL 2 > /dev/null 2>&1
  python - "$L" $(find /tmp/pab/$L -name "*counter_collection.csv") <<'PY'
import csv, sys, collections
tag, f = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    for name in ("align_lean_kernel", "align_kernel", "sketch_sig_kernel"):
        if name + "<" in k:
            acc[name][r["Counter_Name"]] += float(r["Counter_Value"]); n[name].add(r["Dispatch_Id"])
for k, v in acc.items():
    m = max(1, len(n[k]))
    print("== %s %s (%d launches): " % (tag, k, m) + ", ".join("%s %.1fM" % (c.replace("SQ_", ""), x / m / 1e6) for c, x in sorted(v.items())))
PY
done
