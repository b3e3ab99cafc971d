#!/bin/bash
# kernel timeline of the LAST batch of any probe: tools/ktrace_cmd.sh <tag> <python script and args...>  -> gpurun_out/timeline_<tag>.txt
set -u
TAG=${1:-x}; shift
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
P=/tmp/prof_$TAG; rm -rf $P; mkdir -p $P gpurun_out
rocprofv3 --kernel-trace --output-format csv -d $P -o t -- python "$@" > gpurun_out/ktrace_$TAG.log 2>&1
python - "$P" "$TAG" <<'PY'
import csv, glob, sys
P, tag = sys.argv[1], sys.argv[2]
rows = []
for f in glob.glob(P + "/**/*kernel_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "align_kernel" in r["Kernel_Name"]]
if idx:
    last = idx[-1]
    i0 = idx[-2] + 1 if len(idx) > 1 else 0
    # skip the ordering kernels of the batch before
    while i0 < last and ("order" in rows[i0]["Kernel_Name"] or "mask_" in rows[i0]["Kernel_Name"] or "trav_pack" in rows[i0]["Kernel_Name"]): i0 += 1
    t0 = int(rows[i0]["Start_Timestamp"]); prev = t0
    with open("gpurun_out/timeline_%s.txt" % tag, "w") as o:
        for r in rows[i0:]:
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            if e - s > 3000: o.write("%9.1f +%8.1f gap %7.1f  %s\n" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, r["Kernel_Name"].split("(")[0][:80]))
            prev = e
    print(open("gpurun_out/timeline_%s.txt" % tag).read())
PY
