#!/bin/bash
# quick per-kernel times of the HBM-resident bench loop: tools/ktrace.sh <tag> [extra bench args]  -> gpurun_out/kstats_<tag>.txt
set -u
TAG=${1:-x}; shift || true
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
P=/tmp/prof_$TAG; rm -rf $P; mkdir -p $P gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $P -o t -- python bench.py --steps 5 --warmup 2 --no-cpu --no-cli --no-host-fed "$@" > gpurun_out/ktrace_$TAG.log 2>&1
python - "$P" "$TAG" <<'PY'
import csv, glob, sys
P, tag = sys.argv[1], sys.argv[2]
rows = []
for f in glob.glob(P + "/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
with open("gpurun_out/kstats_%s.txt" % tag, "w") as o:
    for r in rows:
        if "at::native" not in r["Name"] and "elementwise" not in r["Name"]:
            o.write("%-80s calls %5s avg_us %10.1f\n" % (r["Name"].split("(")[0][:80], r["Calls"], float(r["AverageNs"]) / 1e3))
print(open("gpurun_out/kstats_%s.txt" % tag).read())
PY
tail -c 600 gpurun_out/ktrace_$TAG.log
# timeline of the last bench step: start offset, duration, gap to the previous kernel's end (us)
python - "$P" "$TAG" <<'PY'
import csv, glob, sys
P, tag = sys.argv[1], sys.argv[2]
rows = []
for f in glob.glob(P + "/**/*kernel_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last occurrence of the signature kernel starts the last step
idx = [i for i, r in enumerate(rows) if "sketch_sig_kernel" in r["Kernel_Name"]]
if idx:
    i0 = idx[-1]
    while i0 > 0 and int(rows[i0]["Start_Timestamp"]) - int(rows[i0 - 1]["End_Timestamp"]) < 200000 and "order" not in rows[i0 - 1]["Kernel_Name"] and "compact" not in rows[i0-1]["Kernel_Name"]:
        i0 -= 1
    t0 = int(rows[i0]["Start_Timestamp"]); prev = t0
    with open("gpurun_out/timeline_%s.txt" % tag, "w") as o:
        for r in rows[i0:]:
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            o.write("%9.1f +%8.1f gap %7.1f  %s\n" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, r["Kernel_Name"].split("(")[0][:70]))
            prev = e
    print(open("gpurun_out/timeline_%s.txt" % tag).read())
PY
