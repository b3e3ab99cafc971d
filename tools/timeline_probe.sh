#!/bin/bash
# kernel timeline of one batch of a kernel_path_probe workload, per stream: start offset, duration, wait behind the previous kernel of the same queue (us)
#   tools/timeline_probe.sh WORKLOAD      (GPU box; rocprofv3 --kernel-trace)
set -u
W=${1:-c2_nomemo}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
P=/tmp/tl_$W; rm -rf $P
rocprofv3 --kernel-trace --output-format csv -d $P -o t -- python tools/kernel_path_probe.py $W 4 > /dev/null 2>&1
python - "$P" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last align kernel but one starts the window shown: one whole batch period
al = [i for i, r in enumerate(rows) if "align_kernel" in r["Kernel_Name"]]
lo, hi = int(rows[al[-3]]["Start_Timestamp"]), int(rows[al[-2]]["Start_Timestamp"])
last_end = {}
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    q = r.get("Queue_Id", "?")
    if lo <= s < hi:
        name = r["Kernel_Name"].split("(")[0].replace("groot::", "").replace("void ", "")[:46]
        print("q%-3s +%8.1f us  dur %8.1f  behind-prev-on-queue %8.1f  %s" % (q, (s - lo) / 1e3, (e - s) / 1e3, (s - last_end.get(q, s)) / 1e3, name))
    last_end[q] = max(last_end.get(q, 0), e)
PY
