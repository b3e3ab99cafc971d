#!/bin/bash
# quick per-kernel timing of the HBM-resident loop: tools/kstats.sh [extra bench.py args]; prints the groot kernels' averages
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
P=/tmp/kstats; rm -rf $P; mkdir -p $P gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $P -o t -- python bench.py --steps 5 --warmup 2 --no-cpu --no-cli --no-host-fed "$@" > gpurun_out/kstats_run.log 2>&1
python - <<'PY'
import csv, glob
for f in glob.glob("/tmp/kstats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "groot" in r["Name"] or "rocprim" in r["Name"]:
            print("%-110s calls %4s avg %10.3f us  %5s%%" % (r["Name"][:110], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
