#!/bin/bash
# PMC pass over a short bench run: prints per-kernel per-launch counters of the groot kernels (run on the GPU box via gpurun)
#   tools/pmc_probe.sh "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU ..." [extra bench args]
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
P=/tmp/pmcprobe; rm -rf $P; mkdir -p $P
rocprofv3 --pmc $1 --output-format csv -d $P -o c -- python bench.py --steps 2 --warmup 1 --no-cpu --no-cli --no-host-fed ${2:-} > $P/log.txt 2>&1
python - <<'PY'
import csv, glob, collections
pmc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob("/tmp/pmcprobe/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        if "groot" not in k: continue
        pmc[k][row["Counter_Name"]] += float(row["Counter_Value"]); calls[k][row["Counter_Name"]].add(row["Dispatch_Id"])
for k, v in pmc.items():
    print(k[:60], {c: round(v[c] / max(1, len(calls[k][c]))) for c in v})
PY
