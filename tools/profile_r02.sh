#!/bin/bash
# Round-2 rocprofv3 evidence for bench.py (run on the GPU box via gpurun from the repo root).
#   1. --kernel-trace --stats  -> profiles/r02_kernel_stats.csv (groot + rocprim kernels of the HBM-resident headline loop)
#   2. PMC passes (own runs, no trace domains): SQ issue counters, then FETCH_SIZE, then WRITE_SIZE, then L2 hits
#      -> profiles/r02_pmc.json (per kernel, per launch; FETCH_SIZE doubled as the gfx950 note in
#         /opt/skills/guides/MI355X_MICROARCH.md prescribes for wide coalesced streams)
#   3. --kernel-trace --stats of the host-fed leg alone (tools/host_fed_probe.py) -> profiles/r02_host_fed_kernel_stats.csv
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
P=/tmp/prof; rm -rf $P; mkdir -p $P profiles gpurun_out
ARGS="--steps 3 --warmup 1 --no-cpu --no-cli --no-host-fed"
rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace -o t -- python bench.py $ARGS > gpurun_out/prof_trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU --output-format csv -d $P/sq -o c -- python bench.py $ARGS > gpurun_out/prof_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/fetch -o c -- python bench.py $ARGS > gpurun_out/prof_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $P/write -o c -- python bench.py $ARGS > gpurun_out/prof_write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --output-format csv -d $P/tcc -o c -- python bench.py $ARGS > gpurun_out/prof_tcc.log 2>&1
READS=10000000 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $P/hf -o t -- python tools/host_fed_probe.py 12 4 > gpurun_out/prof_hf.log 2>&1
python - <<'PY'
import csv, glob, json, collections, os
P = "/tmp/prof"
out = {}
def stats(pattern, dest, note):
    rows = []
    for f in glob.glob(pattern, recursive=True):
        rows = list(csv.DictReader(open(f)))
    keep = [r for r in rows if "groot" in r["Name"] or "rocprim" in r["Name"] or "copy" in r["Name"].lower()]
    with open(dest, "w") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in keep:
            w.writerow([r["Name"].split("(")[0][:90]] + [r[k] for k in list(r.keys())[1:]])
        w.writerow(["# " + note, "", sum(int(r["TotalDurationNs"]) for r in rows)])
stats(P + "/trace/**/*kernel_stats.csv", "profiles/r02_kernel_stats.csv", "total of all kernels in the process (incl. torch input generation)")
stats(P + "/hf/**/*kernel_stats.csv", "profiles/r02_host_fed_kernel_stats.csv", "host-fed leg (tools/host_fed_probe.py 12 4): total of all kernels in the process")
for f in glob.glob(P + "/hf/**/*memory_copy_stats.csv", recursive=True):
    os.replace(f, "profiles/r02_host_fed_memory_copy_stats.csv")
pmc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(lambda: collections.defaultdict(set))
for d in ("sq", "fetch", "write", "tcc"):
    for f in glob.glob(P + "/%s/**/*counter_collection.csv" % d, recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
            if "groot" not in k:
                continue
            pmc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            calls[k][row["Counter_Name"]].add(row["Dispatch_Id"])
for k, v in pmc.items():
    n = {c: max(1, len(calls[k][c])) for c in v}
    per = {c: v[c] / n[c] for c in v}
    e = {"launches_profiled": n.get("SQ_WAVES", n.get("FETCH_SIZE", 1)), "per_launch": per}
    if "FETCH_SIZE" in per and "WRITE_SIZE" in per:
        # FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md, HBM section)
        e["hbm_bytes_per_launch"] = (2.0 * per["FETCH_SIZE"] + per["WRITE_SIZE"]) * 1024.0
        e["fetch_kib_raw"] = per["FETCH_SIZE"]; e["write_kib_raw"] = per["WRITE_SIZE"]
    if "TCC_HIT_sum" in per:
        e["l2_hit_rate"] = per["TCC_HIT_sum"] / max(1.0, per["TCC_HIT_sum"] + per["TCC_MISS_sum"])
    out[k.replace("groot::", "")] = e
json.dump(out, open("profiles/r02_pmc.json", "w"), indent=1, sort_keys=True)
print(open("profiles/r02_kernel_stats.csv").read())
print(open("profiles/r02_host_fed_kernel_stats.csv").read()[:1500])
PY
cp profiles/r02_*.csv profiles/r02_pmc.json gpurun_out/ 2>/dev/null
tail -1 gpurun_out/prof_trace.log | cut -c1-300
