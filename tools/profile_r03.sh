#!/bin/bash
# Round-3 rocprofv3 evidence for bench.py (run on the GPU box via gpurun from the repo root).
#   1. --kernel-trace --stats  -> profiles/r03_kernel_stats.csv (groot + rocprim kernels of the HBM-resident headline loop)
#   2. PMC passes (own runs, no trace domains): SQ issue counters, then FETCH_SIZE, then WRITE_SIZE, then L2 hits
#      -> profiles/r03_pmc.json (per kernel, per launch; FETCH_SIZE doubled as the gfx950 note in
#         /opt/skills/guides/MI355X_MICROARCH.md prescribes for wide coalesced streams)
#   3. --kernel-trace --stats of the host-fed leg alone (tools/host_fed_probe.py) -> profiles/r03_host_fed_kernel_stats.csv
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
P=/tmp/prof; rm -rf $P; mkdir -p $P profiles gpurun_out
ARGS="--steps 3 --warmup 2 --no-cpu --no-cli --no-host-fed --no-legs"
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
    """per-kernel statistics of the launches of the measured loop only: everything before the first text_lookup_kernel launch is
    groot_hip_open (its capture pass runs the same kernels on the memo's strings) or torch generating the reads"""
    rows = []
    for f in glob.glob(pattern, recursive=True):
        rows = list(csv.DictReader(open(f)))
    starts = [int(r["Start_Timestamp"]) for r in rows if "text_lookup_kernel" in r["Kernel_Name"]]
    t0 = min(starts) - 50000 if starts else 0
    per = collections.OrderedDict()
    for r in sorted(rows, key=lambda r: int(r["Start_Timestamp"])):
        if int(r["Start_Timestamp"]) < t0:
            continue
        name = r["Kernel_Name"].split("(")[0][:90]
        if not ("groot" in name or "rocprim" in name or "rocclr" in name):
            continue
        per.setdefault(name, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    total = sum(sum(v) for v in per.values())
    with open(dest, "w") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for name, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([name, len(v), sum(v), "%.1f" % (sum(v) / len(v)), "%.2f" % (100.0 * sum(v) / max(1, total)), min(v), max(v)])
        w.writerow(["# " + note, "", total])
stats(P + "/trace/**/*kernel_trace.csv", "profiles/r03_kernel_stats.csv", "launches from the first text_lookup_kernel on (the bench loop; groot_hip_open and torch's read generation left out), from rocprofv3 --kernel-trace")
stats(P + "/hf/**/*kernel_trace.csv", "profiles/r03_host_fed_kernel_stats.csv", "host-fed leg (tools/host_fed_probe.py 12 4), launches from the first text_lookup_kernel on")
for f in glob.glob(P + "/hf/**/*memory_copy_stats.csv", recursive=True):
    os.replace(f, "profiles/r03_host_fed_memory_copy_stats.csv")
pmc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(lambda: collections.defaultdict(set))
first_lookup = {}
for d in ("sq", "fetch", "write", "tcc"):
    for f in glob.glob(P + "/%s/**/*counter_collection.csv" % d, recursive=True):
        ids = [int(row["Dispatch_Id"]) for row in csv.DictReader(open(f)) if "text_lookup_kernel" in row["Kernel_Name"]]
        first_lookup[d] = min(ids) if ids else 0
for d in ("sq", "fetch", "write", "tcc"):
    for f in glob.glob(P + "/%s/**/*counter_collection.csv" % d, recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
            if "groot" not in k:
                continue
            # (groot_hip_open's capture pass launches the same kernels on the memo's strings: only dispatches after the first
            # text_lookup_kernel dispatch of the run belong to the bench loop)
            if int(row["Dispatch_Id"]) < first_lookup.get(d, 0):
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
json.dump(out, open("profiles/r03_pmc.json", "w"), indent=1, sort_keys=True)
print(open("profiles/r03_kernel_stats.csv").read())
print(open("profiles/r03_host_fed_kernel_stats.csv").read()[:1500])
PY
cp profiles/r03_*.csv profiles/r03_pmc.json gpurun_out/ 2>/dev/null
tail -1 gpurun_out/prof_trace.log | cut -c1-300
