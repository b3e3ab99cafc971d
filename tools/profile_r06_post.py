#!/usr/bin/env python3
"""post-processing of tools/profile_r06.sh: rocprofv3 CSVs -> profiles/r06_kernel_stats.csv, profiles/r06_pmc.json.

The probe runs the measured loop LAST (steps + 2 warm-up batches behind groot_hip_open, whose capture pass launches the same kernels on the
memo's strings).  Every batch launches exactly one assign_q_rows_kernel: the launches of the loop are those from the (steps + 2)-th last
assign_q_rows_kernel launch on, minus the seed-stage kernels in front of it -- cut at the last memset-free gap instead: everything after
the end of the (steps + 3)-th last order_ovf_kernel (the last kernel of the batch before the loop) belongs to the loop."""
import collections
import csv
import glob
import json
import os
import sys

P, workloads = sys.argv[1], sys.argv[2:]
TAG = os.environ.get("PROF_TAG", "r06")
STEPS = {"trace": 4 + 2, "pmc": 2 + 2}      # batches of the loop incl. the two warm-up batches


def short(name):
    n = name.replace("void ", "").replace("groot::", "")
    base = n.split("(")[0]
    if base.startswith("sketch_seed_kernel"):      # keep the LIST flag apart: the list pass is a different launch shape
        args = base[base.index("<") + 1: base.rindex(">")].split(",")
        return "sketch_seed_kernel<LIST>" if len(args) > 4 and args[4].strip() == "true" else "sketch_seed_kernel"
    if base.startswith("rocprim"):
        return "rocprim::" + base.split("::")[-1].split("<")[0]
    return base.split("<")[0]


OPEN_ONLY = ("text_argmin_kernel", "sketch_equal_kernel", "text_table_fill_kernel")    # launched by groot_hip_open only


def loop_rows(rows, n_batches, key):
    """the dispatches of the last n_batches batches: every batch ends with one order_ovf_kernel, so the loop starts behind the
    (n_batches + 1)-th last of them (a capture batch of groot_hip_open) -- or, for a ctx without memo, behind the last kernel that only
    groot_hip_open launches, whichever comes later"""
    rows = sorted(rows, key=key)
    ends = [i for i, r in enumerate(rows) if "order_ovf_kernel" in r["Kernel_Name"]]
    cut = ends[-(n_batches + 1)] if len(ends) > n_batches else -1
    for i, r in enumerate(rows):
        if any(k in r["Kernel_Name"] for k in OPEN_ONLY):
            cut = max(cut, i)
    return rows[cut + 1:]


stats_rows = []
pmc_out = {}
for w in workloads:
    for f in glob.glob(os.path.join(P, w, "trace", "**", "*kernel_trace.csv"), recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "groot" in r["Kernel_Name"] or "rocprim" in r["Kernel_Name"]]
        rows = loop_rows(rows, STEPS["trace"], lambda r: int(r["Start_Timestamp"]))
        per = collections.OrderedDict()
        for r in rows:
            per.setdefault(short(r["Kernel_Name"]), []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        total = sum(sum(v) for v in per.values())
        span = (max(int(r["End_Timestamp"]) for r in rows) - min(int(r["Start_Timestamp"]) for r in rows)) if rows else 0
        for name, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
            stats_rows.append([w, name, len(v), sum(v), "%.1f" % (sum(v) / len(v)), "%.2f" % (100.0 * sum(v) / max(1, total)), min(v), max(v)])
        stats_rows.append([w, "# sum of kernel durations / first start .. last end of the loop (overlap of the two streams shows as sum > span)", "", total, "", "", "", span])
    pmc = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.defaultdict(lambda: collections.defaultdict(set))
    for d in ("sq", "fetch", "write", "tcc"):
        for f in glob.glob(os.path.join(P, w, d, "**", "*counter_collection.csv"), recursive=True):
            rows = [r for r in csv.DictReader(open(f)) if "groot" in r["Kernel_Name"]]
            # (one row per dispatch and counter: cut on the dispatch ids)
            per_dispatch = {}
            for r in rows:
                per_dispatch.setdefault(int(r["Dispatch_Id"]), r)
            keep = {int(r["Dispatch_Id"]) for r in loop_rows(list(per_dispatch.values()), STEPS["pmc"], lambda r: int(r["Dispatch_Id"]))}
            for r in rows:
                if int(r["Dispatch_Id"]) not in keep:
                    continue
                k = short(r["Kernel_Name"])
                pmc[k][r["Counter_Name"]] += float(r["Counter_Value"])
                calls[k][r["Counter_Name"]].add(r["Dispatch_Id"])
    out = {}
    for k, v in pmc.items():
        n = {c: max(1, len(calls[k][c])) for c in v}
        per = {c: v[c] / n[c] for c in v}
        e = {"launches_profiled": n.get("SQ_WAVES", n.get("FETCH_SIZE", 1)), "per_launch": per}
        if "FETCH_SIZE" in per and "WRITE_SIZE" in per:
            # FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md, HBM section)
            e["hbm_bytes_per_launch"] = (2.0 * per["FETCH_SIZE"] + per["WRITE_SIZE"]) * 1024.0
            e["fetch_kib_raw"] = per["FETCH_SIZE"]
            e["write_kib_raw"] = per["WRITE_SIZE"]
        if "TCC_HIT_sum" in per:
            e["l2_hit_rate"] = per["TCC_HIT_sum"] / max(1.0, per["TCC_HIT_sum"] + per["TCC_MISS_sum"])
        out[k] = e
    # average duration per kernel from the trace of the same workload
    for row in stats_rows:
        if row[0] == w and row[1] in out:
            out[row[1]]["avg_ms"] = float(row[4]) / 1e6
    pmc_out[w] = out

os.makedirs("profiles", exist_ok=True)
old = {}
if os.path.exists("profiles/%s_pmc.json" % TAG):
    try:
        old = json.load(open("profiles/%s_pmc.json" % TAG))
    except Exception:
        old = {}
old.update(pmc_out)
old["_meta"] = {"commit": os.environ.get("PROF_COMMIT", ""), "tool": "tools/profile_r06.sh"}
json.dump(old, open("profiles/%s_pmc.json" % TAG, "w"), indent=1, sort_keys=True)
keep = []
if os.path.exists("profiles/%s_kernel_stats.csv" % TAG):
    for r in csv.reader(open("profiles/%s_kernel_stats.csv" % TAG)):
        if r and r[0] != "Workload" and r[0] not in workloads:
            keep.append(r)
with open("profiles/%s_kernel_stats.csv" % TAG, "w") as f:
    wr = csv.writer(f)
    wr.writerow(["Workload", "Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs/SpanNs"])
    for r in keep + stats_rows:
        wr.writerow(r)
print(open("profiles/%s_kernel_stats.csv" % TAG).read())
