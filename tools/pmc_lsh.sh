cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/pl; rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU --output-format csv -d /tmp/pl -o c -- python tools/threshold_probe.py > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/pl/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0]
        if "sketch_seed_kernel" in k or "align_kernel" in k:
            acc[(k, r["Dispatch_Id"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for (k,d),v in sorted(acc.items(), key=lambda x:int(x[0][1])):
    s={c:sum(x) for c,x in v.items()}
    if s.get("SQ_WAVES",0) < 1000: continue
    print(k[-60:], d, "waves", int(s["SQ_WAVES"]), "VALU/wave", round(s["SQ_INSTS_VALU"]/s["SQ_WAVES"]), "SALU/wave", round(s["SQ_INSTS_SALU"]/s["SQ_WAVES"]), "VMEM/wave", round(s["SQ_INSTS_VMEM_RD"]/s["SQ_WAVES"]), "wait", round(s["SQ_WAIT_ANY"]/s["SQ_WAVE_CYCLES"],2), "stall", round(s["SQ_WAIT_INST_ANY"]/s["SQ_WAVE_CYCLES"],2))
PY
