#!/bin/bash
# SQ counters of align_kernel for a probe command: tools/pmc_align.sh <python script and args...>
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/pa /tmp/pb; 
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/pa -o c -- python "$@" > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR SQ_INSTS_SMEM TCC_HIT_sum TCC_MISS_sum --output-format csv -d /tmp/pb -o c -- python "$@" > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections
acc=collections.defaultdict(lambda: collections.defaultdict(float))
for d in ("/tmp/pa", "/tmp/pb"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].split("(")[0]
            if "align_kernel" in k or "sketch_seed_kernel" in k or "text_lookup" in k or "order_first" in k:
                acc[(k[-50:], r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
last = {}
for (k,d),v in sorted(acc.items(), key=lambda x:int(x[0][1])):
    last.setdefault(k, []).append(v)
for k, vs in last.items():
    # counters of the two passes belong to different dispatch ids: report the last dispatch of each pass
    a = [v for v in vs if "SQ_WAVES" in v][-1]; b = [v for v in vs if "SQ_INSTS_LDS" in v][-1]
    w = a["SQ_WAVES"]
    print(k, "waves", int(w), "VALU/w", round(a["SQ_INSTS_VALU"]/w), "SALU/w", round(a["SQ_INSTS_SALU"]/w), "VMEM_RD/w", round(a["SQ_INSTS_VMEM_RD"]/w),
          "wait", round(a["SQ_WAIT_ANY"]/a["SQ_WAVE_CYCLES"],2), "stall", round(a["SQ_WAIT_INST_ANY"]/a["SQ_WAVE_CYCLES"],2), "active", round(a["SQ_ACTIVE_INST_ANY"]/a["SQ_WAVE_CYCLES"],2),
          "cycles/w", round(a["SQ_WAVE_CYCLES"]/w), "LDS/w", round(b["SQ_INSTS_LDS"]/w), "bank_conf", int(b["SQ_LDS_BANK_CONFLICT"]), "VMEM_WR/w", round(b["SQ_INSTS_VMEM_WR"]/w),
          "L2hit", round(b["TCC_HIT_sum"]/max(1,b["TCC_HIT_sum"]+b["TCC_MISS_sum"]),2))
PY
