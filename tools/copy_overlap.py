"""How much of the host-fed leg's copy-in and copy-out ran at the same time?  python tools/copy_overlap.py <dir of rocprofv3 --memory-copy-trace --kernel-trace csv>"""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*memory_copy_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
ks = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    ks += list(csv.DictReader(open(f)))
ev = {"H2D": [], "D2H": []}
for r in rows:
    d = r.get("Direction", "")
    k = "H2D" if "HOST_TO_DEVICE" in d else "D2H" if "DEVICE_TO_HOST" in d else None
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if k and e - s > 200_000:
        ev[k].append((s, e))
blit = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in ks if "copyBuffer" in r["Kernel_Name"] and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 200_000]
print("big copies: H2D %d, D2H %d, blit kernels %d" % (len(ev["H2D"]), len(ev["D2H"]), len(blit)))
for k, v in list(ev.items()) + [("blit", blit)]:
    if v:
        v.sort()
        tail = v[len(v) // 2:]
        print(k, "mean ms %.2f" % (sum(e - s for s, e in tail) / len(tail) / 1e6), "period ms %.2f" % ((tail[-1][0] - tail[0][0]) / max(1, len(tail) - 1) / 1e6))
def overlap(a, b):
    t = 0
    for s1, e1 in a:
        for s2, e2 in b:
            t += max(0, min(e1, e2) - max(s1, s2))
    return t
other = ev["D2H"] + blit
h = ev["H2D"][len(ev["H2D"]) // 2:]
if h and other:
    print("share of copy-in time that ran beside a copy-out: %.2f" % (overlap(h, other) / max(1, sum(e - s for s, e in h))))
    for s, e in h[:4]:
        print("H2D %.2f..%.2f ms" % ((s - h[0][0]) / 1e6, (e - h[0][0]) / 1e6), " out:", ["%.2f..%.2f" % ((s2 - h[0][0]) / 1e6, (e2 - h[0][0]) / 1e6) for s2, e2 in other if e2 > s - 8e6 and s2 < e + 8e6])
