"""Timeline of the host-fed leg from a rocprofv3 trace (csv, --kernel-trace --memory-copy-trace): every copy and kernel longer than
`min_us` in a window of the steady state, with the queue it ran on.   python tools/hostfed_timeline.py <dir> [window_ms] [min_us]"""
import csv
import glob
import sys

d = sys.argv[1]
win = float(sys.argv[2]) if len(sys.argv) > 2 else 22.0
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 100.0
ev = []
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        k = "H2D" if "HOST_TO_DEVICE" in r["Direction"] else "D2H" if "DEVICE_TO_HOST" in r["Direction"] else "D2D"
        ev.append((s, e, "copy " + k, "-"))
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        ev.append((s, e, r["Kernel_Name"].split("(")[0].split("<")[0][-40:], r.get("Queue_Id", "?")))
ev.sort()
big = [x for x in ev if x[2] == "copy H2D" and x[1] - x[0] > 1_000_000]
if not big:
    sys.exit("no large copies found")
t0 = big[len(big) * 2 // 3][0]
print("window starts at the H2D number %d of %d" % (len(big) * 2 // 3, len(big)))
for s, e, n, q in ev:
    if s >= t0 and s < t0 + win * 1e6 and e - s >= min_us * 1e3:
        print("%8.3f .. %8.3f  %7.3f ms  q=%-3s %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, q, n))
