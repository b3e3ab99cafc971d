"""Which reads keep the align stage busy longest?  Run with a -DGROOT_WORK_COUNTERS=2 build (GROOT_HIP_LIB) and GROOT_ROUND_LANES=1:
the library prints the reads whose round took >= 100 wave iterations; this script shows what they are.
    GROOT_HIP_LIB=build/wc2/libgroot_hip.so GROOT_ROUND_LANES=1 python tools/slow_reads_probe.py 2> err.log; python tools/slow_reads_probe.py err.log"""
import os
import re
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from groot_amd import _ffi, device, synth

R = 500000
index, _ = bench.load_index("resfinder.90")
cat, off, lens = synth.reference_sequences(index)
seq, so, _ = synth.reads_np(cat, off, lens, R, 150, min_len=75)
al = device.Aligner(index, threshold=0.99, max_batch_reads=R, max_read_len=256, max_batch_bases=int(so[-1]) + 64)
al.set_profiling(True)
al.submit(seq, so)
c = al.wait()
print("align ms", al.stage_ms()["align"])
if len(sys.argv) > 1:
    txt = open(sys.argv[1]).read()
    m = re.findall(r"slow reads \((\d+)\):((?: \d+:\d+)*)", txt)
    n, lst = m[-1]
    pairs = [tuple(map(int, x.split(":"))) for x in lst.split()]
    sd = al.seeds()
    per = np.bincount(sd["read_id"], minlength=R)
    tr = al.travs()[0]
    ntr = np.bincount(tr["read_id"], minlength=R)
    wg = _ffi.view_arrays(index.view)["win_graph"]
    L = np.diff(so.astype(np.int64))
    print("slow reads in all:", n)
    for r, it in sorted(pairs, key=lambda p: -p[1])[:25]:
        ws = sd["window_id"][sd["read_id"] == r]
        print("read %7d iterations %5d len %3d seeds %3d graphs %d traversals %d" % (r, it, L[r], per[r], len(np.unique(wg[ws])), ntr[r]))
al.close()
