"""Which reads keep the align stage busy longest?  Needs a -DGROOT_WORK_COUNTERS=2 build (python __graft_entry__.py wc 2) and one read per
round (GROOT_DEV_ROUND=1): the library prints the reads whose round took >= 100 wave iterations; this script shows what they are.
    GROOT_HIP_LIB=build/wc2/libgroot_hip.so GROOT_DEV_ROUND=1 python tools/slow_reads_probe.py mixed|sub1 2> err.log; python tools/slow_reads_probe.py mixed|sub1 err.log"""
import os
import re
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from groot_amd import _ffi, device, synth

wl = sys.argv[1]
R = 500000
if wl == "mixed":
    index, _ = bench.load_index("resfinder.90")
    cat, off, lens = synth.reference_sequences(index)
    seq, so, _ = synth.reads_np(cat, off, lens, R, 150, min_len=75)
else:
    index, _ = bench.load_index("arg-annot.90")
    cat, off, lens = synth.reference_sequences(index)
    seq, so, _ = synth.reads_np(cat, off, lens, R, 100)
    rng = np.random.default_rng(7)
    hit = rng.random(len(seq)) < 0.01
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    cur = np.searchsorted(acgt, seq)
    seq = np.where(hit, acgt[(cur + 1 + rng.integers(0, 3, len(seq))) % 4], seq).astype(np.uint8)
    os.environ["GROOT_NO_OUTCOME_TABLE"] = "1"
al = device.Aligner(index, threshold=0.99, max_batch_reads=R, max_read_len=256, max_batch_bases=int(so[-1]) + 64)
al.set_profiling(True)
al.submit(seq, so)
c = al.wait()
print("align ms", al.stage_ms()["align"], "walked", c["walked_reads"])
if len(sys.argv) > 2:
    txt = open(sys.argv[2]).read()
    m = re.findall(r"slow reads \((\d+)\):((?: \d+:\d+:\d+/\d+/\d+)*)", txt)
    n, lst = m[-1]
    pairs, steps = [], {}
    for x in lst.split():
        r_, it_, fsd = x.split(":")
        pairs.append((int(r_), int(it_)))
        steps[int(r_)] = fsd
    sd = al.seeds()
    per = np.bincount(sd["read_id"], minlength=R)
    tr = al.travs()[0]
    ntr = np.bincount(tr["read_id"], minlength=R)
    arr = _ffi.view_arrays(index.view)
    wg = arr["win_graph"]
    cn_off = arr["win_cn_off"]
    L = np.diff(so.astype(np.int64))
    print("slow reads in all:", n)
    for r, it in sorted(pairs, key=lambda p: -p[1])[:30]:
        ws = sd["window_id"][sd["read_id"] == r]
        cn = (cn_off[ws + 1] - cn_off[ws]) if len(ws) else np.zeros(0)
        print("read %7d iterations %5d (FETCH/SCAN/DFS steps %s) len %3d seeds %3d graphs %d traversals %d contained nodes per window: max %d mean %.1f  node len of first seed %d merge_span %d" % (
            r, it, steps[r], L[r], per[r], len(np.unique(wg[ws])), ntr[r], cn.max() if len(cn) else 0, cn.mean() if len(cn) else 0,
            (arr["node_seq_off"][arr["win_node"][ws[0]] + 1] - arr["node_seq_off"][arr["win_node"][ws[0]]]) if len(ws) else 0, arr["win_merge_span"][ws[0]] if len(ws) else 0))
al.close()
