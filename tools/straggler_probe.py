"""Does the align stage of the mixed-length workload wait for its reads with the most seed windows?  The same batch with and without
the reads that bring more than N seed windows (a read's windows are handled one after the other by one lane).
    python tools/straggler_probe.py [threshold] [N]      (GPU box)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from groot_amd import device, synth

t = float(sys.argv[1]) if len(sys.argv) > 1 else 0.99
N = int(sys.argv[2]) if len(sys.argv) > 2 else 16
R = 2_000_000
index, _ = bench.load_index("resfinder.90")
cat, off, lens = synth.reference_sequences(index)
seq, so, _ = synth.reads_np(cat, off, lens, R, 150, min_len=75)


def run(seq, so, n):
    al = device.Aligner(index, threshold=t, max_batch_reads=n, max_read_len=256, max_batch_bases=int(so[-1]) + 64)
    al.set_profiling(True)
    for _ in range(3):
        al.submit(seq, so)
        c = al.wait()
    ms = al.stage_ms()
    sd = al.seeds()
    al.close()
    return ms, c, np.bincount(sd["read_id"], minlength=n)


ms, c, per = run(seq, so, R)
print("all reads      : align %.2f ms, seed stage %.2f ms, max seeds/read %d, reads > %d seeds: %d" % (ms["align"], ms["sketch_seed"], per.max(), N, (per > N).sum()))
keep = np.flatnonzero(per <= N)
lens_r = np.diff(so.astype(np.int64))
idx = np.concatenate([np.arange(so[i], so[i + 1]) for i in keep[:0]]) if False else None
mask = np.repeat(per <= N, lens_r)
seq2 = np.concatenate([seq[: int(so[-1])][mask], np.zeros(64, np.uint8)])
so2 = np.concatenate([[0], np.cumsum(lens_r[keep])]).astype(np.uint64)
ms2, c2, per2 = run(seq2, so2, len(keep))
print("without them   : align %.2f ms, seed stage %.2f ms, max seeds/read %d (%d reads)" % (ms2["align"], ms2["sketch_seed"], per2.max(), len(keep)))
