"""Mixed read lengths (BASELINE configs[4] in miniature: reads of 80..150 bases against the w=100 index): stage times.
    python tools/mixed_probe.py [reads] [threshold]      (GPU box)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from groot_amd import device, synth

R = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
t = float(sys.argv[2]) if len(sys.argv) > 2 else 0.99
index, _ = bench.load_index()
cat, off, lens = synth.reference_sequences(index)
seq, so, _ = synth.reads_np(cat, off, lens, R, 150, min_len=80)
al = device.Aligner(index, threshold=t, max_batch_reads=R, max_read_len=256, max_batch_bases=int(so[-1]) + 64, results_on_device=True)
al.set_profiling(True)
for _ in range(3):
    al.submit(seq, so)
    c = al.wait()
ms = al.stage_ms()
print({"reads": R, "threshold": t, "Mreads_s_kernels": round(R / ms["total"] / 1e3, 1), "stage_ms": {k: round(v, 2) for k, v in ms.items()},
       "mapped": c["mapped"], "seeds_per_read": round(c["seeds"] / R, 2), "full_sketch_reads": c["full_sketch_reads"]})
al.close()
