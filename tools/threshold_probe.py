#!/usr/bin/env python
"""Throughput of the general LSH-Forest path: 2 M synthetic 100 bp reads at containment thresholds 0.99 .. 0.90 (GPU box)."""
import sys, time, numpy as np, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from groot_amd import device, synth
dev = torch.device("cuda", 0)
index, _ = bench.load_index()
cat, off, lens = synth.reference_sequences(index)
cat_t, off_t, lens_t = (torch.from_numpy(x).to(dev) for x in (cat, off, lens))
R, L = 2_000_000, 100
p, _, _ = synth.reads_torch(cat_t, off_t, lens_t, R, L, first=0)
d_seq = torch.zeros(R * L + 64, dtype=torch.uint8, device=dev); d_seq[:R*L] = p[:R*L]
d_off = torch.arange(0, R + 1, dtype=torch.int64, device=dev) * L
torch.cuda.synchronize()
for t in (0.99, 0.97, 0.95, 0.90):
    al = device.Aligner(index, device=0, threshold=t, max_batch_reads=R, max_read_len=256, max_batch_bases=R*L+64)
    al.set_profiling(True)
    for _ in range(2):
        al.submit_device(d_seq.data_ptr(), d_off.data_ptr(), R, first_read_id=0, max_len=L); c = al.wait()
    ms = al.stage_ms()
    print(t, {k: round(v, 2) for k, v in ms.items()}, "seeds/read", round(c["seeds"]/R, 2), "alns", c["alignments"], "Mreads/s", round(R/ms["total"]/1e3, 1))
    al.close()
