"""Seed windows per read in the mixed-length workload (resfinder.90, reads of 75..150 bases): the tail decides the align stage.
    python tools/seed_hist_probe.py      (GPU box)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import bench
from groot_amd import device, host, synth
index, _ = bench.load_index("resfinder.90")
cat, off, lens = synth.reference_sequences(index)
R = 500000
seq, so, _ = synth.reads_np(cat, off, lens, R, 150, min_len=75)
for t in (0.99, 0.90):
    al = device.Aligner(index, threshold=t, max_batch_reads=R, max_read_len=256, max_batch_bases=int(so[-1]) + 64)
    al.submit(seq, so); c = al.wait()
    sd = al.seeds()
    per = np.bincount(sd["read_id"], minlength=R)
    tr = al.travs() if hasattr(al, "travs") else None
    print(t, "seeds/read max", per.max(), "hist>4:", int((per > 4).sum()), ">16:", int((per > 16).sum()), ">64:", int((per > 64).sum()), "top", np.sort(per)[-10:])
    al.close()
