#!/usr/bin/env python3
"""which reads differ between the first runs and the settled ones, and how"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from groot_amd import device, synth
dev = torch.device("cuda", 0)
R = 10_000_000
index, _ = bench.load_index("arg-annot.90")
cat, off, lens = synth.reference_sequences(index)
cat_t, off_t, lens_t = (torch.from_numpy(x).to(dev) for x in (cat, off, lens))
chunks = []
for c0 in range(0, R, 1_000_000):
    p, _, _ = synth.reads_torch(cat_t, off_t, lens_t, 1_000_000, 100, first=c0)
    chunks.append(p[: 1_000_000 * 100])
d_seq = torch.zeros(R * 100 + 64, dtype=torch.uint8, device=dev)
d_seq[: R * 100] = torch.cat(chunks)
del chunks
d_off = torch.arange(0, R + 1, dtype=torch.int64, device=dev) * 100
al = device.Aligner(index, device=0, max_batch_reads=R, max_read_len=256, max_batch_bases=R * 100 + 64, memo_budget_mb=device.MEMO_OFF)
runs = []
for rep in range(5):
    al.submit_device(d_seq.data_ptr(), d_off.data_ptr(), R, first_read_id=0, max_len=100)
    c = al.wait()
    t, m = al.travs()
    cnt = np.bincount(t["read_id"], minlength=R)
    sd = al.seeds()
    sc = np.bincount(sd["read_id"], minlength=R)
    runs.append((cnt, sc, c))
    print("run", rep, c, flush=True)
ref_cnt, ref_sc, _ = runs[-1]
seq_h = d_seq[: R * 100].view(R, 100)
for rep in range(4):
    cnt, sc, _ = runs[rep]
    d = np.flatnonzero((cnt != ref_cnt) | (sc != ref_sc))
    print("run %d: %d reads differ" % (rep, len(d)))
    for r in d[:40]:
        row = seq_h[int(r)].cpu().numpy()
        print("   read %8d (block %6d, lane-in-block %3d): records %d (settled %d), seeds %d (settled %d), N in read: %s" % (r, r // 256, r % 256, cnt[r], ref_cnt[r], sc[r], ref_sc[r], bool((row == ord('N')).any())))
al.close()
