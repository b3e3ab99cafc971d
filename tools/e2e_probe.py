"""cli_e2e leg of bench.py alone: python tools/e2e_probe.py [reads] [bam level]   (GPU box)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from groot_amd import synth

R = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
level = int(sys.argv[2]) if len(sys.argv) > 2 else -2
dev = torch.device("cuda", 0)
index, _ = bench.load_index()
cat, off, lens = synth.reference_sequences(index)
cat_t, off_t, lens_t = (torch.from_numpy(x).to(dev) for x in (cat, off, lens))
chunks = []
for c0 in range(0, R, 1_000_000):
    n = min(1_000_000, R - c0)
    p, _, _ = synth.reads_torch(cat_t, off_t, lens_t, n, 100, first=c0)
    chunks.append(p[: n * 100])
d_seq = torch.cat(chunks)
out = bench.cli_e2e(index, d_seq, R, level)
print(json.dumps({k: v for k, v in out.items() if k != "what"}))
