"""bench.py's host_fed leg alone (pinned host -> GPU -> pinned host, several batches in flight), for tuning the copy-out:
    python tools/host_fed_probe.py [steps] [depth]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from groot_amd import synth

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 3
R = int(os.environ.get("READS", 10_000_000))
index, _ = bench.load_index()
dev = torch.device("cuda", 0)
cat, off, lens = synth.reference_sequences(index)
cat_t, off_t, lens_t = (torch.from_numpy(x).to(dev) for x in (cat, off, lens))
parts = []
for c0 in range(0, R, 1_000_000):
    n = min(1_000_000, R - c0)
    p, _, _ = synth.reads_torch(cat_t, off_t, lens_t, n, bench.READ_LEN, first=c0)
    parts.append(p[: n * bench.READ_LEN])
d_seq = torch.cat(parts)
hf, _ = bench.host_fed(index, d_seq, R, steps, depth)
print(json.dumps({"depth": depth, "value": hf["value"], "ms_per_batch": hf["ms_per_batch"],
                  "stage": {k: round(v, 2) for k, v in hf["stage_ms_per_batch"].items()},
                  "caller": {k: round(v, 2) for k, v in hf["caller_ms_per_batch"].items()}}))
