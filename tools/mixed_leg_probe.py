"""The kernels of bench.py's mixed leg alone (resfinder.90, reads of 75..150 bases resident in HBM), for rocprofv3 --kernel-trace.
    python tools/mixed_leg_probe.py [threshold] [max_read_len] [steps]      (GPU box)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from groot_amd import device, host, synth

t = float(sys.argv[1]) if len(sys.argv) > 1 else 0.99
mrl = int(sys.argv[2]) if len(sys.argv) > 2 else 256
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
R = int(os.environ.get("READS", 2_000_000))
index, _ = bench.load_index("resfinder.90")
dev = torch.device("cuda", 0)
cat, off, lens = synth.reference_sequences(index)
cat_t, off_t, lens_t = (torch.from_numpy(x).to(dev) for x in (cat, off, lens))
d_seq, d_off, _ = synth.reads_torch_mixed(cat_t, off_t, lens_t, R, 150, 75)
total = int(d_off[-1].item())
al = device.Aligner(index, device=0, threshold=t, max_batch_reads=R, max_read_len=mrl, max_batch_bases=total + 64, results_on_device=True, pipeline_depth=2)
al.set_profiling(True)
v, ms, c = bench.resident_rate(al, d_seq.data_ptr(), d_off.data_ptr(), R, 150, steps, 2, mixed=True)
print({"t": t, "max_read_len": mrl, "Mreads_s": round(v, 1), "stage_ms": {k: round(x, 3) for k, x in ms.items()}, "mapped": c["mapped"],
       "full_sketch_reads": c["full_sketch_reads"], "walked_reads": c["walked_reads"]})
al.close()
