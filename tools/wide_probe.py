"""align_kernel<11, *> (graphs with 193..704 paths) on a synthetic 300-allele graph: ms per batch of reads.
    python tools/wide_probe.py [reads]          (GPU box; GROOT_WIDE_WAVES=2|4 picks the occupancy the kernel is launched for)"""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from groot_amd import device, host, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
rng = np.random.default_rng(1)
base = rng.integers(0, 4, 1200)
seqs = []
for i in range(300):
    s = base.copy()
    pos = rng.choice(1200, 30, replace=False)
    s[pos] = (s[pos] + rng.integers(1, 4, 30)) % 4
    seqs.append(bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[s]))
with tempfile.TemporaryDirectory() as td:
    p = os.path.join(td, "wide.msa")
    with open(p, "w") as f:
        for i, s in enumerate(seqs):
            f.write(">allele%d\n%s\n" % (i, s.decode()))
    index = host.Index.from_msa_files([p])
assert index.view.path_words > 3
cat, o, lens = synth.reference_sequences(index)
seq, off, _ = synth.reads_np(cat, o, lens, n, 100)
al = device.Aligner(index, max_batch_reads=n, results_on_device=True)
al.set_profiling(True)
for _ in range(2):
    al.submit(seq, off); c = al.wait()
t = []
for _ in range(5):
    al.submit(seq, off); c = al.wait()
    t.append(al.stage_ms()["align"])
print({"reads": n, "path_words": int(index.view.path_words), "align_ms": round(float(np.mean(t)), 3), "mapped": c["mapped"], "alignments": c["alignments"]})
