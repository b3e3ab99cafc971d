#!/usr/bin/env python3
"""Per-read record / alignment counts of ONE large resident batch, saved for comparison between builds and switches (a random error of one read in a
million is invisible to comparisons with the oracle on 10^5 reads):
    [GROOT_HIP_LIB=... GROOT_NO_SIG=1 ...] python tools/cross_check.py mixed99|mixed90|sub1|c2 OUT.npz      then     python tools/cross_check.py --diff A.npz B.npz ..."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

if sys.argv[1] == "--diff":
    base = np.load(sys.argv[2])
    for f in sys.argv[3:]:
        x = np.load(f)
        d = np.nonzero((x["cnt"] != base["cnt"]) | (x["aln"] != base["aln"]))[0]
        print("%s vs %s: %d of %d reads differ; records %d / %d, alignments %d / %d" % (f, sys.argv[2], len(d), len(base["cnt"]), x["cnt"].sum(), base["cnt"].sum(), x["aln"].sum(), base["aln"].sum()), d[:10])
    sys.exit(0)

import torch  # noqa: E402

import bench  # noqa: E402
from groot_amd import device, synth  # noqa: E402

wl, out = sys.argv[1], sys.argv[2]
dev = torch.device("cuda", 0)
mixed = wl.startswith("mixed")
R = 8_000_000 if mixed else 10_000_000
index, _ = bench.load_index("resfinder.90" if mixed else "arg-annot.90")
cat, off, lens = synth.reference_sequences(index)
cat_t, off_t, lens_t = (torch.from_numpy(x).to(dev) for x in (cat, off, lens))
threshold = 0.99
if mixed:
    d_seq, d_off, _ = synth.reads_torch_mixed(cat_t, off_t, lens_t, R, 150, 75)
    max_len, total = 150, int(d_off[-1].item())
    threshold = int(wl[5:]) / 100.0
else:
    chunks = []
    for c0 in range(0, R, 1_000_000):
        n = min(1_000_000, R - c0)
        p, _, _ = synth.reads_torch(cat_t, off_t, lens_t, n, bench.READ_LEN, first=c0)
        chunks.append(p[: n * bench.READ_LEN])
    d_seq = torch.zeros(R * bench.READ_LEN + 64, dtype=torch.uint8, device=dev)
    d_seq[: R * bench.READ_LEN] = torch.cat(chunks)
    d_off = torch.arange(0, R + 1, dtype=torch.int64, device=dev) * bench.READ_LEN
    max_len, total = bench.READ_LEN, R * bench.READ_LEN
    if wl == "sub1":
        g = torch.Generator(device=dev)
        g.manual_seed(0x67726F6F74)
        d_seq = bench.substituted(d_seq, R, 0.01, g)
al = device.Aligner(index, device=0, threshold=threshold, max_batch_reads=R, max_read_len=256, max_batch_bases=total + 64, memo_budget_mb=device.MEMO_OFF)
cnt = aln = None
for rep in range(2):                                      # twice: the second run must repeat the first
    al.submit_device(d_seq.data_ptr(), d_off.data_ptr(), R, first_read_id=0, max_len=max_len, mixed=mixed)
    c = al.wait()
    t, m = al.travs()
    cn = np.bincount(t["read_id"], minlength=R)
    an = np.zeros(R, dtype=np.int64)
    np.add.at(an, t["read_id"], np.unpackbits(np.ascontiguousarray(m).view(np.uint8), axis=1).sum(axis=1))
    if cnt is not None:
        print("second run: %d reads differ from the first" % int(((cn != cnt) | (an != aln)).sum()))
    cnt, aln = cn, an
print(wl, "mapped", c["mapped"], "alignments", c["alignments"], "records", int(cnt.sum()), "full_sketch_reads", c["full_sketch_reads"], "walked", c["walked_reads"], flush=True)
np.savez(out, cnt=cnt.astype(np.int32), aln=aln.astype(np.int32))
al.close()
