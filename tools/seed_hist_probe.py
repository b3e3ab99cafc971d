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
    from groot_amd import _ffi
    wg = _ffi.view_arrays(index.view)["win_graph"]
    g = wg[sd["window_id"]]
    heavy = np.flatnonzero(per > 8)
    order = np.argsort(sd["read_id"], kind="stable")
    rid, gg = sd["read_id"][order], g[order]
    starts = np.searchsorted(rid, heavy); ends = np.searchsorted(rid, heavy, side="right")
    ng, mx = [], []
    for a, b in zip(starts[:20000], ends[:20000]):
        u, c = np.unique(gg[a:b], return_counts=True)
        ng.append(len(u)); mx.append(c.max())
    ng, mx = np.array(ng), np.array(mx)
    print(t, "reads > 8 seeds:", len(heavy), "graphs per such read: median", np.median(ng), "max", ng.max(), "; largest single-graph group: median", np.median(mx), "p99", np.percentile(mx, 99), "max", mx.max(),
          "; reads whose largest group > 8:", int((mx > 8).sum()), "> 32:", int((mx > 32).sum()))
    print(t, "seeds/read max", per.max(), "hist>4:", int((per > 4).sum()), ">16:", int((per > 16).sum()), ">64:", int((per > 64).sum()), "top", np.sort(per)[-10:])
    al.close()

# how many of the seed windows does graphMinion actually try (IncrementSubPath calls, graphminion.go:60-67) -- it passes over a
# graph's remaining windows after the first alignment there (:96-98)
for t in (0.99,):
    al = device.Aligner(index, threshold=t, max_batch_reads=R, max_read_len=256, max_batch_bases=int(so[-1]) + 64)
    al.submit(seq, so); c = al.wait()
    calls = int(al.attempts().astype(np.int64).sum())
    print(t, "seeds", c["seeds"], "IncrementSubPath calls", calls, "mapped", c["mapped"], "travs", c.get("travs"))
    al.close()

# reads with many seed windows and no alignment at all: every one of their windows is tried and fails
al = device.Aligner(index, threshold=0.99, max_batch_reads=R, max_read_len=256, max_batch_bases=int(so[-1]) + 64)
al.submit(seq, so); c = al.wait()
sd = al.seeds(); per = np.bincount(sd["read_id"], minlength=R)
tr = al.travs()[0]
has = np.zeros(R, bool); has[tr["read_id"]] = True
bad = np.flatnonzero((per > 0) & ~has)
print("reads with seeds and no traversal:", len(bad), "their seed counts: total", int(per[bad].sum()), "top", np.sort(per[bad])[-12:])
lens = np.diff(so.astype(np.int64))
for r in bad[np.argsort(per[bad])[-3:]]:
    print("read", r, "len", lens[r], "seeds", per[r], bytes(seq[int(so[r]):int(so[r + 1])]).decode(errors="replace"))
al.close()
