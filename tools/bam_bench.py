"""The BAM writer alone, on the host: traversals of synthetic reads (start node of the read's own path, path set = every path through
that node) -> groot_bam_write_travs -> /dev/null.   python tools/bam_bench.py [reads] [threads] [level]   (GROOT_BAM_STATS=1 for the split)"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from groot_amd import _ffi, device, host, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 8
level = int(sys.argv[3]) if len(sys.argv) > 3 else -2
index, _ = bench.load_index()
v = index.view
a = _ffi.view_arrays(v)
cat, off, lens = synth.reference_sequences(index)
seq, so, truth = synth.reads_np(cat, off, lens, n, 100)
# per path: (position, node) of its nodes, ascending
npo, npp, nps = a["node_np_off"], a["np_path"].astype(np.int64), a["np_pos"].astype(np.int64)
node_of_pair = np.repeat(np.arange(v.n_nodes), np.diff(npo))
g_of_node = np.repeat(np.arange(v.n_graphs), np.diff(a["graph_node_off"]))
gpath = a["graph_path_off"][g_of_node[node_of_pair]].astype(np.int64) + npp      # global path id of every pair
order = np.lexsort((nps, gpath))
gp_s, pos_s, node_s = gpath[order], nps[order], node_of_pair[order]
key = gp_s * (1 << 32) + pos_s
p, st = truth["seq"].astype(np.int64), truth["start"].astype(np.int64)
j = np.searchsorted(key, p * (1 << 32) + st, side="right") - 1
node = node_s[j]
offset = st - pos_s[j]
pw = v.path_words
masks = a["node_mask"].reshape(v.n_nodes, pw)[node].copy()
max_paths = int(sys.argv[4]) if len(sys.argv) > 4 else 17          # (arg-annot.90 reads: 17 records each on average)
if max_paths:
    for w in range(pw):                                             # keep at most max_paths set bits per traversal (lowest first), word by word
        m = masks[:, w].copy()
        out = np.zeros_like(m)
        left = np.full(len(m), max_paths, dtype=np.int64) if w == 0 else left
        for _ in range(64):
            low = m & (~m + np.uint64(1))
            take = (low != 0) & (left > 0)
            out |= np.where(take, low, np.uint64(0))
            left = left - take.astype(np.int64)
            m = m & ~low
            if not take.any():
                break
        masks[:, w] = out
tr = np.zeros(n, dtype=device.TRAV_DTYPE)
tr["read_id"] = np.arange(n); tr["graph_id"] = g_of_node[node]; tr["node"] = node; tr["offset"] = offset
tr["flags"] = 8 | (truth["strand"] & 1).astype(np.uint8)
names = [b"r%d" % i for i in range(n)]
noff = np.zeros(n + 1, dtype=np.uint64)
noff[1:] = np.cumsum([len(x) for x in names])
names = np.frombuffer(b"".join(names), dtype=np.uint8).copy()
qual = np.full(len(seq), ord("I"), dtype=np.uint8)


class RB(C.Structure):
    _fields_ = [("seq", C.c_void_p), ("qual", C.c_void_p), ("seq_off", C.c_void_p), ("names", C.c_void_p), ("name_off", C.c_void_p),
                ("n_reads", C.c_uint32), ("first_read_id", C.c_uint32)]


rb = RB(seq.ctypes.data, qual.ctypes.data, so.ctypes.data, names.ctypes.data, noff.ctypes.data, n, 0)
H = host.lib()
for rep in range(3):
    h = C.c_void_p()
    host._check(H.groot_bam_open(b"/dev/null", C.byref(v), b"2020-01-01T00:00:00Z", C.byref(h)))
    host._check(H.groot_bam_set_threads(h, C.c_uint32(threads)))
    host._check(H.groot_bam_set_level(h, C.c_int(level)))
    nrec = C.c_uint64()
    t0 = time.perf_counter()
    host._check(H.groot_bam_write_travs(h, C.byref(v), C.byref(rb), tr.ctypes.data_as(C.c_void_p), _ffi.as_ptr(masks, C.c_uint64), C.c_uint64(n), C.byref(nrec)))
    dt = time.perf_counter() - t0
    host._check(H.groot_bam_close(h))
    print("reads %d records %d (%.1f per read) threads %d level %d: %.3f s = %.1f M records/s, %.1f ns per record and thread" % (
        n, nrec.value, nrec.value / n, threads, level, dt, nrec.value / dt / 1e6, dt * threads / nrec.value * 1e9))
