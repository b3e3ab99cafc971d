#!/usr/bin/env python3
"""A host-fed stream of many batches (packed wire format, several in flight, ctx opened in the background -- what `groot-hip align` does) against the same
reads as ONE device-resident batch: per-read record counts, batch by batch; names the reads that differ.
    [GROOT_HIP_LIB=...] python tools/stream_check.py [reads] [batch] [depth]"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402
from groot_amd import device, host, synth  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 6_000_000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 262_144
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 3
L = 100
index, _ = bench.load_index()
cat, off, lens = synth.reference_sequences(index)
seq, so, _ = synth.reads_np(cat, off, lens, R, L)
ref = device.Aligner(index, max_batch_reads=R, max_read_len=128, max_batch_bases=int(so[-1]) + 64, memo_budget_mb=device.MEMO_OFF)
ref.submit(seq, so)
want = ref.wait()
t, m = ref.travs()
ref_cnt = np.bincount(t["read_id"], minlength=R)
ref_aln = np.zeros(R, dtype=np.int64)
np.add.at(ref_aln, t["read_id"], np.unpackbits(np.ascontiguousarray(m).view(np.uint8), axis=1).sum(axis=1))
ref.close()
print("one batch: mapped", want["mapped"], "alignments", want["alignments"], flush=True)
batches = []
for b0 in range(0, R, B):
    n = min(B, R - b0)
    pk, ep, eb = host.pack_reads(seq[b0 * L:(b0 + n) * L])
    batches.append((b0, n, pk, ep, eb))
al = device.Aligner(index, max_batch_reads=B, max_read_len=256, pipeline_depth=depth, memo_budget_mb=device.MEMO_OFF, background=True)
bad_total, aln_total, pending = 0, 0, []


def take():
    global bad_total, aln_total
    b0, n = pending.pop(0)
    r = al.collect(copy=True)
    cnt = np.bincount(r["travs"]["read_id"], minlength=n)[:n]
    aln = np.zeros(n, dtype=np.int64)
    if r["n_travs"]:
        np.add.at(aln, r["travs"]["read_id"], np.unpackbits(np.ascontiguousarray(r["masks"]).view(np.uint8), axis=1).sum(axis=1))
    aln_total += int(aln.sum())
    d = np.nonzero((cnt != ref_cnt[b0:b0 + n]) | (aln != ref_aln[b0:b0 + n]))[0]
    if len(d):
        bad_total += len(d)
        print("batch at read %d: %d reads differ, full_sketch_reads %d" % (b0, len(d), r["counts"]["full_sketch_reads"]))
        for i in d[:12]:
            rd = seq[(b0 + i) * L:(b0 + i + 1) * L]
            print("   read %d (in batch %d: workgroup %d, wavefront %d, lane %d) records %d want %d, alignments %d want %d, bytes other than ACGT %d" % (
                b0 + i, i, i // 256, (i % 256) // 64, i % 64, cnt[i], ref_cnt[b0 + i], aln[i], ref_aln[b0 + i], int(np.isin(rd, np.frombuffer(b"ACGT", np.uint8), invert=True).sum())))
    al.release(r["ticket"])


for b0, n, pk, ep, eb in batches:
    if len(pending) == depth:
        take()
    al.submit_packed16(pk, np.full(n, L, dtype=np.uint16), ep, eb)
    pending.append((b0, n))
while pending:
    take()
al.close()
print("stream: %d batches, alignments %d (want %d), reads that differ: %d" % (len(batches), aln_total, want["alignments"], bad_total))
