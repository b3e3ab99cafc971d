#!/usr/bin/env python
"""PCIe-inclusive rate of the hot path: the boundary is handed HOST buffers (groot_hip_submit), as the reference's
mapReads would hand them over.  Reported beside bench.py's HBM-resident `value`, never instead of it.

  one ctx : submit (H2D on the ctx stream) then wait, back to back -- copy and kernels serialise
  two ctxs: two host threads, one ctx each on the same GPU, half-size batches -- one ctx copies while the other computes
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=10_000_000)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--pageable", action="store_true", help="host buffers in ordinary (not page-locked) memory")
    args = ap.parse_args()
    import torch

    import bench
    from groot_amd import device, synth

    dev = torch.device("cuda", 0)
    index = bench.load_index()
    cat, off, lens = synth.reference_sequences(index)
    cat_t, off_t, lens_t = (torch.from_numpy(x).to(dev) for x in (cat, off, lens))
    R, L = args.reads, bench.READ_LEN
    parts = []
    for c0 in range(0, R, 1_000_000):
        n = min(1_000_000, R - c0)
        p, _, _ = synth.reads_torch(cat_t, off_t, lens_t, n, L, first=c0)
        parts.append(p[: n * L].cpu())
    h_seq = torch.cat(parts)
    h_off = torch.arange(0, R + 1, dtype=torch.int64) * L
    if not args.pageable:
        h_seq, h_off = h_seq.pin_memory(), h_off.pin_memory()
    seq, offs = h_seq.numpy(), h_off.numpy().view(np.uint64)
    out = {"reads": R, "read_len": L, "host_memory": "pageable" if args.pageable else "pinned"}

    from groot_amd import host
    t0 = time.perf_counter()
    packed, exc_pos, exc_byte = host.pack_reads(seq)
    out["pack_ms_all_cores"] = (time.perf_counter() - t0) * 1e3
    packed_t = torch.from_numpy(packed)
    if not args.pageable:
        packed_t = packed_t.pin_memory()
    packed = packed_t.numpy()

    def run(n_ctx, use_packed=False):
        per = R // n_ctx
        als = [device.Aligner(index, device=0, max_batch_reads=per, max_read_len=256, max_batch_bases=per * L + 64) for _ in range(n_ctx)]
        totals = [None] * n_ctx

        def work(i, steps):
            lo = i * per
            s, o = seq[lo * L:(lo + per) * L], offs[lo:lo + per + 1] - np.uint64(lo * L)
            o = np.ascontiguousarray(o)
            pk = packed[lo * L // 4:(lo + per) * L // 4]
            sel = (exc_pos >= lo * L) & (exc_pos < (lo + per) * L)
            ep, eb = exc_pos[sel] - np.uint64(lo * L), exc_byte[sel]
            for _ in range(steps):
                if use_packed:
                    als[i].submit_packed(pk, o, ep, eb, first_read_id=lo)
                else:
                    als[i].submit(s, o, first_read_id=lo)
                totals[i] = als[i].wait()

        def timed(steps):
            th = [threading.Thread(target=work, args=(i, steps)) for i in range(n_ctx)]
            t0 = time.perf_counter()
            for t in th:
                t.start()
            for t in th:
                t.join()
            torch.cuda.synchronize()
            return time.perf_counter() - t0

        timed(1)
        dt = timed(args.steps)
        for a in als:
            a.close()
        return {"Mreads_s": per * n_ctx * args.steps / dt / 1e6, "ms_per_10M": dt / args.steps * 1e3 * (1e7 / (per * n_ctx)),
                "alignments": sum(t["alignments"] for t in totals)}

    out["one_ctx"] = run(1)
    out["two_ctx"] = run(2)
    out["non_acgt_bytes"] = int(len(exc_pos))
    out["one_ctx_packed"] = run(1, True)
    out["two_ctx_packed"] = run(2, True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
