#!/usr/bin/env python
"""Experiment: do the sketch+seed kernel (VALU bound) and the align kernel (latency bound) of two ctxs on ONE GPU overlap
when their batches are staggered?  Two host threads, one ctx each, device-resident reads."""
import argparse, json, os, sys, threading, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=10_000_000)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--stagger-ms", type=float, default=2.0)
    args = ap.parse_args()
    import torch
    import bench
    from groot_amd import device, synth
    dev = torch.device("cuda", 0)
    index, _ = bench.load_index()
    cat, off, lens = synth.reference_sequences(index)
    cat_t, off_t, lens_t = (torch.from_numpy(x).to(dev) for x in (cat, off, lens))
    R, L = args.reads, bench.READ_LEN
    parts = []
    for c0 in range(0, R, 1_000_000):
        n = min(1_000_000, R - c0)
        p, _, _ = synth.reads_torch(cat_t, off_t, lens_t, n, L, first=c0)
        parts.append(p[: n * L])
    d_seq = torch.zeros(R * L + 64, dtype=torch.uint8, device=dev)
    d_seq[: R * L] = torch.cat(parts)
    d_off = torch.arange(0, R + 1, dtype=torch.int64, device=dev) * L
    torch.cuda.synchronize()
    out = {}
    for n_ctx in (1, 2):
        per = R // n_ctx
        als = [device.Aligner(index, device=0, max_batch_reads=per, max_read_len=256, max_batch_bases=R * L + 64, results_on_device=True) for _ in range(n_ctx)]
        res = [None] * n_ctx
        def work(i, steps, delay):
            time.sleep(delay)
            for _ in range(steps):
                als[i].submit_device(d_seq.data_ptr(), d_off.data_ptr() + 8 * i * per, per, first_read_id=i * per, max_len=L)
                res[i] = als[i].wait()
        def timed(steps):
            th = [threading.Thread(target=work, args=(i, steps, i * args.stagger_ms * 1e-3)) for i in range(n_ctx)]
            t0 = time.perf_counter()
            for t in th: t.start()
            for t in th: t.join()
            torch.cuda.synchronize()
            return time.perf_counter() - t0
        timed(1)
        dt = timed(args.steps)
        out[f"{n_ctx}_ctx"] = {"Mreads_s": per * n_ctx * args.steps / dt / 1e6, "alignments": sum(r["alignments"] for r in res)}
        for a in als: a.close()
    print(json.dumps(out))

if __name__ == "__main__":
    main()
