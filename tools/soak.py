#!/usr/bin/env python
"""Soak test (GPU box): many seeds of mutated / mixed-length / both-strand reads on the full arg-annot.90 index, device vs
oracle on everything the parity tests compare.  python tools/soak.py [n_seeds] [reads_per_seed]"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))


def main():
    n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 60000
    import bench
    from groot_amd import synth
    from oracle import oracle_py as O
    import test_gpu_parity as T
    from test_gpu_parity import assert_same, run_both

    index, _ = bench.load_index()
    cat, o, lens = synth.reference_sequences(index)
    comp = bytes.maketrans(b"ACGTN", b"TGCAN")
    for seed in range(n_seeds):
        rng = np.random.default_rng(1000 + seed)
        reads = []
        for i in range(n_reads):
            s = int(rng.integers(0, len(lens)))
            L = min(int(rng.integers(31, 200)) if i % 3 == 0 else 100, int(lens[s]))
            st = int(rng.integers(0, lens[s] - L + 1))
            r = bytearray(cat[int(o[s]) + st:int(o[s]) + st + L].tobytes())
            kind = int(rng.integers(0, 10))
            if kind == 1:
                r[0] = ord("ACGT"[int(rng.integers(0, 4))])
            elif kind == 2:
                r[-1] = ord("ACGT"[int(rng.integers(0, 4))])
            elif kind == 3:
                r[int(rng.integers(0, L))] = ord("ACGTN"[int(rng.integers(0, 5))])
            elif kind == 4:
                r = bytearray(rng.integers(0, 4, L).astype(np.uint8).tobytes().translate(bytes.maketrans(bytes(range(4)), b"ACGT")))
            r = bytes(r)
            reads.append(r.translate(comp)[::-1] if rng.integers(0, 2) else r)
        seq, off = O.pack_reads(reads)
        for t in (0.99, 0.95):
            for keep in (True, False):      # full-width sketch kernel alone / signature kernel in front of it
                T.KEEP_SKETCHES = keep
                al, counts, run = run_both(index, seq, off, threshold=t)
                assert_same(al, counts, run, index)
                al.close()
                print(f"seed {seed} t={t} {'full-width' if keep else 'signature'}: {counts['mapped']} mapped, {counts['alignments']} alignments, "
                      f"{counts['full_sketch_reads']} through the full-width kernel, identical", flush=True)


if __name__ == "__main__":
    main()
