"""Generates tests/golden/oracle_golden.json: digests of the oracle's outputs on the reference's own
fixtures (tests/golden/data).  Run from the repo root:  python tests/golden/make_golden.py
The reference cannot be executed here (Go), so these vectors are oracle output frozen after the
oracle was pinned on the reference's assertions (tests/test_oracle_pins.py)."""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))


def _d(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def run_case(index, reads, threshold=0.99):
    from oracle import oracle_py as O

    cat, off = O.pack_reads([r[1] for r in reads])
    run = O.Run(index, threshold)
    run.batch(cat, off)
    kf, kt = run.weights(order=1)
    att = run.attempts()
    return {
        "counts": {k: int(v) for k, v in run.counts().items()},
        "sketches": _d(run.sketches()),
        "seeds": _d(run.seeds()),
        "alns": _d(run.alns()),
        "attempts_nonzero": int(np.count_nonzero(att)),
        "attempts_sum": int(att.sum()),
        "kmer_freq": _d(kf),
        "kmer_total": _d(kt),
    }


def compute(argannot_index, perfect_reads, variable_reads, genes_index, oxa_reads):
    return {
        "perfect_reads_small@arg-annot.90(k31,s21,w100),t0.99": run_case(argannot_index, perfect_reads),
        "perfect_reads_small_variable_rl@arg-annot.90,t0.99": run_case(argannot_index, variable_reads),
        "perfect_reads_small_variable_rl@arg-annot.90,t0.90": run_case(argannot_index, variable_reads, 0.90),
        "OXA90-OXA106-with-errors@test-genes(k51,s30,w100),t0.99": run_case(genes_index, oxa_reads),
    }


if __name__ == "__main__":
    import tarfile
    import tempfile

    from conftest import DATA, read_fastq
    from groot_amd import host

    with tempfile.TemporaryDirectory() as td:
        with tarfile.open(os.path.join(DATA, "arg-annot.90.tar.gz")) as tf:
            members = [m for m in tf.getmembers() if os.path.basename(m.name).startswith("cluster") and m.name.endswith(".msa")]
            tf.extractall(td, members=members)
        arg = host.Index.from_msa_dir(os.path.join(td, "arg-annot.90"))
    genes = host.Index.from_msa_files([os.path.join(DATA, "test-genes.msa")], host.index_params(k=51, s=30, w=100))
    out = compute(arg, read_fastq(os.path.join(DATA, "full-argannot-perfect-reads-small.fq.gz")),
                  read_fastq(os.path.join(DATA, "full-argannot-perfect-reads-small-variable-rl.fq.gz")), genes,
                  read_fastq(os.path.join(DATA, "test-reads-OXA90-OXA106-100bp-with-errors.fastq.gz")))
    with open(os.path.join(HERE, "oracle_golden.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(json.dumps({k: v["counts"] for k, v in out.items()}, indent=1))
