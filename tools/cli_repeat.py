#!/usr/bin/env python3
"""Does `groot-hip align` give the same counts run after run?  One synthetic FASTQ (bench.py's cli_e2e input), the CLI N times per binary directory:
    python tools/cli_repeat.py N [reads] DIR [DIR...]     (DIR holds groot-hip + libgroot_host.so + libgroot_hip.so; default build/)"""
import json
import os
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402

import bench  # noqa: E402
from groot_amd import synth  # noqa: E402

n_runs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
R = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
dirs = sys.argv[3:] or [os.path.join(REPO, "build")]
dev = torch.device("cuda", 0)
index, _ = bench.load_index()
cat, off, lens = synth.reference_sequences(index)
cat_t, off_t, lens_t = (torch.from_numpy(x).to(dev) for x in (cat, off, lens))
chunks = []
for c0 in range(0, R, 1_000_000):
    n = min(1_000_000, R - c0)
    p, _, _ = synth.reads_torch(cat_t, off_t, lens_t, n, bench.READ_LEN, first=c0)
    chunks.append(p[: n * bench.READ_LEN].cpu())
seq = torch.cat(chunks).numpy()
with tempfile.TemporaryDirectory() as td:
    idx_dir = os.path.join(td, "index")
    os.makedirs(idx_dir)
    index.save(os.path.join(idx_dir, "groot.gidx"))
    fq = os.path.join(td, "reads.fq")
    bench.write_fastq(fq, seq, R)
    for d in dirs:
        for i in range(n_runs):
            stats = os.path.join(td, "stats.json")
            cmd = [os.path.join(d, "groot-hip"), "align", "-i", idx_dir, "-f", fq, "-g", os.path.join(td, "graphs"), "--bam", os.path.join(td, "out.bam"),
                   "--log", os.path.join(td, "groot.log"), "-p", str(bench.usable_cpus()), "--bamLevel", "-2", "--stats", stats, "--batch", "262144"] + os.environ.get("GROOT_CLI_EXTRA", "").split()
            try:
                p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=60)
            except subprocess.TimeoutExpired:
                print(d, i, "HUNG (killed after 60 s)", flush=True)
                continue
            if p.returncode:
                print(d, i, "FAILED", (p.stderr or p.stdout)[-300:])
                continue
            st = json.load(open(stats))
            print(d, i, "mapped", st["mapped"], "alignments", st["alignments"], "bam_bytes", st["bam_bytes"], "full_sketch_reads", st["full_sketch_reads"], "total_s %.3f" % st["total_s"], flush=True)
