#!/bin/bash
# end-to-end `groot-hip align` on a synthetic FASTQ (run on the GPU box): FASTQ in -> BAM + GFA out, wall time
set -u
cd "$GRAFT_REPO_ROOT"
N=${1:-1000000}
W=/tmp/e2e; rm -rf $W; mkdir -p $W
python - <<PY
import sys, tarfile, os, numpy as np
sys.path.insert(0, ".")
from groot_amd import host, synth
with tarfile.open("tests/golden/data/arg-annot.90.tar.gz") as tf:
    ms=[m for m in tf.getmembers() if os.path.basename(m.name).startswith("cluster") and m.name.endswith(".msa")]
    tf.extractall("$W", members=ms)
idx=host.Index.from_msa_dir("$W/arg-annot.90")
os.makedirs("$W/idx"); idx.save("$W/idx/groot.gidx")
cat,off,lens=synth.reference_sequences(idx)
seq,so,_=synth.reads_np(cat,off,lens,$N,100)
q=b"I"*100
with open("$W/reads.fq","wb") as f:
    a=seq.reshape(-1,100)
    for i in range($N):
        f.write(b"@syn_%d\n"%i); f.write(a[i].tobytes()); f.write(b"\n+\n"); f.write(q); f.write(b"\n")
print("fastq written", os.path.getsize("$W/reads.fq")/1e6, "MB")
PY
time ./build/groot-hip align -i $W/idx -f $W/reads.fq --bam $W/out.bam -g $W/graphs --log $W/align.log --batch 1000000 -p ${P:-8}
ls -la $W/out.bam; tail -4 $W/align.log
