"""N>1 path on CPU: world_size 2 over gloo.  Each rank handles its shard of the reads (here through
the oracle, standing in for its GPU) and the call-count table + counters are all-reduced exactly as
bench.py does over RCCL; the reduced result must equal the single-process run."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import REPO


COUNT_KEYS = ("received", "mapped", "multimapped", "alignments", "seeds")


def shard_range(n_reads, rank, world):
    """contiguous, balanced [lo, hi) of reads for `rank` (what bench.py and the CLI do with batches)"""
    base, rem = divmod(n_reads, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def reduce_counts(dist, counts):
    """sum the boss counters (boss.go:24-27) over ranks"""
    t = torch.tensor([int(counts[k]) for k in COUNT_KEYS], dtype=torch.int64)
    dist.all_reduce(t)
    return {k: int(v) for k, v in zip(COUNT_KEYS, t.tolist())}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, msa_files, n_reads, out_dir):
    sys.path.insert(0, REPO)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from groot_amd import host, synth
    from oracle import oracle_py as O

    index = host.Index.from_msa_files(msa_files)          # replicated index
    cat, off, lens = synth.reference_sequences(index)
    lo, hi = shard_range(n_reads, rank, world)
    seq, so, _ = synth.reads_np(cat, off, lens, hi - lo, 100, first=lo)
    run = O.Run(index)
    run.batch(seq, so, first_read_id=lo)
    att = run.attempts()
    n_q = 256 - index.view.kmer_size + 2                  # table shape of a ctx opened with max_read_len=256
    full = np.zeros((n_q, index.view.n_windows), dtype=np.int64)
    full[: att.shape[0]] = att
    t = torch.from_numpy(full)
    dist.all_reduce(t)                                    # the one exchange: the call-count table
    counts = reduce_counts(dist, run.counts())
    if rank == 0:
        np.save(os.path.join(out_dir, "att.npy"), t.numpy())
        np.save(os.path.join(out_dir, "counts.npy"), np.array([counts[k] for k in COUNT_KEYS]))
    al = run.alns()
    np.save(os.path.join(out_dir, f"alns{rank}.npy"), al)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_covers_everything():
    for n, w in [(10, 3), (7, 8), (0, 2), (100, 4)]:
        spans = [shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
        assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


@pytest.mark.timeout(300)
def test_two_ranks_equal_one(msa_dir, tmp_path):
    from groot_amd import device, host, synth
    from oracle import oracle_py as O

    files = host.msa_files(msa_dir)[:24]
    n_reads = 3000
    mp.spawn(_worker, args=(2, _free_port(), files, n_reads, str(tmp_path)), nprocs=2, join=True)
    index = host.Index.from_msa_files(files)
    cat, off, lens = synth.reference_sequences(index)
    seq, so, _ = synth.reads_np(cat, off, lens, n_reads, 100)
    run = O.Run(index)
    run.batch(seq, so)
    att = run.attempts()
    got = np.load(os.path.join(tmp_path, "att.npy"))
    assert np.array_equal(got[: att.shape[0]], att) and not got[att.shape[0]:].any()
    counts = np.load(os.path.join(tmp_path, "counts.npy"))
    c1 = run.counts()
    assert counts.tolist() == [c1[k] for k in COUNT_KEYS]
    # weights from the reduced table = single-process canonical weights (independent of the GPU count)
    kf, kt = device.weights(index, got.astype(np.uint32))
    kf1, kt1 = run.weights(order=1)
    assert np.array_equal(kf, kf1) and np.array_equal(kt, kt1)
    # alignment records of the shards, concatenated in rank order, are the single-process records
    al = np.concatenate([np.load(os.path.join(tmp_path, f"alns{r}.npy")) for r in range(2)])
    assert np.array_equal(al, run.alns())


# ---- the RCCL branch of groot_hip_attempts_allreduce, executed on ONE GPU --------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("n_ctx", [1, 2])
def test_rccl_branch_on_one_gpu(small_index, monkeypatch, hip_lib, n_ctx):
    """GROOT_FORCE_RCCL=1 builds the communicator even though every ctx sits on the same device (ncclCommInitAll over one device),
    so dlopen("librccl"), the six symbols, the hand-copied enum values (ncclUint32 = 3, ncclSum = 0) and the grouped in-place
    ncclAllReduce really run.  One ctx: the table comes back unchanged.  Two ctxs with different shards: the kernel folds the
    second into the first, RCCL all-reduces that single rank, both ctxs end up with the sum = the single-ctx table of all reads."""
    from groot_amd import device, synth

    index = small_index
    cat, off, lens = synth.reference_sequences(index)
    n = 4000
    seq, so, _ = synth.reads_np(cat, off, lens, n, 100)
    monkeypatch.delenv("GROOT_FORCE_RCCL", raising=False)
    ref = device.Aligner(index, max_batch_reads=n, max_read_len=128)
    ref.submit(seq, so)
    ref.wait()
    want_q, want = ref.attempts_rows()
    ref.close()
    assert want.sum() > 0
    monkeypatch.setenv("GROOT_FORCE_RCCL", "1")
    als = [device.Aligner(index, max_batch_reads=n, max_read_len=128) for _ in range(n_ctx)]
    per = n // n_ctx
    for i, al in enumerate(als):
        lo, hi = i * per, (n if i == n_ctx - 1 else (i + 1) * per)
        al.submit(seq[int(so[lo]):int(so[hi])], so[lo:hi + 1] - so[lo], first_read_id=lo)
        al.wait()
    device.attempts_allreduce(als)
    for al in als:
        q, got = al.attempts_rows()
        assert np.array_equal(q, want_q) and np.array_equal(got, want)
    for al in als:
        al.close()
