"""Multi-GPU driver pieces: reads shard across ranks, the index is replicated, and the
only exchange is the sum of the per-(kmerCount, window) IncrementSubPath call counts plus the read
counters after the last batch.  One process per GPU; torch.distributed backend "nccl" (= RCCL over
xGMI) on GPUs, "gloo" in the CPU tests."""
import numpy as np


def shard_range(n_reads, rank, world):
    """contiguous, balanced [lo, hi) of reads for `rank`"""
    base, rem = divmod(n_reads, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


COUNT_KEYS = ("received", "mapped", "multimapped", "alignments", "seeds")


def reduce_counts(dist, counts, device=None):
    """sum the boss counters (boss.go:24-27) over ranks"""
    import torch

    t = torch.tensor([int(counts[k]) for k in COUNT_KEYS], dtype=torch.int64, device=device)
    dist.all_reduce(t)
    return {k: int(v) for k, v in zip(COUNT_KEYS, t.tolist())}


def reduce_attempts(dist, attempts):
    """all-reduce (sum) of the call-count table; `attempts` is a torch int32/int64 tensor (device or host)"""
    dist.all_reduce(attempts)
    return attempts


def attempts_to_numpy(t):
    return t.detach().cpu().numpy().astype(np.uint32, copy=False)
