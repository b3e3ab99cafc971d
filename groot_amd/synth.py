"""Synthetic error-free reads sampled from the indexed reference sequences.

Mirrors what the reference's accuracy script simulates with bbmap randomreads (maxsnps=0,
adderrors=false; testing/run_accuracy_tests.sh:5-6): pick a sequence uniformly, a start uniformly,
emit L bases, flip the strand with p=0.5.  Counter-based (splitmix64 of seed+3*i+j), so any shard of
the read stream can be produced independently and identically with numpy (host) or torch (device).
"""
import numpy as np

SEED = 0x67726F6F74  # "groot"
_M64 = (1 << 64) - 1


def reference_sequences(index):
    """concatenated linear path sequences of every graph (Graph2Seqs) -> (uint8 cat, offsets, lengths)"""
    a = index.arrays
    seqs = []
    for g in range(index.view.n_graphs):
        for lp in range(int(a["graph_path_off"][g + 1] - a["graph_path_off"][g])):
            seqs.append(index.path_sequence(g, lp))
    lens = np.array([len(s) for s in seqs], dtype=np.int64)
    off = np.zeros(len(seqs) + 1, dtype=np.int64)
    off[1:] = np.cumsum(lens)
    cat = np.frombuffer(b"".join(seqs), dtype=np.uint8).copy()
    return cat, off, lens


def _mix_np(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15))
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


_COMP = np.zeros(256, dtype=np.uint8)
for _a, _b in zip(b"ACGTN", b"TGCAN"):
    _COMP[_a] = _b


def plan_np(lens, n_reads, read_len, first=0, seed=SEED, min_len=None):
    """(sequence index, start, strand, length) of reads first..first+n_reads-1"""
    with np.errstate(over="ignore"):
        i = np.arange(first, first + n_reads, dtype=np.uint64) * np.uint64(4) + np.uint64(seed)
        r1, r2, r3, r4 = (_mix_np(i + np.uint64(j)) for j in range(4))
    ok = np.flatnonzero(lens >= read_len)
    sidx = ok[((r1 >> np.uint64(1)) % np.uint64(len(ok))).astype(np.int64)]
    if min_len is None or min_len >= read_len:
        rl = np.full(n_reads, read_len, dtype=np.int64)
    else:
        rl = min_len + ((r4 >> np.uint64(1)) % np.uint64(read_len - min_len + 1)).astype(np.int64)
    span = (lens[sidx] - rl + 1).astype(np.uint64)
    start = ((r2 >> np.uint64(1)) % span).astype(np.int64)
    strand = (r3 & np.uint64(1)).astype(np.int64)
    return sidx, start, strand, rl


def reads_np(cat, off, lens, n_reads, read_len=100, first=0, seed=SEED, min_len=None):
    """host generation: (seq_concat uint8, seq_off uint64[n+1], truth dict)"""
    sidx, start, strand, rl = plan_np(lens, n_reads, read_len, first, seed, min_len)
    seq_off = np.zeros(n_reads + 1, dtype=np.uint64)
    seq_off[1:] = np.cumsum(rl).astype(np.uint64)
    total = int(seq_off[-1])
    # position of every base inside its read
    read_of = np.repeat(np.arange(n_reads), rl)
    within = np.arange(total, dtype=np.int64) - seq_off[:-1].astype(np.int64)[read_of]
    fwd_pos = np.where(strand[read_of] == 0, within, rl[read_of] - 1 - within)
    src = off[sidx][read_of] + start[read_of] + fwd_pos
    b = cat[src]
    out = np.where(strand[read_of] == 0, b, _COMP[b]).astype(np.uint8)
    return out, seq_off, {"seq": sidx, "start": start, "strand": strand, "len": rl}


def reads_torch(cat_t, off_t, lens_t, n_reads, read_len=100, first=0, seed=SEED):
    """device generation of fixed-length reads with torch (same stream as reads_np):
    returns (uint8 tensor [n*L] padded to a multiple of 16 + 16, int64 offsets-as-uint64 bits [n+1], truth)"""
    import torch

    dev = cat_t.device

    def c(v):  # python int -> wrapped int64 scalar
        v &= _M64
        return v - (1 << 64) if v >= (1 << 63) else v

    def lsr(x, s):
        return (x >> s) & ((1 << (64 - s)) - 1)

    def mix(x):
        x = x + c(0x9E3779B97F4A7C15)
        x = (x ^ lsr(x, 30)) * c(0xBF58476D1CE4E5B9)
        x = (x ^ lsr(x, 27)) * c(0x94D049BB133111EB)
        return x ^ lsr(x, 31)

    i = torch.arange(first, first + n_reads, dtype=torch.int64, device=dev) * 4 + c(seed)
    r1, r2, r3 = mix(i), mix(i + 1), mix(i + 2)
    ok = torch.nonzero(lens_t >= read_len).squeeze(1)
    sidx = ok[lsr(r1, 1) % ok.numel()]
    span = lens_t[sidx] - read_len + 1
    start = lsr(r2, 1) % span
    strand = r3 & 1
    within = torch.arange(read_len, dtype=torch.int64, device=dev).unsqueeze(0)
    fwd_pos = torch.where(strand.unsqueeze(1) == 0, within, read_len - 1 - within)
    src = (off_t[sidx] + start).unsqueeze(1) + fwd_pos
    b = cat_t[src.reshape(-1)]
    comp = torch.from_numpy(_COMP).to(dev)
    rc_mask = (strand.unsqueeze(1) == 1).expand(-1, read_len).reshape(-1)
    out = torch.where(rc_mask, comp[b.long()], b)
    total = n_reads * read_len
    padded = torch.zeros(((total + 15) // 16) * 16 + 16, dtype=torch.uint8, device=dev)
    padded[:total] = out
    seq_off = torch.arange(0, n_reads + 1, dtype=torch.int64, device=dev) * read_len
    return padded, seq_off, {"seq": sidx, "start": start, "strand": strand}


def reads_torch_mixed(cat_t, off_t, lens_t, n_reads, read_len=150, min_len=75, first=0, seed=SEED):
    """device generation of reads of lengths U{min_len..read_len} (BASELINE configs[4]); the same stream as
    reads_np(..., read_len, min_len=min_len): (uint8 tensor padded by 64 bytes, int64 offsets [n+1], truth)"""
    import torch

    dev = cat_t.device

    def c(v):
        v &= _M64
        return v - (1 << 64) if v >= (1 << 63) else v

    def lsr(x, s):
        return (x >> s) & ((1 << (64 - s)) - 1)

    def mix(x):
        x = x + c(0x9E3779B97F4A7C15)
        x = (x ^ lsr(x, 30)) * c(0xBF58476D1CE4E5B9)
        x = (x ^ lsr(x, 27)) * c(0x94D049BB133111EB)
        return x ^ lsr(x, 31)

    i = torch.arange(first, first + n_reads, dtype=torch.int64, device=dev) * 4 + c(seed)
    r1, r2, r3, r4 = mix(i), mix(i + 1), mix(i + 2), mix(i + 3)
    ok = torch.nonzero(lens_t >= read_len).squeeze(1)
    sidx = ok[lsr(r1, 1) % ok.numel()]
    rl = min_len + lsr(r4, 1) % (read_len - min_len + 1)
    span = lens_t[sidx] - rl + 1
    start = lsr(r2, 1) % span
    strand = r3 & 1
    seq_off = torch.zeros(n_reads + 1, dtype=torch.int64, device=dev)
    seq_off[1:] = torch.cumsum(rl, 0)
    total = int(seq_off[-1].item())
    read_of = torch.repeat_interleave(torch.arange(n_reads, device=dev), rl)
    within = torch.arange(total, dtype=torch.int64, device=dev) - seq_off[:-1][read_of]
    fwd_pos = torch.where(strand[read_of] == 0, within, rl[read_of] - 1 - within)
    src = off_t[sidx][read_of] + start[read_of] + fwd_pos
    b = cat_t[src]
    comp = torch.from_numpy(_COMP).to(dev)
    out = torch.where(strand[read_of] == 1, comp[b.long()], b)
    padded = torch.zeros(((total + 15) // 16) * 16 + 64, dtype=torch.uint8, device=dev)
    padded[:total] = out
    return padded, seq_off, {"seq": sidx, "start": start, "strand": strand, "len": rl}
