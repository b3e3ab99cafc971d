#!/usr/bin/env python3
"""bench.py -- Mreads/s aligned on MI355X for BASELINE.json's workload (configs[2]).

A step = one pass of the whole `groot align` hot path (seed stage -> align stage -> canonical ordering of the traversal
records) over one batch of synthetic 100 bp reads that is already resident in HBM; the records stay in HBM (`value`, as the
bench contract asks: inputs resident when the timed region starts).  **`value` is the rate THROUGH THE KERNELS north_star names**:
the ctx is opened with the memo of groot_hip_open switched off, so every read is hashed (sketch_sig_kernel; the reads it cannot
decide: the full-width list pass), looked up and walked through its graph (align_kernel).  Reads shard across GPUs (one process
per GPU, index replicated); the only exchange is one RCCL all-reduce of the IncrementSubPath call counts after the last step
(inside the timed region); with --gpus N every rank's own ms per step and its all-reduce time are on the line (per_rank).

`roofline` describes the longest kernel of that step from live HIP events and carries, as FLAT scalars (the driver keeps those):
kernel_path_mreads / _ms_per_step, the per-kernel ms and fractions, valu_issue_* (wave-instructions against the issue ceiling),
and the rates of the legs below (memo_mreads, sub1_mreads, mixed99_mreads, host_fed_mreads, cli_e2e_mreads ...).

Beside `value` the same JSON line carries, at N=1:
  kernel_path   the `value` ctx on reads with 1 % substitutions
  memo_tier     configs[2] on the library's DEFAULT ctx (memo on: error-free window-sized reads are its keys -- a table look-up)
  robustness    the default ctx on reads the memo cannot answer: 1 % substitutions per base; 99 % random reads
  thresholds    configs[4]'s containment-threshold sweep on the 100 bp reads (t = 0.97, 0.95, 0.90)
  mixed         configs[4] at single-GPU scale: resfinder.90, 8 M reads of 75..150 bases, both strands, t = 0.99 .. 0.90,
                and one gzip-streamed run of build/groot-hip align
  host_fed      pinned host buffers -> groot_hip_submit_acquired (2-bit bases over PCIe) -> kernels -> traversal records
                back in pinned host memory (groot_hip_collect), several batches in flight in ONE ctx: SURVEY 8d's
                "first submit -> last collect" rate, PCIe in both directions included; runs for >= 5 s
  cli_e2e       build/groot-hip align: FASTQ file -> BAM file + GFAs, the whole process
  cpu_baseline  the oracle (CPU restatement of the reference path) as one process per granted CPU

  python bench.py --gpus 1 --steps 200 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import subprocess
import sys
import tarfile
import tempfile
import time

import numpy as np

# (before torch initialises the HIP runtime: a ctx runs four streams beside torch's -- groot_hip.hip groot_hw_queues)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

READ_LEN = 100
HBM_PEAK_GBS = 8000.0  # spec peak, /opt/skills/guides/MI355X_MICROARCH.md (6.29 TB/s measured copy rate)
PMC_FILE = next((f for f in (os.path.join(REPO, "profiles", n) for n in ("r06_pmc.json", "r05_pmc.json", "r04_pmc.json")) if os.path.exists(f)), "")   # {workload: {kernel: per-launch counters}}, tools/profile_r05.sh


def usable_cpus():
    """CPUs this process may really use: min(affinity mask, cgroup CPU quota) -- the GPU boxes show 256 hardware threads and grant 16"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, round(int(q) / int(p))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, round(q / p)))
        except Exception:
            pass
    return max(1, n)


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def load_index(db="arg-annot.90"):
    """arg-annot.90 (or resfinder.90), k=31 s=21 w=100 x=8 y=4 (cmd/index.go:45-49 defaults), cached under build/"""
    from groot_amd import host

    cache = host.index_cache_path(db + ".k31.s21.w100")
    if os.path.exists(cache):
        try:
            return host.Index.load(cache), cache
        except Exception:
            pass
    with tempfile.TemporaryDirectory() as td:
        with tarfile.open(os.path.join(REPO, "tests", "golden", "data", db + ".tar.gz")) as tf:
            members = [m for m in tf.getmembers() if os.path.basename(m.name).startswith("cluster") and m.name.endswith(".msa")]
            tf.extractall(td, members=members)
        index = host.Index.from_msa_dir(os.path.join(td, db))
    try:
        tmp = cache + ".%d.tmp" % os.getpid()
        index.save(tmp)
        os.replace(tmp, cache)
    except Exception:
        pass
    return index, cache


# ---------------------------------------------------------------------------------------------------------------------
def cpu_baseline(index_path, seconds):
    """The oracle (C port of the reference path, single thread per instance) on this box's host cores: one PROCESS per
    hardware thread, each on its own slice of the synthetic stream, all started together (oracle/cpu_worker.py).  The
    reference's Go binary cannot be built here (no Go toolchain): kind = "port"."""
    worker = os.path.join(REPO, "oracle", "cpu_worker.py")

    def run(n_proc, per, first0):
        procs = [subprocess.Popen([sys.executable, worker, index_path, str(first0 + i * per), str(per), str(READ_LEN), "25000"],
                                  stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True) for i in range(n_proc)]
        for p in procs:
            if p.stdout.readline().strip() != "ready":
                raise RuntimeError("cpu worker failed to start")
        for p in procs:
            p.stdin.write("go\n")
            p.stdin.flush()
        res = [json.loads(p.stdout.readline()) for p in procs]
        for p in procs:
            p.wait()
        wall = max(r["t_end"] for r in res) - min(r["t_start"] for r in res)
        return sum(r["reads"] for r in res) / wall / 1e6, wall

    one_n = 150_000
    one, one_wall = run(1, one_n, 0)
    cores = usable_cpus()
    per = int(max(50_000, min(600_000, one * 1e6 * seconds * 0.6)) // 25_000 * 25_000)   # SMT siblings run slower than a lone thread
    allv, wall = run(cores, per, 1_000_000)
    return {"value": allv, "unit": "Mreads/s", "cores": cores, "kind": "port", "hardware_threads": os.cpu_count(),
            "cores_note": "cores = CPUs this container may use (min of affinity mask and cgroup CPU quota); the box shows more hardware threads than it grants",
            "sample": f"{cores} oracle processes x {per} reads of the synthetic stream, together, {wall:.1f} s wall",
            "sample_note": "records dropped per 25k-read chunk as the reference streams them to the BAM",
            "single_core": {"value": one, "sample": f"{one_n} reads, one process, {one_wall:.1f} s"},
            "scaling_efficiency": allv / (one * cores),
            "note": "oracle/groot_oracle.c = CPU restatement of the reference path; the Go binary itself cannot be built in this image"}


# ---------------------------------------------------------------------------------------------------------------------
def pmc_of(workload, kernel):
    """per-launch PMC figures of `kernel` on `workload` from the committed profile (separate rocprofv3 --pmc passes of tools/kernel_path_probe.py:
    they cannot run inside this process); {} when absent"""
    try:
        return json.load(open(PMC_FILE)).get(workload, {}).get(kernel, {})
    except Exception:
        return {}


def pmc_commit():
    """the commit the PMC passes of PMC_FILE were taken on: the file's own "_meta" entry, else the known ones"""
    try:
        m = json.load(open(PMC_FILE)).get("_meta", {})
        if m.get("commit"):
            return m["commit"]
    except Exception:
        pass
    return {"r05_pmc.json": "6cad506", "r04_pmc.json": "fb627ca"}.get(os.path.basename(PMC_FILE))


def kernel_blocks(workload, ms, counts, R, mean_len, pw):
    """per-kernel roofline of the hashing / graph-walk path for one workload: live HIP-event time of the kernel (groot_stage_ms), the
    algorithmic bytes of SURVEY 8d for the reads that kernel handles, and -- from the committed PMC passes -- HBM traffic and VALU issue"""
    todo, walked, travs = counts["full_sketch_reads"], counts["walked_reads"], counts["travs"]
    spec = {
        # every read: bases + u64 offset in; seed count, scheduling key, 32-byte read record out
        "sketch_sig_kernel": (ms["first_seed_kernel"], R, R * (mean_len + 8) + R * (4 + 4 + 32)),
        # the reads the first kernel could not decide: bases + offset in, seeds + key + record out
        "sketch_seed_kernel<LIST>": (ms["list_pass"], todo, todo * (mean_len + 8) + todo * (4 + 4 + 32) + 4 * counts["seeds"]),
        # the reads that need the graph walk: record + bases + seeds in; traversal record + path set + count out
        "align_kernel": (ms["align"], walked, walked * (32 + mean_len + 4) + (20 + 8 * pw) * travs + 4 * R),
    }
    out = {}
    for k, (t_ms, n, b) in spec.items():
        e = {"kernel_ms": t_ms, "reads": n, "bytes_per_launch": b}
        if t_ms > 0:
            e["achieved"] = b / (t_ms * 1e-3) / 1e9
            e["frac"] = e["achieved"] / HBM_PEAK_GBS
        p = pmc_of(workload, k) or pmc_of(workload, k.split("<")[0])
        if p:
            e["traffic"] = p.get("hbm_bytes_per_launch")
            pl = p.get("per_launch", {})
            if pl.get("SQ_INSTS_VALU") and p.get("avg_ms"):
                # share of the chip's VALU issue slots the kernel used: wave-instructions x 4 cycles / (1024 SIMDs x 2.4 GHz x duration);
                # duration = the kernel's average in the trace of the same workload (it runs beside the other stage's kernels there too)
                e["valu_issue"] = 4.0 * pl["SQ_INSTS_VALU"] / (1024 * 2.4e9 * p["avg_ms"] * 1e-3)
            if pl.get("SQ_WAVE_CYCLES"):
                e["wait_frac"] = pl.get("SQ_WAIT_ANY", 0.0) / pl["SQ_WAVE_CYCLES"]     # share of the resident waves' cycles spent waiting
            if p.get("l2_hit_rate") is not None:
                e["l2_hit_rate"] = p.get("l2_hit_rate")
            e["pmc_kernel_ms"] = p.get("avg_ms")
            e["traffic_source"] = "profiles/%s[%s]" % (os.path.basename(PMC_FILE), workload)
        out[k] = e
    return out


def substituted(d_seq, R, p, gen):
    """a copy of the batch with every base replaced by another one with probability p"""
    import torch

    dev = d_seq.device
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    err = d_seq.clone()
    rows = err[: R * READ_LEN].view(R, READ_LEN)
    CH = 1_000_000
    for c0 in range(0, R, CH):
        n = min(CH, R - c0)
        hit = torch.rand(n, READ_LEN, generator=gen, device=dev) < p
        cur = torch.searchsorted(acgt, rows[c0:c0 + n].contiguous())          # A C G T -> 0..3
        other = acgt[(cur + 1 + torch.randint(0, 3, (n, READ_LEN), generator=gen, device=dev)) % 4]
        rows[c0:c0 + n] = torch.where(hit, other, rows[c0:c0 + n])
    return err


def resident_rate(al, d_seq_ptr, d_off_ptr, R, max_len, steps, warmup, mixed=False):
    """`steps` batches of R reads that sit in HBM through ctx `al`, two in flight (the ctx is a pipeline: the next batch is
    enqueued while the GPU works on this one); returns (Mreads/s, mean stage ms, counts of the last batch)"""
    import torch

    for _ in range(warmup):
        al.submit_device(d_seq_ptr, d_off_ptr, R, first_read_id=0, max_len=max_len, mixed=mixed)
        al.wait()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    stage, pending, counts = {}, 0, None
    for i in range(steps + 1):
        if i < steps:
            al.submit_device(d_seq_ptr, d_off_ptr, R, first_read_id=0, max_len=max_len, mixed=mixed)
            pending += 1
        if pending == int(os.environ.get("GROOT_PROBE_INFLIGHT", "2")) or (i == steps and pending):   # (1: a batch at a time -- every stage alone on the chip, for tools/)
            counts = al.wait()
            pending -= 1
            for k, v in al.stage_ms().items():
                stage[k] = stage.get(k, 0.0) + v
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return steps * R / dt / 1e6, {k: v / steps for k, v in stage.items()}, counts


def robustness(al, index, d_seq, d_off, R, steps):
    """the same ctx on inputs the memo cannot answer (DESIGN.md "memo"): every path below is the round-2 path"""
    import torch

    dev = d_seq.device
    out = {}
    g = torch.Generator(device=dev)
    g.manual_seed(0x67726F6F74)
    CH = 1_000_000
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    # (a) substitution errors: every base is replaced by another one with probability 0.01 (63 % of the reads hold at least one)
    err = substituted(d_seq, R, 0.01, g)
    v, ms, c = resident_rate(al, err.data_ptr(), d_off.data_ptr(), R, READ_LEN, steps, 2)
    out["substitutions_1pct"] = {"value": v, "unit": "Mreads/s", "stage_ms": ms, "mapped": c["mapped"], "full_sketch_reads": c["full_sketch_reads"],
                                 "walked_reads": c["walked_reads"], "what": "each base replaced with probability 0.01: reads with an error take the hashing kernels",
                                 "kernels": kernel_blocks("sub1", ms, c, R, READ_LEN, index.view.path_words)}
    # (b) metagenome-like: 99 % of the reads are uniform random ACGT
    rows = err[: R * READ_LEN].view(R, READ_LEN)
    rows[:] = d_seq[: R * READ_LEN].view(R, READ_LEN)
    for c0 in range(0, R, CH):
        n = min(CH, R - c0)
        bg = torch.rand(n, generator=g, device=dev) < 0.99
        rnd = acgt[torch.randint(0, 4, (n, READ_LEN), generator=g, device=dev)]
        rows[c0:c0 + n] = torch.where(bg[:, None], rnd, rows[c0:c0 + n])
    v, ms, c = resident_rate(al, err.data_ptr(), d_off.data_ptr(), R, READ_LEN, steps, 3)
    out["background_99pct"] = {"value": v, "unit": "Mreads/s", "stage_ms": ms, "mapped": c["mapped"], "full_sketch_reads": c["full_sketch_reads"],
                               "what": "99 % uniform random reads: the text lookup misses, the ctx goes back to the signature kernel after one batch"}
    del err
    return out


def threshold_sweep(index, d_seq, d_off, R, steps, local_rank):
    """configs[4]'s containment-threshold sweep on the headline reads: a ctx per threshold (its memo is the pipeline's output AT that threshold)"""
    from groot_amd import device

    out = {}
    for t in (0.97, 0.95, 0.90):
        al = device.Aligner(index, device=local_rank, threshold=t, max_batch_reads=R, max_read_len=256, max_batch_bases=R * READ_LEN + 64,
                            results_on_device=True, pipeline_depth=2)
        al.set_profiling(True)
        v, ms, c = resident_rate(al, d_seq.data_ptr(), d_off.data_ptr(), R, READ_LEN, steps, 2)
        out["t=%.2f" % t] = {"value": v, "unit": "Mreads/s", "seeds_per_read": c["seeds"] / R, "alignments": c["alignments"], "stage_ms": ms,
                             "full_sketch_reads": c["full_sketch_reads"], "open_ms": al.open_stats()["open_ms"]}
        al.close()
    return out


def mixed_leg(local_rank, n_reads, steps, cli_reads, bam_level):
    """BASELINE configs[4] at single-GPU scale: resfinder.90 (card.90 is not in the reference tree), reads of 75..150 bases of both
    strands, threshold sweep (kernels, reads resident in HBM), and one gzip-streamed run through build/groot-hip align"""
    import gzip

    import torch

    from groot_amd import device, host, synth

    index, _ = load_index("resfinder.90")
    dev = torch.device("cuda", local_rank)
    cat, off, lens = synth.reference_sequences(index)
    cat_t, off_t, lens_t = (torch.from_numpy(x).to(dev) for x in (cat, off, lens))
    d_seq, d_off, _ = synth.reads_torch_mixed(cat_t, off_t, lens_t, n_reads, 150, 75)
    total = int(d_off[-1].item())
    out = {"index": "resfinder.90 k=31 s=21 w=100: %d graphs, %d windows" % (index.view.n_graphs, index.view.n_windows), "reads": n_reads,
           "read_len": "U{75..150}", "mean_len": total / n_reads, "kernels": {},
           "what": "reads resident in HBM, records stay in HBM (as `value`); reads of other lengths than the window are not in the memo: this is the hashing + graph-walk path"}
    for t in (0.99, 0.97, 0.95, 0.90):
        al = device.Aligner(index, device=local_rank, threshold=t, max_batch_reads=n_reads, max_read_len=256, max_batch_bases=total + 64,
                            results_on_device=True, pipeline_depth=2)
        al.set_profiling(True)
        v, ms, c = resident_rate(al, d_seq.data_ptr(), d_off.data_ptr(), n_reads, 150, steps, 2, mixed=True)
        out["kernels"]["t=%.2f" % t] = {"value": v, "unit": "Mreads/s", "mapped": c["mapped"], "seeds_per_read": c["seeds"] / n_reads,
                                         "alignments": c["alignments"], "walked_reads": c["walked_reads"], "full_sketch_reads": c["full_sketch_reads"], "stage_ms": ms}
        if t in (0.99, 0.90):
            out["kernels"]["t=%.2f" % t]["kernels"] = kernel_blocks("mixed%d" % round(t * 100), ms, c, n_reads, total / n_reads, index.view.path_words)
        if t == 0.99 and n_reads > 2_000_000:
            # the align stage of such a batch lasts at least as long as its slowest read (~2 ms: 150 dependent steps): smaller batches
            # of the same stream pay that floor for fewer reads
            n2 = 2_000_000
            v2, ms2, c2 = resident_rate(al, d_seq.data_ptr(), d_off.data_ptr(), n2, 150, steps, 1, mixed=True)
            out["kernels"]["t=0.99, batches of 2 M reads"] = {"value": v2, "unit": "Mreads/s", "mapped": c2["mapped"], "walked_reads": c2["walked_reads"], "stage_ms": ms2}
        al.close()
    # gzip-streamed through the CLI (t = 0.97)
    try:
        import __graft_entry__ as entry

        exe = entry.build_cli()
        n = min(cli_reads, n_reads)
        seq_h = d_seq[: int(d_off[n].item())].cpu().numpy()
        off_h = d_off[: n + 1].cpu().numpy()
        with tempfile.TemporaryDirectory(dir=os.environ.get("GROOT_BENCH_TMP")) as td:
            idx_dir = os.path.join(td, "index")
            os.makedirs(idx_dir)
            index.save(os.path.join(idx_dir, "groot.gidx"))
            fq = os.path.join(td, "reads.fq.gz")
            t0 = time.perf_counter()
            with gzip.open(fq, "wb", compresslevel=1) as f:
                CHK = 100_000
                for c0 in range(0, n, CHK):
                    c1 = min(n, c0 + CHK)
                    parts = []
                    for i in range(c0, c1):
                        s_ = seq_h[off_h[i]:off_h[i + 1]].tobytes()
                        parts.append(b"@m%d\n%s\n+\n%s\n" % (i, s_, b"I" * len(s_)))
                    f.write(b"".join(parts))
            gz_s = time.perf_counter() - t0
            stats = os.path.join(td, "stats.json")
            cmd = [exe, "align", "-i", idx_dir, "-f", fq, "-g", os.path.join(td, "graphs"), "--bam", os.path.join(td, "out.bam"), "--log", os.path.join(td, "groot.log"),
                   "-p", str(usable_cpus()), "-t", "0.97", "--bamLevel", str(bam_level), "--stats", stats, "--batch", "262144"]
            t0 = time.perf_counter()
            p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            wall = time.perf_counter() - t0
            if p.returncode != 0:
                out["cli_gzip"] = {"error": (p.stderr or p.stdout)[-300:]}
            else:
                st = json.load(open(stats))
                out["cli_gzip"] = {"value": n / wall / 1e6, "unit": "Mreads/s", "reads": n, "threshold": 0.97, "wall_s": wall, "stream_value": n / st["stream_s"] / 1e6,
                                   "fastq_gz_bytes": os.path.getsize(fq), "gzip_write_s": gz_s, "phases_s": st,
                                   "what": "build/groot-hip align on ONE gzip FASTQ (inflated on one thread, as bufio over gzip.Reader in the reference; round 5: by the repo's own decoder, gz_inflate.hpp): whole process"}
    except Exception as e:
        out["cli_gzip"] = {"error": repr(e)}
    return out


# ---------------------------------------------------------------------------------------------------------------------
def host_fed(index, d_seq, R, steps, depth=4):
    """submit -> collect through pinned host memory, `depth` batches in flight in one ctx.  The packed batch is written
    into each slot's pinned staging once (a FASTQ parser's job in the CLI); the timed loop is acquire -> submit_acquired ->
    collect -> release, i.e. H2D of 27 B/read, the kernels, D2H of the traversal records."""
    import torch

    from groot_amd import device, host

    seq_host = d_seq[: R * READ_LEN].cpu().numpy()
    packed, exc_pos, exc_byte = host.pack_reads(seq_host)
    lens = np.full(R, READ_LEN, dtype=np.uint16)
    al = device.Aligner(index, device=torch.cuda.current_device(), max_batch_reads=R, max_read_len=256, max_batch_bases=R * READ_LEN + 64,
                        pipeline_depth=depth)
    al.set_profiling(os.environ.get("GROOT_BENCH_HF_EVENTS", "1") != "0")
    stage = {}
    bufs = [al.acquire() for _ in range(depth)]
    for b in bufs:
        b["packed"][: len(packed)] = packed
        b["seq_len"][:R] = lens
        b["exc_pos"][: len(exc_pos)] = exc_pos
        b["exc_byte"][: len(exc_byte)] = exc_byte
    for b in bufs:                                   # warm-up: also sizes the pinned result buffers
        al.submit_acquired(b["ticket"], R, len(exc_pos))
    first = None
    for _ in range(depth):
        r = al.collect(copy=False)
        first = first or r["counts"]
        al.release(r["ticket"])
    al.attempts_reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    done = 0
    trav_bytes = 0
    host_s = {"submit": 0.0, "collect": 0.0}         # where the calling thread spends the batch period
    for i in range(steps):
        b = al.acquire()                             # a free slot: its staging still holds the packed batch
        ta = time.perf_counter()
        al.submit_acquired(b["ticket"], R, len(exc_pos))
        tb = time.perf_counter()
        host_s["submit"] += tb - ta
        if al.in_flight()[0] == depth:
            r = al.collect(copy=False)
            host_s["collect"] += time.perf_counter() - tb
            trav_bytes += r["n_travs"] * 12 + r["n_mask_bytes"] + (r["n_travs"] // 256 + 1) * 4   # (12-byte records on the wire)
            for k, v in r["ms"].items():
                stage[k] = stage.get(k, 0.0) + v
            al.release(r["ticket"])
            done += 1
    while done < steps:
        r = al.collect(copy=False)
        trav_bytes += r["n_travs"] * 12 + r["n_mask_bytes"] + (r["n_travs"] // 256 + 1) * 4   # (12-byte records on the wire)
        for k, v in r["ms"].items():
            stage[k] = stage.get(k, 0.0) + v
        al.release(r["ticket"])
        done += 1
    dt = time.perf_counter() - t0
    out = {"value": steps * R / dt / 1e6, "unit": "Mreads/s", "steps": steps, "batches_in_flight": depth, "ms_per_batch": dt / steps * 1e3,
           "h2d_bytes_per_read": (len(packed) + 9 * len(exc_pos)) / R,   # (all reads are 100 bp: the length array stays at home)
           "d2h_bytes_per_read": trav_bytes / (steps * R),
           "stage_ms_per_batch": {k: v / steps for k, v in stage.items()},
           "caller_ms_per_batch": {k: v / steps * 1e3 for k, v in host_s.items()},
           "what": "one ctx, one index replica: pinned staging -> H2D (2-bit bases + u16 lengths) -> kernels -> D2H of the traversal "
                   "records (12 bytes each, expanded to groot_trav by collect) into pinned host memory; first submit -> last collect"}
    # the plain-ASCII entry point with pageable caller memory (what a cgo caller handing over Go slices gets)
    off = np.arange(R + 1, dtype=np.uint64) * READ_LEN
    n_ascii = max(3, steps // 4)
    t0 = time.perf_counter()
    done = 0
    for i in range(n_ascii):
        al.submit(seq_host, off)
        if al.in_flight()[0] == depth:
            r = al.collect(copy=False)
            al.release(r["ticket"])
            done += 1
    while done < n_ascii:
        r = al.collect(copy=False)
        al.release(r["ticket"])
        done += 1
    dta = time.perf_counter() - t0
    out["ascii_pageable"] = {"value": n_ascii * R / dta / 1e6, "unit": "Mreads/s", "steps": n_ascii,
                             "what": "groot_hip_submit: ASCII bases + u64 offsets copied from pageable memory into pinned staging, 108 B/read over PCIe"}
    al.close()
    return out, first


# ---------------------------------------------------------------------------------------------------------------------
def write_fastq(path, seq_host, n):
    """n fixed-length records '@r%09d\\nSEQ\\n+\\nIII..\\n' straight from the ASCII read matrix"""
    L = READ_LEN
    rec = np.empty((n, 1 + 10 + 1 + L + 3 + L + 1), dtype=np.uint8)
    rec[:, 0] = ord("@")
    rec[:, 1] = ord("r")
    idx = np.arange(n, dtype=np.int64)
    for d in range(9):
        rec[:, 2 + d] = (idx // 10 ** (8 - d)) % 10 + ord("0")
    rec[:, 11] = ord("\n")
    rec[:, 12:12 + L] = seq_host[: n * L].reshape(n, L)
    rec[:, 12 + L] = ord("\n")
    rec[:, 13 + L] = ord("+")
    rec[:, 14 + L] = ord("\n")
    rec[:, 15 + L:15 + 2 * L] = ord("I")
    rec[:, 15 + 2 * L] = ord("\n")
    with open(path, "wb") as f:
        f.write(rec.tobytes())
    return rec.shape[1] * n


def cli_e2e(index, d_seq, n_reads, bam_level):
    """FASTQ file -> build/groot-hip align -> BAM file + GFAs: the whole process, wall clock"""
    import __graft_entry__ as entry

    exe = entry.build_cli()
    seq_host = d_seq[: n_reads * READ_LEN].cpu().numpy()
    with tempfile.TemporaryDirectory(dir=os.environ.get("GROOT_BENCH_TMP")) as td:
        idx_dir = os.path.join(td, "index")
        os.makedirs(idx_dir)
        index.save(os.path.join(idx_dir, "groot.gidx"))
        fq = os.path.join(td, "reads.fq")
        fq_bytes = write_fastq(fq, seq_host, n_reads)
        bam = os.path.join(td, "out.bam")
        stats = os.path.join(td, "stats.json")
        cmd = [exe, "align", "-i", idx_dir, "-f", fq, "-g", os.path.join(td, "graphs"), "--bam", bam, "--log", os.path.join(td, "groot.log"),
               "-p", str(usable_cpus()), "--bamLevel", str(bam_level), "--stats", stats, "--batch", "262144"]
        t0 = time.perf_counter()
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        wall = time.perf_counter() - t0
        if p.returncode != 0:
            return {"error": (p.stderr or p.stdout)[-400:]}
        if os.environ.get("GROOT_BAM_STATS"):
            for ln in [x for x in p.stderr.splitlines() if "groot bam" in x][-4:]:
                log(ln)
        st = json.load(open(stats))
        out = {"value": n_reads / wall / 1e6, "unit": "Mreads/s", "reads": n_reads, "wall_s": wall, "fastq_bytes": fq_bytes,
               "bam_bytes": os.path.getsize(bam), "bam_level": bam_level, "stream_value": n_reads / st["stream_s"] / 1e6,
               "phases_s": st, "what": "build/groot-hip align: plain FASTQ file -> BAM file + GFAs; value = reads / whole-process wall "
                                       "(index load + device open included), stream_value = reads / (first read parsed -> BAM closed)"}
        return out


# ---------------------------------------------------------------------------------------------------------------------
# ---------------------------------------------------------------------------------------------------------------------
COMPACT_LIMIT = 8000     # the driver keeps an 8 KB tail of stdout and parses the last line; round 5's 20 KB line did not parse


def _r(v, sig=5):
    """floats to `sig` significant digits (the compact line's bytes go to names, not to digits)"""
    if isinstance(v, bool) or v is None or isinstance(v, (int, str)):
        return v
    if isinstance(v, float):
        if v != v or v in (float("inf"), float("-inf")):
            return None
        return float("%.*g" % (sig, v))
    return v


def _flat(d):
    """keep a dict's scalars (rounded); strings cut at 110 characters (the driver cuts at ~120)"""
    out = {}
    for k, v in d.items():
        if isinstance(v, (dict, list, tuple)):
            continue
        out[k] = v[:110] if isinstance(v, str) else _r(v)
    return out


def compact_line(full):
    """The ONE stdout line the driver parses: bench-contract scalars, `config`, `roofline`, `cpu_baseline` -- every value below the three
    objects is a scalar (depth 2), under COMPACT_LIMIT bytes.  Everything else (per-kernel blocks, per-leg stage times, counters) goes to
    bench_full.json and stderr.  tests/test_bench_line.py holds it to that."""
    top = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    line = {k: _r(full.get(k), 6) for k in top}
    cfg = full.get("config", {})
    line["config"] = _flat({k: cfg.get(k) for k in ("workload", "reads_per_gpu_per_step", "read_len", "parallelism", "memo", "walked_reads",
                                                     "full_sketch_reads", "mapped", "alignments", "travs", "seeds", "background_fraction") if cfg.get(k) is not None})
    rf = _flat(full.get("roofline", {}))
    rf.pop("note", None)
    line["roofline"] = rf
    pr = full.get("per_rank")
    if pr:
        # N > 1: every rank's own ms per step and all-reduce ms, flat (rank_ms_per_step_min/max and allreduce_ms_max are already in rf)
        for i, (a, b) in enumerate(zip(pr.get("ms_per_step", []), pr.get("allreduce_ms", []))):
            if i < 16:
                rf["rank%d_ms_per_step" % i] = _r(a)
                rf["rank%d_allreduce_ms" % i] = _r(b)
    cb = full.get("cpu_baseline")
    if cb:
        c = _flat(cb)
        c.pop("note", None)
        c.pop("cores_note", None)
        if isinstance(cb.get("single_core"), dict):
            c["single_core"] = _r(cb["single_core"].get("value"))
        line["cpu_baseline"] = c
    for k in ("host_fed", "cli_e2e", "mixed", "memo_tier", "thresholds", "kernel_path", "lean_first"):
        if isinstance(full.get(k), dict) and "error" in full[k]:
            line.setdefault("leg_errors", {})[k] = str(full[k]["error"])[:110]
    s = json.dumps(line, separators=(",", ":"))
    if len(s) >= COMPACT_LIMIT:      # never expected; drop the least important scalars rather than lose the record again
        for k in sorted(rf, key=lambda k_: (k_ in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic"), -len(k_))):
            if len(s) < COMPACT_LIMIT:
                break
            if k not in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic"):
                rf.pop(k)
                s = json.dumps(line, separators=(",", ":"))
    return s


def emit(full):
    """full object -> bench_full.json (repo root, and gpurun_out/ when there is one) + stderr; compact line -> stdout, last"""
    txt = json.dumps(full)
    for d in (REPO, os.path.join(REPO, "gpurun_out")):
        try:
            if os.path.isdir(d):
                with open(os.path.join(d, "bench_full.json"), "w") as f:
                    f.write(txt + "\n")
        except OSError:
            pass
    print("[bench] full object:", txt, file=sys.stderr, flush=True)
    sys.stdout.flush()
    print(compact_line(full), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--reads", type=int, default=10_000_000, help="reads per GPU per step")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="target wall time of the all-cores CPU baseline run")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-host-fed", action="store_true")
    ap.add_argument("--no-cli", action="store_true")
    ap.add_argument("--host-fed-steps", type=int, default=0, help="0 = as many as run for about --host-fed-seconds")
    ap.add_argument("--host-fed-seconds", type=float, default=5.0)
    ap.add_argument("--no-legs", action="store_true", help="skip the robustness / thresholds / mixed legs")
    ap.add_argument("--leg-steps", type=int, default=10)
    ap.add_argument("--mixed-reads", type=int, default=8_000_000)
    ap.add_argument("--mixed-cli-reads", type=int, default=1_000_000)
    ap.add_argument("--cli-reads", type=int, default=10_000_000)
    ap.add_argument("--cli-bam-level", type=int, default=-2, help="-2 = structural BGZF (include/groot_host.h), -1..9 = zlib")
    ap.add_argument("--no-align", action="store_true", help="diagnostic: --noAlign mode (weights only, no BAM records)")
    ap.add_argument("--background", type=float, default=0.0,
                    help="diagnostic: fraction of reads replaced by uniform random ACGT (metagenome-like input, SURVEY 8d)")
    args = ap.parse_args()

    import torch

    if int(os.environ.get("LOCAL_RANK", "0")) == 0:
        # the native libraries normally travel prebuilt; (re)build whatever is missing or stale (logs go to stderr)
        import __graft_entry__ as entry

        entry.build_host()
        entry.build_hip()
        if args.gpus == 1:
            if not args.no_cli:
                entry.build_cli()
            if not args.no_cpu:
                entry.build_oracle()
    from groot_amd import device, synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # test hook (tests/test_cli.py): all ranks on GPU 0 over gloo, to run the N>1 code path on a one-GPU box
    one_gpu_test = os.environ.get("GROOT_BENCH_TEST_SAME_DEVICE") == "1"
    if one_gpu_test:
        local_rank = 0
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_gpu_test:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    if rank == 0:
        index, index_path = load_index()
    if dist is not None:
        dist.barrier()
    if rank != 0:
        index, index_path = load_index()

    # ---- synthetic reads of this rank's shard, generated straight into HBM ----
    cat, off, lens = synth.reference_sequences(index)
    cat_t, off_t, lens_t = (torch.from_numpy(x).to(dev) for x in (cat, off, lens))
    R = args.reads
    chunks, CH = [], 1_000_000
    for c0 in range(0, R, CH):
        n = min(CH, R - c0)
        p, _, _ = synth.reads_torch(cat_t, off_t, lens_t, n, READ_LEN, first=rank * R + c0)
        chunks.append(p[: n * READ_LEN])
    d_seq = torch.zeros(R * READ_LEN + 64, dtype=torch.uint8, device=dev)
    d_seq[: R * READ_LEN] = torch.cat(chunks)
    del chunks
    if args.background > 0:
        g = torch.Generator(device=dev)
        g.manual_seed(0x67726F6F74 + rank)
        rows = d_seq[: R * READ_LEN].view(R, READ_LEN)
        for c0 in range(0, R, CH):
            n = min(CH, R - c0)
            bg = torch.rand(n, generator=g, device=dev) < args.background
            rnd = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)[torch.randint(0, 4, (n, READ_LEN), generator=g, device=dev)]
            rows[c0:c0 + n] = torch.where(bg[:, None], rnd, rows[c0:c0 + n])
    d_off = torch.arange(0, R + 1, dtype=torch.int64, device=dev) * READ_LEN
    torch.cuda.synchronize()

    # `value` is configs[2] THROUGH THE KERNELS north_star names: the ctx is opened without the memo of groot_hip_open (memo_budget_mb = off), so
    # every read is hashed (sketch_sig_kernel; what it cannot decide: the full-width list pass), looked up and walked through its graph
    # (align_kernel) -- khf.go:35-55, lshe.go:153-175, alignment.go:13-254.  The memo tier (the library's default ctx) is a leg: memo_tier.
    al = device.Aligner(index, device=local_rank, max_batch_reads=R, max_read_len=256, max_batch_bases=R * READ_LEN + 64,
                        no_align=args.no_align, results_on_device=True, pipeline_depth=2, memo_budget_mb=device.MEMO_OFF)
    stream = torch.cuda.current_stream(dev)
    al.set_stream(stream.cuda_stream)
    al.set_profiling(True)
    # the call-count table of this workload has ONE row (kmerCount 70): it lives in a torch tensor so that RCCL can sum it
    # across the ranks in place -- 1.3 MB, the whole multi-GPU exchange
    _, n_w = al.attempts_shape()
    q_row = READ_LEN - index.view.kmer_size + 1
    d_att = torch.zeros(n_w, dtype=torch.int32, device=dev)
    al.attempts_layout([q_row], d_att.data_ptr())

    def step():
        al.submit_device(d_seq.data_ptr(), d_off.data_ptr(), R, first_read_id=0, max_len=READ_LEN)
        return al.wait()

    for _ in range(args.warmup):
        counts = step()
    d_att.zero_()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    stage_sum = {}
    open_stats = al.open_stats()
    # the ctx is a pipeline (submit / collect-the-oldest): the next step is enqueued while the GPU works on this one, so
    # the host's launch latency is not part of a step; every step is waited for and its counters are read
    pending = 0
    for i in range(args.steps + 1):
        if i < args.steps:
            al.submit_device(d_seq.data_ptr(), d_off.data_ptr(), R, first_read_id=0, max_len=READ_LEN)
            pending += 1
        if pending == 2 or (i == args.steps and pending):
            counts = al.wait()
            pending -= 1
            for k_, v_ in al.stage_ms().items():
                stage_sum[k_] = stage_sum.get(k_, 0.0) + v_
    while pending:
        counts = al.wait()
        pending -= 1
    torch.cuda.synchronize()
    t_steps = time.perf_counter() - t0          # this rank's K steps (before the exchange)
    if dist is not None:
        dist.all_reduce(d_att)  # per-(kmerCount, window) IncrementSubPath counts: the only exchange
    torch.cuda.synchronize()
    t_reduce = time.perf_counter() - t0 - t_steps
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    per_rank = None
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # every rank's own step time and what the all-reduce took there, so that a scaling run can be read rank by rank
        mine = torch.tensor([t_steps / args.steps * 1e3, t_reduce * 1e3], dtype=torch.float64, device=dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [[float(x[0].item()), float(x[1].item())] for x in allr]
        log("rank %d: %.3f ms per step, all-reduce of the call counts %.3f ms" % (rank, t_steps / args.steps * 1e3, t_reduce * 1e3))

    if rank == 0:
        total_reads = world * R * args.steps
        value = total_reads / dt / 1e6
        stage = {k_: v_ / args.steps for k_, v_ in stage_sum.items()}
        pw = index.view.path_words
        # per-kernel roofline of the step: live HIP-event durations (groot_stage_ms, taken on the streams the kernels run on), SURVEY 8d's
        # algorithmic bytes for the reads each kernel handles; HBM traffic and VALU instruction counts from the committed rocprofv3 --pmc passes
        blocks = kernel_blocks("c2_nomemo", stage, counts, R, READ_LEN, pw)
        # the dominant kernel = the longest one in the committed trace of this workload (profiles/r05_kernel_stats.csv: align_kernel 3.04 ms, 56 % of the
        # kernel time); the signature kernel's live event pair also counts the time its workgroups wait for CUs that the persistent align grid of the batch
        # before holds (3.4-3.8 ms live, 1.41 in the trace, 1.25 alone), so its live figure is not a kernel duration
        dom = max(blocks, key=lambda k_: blocks[k_].get("pmc_kernel_ms") or blocks[k_]["kernel_ms"])
        b = blocks[dom]
        step_bytes = R * (READ_LEN + 4) + 4 * R + 8 * counts["seeds"] + (20 + 8 * pw) * counts["travs"]     # SURVEY 8d, full pipeline
        step_ms = dt / args.steps * 1e3
        # VALU issue ceiling of the two hashing / walking kernels together: wave-instructions of one batch x 4 cycles over 1024 SIMDs at 2.4 GHz,
        # against the step's duration (they run side by side on two streams): the ceiling SURVEY 7 predicted would bind before HBM does
        valu_insts = sum((pmc_of("c2_nomemo", k_.split("<")[0]) or {}).get("per_launch", {}).get("SQ_INSTS_VALU", 0.0) for k_ in ("sketch_sig_kernel", "align_kernel"))
        valu_floor_ms = 4.0 * valu_insts / (1024 * 2.4e9) * 1e3 if valu_insts else None
        line = {
            "metric": "Mreads/s aligned", "value": value, "unit": "Mreads/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": step_ms, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": "configs[2] memo off: 10M x 100bp reads, arg-annot.90 k31 s21 w100 t0.99; every read hashed, looked up, walked",
                       "reads_per_gpu_per_step": R, "read_len": READ_LEN, **({"background_fraction": args.background} if args.background > 0 else {}), "parallelism": f"reads sharded x{world}, index replicated",
                       "residency": "inputs and records resident in HBM, two batches in flight (host_fed_mreads / cli_e2e_mreads: PCIe- and host-inclusive)",
                       "memo": "off for value (memo_budget_mb = GROOT_MEMO_OFF); the memo tier is roofline.memo_mreads",
                       "full_sketch_reads": counts["full_sketch_reads"], "walked_reads": counts["walked_reads"], "mapped": counts["mapped"],
                       "alignments": counts["alignments"], "travs": counts["travs"], "seeds": counts["seeds"],
                       "per_step_counts": counts, "stage_ms": stage, "open": open_stats},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": b.get("achieved"), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": b.get("frac"), "traffic": b.get("traffic"), "traffic_source": b.get("traffic_source"), "pmc_commit": pmc_commit(),
                         "kernel_selected_by": "longest avg duration in the committed trace (kernel_trace_ms); kernel_ms is live",
                         "bytes_per_launch": b["bytes_per_launch"], "kernel_ms": b["kernel_ms"], "kernel_trace_ms": b.get("pmc_kernel_ms"),
                         "sig_kernel_trace_ms": blocks["sketch_sig_kernel"].get("pmc_kernel_ms"),
                         "valu_issue_frac": b.get("valu_issue"), "wait_frac": b.get("wait_frac"),
                         "kernel_path_mreads": value, "kernel_path_ms_per_step": step_ms,
                         "sig_kernel_ms": stage.get("first_seed_kernel"), "list_pass_ms": stage.get("list_pass"), "sort_ms": stage.get("schedule"),
                         "align_kernel_ms": stage.get("align"), "order_ms": stage.get("sort"),
                         "sig_kernel_frac": blocks["sketch_sig_kernel"].get("frac"), "align_kernel_frac": blocks["align_kernel"].get("frac"),
                         "sig_kernel_valu_issue": blocks["sketch_sig_kernel"].get("valu_issue"), "align_kernel_valu_issue": blocks["align_kernel"].get("valu_issue"),
                         "align_kernel_traffic": blocks["align_kernel"].get("traffic"), "sig_kernel_traffic": blocks["sketch_sig_kernel"].get("traffic"),
                         "whole_step_bytes": step_bytes, "whole_step_frac": step_bytes / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "valu_wave_insts_per_batch": valu_insts or None, "valu_issue_floor_ms": valu_floor_ms,
                         "valu_issue_frac_of_step": (valu_floor_ms / step_ms) if valu_floor_ms else None,
                         "note": "integer hashing + dependent graph walk: VALU issue binds before HBM does (valu_issue_*); frac = algorithmic bytes / live kernel time / 8 TB/s; kernel = the longest one of the committed trace (kernel_trace_ms); sig_kernel_ms (live) includes waiting for CUs held by the align grid (sig_kernel_trace_ms)",
                         "kernels": blocks},
        }
        if per_rank is not None:
            line["per_rank"] = {"ms_per_step": [x[0] for x in per_rank], "allreduce_ms": [x[1] for x in per_rank]}
            line["roofline"]["allreduce_ms_max"] = max(x[1] for x in per_rank)
            line["roofline"]["rank_ms_per_step_max"] = max(x[0] for x in per_rank)
            line["roofline"]["rank_ms_per_step_min"] = min(x[0] for x in per_rank)
        rf = line["roofline"]
        if world == 1:
            if not args.no_legs and args.background == 0:
                # the same ctx (memo off) on reads with 1 % substitutions
                try:
                    g = torch.Generator(device=dev)
                    g.manual_seed(0x67726F6F74)
                    err = substituted(d_seq, R, 0.01, g)
                    v, ms, c = resident_rate(al, err.data_ptr(), d_off.data_ptr(), R, READ_LEN, args.leg_steps, 2)
                    del err
                    line["kernel_path"] = {"what": "the `value` ctx (memo off) on other inputs", "substitutions_1pct": {
                        "value": v, "unit": "Mreads/s", "stage_ms": ms, "full_sketch_reads": c["full_sketch_reads"], "walked_reads": c["walked_reads"],
                        "mapped": c["mapped"], "kernels": kernel_blocks("sub1_nomemo", ms, c, R, READ_LEN, pw)}}
                    rf["sub1_nomemo_mreads"] = v
                    rf["sub1_nomemo_align_ms"] = ms.get("align")
                except Exception as e:   # the headline must still print
                    line["kernel_path"] = {"error": repr(e)}
            al.close()
            al = None
            if not args.no_legs and args.background == 0:
                # the `value` workload with the align stage's first pass in front of align_kernel (GROOT_LEAN=1, kernels_lean.hpp; opt-in: DESIGN.md section 3)
                try:
                    os.environ["GROOT_LEAN"] = "1"
                    all_ = device.Aligner(index, device=local_rank, max_batch_reads=R, max_read_len=256, max_batch_bases=R * READ_LEN + 64,
                                          results_on_device=True, pipeline_depth=2, memo_budget_mb=device.MEMO_OFF)
                    all_.set_profiling(True)
                    v, ms, c = resident_rate(all_, d_seq.data_ptr(), d_off.data_ptr(), R, READ_LEN, args.leg_steps, 3)
                    all_.close()
                    line["lean_first"] = {"value": v, "unit": "Mreads/s", "stage_ms": ms, "lean_reads": c.get("lean_reads"), "walked_reads": c["walked_reads"],
                                          "alignments": c["alignments"], "what": "configs[2], memo off, GROOT_LEAN=1: align_lean_kernel + compaction in front of align_kernel"}
                    rf["lean_first_mreads"] = v
                    rf["lean_pass_ms"] = ms.get("lean_pass")
                    rf["lean_first_align_ms"] = ms.get("align")
                    rf["lean_reads_frac"] = (c.get("lean_reads") or 0) / max(1, c["walked_reads"])
                    try:   # VALU wave-instructions of that batch's hashing + both align passes, from the committed PMC passes taken with GROOT_LEAN=1
                        lp = json.load(open(os.path.join(REPO, "profiles", "r06_lean_pmc.json"))).get("c2_nomemo", {})
                        rf["lean_valu_wave_insts_per_batch"] = sum(lp.get(k_, {}).get("per_launch", {}).get("SQ_INSTS_VALU", 0.0) for k_ in ("sketch_sig_kernel", "align_lean_kernel", "align_kernel")) or None
                    except Exception:
                        pass
                except Exception as e:
                    line["lean_first"] = {"error": repr(e)}
                finally:
                    os.environ.pop("GROOT_LEAN", None)
            if not args.no_legs and args.background == 0:
                # the library's default ctx: the memo of groot_hip_open answers reads that equal an indexed WindowSize-mer
                try:
                    alm = device.Aligner(index, device=local_rank, max_batch_reads=R, max_read_len=256, max_batch_bases=R * READ_LEN + 64,
                                         results_on_device=True, pipeline_depth=2)
                    alm.set_profiling(True)
                    v, ms, c = resident_rate(alm, d_seq.data_ptr(), d_off.data_ptr(), R, READ_LEN, args.leg_steps, 3)
                    line["memo_tier"] = {"value": v, "unit": "Mreads/s", "stage_ms": ms, "full_sketch_reads": c["full_sketch_reads"], "walked_reads": c["walked_reads"],
                                         "mapped": c["mapped"], "alignments": c["alignments"], "open": alm.open_stats(),
                                         "what": "configs[2] on the default ctx: error-free window-sized reads are the memo's keys -- a table look-up, not the kernels"}
                    rf["memo_mreads"] = v
                    line["robustness"] = robustness(alm, index, d_seq, d_off, R, args.leg_steps)
                    rf["sub1_mreads"] = line["robustness"]["substitutions_1pct"]["value"]
                    rf["sub1_align_ms"] = line["robustness"]["substitutions_1pct"]["stage_ms"].get("align")
                    rf["background99_mreads"] = line["robustness"]["background_99pct"]["value"]
                    alm.close()
                except Exception as e:
                    line["memo_tier"] = {"error": repr(e)}
            if not args.no_legs:
                try:
                    line["thresholds"] = threshold_sweep(index, d_seq, d_off, R, args.leg_steps, local_rank)
                except Exception as e:
                    line["thresholds"] = {"error": repr(e)}
                try:
                    line["mixed"] = mixed_leg(local_rank, args.mixed_reads, args.leg_steps, args.mixed_cli_reads, args.cli_bam_level)
                    mk = line["mixed"]["kernels"]
                    rf["mixed99_mreads"] = mk["t=0.99"]["value"]
                    rf["mixed90_mreads"] = mk["t=0.90"]["value"]
                    rf["mixed99_list_pass_ms"] = mk["t=0.99"]["stage_ms"].get("list_pass")
                    rf["mixed99_align_ms"] = mk["t=0.99"]["stage_ms"].get("align")
                    if "t=0.99, batches of 2 M reads" in mk:
                        rf["mixed99_2m_mreads"] = mk["t=0.99, batches of 2 M reads"]["value"]
                    if "value" in line["mixed"].get("cli_gzip", {}):
                        rf["cli_gzip_mreads"] = line["mixed"]["cli_gzip"]["value"]
                        rf["cli_gzip_stream_mreads"] = line["mixed"]["cli_gzip"]["stream_value"]
                except Exception as e:
                    line["mixed"] = {"error": repr(e)}
            if not args.no_host_fed:
                try:
                    hf_steps = args.host_fed_steps or max(40, int(args.host_fed_seconds / 5e-3))   # (a host-fed batch of 10 M reads takes ~5 ms: PCIe)
                    hf, _ = host_fed(index, d_seq, R, hf_steps)
                    hf["frac_of_resident"] = hf["value"] / value
                    line["host_fed"] = hf
                    rf["host_fed_mreads"] = hf["value"]
                except Exception as e:   # the headline must still print
                    line["host_fed"] = {"error": repr(e)}
            if not args.no_cli:
                try:
                    line["cli_e2e"] = cli_e2e(index, d_seq, min(args.cli_reads, R), args.cli_bam_level)
                    rf["cli_e2e_mreads"] = line["cli_e2e"].get("value")
                    rf["cli_stream_mreads"] = line["cli_e2e"].get("stream_value")
                except Exception as e:
                    line["cli_e2e"] = {"error": repr(e)}
            if not args.no_cpu:
                try:
                    line["cpu_baseline"] = cpu_baseline(index_path, args.cpu_seconds)
                except Exception as e:
                    line["cpu_baseline"] = {"error": repr(e)}
        emit(line)
    if al is not None:
        al.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
