#!/usr/bin/env python3
"""bench.py -- Mreads/s aligned on MI355X for BASELINE.json's workload.

A step = one pass of the whole `groot align` hot path (sketch -> LSH-Ensemble seed -> graph DFS
alignment -> canonical ordering of the traversal records) over one batch of synthetic 100 bp
reads that is already resident in HBM.  Reads shard across GPUs (one process per GPU, index
replicated); the only exchange is one RCCL all-reduce of the IncrementSubPath call counts after the
last step (inside the timed region).

  python bench.py --gpus 1 --steps 5 --warmup 2
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import tarfile
import tempfile
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

READ_LEN = 100
HBM_PEAK_GBS = 8000.0  # spec peak, /opt/skills/guides/MI355X_MICROARCH.md (6.29 TB/s measured copy rate)


def load_index():
    """arg-annot.90, k=31 s=21 w=100 x=8 y=4 (cmd/index.go:45-49 defaults), cached under build/"""
    from groot_amd import host

    cache = os.path.join(REPO, "build", "arg-annot.90.k31.s21.w100.gidx")
    if os.path.exists(cache):
        try:
            return host.Index.load(cache)
        except Exception:
            pass
    with tempfile.TemporaryDirectory() as td:
        with tarfile.open(os.path.join(REPO, "tests", "golden", "data", "arg-annot.90.tar.gz")) as tf:
            members = [m for m in tf.getmembers() if os.path.basename(m.name).startswith("cluster") and m.name.endswith(".msa")]
            tf.extractall(td, members=members)
        index = host.Index.from_msa_dir(os.path.join(td, "arg-annot.90"))
    try:
        tmp = cache + ".%d.tmp" % os.getpid()
        index.save(tmp)
        os.replace(tmp, cache)
    except Exception:
        pass
    return index


def cpu_baseline(index, n_sample):
    """the oracle (single-thread C port of the reference path) timed on this box's host cores"""
    from groot_amd import synth
    from oracle import oracle_py as O

    cat, off, lens = synth.reference_sequences(index)
    seq, seq_off, _ = synth.reads_np(cat, off, lens, n_sample, READ_LEN)
    run = O.Run(index, 0.99)
    t0 = time.perf_counter()
    run.batch(seq, seq_off)
    dt = time.perf_counter() - t0
    one = {"value": n_sample / dt / 1e6, "unit": "Mreads/s", "cores": 1, "kind": "port",
           "sample": f"first {n_sample} reads of the same synthetic stream, oracle/groot_oracle.c single thread, {dt:.1f} s"}
    del run
    # the whole host, as the reference's goroutine path would use it (`groot align -p <cores>`): one oracle instance per
    # hardware thread, each on its own slice of the stream (reads are independent; ctypes releases the GIL)
    import threading

    cores = min(os.cpu_count() or 1, 256)
    per = max(20_000, min(100_000, n_sample // 4))
    runs = [O.Run(index, 0.99) for _ in range(cores)]
    slices = []
    for i in range(cores):
        lo = (i * per) % max(1, n_sample - per + 1)
        slices.append((np.ascontiguousarray(seq[lo * READ_LEN:(lo + per) * READ_LEN]), np.ascontiguousarray(seq_off[lo:lo + per + 1] - seq_off[lo])))
    th = [threading.Thread(target=lambda r=r, sl=sl: r.batch(sl[0], sl[1])) for r, sl in zip(runs, slices)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    dta = time.perf_counter() - t0
    allc = {"value": cores * per / dta / 1e6, "unit": "Mreads/s", "cores": cores, "kind": "port",
            "sample": f"{cores} oracle instances x {per} reads of the same synthetic stream in parallel, {dta:.1f} s"}
    return one, allc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads", type=int, default=10_000_000, help="reads per GPU per step")
    ap.add_argument("--cpu-sample", type=int, default=1_000_000)
    ap.add_argument("--no-cpu", action="store_true")
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
        if args.gpus == 1 and not args.no_cpu:
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
        index = load_index()
    if dist is not None:
        dist.barrier()
    if rank != 0:
        index = load_index()

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

    al = device.Aligner(index, device=local_rank, max_batch_reads=R, max_read_len=256, max_batch_bases=R * READ_LEN + 64,
                        no_align=args.no_align)
    stream = torch.cuda.current_stream(dev)
    al.set_stream(stream.cuda_stream)
    al.set_profiling(True)
    n_q, n_w = al.attempts_shape()
    d_att = torch.zeros(n_q * n_w, dtype=torch.int32, device=dev)
    al.attempts_bind(d_att.data_ptr(), d_att.numel())

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
    k_ms, a_ms, s_ms, g_ms = [], [], [], []
    for _ in range(args.steps):
        counts = step()
        ms = al.stage_ms()
        k_ms.append(ms["sketch_seed"]); a_ms.append(ms["align"]); s_ms.append(ms["sort"]); g_ms.append(ms["schedule"])
    if dist is not None:
        dist.all_reduce(d_att)  # per-(kmerCount, window) IncrementSubPath counts: the only exchange
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        total_reads = world * R * args.steps
        value = total_reads / dt / 1e6
        seed_ms, align_ms, order_ms = float(np.mean(k_ms)), float(np.mean(a_ms)), float(np.mean(s_ms))
        # algorithmic bytes per launch (DESIGN.md "Measurement"): what each kernel must read / write once
        #   sketch_seed: bases + u64 offset in; u32 seed count + u32 per seed out
        #   align      : bases + u64 offset + u32 seed count + u32 per seed in; u32 traversal count per read,
        #                44 B per traversal record (20 B header + 3x8 B path set) and one u32 call count per seed tried out
        pw = index.view.path_words
        seed_bytes = R * (READ_LEN + 8 + 4) + 4 * counts["seeds"]
        align_bytes = R * (READ_LEN + 8 + 4 + 4) + 4 * counts["seeds"] + (20 + 8 * pw) * counts["travs"] + 4 * counts["seeds"]
        kernels = {"sketch_seed_kernel<21,4,false,6>": (seed_ms, seed_bytes), "align_kernel<3,true>": (align_ms, align_bytes)}
        dom = max(kernels, key=lambda k: kernels[k][0])
        dom_ms, dom_bytes = kernels[dom]
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9
        traffic, valu = None, None
        pmc = os.path.join(REPO, "profiles", "r01_pmc.json")
        if os.path.exists(pmc):
            try:
                rec = json.load(open(pmc)).get(dom.split("<")[0], {})
                traffic = rec.get("hbm_bytes_per_launch")
                # the ceiling that actually binds: VALU issue = wave-instructions x 4 cycles / (1024 SIMDs x 2.4 GHz)
                insts = rec.get("per_launch", {}).get("SQ_INSTS_VALU")
                if insts:
                    issue_ms = insts * 4 / (1024 * 2.4e9) * 1e3
                    valu = {"wave_insts_per_launch": insts, "issue_ms": issue_ms, "frac_of_kernel": issue_ms / dom_ms,
                            "source": "profiles/r01_pmc.json (rocprofv3 --pmc SQ_INSTS_VALU of the same command)"}
            except Exception:
                traffic, valu = None, None
        line = {
            "metric": "Mreads/s aligned", "value": value, "unit": "Mreads/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": "configs[2]: full pipeline incl. on-GPU graph-traversal alignment, 100 bp error-free reads sampled from arg-annot.90, index k=31 s=21 w=100 x=8 y=4, t=0.99",
                       "reads_per_gpu_per_step": R, "read_len": READ_LEN, **({"background_fraction": args.background} if args.background > 0 else {}), "parallelism": f"reads sharded x{world}, index replicated",
                       "per_step_counts": counts,
                       "stage_ms": {"sketch_seed": seed_ms, "schedule": float(np.mean(g_ms)), "align": align_ms, "order": order_ms}},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "bytes_per_launch": dom_bytes, "kernel_ms": dom_ms,
                         "note": "integer hashing / graph walking: the binding ceiling is VALU issue, not HBM (DESIGN.md)", "valu_issue": valu,
                         "other": {k: {"kernel_ms": v[0], "bytes_per_launch": v[1], "achieved": v[1] / (v[0] * 1e-3) / 1e9}
                                   for k, v in kernels.items() if k != dom}},
        }
        if world == 1 and not args.no_cpu:
            line["cpu_baseline"], line["cpu_baseline_all_cores"] = cpu_baseline(index, args.cpu_sample)
        print(json.dumps(line), flush=True)
    al.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
