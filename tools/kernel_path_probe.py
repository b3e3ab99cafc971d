#!/usr/bin/env python3
"""One workload of bench.py on its own, for rocprofv3 (tools/profile_r04.sh): `python tools/kernel_path_probe.py WORKLOAD [steps] [reads]`.

  headline      configs[2], the product's default ctx (memo on): what bench.py's `value` times
  c2_nomemo     configs[2] with GROOT_NO_OUTCOME_TABLE + GROOT_NO_TEXT_TABLE: every read hashed, looked up, walked (bench.py kernel_path.error_free)
  sub1_nomemo   the same ctx on reads with 1 % substitutions (kernel_path.substitutions_1pct)
  sub1          1 % substitutions on the default ctx (robustness.substitutions_1pct)
  mixed99/90    resfinder.90, reads of 75..150 bases, t = 0.99 / 0.90 (mixed.kernels)

Prints one JSON line: value, mean stage ms, counts.  The first launch of the measured loop is preceded by a marker kernel-free pause
and announced on stderr so that the profile post-processing can cut groot_hip_open's launches (the probe prints the dispatch count
it cannot know; the script cuts by time instead: see tools/profile_r04.sh)."""
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import torch  # noqa: E402

import bench  # noqa: E402
from groot_amd import device, synth  # noqa: E402


def main():
    wl = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    mixed = wl.startswith("mixed")
    R = int(sys.argv[3]) if len(sys.argv) > 3 else (8_000_000 if mixed else 10_000_000)
    index, _ = bench.load_index("resfinder.90" if mixed else "arg-annot.90")
    cat, off, lens = synth.reference_sequences(index)
    cat_t, off_t, lens_t = (torch.from_numpy(x).to(dev) for x in (cat, off, lens))
    threshold = 0.99
    if mixed:
        d_seq, d_off, _ = synth.reads_torch_mixed(cat_t, off_t, lens_t, R, 150, 75)
        max_len, total = 150, int(d_off[-1].item())
        threshold = int(wl[5:]) / 100.0
    else:
        chunks = []
        for c0 in range(0, R, 1_000_000):
            n = min(1_000_000, R - c0)
            p, _, _ = synth.reads_torch(cat_t, off_t, lens_t, n, bench.READ_LEN, first=c0)
            chunks.append(p[: n * bench.READ_LEN])
        d_seq = torch.zeros(R * bench.READ_LEN + 64, dtype=torch.uint8, device=dev)
        d_seq[: R * bench.READ_LEN] = torch.cat(chunks)
        del chunks
        d_off = torch.arange(0, R + 1, dtype=torch.int64, device=dev) * bench.READ_LEN
        max_len, total = bench.READ_LEN, R * bench.READ_LEN
        if wl.startswith("sub1"):
            g = torch.Generator(device=dev)
            g.manual_seed(0x67726F6F74)
            d_seq = bench.substituted(d_seq, R, 0.01, g)
    if wl.endswith("_nomemo"):
        os.environ["GROOT_NO_OUTCOME_TABLE"] = "1"
        os.environ["GROOT_NO_TEXT_TABLE"] = "1"
    al = device.Aligner(index, device=0, threshold=threshold, max_batch_reads=R, max_read_len=256, max_batch_bases=total + 64, results_on_device=True, pipeline_depth=2)
    al.set_profiling(True)
    torch.cuda.synchronize()
    time.sleep(0.2)
    print("[probe] loop starts", time.time_ns(), file=sys.stderr, flush=True)
    t_loop = time.clock_gettime_ns(time.CLOCK_MONOTONIC)
    if os.environ.get("PROBE_SERIAL") == "1":               # one batch at a time: the stages run one after the other, nothing overlaps
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            al.submit_device(d_seq.data_ptr(), d_off.data_ptr(), R, first_read_id=0, max_len=max_len, mixed=mixed)
            c = al.wait()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        v, ms = R * steps / dt / 1e6, al.stage_ms()
    else:
        v, ms, c = bench.resident_rate(al, d_seq.data_ptr(), d_off.data_ptr(), R, max_len, steps, 2, mixed=mixed)
    al.close()
    print(json.dumps({"workload": wl, "reads": R, "steps": steps, "value": v, "stage_ms": ms, "counts": c, "loop_start_monotonic_ns": t_loop}), flush=True)


if __name__ == "__main__":
    main()
