"""Seed-stage time of the full-width kernel for larger sketch sizes (`groot index -s`): 2 M synthetic 100 bp reads per size (GPU box)."""
import os, sys, tarfile, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from groot_amd import device, host, synth
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sizes = [int(x) for x in sys.argv[1:]] or [21, 32, 48, 64]
dev = torch.device("cuda", 0)
with tempfile.TemporaryDirectory() as td:
    with tarfile.open(os.path.join(REPO, "tests", "golden", "data", "arg-annot.90.tar.gz")) as tf:
        members = [m for m in tf.getmembers() if os.path.basename(m.name).startswith("cluster") and m.name.endswith(".msa")]
        tf.extractall(td, members=members)
    files = host.msa_files(os.path.join(td, "arg-annot.90"))
    for s in sizes:
        index = host.Index.from_msa_files(files, host.index_params(s=s))
        cat, off, lens = synth.reference_sequences(index)
        cat_t, off_t, lens_t = (torch.from_numpy(x).to(dev) for x in (cat, off, lens))
        R, L = 2_000_000, 100
        p, _, _ = synth.reads_torch(cat_t, off_t, lens_t, R, L)
        d_seq = torch.zeros(R * L + 64, dtype=torch.uint8, device=dev); d_seq[:R * L] = p[:R * L]
        d_off = torch.arange(0, R + 1, dtype=torch.int64, device=dev) * L
        os.environ["GROOT_NO_OUTCOME_TABLE"] = "1"
        al = device.Aligner(index, device=0, max_batch_reads=R, max_read_len=256, max_batch_bases=R * L + 64, results_on_device=True)
        al.set_profiling(True)
        for _ in range(3):
            al.submit_device(d_seq.data_ptr(), d_off.data_ptr(), R, max_len=L); c = al.wait()
        ms = al.stage_ms()
        print({"s": s, "sketch_seed_ms_per_2M": round(ms["sketch_seed"], 3), "Mreads_s_seed_stage": round(R / ms["sketch_seed"] / 1e3, 1), "mapped": c["mapped"]}, flush=True)
        al.close()
