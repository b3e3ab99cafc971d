"""The flag-compatible command line (`groot-hip index|align`): cmd/index.go, cmd/align.go, and the e2e flow of
testing/run_travis_tests.sh (index -w 150 -k 31 -s 20; align -t 0.99 on bla-b7-150bp-5x.fq)."""
import os
import re
import subprocess

import numpy as np
import pytest

from bamread import read_bam
from conftest import DATA, REPO, read_fastq
from groot_amd import device, host
from oracle import oracle_py as O


@pytest.fixture(scope="module")
def cli(hip_lib):
    import __graft_entry__ as g

    return g.build_cli()


def run(cmd, **kw):
    return subprocess.run(cmd, cwd=REPO, capture_output=True, timeout=600, **kw)


def test_index_subcommand_and_align_without_gpu(cli, msa_dir, tmp_path):
    idx_dir = str(tmp_path / "idx")
    log = str(tmp_path / "index.log")
    r = run([cli, "index", "-m", msa_dir, "-i", idx_dir, "--log", log, "-p", "4"])
    assert r.returncode == 0, r.stderr
    assert not os.path.exists(os.path.join(idx_dir, "groot.gg"))      # the reference's gob files are opt-in (unpinned against Go)
    r = run([cli, "index", "-m", msa_dir, "-i", idx_dir, "--log", log, "-p", "4", "--writeGob"])
    assert r.returncode == 0, r.stderr
    text = open(log).read()
    for line in ("i am groot (version 1.1.2)", "starting the index subcommand", "\tk-mer size: 31", "\tsketch size: 21",
                 "\tgraph window size: 100", "\tnumber of groot graphs built: 583", "\t\tgraphs sketched: 583"):
        assert line in text, line
    again = host.Index.load(os.path.join(idx_dir, "groot.gidx"))
    assert (again.view.n_graphs, again.view.n_paths) == (583, 1749)
    # the directory also holds the reference's own index files (cmd/index.go:130-131), equal to the flat index
    os.rename(os.path.join(idx_dir, "groot.gidx"), str(tmp_path / "moved.gidx"))
    from_gob = host.Index.load_gob(idx_dir)
    for k, arr in again.arrays.items():
        assert np.array_equal(arr, from_gob.arrays[k]), k
    os.rename(str(tmp_path / "moved.gidx"), os.path.join(idx_dir, "groot.gidx"))
    # required flags / bad inputs (cmd/index.go:57-61,161-163; cmd/align.go:56-60)
    assert run([cli, "index", "-i", idx_dir]).returncode != 0
    assert run([cli, "index", "-m", msa_dir, "-i", idx_dir, "-k", "200", "-w", "100", "--log", log]).returncode != 0
    assert run([cli, "align", "-f", os.path.join(DATA, "bla-b7-150bp-5x.fq")]).returncode != 0
    if device.device_count() == 0:
        r = run([cli, "align", "-i", idx_dir, "-f", os.path.join(DATA, "bla-b7-150bp-5x.fq"), "--log", str(tmp_path / "a.log"),
                 "-g", str(tmp_path / "graphs")])
        assert r.returncode != 0 and b"no HIP device" in r.stderr     # never a CPU fallback


@pytest.mark.gpu
@pytest.mark.parametrize("memo", ["auto", "on"])
def test_align_subcommand_against_oracle(cli, argannot_index, perfect_reads, tmp_path, memo):
    """(--memo on: the reads are answered from the memo of groot_hip_open; auto leaves it off for an input this small)"""
    idx_dir = tmp_path / "idx"
    idx_dir.mkdir()
    argannot_index.save(str(idx_dir / "groot.gidx"))
    fq = os.path.join(DATA, "full-argannot-perfect-reads-small.fq.gz")
    log, graphs, bam = str(tmp_path / "align.log"), str(tmp_path / "graphs"), str(tmp_path / "out.bam")
    with open(bam, "wb") as out:
        r = subprocess.run([cli, "align", "-i", str(idx_dir), "-f", fq, "--log", log, "-g", graphs, "--batch", "300", "--memo", memo],
                           cwd=REPO, stdout=out, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr
    # oracle
    cat, off = O.pack_reads([x[1] for x in perfect_reads])
    orun = O.Run(argannot_index)
    orun.batch(cat, off)
    oal, oc = orun.alns(), orun.counts()
    text, refs, recs = read_bam(bam)
    assert len(recs) == len(oal) == oc["alignments"]
    assert [n for n, _ in refs] == [argannot_index.path_name(p) for p in range(argannot_index.view.n_paths)]
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    for a, rec in zip(oal, recs):          # records come out in read order, ids ascending: the canonical order
        name, s, q = perfect_reads[int(a["read_id"])]
        if a["rc"]:
            s, q = s.translate(comp)[::-1], q[::-1]
        assert (rec["name"], rec["ref_id"], rec["pos"]) == (name.decode(), int(a["ref_id"]), int(a["pos"]))
        assert rec["seq"].encode() == s and rec["qual"] == q and rec["cigar"] == "100M"
        assert rec["flag"] == (0x10 if a["rc"] else 0) | (0x100 if a["secondary"] else 0) and rec["mapq"] == 30
    logtxt = open(log).read()
    for line in (f"\tnumber of reads received from input: {oc['received']}", "\tmean read length: 100",
                 f"\ttotal number of unmapped reads: {oc['received'] - oc['mapped']}", f"\ttotal number of mapped reads: {oc['mapped']}",
                 f"\t\tmapped to multiple graphs: {oc['multimapped']}", f"\ttotal number of exact alignments: {oc['alignments']}"):
        assert line in logtxt, line
    # weighted graphs: KC tags = int(KmerFreq) of the canonical replay, pruned at the default coverage 1.0
    kf, kt = orun.weights(order=1)
    gk, pk, nr = orun.prune(kf, 1.0)
    assert f"\ttotal number of k-mers projected onto graphs: {int(kt.sum())}" in logtxt
    a = argannot_index.arrays
    files = sorted(os.listdir(graphs))
    expect = []
    for g in np.flatnonzero(gk):
        nodes = range(int(a["graph_node_off"][g]), int(a["graph_node_off"][g + 1]))
        if any(kf[n] > 0 and not nr[n] for n in nodes):
            expect.append(f"groot-graph-{g}.gfa")
    assert files == sorted(expect) and len(files) > 0
    g = int(re.match(r"groot-graph-(\d+)\.gfa", files[0]).group(1))
    kc = {}
    for line in open(os.path.join(graphs, files[0])):
        f = line.rstrip("\n").split("\t")
        if f[0] == "S":
            kc[int(f[1])] = int(f[4][5:])
    for n in range(int(a["graph_node_off"][g]), int(a["graph_node_off"][g + 1])):
        if not nr[n]:
            assert kc[int(a["node_seg_id"][n])] == int(kf[n])


@pytest.mark.gpu
def test_travis_e2e_flow(cli, msa_dir, tmp_path):
    """testing/run_travis_tests.sh:12-56: index -w 150 -k 31 -s 20, align -t 0.99 on the 150 bp B-7 reads.
    (`groot report` is outside the hot path; here: every alignment is on the Bla-B graph and the B-7 allele is
    the only allele every read aligns to, as the travis assertion implies.)"""
    idx_dir = str(tmp_path / "idx")
    r = run([cli, "index", "-m", msa_dir, "-i", idx_dir, "-w", "150", "-k", "31", "-s", "20", "--log", str(tmp_path / "i.log"), "-p", "8"])
    assert r.returncode == 0, r.stderr
    bam, graphs = str(tmp_path / "o.bam"), str(tmp_path / "g")
    r = run([cli, "align", "-i", idx_dir, "-f", os.path.join(DATA, "bla-b7-150bp-5x.fq"), "-t", "0.99", "--bam", bam, "-g", graphs,
             "--log", str(tmp_path / "a.log")])
    assert r.returncode == 0, r.stderr
    text, refs, recs = read_bam(bam)
    reads = read_fastq(os.path.join(DATA, "bla-b7-150bp-5x.fq"))
    assert recs and all("(Bla)B-" in x["ref"] for x in recs)
    # every simulated read aligns to the B-7 allele it came from (what `groot report` then turns into the one ARG)
    b7 = {x["name"] for x in recs if x["ref"] == "argannot~~~(Bla)B-7~~~AF189304:1-747"}
    assert b7 == {n.decode() for n, _, _ in reads}
    assert all(x["cigar"] == "150M" for x in recs)
    # run_travis_tests.sh:36-56: `groot report -c 0.97` must list exactly one gene, (Bla)B-7
    r = run([cli, "report", "--bamFile", bam, "-c", "0.97", "--log", str(tmp_path / "r.log")])
    assert r.returncode == 0, r.stderr
    lines = r.stdout.decode().splitlines()
    assert len(lines) == 1 and lines[0].split("\t")[0] == "argannot~~~(Bla)B-7~~~AF189304:1-747"
    # the same index as a directory of the reference's own files (groot.gg + groot.lshe, cmd/align.go:93-107)
    import gobenc

    gob_dir = str(tmp_path / "gobidx")
    gobenc.write_index_dir(host.Index.load(os.path.join(idx_dir, "groot.gidx")), gob_dir, shuffle_seed=3)
    bam2 = str(tmp_path / "o2.bam")
    r = run([cli, "align", "-i", gob_dir, "-f", os.path.join(DATA, "bla-b7-150bp-5x.fq"), "-t", "0.99", "--bam", bam2,
             "-g", str(tmp_path / "g2"), "--log", str(tmp_path / "a2.log")])
    assert r.returncode == 0, r.stderr
    text2, refs2, recs2 = read_bam(bam2)
    assert refs2 == refs and recs2 == recs


def _gfas(d):
    out = {}
    for f in sorted(os.listdir(d)):
        out[f] = [ln for ln in open(os.path.join(d, f)) if not ln.startswith("#")]     # the comment lines carry a timestamp
    return out


@pytest.mark.gpu
def test_reads_shard_over_several_contexts(cli, argannot_index, tmp_path):
    """`--gpus N`: batches go round the GPU contexts, records come back in input order, the call counts are summed
    (groot_hip_attempts_allreduce) before the graphs are weighted -- BAM records and GFAs equal the one-context run.  One GPU
    here, so the contexts share it (--ctxPerGpu): the same code path with the RCCL ring replaced by its same-device sum."""
    idx_dir = tmp_path / "idx"
    idx_dir.mkdir()
    argannot_index.save(str(idx_dir / "groot.gidx"))
    fqs = ",".join(os.path.join(DATA, f) for f in ("full-argannot-perfect-reads-small.fq.gz", "full-argannot-perfect-reads-small-variable-rl.fq.gz"))
    outs = []
    for tag, extra in (("one", []), ("three", ["--ctxPerGpu", "3", "--depth", "2"])):
        bam, graphs, stats = str(tmp_path / f"{tag}.bam"), str(tmp_path / f"g_{tag}"), str(tmp_path / f"{tag}.json")
        r = run([cli, "align", "-i", str(idx_dir), "-f", fqs, "--log", str(tmp_path / f"{tag}.log"), "-g", graphs, "--bam", bam, "--batch", "190",
                 "-p", "4", "-t", "0.97", "--bamLevel", "1", "--stats", stats] + extra)
        assert r.returncode == 0, r.stderr
        import json
        st = json.load(open(stats))
        assert st["gpu_contexts"] == (3 if extra else 1) and st["reads"] == 2000 and st["bam_bytes"] > 0
        outs.append((read_bam(bam), _gfas(graphs), open(str(tmp_path / f"{tag}.log")).read()))
    (t1, refs1, recs1), g1, log1 = outs[0]
    (t3, refs3, recs3), g3, log3 = outs[1]
    assert refs1 == refs3 and recs1 == recs3 and len(recs1) > 2000
    assert g1 == g3 and len(g1) > 0
    for key in ("total number of mapped reads", "mapped to multiple graphs", "total number of exact alignments", "total number of k-mers projected"):
        assert [ln.split(" ", 2)[2] for ln in log1.splitlines() if key in ln] == [ln.split(" ", 2)[2] for ln in log3.splitlines() if key in ln] != []


@pytest.mark.gpu
def test_reads_longer_than_the_context_was_opened_for(cli, argannot_index, tmp_path):
    """the reference has no read-length limit: a batch holding a longer read re-opens the GPU context (call counts carried
    over) instead of aborting the run"""
    idx_dir = tmp_path / "idx"
    idx_dir.mkdir()
    argannot_index.save(str(idx_dir / "groot.gidx"))
    short = read_fastq(os.path.join(DATA, "full-argannot-perfect-reads-small.fq.gz"))[:400]
    gene = argannot_index.path_sequence(5, 0)
    long_read = bytes(gene[:600]) if len(gene) >= 600 else bytes(gene)
    fq = tmp_path / "mixed.fq"
    with open(fq, "wb") as f:
        for i, (n, s, q) in enumerate(short):
            f.write(b"@" + n + b"\n" + s + b"\n+\n" + q + b"\n")
            if i == 250:
                f.write(b"@long\n" + long_read + b"\n+\n" + b"I" * len(long_read) + b"\n")
    res = []
    for tag, mrl in (("grow", "160"), ("big", "1024")):
        bam, log = str(tmp_path / f"{tag}.bam"), str(tmp_path / f"{tag}.log")
        r = run([cli, "align", "-i", str(idx_dir), "-f", str(fq), "--log", log, "-g", str(tmp_path / f"g_{tag}"), "--bam", bam, "--batch", "128",
                 "--maxReadLen", mrl, "-p", "2"])
        assert r.returncode == 0, r.stderr
        res.append((read_bam(bam)[2], _gfas(str(tmp_path / f"g_{tag}")), open(log).read()))
    assert "reopening the GPU context" in res[0][2] and "reopening the GPU context" not in res[1][2]
    assert res[0][0] == res[1][0] and res[0][1] == res[1][1] and len(res[0][0]) > 400
    # several contexts, only ONE of which meets the long read: the others follow before the call counts are summed,
    # instead of the run dying after the BAM is written
    bam, log = str(tmp_path / "multi.bam"), str(tmp_path / "multi.log")
    r = run([cli, "align", "-i", str(idx_dir), "-f", str(fq), "--log", log, "-g", str(tmp_path / "g_multi"), "--bam", bam, "--batch", "128",
             "--maxReadLen", "160", "-p", "2", "--ctxPerGpu", "3", "--depth", "2"])
    assert r.returncode == 0, r.stderr
    text = open(log).read()
    assert "reopening the GPU context" in text and "before the call counts are summed" in text
    assert read_bam(bam)[2] == res[1][0] and _gfas(str(tmp_path / "g_multi")) == res[1][1]


@pytest.mark.gpu
def test_bench_multi_rank_code_path(argannot_index, tmp_path):
    """bench.py as the driver launches it for N>1 (torch.distributed.run, one rank per GPU).  This box has one GPU, so both
    ranks share it and talk over gloo (GROOT_BENCH_TEST_SAME_DEVICE): the sharding, barriers, the all-reduce of the call-count
    table and the whole-job rate are the code the 8-GPU run executes."""
    import json
    import sys

    env = dict(os.environ, GROOT_BENCH_TEST_SAME_DEVICE="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29517", os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--reads", "300000"]
    r = subprocess.run(cmd, cwd=REPO, capture_output=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    last = r.stdout.decode().strip().splitlines()[-1]
    compact = json.loads(last)                       # what the driver parses: the LAST stdout line, under its 8 KB tail
    assert len(last) < 8000 and compact["n_gpus"] == 2 and compact["value"] > 0 and "rank1_ms_per_step" in compact["roofline"]
    assert all(not isinstance(v2, (dict, list)) for v in compact.values() if isinstance(v, dict) for v2 in v.values())
    line = json.load(open(os.path.join(REPO, "bench_full.json")))          # the full object of the same run
    assert line["value"] == pytest.approx(compact["value"], rel=1e-4)
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["scaling"] == "weak" and line["value"] > 0
    assert line["config"]["per_step_counts"]["received"] == 300000 and "cpu_baseline" not in line
    # every rank's own step time and its all-reduce time are on the line, so that a scaling run can be read rank by rank
    assert len(line["per_rank"]["ms_per_step"]) == 2 and len(line["per_rank"]["allreduce_ms"]) == 2 and min(line["per_rank"]["ms_per_step"]) > 0
    assert line["roofline"]["allreduce_ms_max"] >= 0 and line["roofline"]["rank_ms_per_step_max"] >= line["roofline"]["rank_ms_per_step_min"]
    one = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "2", "--warmup", "1", "--reads", "300000", "--no-cpu",
                          "--leg-steps", "2", "--mixed-reads", "30000", "--mixed-cli-reads", "5000", "--host-fed-seconds", "0.3"],
                         cwd=REPO, capture_output=True, timeout=900)
    assert one.returncode == 0, one.stderr[-2000:]
    last = one.stdout.decode().strip().splitlines()[-1]
    compact = json.loads(last)
    assert len(last) < 8000 and compact["n_gpus"] == 1 and compact["roofline"]["frac"] > 0 and compact["roofline"]["host_fed_mreads"] > 0
    assert all(not isinstance(v2, (dict, list)) for v in compact.values() if isinstance(v, dict) for v2 in v.values())
    single = json.load(open(os.path.join(REPO, "bench_full.json")))
    assert single["value"] == pytest.approx(compact["value"], rel=1e-4)
    assert single["n_gpus"] == 1 and single["config"]["per_step_counts"]["received"] == 300000
    # the PCIe- and host-inclusive legs ride on the same line
    assert single["host_fed"].get("value", 0) > 0 and single["host_fed"]["d2h_bytes_per_read"] > 10, single["host_fed"]
    # ... and the legs for the inputs the memo does not answer, the threshold sweep and configs[4] in miniature
    assert single["robustness"]["substitutions_1pct"]["value"] > 0 and single["robustness"]["background_99pct"]["value"] > 0, single["robustness"]
    assert set(single["thresholds"]) == {"t=0.97", "t=0.95", "t=0.90"} and all(v["value"] > 0 for v in single["thresholds"].values())
    assert set(single["mixed"]["kernels"]) == {"t=0.99", "t=0.97", "t=0.95", "t=0.90"} and single["mixed"]["cli_gzip"].get("value", 0) > 0, single["mixed"]
    assert single["roofline"]["frac"] > 0 and single["cpu_baseline"] if "cpu_baseline" in single else True
    # `value` is the rate through the hashing and graph-walk kernels (memo off): every read was walked, none answered from a table; the flat
    # scalars the driver keeps say how fast those kernels are and what the legs reached
    c = single["config"]
    assert c["walked_reads"] == c["mapped"] > 0.9 * 300000 and "memo off" in c["workload"], c
    rf = single["roofline"]
    assert rf["kernel"] in ("sketch_sig_kernel", "align_kernel", "sketch_seed_kernel<LIST>") and rf["kernel_path_mreads"] == single["value"]
    for k in ("sig_kernel_ms", "align_kernel_ms", "order_ms", "whole_step_frac", "memo_mreads", "sub1_mreads", "sub1_nomemo_mreads", "mixed99_mreads",
              "mixed90_mreads", "host_fed_mreads", "cli_e2e_mreads"):
        assert isinstance(rf.get(k), float) and rf[k] > 0, (k, rf.get(k))
    assert single["memo_tier"]["walked_reads"] < 0.05 * 300000 and single["kernel_path"]["substitutions_1pct"]["value"] > 0
    assert single["cli_e2e"].get("value", 0) > 0 and single["cli_e2e"]["reads"] == 300000, single["cli_e2e"]


@pytest.mark.gpu
def test_many_batches_through_the_cli_match_one_batch_through_the_library(cli, argannot_index, tmp_path):
    """A stream of many batches -- host-fed, several in flight, the ctx opened in the background so that the first batches meet the full-width
    kernels and later ones the signature kernel -- gives the counts of the same reads submitted as ONE device-resident batch, run after run
    (round 4: a signature kernel that classified its reads before staging them lost a few hundred of 174 M records in this stream only, differently
    every time, while every one-batch comparison with the oracle stayed green)."""
    import bench
    from groot_amd import synth

    index = argannot_index
    n = 1_500_000
    cat, off, lens = synth.reference_sequences(index)
    seq, so, _ = synth.reads_np(cat, off, lens, n, 100)
    al = device.Aligner(index, max_batch_reads=n, max_read_len=128, max_batch_bases=int(so[-1]) + 64, memo_budget_mb=device.MEMO_OFF)
    al.submit(seq, so)
    want = al.wait()
    al.close()
    idx_dir = str(tmp_path / "idx")
    os.makedirs(idx_dir)
    index.save(os.path.join(idx_dir, "groot.gidx"))
    fq = str(tmp_path / "reads.fq")
    bench.write_fastq(fq, seq, n)
    for attempt in range(3):
        stats = str(tmp_path / "stats.json")
        r = run([cli, "align", "-i", idx_dir, "-f", fq, "-g", str(tmp_path / "graphs"), "--bam", str(tmp_path / "out.bam"), "--log", str(tmp_path / "a.log"),
                 "-p", "8", "--stats", stats, "--batch", "65536"])
        assert r.returncode == 0, r.stderr
        import json

        st = json.load(open(stats))
        assert (st["reads"], st["mapped"], st["alignments"]) == (n, want["mapped"], want["alignments"]), (attempt, st)
