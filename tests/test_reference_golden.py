"""Intake for values dumped by the REFERENCE ITSELF (tools/golden/main.go run where Go exists): ntHash / MultiHash,
KHF sketches (src/minhash/khf.go:35-55), ContainmentIndex.Query hits (src/lshe/lshe.go:153-175), read counters and BAM
records (src/pipeline/boss.go:194-240), all on a Go-written groot.gg / groot.lshe read through the gob reader.

Until a case directory exists under tests/golden/reference/ the reference-backed tests xfail: PARITY UNPINNED for the
third-party arithmetic.  The same checks always run on a mock case written by this repo's own code, so
that the day a dump arrives a failure means a real difference, not a broken test."""
import glob
import gzip
import json
import os

import numpy as np
import pytest

from bamread import read_bam
from conftest import DATA, read_fastq
from groot_amd import host
from oracle import oracle_py as O

REF_DIR = os.path.join(os.path.dirname(__file__), "golden", "reference")


def case_dirs():
    return sorted(d for d in glob.glob(os.path.join(REF_DIR, "*")) if os.path.exists(os.path.join(d, "golden.json")))


def load_case(d):
    g = json.load(open(os.path.join(d, "golden.json")))
    fq = os.path.join(d, g["fastq"])
    if not os.path.exists(fq):
        for cand in (g["fastq"], g["fastq"] + ".gz", os.path.splitext(g["fastq"])[0] + ".fastq.gz"):
            if os.path.exists(os.path.join(DATA, cand)):
                fq = os.path.join(DATA, cand)
                break
    return g, host.Index.load_gob(d), read_fastq(fq)


def first_diff(a, b):
    a, b = np.asarray(a), np.asarray(b)
    if a.shape != b.shape:
        return f"shapes {a.shape} vs {b.shape}"
    bad = np.argwhere(a != b)
    return None if len(bad) == 0 else f"first difference at {tuple(bad[0])}: got {a[tuple(bad[0])]} expected {b[tuple(bad[0])]}"


# ---- the checks, parametrised over who computes (oracle on the CPU / HIP path on the GPU) --------------------------
def check_hash_vectors(g, sketch_many):
    """a sequence of exactly k bases has one k-mer: its KHF sketch IS MultiHash(canonical, n) of that k-mer"""
    for hv in g["nthash"]:
        seq, k, n = hv["seq"].encode(), hv["k"], hv["n"]
        kmers = [seq[j:j + k] for j in range(len(seq) - k + 1)]
        got = sketch_many(kmers, k, n)
        exp = np.array(hv["hashes"], dtype=np.uint64)
        d = first_diff(got, exp)
        assert d is None, f"MultiHash k={k} n={n} of {hv['seq']}: {d} (axis 0 = k-mer, axis 1 = hash function)"


def window_tuple(index, w):
    a = index.arrays
    return (int(a["win_graph"][w]), int(a["node_seg_id"][a["win_node"][w]]), int(a["win_offset"][w]), int(a["win_merge_span"][w]))


def check_reads(g, index, reads, sketches, seeds):
    """sketches: uint64[n_reads, s]; seeds: (read_id, window_id) records of the whole FASTQ"""
    assert len(reads) == len(g["reads"])
    exp = np.array([r["sketch"] for r in g["reads"]], dtype=np.uint64)
    d = first_diff(sketches, exp)
    assert d is None, f"KHF sketches: {d} (axis 0 = read, axis 1 = slot)"
    by_read = {}
    for rid, w in zip(seeds["read_id"].tolist(), seeds["window_id"].tolist()):
        by_read.setdefault(rid, []).append(window_tuple(index, w))
    for i, r in enumerate(g["reads"]):
        assert r["name"] == reads[i][0].decode() and r["kmer_count"] == len(reads[i][1]) - g["k"] + 1
        got = sorted(by_read.get(i, []))
        want = sorted((h["graph"], h["node"], h["offset"], h["merge_span"]) for h in r["hits"])
        assert got == want, f"Query hits of read {i} ({r['name']}): got {got[:4]}..., reference {want[:4]}..."


def check_containment(g, index):
    a = index.arrays
    s = g["s"]
    lookup = {}          # several windows may sit at one (graph, node, offset): the "-<i>" list of src/pipeline/index.go:195-203
    for w in range(index.view.n_windows):
        lookup.setdefault(window_tuple(index, w), []).append(w)
    n = 0
    for r in g["reads"][:200]:
        q = np.array(r["sketch"], dtype=np.uint64)
        for h in r["hits"]:
            cands = lookup[(h["graph"], h["node"], h["offset"], h["merge_span"])]
            cs = [O.containment(q, a["win_sketch"][w * s:(w + 1) * s], r["kmer_count"], g["w"] - g["k"] + 1) for w in cands]
            assert h["containment"] in cs, (r["name"], cs, h["containment"])
            assert h["containment"] > g["threshold"]
            n += 1
    return n


def canonical_records(recs):
    """the multiset of (name, ref, pos, cigar, flags without Secondary, seq, qual) + one primary per (read, graph)"""
    return sorted((x["name"], x["ref"], x["pos"], x["cigar"], x["flag"] & ~0x100, x["seq"], bytes(x["qual"])) for x in recs)


def records_from_alns(index, reads, alns):
    comp = bytes.maketrans(b"ACGTN", b"TGCAN")
    out = []
    for a in alns:
        name, s, q = reads[int(a["read_id"])]
        if a["rc"]:
            s, q = s.translate(comp)[::-1], q[::-1]
        n = len(s) - int(a["start_clip"]) - int(a["end_clip"])
        cigar = ("1H" if a["start_clip"] else "") + f"{n}M" + ("1H" if a["end_clip"] else "")
        out.append((name.decode(), index.path_name(int(a["ref_id"])), int(a["pos"]), cigar, 0x10 if a["rc"] else 0, s[:n].decode(), q[:n]))
    return sorted(out)


def check_run(g, d, index, reads, counts, alns, kmer_total, kept_names):
    rs = g["read_stats"]
    assert (counts["received"], counts["mapped"], counts["multimapped"]) == (rs[0], rs[1], rs[2]), (counts, rs)
    assert kmer_total == rs[3]
    assert sorted(kept_names) == sorted(g["kept_paths"])
    bam = os.path.join(d, "out.bam")
    if os.path.exists(bam):
        _, refs, recs = read_bam(bam)
        assert sorted(n for n, _ in refs) == sorted(index.path_name(p) for p in range(index.view.n_paths))
        got, want = records_from_alns(index, reads, alns), canonical_records(recs)
        assert len(got) == len(want)
        for x, y in zip(got, want):
            assert x == y, f"BAM record differs: got {x[:5]} reference {y[:5]}"
        # exactly one record without the Secondary flag per AlignRead call = per (read, graph)
        ref_graph = {}
        for gi in range(index.view.n_graphs):
            for p in range(int(index.arrays["graph_path_off"][gi]), int(index.arrays["graph_path_off"][gi + 1])):
                ref_graph[index.path_name(p)] = gi
        primaries = {}
        for x in recs:
            key = (x["name"], ref_graph[x["ref"]])
            primaries[key] = primaries.get(key, 0) + (0 if x["flag"] & 0x100 else 1)
        assert all(v == 1 for v in primaries.values())


def oracle_everything(g, index, reads):
    seq, off = O.pack_reads([r[1] for r in reads])
    run = O.Run(index, g["threshold"])
    run.batch(seq, off)
    kf, kt = run.weights(order=1)
    gk, pk, nr = run.prune(kf, g["min_kmer_cov"])
    kept = [index.path_name(p) for p in range(index.view.n_paths) if pk[p] and gk[np.searchsorted(index.arrays["graph_path_off"], p, side="right") - 1]]
    return run, int(kt.sum()), kept


def oracle_sketch_many(seqs, k, n):
    return np.array([O.khf_sketch(s, k, n) for s in seqs], dtype=np.uint64)


def run_checks_with_oracle(d):
    g, index, reads = load_case(d)
    check_hash_vectors(g, oracle_sketch_many)
    check_hash_vectors(g, lambda seqs, k, n: np.array([host.window_sketch(s, k, n) for s in seqs], dtype=np.uint64))   # the index builder's copy
    run, kmer_total, kept = oracle_everything(g, index, reads)
    check_reads(g, index, reads, run.sketches(), run.seeds())
    assert check_containment(g, index) > 0
    check_run(g, d, index, reads, run.counts(), run.alns(), kmer_total, kept)


# ---- a mock case written by this repo's own code: keeps the intake honest ------------------------------------------
def write_mock_case(d, msa_files, fastq, k, s, w, t, c):
    """same files and schema as tools/golden/main.go, computed by the oracle + this repo's gob writer / BAM writer"""
    os.makedirs(d, exist_ok=True)
    index = host.Index.from_msa_files(msa_files, host.index_params(k=k, s=s, w=w))
    index.save_gob(d)
    index = host.Index.load_gob(d)
    reads = read_fastq(fastq)
    g = {"groot_version": "1.1.2", "fastq": os.path.basename(fastq), "k": k, "s": s, "w": w, "num_part": 8, "max_k": 4, "threshold": t,
         "min_kmer_cov": c, "nthash": [], "reads": []}
    for sq in (b"ACTGCGTGCGTGAAACGTGCACGTGACGTG", b"NNNNACGTacgtNNNNACGTACGTTTTTGGGGCCCCAAAATGCATGCATGCA"):
        for kk, nn in ((7, 10), (k, s)):
            if len(sq) >= kk:
                g["nthash"].append({"seq": sq.decode(), "k": kk, "n": nn,
                                    "hashes": [[int(v) for v in O.khf_sketch(sq[j:j + kk], kk, nn)] for j in range(len(sq) - kk + 1)]})
    run, kmer_total, kept = oracle_everything({"threshold": t, "min_kmer_cov": c}, index, reads)
    sk, seeds = run.sketches(), run.seeds()
    hits = {}
    a = index.arrays
    for rid, wid in zip(seeds["read_id"].tolist(), seeds["window_id"].tolist()):
        gi, node, offs, ms = window_tuple(index, wid)
        q = sk[rid]
        hits.setdefault(rid, []).append({"graph": gi, "node": node, "offset": offs, "merge_span": ms,
                                         "ref": a["win_ref"][a["win_ref_off"][wid]:a["win_ref_off"][wid + 1]].tolist(),
                                         "containment": O.containment(q, a["win_sketch"][wid * s:(wid + 1) * s], len(reads[rid][1]) - k + 1, w - k + 1)})
    for i, (name, sq, _) in enumerate(reads):
        g["reads"].append({"name": name.decode(), "len": len(sq), "kmer_count": len(sq) - k + 1, "sketch": [int(v) for v in sk[i]],
                           "hits": hits.get(i, [])})
    oc = run.counts()
    g["read_stats"] = [int(oc["received"]), int(oc["mapped"]), int(oc["multimapped"]), kmer_total]
    g["kept_paths"] = sorted(kept)
    g["num_windows"], g["num_graphs"] = int(index.view.n_windows), int(index.view.n_graphs)
    json.dump(g, open(os.path.join(d, "golden.json"), "w"))
    seq, off = O.pack_reads([r[1] for r in reads])
    noff = np.concatenate([[0], np.cumsum([len(r[0]) for r in reads])]).astype(np.uint64)
    batch = {"seq": seq, "qual": np.frombuffer(b"".join(r[2] for r in reads), dtype=np.uint8), "seq_off": off,
             "names": np.frombuffer(b"".join(r[0] for r in reads), dtype=np.uint8), "name_off": noff}
    bw = host.BamWriter(os.path.join(d, "out.bam"), index)
    bw.write(run.alns(), batch)
    bw.close()
    return d


@pytest.fixture(scope="module")
def mock_case(tmp_path_factory, native_libs):
    d = str(tmp_path_factory.mktemp("mockref") / "genes_k51_s30")
    return write_mock_case(d, [os.path.join(DATA, "test-genes.msa")], os.path.join(DATA, "test-reads-OXA90-OXA106-100bp-with-errors.fastq.gz"),
                           51, 30, 100, 0.99, 10.0)


def test_intake_machinery_on_a_mock_case(mock_case):
    run_checks_with_oracle(mock_case)
    # and it does catch a slip: one flipped bit in one MultiHash value / one sketch slot
    g = json.load(open(os.path.join(mock_case, "golden.json")))
    g["nthash"][0]["hashes"][2][3] ^= 1
    with pytest.raises(AssertionError, match="MultiHash"):
        check_hash_vectors(g, oracle_sketch_many)
    g = json.load(open(os.path.join(mock_case, "golden.json")))
    g["reads"][5]["sketch"][7] ^= 1 << 40
    _, index, reads = load_case(mock_case)
    run, _, _ = oracle_everything(g, index, reads)
    with pytest.raises(AssertionError, match="KHF sketches"):
        check_reads(g, index, reads, run.sketches(), run.seeds())


@pytest.mark.parametrize("case", case_dirs() or [None])
def test_oracle_against_the_reference_dump(case, native_libs):
    if case is None:
        pytest.xfail("PARITY UNPINNED: no reference dump under tests/golden/reference/ -- run tools/golden/main.go where Go exists "
                     "(tests/golden/reference/README.md has the exact commands)")
    run_checks_with_oracle(case)


@pytest.mark.gpu
@pytest.mark.parametrize("case", case_dirs() or ["mock"])
def test_hip_path_against_the_reference_dump(case, mock_case, hip_lib):
    """the same comparison with the HIP path doing the computing (on the mock case when no reference dump exists: then it
    is device-vs-oracle once more, through the dump format)"""
    from groot_amd import device

    d = mock_case if case == "mock" else case
    g, index, reads = load_case(d)
    al = device.Aligner(index, threshold=g["threshold"], keep_sketches=True, max_batch_reads=max(1024, len(reads)), max_read_len=512)

    def sketch_many(seqs, k, n):
        if (k, n) != (g["k"], g["s"]):          # a ctx sketches with its index's (k, s); other vectors are the oracle's job
            return oracle_sketch_many(seqs, k, n)
        cat, off = O.pack_reads(list(seqs))
        return al.sketch(cat, off)

    check_hash_vectors(g, sketch_many)
    seq, off = O.pack_reads([r[1] for r in reads])
    al.submit(seq, off)
    counts = al.wait()
    check_reads(g, index, reads, al.sketches(), al.seeds())
    q, rows = al.attempts_rows()
    kf, kt = device.weights_rows(index, q, rows)
    gk, pk, nr = device.prune(index, kf, g["min_kmer_cov"])
    kept = [index.path_name(p) for p in range(index.view.n_paths) if pk[p] and gk[np.searchsorted(index.arrays["graph_path_off"], p, side="right") - 1]]
    check_run(g, d, index, reads, counts, al.alns(), int(kt.sum()), kept)
    al.close()
    if case == "mock":
        pytest.xfail("PARITY UNPINNED: the HIP path matches the dump format of a MOCK case only; no reference dump is present")
