"""Host logic either side of the device path: index build (MSA/GFA -> graphs -> windows), record
expansion, weighting, pruning, GFA/BAM/FASTQ I/O -- checked against the oracle and the reference's fixtures."""
import gzip
import os
import re

import numpy as np
import pytest

from bamread import read_bam
from conftest import DATA
from groot_amd import device, host, synth
from oracle import oracle_py as O


def msa_rows(path):
    rows = []
    for line in open(path):
        line = line.strip()
        if line.startswith(">"):
            rows.append([line[1:], ""])
        elif line:
            rows[-1][1] += line
    return [(n, s) for n, s in rows if n != "consensus"]


def test_version_matches_reference():
    assert host.lib().groot_host_version() == b"1.1.2"   # src/version/version.go, checked at cmd/align.go:96


def test_paths_spell_the_msa_rows(msa_dir, small_index):
    """CreateGrootGraph + Graph2Seqs: every path re-spells its gap-stripped, upper-cased MSA row; Lengths match"""
    a = small_index.arrays
    for g, f in enumerate(host.msa_files(msa_dir)[:24]):
        p0 = int(a["graph_path_off"][g])
        for lp, (name, seq) in enumerate(msa_rows(f)):
            exp = "".join(c.upper() if c.upper() in "ACGTN" else "N" for c in seq.replace("-", ""))
            assert small_index.path_name(p0 + lp) == name
            assert small_index.path_sequence(g, lp).decode() == exp
            assert a["path_len"][p0 + lp] == len(exp)


def test_graph_is_topologically_sorted_dag(small_index):
    a = small_index.arrays
    for n in range(small_index.view.n_nodes):
        for e in a["edges"][a["node_edge_off"][n]:a["node_edge_off"][n + 1]]:
            assert e > n                      # SortedNodes order: every edge goes forward
        es = a["node_seg_id"][a["edges"][a["node_edge_off"][n]:a["node_edge_off"][n + 1]]]
        assert list(es) == sorted(es, reverse=True)   # traverse() leaves OutEdges sorted descending (graph.go:203)
    # Position = offset of the node in the path's linear sequence
    for g in range(3):
        for n in range(int(a["graph_node_off"][g]), int(a["graph_node_off"][g + 1])):
            for j in range(int(a["node_np_off"][n]), int(a["node_np_off"][n + 1])):
                seq = small_index.path_sequence(g, int(a["np_path"][j]))
                pos = int(a["np_pos"][j])
                assert seq[pos:pos + len(small_index.node_seq(n))] == small_index.node_seq(n)


def test_window_sketches_and_quirks(small_index):
    """WindowGraph (graph.go:229-396): Key.Sketch = KHF sketch of the window's first path position;
    windows are numbered in canonical seed order; the last run of a multi-run path is not indexed"""
    a = small_index.arrays
    v = small_index.view
    key = list(zip(a["win_graph"].tolist(), a["node_seg_id"][a["win_node"]].tolist(), a["win_offset"].tolist()))
    assert key == sorted(key)
    sk = a["win_sketch"].reshape(-1, v.sketch_size)
    rng = np.random.default_rng(1)
    for w in rng.choice(v.n_windows, 200, replace=False):
        g = int(a["win_graph"][w])
        lp = int(a["win_ref"][a["win_ref_off"][w]])
        node = int(a["win_node"][w])
        j = [j for j in range(int(a["node_np_off"][node]), int(a["node_np_off"][node + 1])) if a["np_path"][j] == lp][0]
        start = int(a["np_pos"][j]) + int(a["win_offset"][w])
        seq = small_index.path_sequence(g, lp)[start:start + v.window_size]
        assert len(seq) == v.window_size
        assert np.array_equal(sk[w], O.khf_sketch(seq, v.kmer_size, v.sketch_size))
        assert np.array_equal(sk[w], host.window_sketch(seq, v.kmer_size, v.sketch_size))
        # ContainedNodes counts: bases x merged windows of the first path, summed over merged paths
        cnt = a["cn_count"][a["win_cn_off"][w]:a["win_cn_off"][w + 1]]
        assert cnt.sum() % v.window_size == 0 and cnt.sum() >= (int(a["win_merge_span"][w]) + 1) * v.window_size
    # quirk 1: the final window position of a path with >= 2 runs is never the start of an indexed window
    g = 0
    for lp in range(int(a["graph_path_off"][1] - a["graph_path_off"][0])):
        plen = int(a["path_len"][lp])
        covered = set()
        for w in np.flatnonzero(a["win_graph"] == g):
            refs = a["win_ref"][a["win_ref_off"][w]:a["win_ref_off"][w + 1]]
            if lp in refs:
                node = int(a["win_node"][w])
                j = [j for j in range(int(a["node_np_off"][node]), int(a["node_np_off"][node + 1])) if a["np_path"][j] == lp][0]
                s = int(a["np_pos"][j]) + int(a["win_offset"][w])
                covered.update(range(s, s + int(a["win_merge_span"][w]) + 1))
        if len(covered) > 1:
            assert (plen - v.window_size) not in covered


def test_index_save_load_roundtrip(small_index, tmp_path):
    f = str(tmp_path / "x.gidx")
    small_index.save(f)
    again = host.Index.load(f)
    for k, arr in small_index.arrays.items():
        assert np.array_equal(arr, again.arrays[k]), k
    for n in ("kmer_size", "sketch_size", "window_size", "num_part", "max_k", "num_window_kmers", "path_words"):
        assert getattr(small_index.view, n) == getattr(again.view, n)
    open(f, "r+b").truncate(1000)
    with pytest.raises(host.GrootError):
        host.Index.load(f)


def test_index_errors(tmp_path):
    with pytest.raises(host.GrootError):
        host.Index.from_msa_files([str(tmp_path / "missing.msa")])
    bad = tmp_path / "cluster-1.msa"
    bad.write_text(">a\nACGT\n>b\nACG\n")
    with pytest.raises(host.GrootError):
        host.Index.from_msa_files([str(bad)])
    with pytest.raises(host.GrootError):   # k > w (cmd/index.go:161-163)
        host.Index.from_msa_files([os.path.join(DATA, "test.msa")], host.index_params(k=31, w=20))
    short = tmp_path / "cluster-2.msa"
    short.write_text(">a\nACGTACGTAC\n>b\nACGTTCGTAC\n")
    with pytest.raises(host.GrootError):   # every graph masked: "could not create and sketch any graphs"
        host.Index.from_msa_files([str(short)])


def test_gfa_and_msa_fixtures():
    """src/graph/graph_test.go: test.gfa loads (133 S / 176 L / 6 P); test.msa windows with w=150 k=7 s=128"""
    g = host.Index.from_gfa_files([os.path.join(DATA, "test.gfa")], host.index_params(k=7, s=10, w=30))
    assert (g.view.n_nodes, g.view.n_edges, g.view.n_paths) == (133, 176, 6)
    assert g.arrays["path_len"].tolist() == [len(g.path_sequence(0, p)) for p in range(6)]
    assert g.arrays["path_len"][:2].tolist() == [747, 750]
    m = host.Index.from_msa_files([os.path.join(DATA, "test.msa")], host.index_params(k=7, s=128, w=150))
    assert m.view.n_paths == len(msa_rows(os.path.join(DATA, "test.msa"))) and m.view.n_windows > 0
    assert m.view.num_window_kmers == 144


@pytest.fixture(scope="module")
def small_run(small_index):
    cat, off, lens = synth.reference_sequences(small_index)
    seq, so, truth = synth.reads_np(cat, off, lens, 4000, 100)
    run = O.Run(small_index)
    run.batch(seq, so)
    return run, seq, so, truth


def test_weights_prune_match_oracle(small_index, small_run):
    run = small_run[0]
    att = run.attempts()
    kf, kt = device.weights(small_index, att)
    kf1, kt1 = run.weights(order=1)
    assert np.array_equal(kf, kf1) and np.array_equal(kt, kt1)
    for cov in (1.0, 10.0, 1e9):
        got = device.prune(small_index, kf, cov)
        exp = run.prune(kf, cov)
        assert np.array_equal(got[0], exp[0])
        kept_graphs = np.flatnonzero(exp[0])
        a = small_index.arrays
        for g in kept_graphs:
            sl = slice(int(a["graph_path_off"][g]), int(a["graph_path_off"][g + 1]))
            assert np.array_equal(got[1][sl], exp[1][sl])
            nl = slice(int(a["graph_node_off"][g]), int(a["graph_node_off"][g + 1]))
            assert np.array_equal(got[2][nl], exp[2][nl])
    assert not device.prune(small_index, kf, 1e9)[0].any()


def test_truth_is_found(small_index, small_run):
    run, _, _, truth = small_run
    al = run.alns()
    have = set(zip(al["read_id"].tolist(), al["ref_id"].tolist(), al["pos"].tolist(), al["rc"].tolist()))
    hit = sum((i, int(truth["seq"][i]), int(truth["start"][i]), int(truth["strand"][i])) in have for i in range(4000))
    assert hit >= 3900


def test_gfa_writer_dialect(small_index, small_run, tmp_path):
    """SaveGraphAsGFA output follows the dialect of src/graph/test2.gfa and reloads as the pruned graph"""
    run = small_run[0]
    kf, kt = run.weights(order=1)
    gk, pk, nr = device.prune(small_index, kf, 1.0)
    g = int(np.flatnonzero(gk)[0])
    f = str(tmp_path / f"groot-graph-{g}.gfa")
    assert host.save_gfa(small_index, g, kf, pk, nr, int(kt.sum()), f, timestamp="Wed Apr 24 09:14:36 2019")
    lines = open(f).read().split("\n")
    ref = open(os.path.join(DATA, "test2.gfa")).read().split("\n")
    assert lines[0] == ref[0] == "H\tVN:Z:1"
    assert lines[1].startswith("#\tvariation graph created by groot") and lines[2].startswith("#\tthis graph is approximately weighted")
    assert re.search(r"graphs: (\d+)\)", lines[2]).group(1) == str(int(kt.sum()))   # parsed by haplotype.go:45-50
    kinds = [l[0] for l in lines[3:] if l]
    assert "".join(kinds) == "S" * kinds.count("S") + "L" * kinds.count("L") + "P" * kinds.count("P")
    assert all(re.fullmatch(r"S\t\d+\t[ACGTN]+\tLN:i:\d+\tKC:i:\d+", l) for l in lines if l.startswith("S"))
    assert all(re.fullmatch(r"L\t\d+\t\+\t\d+\t\+\t0M", l) for l in lines if l.startswith("L"))
    assert all(re.fullmatch(r"P\t\S+\t(\d+\+,)*\d+\+\t(\d+M,)*\d+M", l) for l in lines if l.startswith("P"))
    again = host.Index.from_gfa_files([f], host.index_params())
    a = small_index.arrays
    kept_paths = [small_index.path_name(p) for p in range(int(a["graph_path_off"][g]), int(a["graph_path_off"][g + 1])) if pk[p]]
    assert [again.path_name(i) for i in range(again.view.n_paths)] == kept_paths
    for i, name in enumerate(kept_paths):
        lp = [small_index.path_name(p) for p in range(int(a["graph_path_off"][g]), int(a["graph_path_off"][g + 1]))].index(name)
        assert again.path_sequence(0, i) == small_index.path_sequence(g, lp)
    # a graph nobody mapped to is not written (graphio.go:67-69)
    unused = int(np.flatnonzero(kt == 0)[0]) if (kt == 0).any() else None
    if unused is not None and not kf[int(a["graph_node_off"][unused]):int(a["graph_node_off"][unused + 1])].any():
        assert not host.save_gfa(small_index, unused, kf, np.ones_like(pk), np.zeros_like(nr), 0, str(tmp_path / "no.gfa"))


def test_fastq_reader(tmp_path, perfect_reads):
    src = os.path.join(DATA, "full-argannot-perfect-reads-small.fq.gz")
    plain = tmp_path / "a.fastq"
    plain.write_bytes(gzip.open(src).read().replace(b"\n", b"\r\n"))      # CRLF + plain text
    for files in ([src], [str(plain)], [src, str(plain)]):
        got = []
        for b in host.FastqReader(files).batches(max_reads=333):
            for i in range(b["n"]):
                got.append((bytes(b["names"][int(b["name_off"][i]):int(b["name_off"][i + 1])]),
                            bytes(b["seq"][int(b["seq_off"][i]):int(b["seq_off"][i + 1])]),
                            bytes(b["qual"][int(b["seq_off"][i]):int(b["seq_off"][i + 1])])))
        assert got == list(perfect_reads) * len(files)
    bad = tmp_path / "bad.fq"
    bad.write_text("@r1\nACGT\n+\nIIII\nr2\nACGT\n+\nIIII\n")
    with pytest.raises(host.GrootError) as e:
        list(host.FastqReader([str(bad)]).batches())
    assert "does not begin with @" in str(e.value)
    with pytest.raises(host.GrootError):
        host.FastqReader([str(tmp_path / "nope.fq")])


def test_bam_writer_roundtrip(small_index, tmp_path):
    cat, off, lens = synth.reference_sequences(small_index)
    seq, so, _ = synth.reads_np(cat, off, lens, 300, 100)
    fq = tmp_path / "r.fq"
    with open(fq, "wb") as f:
        for i in range(300):
            s = bytes(seq[int(so[i]):int(so[i + 1])])
            f.write(b"@read_%d extra\n%s\n+\n%s\n" % (i, s, bytes(33 + (j * 7 + i) % 40 for j in range(len(s)))))
    run = O.Run(small_index)
    batch = next(host.FastqReader([str(fq)]).batches())
    run.batch(batch["seq"], batch["seq_off"])
    al = run.alns().astype(device.ALN_DTYPE)
    out = str(tmp_path / "out.bam")
    bw = host.BamWriter(out, small_index, date="2020-01-01T00:00:00Z")
    bw.write(al, batch)
    bw.close()
    text, refs, recs = read_bam(out)
    assert text.startswith("@HD\tVN:1.5") and "@PG\tID:1\tPN:groot\tCL:groot align\tVN:1.1.2" in text and "@RG\tID:readsID" in text
    assert refs == [(small_index.path_name(p), int(small_index.arrays["path_len"][p])) for p in range(small_index.view.n_paths)]
    assert len(recs) == len(al) > 300
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    for a, r in zip(al, recs):
        i = int(a["read_id"])
        s = bytes(seq[int(so[i]):int(so[i + 1])])
        q = bytes(33 + (j * 7 + i) % 40 for j in range(len(s)))
        if a["rc"]:
            s, q = s.translate(comp)[::-1], q[::-1]
        assert r["name"] == f"read_{i} extra" and r["ref_id"] == int(a["ref_id"]) and r["pos"] == int(a["pos"])
        assert r["seq"].encode() == s and r["qual"] == q and r["cigar"] == "100M" and r["mapq"] == 30
        assert r["flag"] == (0x10 if a["rc"] else 0) | (0x100 if a["secondary"] else 0)
        assert (r["next_ref"], r["next_pos"], r["tlen"]) == (-1, -1, 0)


def test_bam_fast_path_equals_record_path(small_index, tmp_path):
    """groot_bam_write_travs (parallel traversal -> record -> BGZF) writes the same records as expand + write"""
    import ctypes as C

    from groot_amd import _ffi

    cat, off, lens = synth.reference_sequences(small_index)
    seq, so, _ = synth.reads_np(cat, off, lens, 3000, 100)
    qual = np.frombuffer(bytes(33 + (i * 13) % 41 for i in range(len(seq))), dtype=np.uint8).copy()
    names = b"".join(b"r%d" % i for i in range(3000))
    noff = np.zeros(3001, dtype=np.uint64)
    noff[1:] = np.cumsum([len(b"r%d" % i) for i in range(3000)])
    batch = {"seq": seq, "qual": qual, "seq_off": so, "names": np.frombuffer(names, dtype=np.uint8).copy(), "name_off": noff}
    run = O.Run(small_index)
    run.batch(seq, so, first_read_id=100)
    al = run.alns().astype(device.ALN_DTYPE)
    # traversal form of the same alignments: one traversal per (read, graph, first-node) group is enough for the
    # writer, so derive them from the expanded records (path set = the group's path ids)
    pw = small_index.view.path_words
    a = small_index.arrays
    travs, masks = [], []
    i = 0
    while i < len(al):
        j = i
        m = np.zeros(pw, dtype=np.uint64)
        while j < len(al) and al["read_id"][j] == al["read_id"][i] and al["graph_id"][j] == al["graph_id"][i] and (j == i or al["secondary"][j]):
            p = int(al["path_id"][j])
            if m[p // 64] >> np.uint64(p % 64) & np.uint64(1):
                break
            m[p // 64] |= np.uint64(1) << np.uint64(p % 64)
            j += 1
        # first node of the alignment: the node of path p that holds pos
        g, p0, pos = int(al["graph_id"][i]), int(al["path_id"][i]), int(al["pos"][i])
        node = off_in = None
        for n in range(int(a["graph_node_off"][g]), int(a["graph_node_off"][g + 1])):
            for jj in range(int(a["node_np_off"][n]), int(a["node_np_off"][n + 1])):
                if a["np_path"][jj] == p0 and a["np_pos"][jj] <= pos < a["np_pos"][jj] + (a["node_seq_off"][n + 1] - a["node_seq_off"][n]):
                    node, off_in = n, pos - int(a["np_pos"][jj])
        flags = (1 if al["rc"][i] else 0) | (2 if al["start_clip"][i] else 0) | (4 if al["end_clip"][i] else 0) | 8
        travs.append((int(al["read_id"][i]), g, node, off_in, len(travs) & 0xFFFF, flags, 0))
        masks.append(m)
        i = j
    tr = np.array(travs, dtype=device.TRAV_DTYPE)
    mk = np.array(masks, dtype=np.uint64)
    exp = device.expand_alns(small_index, tr, mk)
    if not all(np.array_equal(exp[f], al[f]) for f in al.dtype.names):
        pytest.skip("alignment groups of this sample are not expressible as one traversal each")
    slow, fast = str(tmp_path / "slow.bam"), str(tmp_path / "fast.bam")
    bw = host.BamWriter(slow, small_index, date="2020-01-01T00:00:00Z")
    bw.write(al, batch, first_read_id=100)
    bw.close()
    H = host.lib()
    h = C.c_void_p()
    host._check(H.groot_bam_open(fast.encode(), C.byref(small_index.view), b"2020-01-01T00:00:00Z", C.byref(h)))
    host._check(H.groot_bam_set_threads(h, C.c_uint32(4)))

    class RB(C.Structure):
        _fields_ = [("seq", C.c_void_p), ("qual", C.c_void_p), ("seq_off", C.c_void_p), ("names", C.c_void_p), ("name_off", C.c_void_p),
                    ("n_reads", C.c_uint32), ("first_read_id", C.c_uint32)]

    rb = RB(seq.ctypes.data, qual.ctypes.data, so.ctypes.data, batch["names"].ctypes.data, noff.ctypes.data, 3000, 100)
    nrec = C.c_uint64()
    host._check(H.groot_bam_write_travs(h, C.byref(small_index.view), C.byref(rb), tr.ctypes.data_as(C.c_void_p), _ffi.as_ptr(mk, C.c_uint64),
                                        C.c_uint64(len(tr)), C.byref(nrec)))
    host._check(H.groot_bam_close(h))
    assert nrec.value == len(al)
    assert read_bam(slow) == read_bam(fast)
    # level -2: members written from the records' structure (bgzf_struct.hpp: back-references to the previous record of the same
    # size, fixed Huffman codes, carry-less-multiplication CRC) -- the inflated stream is byte for byte the zlib path's
    import gzip

    struct = str(tmp_path / "struct.bam")
    host._check(H.groot_bam_open(struct.encode(), C.byref(small_index.view), b"2020-01-01T00:00:00Z", C.byref(h)))
    host._check(H.groot_bam_set_threads(h, C.c_uint32(3)))
    host._check(H.groot_bam_set_level(h, C.c_int(-2)))
    host._check(H.groot_bam_write_travs(h, C.byref(small_index.view), C.byref(rb), tr.ctypes.data_as(C.c_void_p), _ffi.as_ptr(mk, C.c_uint64),
                                        C.c_uint64(len(tr)), C.byref(nrec)))
    host._check(H.groot_bam_close(h))
    assert gzip.open(struct).read() == gzip.open(fast).read()
    assert read_bam(struct) == read_bam(fast)
    assert os.path.getsize(struct) < 1.02 * len(gzip.open(fast).read())       # (few records per traversal here: first records travel as stored blocks)


def test_pack_reads():
    """2-bit packing for groot_hip_submit_packed: code (byte >> 1) & 3, everything that is not ACGT listed as an exception"""
    rng = np.random.default_rng(5)
    seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 1_000_003)].copy()
    odd = rng.choice(len(seq), 500, replace=False)
    seq[odd] = np.frombuffer(b"NnacgtRY*\x00", dtype=np.uint8)[rng.integers(0, 10, 500)]
    for threads in (1, 0):
        packed, pos, byte = host.pack_reads(seq, threads)
        assert len(packed) == (len(seq) + 3) // 4
        assert np.array_equal(pos, np.sort(odd).astype(np.uint64)) and np.array_equal(byte, seq[np.sort(odd)])
        codes = (packed[:, None] >> np.array([0, 2, 4, 6], dtype=np.uint8)) & 3
        back = np.frombuffer(b"ACTG", dtype=np.uint8)[codes.reshape(-1)[: len(seq)]]
        keep = np.ones(len(seq), bool)
        keep[odd] = False
        assert np.array_equal(back[keep], seq[keep])
    empty = host.pack_reads(np.zeros(0, np.uint8))
    assert len(empty[0]) == 0 and len(empty[1]) == 0
