"""Parity tests proper: the HIP path (through the C ABI of libgroot_hip.so) against the CPU oracle on
the same inputs -- bit-exact sketches, seed sets, alignment records, IncrementSubPath call counts,
counters, graph weights and pruning -- plus size-independent properties at BASELINE.json's full size."""
import json
import os

import numpy as np
import pytest

from conftest import DATA, digest
from groot_amd import device, host, synth
from oracle import oracle_py as O

pytestmark = pytest.mark.gpu

@pytest.fixture(autouse=True, params=["align_kernel", "lean_first"])
def align_stage(request, monkeypatch):
    """every test of this module twice: the align stage as shipped (align_kernel alone), and with its first pass in front (GROOT_LEAN=1,
    kernels_lean.hpp) -- the results must not depend on it"""
    if request.param == "lean_first":
        if request.node.name.startswith("test_kernel_path_at_benchmark_size") or "background" in request.node.name:
            pytest.skip("compares the two itself / not about the align stage")
        monkeypatch.setenv("GROOT_LEAN", "1")
    else:
        monkeypatch.delenv("GROOT_LEAN", raising=False)
    return request.param


GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "oracle_golden.json")


@pytest.fixture(scope="module", autouse=True)
def need_gpu(hip_lib):
    assert device.device_count() > 0, "no MI355X visible: the HIP path has no CPU fallback"


# Every test that goes through run_both runs twice: with the sketches kept (the full-width sketch kernel decides every read, and the
# sketches themselves are compared) and without (the signature kernel runs in front of it wherever it applies: exact-table
# reads of an index with a compiled instance; the full-width kernel takes what it cannot decide).  Same expectations.
KEEP_SKETCHES = True


@pytest.fixture(params=["full-width", "signature"])
def sketch_path(request):
    global KEEP_SKETCHES
    KEEP_SKETCHES = request.param == "full-width"
    yield request.param


def run_both(index, seq, off, threshold=0.99, no_align=False, first=0, **kw):
    al = device.Aligner(index, threshold=threshold, no_align=no_align, keep_sketches=KEEP_SKETCHES,
                        max_batch_reads=max(1024, len(off) - 1), **kw)
    al.submit(seq, off, first_read_id=first)
    counts = al.wait()
    run = O.Run(index, threshold, no_align)
    run.batch(seq, off, first_read_id=first)
    return al, counts, run


def assert_same(al, counts, run, index):
    oc = run.counts()
    for k in ("received", "mapped", "multimapped", "alignments", "seeds", "revcomp_panics"):
        assert counts[k] == oc[k], (k, counts[k], oc[k])
    if KEEP_SKETCHES:
        assert np.array_equal(al.sketches(), run.sketches())
        assert counts["full_sketch_reads"] == counts["received"]
    assert np.array_equal(al.seeds(), run.seeds().astype(device.SEED_DTYPE))
    got, exp = al.alns(), run.alns()
    assert len(got) == len(exp)
    for f in exp.dtype.names:
        assert np.array_equal(got[f], exp[f]), f
    att, oatt = al.attempts(), run.attempts()
    assert np.array_equal(att[: oatt.shape[0]], oatt) and not att[oatt.shape[0]:].any()
    kf, kt = device.weights(index, att)
    okf, okt = run.weights(order=1)
    assert np.array_equal(kf, okf) and np.array_equal(kt, okt)
    return got


def pack(reads):
    return O.pack_reads([r[1] for r in reads])


def test_perfect_reads_fixture_and_golden(argannot_index, perfect_reads, sketch_path):
    seq, off = pack(perfect_reads)
    al, counts, run = run_both(argannot_index, seq, off)
    assert_same(al, counts, run, argannot_index)
    exp = json.load(open(GOLDEN))["perfect_reads_small@arg-annot.90(k31,s21,w100),t0.99"]
    if KEEP_SKETCHES:
        assert digest(al.sketches()) == exp["sketches"]
    else:
        assert counts["full_sketch_reads"] < 0.05 * counts["received"]     # error-free 100-mers: decided by signature + text
    assert digest(al.seeds().astype(O.SEED_DTYPE)) == exp["seeds"]
    assert digest(al.alns().astype(O.ALN_DTYPE)) == exp["alns"]
    kf, kt = device.weights(argannot_index, al.attempts())
    assert digest(kf) == exp["kmer_freq"] and digest(kt) == exp["kmer_total"]
    assert {k: counts[k] for k in ("received", "mapped", "multimapped", "alignments", "seeds")} == \
           {k: exp["counts"][k] for k in ("received", "mapped", "multimapped", "alignments", "seeds")}
    al.close()


@pytest.mark.parametrize("threshold", [0.99, 0.90])
def test_variable_length_reads_general_lsh_path(argannot_index, variable_reads, threshold, sketch_path):
    """50-100 bp reads: kmerCount < NumWindowKmers, so hits need fewer than all slots equal and the LSH
    forest (K, L) prefix search is used instead of the exact-sketch table; seed lists overflow their slots"""
    seq, off = pack(variable_reads)
    al, counts, run = run_both(argannot_index, seq, off, threshold=threshold, max_seeds_per_read=2)
    assert_same(al, counts, run, argannot_index)
    key = f"perfect_reads_small_variable_rl@arg-annot.90,t{threshold:.2f}"
    assert digest(al.alns().astype(O.ALN_DTYPE)) == json.load(open(GOLDEN))[key]["alns"]
    al.close()


def test_reads_with_errors_and_clipping(genes_index, oxa_reads, sketch_path):
    """src/pipeline/3_sketch_test.go input (k=51, s=30): failing seeds, level-3/4 hard clips, pruning"""
    seq, off = pack(oxa_reads)
    al, counts, run = run_both(genes_index, seq, off)
    got = assert_same(al, counts, run, genes_index)
    assert got["start_clip"].any() and got["end_clip"].any()
    kf, kt = device.weights(genes_index, al.attempts())
    gk, pk, nr = device.prune(genes_index, kf, 10.0)
    kept = [genes_index.path_name(i) for i in range(genes_index.view.n_paths) if pk[i]]
    assert "argannot~~~(Bla)OXA-90~~~EU547443:1-825" in kept and gk.tolist() == [1]
    al.close()


def test_config0_synthetic_10k(argannot_index, sketch_path):
    """BASELINE configs[0]: 10k synthetic 100 bp reads on arg-annot.90"""
    cat, o, lens = synth.reference_sequences(argannot_index)
    seq, off, truth = synth.reads_np(cat, o, lens, 10_000, 100)
    al, counts, run = run_both(argannot_index, seq, off, first=777)
    got = assert_same(al, counts, run, argannot_index)
    have = set(zip(got["read_id"].tolist(), got["ref_id"].tolist(), got["pos"].tolist(), got["rc"].tolist()))
    hits = sum((777 + i, int(truth["seq"][i]), int(truth["start"][i]), int(truth["strand"][i])) in have for i in range(10_000))
    assert hits >= 9900
    al.close()


def test_small_graph_fixture_all_windows(testgfa_index, sketch_path):
    """src/graph/test.gfa (k=7, s=10, w=30): every window of every path, both strands, plus shifted reads"""
    idx = testgfa_index
    reads = []
    for p in range(6):
        s = idx.path_sequence(0, p)
        for i in range(0, len(s) - 30, 3):
            r = s[i:i + 30]
            reads.append(r)
            reads.append(O.revcomp(r)[0])
            reads.append(s[i:i + 25])      # shorter than the window
    seq, off = O.pack_reads(reads)
    for t in (0.99, 0.8):
        al, counts, run = run_both(idx, seq, off, threshold=t, max_seeds_per_read=1)
        assert_same(al, counts, run, idx)
        assert counts["alignments"] > len(reads)
        al.close()


def test_wide_graph_uses_the_wide_node_records(tmp_path, sketch_path):
    """a cluster with 260 alleles (5 path words): the 128-byte NodeRec<11> variant of the align kernel"""
    rng = np.random.default_rng(11)
    base = rng.choice(list(b"ACGT"), size=420).astype(np.uint8)
    rows = []
    for i in range(260):
        s = base.copy()
        for p in rng.choice(420, size=4, replace=False):
            s[p] = rng.choice([c for c in b"ACGT" if c != s[p]])
        row = bytearray(s.tobytes())
        if i % 7 == 0:                       # a deletion
            d = int(rng.integers(50, 360))
            row[d:d + 3] = b"---"
        rows.append(bytes(row))
    f = tmp_path / "cluster-0.msa"
    f.write_bytes(b"".join(b">*seq%d\n%s\n" % (i, r) if i == 0 else b">seq%d\n%s\n" % (i, r) for i, r in enumerate(rows)))
    idx = host.Index.from_msa_files([str(f)])
    assert idx.view.path_words == 5 and idx.view.n_paths == 260
    cat, o, lens = synth.reference_sequences(idx)
    seq, off, _ = synth.reads_np(cat, o, lens, 6000, 100)
    al, counts, run = run_both(idx, seq, off)
    got = assert_same(al, counts, run, idx)
    assert counts["alignments"] > 10 * counts["mapped"] > 0      # many alleles share every window
    al.close()


def test_index_build_with_gpu_sketches(msa_dir, argannot_index, tmp_path):
    """`groot index` with the window sketches from the device (groot_hip_sketch via the builder's callback, and
    `groot-hip index --gpu 0`): bit-identical to the host-built index"""
    import subprocess

    import __graft_entry__ as g
    from _ffi_empty import empty_view_index

    eng = device.Aligner(empty_view_index(31, 21, 100), max_batch_reads=1 << 16, max_read_len=128)
    idx = host.Index.from_msa_dir(msa_dir, sketcher=eng.sketch)
    eng.close()
    for k, arr in argannot_index.arrays.items():
        assert np.array_equal(arr, idx.arrays[k]), k
    cli = g.build_cli()
    out = tmp_path / "idx"
    r = subprocess.run([cli, "index", "-m", msa_dir, "-i", str(out), "--gpu", "0", "--log", str(tmp_path / "i.log"), "-p", "8"],
                       capture_output=True, timeout=600)
    assert r.returncode == 0, r.stderr
    again = host.Index.load(str(out / "groot.gidx"))
    for k, arr in argannot_index.arrays.items():
        assert np.array_equal(arr, again.arrays[k]), k


def test_edge_cases(small_index, sketch_path):
    cat, o, lens = synth.reference_sequences(small_index)
    seq, off, _ = synth.reads_np(cat, o, lens, 64, 100)
    base = [bytes(seq[int(off[i]):int(off[i + 1])]) for i in range(64)]
    k = small_index.view.kmer_size
    reads = list(base)
    reads[3] = reads[3][:50] + b"N" + reads[3][51:]           # N in the read: matches only a graph N
    reads[4] = b"N" * 100
    reads[5] = reads[5][:k]                                    # exactly one k-mer
    reads[6] = reads[6] + reads[7]                             # 200 bp: longer than the window
    reads[8] = b"ACGT" * 25                                    # low complexity
    reads[9] = reads[9][:99]                                   # ragged lengths
    reads[10] = reads[10][:31] + b"R" + reads[10][32:]         # IUPAC code <= 'T': complement is 0
    seq2, off2 = O.pack_reads(reads)
    al, counts, run = run_both(small_index, seq2, off2)
    assert_same(al, counts, run, small_index)
    # empty batch
    al.submit(np.zeros(0, np.uint8), np.zeros(1, np.uint64))
    c = al.wait()
    assert c["received"] == 0 and len(al.seeds()) == 0 and len(al.travs()[0]) == 0
    al.close()


@pytest.mark.parametrize("threshold", [0.99, 0.9])
def test_mutated_reads_stress(small_index, threshold, sketch_path):
    """substrings of every length 40..160 in both orientations with a wrong first / last / inner base, an N, or both
    ends wrong: every level of the AlignRead hierarchy (and the seed stage's verdict bits that skip levels) is hit"""
    cat, o, lens = synth.reference_sequences(small_index)
    rng = np.random.default_rng(20240 + int(threshold * 100))
    comp = bytes.maketrans(b"ACGTN", b"TGCAN")
    reads = []
    for i in range(12000):
        s = int(rng.integers(0, len(lens)))
        L = int(rng.integers(40, 161))
        if lens[s] < L:
            L = int(lens[s])
        st = int(rng.integers(0, lens[s] - L + 1))
        r = bytearray(cat[int(o[s]) + st:int(o[s]) + st + L].tobytes())
        kind = i % 8
        flip = lambda b: b"ACGT"[(b"ACGT".index(bytes([b])) + 1 + int(rng.integers(0, 3))) % 4] if bytes([b]) in b"ACGT" else ord("A")
        if kind == 1:
            r[0] = flip(r[0])
        elif kind == 2:
            r[-1] = flip(r[-1])
        elif kind == 3:
            r[0] = flip(r[0]); r[-1] = flip(r[-1])
        elif kind == 4:
            j = int(rng.integers(1, L - 1)); r[j] = flip(r[j])
        elif kind == 5:
            r[int(rng.integers(0, L))] = ord("N")
        elif kind == 6:
            r[0] = ord("N")
        r = bytes(r)
        if rng.integers(0, 2):
            r = r.translate(comp)[::-1]
        reads.append(r)
    seq, off = O.pack_reads(reads)
    al, counts, run = run_both(small_index, seq, off, threshold=threshold)
    got = assert_same(al, counts, run, small_index)
    clips = (int(got["start_clip"].sum()), int(got["end_clip"].sum()))
    assert clips[0] > 0 and clips[1] > 0, clips                  # both hard-clip levels produced records
    al.close()


@pytest.mark.parametrize("threshold", [0.99, 0.97, 0.95, 0.90])
def test_mixed_length_threshold_sweep_on_resfinder(resfinder_index, threshold, sketch_path):
    """BASELINE.json configs[4] in miniature: 75-150 bp mixed-length reads of both strands on the larger database,
    containment-threshold sweep (general LSH-Forest path with per-length K/L/min-equal tables)"""
    index = resfinder_index
    cat, o, lens = synth.reference_sequences(index)
    rng = np.random.default_rng(9000 + int(threshold * 100))
    comp = bytes.maketrans(b"ACGTN", b"TGCAN")
    reads = []
    for _ in range(15000):
        s = int(rng.integers(0, len(lens)))
        L = min(int(rng.integers(75, 151)), int(lens[s]))
        st = int(rng.integers(0, lens[s] - L + 1))
        r = cat[int(o[s]) + st:int(o[s]) + st + L].tobytes()
        reads.append(r.translate(comp)[::-1] if rng.integers(0, 2) else r)
    seq, off = O.pack_reads(reads)
    al, counts, run = run_both(index, seq, off, threshold=threshold)
    assert_same(al, counts, run, index)
    assert counts["mapped"] > 1000                              # reads shorter than the window rarely reach t=0.99
    al.close()


def test_packed_submit_equals_byte_submit(small_index, sketch_path):
    """groot_hip_submit_packed (2 bits per base + exception list over PCIe) gives what groot_hip_submit gives"""
    cat, o, lens = synth.reference_sequences(small_index)
    seq, off, _ = synth.reads_np(cat, o, lens, 5000, 100)
    seq = seq.copy()
    rng = np.random.default_rng(3)
    odd = rng.choice(len(seq), 300, replace=False)
    seq[odd] = np.frombuffer(b"NnacgtRY", dtype=np.uint8)[rng.integers(0, 8, 300)]      # bytes <= 'T' or lower case: no panic below
    seq[odd[seq[odd] > ord("T")]] = ord("N")
    al, counts, run = run_both(small_index, seq, off)
    got = assert_same(al, counts, run, small_index)
    sk, sd = (al.sketches().copy() if KEEP_SKETCHES else None), al.seeds().copy()
    al.attempts_reset()
    packed, pos, byte = host.pack_reads(seq)
    assert len(pos) == 300
    al.submit_packed(packed, off, pos, byte)
    c2 = al.wait()
    assert c2 == counts
    assert (sk is None or np.array_equal(al.sketches(), sk)) and np.array_equal(al.seeds(), sd)
    again = al.alns()
    for f in got.dtype.names:
        assert np.array_equal(again[f], got[f]), f
    al.close()


@pytest.mark.parametrize("threshold", [0.99, 0.9])
def test_long_reads_take_the_unstaged_variant(msa_dir, threshold, sketch_path):
    """index with 300-base windows, reads of 300-900 bp (max_read_len 1024): too long for the per-lane LDS slices, so the
    align kernel variant that reads the bases from HBM runs; reads that hang off the graph end included"""
    index = host.Index.from_msa_files(host.msa_files(msa_dir)[:24], host.index_params(w=300))
    cat, o, lens = synth.reference_sequences(index)
    rng = np.random.default_rng(77)
    comp = bytes.maketrans(b"ACGTN", b"TGCAN")
    reads = []
    for i in range(3000):
        s = int(rng.integers(0, len(lens)))
        L = min(300 if i % 3 else int(rng.integers(300, 901)), int(lens[s]))   # window-sized reads seed at any threshold
        st = int(rng.integers(0, lens[s] - L + 1))
        r = cat[int(o[s]) + st:int(o[s]) + st + L].tobytes()
        if i % 7 == 0:
            r = cat[int(o[s]):int(o[s]) + int(lens[s])].tobytes()[-L:] + b"ACGTACGTAC"   # runs past the last node
        reads.append(r.translate(comp)[::-1] if rng.integers(0, 2) else r)
    seq, off = O.pack_reads(reads)
    al, counts, run = run_both(index, seq, off, threshold=threshold, max_read_len=1024)
    assert_same(al, counts, run, index)
    assert counts["mapped"] > 100
    al.close()


@pytest.mark.parametrize("k,sketch,w,y", [(11, 8, 50, 4), (15, 16, 60, 4), (21, 24, 80, 4), (31, 32, 100, 4), (25, 42, 120, 4), (31, 64, 100, 4),
                                            (9, 10, 40, 4), (13, 12, 50, 4), (27, 28, 90, 4), (31, 36, 100, 4), (31, 40, 100, 4), (21, 48, 100, 4),
                                            (31, 50, 100, 4), (31, 56, 110, 4),
                                            # any `groot index -s / -y` (cmd/index.go:45-49): the run-time-sized kernel instance
                                            (31, 25, 100, 2), (31, 25, 100, 8), (21, 7, 60, 3), (31, 21, 100, 1), (31, 21, 100, 7), (25, 100, 90, 5),
                                            (31, 33, 100, 4)])
def test_other_index_parameters(msa_dir, k, sketch, w, y, sketch_path):
    """every compiled sketch size (and k-mer sizes that take the generic multiplier path / leave fewer than 12 bases to the
    prefix tables' second 6-mer) and sizes / hash-functions-per-band without a compiled instance: window-sized and shorter
    reads, both strands, two thresholds"""
    index = host.Index.from_msa_files(host.msa_files(msa_dir)[:8], host.index_params(k=k, s=sketch, w=w, y=y))
    cat, o, lens = synth.reference_sequences(index)
    rng = np.random.default_rng(k * 1000 + sketch)
    comp = bytes.maketrans(b"ACGTN", b"TGCAN")
    reads = []
    for i in range(2500):
        sq = int(rng.integers(0, len(lens)))
        L = min(w if i % 2 else int(rng.integers(max(k, w // 2), w + 30)), int(lens[sq]))
        st = int(rng.integers(0, lens[sq] - L + 1))
        r = bytearray(cat[int(o[sq]) + st:int(o[sq]) + st + L].tobytes())
        if i % 11 == 0:
            r[0] = ord("A") if r[0] != ord("A") else ord("C")
        r = bytes(r)
        reads.append(r.translate(comp)[::-1] if rng.integers(0, 2) else r)
    seq, off = O.pack_reads(reads)
    for t in (0.99, 0.9):
        al, counts, run = run_both(index, seq, off, threshold=t)
        assert_same(al, counts, run, index)
        al.close()


def test_accuracy_fixture(accuracy_index, accuracy_reads, sketch_path):
    """the input of testing/run_accuracy_tests.sh (k=41 s=21 w=150): device == oracle, and groot-accuracy.go's tallies"""
    from conftest import accuracy_stats

    seq, off = O.pack_reads([r[1] for r in accuracy_reads])
    al, counts, run = run_both(accuracy_index, seq, off)
    got = assert_same(al, counts, run, accuracy_index)
    st = accuracy_stats(accuracy_index, accuracy_reads, got)
    assert st == accuracy_stats(accuracy_index, accuracy_reads, run.alns())
    assert st["aligned"] >= 9900 and st["misaligned"] <= 20
    al.close()


def test_error_behaviour_matches_reference_panics(small_index):
    cat, o, lens = synth.reference_sequences(small_index)
    seq, off, _ = synth.reads_np(cat, o, lens, 32, 100)
    reads = [bytes(seq[int(off[i]):int(off[i + 1])]) for i in range(32)]
    # a read shorter than k: NewHasher error -> panic (boss.go:164-166)
    al = device.Aligner(small_index, max_batch_reads=1024)
    s2, o2 = O.pack_reads(reads[:5] + [reads[5][:10]] + reads[6:])
    al.submit(s2, o2)
    with pytest.raises(host.GrootError) as e:
        al.wait()
    assert e.value.code == -7
    with pytest.raises(ValueError):
        run = O.Run(small_index)
        run.batch(s2, o2)
    # lower-case read: sketches like upper case (seedTab), fails forward alignment, RevComplement panics (seqio.go:126)
    strand0 = [i for i in range(32) if True][0]
    s3, o3 = O.pack_reads([reads[strand0].lower()] + reads[1:])
    al.attempts_reset()
    al.submit(s3, o3)
    with pytest.raises(host.GrootError) as e:
        al.wait()
    assert e.value.code == -8
    c = al.wait(check=False)
    run = O.Run(small_index)
    run.batch(s3, o3)
    assert c["revcomp_panics"] == run.counts()["revcomp_panics"] >= 1
    assert np.array_equal(al.seeds(), run.seeds().astype(device.SEED_DTYPE))
    # a read longer than max_read_len is refused, not truncated
    al2 = device.Aligner(small_index, max_batch_reads=1024, max_read_len=120)
    s4, o4 = O.pack_reads(reads[:3] + [reads[3] + reads[4]])
    al2.submit(s4, o4)
    with pytest.raises(host.GrootError) as e:
        al2.wait()
    assert e.value.code == -6
    # too many reads for the ctx
    with pytest.raises(host.GrootError):
        al2.submit(*O.pack_reads(reads * 40))
    al.close(); al2.close()


def test_no_align_mode_and_batch_accumulation(small_index, sketch_path):
    """--noAlign: every seed is weighted, nothing is aligned (graphminion.go:70-72); weights add up over batches"""
    cat, o, lens = synth.reference_sequences(small_index)
    seq, off, _ = synth.reads_np(cat, o, lens, 3000, 100)
    al, counts, run = run_both(small_index, seq, off, no_align=True)
    assert_same(al, counts, run, small_index)
    assert counts["alignments"] == 0 and counts["travs"] == 0
    al.close()
    al = device.Aligner(small_index, max_batch_reads=1024)
    run = O.Run(small_index)
    for b in range(3):
        lo, hi = b * 1000, (b + 1) * 1000
        s = seq[int(off[lo]):int(off[hi])]
        o2 = off[lo:hi + 1] - off[lo]
        al.submit(s, o2, first_read_id=lo)
        al.wait()
        run.batch(s, o2, first_read_id=lo)
    att, oatt = al.attempts(), run.attempts()
    assert np.array_equal(att[: oatt.shape[0]], oatt)
    al.close()


def test_sketch_mirror_of_run_minhash(small_index, genes_index, testgfa_index):
    """groot_hip_sketch = Sequence.RunMinHash(k, s, false, nil) for several (k, s)"""
    rng = np.random.default_rng(7)
    for idx in (small_index, genes_index, testgfa_index):
        k, s = idx.view.kmer_size, idx.view.sketch_size
        seqs = [bytes(rng.choice(list(b"ACGT"), size=int(n)).astype(np.uint8)) for n in rng.integers(k, 250, size=300)]
        seqs += [b"ACGTN" * 20, b"acgtacgtnn" * 10, bytes(range(48, 48 + 80))]
        seqs = [x for x in seqs if len(x) >= k]
        cat, off = O.pack_reads(seqs)
        al = device.Aligner(idx, max_batch_reads=1024)
        got = al.sketch(cat, off)
        for i, x in enumerate(seqs):
            assert np.array_equal(got[i], O.khf_sketch(x, k, s)), (k, s, i)
        al.close()


@pytest.mark.timeout(1200)
def test_full_size_properties(argannot_index):
    """BASELINE configs[2] size (10M x 100 bp) through size-independent properties: determinism, batch-split
    invariance (checksum of checksums), truth containment, and an oracle comparison on a random sample"""
    import torch

    dev = torch.device("cuda", 0)
    cat, o, lens = synth.reference_sequences(argannot_index)
    cat_t, off_t, lens_t = (torch.from_numpy(x).to(dev) for x in (cat, o, lens))
    R, L = 10_000_000, 100
    parts = []
    for c0 in range(0, R, 1_000_000):
        p, _, _ = synth.reads_torch(cat_t, off_t, lens_t, 1_000_000, L, first=c0)
        parts.append(p[: 1_000_000 * L])
    d_seq = torch.zeros(R * L + 64, dtype=torch.uint8, device=dev)
    d_seq[: R * L] = torch.cat(parts)
    del parts
    d_off = torch.arange(0, R + 1, dtype=torch.int64, device=dev) * L
    torch.cuda.synchronize()
    al = device.Aligner(argannot_index, max_batch_reads=R, max_batch_bases=R * L + 64)

    def run(lo, hi):
        # offsets are absolute into d_seq, so a sub-batch is just a window of the offset array
        al.submit_device(d_seq.data_ptr(), d_off.data_ptr() + 8 * lo, hi - lo, first_read_id=lo, max_len=L)
        c = al.wait()
        t, m = al.travs()
        return c, t, m

    c1, t1, m1 = run(0, R)
    att1 = al.attempts()
    al.attempts_reset()
    c2, t2, m2 = run(0, R)
    assert c1 == c2 and np.array_equal(t1, t2) and np.array_equal(m1, m2)          # deterministic
    assert np.array_equal(att1, al.attempts())
    assert c1["received"] == R and c1["mapped"] > 0.99 * R and c1["travs"] == len(t1)
    assert np.all(np.diff(t1["read_id"].astype(np.int64)) >= 0)                    # canonical read order
    # batch-split invariance: 4 unequal batches give the same records and the same call counts
    al.attempts_reset()
    cuts = [0, 1_000_000, 3_500_000, 3_500_001, R]
    ts, ms, tot = [], [], {k: 0 for k in c1}
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        c, t, m = run(lo, hi)
        ts.append(t); ms.append(m)
        for k in tot:
            tot[k] += c[k]
    assert tot == c1
    assert digest(np.concatenate(ts)) == digest(t1) and digest(np.concatenate(ms)) == digest(m1)
    assert np.array_equal(att1, al.attempts())
    # truth containment on the expanded records of a slice
    sl = slice(0, int(np.searchsorted(t1["read_id"], 200_000)))
    recs = device.expand_alns(argannot_index, t1[sl], m1[sl])
    truth = synth.plan_np(lens, 200_000, L)
    have = set(zip(recs["read_id"].tolist(), recs["ref_id"].tolist(), recs["pos"].tolist(), recs["rc"].tolist()))
    hits = sum((i, int(truth[0][i]), int(truth[1][i]), int(truth[2][i])) in have for i in range(200_000))
    assert hits >= 0.995 * 200_000
    # oracle on a random sample of reads
    rng = np.random.default_rng(3)
    pick = np.sort(rng.choice(R, 20_000, replace=False))
    host_seq = d_seq[: R * L].view(R, L)[torch.from_numpy(pick).to(dev)].cpu().numpy().reshape(-1)
    orun = O.Run(argannot_index)
    orun.batch(host_seq, np.arange(0, 20_001, dtype=np.uint64) * L)
    oal = orun.alns()
    full = device.expand_alns(argannot_index, t1, m1) if len(t1) < 20_000_000 else None
    sel = np.isin(full["read_id"], pick)
    got = full[sel]
    remap = np.searchsorted(pick, got["read_id"])
    assert len(got) == len(oal)
    assert np.array_equal(remap, oal["read_id"])
    for f in ("graph_id", "path_id", "ref_id", "pos", "start_clip", "end_clip", "rc", "secondary"):
        assert np.array_equal(got[f], oal[f]), f
    al.close()


def _per_read(t, m, R):
    """records and alignments (paths of the records) per read"""
    cnt = np.bincount(t["read_id"], minlength=R).astype(np.int64)
    aln = np.bincount(t["read_id"], weights=np.bitwise_count(m).sum(axis=1), minlength=R).astype(np.int64)
    return cnt, aln


@pytest.mark.timeout(1800)
@pytest.mark.parametrize("workload", ["c2", "sub1", "mixed99", "mixed90"])
def test_kernel_path_at_benchmark_size(argannot_index, resfinder_index, monkeypatch, workload):
    """The hashing and graph-walk kernels at the size bench.py times them (memo off: GROOT_MEMO_OFF), read by read: the product -- signature
    kernel on a few slots of the sketch, list pass, align kernel -- against the same library with GROOT_NO_SIG=1 (every read through the
    full-width kernel: all S slots at 64 bits, exact table / LSH Forest: an independent seeding path), on 10 M error-free 100 bp reads
    (configs[2]), the same with 1 % substitutions, and 8 M reads of 75..150 bases on resfinder.90 (configs[4]) at t = 0.99 and at t = 0.90 (the
    LSH-Forest branch: lsh_heavy_kernel, the longest list pass); against the same library with GROOT_LEAN=1 (the align stage with its first
    pass, kernels_lean.hpp: every record, path set and call count must be the same); then the oracle on 20 000 of them.  A kernel that is wrong on one read in a million is invisible to oracle comparisons on 10^4..10^5 reads (round 4 had one): at this
    size it shows.  khf.go:35-55, lshe.go:153-175, graphminion.go:46-102, alignment.go:13-254."""
    import torch

    mixed = workload.startswith("mixed")
    threshold = 0.90 if workload == "mixed90" else 0.99
    index = resfinder_index if mixed else argannot_index
    dev = torch.device("cuda", 0)
    cat, o, lens = synth.reference_sequences(index)
    cat_t, off_t, lens_t = (torch.from_numpy(x).to(dev) for x in (cat, o, lens))
    if mixed:
        R = 8_000_000
        d_seq, d_off, _ = synth.reads_torch_mixed(cat_t, off_t, lens_t, R, 150, 75)
        max_len, total = 150, int(d_off[-1].item())
    else:
        R, L = 10_000_000, 100
        parts = []
        for c0 in range(0, R, 1_000_000):
            p, _, _ = synth.reads_torch(cat_t, off_t, lens_t, 1_000_000, L, first=c0)
            parts.append(p[: 1_000_000 * L])
        d_seq = torch.zeros(R * L + 64, dtype=torch.uint8, device=dev)
        d_seq[: R * L] = torch.cat(parts)
        del parts
        d_off = torch.arange(0, R + 1, dtype=torch.int64, device=dev) * L
        max_len, total = L, R * L
        if workload == "sub1":
            g = torch.Generator(device=dev)
            g.manual_seed(0x67726F6F74)
            rows = d_seq[: R * L].view(R, L)
            acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
            for c0 in range(0, R, 1_000_000):
                blk = rows[c0:c0 + 1_000_000]
                hit = torch.rand(blk.shape, generator=g, device=dev) < 0.01
                cur = torch.searchsorted(acgt, torch.where(torch.isin(blk, acgt), blk, acgt[0]).contiguous())
                other = acgt[(cur + 1 + torch.randint(0, 3, blk.shape, generator=g, device=dev)) % 4]
                rows[c0:c0 + 1_000_000] = torch.where(hit & torch.isin(blk, acgt), other, blk)
    torch.cuda.synchronize()
    for v in ("GROOT_NO_TEXT_TABLE", "GROOT_NO_OUTCOME_TABLE", "GROOT_NO_SIG", "GROOT_LEAN"):
        monkeypatch.delenv(v, raising=False)

    def run(no_sig, lean=False):
        for var, on in (("GROOT_NO_SIG", no_sig), ("GROOT_LEAN", lean)):
            if on:
                monkeypatch.setenv(var, "1")
            else:
                monkeypatch.delenv(var, raising=False)
        al = device.Aligner(index, threshold=threshold, max_batch_reads=R, max_read_len=256, max_batch_bases=total + 64, memo_budget_mb=device.MEMO_OFF)
        out = []
        for rep in range(2):                               # twice: the second run must repeat the first
            al.attempts_reset()
            al.submit_device(d_seq.data_ptr(), d_off.data_ptr(), R, first_read_id=0, max_len=max_len, mixed=mixed)
            c = al.wait()
            t, m = al.travs()
            out.append((c, _per_read(t, m, R), al.attempts().copy(), t, m))
        (c0, (n0, a0), att0, t, m), (c1, (n1, a1), att1, _, _) = out
        assert c0 == c1 and np.array_equal(n0, n1) and np.array_equal(a0, a1) and np.array_equal(att0, att1), "a run does not repeat itself"
        al.close()
        return c0, n0, a0, att0, t, m

    c, n, a, att, t, m = run(False)
    # the align stage with its first pass (align_lean_kernel in front of align_kernel; it takes batches of one read length most of whose reads are
    # walked): the same records, path sets, call counts
    cl, _, _, attl, tl, ml = run(False, lean=True)
    assert c["lean_reads"] == 0 and (cl["lean_reads"] > 0.9 * cl["walked_reads"] if workload == "c2" else True), (c, cl)
    assert np.array_equal(t, tl) and np.array_equal(m, ml) and np.array_equal(att, attl)
    assert {k: v for k, v in c.items() if k != "lean_reads"} == {k: v for k, v in cl.items() if k != "lean_reads"}
    del tl, ml
    cf, nf, af, attf, _, _ = run(True)
    assert cf["full_sketch_reads"] == R and (c["full_sketch_reads"] < (0.5 if mixed else 0.2) * R or threshold < 0.99)
    assert c["walked_reads"] == c["mapped"] or mixed or workload == "sub1"     # memo off: nothing is answered from a table
    bad = np.flatnonzero((n != nf) | (a != af))
    assert len(bad) == 0, "%d of %d reads differ between the signature path and the full-width path, first: %s" % (len(bad), R, bad[:10])
    for k in ("received", "mapped", "multimapped", "alignments", "seeds", "travs"):
        assert c[k] == cf[k], k
    assert np.array_equal(att, attf)
    # ... and the oracle on a random sample of the reads
    rng = np.random.default_rng(5)
    pick = np.sort(rng.choice(R, 20_000, replace=False))
    off_h = d_off.cpu().numpy().astype(np.int64)
    lens_h = (off_h[pick + 1] - off_h[pick])
    idx = (np.repeat(off_h[pick], lens_h) + (np.arange(int(lens_h.sum())) - np.repeat(np.cumsum(lens_h) - lens_h, lens_h)))
    host_seq = d_seq[torch.from_numpy(idx).to(dev)].cpu().numpy()
    orun = O.Run(index, threshold)
    orun.batch(host_seq, np.concatenate([[0], np.cumsum(lens_h)]).astype(np.uint64))
    oal = orun.alns()
    sel = np.isin(t["read_id"], pick)
    got = device.expand_alns(index, t[sel], m[sel])
    assert len(got) == len(oal)
    assert np.array_equal(np.searchsorted(pick, got["read_id"]), oal["read_id"])
    for f in ("graph_id", "path_id", "ref_id", "pos", "start_clip", "end_clip", "rc", "secondary"):
        assert np.array_equal(got[f], oal[f]), f


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("threshold", [0.99, 0.95])
def test_configs4_at_single_gpu_scale(resfinder_index, threshold):
    """BASELINE configs[4] at the size one GPU takes: resfinder.90 (card.90 is not in the reference tree), 2 M reads of 75..150
    bases of both strands, two points of the containment-threshold sweep -- through size-independent properties (determinism,
    batch-split invariance of records and call counts, canonical order) and an oracle comparison on 20 000 sampled reads
    (lshe.go:153-175 with per-length K / L / min-equal-slots, graphminion.go:46-102, alignment.go:13-159)"""
    import torch

    index = resfinder_index
    dev = torch.device("cuda", 0)
    cat, o, lens = synth.reference_sequences(index)
    cat_t, off_t, lens_t = (torch.from_numpy(x).to(dev) for x in (cat, o, lens))
    R = 2_000_000
    d_seq, d_off, _ = synth.reads_torch_mixed(cat_t, off_t, lens_t, R, 150, 75)
    total = int(d_off[-1].item())
    torch.cuda.synchronize()
    al = device.Aligner(index, threshold=threshold, max_batch_reads=R, max_batch_bases=total + 64, max_read_len=256)
    keys = ("received", "mapped", "multimapped", "alignments", "seeds", "travs", "revcomp_panics", "short_reads")

    def run(lo, hi):
        al.submit_device(d_seq.data_ptr(), d_off.data_ptr() + 8 * lo, hi - lo, first_read_id=lo, max_len=150, mixed=True)
        c = al.wait()
        t, m = al.travs()
        return {k: c[k] for k in keys}, t, m

    c1, t1, m1 = run(0, R)
    att1 = al.attempts()
    al.attempts_reset()
    c2, t2, m2 = run(0, R)
    assert c1 == c2 and np.array_equal(t1, t2) and np.array_equal(m1, m2)          # deterministic
    assert np.array_equal(att1, al.attempts())
    assert c1["received"] == R and c1["mapped"] > 0.1 * R and c1["travs"] == len(t1)
    assert np.all(np.diff(t1["read_id"].astype(np.int64)) >= 0)                    # canonical read order
    al.attempts_reset()
    cuts = [0, 300_000, 1_100_001, R]
    ts, ms, tot = [], [], {k: 0 for k in keys}
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        c, t, m = run(lo, hi)
        ts.append(t); ms.append(m)
        for k in tot:
            tot[k] += c[k]
    assert tot == c1
    assert digest(np.concatenate(ts)) == digest(t1) and digest(np.concatenate(ms)) == digest(m1)
    assert np.array_equal(att1, al.attempts())
    # oracle on a random sample of reads
    rng = np.random.default_rng(int(threshold * 100))
    pick = np.sort(rng.choice(R, 20_000, replace=False))
    off_h = d_off.cpu().numpy()
    seq_h = d_seq[:total].cpu().numpy()
    reads = [seq_h[off_h[i]:off_h[i + 1]].tobytes() for i in pick]
    s2, o2 = O.pack_reads(reads)
    orun = O.Run(index, threshold)
    orun.batch(s2, o2)
    oal = orun.alns()
    full = device.expand_alns(index, t1, m1)
    got = full[np.isin(full["read_id"], pick)]
    assert len(got) == len(oal) > 1000
    assert np.array_equal(np.searchsorted(pick, got["read_id"]), oal["read_id"])
    for f in ("graph_id", "path_id", "ref_id", "pos", "start_clip", "end_clip", "rc", "secondary"):
        assert np.array_equal(got[f], oal[f]), f
    al.close()


@pytest.mark.parametrize("variant", ["default", "small_buffers"])
def test_reads_with_many_seed_windows(argannot_index, monkeypatch, variant):
    """reads shorter than the windows at a low threshold bring dozens of seed windows, of several graphs: their lists are sorted and
    cut at graph boundaries into items that different lanes of the align stage take (device_types.hpp kSplitMin; records and
    counters put right by split_fix / order_split / order_ovf), and reads with many candidate rows on the LSH-Forest branch get a
    wavefront each (lsh_heavy_kernel).  Seeds, records in (read, ord) order, counters and call counts equal the oracle's
    (lshe.go:153-175, graphminion.go:46-102) -- and equal what the ctx produces when the item list holds 8 items and the heavy list 4
    reads (GROOT_TEST_SMALL_BUFFERS): most such reads then find no room and are handled whole / walk their own rows"""
    monkeypatch.delenv("GROOT_TEST_SMALL_BUFFERS", raising=False)
    if variant == "small_buffers":
        monkeypatch.setenv("GROOT_TEST_SMALL_BUFFERS", "1")
    index = argannot_index
    cat, o, lens = synth.reference_sequences(index)
    seq, off, _ = synth.reads_np(cat, o, lens, 12000, 99, min_len=70)
    al, counts, run = run_both(index, seq, off, threshold=0.9, max_read_len=128)
    per = np.bincount(run.seeds()["read_id"], minlength=12000)
    assert (per > 16).sum() >= 20, "the sample holds too few reads with many seed windows: enlarge it"
    assert_same(al, counts, run, index)
    # the same batch again, twice: nothing of the first pass lingers (items, lists, counters of the split)
    before = al.attempts().copy()
    al.submit(seq, off)
    c2 = al.wait()
    assert c2["alignments"] == counts["alignments"] and c2["multimapped"] == counts["multimapped"]
    assert np.array_equal(al.attempts().astype(np.int64), 2 * before.astype(np.int64))
    al.close()
