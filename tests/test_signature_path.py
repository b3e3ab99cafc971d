"""The signature kernel in front of the full-width sketch kernel (sketch_sig_kernel: top 27 bits of every KHF slot ->
signature table -> confirmation against the window's text; khf.go:35-55, lshe.go:153-175, graph.go:293-333): whatever mix
of reads it is given, seeds / records / counters / call counts equal the full-width kernel's and the oracle's, and it
hands exactly the undecidable reads to the full-width kernel."""
import numpy as np
import pytest

from groot_amd import device, synth
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


@pytest.fixture(scope="module", autouse=True)
def need_gpu(hip_lib):
    assert device.device_count() > 0, "no MI355X visible: the HIP path has no CPU fallback"


def mixed_batch(index, n=6000, seed=11):
    """error-free 100-mers of both strands, plus: one substitution, an N, an IUPAC code (bytes > 'T' are left out: panic),
    99 / 101 / 60-bp reads, a read that is all one base"""
    cat, o, lens = synth.reference_sequences(index)
    seq, off, _ = synth.reads_np(cat, o, lens, n, 100)
    rng = np.random.default_rng(seed)
    reads = [bytearray(seq[int(off[i]):int(off[i + 1])]) for i in range(n)]
    kinds = rng.integers(0, 12, n)
    for i, kind in enumerate(kinds):
        r = reads[i]
        if kind == 0:                                   # substitution somewhere (most lose a minimiser, a few keep all of them)
            p = int(rng.integers(0, len(r)))
            r[p] = b"ACGT"[(b"ACGT".index(r[p]) + 1 + int(rng.integers(0, 3))) % 4]
        elif kind == 1 and i % 3 == 0:
            r[int(rng.integers(0, len(r)))] = ord("N")
        elif kind == 1 and i % 3 == 1:
            r[int(rng.integers(0, len(r)))] = ord("R")  # IUPAC code <= 'T': hashes with seed 0, complements to 0
        elif kind == 2 and i % 4 == 0:
            del r[-1]                                   # 99 bp
        elif kind == 2 and i % 4 == 1:
            r.append(ord("A"))                          # 101 bp
        elif kind == 2 and i % 4 == 2:
            del r[60:]                                  # 60 bp: LSH-Forest branch
        elif kind == 3 and i % 50 == 0:
            r[:] = b"A" * 100
    return O.pack_reads([bytes(r) for r in reads])


def run(index, seq, off, monkeypatch, no_sig, threshold=0.99):
    if no_sig:
        monkeypatch.setenv("GROOT_NO_SIG", "1")
    else:
        monkeypatch.delenv("GROOT_NO_SIG", raising=False)
    al = device.Aligner(index, threshold=threshold, max_batch_reads=max(1024, len(off) - 1), max_read_len=128)
    al.submit(seq, off)
    counts = al.wait()
    out = dict(counts=counts, seeds=al.seeds().copy(), alns=al.alns().copy(), att=al.attempts().copy())
    al.close()
    return out


@pytest.mark.parametrize("which", ["small", "argannot"])
def test_mixed_reads_signature_equals_full_width_equals_oracle(small_index, argannot_index, monkeypatch, which):
    index = small_index if which == "small" else argannot_index
    seq, off = mixed_batch(index)
    n = len(off) - 1
    a = run(index, seq, off, monkeypatch, no_sig=False)
    b = run(index, seq, off, monkeypatch, no_sig=True)
    orc = O.Run(index, 0.99)
    orc.batch(seq, off)
    for got in (a, b):
        assert np.array_equal(got["seeds"], orc.seeds().astype(device.SEED_DTYPE))
        exp = orc.alns()
        assert len(got["alns"]) == len(exp) and all(np.array_equal(got["alns"][f], exp[f]) for f in exp.dtype.names)
        oatt = orc.attempts()
        assert np.array_equal(got["att"][: oatt.shape[0]], oatt)
        for k in ("received", "mapped", "multimapped", "alignments", "seeds", "revcomp_panics"):
            assert got["counts"][k] == orc.counts()[k], k
    assert b["counts"]["full_sketch_reads"] == n
    # the signature kernel keeps the error-free 100-mers (about three quarters of this batch) and passes on the rest
    assert 0.05 * n < a["counts"]["full_sketch_reads"] < 0.45 * n


def test_general_lsh_threshold_goes_to_the_full_width_kernel(small_index, monkeypatch):
    seq, off = mixed_batch(small_index, n=3000, seed=5)
    a = run(small_index, seq, off, monkeypatch, no_sig=False, threshold=0.9)
    b = run(small_index, seq, off, monkeypatch, no_sig=True, threshold=0.9)
    # fewer than all slots must agree: not the signature kernel's case.  What the memo of open knows (error-free window-sized reads)
    # is answered by the text lookup at any threshold; everything else takes the full-width kernel
    assert b["counts"]["full_sketch_reads"] == len(off) - 1 and 0 < a["counts"]["full_sketch_reads"] < len(off) - 1
    assert np.array_equal(a["seeds"], b["seeds"]) and np.array_equal(a["att"], b["att"])
    assert len(a["alns"]) == len(b["alns"]) and all(np.array_equal(a["alns"][f], b["alns"][f]) for f in a["alns"].dtype.names)
    diag = ("full_sketch_reads", "walked_reads", "lean_reads")          # which kernels the reads went through: differs by design
    assert {k: v for k, v in a["counts"].items() if k not in diag} == {k: v for k, v in b["counts"].items() if k not in diag}


def test_reads_with_n_next_to_clean_reads(small_index, monkeypatch):
    """a byte other than ACGT marks its 16-byte chunk: the read and (conservatively) a neighbour sharing the chunk take the
    full-width kernel, everything else stays with the signature kernel -- results unchanged"""
    cat, o, lens = synth.reference_sequences(small_index)
    seq, off, _ = synth.reads_np(cat, o, lens, 2048, 100)
    seq = seq.copy()
    for r in range(0, 2048, 64):
        seq[int(off[r]) + (r % 100)] = ord("N")
    a = run(small_index, seq, off, monkeypatch, no_sig=False)
    b = run(small_index, seq, off, monkeypatch, no_sig=True)
    assert np.array_equal(a["seeds"], b["seeds"]) and np.array_equal(a["att"], b["att"])
    assert all(np.array_equal(a["alns"][f], b["alns"][f]) for f in a["alns"].dtype.names)
    assert 32 <= a["counts"]["full_sketch_reads"] < 300


@pytest.mark.parametrize("k,s,w", [(31, 21, 200), (41, 21, 150), (51, 30, 120), (21, 21, 60), (31, 20, 100), (31, 21, 256)])
def test_every_compiled_instance_and_long_windows(msa_dir, monkeypatch, k, s, w):
    """the (sketch size, k) pairs with a signature kernel, window sizes on both sides of 128 bases (two text widths)"""
    from groot_amd import host

    monkeypatch.delenv("GROOT_NO_SIG", raising=False)
    index = host.Index.from_msa_files(host.msa_files(msa_dir)[:8], host.index_params(k=k, s=s, w=w))
    cat, o, lens = synth.reference_sequences(index)
    rng = np.random.default_rng(k + s + w)
    comp = bytes.maketrans(b"ACGTN", b"TGCAN")
    reads = []
    for i in range(3000):
        sq = int(rng.integers(0, len(lens)))
        L = min(w, int(lens[sq]))
        st = int(rng.integers(0, lens[sq] - L + 1))
        r = bytearray(cat[int(o[sq]) + st:int(o[sq]) + st + L].tobytes())
        if i % 7 == 0:
            p = int(rng.integers(0, L))
            r[p] = ord("A") if r[p] != ord("A") else ord("C")
        r = bytes(r)
        reads.append(r.translate(comp)[::-1] if rng.integers(0, 2) else r)
    seq, off = O.pack_reads(reads)
    al = device.Aligner(index, max_batch_reads=4096, max_read_len=max(256, w))
    al.submit(seq, off)
    counts = al.wait()
    orc = O.Run(index, 0.99)
    orc.batch(seq, off)
    assert np.array_equal(al.seeds(), orc.seeds().astype(device.SEED_DTYPE))
    got, exp = al.alns(), orc.alns()
    assert len(got) == len(exp) and all(np.array_equal(got[f], exp[f]) for f in exp.dtype.names)
    oatt = orc.attempts()
    assert np.array_equal(al.attempts()[: oatt.shape[0]], oatt)
    for f in ("received", "mapped", "multimapped", "alignments", "seeds"):
        assert counts[f] == orc.counts()[f], f
    if w <= 200:    # (256-base windows leave no room for the merged neighbours in a 256-base text: only reads at offset 0 stay)
        assert counts["full_sketch_reads"] < 0.4 * counts["received"]  # the signature kernel decided the error-free reads
    else:
        assert counts["full_sketch_reads"] < counts["received"]
    al.close()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("memo", ["off", "on"])
def test_no_read_is_left_out_by_the_seed_stage(argannot_index, monkeypatch, memo):
    """Every read of every batch is handled by exactly one kernel of the seed stage -- text lookup, signature kernel, list pass, heavy LSH-Forest
    reads --: with GROOT_TEST_POISON the per-read outputs of the seed stage are wiped before each batch, so a read that fell between the kernels
    (round 4: a signature kernel whose workgroup-level list bookkeeping dropped a dozen reads per 10 M from the full-width pass's list) shows as a
    read without seeds.  Without the wipe a stream of equal batches hides it: the work set still holds the right values from its last use, and only
    the FIRST batch through each of the two work sets is wrong (tools/first_use_check.py).  3 M reads per batch, six batches, all alike."""
    for v in ("GROOT_NO_TEXT_TABLE", "GROOT_NO_OUTCOME_TABLE", "GROOT_NO_SIG"):
        monkeypatch.delenv(v, raising=False)
    monkeypatch.setenv("GROOT_TEST_POISON", "1")
    index = argannot_index
    cat, o, lens = synth.reference_sequences(index)
    R = 3_000_000
    seq, off, _ = synth.reads_np(cat, o, lens, R, 100, first=12345)
    al = device.Aligner(index, max_batch_reads=R, max_read_len=128, max_batch_bases=R * 100 + 64, memo_budget_mb=device.MEMO_OFF if memo == "off" else 0)
    ref = None
    for b in range(6):
        al.submit(seq, off)
        c = al.wait()
        t, m = al.travs()
        cnt = np.bincount(t["read_id"], minlength=R)
        if ref is None:
            ref, c0, t0, m0 = cnt, c, t, m
        else:
            bad = np.flatnonzero(cnt != ref)
            assert len(bad) == 0 and c == c0, "batch %d: %d reads differ from batch 0, first: %s" % (b, len(bad), bad[:8])
    al.close()
    # ... and batch 0 itself is right: the oracle on a sample
    pick = np.sort(np.random.default_rng(9).choice(R, 20_000, replace=False))
    rows = seq[: R * 100].reshape(R, 100)[pick].reshape(-1).copy()
    orc = O.Run(index, 0.99)
    orc.batch(rows, np.arange(20_001, dtype=np.uint64) * 100)
    oal = orc.alns()
    sel = np.isin(t0["read_id"], pick)
    got = device.expand_alns(index, t0[sel], m0[sel])
    assert len(got) == len(oal) and np.array_equal(np.searchsorted(pick, got["read_id"]), oal["read_id"])
    for f in ("graph_id", "path_id", "ref_id", "pos", "start_clip", "end_clip", "rc", "secondary"):
        assert np.array_equal(got[f], oal[f]), f
