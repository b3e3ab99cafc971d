"""Round 3: the memo of the pipeline's own results (groot_hip.hip build_outcome_table).  Every WindowSize-mer of every indexed
path goes through the ctx's pipeline once at open; at run time a read that equals such a string is answered from the outcome
table -- found by its bases (text_lookup_kernel) or by its signature (sketch_sig_kernel + sig_info) -- and never reaches the
align stage.  Whatever the route, seeds / records / counters / IncrementSubPath call counts equal the oracle's
(lshe.go:153-175, graphminion.go:46-102, alignment.go:13-159), batch after batch, and equal the ctx opened without the tables."""
import numpy as np
import pytest

from groot_amd import device, synth
from oracle import oracle_py as O
from test_signature_path import mixed_batch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def need_gpu(hip_lib):
    assert device.device_count() > 0, "no MI355X visible: the HIP path has no CPU fallback"


def perfect_batch(index, n, first):
    cat, o, lens = synth.reference_sequences(index)
    seq, off, _ = synth.reads_np(cat, o, lens, n, 100, first=first)
    return seq, off


def check_batch(al, orc_index, seq, off, att_before):
    """one batch through `al`, compared with a fresh oracle run; returns the counts and the call-count table after it"""
    al.submit(seq, off)
    counts = al.wait()
    orc = O.Run(orc_index, 0.99)
    orc.batch(seq, off)
    assert np.array_equal(al.seeds(), orc.seeds().astype(device.SEED_DTYPE))
    got, exp = al.alns(), orc.alns()
    assert len(got) == len(exp) and all(np.array_equal(got[f], exp[f]) for f in exp.dtype.names)
    for k in ("received", "mapped", "multimapped", "alignments", "seeds", "revcomp_panics"):
        assert counts[k] == orc.counts()[k], k
    att = al.attempts().copy()
    oatt = orc.attempts()
    delta = att.astype(np.int64)
    delta[: att_before.shape[0]] -= att_before
    assert np.array_equal(delta[: oatt.shape[0]], oatt) and not delta[oatt.shape[0]:].any()
    return counts, att


@pytest.mark.parametrize("tables", ["all", "no_text", "none"])
def test_batches_of_changing_composition(argannot_index, monkeypatch, tables):
    """error-free reads (the text lookup answers them), then a mixed batch (errors, N, other lengths: the lookup's share drops and
    the ctx goes back to the signature kernel), then error-free reads again: every batch equals the oracle, and the call
    counts accumulate over the batches whichever kernel counted them"""
    for v in ("GROOT_NO_TEXT_TABLE", "GROOT_NO_OUTCOME_TABLE", "GROOT_NO_SIG"):
        monkeypatch.delenv(v, raising=False)
    if tables == "no_text":
        monkeypatch.setenv("GROOT_NO_TEXT_TABLE", "1")
    if tables == "none":
        monkeypatch.setenv("GROOT_NO_OUTCOME_TABLE", "1")
    index = argannot_index
    al = device.Aligner(index, max_batch_reads=8192, max_read_len=128)
    att = np.zeros((0, index.view.n_windows), dtype=np.uint32)
    shares = []
    for b, (seq, off) in enumerate([perfect_batch(index, 6000, 0), mixed_batch(index, 6000, seed=3), mixed_batch(index, 5000, seed=4),
                                    perfect_batch(index, 7000, 50_000), perfect_batch(index, 7000, 90_000)]):
        counts, att = check_batch(al, index, seq, off, att)
        shares.append(counts["full_sketch_reads"] / counts["received"])
    if tables == "all":
        assert shares[0] < 0.03 and shares[-1] < 0.03     # error-free reads: found by their bases, or by their signature
    al.close()


def test_reads_without_any_record_are_tabulated_too(argannot_index, monkeypatch):
    """strings for which nothing is reported (no seed window at all, or seeds and no traversal in either orientation) have an
    entry of their own -- their counters, their IncrementSubPath calls, no record -- instead of a trip through the align stage"""
    monkeypatch.delenv("GROOT_NO_OUTCOME_TABLE", raising=False)
    index = argannot_index
    seq, off = perfect_batch(index, 60000, 0)
    orc = O.Run(index, 0.99)
    orc.batch(seq, off)
    assert np.unique(orc.alns()["read_id"]).size < 60000, "the sample holds no read without records: enlarge it"
    al = device.Aligner(index, max_batch_reads=65536, max_read_len=128)
    counts, _ = check_batch(al, index, seq, off, np.zeros((0, index.view.n_windows), dtype=np.uint32))
    assert counts["full_sketch_reads"] < 0.01 * 60000      # (what is left: reads holding a byte other than ACGT)
    # a second, identical batch: everything doubles in the call-count table
    a1 = al.attempts().copy()
    al.submit(seq, off)
    al.wait()
    assert np.array_equal(al.attempts(), 2 * a1)
    al.close()


def test_small_index_and_small_buffers(small_index, monkeypatch):
    """the capture pass of open and the run-time path with every buffer starting too small (grow-and-redo paths)"""
    monkeypatch.setenv("GROOT_TEST_SMALL_BUFFERS", "1")
    index = small_index
    al = device.Aligner(index, max_batch_reads=4096, max_read_len=128, max_seeds_per_read=1)
    att = np.zeros((0, index.view.n_windows), dtype=np.uint32)
    for seq, off in (perfect_batch(index, 3000, 0), mixed_batch(index, 3000, seed=8), perfect_batch(index, 3000, 7000)):
        _, att = check_batch(al, index, seq, off, att)
    al.close()


def test_reads_through_an_N_of_an_indexed_sequence(argannot_index, monkeypatch):
    """a path string with a few bytes other than ACGT has a text-table entry too (bases with code 0 at those positions, then the
    bytes and their positions: device_types.hpp text_exc_dwords): error-free reads sampled across an N of an indexed sequence are
    answered by the lookup, with the oracle's seeds / records / counters / call counts; the same reads with the N somewhere
    else, or another byte in its place, are not in the table and take the hashing path -- same equality"""
    for v in ("GROOT_NO_TEXT_TABLE", "GROOT_NO_OUTCOME_TABLE", "GROOT_NO_SIG"):
        monkeypatch.delenv(v, raising=False)
    index = argannot_index
    cat, o, lens = synth.reference_sequences(index)
    seq, off, _ = synth.reads_np(cat, o, lens, 400000, 100)
    reads = seq[: 400000 * 100].reshape(-1, 100)
    with_n = reads[(reads == ord("N")).any(axis=1)]
    assert len(with_n) >= 100, "the sample holds too few reads through an N: enlarge it"
    clean = reads[:4000]
    batch = np.concatenate([with_n, clean])
    np.random.default_rng(5).shuffle(batch, axis=0)
    off = np.arange(len(batch) + 1, dtype=np.uint64) * 100
    al = device.Aligner(index, max_batch_reads=8192, max_read_len=128)
    att = np.zeros((0, index.view.n_windows), dtype=np.uint32)
    counts = None
    for _ in range(3):                                     # (the ctx picks the text lookup from the second batch on)
        counts, att = check_batch(al, index, batch.reshape(-1).copy(), off, att)
    assert counts["full_sketch_reads"] == 0, counts
    # the N moved by one base / replaced by another byte: no entry, the hashing path answers
    moved = with_n.copy()
    for row in moved:
        p = int(np.flatnonzero(row == ord("N"))[0])
        row[p], row[(p + 1) % 100] = row[(p + 1) % 100], row[p]
    other = with_n.copy()
    other[other == ord("N")] = ord("R")
    batch2 = np.concatenate([moved, other, clean])
    off2 = np.arange(len(batch2) + 1, dtype=np.uint64) * 100
    counts, att = check_batch(al, index, batch2.reshape(-1).copy(), off2, att)
    assert counts["full_sketch_reads"] >= len(other)
    al.close()


@pytest.mark.parametrize("budget", ["off", "too_small"])
def test_memo_switched_off_by_the_caller_or_by_its_budget(argannot_index, monkeypatch, budget):
    """groot_params.memo_budget_mb: GROOT_MEMO_OFF, or a budget the index's path strings do not fit (1 MiB) -- the ctx opens without
    the memo (groot_open_stats says so), every read goes through the hashing and graph-walk kernels, results equal the oracle's"""
    for v in ("GROOT_NO_TEXT_TABLE", "GROOT_NO_OUTCOME_TABLE", "GROOT_NO_SIG"):
        monkeypatch.delenv(v, raising=False)
    al = device.Aligner(argannot_index, max_batch_reads=8192, max_read_len=128, memo_budget_mb=device.MEMO_OFF if budget == "off" else 1)
    st = al.open_stats()
    assert st["memo_strings"] == 0 and st["memo_entries"] == 0 and st["text_entries"] == 0
    seq, off = perfect_batch(argannot_index, 6000, 4242)
    counts, _ = check_batch(al, argannot_index, seq, off, np.zeros((0, argannot_index.view.n_windows), dtype=np.uint32))
    assert counts["walked_reads"] > 5000          # nothing was answered from a table
    al.close()
    # the default budget holds this index's memo
    al = device.Aligner(argannot_index, max_batch_reads=8192, max_read_len=128)
    assert al.open_stats()["memo_strings"] > 1_000_000
    al.close()


def test_background_open(argannot_index, monkeypatch):
    """groot_hip_open_flags(GROOT_OPEN_BACKGROUND): the ctx takes batches while its prefix tables and signature index are still being
    built (full-width kernel, no verdicts), then with them -- every batch equals the oracle, and the call counts add up across the switch"""
    for v in ("GROOT_NO_TEXT_TABLE", "GROOT_NO_OUTCOME_TABLE", "GROOT_NO_SIG"):
        monkeypatch.delenv(v, raising=False)
    index = argannot_index
    al = device.Aligner(index, max_batch_reads=8192, max_read_len=128, memo_budget_mb=device.MEMO_OFF, background=True)
    att = np.zeros((0, index.view.n_windows), dtype=np.uint32)
    seq, off = perfect_batch(index, 5000, 11)
    c0, att = check_batch(al, index, seq, off, att)                    # (most likely before the tables are there)
    seq2, off2 = mixed_batch(index, 6000, 5)
    c1, att = check_batch(al, index, seq2, off2, att)
    al.open_wait()
    c2, att = check_batch(al, index, seq, off, att)                    # ... and certainly with them: the signature kernel decides most reads
    assert c2["full_sketch_reads"] < 0.2 * 5000 and c2["mapped"] == c0["mapped"] and c2["alignments"] == c0["alignments"]
    c3, att = check_batch(al, index, seq2, off2, att)
    assert c3["alignments"] == c1["alignments"]
    al.close()
    # a ctx that is closed while its background part is still running
    al = device.Aligner(index, max_batch_reads=4096, max_read_len=128, memo_budget_mb=device.MEMO_OFF, background=True)
    al.close()
    # a ctx whose background part is abandoned (groot_hip_open_abandon: the input ended first) keeps answering, through the full-width kernels
    al = device.Aligner(index, max_batch_reads=8192, max_read_len=128, memo_budget_mb=device.MEMO_OFF, background=True)
    al.open_abandon()
    att = np.zeros((0, index.view.n_windows), dtype=np.uint32)
    c4, att = check_batch(al, index, seq, off, att)
    al.open_wait()                                                      # (returns once the builder has stopped; nothing is installed)
    c5, att = check_batch(al, index, seq2, off2, att)
    assert c4["alignments"] == c0["alignments"] and c5["alignments"] == c1["alignments"]
    al.open_abandon()                                                   # (no-op without a build in progress)
    al.close()


@pytest.mark.parametrize("tables", ["all", "no_text"])
def test_call_counts_of_a_stream_with_the_memo_on(argannot_index, monkeypatch, tables):
    """several batches in flight, each a mix of error-free reads (answered from the memo: their IncrementSubPath calls are counted by
    the seed stage's histogram / by order_first_kernel) and reads with an error (walked by the align stage), all of ONE length, i.e.
    one row of the call-count table: the seed stage of batch b+1 (fold_tab_hist_kernel) adds to the cells the align and order stages
    of batch b are adding to on the other stream -- every addition must be atomic.  The table after the stream equals the oracle's
    for the whole input (graphminion.go:60-67)."""
    for v in ("GROOT_NO_TEXT_TABLE", "GROOT_NO_OUTCOME_TABLE", "GROOT_NO_SIG"):
        monkeypatch.delenv(v, raising=False)
    if tables == "no_text":
        monkeypatch.setenv("GROOT_NO_TEXT_TABLE", "1")
    index = argannot_index
    cat, o, lens = synth.reference_sequences(index)
    n_b, per = 10, 40_000
    seq, off, _ = synth.reads_np(cat, o, lens, n_b * per, 100, first=777)
    rows = seq[: n_b * per * 100].reshape(-1, 100).copy()
    rng = np.random.default_rng(21)
    hit = rng.random(rows.shape) < 0.006                   # about every second read holds a substitution
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    code = np.searchsorted(acgt, np.where(np.isin(rows, acgt), rows, ord("A")))
    other = acgt[(code + 1 + rng.integers(0, 3, rows.shape)) % 4]
    rows = np.where(hit & np.isin(rows, acgt), other, rows).astype(np.uint8)
    al = device.Aligner(index, max_batch_reads=per, max_read_len=128, pipeline_depth=3)
    o1 = np.arange(per + 1, dtype=np.uint64) * 100
    pending, mapped = 0, 0
    for b in range(n_b):
        al.submit(rows[b * per:(b + 1) * per].reshape(-1).copy(), o1, first_read_id=b * per)
        pending += 1
        if pending == 3:
            r = al.collect(copy=False)
            mapped += r["counts"]["mapped"]
            al.release(r["ticket"])
            pending -= 1
    while pending:
        r = al.collect(copy=False)
        mapped += r["counts"]["mapped"]
        al.release(r["ticket"])
        pending -= 1
    att = al.attempts().copy()
    al.close()
    orc = O.Run(index, 0.99)
    orc.batch(rows.reshape(-1).copy(), np.arange(n_b * per + 1, dtype=np.uint64) * 100)
    oatt = orc.attempts()
    assert mapped == orc.counts()["mapped"]
    assert np.array_equal(att[: oatt.shape[0]], oatt) and not att[oatt.shape[0]:].any()
