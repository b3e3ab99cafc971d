"""The pipelined boundary (submit / collect-the-oldest, several batches in flight in ONE ctx) and the compact
call-count table, against the one-batch-at-a-time path and the CPU oracle.  The reference seam is the continuous
stream of boss.go:145-203; what must hold is that batching, pipelining and the wire format change nothing."""
import os

import numpy as np
import pytest

from groot_amd import device, host, synth
from oracle import oracle_py as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def need_gpu(hip_lib):
    assert device.device_count() > 0, "no MI355X visible: the HIP path has no CPU fallback"


def make_batches(index, n_batches, n, min_len=None, read_len=100):
    cat, o, lens = synth.reference_sequences(index)
    out = []
    for b in range(n_batches):
        seq, off, _ = synth.reads_np(cat, o, lens, n + 37 * b, read_len, first=b * 100_000, min_len=min_len)
        out.append((seq, off))
    return out


def oracle_of(index, batches, threshold=0.99):
    run = O.Run(index, threshold)
    first = 0
    per = []
    for seq, off in batches:
        before = len(run.alns())
        run.batch(seq, off, first_read_id=first)
        per.append(run.alns()[before:])
        first += len(off) - 1
    return run, per


def wire(seq, off):
    packed, exc_pos, exc_byte = host.pack_reads(seq)
    return packed, np.diff(off.astype(np.int64)).astype(np.uint16), exc_pos, exc_byte


def test_three_in_flight_equal_oracle_and_serial(small_index):
    batches = make_batches(small_index, 7, 1500)
    run, per = oracle_of(small_index, batches)
    al = device.Aligner(small_index, max_batch_reads=4096, pipeline_depth=3)
    got, counts = [], []
    first = 0
    firsts = []
    for i, (seq, off) in enumerate(batches):
        pk, ln, ep, eb = wire(seq, off)
        if i % 3 == 0:
            al.submit_packed16(pk, ln, ep, eb, first_read_id=first)          # the wire format, copied into staging
        elif i % 3 == 1:
            b = al.acquire()                                                  # zero-copy producer side of the same format
            b["packed"][: len(pk)] = pk
            b["seq_len"][: len(ln)] = ln
            b["exc_pos"][: len(ep)] = ep
            b["exc_byte"][: len(eb)] = eb
            al.submit_acquired(b["ticket"], len(ln), len(ep), first_read_id=first)
        else:
            al.submit(seq, off, first_read_id=first)                          # ASCII + u64 offsets
        firsts.append(first)
        first += len(off) - 1
        if al.in_flight()[0] == 3:
            r = al.collect()
            got.append(r); al.release(r["ticket"])
    while al.in_flight()[0]:
        r = al.collect()
        got.append(r); al.release(r["ticket"])
    assert [r["first_read_id"] for r in got] == firsts                        # oldest first
    for r, exp, (seq, off) in zip(got, per, batches):
        assert r["status"] == 0 and r["n_reads"] == len(off) - 1
        recs = device.expand_alns(small_index, r["travs"], r["masks"])
        assert len(recs) == len(exp)
        for f in exp.dtype.names:
            assert np.array_equal(recs[f], exp[f]), f
    tot = {k: sum(r["counts"][k] for r in got) for k in ("received", "mapped", "multimapped", "alignments", "seeds")}
    oc = run.counts()
    assert all(tot[k] == oc[k] for k in tot)
    att, oatt = al.attempts(), run.attempts()
    assert np.array_equal(att[: oatt.shape[0]], oatt) and not att[oatt.shape[0]:].any()
    # a fourth submit without collecting is refused, nothing is lost
    for seq, off in batches[:3]:
        al.submit(seq, off)
    with pytest.raises(host.GrootError) as e:
        al.submit(*batches[3])
    assert e.value.code == -9
    for _ in range(3):
        al.release(al.collect()["ticket"])
    al.close()


def test_held_results_survive_later_batches(small_index):
    """collected batches stay valid until released, whatever runs meanwhile (BAM writer threads work on them)"""
    batches = make_batches(small_index, 4, 2000)
    _, per = oracle_of(small_index, batches[:1])
    al = device.Aligner(small_index, max_batch_reads=4096, pipeline_depth=3)
    al.submit(*batches[0])
    held = al.collect(copy=False)
    for seq, off in batches[1:3]:
        al.submit(seq, off)
    for _ in range(2):
        al.release(al.collect()["ticket"])
    al.submit(*batches[3])
    al.release(al.collect()["ticket"])
    recs = device.expand_alns(small_index, held["travs"], device.unpack_masks(small_index, held["travs"], held["masks"]))   # views of pinned memory
    assert len(recs) == len(per[0]) and all(np.array_equal(recs[f], per[0][f]) for f in per[0].dtype.names)
    al.release(held["ticket"])
    with pytest.raises(host.GrootError):
        al.release(held["ticket"])
    al.close()


def test_call_count_table_has_one_row_per_kmer_count(small_index):
    """mixed read lengths: rows appear on the device as kmerCounts do (more than the initial capacity: the table grows and
    the batch is redone), the export is ascending, and the replay equals the dense one and the oracle's"""
    cat, o, lens = synth.reference_sequences(small_index)
    seq, off, _ = synth.reads_np(cat, o, lens, 6000, 150, min_len=60)
    run = O.Run(small_index, 0.97)
    run.batch(seq, off)
    al = device.Aligner(small_index, threshold=0.97, max_batch_reads=8192, max_read_len=160)
    al.submit(seq, off)
    c = al.wait()
    assert c["seeds"] == run.counts()["seeds"]
    q, rows = al.attempts_rows()
    assert len(q) > 4 and np.all(np.diff(q.astype(np.int64)) > 0)
    oatt = run.attempts()
    dense = al.attempts()
    assert np.array_equal(dense[: oatt.shape[0]], oatt)
    assert np.array_equal(rows, dense[q])
    assert set(q.tolist()) == set(np.nonzero(oatt.any(axis=1))[0].tolist())
    kf, kt = device.weights_rows(small_index, q, rows)
    okf, okt = run.weights(order=1)
    assert np.array_equal(kf, okf) and np.array_equal(kt, okt)
    # a second batch accumulates into the same rows
    al.submit(seq, off)
    al.wait()
    q2, rows2 = al.attempts_rows()
    assert np.array_equal(q2, q) and np.array_equal(rows2, 2 * rows)
    al.close()


def test_allreduce_over_ctxs_and_fixed_layout(small_index):
    """groot_hip_attempts_allreduce: two ctxs (here on one device: summed by a kernel; on distinct devices: RCCL) end up
    with the union layout and the totals; a caller-owned table (the torch tensor bench.py all-reduces) holds the counts"""
    import torch

    cat, o, lens = synth.reference_sequences(small_index)
    a_seq, a_off, _ = synth.reads_np(cat, o, lens, 3000, 100)
    b_seq, b_off, _ = synth.reads_np(cat, o, lens, 2500, 120, first=50_000, min_len=90)
    run = O.Run(small_index)
    run.batch(a_seq, a_off)
    run.batch(b_seq, b_off, first_read_id=3000)
    als = [device.Aligner(small_index, max_batch_reads=4096) for _ in range(2)]
    als[0].submit(a_seq, a_off); als[0].wait()
    als[1].submit(b_seq, b_off); als[1].wait()
    device.attempts_allreduce(als)
    oatt = run.attempts()
    for al in als:
        dense = al.attempts()
        assert np.array_equal(dense[: oatt.shape[0]], oatt)
    q0, r0 = als[0].attempts_rows()
    q1, r1 = als[1].attempts_rows()
    assert np.array_equal(q0, q1) and np.array_equal(r0, r1)
    for al in als:
        al.close()
    # fixed layout in a caller-owned buffer
    al = device.Aligner(small_index, max_batch_reads=4096)
    nw = al.attempts_shape()[1]
    table = torch.zeros(nw, dtype=torch.int32, device="cuda")
    al.attempts_layout([70], table.data_ptr())
    al.submit(a_seq, a_off); al.wait()
    run_a = O.Run(small_index)
    run_a.batch(a_seq, a_off)
    assert np.array_equal(table.cpu().numpy().astype(np.uint32), run_a.attempts()[70])
    # re-laying out the SAME caller-owned buffer (a wider layout, rows move): the counts survive
    wide = torch.zeros(3 * nw, dtype=torch.int32, device="cuda")
    al.attempts_layout([70], wide.data_ptr())
    assert np.array_equal(wide[:nw].cpu().numpy().astype(np.uint32), run_a.attempts()[70])
    al.attempts_layout([60, 65, 70], wide.data_ptr())
    got = wide.cpu().numpy().astype(np.uint32).reshape(3, nw)
    assert np.array_equal(got[2], run_a.attempts()[70]) and not got[:2].any()
    al.submit(b_seq, b_off)                     # kmerCounts outside the fixed layout: refused, not dropped silently
    with pytest.raises(host.GrootError) as e:
        al.wait()
    assert e.value.code == -6
    al.close()


def test_results_on_device_and_legacy_reads(small_index):
    batches = make_batches(small_index, 2, 1800)
    _, per = oracle_of(small_index, batches[:1])
    al = device.Aligner(small_index, max_batch_reads=4096, results_on_device=True, keep_sketches=True)
    al.submit(*batches[0])
    al.wait()
    recs = al.alns()                           # read_travs copies out of HBM on demand
    assert len(recs) == len(per[0]) and all(np.array_equal(recs[f], per[0][f]) for f in per[0].dtype.names)
    seeds0 = al.seeds()
    assert len(seeds0)
    # seeds / sketches live in one of the two work sets, which batches take in turn: gone once the second newer batch has been submitted
    al.submit(*batches[0]); al.submit(*batches[1]); al.submit(*batches[0])
    al.wait()
    with pytest.raises(host.GrootError) as e:
        al.seeds()
    assert e.value.code == -9
    al.wait()
    assert len(al.seeds())
    al.wait()
    assert np.array_equal(al.seeds(), seeds0)
    al.close()


def test_stream_join_orders_a_caller_stream_behind_the_results(small_index):
    """groot_hip_set_stream puts only the seed stage on the caller's stream; the align and order stages run on a stream of the ctx.  A caller that
    consumes results_on_device records from its own stream orders it with groot_hip_stream_join: the copy below is enqueued on the caller's stream
    right after the submit, without any host wait, and must see the finished records."""
    import torch

    batches = make_batches(small_index, 1, 3000)
    _, per = oracle_of(small_index, batches)
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(dev)
    seq, off = batches[0]
    d_seq = torch.zeros(len(seq) + 64, dtype=torch.uint8, device=dev)
    d_seq[: len(seq)] = torch.from_numpy(np.ascontiguousarray(seq)).to(dev)
    d_off = torch.from_numpy(np.ascontiguousarray(off).astype(np.int64)).to(dev)
    torch.cuda.synchronize()
    al = device.Aligner(small_index, max_batch_reads=4096, results_on_device=True)
    al.set_stream(st.cuda_stream)
    al.stream_join()                                   # (nothing submitted yet: a no-op)
    al.submit_device(d_seq.data_ptr(), d_off.data_ptr(), len(off) - 1, first_read_id=0, max_len=int(np.diff(off).max()))
    al.stream_join()                                   # the caller's stream now waits for the batch's order stage
    with torch.cuda.stream(st):
        marker = torch.ones(1, device=dev)             # work of the caller behind the join
    st.synchronize()                                   # ... so when ITS stream has drained, the batch is through the device
    assert float(marker.item()) == 1.0
    c = al.wait()
    recs = al.alns()
    assert len(recs) == len(per[0]) and all(np.array_equal(recs[f], per[0][f]) for f in per[0].dtype.names) and c["received"] == len(off) - 1
    al.close()


@pytest.mark.parametrize("small", [False, True])
def test_stream_join_covers_the_first_pass_only_and_says_so(small_index, monkeypatch, small):
    """groot_hip_stream_join orders a caller's stream behind a batch's KERNELS.  A batch for which a growable buffer was too small is redone by
    wait / collect on the host (it reads its inputs again, rewrites its results): a consumer ordered by the join alone reads the batch's status word
    (groot_hip_redo_status) behind the join -- non-zero under the mask = not final.  GROOT_TEST_SMALL_BUFFERS makes every buffer start too small."""
    import torch

    monkeypatch.delenv("GROOT_TEST_SMALL_BUFFERS", raising=False)
    if small:
        monkeypatch.setenv("GROOT_TEST_SMALL_BUFFERS", "1")
    batches = make_batches(small_index, 1, 3000)
    _, per = oracle_of(small_index, batches)
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(dev)
    seq, off = batches[0]
    d_seq = torch.zeros(len(seq) + 64, dtype=torch.uint8, device=dev)
    d_seq[: len(seq)] = torch.from_numpy(np.ascontiguousarray(seq)).to(dev)
    d_off = torch.from_numpy(np.ascontiguousarray(off).astype(np.int64)).to(dev)
    torch.cuda.synchronize()
    # (memo off: groot_hip_open then runs no batch of its own, which would have grown the buffers already)
    al = device.Aligner(small_index, max_batch_reads=4096, results_on_device=True, memo_budget_mb=device.MEMO_OFF)
    al.set_stream(st.cuda_stream)
    al.submit_device(d_seq.data_ptr(), d_off.data_ptr(), len(off) - 1, first_read_id=0, max_len=int(np.diff(off).max()))
    al.stream_join()
    ptr, mask = al.redo_status()

    class Word:                                        # the status word as a CUDA array: read on the caller's stream, behind the join
        __cuda_array_interface__ = {"shape": (1,), "typestr": "<i4", "data": (ptr, False), "version": 2}

    with torch.cuda.stream(st):
        status = torch.as_tensor(Word(), device=dev).clone()
    st.synchronize()
    redo = int(status.item()) & mask
    assert (redo != 0) == small, (hex(int(status.item())), hex(mask))
    c = al.wait()                                      # ... which grows what was too small and runs the batch again
    recs = al.alns()
    assert len(recs) == len(per[0]) and all(np.array_equal(recs[f], per[0][f]) for f in per[0].dtype.names) and c["received"] == len(off) - 1
    al.close()


def test_corrupt_view_is_refused(small_index):
    """groot_hip_open runs the consistency pass before uploading anything"""
    import ctypes as C

    from groot_amd._ffi import IndexView

    v = IndexView()
    C.memmove(C.byref(v), C.byref(small_index.view), C.sizeof(IndexView))
    bad = small_index.arrays["edges"].copy()
    bad[0] = 0xFFFFFFF0
    v.edges = bad.ctypes.data_as(C.POINTER(C.c_uint32))

    class Fake:
        view = v

    with pytest.raises(host.GrootError) as e:
        device.Aligner(Fake, max_batch_reads=1024)
    assert e.value.code == -3


def test_grow_and_redo_paths_inside_the_pipeline(small_index, monkeypatch):
    """every buffer starts too small (seed slots, overflow lists, ordered output, call-count rows): each batch is redone after
    the buffers grew, with other batches in flight, and still equals the oracle -- records, path sets, counters, call counts"""
    monkeypatch.setenv("GROOT_TEST_SMALL_BUFFERS", "1")
    cat, o, lens = synth.reference_sequences(small_index)
    batches = []
    for b in range(5):
        seq, off, _ = synth.reads_np(cat, o, lens, 900 + 50 * b, 140, first=b * 77_000, min_len=70)
        batches.append((seq, off))
    run, per = oracle_of(small_index, batches, threshold=0.9)
    al = device.Aligner(small_index, threshold=0.9, max_batch_reads=2048, max_read_len=160, max_seeds_per_read=1, pipeline_depth=3)
    got = []
    first = 0
    for seq, off in batches:
        pk, ln, ep, eb = wire(seq, off)
        al.submit_packed16(pk, ln, ep, eb, first_read_id=first)
        first += len(off) - 1
        if al.in_flight()[0] == 3:
            r = al.collect()
            got.append(r); al.release(r["ticket"])
    while al.in_flight()[0]:
        r = al.collect()
        got.append(r); al.release(r["ticket"])
    for r, exp in zip(got, per):
        recs = device.expand_alns(small_index, r["travs"], r["masks"])
        assert len(recs) == len(exp) == r["counts"]["alignments"]
        for f in exp.dtype.names:
            assert np.array_equal(recs[f], exp[f]), f
    oc = run.counts()
    assert all(sum(r["counts"][k] for r in got) == oc[k] for k in ("received", "mapped", "multimapped", "alignments", "seeds"))
    att, oatt = al.attempts(), run.attempts()
    assert np.array_equal(att[: oatt.shape[0]], oatt) and not att[oatt.shape[0]:].any()
    al.close()


def test_cgo_call_sequence(small_index, tmp_path):
    """cgo/ctest.c makes the calls package groothip makes (cgo/groothip/groothip.go: LoadGob -> Open -> packed16 Submit with three
    batches in flight, caller buffers scribbled over after every submit -> Collect / unpack / Release -> Reopen for a longer
    read -> Weights) on a Go-format index directory; its counters, records and weights equal the Python binding's.
    The seam is theBoss.mapReads, src/pipeline/boss.go:108-242."""
    import json
    import subprocess

    from conftest import REPO
    subprocess.check_call(["make", "-s", "-C", os.path.join(REPO, "cgo")])
    index = small_index
    gob_dir = str(tmp_path / "idx")
    os.makedirs(gob_dir)
    index.save_gob(gob_dir)
    index2 = host.Index.load_gob(gob_dir)                     # (node ids as the gob files give them)
    cat, o, lens = synth.reference_sequences(index2)
    seq, off, _ = synth.reads_np(cat, o, lens, 5000, 100)
    reads = [bytes(seq[int(off[i]):int(off[i + 1])]) for i in range(5000)]
    reads[1234] = reads[1234][:50] + b"N" + reads[1234][51:]
    # one read longer than the ctxs are opened for (128): the harness reopens them, call counts carried over
    longest = max(range(len(lens)), key=lambda i: int(lens[i]))
    reads[4100] = bytes(cat[int(o[longest]):int(o[longest]) + min(300, int(lens[longest]))])
    path = str(tmp_path / "reads.txt")
    with open(path, "wb") as f:
        f.write(b"\n".join(reads) + b"\n")
    batch = 1000
    out = subprocess.run([os.path.join(REPO, "build", "cgo_ctest"), gob_dir, path, str(batch)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert out.returncode == 0, out.stderr[-800:]
    got = json.loads(out.stdout.strip().splitlines()[-1])
    assert got["batches"] == 5 and got["reopened"] == 1

    def mix(h, v):
        h = ((h ^ v) * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        return h ^ (h >> 29)

    al = device.Aligner(index2, max_batch_reads=batch, max_read_len=512)
    tot = dict(received=0, mapped=0, multimapped=0, alignments=0, travs=0, records=0)
    rh = 0
    for b0 in range(0, 5000, batch):
        s2, o2 = O.pack_reads(reads[b0:b0 + batch])
        al.submit(s2, o2, first_read_id=0)
        c = al.wait()
        for k in ("received", "mapped", "multimapped", "alignments"):
            tot[k] += c[k]
        t, m = al.travs()
        tot["travs"] += len(t)
        for i in range(len(t)):
            for w in range(m.shape[1]):
                word = int(m[i, w])
                while word:
                    pid = 64 * w + (word & -word).bit_length() - 1
                    word &= word - 1
                    for v in (int(t["read_id"][i]), int(t["graph_id"][i]), pid, int(t["node"][i]), (int(t["offset"][i]) << 8) | int(t["flags"][i])):
                        rh = mix(rh, v)
                    tot["records"] += 1
    for k, v in tot.items():
        assert got[k] == v, k
    assert got["record_hash"] == "%016x" % rh
    q, counts = al.attempts_rows()
    kf, kt = device.weights_rows(index2, q, counts)
    wh = 0
    for bits in kf.view(np.uint64):
        wh = mix(wh, int(bits))
    assert got["weights_hash"] == "%016x" % wh and got["kmer_total"] == int(kt.sum()) and got["rows"] == len(q)
    al.close()
