"""Python mirror of libgroot_hip.so (include/groot_hip.h): the MI355X device path of `groot align`.

Fails loudly when the HIP library or a GPU is missing -- there is no CPU fallback."""
import ctypes as C
import os

import numpy as np

from . import _ffi, host
from ._ffi import IndexView
from .host import GrootError

TRAV_DTYPE = np.dtype([("read_id", "<u4"), ("graph_id", "<u4"), ("node", "<u4"), ("offset", "<u4"), ("ord", "<u2"),
                       ("flags", "u1"), ("reserved", "u1")])
ALN_DTYPE = np.dtype([("read_id", "<u4"), ("graph_id", "<u4"), ("path_id", "<u4"), ("ref_id", "<u4"), ("pos", "<u4"),
                      ("start_clip", "u1"), ("end_clip", "u1"), ("rc", "u1"), ("secondary", "u1")])
SEED_DTYPE = np.dtype([("read_id", "<u4"), ("window_id", "<u4")])
TRAV_RC, TRAV_START_CLIP, TRAV_END_CLIP, TRAV_FIRST = 1, 2, 4, 8


class Params(C.Structure):
    _fields_ = [("containment_threshold", C.c_double), ("no_exact_align", C.c_uint32), ("max_read_len", C.c_uint32),
                ("max_batch_reads", C.c_uint32), ("max_seeds_per_read", C.c_uint32), ("max_batch_bases", C.c_uint64),
                ("keep_sketches", C.c_uint32), ("pipeline_depth", C.c_uint32), ("results_on_device", C.c_uint32),
                ("memo_budget_mb", C.c_uint32)]


MEMO_OFF = 0xFFFFFFFF


class Counts(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("received", "mapped", "multimapped", "alignments", "seeds", "travs",
                                            "revcomp_panics", "short_reads", "full_sketch_reads", "walked_reads", "lean_reads")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class OpenStats(C.Structure):
    _fields_ = [("open_ms", C.c_double), ("memo_ms", C.c_double)] + [(n, C.c_uint64) for n in ("memo_strings", "memo_tabulated", "memo_entries", "text_entries",
                                                                                               "memo_hbm_bytes")]


class StageMs(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("h2d", "sketch_seed", "align", "sort", "total", "schedule", "d2h", "unpack", "first_seed_kernel", "order_kernel", "list_pass", "wall", "lean_pass")]


class BatchBuffers(C.Structure):
    """groot_batch_buffers: the pinned staging of one pipeline slot (groot_hip_acquire)"""
    _fields_ = [("ticket", C.c_uint64), ("packed", C.POINTER(C.c_uint8)), ("seq_len", C.POINTER(C.c_uint16)),
                ("exc_pos", C.POINTER(C.c_uint64)), ("exc_byte", C.POINTER(C.c_uint8)), ("packed_cap", C.c_uint64),
                ("exc_cap", C.c_uint64), ("reads_cap", C.c_uint32), ("reserved", C.c_uint32)]


class BatchResult(C.Structure):
    """groot_batch_result"""
    _fields_ = [("ticket", C.c_uint64), ("first_read_id", C.c_uint32), ("n_reads", C.c_uint32), ("counts", Counts),
                ("travs", C.c_void_p), ("masks", C.c_void_p), ("mask_ckpt", C.c_void_p), ("n_mask_bytes", C.c_uint64),
                ("n_travs", C.c_uint64), ("d_travs", C.c_void_p),
                ("d_masks", C.c_void_p), ("path_words", C.c_uint32), ("status", C.c_int32), ("ms", StageMs)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.environ.get("GROOT_HIP_LIB") or _ffi.lib_path("libgroot_hip.so")   # (GROOT_HIP_LIB: instrumented builds, tools/)
        if not os.path.exists(path):
            raise ImportError(f"{path} missing: the HIP extension was not built (python -c 'import __graft_entry__ as g; g.build()')")
        L = C.CDLL(path)
        L.groot_hip_last_error.restype = C.c_char_p
        L.groot_hip_last_error.argtypes = [C.c_void_p]
        L.groot_hip_close.argtypes = [C.c_void_p]
        L.groot_hip_close.restype = None
        _lib = L
    return _lib


def device_count():
    n = C.c_int(0)
    rc = lib().groot_hip_device_count(C.byref(n))
    return n.value if rc == 0 else 0


def unpack_masks(index, travs, compact_masks):
    """groot_host_unpack_masks: the compact path sets of collect(copy=False) -> [n, path_words]"""
    travs = np.ascontiguousarray(travs, dtype=TRAV_DTYPE)
    cm = np.ascontiguousarray(compact_masks, dtype=np.uint8)
    out = np.zeros((len(travs), index.view.path_words), dtype=np.uint64)
    host._check(host.lib().groot_host_unpack_masks(C.byref(index.view), travs.ctypes.data_as(C.c_void_p), C.c_uint64(len(travs)),
                                                   _ffi.as_ptr(cm, C.c_uint8), _ffi.as_ptr(out, C.c_uint64)))
    return out


def expand_alns(index, travs, masks):
    """groot_host_expand_alns: traversal records -> one tuple per sam.Record."""
    travs = np.ascontiguousarray(travs, dtype=TRAV_DTYPE)
    masks = np.ascontiguousarray(masks, dtype=np.uint64)
    n = C.c_uint64(0)
    H = host.lib()
    rc = H.groot_host_expand_alns(C.byref(index.view), travs.ctypes.data_as(C.c_void_p), _ffi.as_ptr(masks, C.c_uint64),
                                  C.c_uint64(len(travs)), None, C.c_uint64(0), C.byref(n))
    host._check(rc)
    out = np.zeros(n.value, dtype=ALN_DTYPE)
    rc = H.groot_host_expand_alns(C.byref(index.view), travs.ctypes.data_as(C.c_void_p), _ffi.as_ptr(masks, C.c_uint64),
                                  C.c_uint64(len(travs)), out.ctypes.data_as(C.c_void_p), C.c_uint64(len(out)), C.byref(n))
    host._check(rc)
    return out


class Aligner:
    """One groot_ctx: the replacement for theBoss.mapReads (src/pipeline/boss.go:108-242) on one GPU."""

    def __init__(self, index, device=0, threshold=0.99, no_align=False, max_read_len=256, max_batch_reads=1 << 20,
                 max_seeds_per_read=8, keep_sketches=False, max_batch_bases=0, pipeline_depth=0, results_on_device=False, memo_budget_mb=0,
                 background=False):
        self.index = index
        self.params = Params(threshold, 1 if no_align else 0, max_read_len, max_batch_reads, max_seeds_per_read,
                             max_batch_bases, 1 if keep_sketches else 0, pipeline_depth, 1 if results_on_device else 0, memo_budget_mb)
        self._h = C.c_void_p()
        # (background: GROOT_OPEN_BACKGROUND -- the index arrays must outlive the build: self.index holds them)
        rc = lib().groot_hip_open_flags(C.byref(self._h), C.c_int(device), C.byref(index.view), C.byref(self.params), C.c_uint32(1 if background else 0))
        if rc:
            raise GrootError(rc, lib().groot_hip_last_error(None).decode(errors="replace"))
        self.s = index.view.sketch_size
        self.path_words = index.view.path_words

    def close(self):
        if getattr(self, "_h", None) and self._h.value and _lib is not None:
            _lib.groot_hip_close(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def _check(self, rc):
        if rc < 0:
            raise GrootError(rc, lib().groot_hip_last_error(self._h).decode(errors="replace"))
        return rc

    # ---- batch API ----------------------------------------------------------------------------
    def open_wait(self):
        """groot_hip_open_wait: the background part of the open (prefix tables, signature index) is in place"""
        self._check(lib().groot_hip_open_wait(self._h))

    def open_abandon(self):
        """groot_hip_open_abandon: a background open stops at its next checkpoint; the ctx keeps working without what it was building"""
        self._check(lib().groot_hip_open_abandon(self._h))

    def open_stats(self):
        """groot_hip_open_stats: what open built besides the uploaded index (the memo) and how long it took"""
        st = OpenStats()
        self._check(lib().groot_hip_open_stats(self._h, C.byref(st)))
        return {n: getattr(st, n) for n, _ in OpenStats._fields_}

    def set_stream(self, hip_stream):
        self._check(lib().groot_hip_set_stream(self._h, C.c_void_p(hip_stream)))

    def stream_join(self, hip_stream=None):
        """device-side: `hip_stream` (None = the stream of set_stream) waits for the results of every batch submitted so far"""
        self._check(lib().groot_hip_stream_join(self._h, C.c_void_p(hip_stream)))

    def redo_status(self):
        """(device address of the newest batch's status word, mask of its bits that mean `collect will redo the batch`)"""
        p, m = C.c_void_p(), C.c_uint32()
        self._check(lib().groot_hip_redo_status(self._h, C.byref(p), C.byref(m)))
        return p.value, m.value

    def set_profiling(self, on=True):
        self._check(lib().groot_hip_set_profiling(self._h, C.c_int(1 if on else 0)))

    def submit(self, seq_concat, seq_off, first_read_id=0):
        seq = np.ascontiguousarray(seq_concat, dtype=np.uint8)
        off = np.ascontiguousarray(seq_off, dtype=np.uint64)
        self._keep = (seq, off)
        self._check(lib().groot_hip_submit(self._h, _ffi.as_ptr(seq, C.c_uint8), _ffi.as_ptr(off, C.c_uint64),
                                           C.c_uint32(len(off) - 1), C.c_uint32(first_read_id)))

    def submit_packed(self, packed, seq_off, exc_pos, exc_byte, first_read_id=0):
        """groot_hip_submit_packed: the arguments host.pack_reads() builds from seq_concat"""
        pk = np.ascontiguousarray(packed, dtype=np.uint8)
        off = np.ascontiguousarray(seq_off, dtype=np.uint64)
        ep = np.ascontiguousarray(exc_pos, dtype=np.uint64)
        eb = np.ascontiguousarray(exc_byte, dtype=np.uint8)
        self._keep = (pk, off, ep, eb)
        self._check(lib().groot_hip_submit_packed(self._h, _ffi.as_ptr(pk, C.c_uint8), _ffi.as_ptr(off, C.c_uint64), C.c_uint32(len(off) - 1),
                                                  C.c_uint32(first_read_id), _ffi.as_ptr(ep, C.c_uint64), _ffi.as_ptr(eb, C.c_uint8),
                                                  C.c_uint64(len(ep))))

    def submit_packed16(self, packed, seq_len, exc_pos, exc_byte, first_read_id=0):
        """groot_hip_submit_packed16: 2-bit bases + one u16 length per read + exception list (the wire format)"""
        pk = np.ascontiguousarray(packed, dtype=np.uint8)
        ln = np.ascontiguousarray(seq_len, dtype=np.uint16)
        ep = np.ascontiguousarray(exc_pos, dtype=np.uint64)
        eb = np.ascontiguousarray(exc_byte, dtype=np.uint8)
        self._check(lib().groot_hip_submit_packed16(self._h, _ffi.as_ptr(pk, C.c_uint8), _ffi.as_ptr(ln, C.c_uint16), C.c_uint32(len(ln)),
                                                    C.c_uint32(first_read_id), _ffi.as_ptr(ep, C.c_uint64), _ffi.as_ptr(eb, C.c_uint8),
                                                    C.c_uint64(len(ep))))

    def acquire(self):
        """groot_hip_acquire: numpy views of a free slot's pinned staging (packed, seq_len, exc_pos, exc_byte) + ticket"""
        b = BatchBuffers()
        self._check(lib().groot_hip_acquire(self._h, C.byref(b)))
        views = {"ticket": int(b.ticket),
                 "packed": _ffi._np_view(b.packed, b.packed_cap, np.uint8), "seq_len": _ffi._np_view(b.seq_len, b.reads_cap, np.uint16),
                 "exc_pos": _ffi._np_view(b.exc_pos, b.exc_cap, np.uint64), "exc_byte": _ffi._np_view(b.exc_byte, b.exc_cap, np.uint8)}
        return views

    def submit_acquired(self, ticket, n_reads, n_exc=0, first_read_id=0):
        self._check(lib().groot_hip_submit_acquired(self._h, C.c_uint64(ticket), C.c_uint32(n_reads), C.c_uint64(n_exc), C.c_uint32(first_read_id)))

    def collect(self, check=True, copy=True):
        """groot_hip_collect: blocks for the oldest batch.  Returns a dict with counts, ticket, status and the traversal
        records (copies by default; copy=False gives views of the ctx's pinned memory, valid until release(ticket))."""
        r = BatchResult()
        rc = lib().groot_hip_collect(self._h, C.byref(r))
        if rc < 0 and (check or not r.ticket):
            self._check(rc)
        n = int(r.n_travs)
        if r.travs and n:
            t = _ffi._np_view(C.cast(r.travs, C.POINTER(C.c_uint8)), n * TRAV_DTYPE.itemsize, np.uint8).view(TRAV_DTYPE)
            if copy:
                # the compact path sets widened to path_words words per traversal (groot_host_unpack_masks)
                m = np.zeros((n, r.path_words), dtype=np.uint64)
                host._check(host.lib().groot_host_unpack_masks(C.byref(self.index.view), C.c_void_p(r.travs), C.c_uint64(n), C.c_void_p(r.masks),
                                                               _ffi.as_ptr(m, C.c_uint64)))
                t = t.copy()
            else:
                m = _ffi._np_view(C.cast(r.masks, C.POINTER(C.c_uint8)), int(r.n_mask_bytes), np.uint8)   # compact (bytes), as handed out
        else:
            t, m = np.zeros(0, dtype=TRAV_DTYPE), np.zeros((0, self.path_words), dtype=np.uint64)
        return {"ticket": int(r.ticket), "first_read_id": int(r.first_read_id), "n_reads": int(r.n_reads), "counts": r.counts.as_dict(),
                "status": int(r.status), "n_travs": n, "travs": t, "masks": m, "n_mask_bytes": int(r.n_mask_bytes), "d_travs": r.d_travs,
                "d_masks": r.d_masks,
                "ms": {k: float(getattr(r.ms, k)) for k, _ in StageMs._fields_}}

    def release(self, ticket):
        self._check(lib().groot_hip_release(self._h, C.c_uint64(ticket)))

    def in_flight(self):
        a, b = C.c_uint32(), C.c_uint32()
        self._check(lib().groot_hip_in_flight(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def submit_device(self, d_seq_ptr, d_off_ptr, n_reads, first_read_id=0, max_len=0, mixed=False):
        if mixed:
            max_len |= 0x80000000      # GROOT_MAXLEN_MIXED
        self._check(lib().groot_hip_submit_device(self._h, C.c_void_p(d_seq_ptr), C.c_void_p(d_off_ptr), C.c_uint32(n_reads),
                                                  C.c_uint32(first_read_id), C.c_uint32(max_len)))

    def wait(self, check=True):
        c = Counts()
        rc = lib().groot_hip_wait(self._h, C.byref(c))
        self.last_rc = rc
        if check:
            self._check(rc)
        return c.as_dict()

    def seeds(self):
        n = C.c_uint64(0)
        self._check(lib().groot_hip_read_seeds(self._h, None, C.c_uint64(0), C.byref(n)))
        out = np.zeros(n.value, dtype=SEED_DTYPE)
        self._check(lib().groot_hip_read_seeds(self._h, out.ctypes.data_as(C.c_void_p), C.c_uint64(len(out)), C.byref(n)))
        return out

    def travs(self):
        n = C.c_uint64(0)
        self._check(lib().groot_hip_read_travs(self._h, None, None, C.c_uint64(0), C.byref(n)))
        t = np.zeros(n.value, dtype=TRAV_DTYPE)
        m = np.zeros((n.value, self.path_words), dtype=np.uint64)
        self._check(lib().groot_hip_read_travs(self._h, t.ctypes.data_as(C.c_void_p), _ffi.as_ptr(m, C.c_uint64),
                                               C.c_uint64(len(t)), C.byref(n)))
        return t, m

    def alns(self):
        t, m = self.travs()
        return expand_alns(self.index, t, m)

    def sketches(self):
        n = C.c_uint64(0)
        self._check(lib().groot_hip_read_sketches(self._h, None, C.c_uint64(0), C.byref(n)))
        out = np.zeros((n.value, self.s), dtype=np.uint64)
        self._check(lib().groot_hip_read_sketches(self._h, _ffi.as_ptr(out, C.c_uint64), C.c_uint64(n.value), C.byref(n)))
        return out

    def stage_ms(self):
        m = StageMs()
        self._check(lib().groot_hip_stage_ms(self._h, C.byref(m)))
        return {n: float(getattr(m, n)) for n, _ in StageMs._fields_}

    # ---- weights ------------------------------------------------------------------------------
    def attempts_shape(self):
        nq, nw = C.c_uint32(), C.c_uint32()
        self._check(lib().groot_hip_attempts_shape(self._h, C.byref(nq), C.byref(nw)))
        return nq.value, nw.value

    def attempts_device(self):
        """(device pointer of the [rows][n_windows] uint32 call-count table, rows, n_windows)"""
        p, nr, nw = C.c_void_p(), C.c_uint32(), C.c_uint32()
        self._check(lib().groot_hip_attempts_device(self._h, C.byref(p), C.byref(nr), C.byref(nw)))
        return p.value, nr.value, nw.value

    def attempts(self):
        """dense compatibility view [max_read_len-k+2][n_windows] (rows of kmerCounts that never occurred are zero)"""
        nq, nw = self.attempts_shape()
        out = np.zeros((nq, nw), dtype=np.uint32)
        self._check(lib().groot_hip_attempts_read(self._h, _ffi.as_ptr(out, C.c_uint32), C.c_uint64(out.size)))
        return out

    def attempts_rows(self):
        """groot_hip_attempts_export: (kmerCounts ascending, counts[rows][n_windows])"""
        nr, nw = C.c_uint32(), C.c_uint32()
        self._check(lib().groot_hip_attempts_export(self._h, None, None, C.c_uint32(0), C.byref(nr), C.byref(nw)))
        q = np.zeros(nr.value, dtype=np.uint32)
        cnt = np.zeros((nr.value, nw.value), dtype=np.uint32)
        self._check(lib().groot_hip_attempts_export(self._h, _ffi.as_ptr(q, C.c_uint32), _ffi.as_ptr(cnt, C.c_uint32), C.c_uint32(nr.value),
                                                    C.byref(nr), C.byref(nw)))
        return q, cnt

    def attempts_import(self, q_values, counts):
        """groot_hip_attempts_import: add exported rows back (a re-opened ctx carries its counts over)"""
        q = np.ascontiguousarray(q_values, dtype=np.uint32)
        cnt = np.ascontiguousarray(counts, dtype=np.uint32)
        self._check(lib().groot_hip_attempts_import(self._h, _ffi.as_ptr(q, C.c_uint32), _ffi.as_ptr(cnt, C.c_uint32), C.c_uint32(len(q))))

    def attempts_layout(self, q_values, d_table=None):
        """groot_hip_attempts_layout: fix the rows (ascending kmerCounts); d_table = caller-owned device buffer or None"""
        q = np.ascontiguousarray(q_values, dtype=np.uint32)
        self._check(lib().groot_hip_attempts_layout(self._h, _ffi.as_ptr(q, C.c_uint32), C.c_uint32(len(q)), C.c_void_p(d_table or 0)))

    def attempts_reset(self):
        self._check(lib().groot_hip_attempts_reset(self._h))

    # ---- fine-grained mirror of Sequence.RunMinHash ------------------------------------------
    def sketch(self, seq_concat, seq_off):
        seq = np.ascontiguousarray(seq_concat, dtype=np.uint8)
        off = np.ascontiguousarray(seq_off, dtype=np.uint64)
        out = np.zeros((len(off) - 1, self.s), dtype=np.uint64)
        self._check(lib().groot_hip_sketch(self._h, _ffi.as_ptr(seq, C.c_uint8), _ffi.as_ptr(off, C.c_uint64),
                                           C.c_uint32(len(off) - 1), _ffi.as_ptr(out, C.c_uint64)))
        return out


def attempts_allreduce(aligners):
    """groot_hip_attempts_allreduce over the ctxs of one process: RCCL across devices, a kernel within a device"""
    arr = (C.c_void_p * len(aligners))(*[a._h for a in aligners])
    rc = lib().groot_hip_attempts_allreduce(arr, C.c_int(len(aligners)))
    aligners[0]._check(rc)


def weights_rows(index, q_values, counts):
    """groot_host_weights_rows: canonical replay of IncrementSubPath from the per-kmerCount rows of the call-count table"""
    q = np.ascontiguousarray(q_values, dtype=np.uint32)
    cnt = np.ascontiguousarray(counts, dtype=np.uint32)
    v = index.view
    kf = np.zeros(v.n_nodes, dtype=np.float64)
    kt = np.zeros(v.n_graphs, dtype=np.uint64)
    host._check(host.lib().groot_host_weights_rows(C.byref(v), _ffi.as_ptr(q, C.c_uint32), C.c_uint32(len(q)), _ffi.as_ptr(cnt, C.c_uint32),
                                                   _ffi.as_ptr(kf, C.c_double), _ffi.as_ptr(kt, C.c_uint64)))
    return kf, kt


def weights(index, attempts):
    """groot_host_weights: canonical replay of IncrementSubPath from the call counts."""
    att = np.ascontiguousarray(attempts, dtype=np.uint32)
    v = index.view
    kf = np.zeros(v.n_nodes, dtype=np.float64)
    kt = np.zeros(v.n_graphs, dtype=np.uint64)
    host._check(host.lib().groot_host_weights(C.byref(v), _ffi.as_ptr(att, C.c_uint32), C.c_uint32(att.shape[0]),
                                              _ffi.as_ptr(kf, C.c_double), _ffi.as_ptr(kt, C.c_uint64)))
    return kf, kt


def prune(index, kmer_freq, min_cov=1.0):
    v = index.view
    gk = np.zeros(v.n_graphs, dtype=np.uint8)
    pk = np.zeros(v.n_paths, dtype=np.uint8)
    nr = np.zeros(v.n_nodes, dtype=np.uint8)
    kf = np.ascontiguousarray(kmer_freq, dtype=np.float64)
    host._check(host.lib().groot_host_prune(C.byref(v), _ffi.as_ptr(kf, C.c_double), C.c_double(min_cov), _ffi.as_ptr(gk, C.c_uint8),
                                            _ffi.as_ptr(pk, C.c_uint8), _ffi.as_ptr(nr, C.c_uint8)))
    return gk, pk, nr
