"""Python mirror of libgroot_host.so (include/groot_host.h): index build/load/save and the
host steps either side of the device path.  Names follow the reference's packages
(src/pipeline/index.go, src/graph, src/lshe)."""
import ctypes as C
import glob
import os

import numpy as np

from . import _ffi
from ._ffi import IndexParams, IndexView


class GrootError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"[{code}] {msg}")
        self.code = code


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = _ffi.lib_path("libgroot_host.so")
        if not os.path.exists(path):
            raise ImportError(f"{path} missing: run `python -c 'import __graft_entry__ as g; g.build()'` first")
        L = C.CDLL(path)
        L.groot_host_last_error.restype = C.c_char_p
        L.groot_host_version.restype = C.c_char_p
        L.groot_index_free.argtypes = [C.c_void_p]
        L.groot_index_free.restype = None
        L.groot_index_get_view.argtypes = [C.c_void_p, C.POINTER(IndexView)]
        L.groot_index_get_view.restype = None
        _lib = L
    return _lib


def _check(rc):
    if rc < 0:
        raise GrootError(rc, lib().groot_host_last_error().decode(errors="replace"))
    return rc


def index_cache_path(stem):
    """build/<stem>.<key>.gidx, key = hash of the index-builder sources and the view layout: a cached index never
    outlives the code that built it (windowing / merge quirks / file format)."""
    import hashlib

    h = hashlib.sha1()
    for rel in ("groot_amd/csrc/host/index.cpp", "groot_amd/csrc/host/host_common.hpp", "include/groot_index.h"):
        with open(os.path.join(_ffi.REPO, rel), "rb") as f:
            h.update(f.read())
    return os.path.join(_ffi.BUILD_DIR, f"{stem}.{h.hexdigest()[:12]}.gidx")


def index_params(k=31, s=21, w=100, x=8, y=4, max_sketch_span=30, threads=0):
    """defaults of `groot index` (cmd/index.go:45-50)"""
    return IndexParams(k, s, w, x, y, max_sketch_span, threads, 0)


class Index:
    """Owns a groot_index handle; .view is the groot_index_view, .arrays numpy views of it."""

    def __init__(self, handle):
        self._h = C.c_void_p(handle)
        self.view = IndexView()
        lib().groot_index_get_view(self._h, C.byref(self.view))
        self.arrays = _ffi.view_arrays(self.view)

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.groot_index_free(self._h)
            self._h = None

    # ---- constructors -------------------------------------------------------------------
    @staticmethod
    def _build(fn, files, params):
        arr = (C.c_char_p * len(files))(*[f.encode() for f in files])
        out = C.c_void_p()
        _check(fn(arr, C.c_uint32(len(files)), C.byref(params), C.byref(out)))
        return Index(out.value)

    @classmethod
    def from_msa_files(cls, files, params=None):
        return cls._build(lib().groot_index_build_msa_files, list(files), params or index_params())

    @classmethod
    def from_gfa_files(cls, files, params=None):
        return cls._build(lib().groot_index_build_gfa_files, list(files), params or index_params())

    @classmethod
    def from_msa_dir(cls, msa_dir, params=None, sketcher=None):
        """sketcher: optional callable (seq_concat uint8[], seq_off uint64[n+1]) -> uint64[n, s] that computes
        the window sketches (e.g. device.Aligner.sketch): groot_index_build_msa_dir_with"""
        out = C.c_void_p()
        p = params or index_params()
        if sketcher is None:
            _check(lib().groot_index_build_msa_dir(msa_dir.encode(), C.byref(p), C.byref(out)))
            return cls(out.value)
        s = p.sketch_size
        FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint8), C.POINTER(C.c_uint64), C.c_uint32, C.POINTER(C.c_uint64))

        def cb(_user, seq, off, n, dst):
            try:
                o = np.ctypeslib.as_array(off, shape=(n + 1,)).copy()
                sq = np.ctypeslib.as_array(seq, shape=(int(o[n]),)).copy()
                res = np.ascontiguousarray(sketcher(sq, o), dtype=np.uint64)
                np.ctypeslib.as_array(dst, shape=(n * s,))[:] = res.reshape(-1)
                return 0
            except Exception:
                return -1

        fn = FN(cb)
        _check(lib().groot_index_build_msa_dir_with(msa_dir.encode(), C.byref(p), fn, None, C.byref(out)))
        return cls(out.value)

    @classmethod
    def load(cls, path):
        out = C.c_void_p()
        _check(lib().groot_index_load(path.encode(), C.byref(out)))
        return cls(out.value)

    @classmethod
    def load_gob(cls, index_dir):
        """an index directory written by the reference's `groot index`: groot.gg + groot.lshe (cmd/align.go:93-107)"""
        out = C.c_void_p()
        _check(lib().groot_index_load_gob(os.path.join(index_dir, "groot.gg").encode(), os.path.join(index_dir, "groot.lshe").encode(),
                                          C.byref(out)))
        return cls(out.value)

    def save(self, path):
        _check(lib().groot_index_save(self._h, path.encode()))

    def save_gob(self, index_dir, max_sketch_span=30):
        """groot.gg + groot.lshe as the reference's `groot index` writes them (cmd/index.go:130-131)"""
        os.makedirs(index_dir, exist_ok=True)
        _check(lib().groot_index_save_gob(self._h, index_dir.encode(), C.c_uint32(max_sketch_span)))

    # ---- convenience accessors ------------------------------------------------------------
    def path_name(self, global_path):
        a = self.arrays
        o0, o1 = int(a["path_name_off"][global_path]), int(a["path_name_off"][global_path + 1])
        return bytes(a["path_names"][o0:o1]).decode()

    def node_seq(self, node):
        a = self.arrays
        return bytes(a["bases"][int(a["node_seq_off"][node]):int(a["node_seq_off"][node + 1])])

    def path_sequence(self, graph, local_path):
        """Graph2Seqs (graph.go:625-644) for one path."""
        a = self.arrays
        n0, n1 = int(a["graph_node_off"][graph]), int(a["graph_node_off"][graph + 1])
        out = []
        for n in range(n0, n1):
            ps = a["np_path"][int(a["node_np_off"][n]):int(a["node_np_off"][n + 1])]
            if local_path in ps:
                out.append(self.node_seq(n))
        return b"".join(out)


def gob_to_json(data):
    """every top-level value of a Go encoding/gob stream, decoded by the reader behind Index.load_gob"""
    import json

    b = np.frombuffer(bytes(data), dtype=np.uint8)
    need = C.c_uint64(0)
    _check(lib().groot_gob_to_json(_ffi.as_ptr(b, C.c_uint8), C.c_uint64(len(b)), None, C.c_uint64(0), C.byref(need)))
    buf = C.create_string_buffer(need.value)
    _check(lib().groot_gob_to_json(_ffi.as_ptr(b, C.c_uint8), C.c_uint64(len(b)), buf, C.c_uint64(need.value), C.byref(need)))
    return json.loads(buf.value.decode())


def window_sketch(seq, k, s):
    out = np.empty(s, dtype=np.uint64)
    b = np.frombuffer(seq, dtype=np.uint8)
    _check(lib().groot_host_window_sketch(_ffi.as_ptr(b, C.c_uint8), C.c_uint32(len(b)), C.c_uint32(k), C.c_uint32(s),
                                          _ffi.as_ptr(out, C.c_uint64)))
    return out


def msa_files(msa_dir):
    """cluster*.msa in filepath.Glob (lexical) order: cmd/index.go:143"""
    return sorted(glob.glob(os.path.join(msa_dir, "cluster*.msa")))


# ---- FASTQ in / BAM + GFA out (host side of `groot align`) ------------------------------------------
class AlnRecord(C.Structure):
    """groot_aln_record"""
    _fields_ = [("name", C.c_char_p), ("name_len", C.c_uint32), ("seq", C.POINTER(C.c_uint8)), ("qual", C.POINTER(C.c_uint8)),
                ("seq_len", C.c_uint32), ("ref_id", C.c_uint32), ("pos", C.c_uint32), ("start_clip", C.c_uint8),
                ("end_clip", C.c_uint8), ("reverse", C.c_uint8), ("secondary", C.c_uint8)]


class FastqReader:
    """DataStreamer + FastqHandler (src/pipeline/sketch.go:41-77,175-238): batches of reads from FASTQ files"""

    def __init__(self, files):
        self._h = C.c_void_p()
        arr = (C.c_char_p * len(files))(*[f.encode() for f in files])
        _check(lib().groot_fastq_open(arr, C.c_uint32(len(files)), C.byref(self._h)))

    def batches(self, max_reads=1 << 16, max_bases=1 << 24, max_name_bytes=1 << 23):
        L = lib()
        L.groot_fastq_next_batch.restype = C.c_int64
        seq = np.empty(max_bases, dtype=np.uint8)
        qual = np.empty(max_bases, dtype=np.uint8)
        names = np.empty(max_name_bytes, dtype=np.uint8)
        soff = np.empty(max_reads + 1, dtype=np.uint64)
        noff = np.empty(max_reads + 1, dtype=np.uint64)
        while True:
            n = L.groot_fastq_next_batch(self._h, C.c_uint32(max_reads), _ffi.as_ptr(seq, C.c_uint8), _ffi.as_ptr(qual, C.c_uint8),
                                         _ffi.as_ptr(soff, C.c_uint64), C.c_uint64(max_bases), names.ctypes.data_as(C.c_char_p),
                                         _ffi.as_ptr(noff, C.c_uint64), C.c_uint64(max_name_bytes))
            _check(int(n))
            if n == 0:
                return
            nb, nn = int(soff[n]), int(noff[n])
            yield {"n": int(n), "seq": seq[:nb].copy(), "qual": qual[:nb].copy(), "seq_off": soff[: n + 1].copy(),
                   "names": names[:nn].copy(), "name_off": noff[: n + 1].copy()}

    def close(self):
        if self._h:
            lib().groot_fastq_close(self._h)
            self._h = C.c_void_p()

    __del__ = close


class ReadsView(C.Structure):
    """groot_reads_view"""
    _fields_ = [("n_reads", C.c_uint32), ("max_len", C.c_uint32), ("n_bases", C.c_uint64), ("n_exc", C.c_uint64),
                ("packed", C.POINTER(C.c_uint8)), ("seq_len", C.POINTER(C.c_uint16)), ("exc_pos", C.POINTER(C.c_uint64)),
                ("exc_byte", C.POINTER(C.c_uint8)), ("text", C.POINTER(C.c_uint8)), ("name_pos", C.POINTER(C.c_uint32)),
                ("name_len", C.POINTER(C.c_uint32)), ("seq_pos", C.POINTER(C.c_uint32)), ("qual_pos", C.POINTER(C.c_uint32)),
                ("qual_len", C.POINTER(C.c_uint32))]


class ParallelReads:
    """groot_reads_*: parallel FASTQ ingest (reader thread per file, parse + 2-bit pack over all cores); batches carry the
    wire format of groot_hip_submit_packed16 and positions into the FASTQ text"""

    def __init__(self, files, threads=0, block_bytes=0, max_batch_reads=0, max_batch_bases=0):
        self._h = C.c_void_p()
        arr = (C.c_char_p * len(files))(*[f.encode() for f in files])
        _check(lib().groot_reads_open(arr, C.c_uint32(len(files)), C.c_uint32(threads), C.c_uint64(block_bytes), C.c_uint32(max_batch_reads),
                                      C.c_uint64(max_batch_bases), C.byref(self._h)))

    def batches(self):
        L = lib()
        L.groot_reads_batch_free.argtypes = [C.c_void_p]
        L.groot_reads_batch_free.restype = None
        L.groot_reads_batch_view.argtypes = [C.c_void_p, C.POINTER(ReadsView)]
        L.groot_reads_batch_view.restype = None
        while True:
            b = C.c_void_p()
            _check(L.groot_reads_next(self._h, C.byref(b)))
            if not b.value:
                return
            v = ReadsView()
            L.groot_reads_batch_view(b, C.byref(v))
            n = v.n_reads
            out = {"n": n, "max_len": v.max_len, "n_bases": int(v.n_bases),
                   "packed": _ffi._np_view(v.packed, (v.n_bases + 3) // 4, np.uint8).copy(),
                   "seq_len": _ffi._np_view(v.seq_len, n, np.uint16).copy(),
                   "exc_pos": _ffi._np_view(v.exc_pos, v.n_exc, np.uint64).copy(),
                   "exc_byte": _ffi._np_view(v.exc_byte, v.n_exc, np.uint8).copy()}
            pos = {k: _ffi._np_view(getattr(v, k), n, np.uint32) for k in ("name_pos", "name_len", "seq_pos", "qual_pos", "qual_len")}
            text_end = max(int((pos["qual_pos"] + pos["qual_len"]).max()), int((pos["seq_pos"] + out["seq_len"]).max())) if n else 0
            text = _ffi._np_view(v.text, text_end, np.uint8)
            out["names"] = [bytes(text[int(p):int(p) + int(l)]) for p, l in zip(pos["name_pos"], pos["name_len"])]
            out["seqs"] = [bytes(text[int(p):int(p) + int(l)]) for p, l in zip(pos["seq_pos"], out["seq_len"])]
            out["quals"] = [bytes(text[int(p):int(p) + int(l)]) for p, l in zip(pos["qual_pos"], pos["qual_len"])]
            L.groot_reads_batch_free(b)
            yield out

    def close(self):
        if self._h:
            lib().groot_reads_close.argtypes = [C.c_void_p]
            lib().groot_reads_close.restype = None
            lib().groot_reads_close(self._h)
            self._h = C.c_void_p()

    __del__ = close


class BamWriter:
    """setupBAM + the record collector of theBoss (src/pipeline/boss.go:45-105,225-240)"""

    def __init__(self, path, index, date=None):
        self._h = C.c_void_p()
        self.index = index
        _check(lib().groot_bam_open(path.encode() if path else None, C.byref(index.view), date.encode() if date else None,
                                    C.byref(self._h)))

    def write(self, alns, batch, first_read_id=0):
        """alns: expanded records (ALN_DTYPE) of reads held in `batch` (a FastqReader batch dict).
        Seq/Qual of a reverse-complemented read are the reverse complement / reverse (seqio.go:120-133);
        a start-clipped record still carries read.Seq[0:seqLen] (alignment.go:120)."""
        comp = np.zeros(256, dtype=np.uint8)
        for a, b in zip(b"ACGTN", b"TGCAN"):
            comp[a] = b
        recs = (AlnRecord * len(alns))()
        keep = []
        for i, a in enumerate(alns):
            r = int(a["read_id"]) - first_read_id
            s0, s1 = int(batch["seq_off"][r]), int(batch["seq_off"][r + 1])
            n0, n1 = int(batch["name_off"][r]), int(batch["name_off"][r + 1])
            seq, qual = batch["seq"][s0:s1], batch["qual"][s0:s1]
            if a["rc"]:
                seq, qual = comp[seq][::-1], qual[::-1]
            seq_len = (s1 - s0) - int(a["start_clip"]) - int(a["end_clip"])
            seq = np.ascontiguousarray(seq[:seq_len])
            qual = np.ascontiguousarray(qual[:seq_len])
            name = bytes(batch["names"][n0:n1])
            keep.append((seq, qual, name))
            recs[i] = AlnRecord(name, len(name), _ffi.as_ptr(seq, C.c_uint8), _ffi.as_ptr(qual, C.c_uint8), seq_len, int(a["ref_id"]),
                                int(a["pos"]), int(a["start_clip"]), int(a["end_clip"]), int(a["rc"]), int(a["secondary"]))
        _check(lib().groot_bam_write(self._h, recs, C.c_uint64(len(alns))))

    def close(self):
        if self._h:
            h, self._h = self._h, C.c_void_p()
            _check(lib().groot_bam_close(h))

    def __del__(self):
        if getattr(self, "_h", None):
            try:
                self.close()
            except Exception:
                pass


def pack_reads(seq_concat, threads=0):
    """2 bits per base + the list of bytes that are not ACGT (groot_host_pack_reads) -> (packed, exc_pos, exc_byte)"""
    seq = np.ascontiguousarray(seq_concat, dtype=np.uint8)
    packed = np.empty((len(seq) + 3) // 4, dtype=np.uint8)
    cap = 1024
    while True:
        pos, byte = np.empty(cap, dtype=np.uint64), np.empty(cap, dtype=np.uint8)
        n = C.c_uint64(0)
        rc = lib().groot_host_pack_reads(_ffi.as_ptr(seq, C.c_uint8), C.c_uint64(len(seq)), _ffi.as_ptr(packed, C.c_uint8),
                                         _ffi.as_ptr(pos, C.c_uint64), _ffi.as_ptr(byte, C.c_uint8), C.c_uint64(cap), C.byref(n),
                                         C.c_uint32(threads))
        if rc == -6 and n.value > cap:      # GROOT_E_NOSPACE: retry with the size it asked for
            cap = int(n.value)
            continue
        _check(rc)
        return packed, pos[: n.value].copy(), byte[: n.value].copy()


def report(bam_path, cov_cutoff=0.97, low_cov=False, out_path=None):
    """`groot report` (src/reporting/reporting.go): list of (name, read count, length, coverage cigar)"""
    import tempfile

    tmp = None
    if out_path is None:
        fd, tmp = tempfile.mkstemp(suffix=".report")
        os.close(fd)
    n = C.c_uint64(0)
    try:
        _check(lib().groot_host_report(bam_path.encode(), C.c_double(cov_cutoff), C.c_int(1 if low_cov else 0), (out_path or tmp).encode(),
                                       C.byref(n)))
        rows = [ln.rstrip("\n").split("\t") for ln in open(out_path or tmp)]
    finally:
        if tmp:
            os.unlink(tmp)
    assert len(rows) == n.value
    return [(r[0], int(r[1]), int(r[2]), r[3]) for r in rows]


def save_gfa(index, graph, kmer_freq, path_kept, node_removed, total_kmers, file_name, timestamp=None):
    """GrootGraph.SaveGraphAsGFA (src/graph/graphio.go:19-112); returns True if a file was written"""
    kf = np.ascontiguousarray(kmer_freq, dtype=np.float64)
    pk = np.ascontiguousarray(path_kept, dtype=np.uint8)
    nr = np.ascontiguousarray(node_removed, dtype=np.uint8)
    written = C.c_int(0)
    _check(lib().groot_host_save_gfa(C.byref(index.view), C.c_uint32(graph), _ffi.as_ptr(kf, C.c_double), _ffi.as_ptr(pk, C.c_uint8),
                                     _ffi.as_ptr(nr, C.c_uint8), C.c_uint64(total_kmers), timestamp.encode() if timestamp else None,
                                     file_name.encode(), C.byref(written)))
    return bool(written.value)
