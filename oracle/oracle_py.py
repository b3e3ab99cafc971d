"""ctypes loader for the CPU oracle (oracle/_build/libgroot_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg -- never by the groot_amd package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_DIR, "_build", "libgroot_oracle.so")


class OracleAln(C.Structure):
    _fields_ = [("read_id", C.c_uint32), ("graph_id", C.c_uint32), ("path_id", C.c_uint32), ("ref_id", C.c_uint32),
                ("pos", C.c_uint32), ("start_clip", C.c_uint8), ("end_clip", C.c_uint8), ("rc", C.c_uint8),
                ("secondary", C.c_uint8)]


ALN_DTYPE = np.dtype([("read_id", "<u4"), ("graph_id", "<u4"), ("path_id", "<u4"), ("ref_id", "<u4"), ("pos", "<u4"),
                      ("start_clip", "u1"), ("end_clip", "u1"), ("rc", "u1"), ("secondary", "u1")])
SEED_DTYPE = np.dtype([("read_id", "<u4"), ("window_id", "<u4")])


class OracleCounts(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("received", "mapped", "multimapped", "alignments", "seeds", "revcomp_panics")]


def build():
    src = [os.path.join(_DIR, f) for f in ("groot_oracle.c", "groot_oracle.h")]
    if os.path.exists(_SO) and all(os.path.getmtime(_SO) >= os.path.getmtime(s) for s in src if os.path.exists(s)):
        return _SO
    subprocess.check_call(["make", "-C", _DIR, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        L.oracle_containment.restype = C.c_double
        L.oracle_lshe_build.restype = C.c_void_p
        L.oracle_lshe_free.argtypes = [C.c_void_p]
        L.oracle_run_new.restype = C.c_void_p
        L.oracle_run_new.argtypes = [C.c_void_p, C.c_double, C.c_int]
        L.oracle_run_free.argtypes = [C.c_void_p]
        L.oracle_run_drop_records.argtypes = [C.c_void_p]
        L.oracle_run_drop_records.restype = None
        L.oracle_run_seeds.restype = C.c_uint64
        L.oracle_run_alns.restype = C.c_uint64
        L.oracle_run_sketches.restype = C.c_uint64
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def khf_sketch(seq, k, s):
    b = np.frombuffer(bytes(seq), dtype=np.uint8)
    out = np.empty(s, dtype=np.uint64)
    rc = lib().oracle_khf_sketch(_p(b, C.c_uint8), C.c_uint32(len(b)), C.c_uint32(k), C.c_uint32(s), _p(out, C.c_uint64))
    if rc:
        raise ValueError("k size is greater than sequence length")
    return out


def nthash_canonical(seq, k):
    b = np.frombuffer(bytes(seq), dtype=np.uint8)
    out = np.empty(max(0, len(b) - k + 1), dtype=np.uint64)
    if lib().oracle_nthash_canonical(_p(b, C.c_uint8), C.c_uint32(len(b)), C.c_uint32(k), _p(out, C.c_uint64)):
        raise ValueError("k size is greater than sequence length")
    return out


def revcomp(seq, qual=None):
    s = np.frombuffer(bytes(seq), dtype=np.uint8).copy()
    q = np.frombuffer(bytes(qual), dtype=np.uint8).copy() if qual is not None else None
    panics = lib().oracle_revcomp(_p(s, C.c_uint8), _p(q, C.c_uint8) if q is not None else None, C.c_uint32(len(s)))
    return bytes(s), (bytes(q) if q is not None else None), panics


def optimal_kl(max_k, max_l, x, q, t):
    k, l = C.c_int(), C.c_int()
    lib().oracle_optimal_kl(max_k, max_l, x, q, C.c_double(t), C.byref(k), C.byref(l))
    return k.value, l.value


def containment(q, x, q_size, x_size):
    q = np.ascontiguousarray(q, dtype=np.uint64)
    x = np.ascontiguousarray(x, dtype=np.uint64)
    return lib().oracle_containment(_p(q, C.c_uint64), _p(x, C.c_uint64), len(q), q_size, x_size)


def align_read(index, read, window, rc=False):
    """AlignRead (alignment.go:13-159) of one oriented read against one seed window"""
    b = np.frombuffer(bytes(read), dtype=np.uint8)
    cap = 4096
    out = np.zeros(cap, dtype=ALN_DTYPE)
    n = lib().oracle_align_read(C.cast(C.byref(index.view), C.c_void_p), _p(b, C.c_uint8), C.c_uint32(len(b)),
                                C.c_int(1 if rc else 0), C.c_uint32(window), out.ctypes.data_as(C.c_void_p), C.c_uint32(cap))
    assert n <= cap
    return out[:n].copy()


class Lshe:
    def __init__(self, sketches, s, num_part, max_k, num_window_kmers):
        self.sk = np.ascontiguousarray(sketches, dtype=np.uint64).reshape(-1)
        self.s = s
        self.h = C.c_void_p(lib().oracle_lshe_build(_p(self.sk, C.c_uint64), C.c_uint32(len(self.sk) // s), C.c_uint32(s),
                                                    C.c_uint32(num_part), C.c_uint32(max_k), C.c_uint32(num_window_kmers)))

    def query(self, sig, query_size, threshold):
        sig = np.ascontiguousarray(sig, dtype=np.uint64)
        cap = 4096
        out = np.empty(cap, dtype=np.uint32)
        n = lib().oracle_lshe_query(self.h, _p(sig, C.c_uint64), C.c_int(query_size), C.c_double(threshold),
                                    _p(out, C.c_uint32), C.c_uint32(cap))
        assert n <= cap
        return out[:n].copy()

    def __del__(self):
        if self.h and _lib is not None:
            _lib.oracle_lshe_free(self.h)
            self.h = None


def pack_reads(seqs):
    """list of bytes -> (concat uint8, offsets uint64[n+1])"""
    off = np.zeros(len(seqs) + 1, dtype=np.uint64)
    if seqs:
        off[1:] = np.cumsum([len(s) for s in seqs], dtype=np.uint64)
    cat = np.frombuffer(b"".join(seqs), dtype=np.uint8).copy() if seqs else np.zeros(0, dtype=np.uint8)
    return cat, off


class Run:
    """oracle_run: the whole align path on the CPU for a flat index view (any object exposing the
    groot_index_view ctypes struct as .view)."""

    def __init__(self, index, threshold=0.99, no_align=False):
        self.index = index  # keep the arrays alive
        self.h = C.c_void_p(lib().oracle_run_new(C.cast(C.byref(index.view), C.c_void_p), C.c_double(threshold),
                                                 C.c_int(1 if no_align else 0)))
        self.n_windows = index.view.n_windows
        self.s = index.view.sketch_size

    def batch(self, cat, off, first_read_id=0):
        cat = np.ascontiguousarray(cat, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        rc = lib().oracle_run_batch(self.h, _p(cat, C.c_uint8), _p(off, C.c_uint64), C.c_uint32(len(off) - 1),
                                    C.c_uint32(first_read_id))
        if rc:
            raise ValueError("read shorter than k (reference panics, boss.go:164-166)")

    def drop_records(self):
        """streaming use: forget the per-read outputs collected so far (counters and call counts stay)"""
        lib().oracle_run_drop_records(self.h)

    def counts(self):
        c = OracleCounts()
        lib().oracle_run_counts(self.h, C.byref(c))
        return {n: getattr(c, n) for n, _ in OracleCounts._fields_}

    def _arr(self, fn, dtype, mult=1):
        ptr = C.c_void_p()
        n = fn(self.h, C.byref(ptr))
        if not n:
            return np.zeros(0, dtype=dtype)
        buf = (C.c_char * (int(n) * np.dtype(dtype).itemsize)).from_address(ptr.value)
        return np.frombuffer(buf, dtype=dtype).copy()

    def seeds(self):
        return self._arr(lib().oracle_run_seeds, SEED_DTYPE)

    def alns(self):
        return self._arr(lib().oracle_run_alns, ALN_DTYPE)

    def sketches(self):
        return self._arr(lib().oracle_run_sketches, np.uint64).reshape(-1, self.s)

    def attempts(self):
        ptr = C.c_void_p()
        nq = lib().oracle_run_attempts(self.h, C.byref(ptr))
        if not nq:
            return np.zeros((0, self.n_windows), dtype=np.uint32)
        buf = (C.c_char * (int(nq) * self.n_windows * 4)).from_address(ptr.value)
        return np.frombuffer(buf, dtype=np.uint32).copy().reshape(nq, self.n_windows)

    def weights(self, order=1):
        v = self.index.view
        kf = np.zeros(v.n_nodes, dtype=np.float64)
        kt = np.zeros(v.n_graphs, dtype=np.uint64)
        lib().oracle_run_weights(self.h, C.c_int(order), _p(kf, C.c_double), _p(kt, C.c_uint64))
        return kf, kt

    def prune(self, kmer_freq, min_cov=1.0):
        v = self.index.view
        gk = np.zeros(v.n_graphs, dtype=np.uint8)
        pk = np.zeros(v.n_paths, dtype=np.uint8)
        nr = np.zeros(v.n_nodes, dtype=np.uint8)
        kf = np.ascontiguousarray(kmer_freq, dtype=np.float64)
        lib().oracle_prune(C.cast(C.byref(v), C.c_void_p), _p(kf, C.c_double), C.c_double(min_cov), _p(gk, C.c_uint8),
                           _p(pk, C.c_uint8), _p(nr, C.c_uint8))
        return gk, pk, nr

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.oracle_run_free(self.h)
            self.h = None
