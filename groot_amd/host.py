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
    def from_msa_dir(cls, msa_dir, params=None):
        out = C.c_void_p()
        p = params or index_params()
        _check(lib().groot_index_build_msa_dir(msa_dir.encode(), C.byref(p), C.byref(out)))
        return cls(out.value)

    @classmethod
    def load(cls, path):
        out = C.c_void_p()
        _check(lib().groot_index_load(path.encode(), C.byref(out)))
        return cls(out.value)

    def save(self, path):
        _check(lib().groot_index_save(self._h, path.encode()))

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


def window_sketch(seq, k, s):
    out = np.empty(s, dtype=np.uint64)
    b = np.frombuffer(seq, dtype=np.uint8)
    _check(lib().groot_host_window_sketch(_ffi.as_ptr(b, C.c_uint8), C.c_uint32(len(b)), C.c_uint32(k), C.c_uint32(s),
                                          _ffi.as_ptr(out, C.c_uint64)))
    return out


def msa_files(msa_dir):
    """cluster*.msa in filepath.Glob (lexical) order: cmd/index.go:143"""
    return sorted(glob.glob(os.path.join(msa_dir, "cluster*.msa")))
