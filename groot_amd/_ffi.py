"""ctypes declarations of the C ABI in include/*.h (groot_index.h, groot_host.h, groot_hip.h).

This is the Python twin of the cgo stub in INTEGRATION.md: plain pointers and sizes only.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(_HERE)
BUILD_DIR = os.path.join(REPO, "build")

u8p, u32p, u64p, f64p = (C.POINTER(t) for t in (C.c_uint8, C.c_uint32, C.c_uint64, C.c_double))


class IndexView(C.Structure):
    """groot_index_view (include/groot_index.h)"""

    _scalars32 = ["kmer_size", "sketch_size", "window_size", "num_part", "max_k", "num_window_kmers", "path_words",
                  "reserved0", "n_graphs", "n_nodes", "n_edges", "n_paths", "n_windows", "reserved1"]
    _scalars64 = ["n_bases", "n_np", "n_cn", "n_wref", "n_name_bytes"]
    _arrays = [("graph_node_off", u32p), ("graph_path_off", u32p), ("graph_masked", u8p), ("node_seg_id", u32p),
               ("node_seq_off", u32p), ("node_edge_off", u32p), ("node_np_off", u32p), ("node_mask", u64p),
               ("bases", u8p), ("edges", u32p), ("np_path", u32p), ("np_pos", u32p), ("path_len", u32p),
               ("path_name_off", u32p), ("path_names", u8p), ("win_graph", u32p), ("win_node", u32p),
               ("win_offset", u32p), ("win_merge_span", u32p), ("win_cn_off", u32p), ("cn_node", u32p),
               ("cn_count", u32p), ("win_ref_off", u32p), ("win_ref", u32p), ("win_sketch", u64p)]
    _fields_ = ([(n, C.c_uint32) for n in _scalars32] + [(n, C.c_uint64) for n in _scalars64] + _arrays)


class IndexParams(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("kmer_size", "sketch_size", "window_size", "num_part", "max_k",
                                            "max_sketch_span", "n_threads", "reserved")]


def _np_view(ptr, n, dtype):
    """zero-copy numpy view of a C array (borrowed; keep the owner alive)."""
    if n == 0:
        return np.zeros(0, dtype=dtype)
    addr = C.cast(ptr, C.c_void_p).value
    buf = (C.c_char * (int(n) * np.dtype(dtype).itemsize)).from_address(addr)
    return np.frombuffer(buf, dtype=dtype)


def view_arrays(v):
    """dict of numpy views over every array of an IndexView."""
    s = v.sketch_size
    sizes = {
        "graph_node_off": v.n_graphs + 1, "graph_path_off": v.n_graphs + 1, "graph_masked": v.n_graphs,
        "node_seg_id": v.n_nodes, "node_seq_off": v.n_nodes + 1, "node_edge_off": v.n_nodes + 1,
        "node_np_off": v.n_nodes + 1, "node_mask": v.n_nodes * v.path_words, "bases": v.n_bases, "edges": v.n_edges,
        "np_path": v.n_np, "np_pos": v.n_np, "path_len": v.n_paths, "path_name_off": v.n_paths + 1,
        "win_graph": v.n_windows, "win_node": v.n_windows, "win_offset": v.n_windows, "win_merge_span": v.n_windows,
        "win_cn_off": v.n_windows + 1, "cn_node": v.n_cn, "cn_count": v.n_cn, "win_ref_off": v.n_windows + 1,
        "win_ref": v.n_wref, "win_sketch": v.n_windows * s, "path_names": v.n_name_bytes,
    }
    dt = {u8p: np.uint8, u32p: np.uint32, u64p: np.uint64}
    out = {}
    for name, ctype in IndexView._arrays:
        out[name] = _np_view(getattr(v, name), sizes[name], dt[ctype])
    return out


def lib_path(name):
    return os.path.join(BUILD_DIR, name)


def as_ptr(arr, ctype):
    return arr.ctypes.data_as(C.POINTER(ctype))
