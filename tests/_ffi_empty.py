"""an index view without graphs: a ctx opened on it is a pure RunMinHash engine (used to build indexes on the GPU)"""
from groot_amd._ffi import IndexView


class _Empty:
    def __init__(self, view):
        self.view = view


def empty_view_index(k, s, w, x=8, y=4):
    v = IndexView()
    v.kmer_size, v.sketch_size, v.window_size, v.num_part, v.max_k = k, s, w, x, y
    v.num_window_kmers = w - k + 1
    v.path_words = 1
    return _Empty(v)
