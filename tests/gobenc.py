"""Test-side encoder for Go's encoding/gob wire format (independent of the C++ reader it exercises).

Written from the format description in the gob package documentation; `tests/test_gob.py` pins it byte-for-byte on
the worked example printed there (type Point struct{X, Y int}; Point{22, 33}).  Only what the reference's index
files use: bool/int/uint/float64/[]byte/string, structs, slices, maps (src/pipeline/runtime.go:15-28,
src/graph/graph.go:18-27, src/graph/node.go:13-22, src/lshe/lshe.go:17-44)."""
import random
import struct

BOOL, INT, UINT, FLOAT, BYTES, STRING = 1, 2, 3, 4, 5, 6


class Struct:
    def __init__(self, name, fields):
        self.name, self.fields = name, fields


class Slice:
    def __init__(self, name, elem):
        self.name, self.elem = name, elem


class Map:
    def __init__(self, name, key, elem):
        self.name, self.key, self.elem = name, key, elem


def enc_uint(n):
    if n < 128:
        return bytes([n])
    b = n.to_bytes((n.bit_length() + 7) // 8, "big")
    return bytes([256 - len(b)]) + b


def enc_int(i):
    u = ((~i) << 1) | 1 if i < 0 else i << 1
    return enc_uint(u & 0xFFFFFFFFFFFFFFFF)


def enc_float(f):
    return enc_uint(int.from_bytes(struct.pack(">d", float(f)), "little"))


class Encoder:
    """One gob stream: Encoder().encode(type, value) appends the type definitions not sent yet, then the value."""

    def __init__(self, shuffle_seed=None):
        self.out = bytearray()
        self.ids = {}
        self.next_id = 65
        self.rng = random.Random(shuffle_seed) if shuffle_seed is not None else None

    # ---- types ---------------------------------------------------------------------------------
    def _id(self, t):
        if isinstance(t, int):
            return t
        if id(t) not in self.ids:
            self.ids[id(t)] = self.next_id
            self.next_id += 1
            self._send_type(t)
        return self.ids[id(t)]

    def _common(self, name, tid):
        b = bytearray()
        if name:
            b += b"\x01" + enc_uint(len(name)) + name.encode()
            b += b"\x01" + enc_int(tid)
        else:
            b += b"\x02" + enc_int(tid)
        return bytes(b) + b"\x00"

    def _send_type(self, t):
        tid = self.ids[id(t)]
        inner = []
        w = bytearray()
        if isinstance(t, Struct):
            # field ids of inner types must exist before we can write them; Go also numbers before sending
            for _, ft in t.fields:
                if not isinstance(ft, int) and id(ft) not in self.ids:
                    self.ids[id(ft)] = self.next_id
                    self.next_id += 1
                    inner.append(ft)
            w += b"\x03" + b"\x01" + self._common(t.name, tid)
            if t.fields:
                w += b"\x01" + enc_uint(len(t.fields))
                for fn, ft in t.fields:
                    w += b"\x01" + enc_uint(len(fn)) + fn.encode() + b"\x01" + enc_int(ft if isinstance(ft, int) else self.ids[id(ft)]) + b"\x00"
            w += b"\x00"
        else:
            parts = [t.elem] if isinstance(t, Slice) else [t.key, t.elem]
            for ft in parts:
                if not isinstance(ft, int) and id(ft) not in self.ids:
                    self.ids[id(ft)] = self.next_id
                    self.next_id += 1
                    inner.append(ft)
            w += (b"\x02" if isinstance(t, Slice) else b"\x04") + b"\x01" + self._common(t.name, tid)
            for ft in parts:
                w += b"\x01" + enc_int(ft if isinstance(ft, int) else self.ids[id(ft)])
            w += b"\x00"
        w += b"\x00"
        msg = enc_int(-tid) + bytes(w)
        self.out += enc_uint(len(msg)) + msg
        for ft in inner:
            self._send_type(ft)

    # ---- values --------------------------------------------------------------------------------
    @staticmethod
    def _is_zero(t, v):
        if isinstance(t, int):
            return v in (0, False, b"", "", 0.0, None)
        if isinstance(t, Slice):
            return v is None or len(v) == 0
        if isinstance(t, Map):
            return v is None       # an empty non-nil map is sent (count 0)
        return False               # struct-typed fields are always sent

    def _value(self, t, v):
        if t == BOOL:
            return enc_uint(1 if v else 0)
        if t == INT:
            return enc_int(int(v))
        if t == UINT:
            return enc_uint(int(v))
        if t == FLOAT:
            return enc_float(v)
        if t in (BYTES, STRING):
            b = v.encode() if isinstance(v, str) else bytes(v)
            return enc_uint(len(b)) + b
        if isinstance(t, Struct):
            b = bytearray()
            last = -1
            for i, (fn, ft) in enumerate(t.fields):
                fv = v.get(fn)
                if fv is None or self._is_zero(ft, fv):
                    continue
                b += enc_uint(i - last) + self._value(ft, fv)
                last = i
            return bytes(b) + b"\x00"
        if isinstance(t, Slice):
            b = bytearray(enc_uint(len(v)))
            for e in v:
                b += self._value(t.elem, e)
            return bytes(b)
        if isinstance(t, Map):
            items = list(v.items())
            if self.rng:
                self.rng.shuffle(items)   # Go map iteration order is random
            b = bytearray(enc_uint(len(items)))
            for k, e in items:
                b += self._value(t.key, k) + self._value(t.elem, e)
            return bytes(b)
        raise TypeError(t)

    def encode(self, t, v):
        tid = self._id(t)
        body = self._value(t, v)
        if not isinstance(t, Struct):
            body = b"\x00" + body
        msg = enc_int(tid) + body
        self.out += enc_uint(len(msg)) + msg
        return self


# ---- the reference's two index types --------------------------------------------------------------
def info_type():
    nodes = Slice("Nodes", UINT)                                               # node.go:8 type Nodes []uint64
    node = Struct("GrootGraphNode", [("SegmentID", UINT), ("SegmentLength", FLOAT), ("Sequence", BYTES), ("OutEdges", nodes),
                                     ("PathIDs", Slice("[]uint32", UINT)), ("Position", Map("map[int]int", INT, INT)),
                                     ("KmerFreq", FLOAT), ("Marked", BOOL)])
    graph = Struct("GrootGraph", [("GrootVersion", STRING), ("GraphID", UINT), ("SortedNodes", Slice("[]*graph.GrootGraphNode", node)),
                                  ("Paths", Map("map[uint32][]uint8", UINT, BYTES)), ("Lengths", Map("map[uint32]int", UINT, INT)),
                                  ("NodeLookup", Map("map[uint64]int", UINT, INT)), ("Masked", BOOL), ("KmerTotal", UINT),
                                  ("EMiterations", INT)])
    align = Struct("AlignCmd", [("Fasta", BOOL), ("BloomFilter", BOOL), ("MinKmerCoverage", FLOAT), ("BAMout", STRING),
                                ("NoExactAlign", BOOL)])
    haplo = Struct("HaploCmd", [("Cutoff", FLOAT), ("MinIterations", INT), ("MaxIterations", INT), ("TotalKmers", INT),
                                ("HaploDir", STRING)])
    return Struct("Info", [("Version", STRING), ("NumProc", INT), ("Profiling", BOOL), ("KmerSize", INT), ("SketchSize", INT),
                           ("WindowSize", INT), ("NumPart", INT), ("MaxK", INT), ("MaxSketchSpan", INT),
                           ("ContainmentThreshold", FLOAT), ("IndexDir", STRING), ("Store", Map("Store", UINT, graph)),
                           ("Sketch", align), ("Haplotype", haplo)])


def lshe_type():
    key = Struct("Key", [("GraphID", UINT), ("Node", UINT), ("OffSet", UINT), ("ContainedNodes", Map("map[uint64]float64", UINT, FLOAT)),
                         ("Ref", Slice("[]uint32", UINT)), ("RC", BOOL), ("Sketch", Slice("[]uint64", UINT)), ("Freq", FLOAT),
                         ("MergeSpan", UINT), ("WindowSize", UINT)])
    return Struct("ContainmentIndex", [("NumPart", INT), ("MaxK", INT), ("NumWindowKmers", INT), ("SketchSize", INT),
                                       ("WindowLookup", Map("map[string]lshe.Key", STRING, key))])


def index_to_go_values(index, max_sketch_span=30, as_written_by_index=None):
    """the values `groot index` would hold in memory for this flat index (field by field).  as_written_by_index = the index
    directory name: only the fields cmd/index.go:96-106 sets (what groot_index_save_gob writes); otherwise a few more
    fields are filled in to exercise the reader"""
    a, v = index.arrays, index.view
    store, lookup = {}, {}
    for g in range(v.n_graphs):
        n0, n1 = int(a["graph_node_off"][g]), int(a["graph_node_off"][g + 1])
        p0, p1 = int(a["graph_path_off"][g]), int(a["graph_path_off"][g + 1])
        nodes = []
        for n in range(n0, n1):
            seq = bytes(a["bases"][int(a["node_seq_off"][n]):int(a["node_seq_off"][n + 1])])
            e0, e1 = int(a["node_edge_off"][n]), int(a["node_edge_off"][n + 1])
            q0, q1 = int(a["node_np_off"][n]), int(a["node_np_off"][n + 1])
            pids = [int(x) for x in a["np_path"][q0:q1]]
            nodes.append({"SegmentID": int(a["node_seg_id"][n]), "SegmentLength": float(len(seq)), "Sequence": seq,
                          "OutEdges": [int(a["node_seg_id"][int(e)]) for e in a["edges"][e0:e1]], "PathIDs": pids,
                          "Position": {p: int(x) for p, x in zip(pids, a["np_pos"][q0:q1])}})
        store[g] = {"GrootVersion": None if as_written_by_index else "1.1.2", "GraphID": g, "SortedNodes": nodes,
                    "Paths": {p - p0: index.path_name(p).encode() for p in range(p0, p1)},
                    "Lengths": {p - p0: int(a["path_len"][p]) for p in range(p0, p1)},
                    "NodeLookup": {int(a["node_seg_id"][n]): n - n0 for n in range(n0, n1)},
                    "Masked": bool(a["graph_masked"][g])}
    s = v.sketch_size
    dup = {}
    for w in range(v.n_windows):
        g, node, off = int(a["win_graph"][w]), int(a["node_seg_id"][int(a["win_node"][w])]), int(a["win_offset"][w])
        base = "g%dn%do%d" % (g, node, off)
        i = dup.get(base, 0)
        dup[base] = i + 1
        c0, c1 = int(a["win_cn_off"][w]), int(a["win_cn_off"][w + 1])
        r0, r1 = int(a["win_ref_off"][w]), int(a["win_ref_off"][w + 1])
        lookup["%s-%d" % (base, i)] = {
            "GraphID": g, "Node": node, "OffSet": off,
            "ContainedNodes": {int(a["node_seg_id"][int(n)]): float(c) for n, c in zip(a["cn_node"][c0:c1], a["cn_count"][c0:c1])},
            "Ref": [int(x) for x in a["win_ref"][r0:r1]], "Sketch": [int(x) for x in a["win_sketch"][w * s:(w + 1) * s]],
            "MergeSpan": int(a["win_merge_span"][w]), "WindowSize": int(v.window_size)}
    info = {"Version": "1.1.2", "NumProc": 8, "KmerSize": int(v.kmer_size), "SketchSize": int(s), "WindowSize": int(v.window_size),
            "NumPart": int(v.num_part), "MaxK": int(v.max_k), "MaxSketchSpan": max_sketch_span, "ContainmentThreshold": 0.99,
            "IndexDir": "index-dir", "Store": store, "Sketch": {}, "Haplotype": {}}
    if as_written_by_index:
        info.update({"NumProc": 0, "ContainmentThreshold": 0.0, "IndexDir": as_written_by_index})
    ci = {"NumPart": int(v.num_part), "MaxK": int(v.max_k), "NumWindowKmers": int(v.num_window_kmers), "SketchSize": int(s),
          "WindowLookup": lookup}
    return info, ci


def write_index_dir(index, out_dir, shuffle_seed=1):
    import os
    info, ci = index_to_go_values(index)
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, "groot.gg"), "wb") as f:
        f.write(Encoder(shuffle_seed).encode(info_type(), info).out)
    with open(os.path.join(out_dir, "groot.lshe"), "wb") as f:
        f.write(Encoder(shuffle_seed + 1).encode(lshe_type(), ci).out)
