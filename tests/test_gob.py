"""Reader for the reference's own index files (Go encoding/gob: <indexDir>/groot.gg + groot.lshe, cmd/index.go:130-131).

Go is not installed here, so no Go-written file exists to test against.  The decoder is pinned on the byte vectors
printed in the encoding/gob package documentation (typed in below), and the index mapping on round trips through the
independent test-side encoder of tests/gobenc.py, itself pinned byte-for-byte on the same documentation example."""
import os

import numpy as np
import pytest

import gobenc
from groot_amd import host

# "type Point struct { X, Y int }" then Point{22, 33}: the annotated example of the gob documentation
DOC_POINT_TYPE = bytes.fromhex("1f ff 81 03 01 01 05 50 6f 69 6e 74 01 ff 82 00 01 02 01 01 58 01 04 00 01 01 59 01 04 00 00 00".replace(" ", ""))
DOC_POINT_VALUE = bytes.fromhex("07 ff 82 01 2c 01 42 00".replace(" ", ""))


def test_decoder_on_documentation_example():
    assert host.gob_to_json(DOC_POINT_TYPE + DOC_POINT_VALUE) == [{"X": 22, "Y": 33}]
    # two values on one stream; the second omits the zero-valued X
    assert host.gob_to_json(DOC_POINT_TYPE + DOC_POINT_VALUE + bytes.fromhex("05ff82022c00")) == [{"X": 22, "Y": 33}, {"Y": 22}]


def test_scalar_encodings_of_the_documentation():
    # "7 is transmitted as 07, 256 as FE 01 00, -129 as FE 01 01, -7 as 0D; 17.0 is encoded in three bytes FE 31 40"
    assert gobenc.enc_uint(7) == b"\x07" and gobenc.enc_uint(256) == b"\xfe\x01\x00"
    assert gobenc.enc_int(-129) == b"\xfe\x01\x01" and gobenc.enc_int(-7) == b"\x0d"
    assert gobenc.enc_float(17.0) == b"\xfe\x31\x40"
    v = bytes.fromhex("09ff8201fe010101 0d00".replace(" ", ""))                # Point{-129, -7}
    assert host.gob_to_json(DOC_POINT_TYPE + v) == [{"X": -129, "Y": -7}]
    # same struct with float fields (type id 4 -> 08) and uint fields (type id 3 -> 06)
    as_float = DOC_POINT_TYPE.replace(b"\x01\x58\x01\x04", b"\x01\x58\x01\x08").replace(b"\x01\x59\x01\x04", b"\x01\x59\x01\x06")
    v = bytes.fromhex("0bff8201fe314001fe010000")                                # {X: 17.0, Y: 256}
    assert host.gob_to_json(as_float + v) == [{"X": 17.0, "Y": 256}]


def test_test_encoder_reproduces_documentation_bytes():
    point = gobenc.Struct("Point", [("X", gobenc.INT), ("Y", gobenc.INT)])
    out = gobenc.Encoder().encode(point, {"X": 22, "Y": 33}).out
    assert bytes(out) == DOC_POINT_TYPE + DOC_POINT_VALUE


def test_containers_roundtrip():
    inner = gobenc.Struct("Inner", [("A", gobenc.UINT), ("B", gobenc.Slice("[]float64", gobenc.FLOAT))])
    outer = gobenc.Struct("Outer", [("Name", gobenc.STRING), ("Raw", gobenc.BYTES), ("Flag", gobenc.BOOL),
                                    ("M", gobenc.Map("map[string]Inner", gobenc.STRING, inner)),
                                    ("P", gobenc.Map("map[int]int", gobenc.INT, gobenc.INT)),
                                    ("S", gobenc.Slice("[]Inner", inner)), ("Big", gobenc.UINT), ("Neg", gobenc.INT)])
    val = {"Name": "x\"y", "Raw": b"\x00\xffA", "Flag": True, "M": {"k1": {"A": 5, "B": [1.5, -2.0]}, "k2": {}},
           "P": {-3: 4, 1 << 40: -(1 << 50)}, "S": [{"A": 1}, {}, {"B": [0.0]}], "Big": (1 << 64) - 1, "Neg": -(1 << 63)}
    got = host.gob_to_json(gobenc.Encoder().encode(outer, val).out)
    assert got == [{"Name": "x\"y", "Raw": "\x00\xffA", "Flag": True, "M": [["k1", {"A": 5, "B": [1.5, -2.0]}], ["k2", {}]],
                    "P": [[-3, 4], [1 << 40, -(1 << 50)]], "S": [{"A": 1}, {}, {"B": [0.0]}], "Big": (1 << 64) - 1, "Neg": -(1 << 63)}]
    # a top-level non-struct value carries a zero delta in front
    assert host.gob_to_json(gobenc.Encoder().encode(gobenc.Slice("[]uint32", gobenc.UINT), [1, 2, 300]).out) == [[1, 2, 300]]


def _same_index(a, b):
    for n in ("kmer_size", "sketch_size", "window_size", "num_part", "max_k", "num_window_kmers", "path_words", "n_graphs",
              "n_nodes", "n_windows", "n_paths"):
        assert getattr(a.view, n) == getattr(b.view, n), n
    for k, arr in a.arrays.items():
        assert np.array_equal(arr, b.arrays[k]), k


@pytest.mark.parametrize("which", ["small", "testgfa"])
def test_reference_index_dir_loads_to_the_same_flat_index(which, small_index, testgfa_index, tmp_path):
    index = small_index if which == "small" else testgfa_index
    d = str(tmp_path / "idx")
    gobenc.write_index_dir(index, d, shuffle_seed=7)   # map entries in shuffled order, as Go writes them
    _same_index(index, host.Index.load_gob(d))


@pytest.mark.parametrize("which", ["small", "testgfa"])
def test_writer_matches_the_test_encoder_byte_for_byte(which, small_index, testgfa_index, tmp_path):
    """groot_index_save_gob (C++) and tests/gobenc.py (Python) were written independently from the format description: for
    the same values in the same map order they must emit the same bytes; and the reader takes the files back"""
    index = small_index if which == "small" else testgfa_index
    d = str(tmp_path / "out")
    index.save_gob(d)
    info, ci = gobenc.index_to_go_values(index, as_written_by_index=d)
    assert open(os.path.join(d, "groot.gg"), "rb").read() == bytes(gobenc.Encoder().encode(gobenc.info_type(), info).out)
    assert open(os.path.join(d, "groot.lshe"), "rb").read() == bytes(gobenc.Encoder().encode(gobenc.lshe_type(), ci).out)
    _same_index(index, host.Index.load_gob(d))


def test_gob_index_errors(testgfa_index, tmp_path):
    d = str(tmp_path / "idx")
    with pytest.raises(host.GrootError) as e:
        host.Index.load_gob(d)
    assert e.value.code == -2                                   # GROOT_E_IO
    gobenc.write_index_dir(testgfa_index, d)
    gg, db = os.path.join(d, "groot.gg"), os.path.join(d, "groot.lshe")
    good_gg, good_db = open(gg, "rb").read(), open(db, "rb").read()
    for path, good in ((gg, good_gg), (db, good_db)):
        for cut in (0, 10, len(good) // 2, len(good) - 1):       # empty (runtime.go:84-86) and truncated streams
            open(path, "wb").write(good[:cut])
            with pytest.raises(host.GrootError) as e:
                host.Index.load_gob(d)
            assert e.value.code == -3, cut                       # GROOT_E_FORMAT
        open(path, "wb").write(good)
    host.Index.load_gob(d)
    # a window that points at a segment its graph does not hold
    info, ci = gobenc.index_to_go_values(testgfa_index)
    next(iter(ci["WindowLookup"].values()))["Node"] = 10 ** 6
    open(db, "wb").write(gobenc.Encoder().encode(gobenc.lshe_type(), ci).out)
    with pytest.raises(host.GrootError) as e:
        host.Index.load_gob(d)
    assert e.value.code == -3
    # LSH parameters that disagree between the two files
    info, ci = gobenc.index_to_go_values(testgfa_index)
    ci["SketchSize"] += 1
    open(db, "wb").write(gobenc.Encoder().encode(gobenc.lshe_type(), ci).out)
    with pytest.raises(host.GrootError):
        host.Index.load_gob(d)
