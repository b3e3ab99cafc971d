"""The gzip reader of the FASTQ ingest (groot_amd/csrc/host/gz_inflate.hpp) against zlib: tools/gz_check.cpp inflates a file with both and
compares.  Streams of every block type and compression setting, several members, header fields, corrupt and truncated input; then the reader
itself on a gzip FASTQ (sketch.go:41-77 wraps named *.gz files in a gzip reader)."""
import gzip
import io
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("gz") / "gz_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(REPO, "tools", "gz_check.cpp"), "-lz"], check=True)
    return exe


@pytest.fixture(scope="module")
def checker_san(tmp_path_factory):
    """the same tool under AddressSanitizer + UBSan: the decoder works with slack (4-byte literal stores, 16-byte match copies, 8-byte loads past
    the input's end) -- an access outside its buffers that stays inside the heap would go unnoticed in the plain build"""
    exe = str(tmp_path_factory.mktemp("gz") / "gz_check_san")
    r = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-o", exe,
                        os.path.join(REPO, "tools", "gz_check.cpp"), "-lz"], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("no sanitizer runtime in this image: " + r.stderr[-200:])
    return exe


def gz(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, wbits=31, memlevel=8):
    c = zlib.compressobj(level, zlib.DEFLATED, wbits, memlevel, strategy)
    return c.compress(data) + c.flush()


def fastq_text(n, seed=5):
    rng = np.random.default_rng(seed)
    out = bytearray()
    for i in range(n):
        L = int(rng.integers(60, 151))
        out += b"@read%d/%d\n" % (i, L) + rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), size=L, p=[0.248, 0.248, 0.248, 0.248, 0.008]).tobytes()
        out += b"\n+\n" + (rng.integers(33, 74, L, dtype=np.uint8)).tobytes() + b"\n"
    return bytes(out)


def run(checker, path, chunk=1 << 20):
    return subprocess.run([checker, str(path), str(chunk)], capture_output=True, text=True, check=True).stdout.strip()


def test_streams_of_every_kind(checker, tmp_path):
    text = fastq_text(12000)
    rng = np.random.default_rng(3)
    rnd = rng.integers(0, 256, 700_000, dtype=np.uint8).tobytes()
    pat = bytes(rng.integers(65, 91, 40000, dtype=np.uint8))

    def bgzf_member(d):
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        c = co.compress(d) + co.flush()
        return b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(c) + 25) + c + struct.pack("<II", zlib.crc32(d), len(d))

    bio = io.BytesIO()
    with gzip.GzipFile(filename="some_name.fq", mode="wb", fileobj=bio, mtime=0) as f:
        f.write(text[:300000])
    h = bytearray(gz(text)[:10])
    h[3] |= 16 | 2
    hdr = bytes(h) + b"a comment\0"
    hdr += struct.pack("<H", zlib.crc32(hdr) & 0xFFFF)
    cases = {
        "l1": gz(text, 1), "l6": gz(text, 6), "l9": gz(text, 9), "fixed": gz(text, 6, zlib.Z_FIXED), "huffman_only": gz(text, 6, zlib.Z_HUFFMAN_ONLY),
        "rle": gz(text, 6, zlib.Z_RLE), "stored": gz(text, 0), "memlevel1": gz(text, 6, memlevel=1), "window512": gz(text, 6, wbits=16 + 9),
        "random": gz(rnd, 6), "random_stored": gz(rnd, 0), "zeros": gz(bytes(3_000_000), 9), "empty": gz(b""), "one_byte": gz(b"x"),
        "members": gz(text[:100000]) + gz(b"") + gz(text[100000:300000], 1) + gz(rnd[:70000], 0) + gz(text[300000:900000], 9),
        "bgzf": b"".join(bgzf_member(text[i:i + 60000]) for i in range(0, 900_000, 60000)) + bgzf_member(b""),
        "fname": bio.getvalue(), "fcomment_fhcrc": hdr + gz(text)[10:], "trailing_bytes": gz(text) + b"\0\0\0\0garbage",
        "far_matches": gz(pat + rnd[:1000] + pat + pat[:77] + rnd[:3] + pat[5:], 9),
        "short_distances": gz((b"ab" * 50000) + (b"abc" * 30000) + (b"abcdefg" * 20000) + b"z" * 100000, 6),
    }
    for name, data in cases.items():
        p = tmp_path / (name + ".gz")
        p.write_bytes(data)
        for chunk in ((1 << 20, 4099, 1) if len(data) < 400_000 else (1 << 20, 4099)):
            out = run(checker, p, chunk)
            assert out.startswith("same") and out.endswith("(zlib ok)"), (name, chunk, out)


def test_corrupt_and_truncated_streams_end_in_an_error(checker, tmp_path):
    good = gz(fastq_text(3000), 6)
    bad = {"half": good[:len(good) // 2], "no_trailer": good[:-3], "short_header": good[:5]}
    b = bytearray(good); b[len(b) // 3] ^= 0x55; bad["flipped_byte"] = bytes(b)
    b = bytearray(good); b[-6] ^= 1; bad["crc"] = bytes(b)
    b = bytearray(good); b[-1] ^= 1; bad["length"] = bytes(b)
    for name, data in bad.items():
        p = tmp_path / (name + ".gz")
        p.write_bytes(data)
        assert run(checker, p).startswith("error:"), name
    # a thousand damaged streams: an error or the right bytes, never anything else (and never a crash: check=True)
    rng = np.random.default_rng(17)
    for it in range(300):
        b = bytearray(good)
        if rng.random() < 0.3:
            b = b[: int(rng.integers(1, len(b) + 1))]
        for _ in range(int(rng.integers(1, 4))):
            b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))
        p = tmp_path / "damaged.gz"
        p.write_bytes(bytes(b))
        out = run(checker, p, [1 << 20, 4099, 13][it % 3])
        assert out.startswith("error:") or out.startswith("same"), (it, out)


def test_input_that_ends_inside_a_later_members_header_is_an_error(checker, tmp_path):
    """behind a member the input may end, or go on with bytes that are not a gzip header; once the 1f 8b of another member has matched, the file
    is truncated if it ends before that member's header does (a bgzip FASTQ cut inside a block header) -- gzread and Go's gzip.Reader (the
    reference's reader, sketch.go:175-238) report an error there, and so does this one"""
    text = fastq_text(2000)
    a, b = gz(text[:50000]), gz(text[50000:])
    fextra = b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0\x1b\0"
    fname = bytearray(b[:10]); fname[3] |= 8
    cases = {"magic_only": a + b[:2], "nine_bytes": a + b[:9], "inside_fextra": a + fextra[:14], "inside_fname": a + bytes(fname) + b"reads.fq"}
    for name, data in cases.items():
        p = tmp_path / (name + ".gz")
        p.write_bytes(data)
        out = run(checker, p)
        assert out.startswith("error:") and "header" in out, (name, out)
    # ... while stray bytes that are not a header are still ignored, and a lone 0x1f is such a byte
    for name, data in {"stray": a + b"\0\0garbage", "lone_1f": a + b"\x1f", "not_magic": a + b"\x1f\x8c\x08"}.items():
        p = tmp_path / (name + ".gz")
        p.write_bytes(data)
        assert run(checker, p).startswith("same 50000"), name


def test_damaged_streams_of_every_block_type_under_the_sanitizers(checker_san, tmp_path):
    """stored, fixed, dynamic and BGZF streams -- one of them with more than 2 MiB of compressed data, so that the decoder's refill of its 1 MiB
    input buffer and its end-of-input path run under mutation too -- damaged a few hundred times: an error or the right bytes, and no report
    from AddressSanitizer / UBSan (a report ends the tool with a non-zero status: check=True)"""
    text = fastq_text(3000, seed=21)
    rng = np.random.default_rng(23)

    def bgzf_member(d):
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        c = co.compress(d) + co.flush()
        return b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(c) + 25) + c + struct.pack("<II", zlib.crc32(d), len(d))

    big = rng.integers(0, 256, 2_600_000, dtype=np.uint8).tobytes()        # incompressible: > 2 MiB of deflate data
    streams = {"stored": gz(text[:200000], 0), "fixed": gz(text, 6, zlib.Z_FIXED), "dynamic": gz(text, 6),
               "bgzf": b"".join(bgzf_member(text[i:i + 30000]) for i in range(0, 300000, 30000)) + bgzf_member(b""),
               "big": gz(big[:1_300_000], 6) + gz(big[1_300_000:], 1)}
    for name, good in streams.items():
        p = tmp_path / (name + ".gz")
        p.write_bytes(good)
        assert run(checker_san, p, 4099 if len(good) < 400_000 else 1 << 20).startswith("same"), name
        n_it = 25 if name == "big" else 90
        for it in range(n_it):
            b = bytearray(good)
            kind = it % 3
            if kind == 0:                                  # cut somewhere (block headers and the refill boundary included)
                b = b[: int(rng.integers(1, len(b)))]
            elif kind == 1:                                # damage near the start of a block / member header
                at = int(rng.integers(0, min(len(b), 64)))
                b[at] ^= 1 << int(rng.integers(0, 8))
            for _ in range(int(rng.integers(0, 3))):
                b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))
            p.write_bytes(bytes(b))
            out = run(checker_san, p, [1 << 20, 4099, 13][it % 3] if len(b) < 400_000 else 1 << 20)
            assert out.startswith("error:") or out.startswith("same"), (name, it, out)


def test_reader_on_gzip_fastq(tmp_path):
    """the ingest itself: the batches of a gzip FASTQ (several members) equal those of the plain file"""
    from groot_amd import host

    text = fastq_text(20000, seed=9)
    plain, packed = tmp_path / "r.fq", tmp_path / "r.fq.gz"
    plain.write_bytes(text)
    packed.write_bytes(gz(text[:1_000_003], 6) + gz(text[1_000_003:], 1))

    def records(path):
        rd = host.ParallelReads([str(path)], threads=4, block_bytes=1 << 20, max_batch_reads=7000, max_batch_bases=1 << 24)
        names, seqs, quals, packed = [], [], [], []
        for b in rd.batches():
            names += b["names"]; seqs += b["seqs"]; quals += b["quals"]
            packed.append((b["n"], b["packed"].tobytes(), b["exc_pos"].tobytes()))
        rd.close()
        return names, seqs, quals, packed

    a, z = records(plain), records(packed)
    assert len(a[0]) == 20000
    assert a[0] == z[0] and a[1] == z[1] and a[2] == z[2]
    assert sum(x[0] for x in a[3]) == sum(x[0] for x in z[3]) == 20000
    # (batch boundaries follow the text blocks, which differ between the two files: the wire format is compared where they agree)
    if [x[0] for x in a[3]] == [x[0] for x in z[3]]:
        assert a[3] == z[3]
    # a damaged file is an error of the reader, with the decoder's reason
    bad = tmp_path / "bad.fq.gz"
    data = bytearray(packed.read_bytes())
    data[len(data) // 2] ^= 0x10
    bad.write_bytes(bytes(data))
    with pytest.raises(Exception) as ei:
        records(bad)
    assert "gzip" in str(ei.value) or "FASTQ" in str(ei.value) or "fastq" in str(ei.value)
