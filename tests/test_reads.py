"""Parallel FASTQ ingest (groot_reads_*, the CLI's input side) against the serial line-scanner restatement
(groot_fastq_*): same reads, names, qualities in the same order for any block size / thread count, and the packed wire
format decodes back to the sequences.  Reference semantics: src/pipeline/sketch.go:41-77,213-236, seqio.go:173-188."""
import gzip
import os

import numpy as np
import pytest

from conftest import DATA
from groot_amd import host


def serial(files):
    fq = host.FastqReader(files)
    names, seqs, quals = [], [], []
    for b in fq.batches(max_reads=4096):
        for i in range(b["n"]):
            s0, s1 = int(b["seq_off"][i]), int(b["seq_off"][i + 1])
            n0, n1 = int(b["name_off"][i]), int(b["name_off"][i + 1])
            names.append(bytes(b["names"][n0:n1])); seqs.append(bytes(b["seq"][s0:s1])); quals.append(bytes(b["qual"][s0:s1]))
    fq.close()
    return names, seqs, quals


def parallel(files, **kw):
    rd = host.ParallelReads(files, **kw)
    names, seqs, quals, batches = [], [], [], []
    for b in rd.batches():
        names += b["names"]; seqs += b["seqs"]; quals += b["quals"]
        batches.append(b)
    rd.close()
    return names, seqs, quals, batches


def unpack(b):
    """decode the wire format back to ASCII"""
    codes = np.frombuffer(b"ACTG", dtype=np.uint8)
    n = b["n_bases"]
    pk = b["packed"]
    idx = np.arange(n)
    out = codes[(pk[idx // 4] >> (2 * (idx % 4))) & 3].copy()
    out[b["exc_pos"].astype(np.int64)] = b["exc_byte"]
    return out


@pytest.mark.parametrize("block,threads", [(0, 0), (1 << 16, 3), (4099, 2), (700, 1)])
def test_same_reads_as_the_line_scanner(native_libs, block, threads):
    files = [os.path.join(DATA, "full-argannot-perfect-reads-small.fq.gz"), os.path.join(DATA, "bla-b7-150bp-5x.fq"),
             os.path.join(DATA, "full-argannot-perfect-reads-small-variable-rl.fq.gz")]
    en, es, eq = serial(files)
    gn, gs, gq, batches = parallel(files, threads=threads, block_bytes=block, max_batch_reads=777)
    assert gn == en and gs == es
    assert [q[: len(s)] for q, s in zip(gq, gs)] == [q[: len(s)] for q, s in zip(eq, es)]
    assert all(b["n"] <= 777 for b in batches)
    # the wire format decodes back to the sequences, read by read
    k = 0
    for b in batches:
        asc = unpack(b)
        off = np.concatenate([[0], np.cumsum(b["seq_len"].astype(np.int64))])
        assert off[-1] == b["n_bases"] and b["max_len"] == b["seq_len"].max()
        for i in range(b["n"]):
            assert bytes(asc[off[i]:off[i + 1]]) == gs[k + i]
        k += b["n"]


def test_edge_cases(native_libs, tmp_path):
    # CRLF line ends, lower case + N (exceptions), a last line without '\n', a partial trailing record, records split across files
    a = tmp_path / "a.fq"
    b = tmp_path / "b.fastq.gz"
    c = tmp_path / "c.fq"
    a.write_bytes(b"@r1 desc\r\nACGTNNacgt\r\n+\r\nIIIIIIIIII\r\n@r2\nAC\n")           # r2 continues in the next file
    with gzip.open(b, "wb") as f:
        f.write(b"+\nII\n@r3\nTTTTGGGGCCCCAAAA\n+r3\nIIIIIIIIIIIIIIII")                  # no trailing newline
    c.write_bytes(b"@r4\nGATTACA\n+\nIIIIIII\n@r5\nACGT\n")                             # r5 is partial: dropped
    files = [str(a), str(b), str(c)]
    en, es, eq = serial(files)
    assert en == [b"r1 desc", b"r2", b"r3", b"r4"] and es == [b"ACGTNNacgt", b"AC", b"TTTTGGGGCCCCAAAA", b"GATTACA"]
    for block in (0, 16, 37):
        gn, gs, gq, batches = parallel(files, threads=2, block_bytes=block)
        assert (gn, gs) == (en, es), block
        assert [q[: len(s)] for q, s in zip(gq, gs)] == eq
        asc = np.concatenate([unpack(x) for x in batches])
        assert bytes(asc) == b"".join(es)
        exc = sum(len(x["exc_pos"]) for x in batches)
        assert exc == 6                                                                  # N N a c g t
    # blank lines: FastqHandler.Run skips them before an ID, a sequence or a '+' line (an empty line is a nil slice there,
    # sketch.go:217-222) and takes the fourth line as it comes -- a file ending in an extra newline followed by a second file,
    # blank lines between records, an empty quality line
    d = tmp_path / "d.fq"
    e2 = tmp_path / "e.fq"
    d.write_bytes(b"@b1\nACGTACGT\n+\nIIIIIIII\n\n\n@b2\n\nGGGG\n\r\n+\nIIII\n\n")
    e2.write_bytes(b"@b3\nTTTTT\n+\n\n@b4\nCC\n+\nII\n")
    en, es, eq = serial([str(d), str(e2)])
    assert en == [b"b1", b"b2", b"b3", b"b4"] and es == [b"ACGTACGT", b"GGGG", b"TTTTT", b"CC"]
    for block in (0, 9, 16, 37):
        gn, gs, gq, _ = parallel([str(d), str(e2)], threads=2, block_bytes=block)
        assert (gn, gs) == (en, es), block
        # (a quality line shorter than its sequence: the line scanner pads with '!', the parallel reader reports the short line)
        assert [q[: len(s)].ljust(len(s), b"!") for q, s in zip(gq, gs)] == [q[: len(s)].ljust(len(s), b"!") for q, s in zip(eq, es)]
    # an ID line that does not start with '@' is an error (seqio.go:179-181)
    bad = tmp_path / "bad.fq"
    bad.write_bytes(b"@ok\nACGT\n+\nIIII\nr2\nACGT\n+\nIIII\n")
    with pytest.raises(host.GrootError) as e:
        parallel([str(bad)])
    assert e.value.code == -3
    with pytest.raises(host.GrootError):
        host.ParallelReads([str(tmp_path / "missing.fq")])
    # empty input
    empty = tmp_path / "empty.fq"
    empty.write_bytes(b"")
    assert parallel([str(empty)])[0] == []
