"""`groot report` restatement (src/reporting/reporting.go) and the reference's end-to-end assertion built on it:
testing/run_travis_tests.sh:12-56 -- index -w 150 -k 31 -s 20, align -t 0.99 on bla-b7-150bp-5x.fq, report -c 0.97 must
list exactly one gene, argannot~~~(Bla)B-7~~~AF189304:1-747.  CPU side: the alignments come from the oracle; the GPU run
of the same flow is tests/test_cli.py::test_travis_e2e_flow."""
import os
import subprocess

import numpy as np
import pytest

from conftest import DATA, REPO
from groot_amd import device, host
from oracle import oracle_py as O

B7 = "argannot~~~(Bla)B-7~~~AF189304:1-747"


@pytest.fixture(scope="module")
def travis_index(msa_dir):
    return host.Index.from_msa_dir(msa_dir, host.index_params(k=31, s=20, w=150))


def _oracle_bam(index, fq, out, threshold=0.99):
    run = O.Run(index, threshold)
    w = host.BamWriter(out, index, date="2020-01-01T00:00:00Z")
    first = 0
    for batch in host.FastqReader([fq]).batches():
        run.batch(batch["seq"], batch["seq_off"])
    al = run.alns().astype(device.ALN_DTYPE)
    # one batch holds the whole fixture
    batch = next(host.FastqReader([fq]).batches())
    w.write(al, batch, first)
    w.close()
    return al


def test_travis_assertion_on_the_oracle_alignments(travis_index, tmp_path):
    bam = str(tmp_path / "groot.bam")
    al = _oracle_bam(travis_index, os.path.join(DATA, "bla-b7-150bp-5x.fq"), bam)
    rows = host.report(bam, 0.97)
    assert [r[0] for r in rows] == [B7]                       # "bla-b7 is the only arg reported"
    name, count, length, cigar = rows[0]
    b7 = [p for p in range(travis_index.view.n_paths) if travis_index.path_name(p) == B7][0]
    assert count == int((al["ref_id"] == b7).sum()) and length == 747
    import re

    runs = [(int(n), c) for n, c in re.findall(r"(\d+)([MD])", cigar)]
    assert sum(n for n, _ in runs) == 747 and sum(n for n, c in runs if c == "M") / 747 >= 0.97
    # every alignment is on the Bla-B graph; with no cutoff each allele that received a record is listed
    lax = host.report(bam, 0.0)
    assert B7 in [r[0] for r in lax] and all("(Bla)B-" in r[0] for r in lax)
    assert len(lax) == len(set(int(x) for x in al["ref_id"]))
    assert host.report(bam, 1.0) == []                         # the 3' end of B-7 is never seeded (last window run dropped)
    with pytest.raises(host.GrootError):
        host.report(bam, 1.5)                                  # cmd/report.go:95-97
    with pytest.raises(host.GrootError):
        host.report(str(tmp_path / "missing.bam"))


def _write_bam(index, path, records):
    """records: (ref path id, pos, length) -> one forward primary record each"""
    import ctypes as C

    from groot_amd import _ffi

    w = host.BamWriter(path, index, date="2020-01-01T00:00:00Z")
    recs = (host.AlnRecord * len(records))()
    keep = []
    for i, (ref, pos, n) in enumerate(records):
        seq = np.frombuffer(b"A" * n, dtype=np.uint8).copy()
        qual = np.full(n, 40, dtype=np.uint8)
        name = b"r%d" % i
        keep.append((seq, qual, name))
        recs[i] = host.AlnRecord(name, len(name), _ffi.as_ptr(seq, C.c_uint8), _ffi.as_ptr(qual, C.c_uint8), n, ref, pos, 0, 0, 0, 0)
    host._check(host.lib().groot_bam_write(w._h, recs, C.c_uint64(len(records))))
    w.close()


def test_pileup_and_cigar_quirks(testgfa_index, tmp_path):
    """reporting.go:104-127 covers [Start, Start+Len] inclusive (one base past the read), clipped to the last base;
    cigarClean (:178-213) run-length encodes the pileup with its own end-of-string rule"""
    idx = testgfa_index
    L = int(idx.arrays["path_len"][0])
    name = idx.path_name(0).lstrip("*")
    bam = str(tmp_path / "a.bam")
    _write_bam(idx, bam, [(0, 0, 100), (0, 300, 50)])
    rows = host.report(bam, 0.0)
    assert rows == [(name, 2, L, f"101M199D51M{L - 351}D")]
    assert host.report(bam, 0.5) == []
    assert host.report(bam, 0.0, low_cov=True) == []           # cutoff forced to 0.97 (cmd/report.go:119-122)
    # covered to the end except the last base: "…M1D" form of the end rule
    _write_bam(idx, bam, [(0, 0, L - 2)])
    assert host.report(bam, 0.9) == [(name, 1, L, f"{L - 1}M1D")]
    # the whole gene: runs of one symbol end with the count including the last element
    _write_bam(idx, bam, [(0, 0, L), (1, 5, 10)])
    rows = host.report(bam, 0.97)
    assert rows == [(name, 1, L, f"{L}M")]
    # internal gap + --lowCov drops the gene even above the cutoff
    _write_bam(idx, bam, [(0, 0, L // 2), (0, L // 2 + 4, L)])
    assert len(host.report(bam, 0.97)) == 1 and host.report(bam, 0.97, low_cov=True) == []


def test_report_subcommand(testgfa_index, tmp_path):
    import __graft_entry__ as g

    cli = g.build_cli()
    L = int(testgfa_index.arrays["path_len"][0])
    bam = str(tmp_path / "x.bam")
    _write_bam(testgfa_index, bam, [(0, 0, L)])
    log = str(tmp_path / "r.log")
    r = subprocess.run([cli, "report", "--bamFile", bam, "-c", "0.9", "--log", log], cwd=REPO, capture_output=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert r.stdout.decode() == f"{testgfa_index.path_name(0).lstrip('*')}\t1\t{L}\t{L}M\n"
    text = open(log).read()
    for line in ("starting the report subcommand", "\tcoverage cutoff: 0.90", "finished"):
        assert line in text
    with open(bam, "rb") as f:                                   # BAM on stdin, as `groot align | groot report`
        r = subprocess.run([cli, "report", "-c", "0.9", "--log", log], cwd=REPO, stdin=f, capture_output=True, timeout=120)
    assert r.returncode == 0 and r.stdout.decode().count("\n") == 1
    assert subprocess.run([cli, "report", "--bamFile", str(tmp_path / "x.sam"), "--log", log], cwd=REPO, capture_output=True).returncode != 0
    assert subprocess.run([cli, "report", "--bamFile", bam, "-c", "1.2", "--log", log], cwd=REPO, capture_output=True).returncode != 0
