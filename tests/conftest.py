"""Shared fixtures.  `-m "not gpu"` runs here on the CPU (oracle, host logic, ABI); `-m gpu` tests are
the parity tests proper and call the HIP path through the C ABI."""
import gzip
import os
import sys
import tarfile

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
DATA = os.path.join(REPO, "tests", "golden", "data")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (HIP path through libgroot_hip.so)")


@pytest.fixture(scope="session", autouse=True)
def native_libs():
    """host library + oracle are needed by every test; both build in seconds"""
    import __graft_entry__ as g

    g.build_host()
    g.build_oracle()


@pytest.fixture(scope="session")
def hip_lib(native_libs):
    import __graft_entry__ as g

    # torch bundles its own HIP runtime: when a test uses torch tensors next to libgroot_hip.so, torch has to bring its copy
    # up first (the other order leaves torch with "No HIP GPUs are available"), whatever order the tests run in
    try:
        import torch

        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass
    return g.build_hip()


def read_fastq(path):
    op = gzip.open if path.endswith(".gz") else open
    with op(path, "rb") as f:
        lines = f.read().split(b"\n")
    return [(lines[i][1:], lines[i + 1], lines[i + 3]) for i in range(0, len(lines) - 3, 4)]


@pytest.fixture(scope="session")
def msa_dir(tmp_path_factory):
    d = tmp_path_factory.mktemp("argannot")
    with tarfile.open(os.path.join(DATA, "arg-annot.90.tar.gz")) as tf:
        members = [m for m in tf.getmembers() if os.path.basename(m.name).startswith("cluster") and m.name.endswith(".msa")]
        tf.extractall(d, members=members)
    return os.path.join(d, "arg-annot.90")


@pytest.fixture(scope="session")
def argannot_index(msa_dir):
    """arg-annot.90 with the `groot index` defaults k=31 s=21 w=100 x=8 y=4 (cmd/index.go:45-49)"""
    from groot_amd import host

    cache = host.index_cache_path("arg-annot.90.k31.s21.w100")
    if os.path.exists(cache):
        try:
            return host.Index.load(cache)
        except Exception:
            pass
    idx = host.Index.from_msa_dir(msa_dir)
    try:
        idx.save(cache)
    except Exception:
        pass
    return idx


@pytest.fixture(scope="session")
def small_index(msa_dir):
    """first 24 clusters (lexical order) of arg-annot.90"""
    from groot_amd import host

    return host.Index.from_msa_files(host.msa_files(msa_dir)[:24])


@pytest.fixture(scope="session")
def testgfa_index():
    """the reference's src/graph/test.gfa (6 Bla-B alleles) windowed with small parameters"""
    from groot_amd import host

    return host.Index.from_gfa_files([os.path.join(DATA, "test.gfa")], host.index_params(k=7, s=10, w=30))


@pytest.fixture(scope="session")
def resfinder_index(tmp_path_factory):
    """resfinder.90 (669 MSAs) with the `groot index` defaults: the database of BASELINE.json's mixed-length configuration
    (card.90 itself is not in the reference tree)"""
    from groot_amd import host

    d = tmp_path_factory.mktemp("resfinder")
    with tarfile.open(os.path.join(DATA, "resfinder.90.tar.gz")) as tf:
        members = [m for m in tf.getmembers() if os.path.basename(m.name).startswith("cluster") and m.name.endswith(".msa")]
        tf.extractall(d, members=members)
    return host.Index.from_msa_dir(os.path.join(d, "resfinder.90"))


@pytest.fixture(scope="session")
def accuracy_index(msa_dir):
    """arg-annot.90 with the parameters of testing/run_accuracy_tests.sh:13-20 (k=41 s=21 w=150 x=8 y=4)"""
    from groot_amd import host

    return host.Index.from_msa_dir(msa_dir, host.index_params(k=41, s=21, w=150))


@pytest.fixture(scope="session")
def accuracy_reads():
    return read_fastq(os.path.join(DATA, "argannot-150bp-10000-reads.fq.gz"))


def accuracy_stats(index, reads, alns):
    """testing/groot-accuracy.go:56-140: aligned / multi-aligned reads, reads without a record on the reference named in the
    read header (field 9 of the randomreads.sh name, up to '$'; randomreads writes '{' for '_'), records with the right start"""
    names = [index.path_name(p).lstrip("*") for p in range(index.view.n_paths)]
    hits = {}
    for a in alns:
        hits.setdefault(int(a["read_id"]), []).append(a)
    multi = sum(1 for h in hits.values() if len(h) > 1)
    wrong_as_the_tool_counts, wrong, right_start = 0, 0, 0
    for rid, hs in hits.items():
        parts = reads[rid][0].decode().split("_")
        ref, pos = parts[9].split("$")[0].split(" ")[0], int(parts[2])
        refs = [names[int(a["ref_id"])] for a in hs]
        wrong_as_the_tool_counts += ref not in refs
        wrong += ref.replace("{", "_") not in refs
        right_start += sum(1 for a, n in zip(hs, refs) if n == ref.replace("{", "_") and int(a["pos"]) == pos)
    return {"aligned": len(hits), "multi": multi, "misaligned_tool": wrong_as_the_tool_counts, "misaligned": wrong, "right_start": right_start}


@pytest.fixture(scope="session")
def genes_index():
    """src/pipeline/test-data/test-genes.msa with the parameters of 1_pipeline_test.go:32-40"""
    from groot_amd import host

    return host.Index.from_msa_files([os.path.join(DATA, "test-genes.msa")], host.index_params(k=51, s=30, w=100))


@pytest.fixture(scope="session")
def perfect_reads():
    return read_fastq(os.path.join(DATA, "full-argannot-perfect-reads-small.fq.gz"))


@pytest.fixture(scope="session")
def variable_reads():
    return read_fastq(os.path.join(DATA, "full-argannot-perfect-reads-small-variable-rl.fq.gz"))


@pytest.fixture(scope="session")
def oxa_reads():
    return read_fastq(os.path.join(DATA, "test-reads-OXA90-OXA106-100bp-with-errors.fastq.gz"))


def digest(arr):
    import hashlib

    return hashlib.sha256(np.ascontiguousarray(arr).tobytes()).hexdigest()
