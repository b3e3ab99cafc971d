"""The C-ABI libraries load and export every symbol include/*.h declares (no compute without a GPU)."""
import ctypes as C
import os
import re

import pytest

from conftest import REPO


def declared(header):
    txt = open(os.path.join(REPO, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(groot_[a-z0-9_]+)\s*\(", txt)))


def test_host_library_exports_every_declared_symbol():
    L = C.CDLL(os.path.join(REPO, "build", "libgroot_host.so"))
    names = declared("groot_host.h")
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), n


def test_hip_library_exports_every_declared_symbol(hip_lib):
    L = C.CDLL(hip_lib)
    names = declared("groot_hip.h")
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), n


def test_no_cpu_fallback(hip_lib, small_index):
    """without a HIP device open must fail loudly (GROOT_E_DEVICE), never fall back to the CPU"""
    from groot_amd import device
    from groot_amd.host import GrootError

    if device.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(GrootError) as e:
        device.Aligner(small_index, max_batch_reads=1024)
    assert e.value.code == -5


def test_product_does_not_touch_the_oracle():
    """only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use oracle/"""
    for root, _, files in os.walk(os.path.join(REPO, "groot_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")):
                assert "oracle" not in open(os.path.join(root, f), errors="replace").read().lower(), os.path.join(root, f)
    for f in os.listdir(os.path.join(REPO, "include")):
        assert "oracle" not in open(os.path.join(REPO, "include", f)).read().lower()
