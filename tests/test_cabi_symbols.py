"""CPU-only: the C-ABI library builds, loads and exports every symbol that
include/driftmi.h declares (no compute calls without a GPU), and the product
refuses to run without a GPU instead of falling back."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "driftmi.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(dm_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_exported():
    from driftscan_amd import _lib

    lib = _lib.load()
    names = _declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), "libdriftmi.so does not export %s" % n
    # and the ctypes table covers the header one to one
    assert sorted(_lib.SIGNATURES) == names
    assert lib.dm_version() >= 100


def test_driftcomm_header_symbols_exported():
    """include/driftcomm.h <-> libdriftcomm.so <-> driftscan_amd/comm.py (no collective is called without a GPU)."""
    from driftscan_amd import comm

    txt = open(os.path.join(ROOT, "include", "driftcomm.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = sorted(set(re.findall(r"\b(dm_[a-z0-9_]+)\s*\(", txt)))
    if not os.path.exists(comm.LIBPATH):
        pytest.skip("libdriftcomm.so not built (no RCCL headers)")
    lib = comm.load()
    assert len(names) >= 8 and sorted(comm.SIGNATURES) == names
    for n in names:
        assert hasattr(lib, n)
    assert lib.dm_comm_size(None) == -1 and lib.dm_comm_rank(None) == -1


def test_no_cpu_fallback():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from driftscan_amd import _lib

    with pytest.raises(_lib.DriftMIError):
        _lib.Context(0)


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under driftscan_amd/ may import it."""
    pkg = os.path.join(ROOT, "driftscan_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
