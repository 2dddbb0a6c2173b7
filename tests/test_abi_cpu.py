"""CPU-side checks of the C-ABI boundary: the library loads, exports every
symbol include/blsq.h declares, and refuses to run without a GPU (no compute
calls here)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "blsq.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(blsq_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from bounded_lsq import _abi
    lib = _abi.load()
    names = _header_symbols()
    assert len(names) >= 30
    for name in names:
        assert hasattr(lib, name), "missing export: " + name
    # and the binding table covers exactly the header
    assert sorted(_abi.SIGNATURES) == names


def test_version_and_pure_helpers():
    from bounded_lsq import _abi
    lib = _abi.load()
    assert lib.blsq_version() >= 100
    assert lib.blsq_tsqr_tri_ld(128) == 144
    assert lib.blsq_tsqr_tri_ld(256) == 272


def test_no_gpu_means_loud_failure():
    from bounded_lsq import _abi
    lib = _abi.load()
    if lib.blsq_device_count() > 0:
        pytest.skip("a GPU is visible here")
    with pytest.raises(_abi.BlsqError):
        _abi.Context(0)
    import bounded_lsq
    with pytest.raises(_abi.BlsqError):
        bounded_lsq.TrfStepSolver(1, 8, 2)


def test_product_package_does_not_import_oracle():
    pkg = os.path.join(ROOT, "bounded-lsq_amd", "bounded_lsq")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            txt = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+[^\n]*oracle", txt, flags=re.M), fn
