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


class _MockLib:
    """Counts the destroy calls of the C-ABI (no GPU needed)."""

    def __init__(self):
        self.calls = []

    def blsq_ctx_destroy(self, h):
        self.calls.append(("ctx", h))
        return 0 if h else -1

    def blsq_outer_destroy(self, h):
        self.calls.append(("outer", h))
        return 0

    def blsq_trf_plan_destroy(self, h):
        self.calls.append(("trf", h))
        return 0


def _mock_ctx_and_driver(own_ctx):
    import weakref
    from bounded_lsq import _abi, _outer, _hip_step
    lib = _MockLib()
    ctx = object.__new__(_abi.Context)
    ctx.lib, ctx.h, ctx.device_id, ctx._plans = lib, 11, 0, weakref.WeakSet()
    drv = object.__new__(_outer.OuterDriver)
    drv.ctx, drv._own_ctx, drv.h = ctx, own_ctx, 22
    ctx.adopt(drv)
    sol = object.__new__(_hip_step.TrfStepSolver)
    sol.ctx, sol.lib, sol.h = ctx, lib, 33
    ctx.adopt(sol)
    return lib, ctx, drv, sol


@pytest.mark.parametrize("first", ["driver", "ctx"])
def test_close_of_a_driver_that_owns_its_context_destroys_everything_once(first):
    """OuterDriver(ctx=None).close() closes its ctx, whose close() walks its plans, the driver
    included: each handle must be destroyed exactly once, with no recursion, whichever end starts."""
    lib, ctx, drv, sol = _mock_ctx_and_driver(own_ctx=True)
    (drv if first == "driver" else ctx).close()
    assert sorted(lib.calls) == [("ctx", 11), ("outer", 22), ("trf", 33)]
    assert lib.calls[-1] == ("ctx", 11)                      # plans go before their context
    drv.close(); ctx.close(); sol.close()                    # idempotent
    assert len(lib.calls) == 3
    assert ctx.h is None and drv.h is None and sol.h is None


def test_close_of_a_driver_on_a_borrowed_context_leaves_the_context_alone():
    lib, ctx, drv, sol = _mock_ctx_and_driver(own_ctx=False)
    drv.close()
    assert lib.calls == [("outer", 22)] and ctx.h == 11
    ctx.close()
    assert sorted(lib.calls) == [("ctx", 11), ("outer", 22), ("trf", 33)]


def test_unknown_negative_status_raises_value_error_not_key_error():
    import numpy as np
    from bounded_lsq._outer import raise_step_errors
    with pytest.raises(ValueError, match="problem 1"):
        raise_step_errors(np.array([0, -77, 0]))
    with pytest.raises(ValueError, match="`s` is zero"):
        raise_step_errors(np.array([-1]))
    raise_step_errors(np.array([0, 1, 3]))                   # termination codes: nothing raised


def test_option_table_is_documented():
    """Every switch of the library's option table (blsq_option_info needs no GPU) appears in INTEGRATION.md with its
    environment variable, and no getenv("BLSQ_...") is left in the sources outside the table and the RCCL loader."""
    import ctypes as C
    import glob
    import re
    from bounded_lsq import _abi
    lib = _abi.load()
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    n = lib.blsq_option_count()
    assert n >= 26
    for i in range(n):
        nm, ev, dc = C.c_char_p(), C.c_char_p(), C.c_char_p()
        df = C.c_double()
        assert lib.blsq_option_info(i, C.byref(nm), C.byref(ev), C.byref(df), C.byref(dc)) == 0
        assert "`%s`" % nm.value.decode() in doc and "`%s`" % ev.value.decode() in doc, nm.value
        assert ev.value.decode() == "BLSQ_" + nm.value.decode().upper()
    assert lib.blsq_option_info(n, None, None, None, None) != 0
    left = []
    for path in glob.glob(os.path.join(ROOT, "bounded-lsq_amd", "csrc", "*")):
        if not path.endswith((".hip", ".h", ".cpp")) or path.endswith("blsq_options.cpp"):
            continue
        for m in re.finditer(r'getenv\("(BLSQ_[A-Z0-9_]+)"\)', open(path).read()):
            if m.group(1) != "BLSQ_RCCL_PATH":
                left.append((os.path.basename(path), m.group(1)))
    assert not left, left
