"""Error / warning contract of the least_squares front-end (no GPU needed:
every check fires before the first solver is created).  Mirrors the
behaviours the reference tests in bounded_lsq/tests/test_least_squares.py
(:65-88,103-110,171-199,203-220)."""
import warnings

import numpy as np
import pytest

from bounded_lsq import least_squares
import bounded_lsq


def fun_trivial(x, a=0):
    return (x - a) ** 2 + 5.0


def test_bad_method_and_jac():
    with pytest.raises(ValueError):
        least_squares(fun_trivial, 2.0, method='abc')
    with pytest.raises(ValueError):
        least_squares(fun_trivial, 2.0, jac='oops')
    with pytest.raises(ValueError):
        least_squares(fun_trivial, 2.0, jac=3)


def test_lm_is_outside_the_path():
    with pytest.raises(NotImplementedError):
        least_squares(fun_trivial, 2.0, method='lm')


@pytest.mark.parametrize("method", ["trf", "dogbox"])
def test_bounds_validation(method):
    with pytest.raises(ValueError):          # wrong number of bounds
        least_squares(fun_trivial, 2.0, bounds=(1.0, 2.0, 3.0), method=method)
    with pytest.raises(ValueError):          # lb >= ub
        least_squares(fun_trivial, 2.0, bounds=(3.0, 2.0), method=method)
    with pytest.raises(ValueError):          # lb == ub
        least_squares(fun_trivial, 2.0, bounds=(2.0, 2.0), method=method)
    with pytest.raises(ValueError):          # shape mismatch
        least_squares(fun_trivial, [2.0, 1.0], bounds=([1.0, 2.0, 3.0], [4.0, 5.0, 6.0]),
                      method=method)
    with pytest.raises(ValueError):          # infeasible x0
        least_squares(fun_trivial, 2.0, bounds=(3.0, 4.0), method=method)
    with pytest.raises(ValueError):          # x0 with 2 dims
        least_squares(fun_trivial, [[1.0, 2.0]], method=method)


@pytest.mark.parametrize("method", ["trf", "dogbox"])
def test_scaling_validation(method):
    for bad in ('auto', -1.0, [1.0, -2.0], [1.0, 2.0, 3.0]):
        with pytest.raises(ValueError):
            least_squares(lambda x: x, [2.0, 1.0], scaling=bad, method=method)


def test_low_tolerances_warn_before_any_gpu_work():
    """least_squares.py:15-27: each tolerance below eps warns (then the run
    proceeds; without a GPU it stops with the loud BlsqError, not silently)."""
    from bounded_lsq import _abi
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        try:
            least_squares(fun_trivial, 2.0, ftol=1e-20, xtol=1e-20, gtol=1e-20, max_nfev=2)
        except _abi.BlsqError:
            pass
    assert sum("too low" in str(w.message) for w in rec) == 3


def test_exported_helpers_known_answers():
    """Known answers of the helper functions."""
    a = bounded_lsq.find_active_constraints(np.array([1e-13, .5, 1 - 1e-13]), np.zeros(3),
                                            np.ones(3))
    assert list(a) == [-1, 0, 1]
    r = bounded_lsq.make_strictly_feasible(np.array([0., 1.]), np.zeros(2), np.ones(2))
    assert r[0] == 5e-324 and r[1] == 0.9999999999999999
    r = bounded_lsq.make_strictly_feasible(np.array([0., 1.]), np.zeros(2), np.ones(2),
                                           rstep=1e-10)
    np.testing.assert_array_equal(r, [1e-10, 1 - 2e-10])
    lb, ub = bounded_lsq.prepare_bounds((0.0, [1.0, 2.0]), np.zeros(2))
    np.testing.assert_array_equal(lb, [0, 0])
    np.testing.assert_array_equal(ub, [1, 2])
    assert bounded_lsq.CL_optimality(np.array([.2, .2, .2]), np.array([-1., 1., 0.]), 0.0,
                                     np.array([1., np.inf, 1.])) == pytest.approx(0.8)


def test_step_status_maps_to_the_references_exceptions():
    """BLSQ_STATUS_* -> the ValueError the reference raises (trust_region.py:28-29, 34-35); in a
    batch the first offending ACTIVE problem is named."""
    import numpy as np
    import pytest
    from bounded_lsq import _hip_step
    _hip_step._raise_status(0)
    with pytest.raises(ValueError, match="`s` is zero"):
        _hip_step._raise_status(1)
    with pytest.raises(ValueError, match="`x` is not within the trust region"):
        _hip_step._raise_status(2)
    _hip_step.raise_batch_status(np.array([0, 0, 0]))
    _hip_step.raise_batch_status(np.array([0, 2, 0]), active=np.array([True, False, True]))
    with pytest.raises(ValueError, match="problem 2: `x` is not within"):
        _hip_step.raise_batch_status(np.array([0, 1, 2]), active=np.array([True, False, True]))
    from bounded_lsq._outer import raise_step_errors
    raise_step_errors(np.array([1, 3, 0, 4]))
    with pytest.raises(ValueError, match="problem 1: `s` is zero"):
        raise_step_errors(np.array([2, -1, -2]))
