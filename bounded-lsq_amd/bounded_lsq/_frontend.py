"""`least_squares` front-end: argument validation, callback wrapping, dispatch.

Same call signature, result fields, messages and error contract as the
reference's ``bounded_lsq.least_squares`` (least_squares.py:120-383) for the
'trf' and 'dogbox' methods.  Method 'lm' (a MINPACK bridge in the reference,
least_squares.py:52-97) is outside the GPU path and is not provided.
"""
from warnings import warn

import numpy as np
from scipy.optimize._numdiff import approx_derivative

from ._drivers import trf, dogbox
from ._hostmath import in_bounds, prepare_bounds

EPS = np.finfo(float).eps

TERMINATION_MESSAGES = {
    0: "The maximum number of function evaluations is exceeded.",
    1: "`gtol` termination condition is satisfied.",
    2: "`ftol` termination condition is satisfied.",
    3: "`xtol` termination condition is satisfied.",
    4: "Both `ftol` and `xtol` termination conditions are satisfied.",
}


def _clamp_tolerances(ftol, xtol, gtol):
    """least_squares.py:15-27: warn and clamp tolerances below machine eps."""
    out = []
    for name, val in (("`ftol`", ftol), ("`xtol`", xtol), ("`gtol`", gtol)):
        if val < EPS:
            warn("{} is too low, setting to machine epsilon {}.".format(name, EPS))
            val = EPS
        out.append(val)
    return tuple(out)


def _checked_scaling(scaling, x0):
    """least_squares.py:100-117."""
    if isinstance(scaling, str) and scaling == 'jac':
        return scaling
    try:
        scaling = np.asarray(scaling, dtype=float)
    except ValueError:
        raise ValueError("`scaling` must be 'jac' or array-like with numbers.")
    if np.any(scaling <= 0):
        raise ValueError("`scaling` must contain only positive values.")
    if scaling.ndim == 0:
        scaling = np.resize(scaling, x0.shape)
    if scaling.shape != x0.shape:
        raise ValueError("Inconsistent shapes between `scaling` and `x0`.")
    return scaling


def least_squares(fun, x0, jac='2-point', bounds=(-np.inf, np.inf), method='trf',
                  ftol=EPS ** 0.5, xtol=EPS ** 0.5, gtol=EPS ** 0.5, max_nfev=None,
                  scaling=1.0, diff_step=None, args=(), kwargs={}, options={}):
    """Minimise ``sum(fun(x)**2)`` subject to ``lb <= x <= ub``.

    Parameters and the returned ``OptimizeResult`` fields (x, fun, jac,
    obj_value, optimality, active_mask, nfev, njev, status, message, success,
    x_covariance) are those of the reference (least_squares.py:120-305).  The
    per-iteration linear algebra runs on the GPU; ``options`` may carry
    ``ctx`` (a ``bounded_lsq._abi.Context``) to choose the device.
    """
    if method not in ['trf', 'dogbox', 'lm']:
        raise ValueError("`method` must be 'trf', 'dogbox' or 'lm'.")
    if method == 'lm':
        raise NotImplementedError(
            "method='lm' is a MINPACK bridge in the reference and is outside "
            "the MI355X step path; use 'trf' or 'dogbox'.")
    if len(bounds) != 2:
        raise ValueError("`bounds` must contain 2 elements.")
    x0 = np.atleast_1d(x0).astype(float)
    if x0.ndim > 1:
        raise ValueError("`x0` must have at most 1 dimension.")
    lb, ub = prepare_bounds(bounds, x0)
    if lb.shape != x0.shape or ub.shape != x0.shape:
        raise ValueError("Inconsistent shapes between bounds and `x0`.")
    if np.any(lb >= ub):
        raise ValueError("Each lower bound mush be strictly less than each "
                         "upper bound.")
    if not (isinstance(jac, str) and jac in ['2-point', '3-point']) and not callable(jac):
        raise ValueError("`jac` must be '2-point', '3-point' or callable.")
    scaling = _checked_scaling(scaling, x0)
    ftol, xtol, gtol = _clamp_tolerances(ftol, xtol, gtol)
    if not in_bounds(x0, lb, ub):
        raise ValueError("`x0` is infeasible.")

    def residuals(x):
        f = np.atleast_1d(fun(x, *args, **kwargs))
        if f.ndim > 1:
            raise RuntimeError("`fun` must return at most 1-d array_like.")
        return np.ascontiguousarray(f, dtype=float)

    if callable(jac):
        def jacobian(x, f):
            J = np.atleast_2d(jac(x, *args, **kwargs))
            if J.ndim > 2:
                raise RuntimeError("`jac` must return at most 2-d array_like.")
            return np.ascontiguousarray(J, dtype=float)
    else:
        def jacobian(x, f):
            J = approx_derivative(fun, x, rel_step=diff_step, method=jac, f0=f,
                                  bounds=bounds, args=args, kwargs=kwargs)
            J = np.atleast_2d(J)
            if J.ndim > 2:
                raise RuntimeError("`jac` must return at most 2-d array_like.")
            return np.ascontiguousarray(J, dtype=float)

    driver = trf if method == 'trf' else dogbox
    result = driver(residuals, jacobian, x0, lb, ub, ftol, xtol, gtol, max_nfev, scaling,
                    **options)
    result.message = TERMINATION_MESSAGES[result.status]
    result.success = result.status > 0
    return result
