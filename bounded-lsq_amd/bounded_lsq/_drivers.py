"""Outer trust-region drivers on top of the GPU step path.

The reference keeps the accept/reject logic, the Delta/alpha updates, the
termination tests and the user callbacks in Python (trf.py:173-237,309-358;
dogbox.py:100-163,221-272); so does this module.  Everything between two
callbacks — the factorisation of J and the step computation — is two C-ABI
calls into libblsq_hip.so (`factor` once per new Jacobian, `step` once per
trial radius, without refactorising).
"""
import numpy as np
from numpy.linalg import norm
from scipy.optimize import OptimizeResult

from ._hip_step import (TrfStepSolver, DogboxStepSolver, SCALE_GIVEN, SCALE_JAC_INIT,
                        SCALE_JAC_UPDATE, lease_solver, return_solver)
from ._hostmath import shift_into_interior, active_mask, cl_vector

EPS = np.finfo(float).eps


def _is_jac(scaling):
    return isinstance(scaling, str) and scaling == 'jac'


def _termination(ftol_ok, xtol_ok):
    if ftol_ok and xtol_ok:
        return 4
    if ftol_ok:
        return 2
    if xtol_ok:
        return 3
    return None


def trf(fun, jac, x0, lb, ub, ftol, xtol, gtol, max_nfev, scaling, ctx=None):
    """Trust Region Reflective driver (same signature / result fields as the
    reference's ``trf``, trf.py:173)."""
    x = shift_into_interior(x0, lb, ub, rstep=1e-10)          # trf.py:201
    f = fun(x)
    nfev = 1
    J = jac(x, f)
    njev = 1
    if f.shape[0] != J.shape[0]:
        raise RuntimeError("Inconsistent dimensions between the returns of "
                           "`fun` and `jac` on the first iteration.")
    m, n = J.shape
    solver = lease_solver(TrfStepSolver, 1, m, n, ctx=ctx)
    try:
        use_jac = _is_jac(scaling)
        scale = np.ones(n) if use_jac else 1 / np.asarray(scaling, dtype=float)
        F = solver.factor(J[None], f[None], x[None], lb[None], ub[None], scale[None],
                          SCALE_JAC_INIT if use_jac else SCALE_GIVEN)
        scale = F.scale[0]
        v = cl_vector(x, F.g[0], lb, ub)
        Delta = norm(x0 / (scale * v ** 0.5))                 # trf.py:223-226 (x0, not x)
        if Delta == 0:
            Delta = 1.0
        obj_value = np.dot(f, f)
        alpha = 0.0
        if max_nfev is None:
            max_nfev = x0.size * 100
        status = None
        g_norm = float(F.g_norm[0])
        have_factor = True
        while nfev < max_nfev:
            if not have_factor:
                F = solver.factor(J[None], f[None], x[None], lb[None], ub[None], scale[None],
                                  SCALE_JAC_UPDATE if use_jac else SCALE_GIVEN)
                scale = F.scale[0]
                have_factor = True
            g_norm = float(F.g_norm[0])
            if g_norm < gtol:
                status = 1
            if status is not None:
                return OptimizeResult(
                    x=x, fun=f, jac=J, obj_value=obj_value, optimality=g_norm,
                    active_mask=active_mask(x, lb, ub, rtol=xtol), nfev=nfev, njev=njev,
                    status=status, x_covariance=None)
            actual_reduction = -1
            while actual_reduction <= 0 and nfev < max_nfev:
                S = solver.step(np.array([Delta]), np.array([alpha]), active_rtol=xtol)
                alpha = float(S.alpha[0])
                step_h_norm = float(S.step_h_norm[0])
                x_new = S.x_new[0]
                f_new = fun(x_new)
                nfev += 1
                obj_value_new = np.dot(f_new, f_new)
                actual_reduction = obj_value - obj_value_new
                predicted = float(S.predicted_reduction[0])
                if predicted > 0:
                    ratio = (actual_reduction - float(S.correction[0])) / predicted
                else:
                    ratio = 0
                if ratio < 0.25:                               # trf.py:325-331
                    Delta_new = 0.25 * step_h_norm
                    alpha *= Delta / Delta_new
                    Delta = Delta_new
                elif ratio > 0.75 and step_h_norm > 0.95 * Delta:
                    Delta *= 2.0
                    alpha *= 0.5
                ftol_ok = abs(actual_reduction) < ftol * obj_value and ratio > 0.25
                xtol_ok = norm(S.step[0]) < xtol * max(EPS ** 0.5, norm(x))
                status = _termination(ftol_ok, xtol_ok)
                if status is not None:
                    break
            if actual_reduction > 0:
                x = x_new
                f = f_new
                obj_value = obj_value_new
                J = jac(x, f)
                njev += 1
                have_factor = False
        return OptimizeResult(
            x=x, fun=f, jac=J, obj_value=obj_value, optimality=g_norm,
            active_mask=active_mask(x, lb, ub, rtol=xtol), nfev=nfev, njev=njev, status=0,
            x_covariance=None)
    finally:
        return_solver(solver)


def dogbox(fun, jac, x0, lb, ub, ftol, xtol, gtol, max_nfev, scaling, ctx=None):
    """Rectangular trust-region dogleg driver (reference: dogbox.py:100)."""
    f = fun(x0)
    nfev = 1
    J = jac(x0, f)
    njev = 1
    if f.shape[0] != J.shape[0]:
        raise RuntimeError("Inconsistent dimensions between the returns of "
                           "`fun` and `jac` on the first iteration.")
    m, n = J.shape
    solver = lease_solver(DogboxStepSolver, 1, m, n, ctx=ctx)
    try:
        use_jac = _is_jac(scaling)
        scale = np.ones(n) if use_jac else 1 / np.asarray(scaling, dtype=float)
        on_bound = np.zeros_like(x0, dtype=int)               # dogbox.py:152-154
        on_bound[np.equal(x0, lb)] = -1
        on_bound[np.equal(x0, ub)] = 1
        x = x0.copy()
        F = solver.factor(J[None], f[None], x[None], lb[None], ub[None], scale[None],
                          on_bound[None], SCALE_JAC_INIT if use_jac else SCALE_GIVEN)
        scale = F.scale[0]
        Delta = norm(x0 / scale, ord=np.inf)                  # dogbox.py:148-150
        if Delta == 0:
            Delta = 1.0
        obj_value = np.dot(f, f)
        if max_nfev is None:
            max_nfev = x0.size * 100
        status = None
        g_norm = float(F.g_norm[0])
        have_factor = True
        while nfev < max_nfev:
            if not have_factor:
                F = solver.factor(J[None], f[None], x[None], lb[None], ub[None], scale[None],
                                  on_bound[None],
                                  SCALE_JAC_UPDATE if use_jac else SCALE_GIVEN)
                scale = F.scale[0]
                have_factor = True
            if int(F.all_active[0]):                           # dogbox.py:182-188
                g_norm = 0.0
                status = 1
            else:
                g_norm = float(F.g_norm[0])
                if g_norm < gtol:
                    status = 1
            if status is not None:
                return OptimizeResult(
                    x=x, fun=f, jac=J, obj_value=obj_value, optimality=g_norm,
                    active_mask=on_bound, nfev=nfev, njev=njev, status=status,
                    x_covariance=None)
            actual_reduction = -1.0
            while actual_reduction <= 0 and nfev < max_nfev:
                S = solver.step(np.array([Delta]))
                x_new = S.x_new[0]
                f_new = fun(x_new)
                nfev += 1
                obj_value_new = np.dot(f_new, f_new)
                actual_reduction = obj_value - obj_value_new
                predicted = float(S.predicted_reduction[0])
                ratio = actual_reduction / predicted if predicted > 0 else 0
                if ratio < 0.25:                               # dogbox.py:234-237
                    Delta = 0.25 * float(S.step_scaled_norm[0])
                elif ratio > 0.75 and bool(S.tr_hit[0]):
                    Delta *= 2.0
                ftol_ok = abs(actual_reduction) < ftol * obj_value and ratio > 0.25
                xtol_ok = Delta < xtol * max(EPS ** 0.5, norm(x / scale, ord=np.inf))
                status = _termination(ftol_ok, xtol_ok)
                if status is not None:
                    break
            if actual_reduction > 0:
                on_bound = S.on_bound_new[0].astype(int)
                x = x_new.copy()
                x[on_bound == -1] = lb[on_bound == -1]        # dogbox.py:257-261
                x[on_bound == 1] = ub[on_bound == 1]
                f = f_new
                obj_value = obj_value_new
                J = jac(x, f)
                njev += 1
                have_factor = False
        return OptimizeResult(
            x=x, fun=f, jac=J, obj_value=obj_value, optimality=g_norm, active_mask=on_bound,
            nfev=nfev, njev=njev, status=0, x_covariance=None)
    finally:
        return_solver(solver)
