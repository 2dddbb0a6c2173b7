"""`least_squares_batch`: B independent problems of one shape, solved together.

The reference solves one problem at a time; its drivers'
control flow (trf.py:238-352, dogbox.py:164-267) is kept here PER PROBLEM, but
the problems advance in lock-step so that every factorisation / step is ONE
batched C-ABI call (`blsq_*_factor`, `blsq_*_step` with B > 1):

    tick:  factor (problems with a fresh Jacobian)  ->  step (all active)
           ->  fun(x_new)  ->  accept / reject per problem  ->  jac for accepted

Callbacks are vectorised: ``fun(X) -> (B, m)``, ``jac(X) -> (B, m, n)`` for
``X`` of shape (B, n).  Each problem's result (x, nfev, njev, status, masks, ...)
is what ``least_squares`` returns for that problem alone: evaluations made for
other problems in the same tick are not counted, and a problem that has
terminated is frozen.  The batched factor call re-factors problems whose
Jacobian did not change in that tick (idempotent; costs time, not accuracy).
"""
import numpy as np
from numpy.linalg import norm
from scipy.optimize import OptimizeResult

from ._frontend import (TERMINATION_MESSAGES, _clamp_tolerances, _checked_scaling, EPS)
from ._hip_step import (TrfStepSolver, DogboxStepSolver, SCALE_GIVEN, SCALE_JAC_INIT,
                        SCALE_JAC_UPDATE, raise_batch_status)
from ._hostmath import shift_into_interior, active_mask, cl_vector


def _bounds_2d(bounds, B, n):
    if len(bounds) != 2:
        raise ValueError("`bounds` must contain 2 elements.")
    lb = np.broadcast_to(np.asarray(bounds[0], dtype=float), (B, n)).copy()
    ub = np.broadcast_to(np.asarray(bounds[1], dtype=float), (B, n)).copy()
    if np.any(lb >= ub):
        raise ValueError("Each lower bound mush be strictly less than each "
                         "upper bound.")
    return lb, ub


def least_squares_batch(fun, x0, jac, bounds=(-np.inf, np.inf), method='trf',
                        ftol=EPS ** 0.5, xtol=EPS ** 0.5, gtol=EPS ** 0.5, max_nfev=None,
                        scaling=1.0, diff_step=None, args=(), kwargs=None, ctx=None, driver='host'):
    """Solve B bound-constrained least-squares problems of identical shape.

    fun : callable, ``fun(X) -> (B, m)`` residuals for ``X`` (B, n)
    x0  : (B, n) initial guesses;  jac : callable ``jac(X) -> (B, m, n)``, or '2-point' / '3-point' 
    bounds : pair broadcastable to (B, n);  scaling : 'jac' or broadcastable to (n,)
    diff_step : relative step of the finite-difference Jacobian (as `least_squares`, least_squares.py:357-365)
    args, kwargs : passed on to `fun` and `jac` (``fun(X, *args, **kwargs)``)
    driver : 'host' — the per-problem accept / update logic runs here in Python around batched
             C-ABI calls; 'device' — it runs on the GPU (``OuterDriver``, blsq_outer_*), x / f / J
             stay resident and only fresh Jacobians are uploaded and factored.
    Returns a list of B ``OptimizeResult`` (fields as ``least_squares``).
    """
    if method not in ('trf', 'dogbox'):
        raise ValueError("`method` must be 'trf' or 'dogbox'.")
    X0 = np.array(x0, dtype=float)
    if X0.ndim != 2:
        raise ValueError("`x0` must have shape (B, n).")
    B, n = X0.shape
    lb, ub = _bounds_2d(bounds, B, n)
    if not callable(fun):
        raise ValueError("`fun` must be callable (vectorised over the batch).")
    kwargs = dict(kwargs) if kwargs else {}
    if args or kwargs:                                        # least_squares.py:351-355, 367-371
        user_fun, user_jac = fun, jac

        def fun(X):                                           # noqa: F811
            return user_fun(X, *args, **kwargs)
        if callable(user_jac):
            def jac(X):                                       # noqa: F811
                return user_jac(X, *args, **kwargs)
    fd_state = {}                                  # lazily created FdJacobian (+ ctx if we own it)
    if isinstance(jac, str) and jac in ('2-point', '3-point'):
        # the reference's FD Jacobian (third-party approx_derivative, least_squares.py:357-365)
        # restated for the batch on the device (`FdJacobian`: bit-identical steps, points and
        # quotients); `fun` is called once per perturbed coordinate with all B problems.  FD
        # evaluations are not counted in nfev (:229-232).
        from . import _abi
        from ._fd import FdJacobian
        fd_method = jac
        user_ctx = ctx

        def jac(X):                                                     # noqa: F811
            F = np.ascontiguousarray(fun(X), dtype=float)
            if "fd" not in fd_state:
                if user_ctx is None:
                    fd_state["own_ctx"] = _abi.Context(0)
                fd_state["fd"] = FdJacobian(user_ctx or fd_state["own_ctx"], B, F.shape[1], n,
                                            fd_method, diff_step)

            def fun_points(Xp):
                return np.stack([np.asarray(fun(np.ascontiguousarray(Xp[:, p, :])), dtype=float)
                                 for p in range(Xp.shape[1])], axis=1)
            return fd_state["fd"].jac_host(fun_points, X, F, lb, ub)
    elif not callable(jac):
        raise ValueError("`jac` must be '2-point', '3-point' or callable.")
    scaling = _checked_scaling(scaling, X0[0])
    ftol, xtol, gtol = _clamp_tolerances(ftol, xtol, gtol)
    if not np.all((X0 >= lb) & (X0 <= ub)):
        raise ValueError("`x0` is infeasible.")
    use_jac = isinstance(scaling, str)
    trf = method == 'trf'
    if max_nfev is None:
        max_nfev = n * 100
    if driver not in ('host', 'device'):
        raise ValueError("`driver` must be 'host' or 'device'.")

    def _release_fd():
        if "fd" in fd_state:
            fd_state.pop("fd").close()
        if "own_ctx" in fd_state:
            fd_state.pop("own_ctx").close()

    if driver == 'device':
        try:
            return _device_batch(fun, jac, X0, lb, ub, trf, use_jac, scaling, ftol, xtol, gtol,
                                 max_nfev, ctx)
        finally:
            _release_fd()

    def feval(X):
        F = np.ascontiguousarray(fun(X), dtype=float)
        if F.ndim != 2 or F.shape[0] != B:
            raise RuntimeError("`fun` must return an array of shape (B, m).")
        return F

    def jeval(X):
        J = np.ascontiguousarray(jac(X), dtype=float)
        if J.ndim != 3 or J.shape[0] != B or J.shape[2] != n:
            raise RuntimeError("`jac` must return an array of shape (B, m, n).")
        return J

    if trf:                                                   # trf.py:201
        x = np.stack([shift_into_interior(X0[b], lb[b], ub[b], rstep=1e-10) for b in range(B)])
    else:
        x = X0.copy()
    f = feval(x)
    J = jeval(x)
    m = f.shape[1]
    if J.shape[1] != m:
        raise RuntimeError("Inconsistent dimensions between the returns of "
                           "`fun` and `jac` on the first iteration.")
    nfev = np.ones(B, dtype=int)
    njev = np.ones(B, dtype=int)
    scale = np.ones((B, n)) if use_jac else np.broadcast_to(1 / np.asarray(scaling, float),
                                                            (B, n)).copy()
    on_bound = np.zeros((B, n), dtype=np.int64)
    if not trf:                                               # dogbox.py:152-154
        on_bound[X0 == lb] = -1
        on_bound[X0 == ub] = 1
    solver = (TrfStepSolver if trf else DogboxStepSolver)(B, m, n, ctx=ctx)
    try:
        def factor(mode):
            if trf:
                return solver.factor(J, f, x, lb, ub, scale, mode)
            return solver.factor(J, f, x, lb, ub, scale, on_bound, mode)

        F = factor(SCALE_JAC_INIT if use_jac else SCALE_GIVEN)
        scale = F.scale.copy()
        if trf:                                               # trf.py:223-226
            Delta = np.array([norm(X0[b] / (scale[b] * cl_vector(x[b], F.g[b], lb[b], ub[b]) ** 0.5))
                              for b in range(B)])
        else:                                                 # dogbox.py:148-150
            Delta = np.array([norm(X0[b] / scale[b], ord=np.inf) for b in range(B)])
        Delta[Delta == 0] = 1.0
        alpha = np.zeros(B)
        obj = np.einsum('bi,bi->b', f, f)
        status = [None] * B          # termination status found in the inner loop
        done = np.zeros(B, dtype=bool)
        result_status = np.zeros(B, dtype=int)
        g_norm = np.zeros(B)
        # per-problem phase: True -> at the top of the outer loop (needs the gtol / status check)
        at_top = np.ones(B, dtype=bool)
        need_factor = np.zeros(B, dtype=bool)                 # fresh J since the last factor
        actual = np.full(B, -1.0)

        while not np.all(done):
            if np.any(need_factor & ~done):
                F = factor(SCALE_JAC_UPDATE if use_jac else SCALE_GIVEN)
                scale = np.where((need_factor & ~done)[:, None], F.scale, scale)
                need_factor[:] = False
            # ---- top of the outer loop (trf.py:238-261 / dogbox.py:164-194) ----
            for b in np.nonzero(at_top & ~done)[0]:
                if nfev[b] >= max_nfev:                       # `while nfev < max_nfev` failed
                    done[b] = True
                    result_status[b] = 0
                    continue
                if trf:
                    g_norm[b] = F.g_norm[b]
                    if g_norm[b] < gtol:
                        status[b] = 1
                else:
                    if int(F.all_active[b]):
                        g_norm[b] = 0.0
                        status[b] = 1
                    else:
                        g_norm[b] = F.g_norm[b]
                        if g_norm[b] < gtol:
                            status[b] = 1
                if status[b] is not None:
                    done[b] = True
                    result_status[b] = status[b]
                    continue
                at_top[b] = False
                actual[b] = -1.0
            act = ~done
            if not np.any(act):
                break
            # ---- one inner iteration for every active problem --------------------
            if trf:
                S = solver.step(Delta, alpha, active_rtol=xtol)
                alpha = np.where(act, S.alpha, alpha)
            else:
                S = solver.step(Delta)
            # where the reference raises ValueError out of the step (trust_region.py:28-29,
            # 34-35) its solve aborts: so does the batch, naming the problem
            raise_batch_status(S.status, act)
            x_new = np.where(act[:, None], S.x_new, x)
            f_new = feval(x_new)
            accepted = np.zeros(B, dtype=bool)
            for b in np.nonzero(act)[0]:
                nfev[b] += 1
                obj_new = np.dot(f_new[b], f_new[b])
                actual[b] = obj[b] - obj_new
                pred = float(S.predicted_reduction[b])
                if trf:
                    ratio = (actual[b] - float(S.correction[b])) / pred if pred > 0 else 0
                    shn = float(S.step_h_norm[b])
                    if ratio < 0.25:
                        D_new = 0.25 * shn
                        alpha[b] *= Delta[b] / D_new
                        Delta[b] = D_new
                    elif ratio > 0.75 and shn > 0.95 * Delta[b]:
                        Delta[b] *= 2.0
                        alpha[b] *= 0.5
                    xtol_ok = norm(S.step[b]) < xtol * max(EPS ** 0.5, norm(x[b]))
                else:
                    ratio = actual[b] / pred if pred > 0 else 0
                    if ratio < 0.25:
                        Delta[b] = 0.25 * float(S.step_scaled_norm[b])
                    elif ratio > 0.75 and bool(S.tr_hit[b]):
                        Delta[b] *= 2.0
                    xtol_ok = Delta[b] < xtol * max(EPS ** 0.5, norm(x[b] / scale[b], ord=np.inf))
                ftol_ok = abs(actual[b]) < ftol * obj[b] and ratio > 0.25
                if ftol_ok and xtol_ok:
                    status[b] = 4
                elif ftol_ok:
                    status[b] = 2
                elif xtol_ok:
                    status[b] = 3
                leave_inner = (status[b] is not None) or actual[b] > 0 or nfev[b] >= max_nfev
                if actual[b] > 0:
                    accepted[b] = True
                    if trf:
                        x[b] = x_new[b]
                    else:
                        on_bound[b] = S.on_bound_new[b]
                        xb = x_new[b].copy()
                        xb[on_bound[b] == -1] = lb[b][on_bound[b] == -1]
                        xb[on_bound[b] == 1] = ub[b][on_bound[b] == 1]
                        x[b] = xb
                    f[b] = f_new[b]
                    obj[b] = obj_new
                if leave_inner:
                    at_top[b] = True
            if np.any(accepted):
                J_new = jeval(x)
                for b in np.nonzero(accepted)[0]:
                    J[b] = J_new[b]
                    njev[b] += 1
                    need_factor[b] = True
        results = []
        for b in range(B):
            mask = active_mask(x[b], lb[b], ub[b], rtol=xtol) if trf else on_bound[b].astype(int)
            r = OptimizeResult(x=x[b].copy(), fun=f[b].copy(), jac=J[b].copy(), obj_value=obj[b],
                               optimality=g_norm[b], active_mask=mask, nfev=int(nfev[b]),
                               njev=int(njev[b]), status=int(result_status[b]), x_covariance=None)
            r.message = TERMINATION_MESSAGES[r.status]
            r.success = r.status > 0
            results.append(r)
        return results
    finally:
        solver.close()
        _release_fd()


def _device_batch(fun, jac, X0, lb, ub, trf, use_jac, scaling, ftol, xtol, gtol, max_nfev, ctx):
    """`least_squares_batch` on the device-resident outer driver (same results, same counts)."""
    from ._outer import OuterDriver
    B, n = X0.shape
    if trf:                                                   # trf.py:201
        xs = np.stack([shift_into_interior(X0[b], lb[b], ub[b], rstep=1e-10) for b in range(B)])
    else:
        xs = X0.copy()
    F0 = np.ascontiguousarray(fun(xs), dtype=float)
    if F0.ndim != 2 or F0.shape[0] != B:
        raise RuntimeError("`fun` must return an array of shape (B, m).")
    m = F0.shape[1]
    calls = {"first": True}

    def fun_cached(X):                     # the driver asks for fun(x_start) first: reuse F0
        if calls["first"]:
            calls["first"] = False
            return F0
        F = np.ascontiguousarray(fun(X), dtype=float)
        if F.shape != (B, m):
            raise RuntimeError("`fun` must return an array of shape (B, m).")
        return F

    def jac_checked(X):
        J = np.ascontiguousarray(jac(X), dtype=float)
        if J.ndim != 3 or J.shape[0] != B or J.shape[2] != n:
            raise RuntimeError("`jac` must return an array of shape (B, m, n).")
        if J.shape[1] != m:
            raise RuntimeError("Inconsistent dimensions between the returns of "
                               "`fun` and `jac` on the first iteration.")
        return J

    scale = np.ones((B, n)) if use_jac else np.broadcast_to(1 / np.asarray(scaling, float), (B, n))
    drv = OuterDriver('trf' if trf else 'dogbox', B, m, n, ctx=ctx)
    try:
        drv.start(X0, xs, lb, ub, scale, use_jac, ftol, xtol, gtol, max_nfev)
        R = drv.run_host(fun_cached, jac_checked)
        Jfin = drv._down(drv.d_J, (B, m, n))
    finally:
        drv.close()
    results = []
    for b in range(B):
        x = R["x"][b]
        mask = active_mask(x, lb[b], ub[b], rtol=xtol) if trf else R["on_bound"][b].astype(int)
        r = OptimizeResult(x=x.copy(), fun=R["f"][b].copy(), jac=Jfin[b].copy(),
                           obj_value=float(R["obj"][b]), optimality=float(R["optimality"][b]),
                           active_mask=mask, nfev=int(R["nfev"][b]), njev=int(R["njev"][b]),
                           status=int(R["status"][b]), x_covariance=None)
        r.message = TERMINATION_MESSAGES[r.status]
        r.success = r.status > 0
        results.append(r)
    return results
