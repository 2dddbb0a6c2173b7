"""CPU oracle for the trust-region step path of nmayorov/bounded-lsq.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package imports this file;
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may use it, and only as the checker / reported baseline.

This is a restatement (not a copy) of the reference's per-iteration linear
algebra, written from its semantics with numpy/scipy — the same third-party
calls the reference makes (``scipy.linalg.svd`` -> LAPACK gesdd at
trf.py:272, ``numpy.linalg.lstsq`` -> LAPACK gelsd at dogbox.py:197).  Each
function cites the reference lines it follows (paths under /root/reference).

Pinning: the reference's own tests hold no iteration-level vectors (SURVEY.md
section 4), so this oracle is pinned by golden vectors captured by importing the
reference in the build container (``tests/golden/make_golden.py``), committed
under ``tests/golden/`` and checked by ``tests/test_oracle_golden.py``.

The floating-point operation ORDER of the reference is kept wherever it is
observable (masks use exact ``==`` on computed minima).
"""
from __future__ import annotations

import math
from typing import NamedTuple, Optional

import numpy as np
from scipy.linalg import svd as _svd

EPS = float(np.finfo(float).eps)


# --------------------------------------------------------------------------
# bounds geometry  (reference: bounded_lsq/bounds.py)
# --------------------------------------------------------------------------

def within(x, lb, ub) -> bool:
    """bounds.py:19-21 (in_bounds)."""
    return bool(np.all(x >= lb) and np.all(x <= ub))


def step_to_bound(x, d, lb, ub):
    """bounds.py:24-48 (step_size_to_bound).

    t_i = max((lb-x)_i/d_i, (ub-x)_i/d_i) where d_i != 0, +inf elsewhere;
    returns (min_i t_i, hits) with hits = [t_i == min] * sign(d_i) (int64),
    so every tie is flagged.
    """
    x = np.asarray(x, dtype=float)
    d = np.asarray(d, dtype=float)
    t = np.full(x.shape, np.inf)
    nz = d != 0
    with np.errstate(over="ignore", invalid="ignore"):
        lo = (lb - x)[nz] / d[nz]
        hi = (ub - x)[nz] / d[nz]
        t[nz] = np.maximum(lo, hi)
    tmin = np.min(t) if t.size else np.inf
    hits = (t == tmin).astype(np.int64) * np.sign(d).astype(np.int64)
    return float(tmin), hits


def active_constraints(x, lb, ub, rtol=1e-12):
    """bounds.py:51-76 (find_active_constraints): -1/0/+1 by distance to the
    NEARER bound against rtol*max(1,|bound|)."""
    x = np.asarray(x, dtype=float)
    out = np.zeros(x.shape, dtype=np.int64)
    below = x - lb
    above = ub - x
    nearer_lower = below < above
    with np.errstate(invalid="ignore"):
        low_on = below < rtol * np.maximum(1.0, np.abs(lb))
        up_on = above < rtol * np.maximum(1.0, np.abs(ub))
    out[nearer_lower & low_on] = -1
    out[~nearer_lower & up_on] = 1
    return out


def nudge_inside(x, lb, ub, rstep=0.0):
    """bounds.py:79-103 (make_strictly_feasible).  rstep == 0: one ulp toward
    the opposite bound; else a relative shift rstep*(1+|bound|).  The upper
    test is evaluated on the ORIGINAL x (bounds.py:97), after the lower fix."""
    x = np.asarray(x, dtype=float)
    out = x.copy()
    at_low = x <= lb
    at_up = x >= ub
    if rstep == 0:
        out[at_low] = np.nextafter(lb[at_low], ub[at_low])
        out[at_up] = np.nextafter(ub[at_up], lb[at_up])
    else:
        out[at_low] = lb[at_low] + rstep * (1 + np.abs(lb[at_low]))
        out[at_up] = ub[at_up] - rstep * (1 + np.abs(ub[at_up]))
    return out


def cl_scaling(x, g, lb, ub):
    """bounds.py:106-149 (scaling_vector): Coleman-Li v and dv/dx.
    The g>0 rule is applied second and so wins where both could apply
    (they cannot: g<0 and g>0 are exclusive)."""
    x = np.asarray(x, dtype=float)
    v = np.ones_like(x)
    jv = np.zeros_like(x)
    up = (g < 0) & np.isfinite(ub)
    v[up] = ub[up] - x[up]
    jv[up] = -1.0
    lo = (g > 0) & np.isfinite(lb)
    v[lo] = x[lo] - lb[lo]
    jv[lo] = 1.0
    return v, jv


def cl_optimality(x, g, lb, ub):
    """bounds.py:152-156 (CL_optimality)."""
    lb = np.resize(lb, np.shape(x))
    ub = np.resize(ub, np.shape(x))
    v, _ = cl_scaling(x, g, lb, ub)
    return float(np.linalg.norm(v * g, ord=np.inf))


# --------------------------------------------------------------------------
# trust-region sub-problem  (reference: bounded_lsq/trust_region.py)
# --------------------------------------------------------------------------

def sphere_intersections(x, s, Delta):
    """trust_region.py:11-44 (intersect_trust_region): roots of
    ||x + t s||^2 = Delta^2 by the cancellation-free quadratic formula."""
    a = float(np.dot(s, s))
    if a == 0:
        raise ValueError("`s` is zero.")
    b = float(np.dot(x, s))
    c = float(np.dot(x, x)) - Delta ** 2
    if c > 0:
        raise ValueError("`x` is not within the trust region.")
    disc = math.sqrt(b * b - a * c)
    q = -(b + math.copysign(disc, b))
    r1 = q / a
    r2 = c / q
    return (r1, r2) if r1 < r2 else (r2, r1)


def _secular(alpha, suf, s, Delta):
    """trust_region.py:47-53 (_phi_and_derivative)."""
    den = s ** 2 + alpha
    pn = float(np.linalg.norm(suf / den))
    phi = pn - Delta
    dphi = -float(np.sum(suf ** 2 / den ** 3)) / pn
    return phi, dphi


def tr_subproblem(n, m, uf, s, V, Delta, initial_alpha=None, rtol=0.01,
                  max_iter=10):
    """trust_region.py:56-152 (solve_lsq_trust_region), incl. its quirks:
    the precedence of ``None or (not full_rank and alpha0 == 0)`` (:127) and
    the stale-phi rescale when max_iter is exhausted (:132-150)."""
    suf = s * uf
    if m >= n:
        full_rank = bool(s[-1] > EPS * m * s[0])
    else:
        full_rank = False

    if full_rank:
        p = -V.dot(uf / s)
        if np.linalg.norm(p) <= Delta:
            return p, 0.0, 0

    hi = float(np.linalg.norm(suf)) / Delta
    if full_rank:
        phi, dphi = _secular(0.0, suf, s, Delta)
        lo = -phi / dphi
    else:
        lo = 0.0

    def _restart():
        return max(0.001 * hi, (lo * hi) ** 0.5)

    if initial_alpha is None or (not full_rank and initial_alpha == 0):
        alpha = _restart()
    else:
        alpha = initial_alpha

    it = 0
    for it in range(max_iter):
        if alpha < lo or alpha > hi:
            alpha = _restart()
        phi, dphi = _secular(alpha, suf, s, Delta)
        if abs(phi) < rtol * Delta:
            break
        if phi < 0:
            hi = alpha
        ratio = phi / dphi
        lo = max(lo, alpha - ratio)
        alpha -= (phi + Delta) * ratio / Delta

    p = -V.dot(suf / (s ** 2 + alpha))
    if phi > 0:
        p *= Delta / np.linalg.norm(p)
    return p, float(alpha), it + 1


# --------------------------------------------------------------------------
# TRF step helpers  (reference: bounded_lsq/trf.py:15-170)
# --------------------------------------------------------------------------

def quad_1d_min(a, b, lo, hi):
    """trf.py:15-34 (minimize_quadratic): argmin of a t^2 + b t over
    {lo, hi, interior extremum}; first index wins ties."""
    cand = [lo, hi]
    if a != 0:
        ext = -0.5 * b / a
        if lo <= ext <= hi:
            cand.append(ext)
    t = np.array(cand, dtype=float)
    y = a * t ** 2 + b * t
    k = int(np.argmin(y))
    return float(t[k]), float(y[k])


def line_quadratic(J, diag, g, s, s0=None):
    """trf.py:37-76 (build_1d_quadratic_function)."""
    Js = J.dot(s)
    a = 0.5 * (np.dot(Js, Js) + np.dot(s * diag, s))
    b = np.dot(g, s)
    if s0 is not None:
        Js0 = J.dot(s0)
        b += np.dot(Js0, Js) + np.dot(s0 * diag, s)
    return float(a), float(b)


def model_values(J, diag, g, steps):
    """trf.py:79-102 (evaluate_quadratic_function); steps is (k, n)."""
    JS = J.dot(steps.T)
    return 0.5 * (np.sum(JS ** 2, axis=0) +
                  np.sum(diag * steps ** 2, axis=1)) + np.dot(steps, g)


def reflected_step(x, J_h, diag_h, g_h, p, p_h, d, Delta, lb, ub, theta):
    """trf.py:105-156 (find_reflected_step).  Returns (p_h', r_h) as new
    arrays (the reference mutates p / p_h in place; callers here use the
    returned values only).  r_h is p_h' itself when no reflection exists."""
    p_stride, hits = step_to_bound(x, p, lb, ub)
    r_h = p_h.copy()
    r_h[hits.astype(bool)] *= -1
    r = d * r_h

    p = p * p_stride
    p_h = p_h * p_stride
    x_face = x + p

    _, to_tr = sphere_intersections(p_h, r_h, Delta)
    to_face, _ = step_to_bound(x_face, r, lb, ub)
    to_face *= theta
    r_hi = min(to_face, to_tr)
    if r_hi > 0:
        r_lo = (1 - theta) * p_stride / r_hi
    else:
        r_lo = -1

    if r_lo <= r_hi:
        a, b = line_quadratic(J_h, diag_h, g_h, r_h, s0=p_h)
        t, _ = quad_1d_min(a, b, r_lo, r_hi)
        refl = p_h + r_h * t
    else:
        refl = None

    p_h = p_h * theta
    return (p_h, p_h) if refl is None else (p_h, refl)


def gradient_step(x, J_h, diag_h, g_h, d, Delta, lb, ub, theta):
    """trf.py:159-170 (find_gradient_step)."""
    to_face, _ = step_to_bound(x, -g_h * d, lb, ub)
    to_face *= theta
    to_tr = Delta / np.linalg.norm(g_h)
    t_max = min(to_face, to_tr)
    a, b = line_quadratic(J_h, diag_h, g_h, -g_h)
    t, _ = quad_1d_min(a, b, 0.0, t_max)
    return -t * g_h


# --------------------------------------------------------------------------
# TRF step-solve = trf.py:244-308, split at the seam the C-ABI uses:
#   factor (once per outer iteration)  +  step (once per inner iteration)
# --------------------------------------------------------------------------

class TrfFactor(NamedTuple):
    m: int
    n: int
    x: np.ndarray
    lb: np.ndarray
    ub: np.ndarray
    g: np.ndarray
    v: np.ndarray
    jv: np.ndarray
    d: np.ndarray
    g_h: np.ndarray
    diag_h: np.ndarray
    g_norm: float
    theta: float
    J_h: np.ndarray
    s: np.ndarray
    V: np.ndarray
    uf: np.ndarray


def trf_factor(J, f, x, lb, ub, scale) -> TrfFactor:
    """trf.py:244-277: gradient, Coleman-Li hat variables, augmented SVD."""
    J = np.asarray(J, dtype=float)
    m, n = J.shape
    g = J.T.dot(f)
    v, jv = cl_scaling(x, g, lb, ub)
    d = v ** 0.5 * scale
    g_h = d * g
    diag_h = g * jv * scale ** 2
    g_norm = float(np.linalg.norm(g * v, ord=np.inf))
    J_h = J * d
    J_aug = np.empty((m + n, n))
    J_aug[:m] = J_h
    J_aug[m:] = np.diag(diag_h ** 0.5)
    f_aug = np.zeros(m + n)
    f_aug[:m] = f
    U, s, Vt = _svd(J_aug, full_matrices=False)
    uf = U.T.dot(f_aug)
    theta = max(0.995, 1 - g_norm)
    return TrfFactor(m, n, np.asarray(x, float), lb, ub, g, v, jv, d, g_h,
                     diag_h, g_norm, theta, J_h, s, Vt.T, uf)


class TrfStep(NamedTuple):
    p_h_tr: np.ndarray        # raw trust-region solution (hat space)
    alpha: float
    n_iter: int
    to_bound: float
    hits: np.ndarray          # hits of x + p against the box (int64)
    branch: int               # 0 feasible, 1 reflective
    steps_h: np.ndarray       # (1,n) or (3,n): p_h, r_h, c_h
    qp: np.ndarray
    choice: int
    step_h: np.ndarray
    predicted_reduction: float
    step: np.ndarray
    x_new: np.ndarray
    step_h_norm: float
    correction: float


def trf_step(F: TrfFactor, Delta, alpha) -> TrfStep:
    """trf.py:284-308 (+ the correction / norm terms of :318-324 that the
    outer driver consumes)."""
    p_h, alpha, n_iter = tr_subproblem(F.n, F.m, F.uf, F.s, F.V, Delta,
                                       initial_alpha=alpha)
    p_h_tr = p_h.copy()
    p = F.d * p_h
    to_bound, hits = step_to_bound(F.x, p, F.lb, F.ub)
    if to_bound >= 1:
        p_h = p_h * min(F.theta * to_bound, 1)
        steps_h = np.atleast_2d(p_h)
        branch = 0
    else:
        p_h, r_h = reflected_step(F.x, F.J_h, F.diag_h, F.g_h, p, p_h, F.d,
                                  Delta, F.lb, F.ub, F.theta)
        c_h = gradient_step(F.x, F.J_h, F.diag_h, F.g_h, F.d, Delta, F.lb,
                            F.ub, F.theta)
        steps_h = np.array([p_h, r_h, c_h])
        branch = 1
    qp = model_values(F.J_h, F.diag_h, F.g_h, steps_h)
    k = int(np.argmin(qp))
    step_h = steps_h[k]
    pred = -2 * qp[k]
    step = F.d * step_h
    x_new = nudge_inside(F.x + step, F.lb, F.ub)
    return TrfStep(p_h_tr, float(alpha), int(n_iter), float(to_bound), hits,
                   branch, steps_h, qp, k, step_h, float(pred), step, x_new,
                   float(np.linalg.norm(step_h)),
                   float(np.dot(step_h * F.diag_h, step_h)))


def trf_step_solve(J, f, x, lb, ub, scale, Delta, alpha):
    """One 'step-solve' in the sense of SURVEY.md section 8(d)."""
    F = trf_factor(J, f, x, lb, ub, scale)
    return F, trf_step(F, Delta, alpha)


# --------------------------------------------------------------------------
# dogbox step helpers  (reference: bounded_lsq/dogbox.py:9-97)
# --------------------------------------------------------------------------

def box_tr_intersection(x, tr, lb, ub):
    """dogbox.py:9-35 (find_intersection)."""
    lc = lb - x
    uc = ub - x
    lt = np.maximum(lc, -tr)
    ut = np.minimum(uc, tr)
    return lt, ut, lt == lc, ut == uc, lt == -tr, ut == tr


def dogleg(x, cauchy, newton, tr, lb, ub):
    """dogbox.py:38-75 (dogleg_step)."""
    lt, ut, orig_l, orig_u, tr_l, tr_u = box_tr_intersection(x, tr, lb, ub)
    face = np.zeros(x.shape, dtype=np.int64)
    if within(newton, lt, ut):
        return newton, face, False
    if not within(cauchy, lt, ut):
        beta, _ = step_to_bound(np.zeros_like(cauchy), cauchy, lt, ut)
        cauchy = beta * cauchy
    diff = newton - cauchy
    t, hits = step_to_bound(cauchy, diff, lt, ut)
    face[(hits < 0) & orig_l] = -1
    face[(hits > 0) & orig_u] = 1
    tr_hit = bool(np.any(((hits < 0) & tr_l) | ((hits > 0) & tr_u)))
    return cauchy + t * diff, face, tr_hit


def clipped_cauchy(x, cauchy, tr, lb, ub):
    """dogbox.py:78-97 (constrained_cauchy_step)."""
    lt, ut, orig_l, orig_u, tr_l, tr_u = box_tr_intersection(x, tr, lb, ub)
    face = np.zeros(x.shape, dtype=np.int64)
    if within(cauchy, lt, ut):
        return cauchy, face, False
    beta, hits = step_to_bound(np.zeros_like(cauchy), cauchy, lt, ut)
    face[(hits < 0) & orig_l] = -1
    face[(hits > 0) & orig_u] = 1
    tr_hit = bool(np.any(((hits < 0) & tr_l) | ((hits > 0) & tr_u)))
    return beta * cauchy, face, tr_hit


class DogboxFactor(NamedTuple):
    m: int
    n: int
    f: np.ndarray
    x: np.ndarray
    lb: np.ndarray
    ub: np.ndarray
    scale: np.ndarray
    g: np.ndarray
    active: np.ndarray        # bool (n,)
    free: np.ndarray          # bool (n,)
    g_norm: float
    all_active: bool
    J_free: np.ndarray
    newton: Optional[np.ndarray]
    cauchy: Optional[np.ndarray]


def dogbox_factor(J, f, x, lb, ub, scale, on_bound) -> DogboxFactor:
    """dogbox.py:170-199: gradient, active/free split, Newton (lstsq, numpy
    default rcond) and Cauchy steps on the free columns."""
    J = np.asarray(J, dtype=float)
    m, n = J.shape
    g = J.T.dot(f)
    active = on_bound * g < 0
    free = ~active
    J_free = J[:, free]
    g_free = g[free]
    if np.all(active):
        return DogboxFactor(m, n, f, x, lb, ub, scale, g, active, free, 0.0,
                            True, J_free, None, None)
    g_norm = float(np.linalg.norm(g_free, ord=np.inf))
    newton = np.linalg.lstsq(J_free, -f, rcond=None)[0]
    Jg = J_free.dot(g_free)
    with np.errstate(divide="ignore", invalid="ignore"):
        cauchy = -np.dot(g_free, g_free) / np.dot(Jg, Jg) * g_free
    return DogboxFactor(m, n, f, x, lb, ub, scale, g, active, free, g_norm,
                        False, J_free, newton, cauchy)


class DogboxStep(NamedTuple):
    step_free: np.ndarray
    on_bound_free: np.ndarray
    tr_hit: bool
    predicted_reduction: float
    fallback: bool
    step: np.ndarray
    x_new: np.ndarray
    on_bound_new: np.ndarray   # on_bound after `on_bound[free] = on_bound_free`
    step_scaled_norm: float    # ||step/scale||_inf (dogbox.py:235)


def dogbox_step(F: DogboxFactor, Delta, on_bound) -> DogboxStep:
    """dogbox.py:203-220, incl. the quirk that Js is NOT recomputed after
    the constrained-Cauchy fallback (:216)."""
    fs = F.free
    tr = Delta * F.scale[fs]
    step_free, obf, tr_hit = dogleg(F.x[fs], F.cauchy, F.newton, tr,
                                    F.lb[fs], F.ub[fs])
    Js = F.J_free.dot(step_free)
    pred = -np.dot(Js, Js) - 2 * np.dot(Js, F.f)
    fallback = False
    if pred <= 0:
        step_free, obf, tr_hit = clipped_cauchy(F.x[fs], F.cauchy, tr,
                                                F.lb[fs], F.ub[fs])
        pred = -np.dot(Js, Js) - 2 * np.dot(Js, F.f)
        fallback = True
    step = np.zeros(F.n)
    step[fs] = step_free
    x_new = F.x + step
    ob_new = np.array(on_bound, dtype=np.int64, copy=True)
    ob_new[fs] = obf
    return DogboxStep(step_free, obf, bool(tr_hit), float(pred), fallback,
                      step, x_new, ob_new,
                      float(np.linalg.norm(step / F.scale, ord=np.inf)))


def dogbox_step_solve(J, f, x, lb, ub, scale, on_bound, Delta):
    F = dogbox_factor(J, f, x, lb, ub, scale, on_bound)
    return F, (None if F.all_active else dogbox_step(F, Delta, on_bound))
