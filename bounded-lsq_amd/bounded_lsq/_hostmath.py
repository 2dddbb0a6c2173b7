"""Tiny n-vector helpers the host drivers need OUTSIDE the per-iteration path
(initial interior shift, final active mask, Coleman-Li v for Delta_0).  Inside
the iteration the same quantities are produced on the GPU."""
import numpy as np


def in_bounds(x, lb, ub):
    """bounds.py:19-21."""
    return bool(np.all((x >= lb) & (x <= ub)))


def prepare_bounds(bounds, x0):
    """bounds.py:7-16: scalar bounds are broadcast to x0's shape."""
    lb, ub = (np.asarray(b, dtype=float) for b in bounds)
    if lb.ndim == 0:
        lb = np.resize(lb, x0.shape)
    if ub.ndim == 0:
        ub = np.resize(ub, x0.shape)
    return lb, ub


def shift_into_interior(x, lb, ub, rstep=0.0):
    """bounds.py:79-103 (make_strictly_feasible)."""
    out = np.array(x, dtype=float, copy=True)
    low = x <= lb
    up = x >= ub
    if rstep == 0:
        out[low] = np.nextafter(lb[low], ub[low])
        out[up] = np.nextafter(ub[up], lb[up])
    else:
        out[low] = lb[low] + rstep * (1 + np.abs(lb[low]))
        out[up] = ub[up] - rstep * (1 + np.abs(ub[up]))
    return out


def active_mask(x, lb, ub, rtol=1e-12):
    """bounds.py:51-76 (find_active_constraints)."""
    mask = np.zeros(np.shape(x), dtype=int)
    dl = x - lb
    du = ub - x
    lower_nearer = dl < du
    with np.errstate(invalid="ignore"):
        on_l = dl < rtol * np.maximum(1, np.abs(lb))
        on_u = du < rtol * np.maximum(1, np.abs(ub))
    mask[lower_nearer & on_l] = -1
    mask[~lower_nearer & on_u] = 1
    return mask


def cl_vector(x, g, lb, ub):
    """bounds.py:106-149 (scaling_vector), v only."""
    v = np.ones_like(x)
    sel = (g < 0) & np.isfinite(ub)
    v[sel] = ub[sel] - x[sel]
    sel = (g > 0) & np.isfinite(lb)
    v[sel] = x[sel] - lb[sel]
    return v


def cl_optimality(x, g, lb, ub):
    """bounds.py:152-156 (CL_optimality)."""
    lb = np.resize(lb, np.shape(x))
    ub = np.resize(ub, np.shape(x))
    return float(np.linalg.norm(cl_vector(np.asarray(x, float), g, lb, ub) * g, ord=np.inf))
