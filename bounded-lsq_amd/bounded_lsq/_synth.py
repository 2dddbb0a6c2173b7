"""Seeded synthetic step-solve inputs.

Bench / test support only: the same seeded batch feeds the HIP path, the
oracle and the CPU baseline.  Problem ``b`` of a batch uses seed ``base + b``
and a fixed draw order, so any host regenerates the inputs bit-exactly
(numpy PCG64 streams are version-stable).

    J_ij ~ N(0,1)   f_i ~ N(0,1)   x_j ~ U(-1,1)
    lb = x - U(1e-3, 0.05)         ub = x + U(1e-3, 0.05)       scale = 1
"""
import numpy as np


def trf_problem(seed, m, n, unbounded=False):
    rng = np.random.default_rng(int(seed))
    J = rng.standard_normal((m, n))
    f = rng.standard_normal(m)
    x = rng.uniform(-1.0, 1.0, n)
    lo = rng.uniform(1e-3, 0.05, n)
    hi = rng.uniform(1e-3, 0.05, n)
    if unbounded:
        lb = np.full(n, -np.inf)
        ub = np.full(n, np.inf)
    else:
        lb = x - lo
        ub = x + hi
    return dict(J=J, f=f, x=x, lb=lb, ub=ub, scale=np.ones(n))


def trf_batch(base_seed, B, m, n, unbounded=False):
    """Batch-major arrays: J (B,m,n), f (B,m), x/lb/ub/scale (B,n)."""
    out = dict(J=np.empty((B, m, n)), f=np.empty((B, m)), x=np.empty((B, n)),
               lb=np.empty((B, n)), ub=np.empty((B, n)),
               scale=np.ones((B, n)))
    for b in range(B):
        p = trf_problem(base_seed + b, m, n, unbounded)
        for k in ("J", "f", "x", "lb", "ub"):
            out[k][b] = p[k]
    return out


def dogbox_problem(seed, m, n, frac_on_bound=0.10):
    """As trf_problem, with ~frac_on_bound of the variables placed exactly on
    their lower bound (on_bound = -1) as dogbox.py:152-154 would mark them."""
    p = trf_problem(seed, m, n)
    rng = np.random.default_rng(int(seed) + 0x5EED)
    k = max(1, int(round(frac_on_bound * n))) if frac_on_bound > 0 else 0
    idx = rng.choice(n, size=k, replace=False)
    p["x"][idx] = p["lb"][idx]
    ob = np.zeros(n, dtype=np.int64)
    ob[idx] = -1
    p["on_bound"] = ob
    return p


def dogbox_batch(base_seed, B, m, n, frac_on_bound=0.10):
    out = dict(J=np.empty((B, m, n)), f=np.empty((B, m)), x=np.empty((B, n)),
               lb=np.empty((B, n)), ub=np.empty((B, n)),
               scale=np.ones((B, n)), on_bound=np.zeros((B, n), np.int64))
    for b in range(B):
        p = dogbox_problem(base_seed + b, m, n, frac_on_bound)
        for k in ("J", "f", "x", "lb", "ub", "on_bound"):
            out[k][b] = p[k]
    return out
