"""Problems of tests/test_ranks_gpu.py: every rank builds ITS rows from a per-block seed (no rank ever
holds the whole matrix), the parent builds the whole problem for the oracle."""
import numpy as np


def _spec(case):
    name, shape = case.rsplit("_", 1)
    m, n = (int(v) for v in shape.split("x"))
    return name, m, n


def _block(case, r, world):
    """rows [lo, hi) of rank r: Gaussian rows from a per-block seed"""
    name, m, n = _spec(case)
    lo, hi = (r * m) // world, ((r + 1) * m) // world
    rng = np.random.default_rng(9000 + 17 * r + n)
    J = rng.standard_normal((hi - lo, n))
    f = rng.standard_normal(hi - lo)
    if name.startswith("reject"):                      # J <- J V diag(s) V^T, s log-spaced to 1 / 3e4
        r0 = np.random.default_rng(31 + n)
        V, _ = np.linalg.qr(r0.standard_normal((n, n)))
        J = (J @ (V * np.logspace(0.0, -4.5, n))) @ V.T
    return J, f


def _vectors(case):
    name, m, n = _spec(case)
    r0 = np.random.default_rng(77 + n)
    x = r0.uniform(-1.0, 1.0, n)
    lb = x - r0.uniform(1e-3, 0.05, n)
    ub = x + r0.uniform(1e-3, 0.05, n)
    if name.startswith("reject"):
        lb = np.full(n, -np.inf); ub = np.full(n, np.inf)
    return x, lb, ub


def make_case(case, rank, world):
    name, m, n = _spec(case)
    J, f = _block(case, rank, world)
    x, lb, ub = _vectors(case)
    mode = 1 if "jac" in name else 0                  # BLSQ_SCALE_JAC_INIT / BLSQ_SCALE_GIVEN
    if name == "modes":
        mode = rank % 2                               # the ranks DISAGREE (error test)
    C = dict(J=J, f=f, x=x, lb=lb, ub=ub, scale=np.ones(n), n=n, m_total=m, scale_mode=mode,
             deltas=(0.5, 10.0) if not name.startswith("reject") else (0.05, 1e3))
    if name == "gram_jac_twice":
        C["factor_calls"] = 2                         # (the second call re-initialises from the same J: same scale)
    return C


def whole_problem(case, world):
    name, m, n = _spec(case)
    blocks = [_block(case, r, world) for r in range(world)]
    J = np.vstack([b[0] for b in blocks]); f = np.concatenate([b[1] for b in blocks])
    x, lb, ub = _vectors(case)
    scale = np.ones(n)
    if "jac" in name:                                 # trf.py:216-219
        nrm = np.linalg.norm(J, axis=0)
        nrm[nrm == 0] = 1.0
        scale = 1.0 / nrm
    return dict(J=J, f=f, x=x, lb=lb, ub=ub, scale_oracle=scale)
