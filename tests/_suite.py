"""A small nonlinear least-squares problem suite for end-to-end convergence regression
(SURVEY.md 8f-3).

The reference ships a 58-problem benchmark module (benchmarks/lsq_problems.py); its code is not
copied.  These are the classic Moré–Garbow–Hillstrom style test functions written down here from
their mathematical definitions, each with an unbounded and a bounded variant.  What the tests
compare is DATA: tests/golden/suite.json holds what the REFERENCE's public drivers returned
(nfev, njev, status, x, ...) when tests/golden/make_golden.py ran them on exactly these
functions, start points and bounds.
"""
import numpy as np

inf = np.inf


def _p(name, fun, jac, x0, boxes):
    return dict(name=name, fun=fun, jac=jac, x0=np.array(x0, float), boxes=boxes)


# ---- Powell singular (n = 4, m = 4) -------------------------------------------------------
def powell_singular():
    s5, s10 = 5.0 ** 0.5, 10.0 ** 0.5

    def fun(x):
        return np.array([x[0] + 10 * x[1], s5 * (x[2] - x[3]), (x[1] - 2 * x[2]) ** 2,
                         s10 * (x[0] - x[3]) ** 2])

    def jac(x):
        return np.array([[1, 10, 0, 0], [0, 0, s5, -s5],
                         [0, 2 * (x[1] - 2 * x[2]), -4 * (x[1] - 2 * x[2]), 0],
                         [2 * s10 * (x[0] - x[3]), 0, 0, -2 * s10 * (x[0] - x[3])]], float)
    return _p("powell_singular", fun, jac, [3, -1, 0, 1],
              [([-inf] * 4, [inf] * 4), ([0.1, -2, -1, 0.5], [4, 2, 1, 2])])


# ---- Freudenstein & Roth (n = 2, m = 2) ---------------------------------------------------
def freudenstein_roth():
    def fun(x):
        return np.array([-13 + x[0] + ((5 - x[1]) * x[1] - 2) * x[1],
                         -29 + x[0] + ((x[1] + 1) * x[1] - 14) * x[1]])

    def jac(x):
        return np.array([[1, 10 * x[1] - 3 * x[1] ** 2 - 2], [1, 3 * x[1] ** 2 + 2 * x[1] - 14]], float)
    return _p("freudenstein_roth", fun, jac, [0.5, -2],
              [([-inf] * 2, [inf] * 2), ([0, -3], [6, 0.5])])


# ---- helical valley (n = 3, m = 3) --------------------------------------------------------
def helical_valley():
    def theta(x):
        t = np.arctan2(x[1], x[0]) / (2 * np.pi)
        return t

    def fun(x):
        r = np.hypot(x[0], x[1])
        return np.array([10 * (x[2] - 10 * theta(x)), 10 * (r - 1), x[2]])

    def jac(x):
        r2 = x[0] ** 2 + x[1] ** 2
        r = r2 ** 0.5
        dt0 = -x[1] / (2 * np.pi * r2)
        dt1 = x[0] / (2 * np.pi * r2)
        return np.array([[-100 * dt0, -100 * dt1, 10], [10 * x[0] / r, 10 * x[1] / r, 0],
                         [0, 0, 1]], float)
    return _p("helical_valley", fun, jac, [-1, 0.5, 0.3],
              [([-inf] * 3, [inf] * 3), ([-2, 0.1, -1], [0.8, 2, 2])])


# ---- Wood (n = 4, m = 6) ------------------------------------------------------------------
def wood():
    s10, s90 = 10.0 ** 0.5, 90.0 ** 0.5

    def fun(x):
        return np.array([10 * (x[1] - x[0] ** 2), 1 - x[0], s90 * (x[3] - x[2] ** 2), 1 - x[2],
                         s10 * (x[1] + x[3] - 2), (x[1] - x[3]) / s10])

    def jac(x):
        return np.array([[-20 * x[0], 10, 0, 0], [-1, 0, 0, 0], [0, 0, -2 * s90 * x[2], s90],
                         [0, 0, -1, 0], [0, s10, 0, s10], [0, 1 / s10, 0, -1 / s10]], float)
    return _p("wood", fun, jac, [-3, -1, -3, -1],
              [([-inf] * 4, [inf] * 4), ([-4, -2, -4, -2], [0.5, 3, 2, 3])])


# ---- Beale (n = 2, m = 3) -----------------------------------------------------------------
def beale():
    y = np.array([1.5, 2.25, 2.625])

    def fun(x):
        i = np.arange(1, 4)
        return y - x[0] * (1 - x[1] ** i)

    def jac(x):
        i = np.arange(1, 4)
        return np.stack([-(1 - x[1] ** i), x[0] * i * x[1] ** (i - 1)], axis=1)
    return _p("beale", fun, jac, [1, 1], [([-inf] * 2, [inf] * 2), ([0.6, -1], [2.5, 1.0])])


# ---- Box three-dimensional (n = 3, m = 10) ------------------------------------------------
def box3d():
    t = 0.1 * np.arange(1, 11)

    def fun(x):
        return np.exp(-t * x[0]) - np.exp(-t * x[1]) - x[2] * (np.exp(-t) - np.exp(-10 * t))

    def jac(x):
        return np.stack([-t * np.exp(-t * x[0]), t * np.exp(-t * x[1]),
                         -(np.exp(-t) - np.exp(-10 * t))], axis=1)
    return _p("box3d", fun, jac, [0, 10, 20],
              [([-inf] * 3, [inf] * 3), ([0, 5, 0], [2, 10, 20])])


# ---- Kowalik & Osborne (n = 4, m = 11) ----------------------------------------------------
def kowalik_osborne():
    y = np.array([0.1957, 0.1947, 0.1735, 0.1600, 0.0844, 0.0627, 0.0456, 0.0342, 0.0323,
                  0.0235, 0.0246])
    u = np.array([4, 2, 1, 0.5, 0.25, 0.167, 0.125, 0.1, 0.0833, 0.0714, 0.0625])

    def fun(x):
        return y - x[0] * (u ** 2 + u * x[1]) / (u ** 2 + u * x[2] + x[3])

    def jac(x):
        num = u ** 2 + u * x[1]
        den = u ** 2 + u * x[2] + x[3]
        return np.stack([-num / den, -x[0] * u / den, x[0] * num * u / den ** 2,
                         x[0] * num / den ** 2], axis=1)
    return _p("kowalik_osborne", fun, jac, [0.25, 0.39, 0.415, 0.39],
              [([-inf] * 4, [inf] * 4), ([0.2, 0, 0.1, 0.1], [0.3, 1, 1, 0.5])])


# ---- Bard (n = 3, m = 15) -----------------------------------------------------------------
def bard():
    y = np.array([0.14, 0.18, 0.22, 0.25, 0.29, 0.32, 0.35, 0.39, 0.37, 0.58, 0.73, 0.96, 1.34,
                  2.10, 4.39])
    u = np.arange(1.0, 16.0)
    v = 16.0 - u
    w = np.minimum(u, v)

    def fun(x):
        return y - (x[0] + u / (v * x[1] + w * x[2]))

    def jac(x):
        d = v * x[1] + w * x[2]
        return np.stack([-np.ones(15), u * v / d ** 2, u * w / d ** 2], axis=1)
    return _p("bard", fun, jac, [1, 1, 1], [([-inf] * 3, [inf] * 3), ([0.1, 0.5, 0], [1, 1.4, 2])])


# ---- Brown almost-linear (n = 5, m = 5) ---------------------------------------------------
def brown_almost_linear():
    n = 5

    def fun(x):
        f = x + x.sum() - (n + 1)
        f[-1] = np.prod(x) - 1
        return f

    def jac(x):
        J = np.ones((n, n)) + np.eye(n)
        J[-1] = [np.prod(np.delete(x, j)) for j in range(n)]
        return J
    return _p("brown_almost_linear", fun, jac, [0.5] * 5,
              [([-inf] * 5, [inf] * 5), ([0, 0, 0, 0.3, 0], [0.9, 2, 2, 2, 2])])


# ---- extended Rosenbrock (n = 10, m = 10) -------------------------------------------------
def ext_rosenbrock():
    n = 10

    def fun(x):
        f = np.empty(n)
        f[0::2] = 10 * (x[1::2] - x[0::2] ** 2)
        f[1::2] = 1 - x[0::2]
        return f

    def jac(x):
        J = np.zeros((n, n))
        for k in range(0, n, 2):
            J[k, k] = -20 * x[k]
            J[k, k + 1] = 10
            J[k + 1, k] = -1
        return J
    lo = np.full(n, -2.0); lo[1::2] = -1.0
    hi = np.full(n, 0.9); hi[1::2] = 2.0
    return _p("ext_rosenbrock", fun, jac, [-1.2, 1] * 5, [([-inf] * n, [inf] * n), (lo, hi)])


# ---- Watson (n = 6, m = 31) ---------------------------------------------------------------
def watson():
    n = 6
    t = np.arange(1, 30) / 29.0

    def fun(x):
        j = np.arange(n)
        s1 = ((j[1:] * x[1:])[None, :] * t[:, None] ** (j[1:] - 1)[None, :]).sum(1)
        s2 = (x[None, :] * t[:, None] ** j[None, :]).sum(1)
        return np.concatenate([s1 - s2 ** 2 - 1, [x[0], x[1] - x[0] ** 2 - 1]])

    def jac(x):
        j = np.arange(n)
        P = t[:, None] ** j[None, :]
        s2 = (x[None, :] * P).sum(1)
        D = np.zeros((29, n))
        D[:, 1:] = j[1:][None, :] * t[:, None] ** (j[1:] - 1)[None, :]
        D -= 2 * s2[:, None] * P
        r30 = np.zeros(n); r30[0] = 1
        r31 = np.zeros(n); r31[0] = -2 * x[0]; r31[1] = 1
        return np.vstack([D, r30, r31])
    return _p("watson", fun, jac, [0.0] * 6,
              [([-inf] * n, [inf] * n), ([-0.1, 0, 0, 0, -1, 0], [0.1, 2, 0.1, 1, 1, 2])])


# ---- Osborne-type exponential fit (n = 5, m = 33) -----------------------------------------
def exp_sum():
    t = 10.0 * np.arange(33)
    y = 0.375 + 1.93 * np.exp(-0.013 * t) - 1.46 * np.exp(-0.022 * t) \
        + 0.002 * np.cos(0.7 * np.arange(33))

    def fun(x):
        return y - (x[0] + x[1] * np.exp(-x[3] * t) + x[2] * np.exp(-x[4] * t))

    def jac(x):
        e3, e4 = np.exp(-x[3] * t), np.exp(-x[4] * t)
        return np.stack([-np.ones_like(t), -e3, -e4, x[1] * t * e3, x[2] * t * e4], axis=1)
    return _p("exp_sum", fun, jac, [0.5, 1.5, -1, 0.01, 0.02],
              [([-inf] * 5, [inf] * 5), ([0.3, 1, -2, 0.005, 0.015], [0.6, 2.5, -0.5, 0.02, 0.05])])


SUITE = [powell_singular(), freudenstein_roth(), helical_valley(), wood(), beale(), box3d(),
         kowalik_osborne(), bard(), brown_almost_linear(), ext_rosenbrock(), watson(), exp_sum()]
SUITE_BY_NAME = {p["name"]: p for p in SUITE}
