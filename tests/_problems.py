"""Small nonlinear least-squares problems shared by the golden-vector
generator (run against the reference) and the end-to-end tests (run against
this repo's drivers).  Bounded Rosenbrock specs follow the six cases the
reference tests use (bounded_lsq/tests/test_least_squares.py:252-264,
benchmarks/lsq_problems.py:454-462)."""
import numpy as np

inf = np.inf

ROSEN_SPECS = [
    ([-2.0, 1.0], [-inf, -1.5], [inf, inf]),
    ([2.0, 2.0], [-inf, 1.5], [inf, inf]),
    ([-2.0, 2.0], [-inf, 1.5], [inf, inf]),
    ([0.0, 2.0], [-inf, 1.5], [1.0, inf]),
    ([2.0, 2.0], [1.0, 1.5], [3.0, 3.0]),
    ([-1.2, 1.0], [-50.0, 0.0], [0.5, 100]),
]


def rosen(x):
    return np.array([10 * (x[1] - x[0] ** 2), 1 - x[0]])


def rosen_jac(x):
    return np.array([[-20 * x[0], 10.0], [-1.0, 0.0]])


def expfit_problem(seed, m=40):
    """y = a exp(b t) + c cos(d t) + noise; 4 parameters."""
    rng = np.random.default_rng(seed)
    t = np.linspace(0, 4, m)
    truth = np.array([2.0, -0.7, 0.5, 0.3])
    y = truth[0] * np.exp(truth[1] * t) + truth[2] * np.cos(truth[3] * t)
    y = y + 0.01 * rng.standard_normal(m)

    def fun(p):
        return p[0] * np.exp(p[1] * t) + p[2] * np.cos(p[3] * t) - y

    def jac(p):
        return np.stack([np.exp(p[1] * t), p[0] * t * np.exp(p[1] * t),
                         np.cos(p[3] * t), -p[2] * t * np.sin(p[3] * t)], 1)
    return fun, jac


EXPFIT_X0 = [1.0, -0.1, 1.0, 1.0]
EXPFIT_BOX = ([0.0, -2.0, 0.0, 0.0], [1.5, 0.0, 3.0, 2.0])
