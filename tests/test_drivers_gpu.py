"""End-to-end parity of the host drivers (least_squares / trf / dogbox over the
GPU step path) against records captured from the reference's own public
drivers (tests/golden/e2e.json, first_iter.npz) and against the behaviours
the reference's test-suite asserts (test_least_squares.py)."""
import numpy as np
import pytest

from _golden import load_json, load_npz, unhex
from _problems import ROSEN_SPECS, rosen, rosen_jac, expfit_problem, EXPFIT_X0, EXPFIT_BOX

pytestmark = pytest.mark.gpu

E2E = load_json("e2e.json")
TOL = float.fromhex(E2E["tol"])


def _problem(tag):
    if tag.startswith("rosen"):
        return rosen, rosen_jac
    return expfit_problem(E2E["expfit_seed"])


@pytest.mark.parametrize("rec", E2E["records"],
                         ids=["%s-%s-%s" % (r["tag"], r["method"], r["scaling"])
                              for r in E2E["records"]])
def test_end_to_end_records(rec):
    import bounded_lsq
    fun, jac = _problem(rec["tag"])
    scaling = rec["scaling"] if isinstance(rec["scaling"], str) else np.array(rec["scaling"])
    if not isinstance(scaling, str) and scaling.size == 1:
        scaling = float(scaling[0])
    res = bounded_lsq.least_squares(fun, unhex(rec["x0"]), jac=jac,
                                    bounds=(unhex(rec["lb"]), unhex(rec["ub"])),
                                    method=rec["method"], ftol=TOL, xtol=TOL, gtol=TOL,
                                    scaling=scaling)
    assert res.nfev == rec["nfev"] and res.njev == rec["njev"]
    assert res.status == rec["status"]
    np.testing.assert_allclose(res.x, unhex(rec["x"]), rtol=1e-8, atol=1e-12)
    np.testing.assert_allclose(res.obj_value, float.fromhex(rec["obj_value"]), rtol=1e-8,
                               atol=1e-14)
    np.testing.assert_array_equal(res.active_mask, rec["active_mask"])   # bit-exact mask
    assert res.success == (rec["status"] > 0) and isinstance(res.message, str)


FIRST = load_npz("first_iter.npz")


@pytest.mark.parametrize("name,ins,out", FIRST, ids=[c[0] for c in FIRST])
def test_first_inner_iteration_matches_reference_driver(name, ins, out):
    """x passed to the 2nd fun() call == x_new of the reference's inline block."""
    import bounded_lsq
    from bounded_lsq import _synth
    P = _synth.trf_problem(int(ins["seed"]), int(ins["m"]), int(ins["n"]))
    calls = []

    class Stop(Exception):
        pass

    def fun(x):
        calls.append(x.copy())
        if len(calls) == 2:
            raise Stop
        return P["f"].copy()

    def jac(x, f):
        return P["J"].copy()
    drv = bounded_lsq.trf if name.startswith("trf") else bounded_lsq.dogbox
    with pytest.raises(Stop):
        drv(fun, jac, P["x"].copy(), P["lb"], P["ub"], 1e-8, 1e-8, 1e-8, None,
            np.ones(P["x"].size))
    np.testing.assert_array_equal(calls[0], out["x_first"])
    dx_ref = out["x_new"] - out["x_first"]
    dx = calls[1] - calls[0]
    assert np.linalg.norm(dx - dx_ref) <= 1e-10 * np.linalg.norm(dx_ref)


# ---- behaviours the reference's own tests assert (test_least_squares.py) ----
def fun_trivial(x, a=0):
    return (x - a) ** 2 + 5.0


def jac_trivial(x, a=0.0):
    return 2 * (x - a)


def fun_rosenbrock(x):
    return np.array([10 * (x[1] - x[0] ** 2), (1 - x[0])])


def jac_rosenbrock(x):
    return np.array([[-20 * x[0], 10], [-1, 0]])


@pytest.mark.parametrize("method", ["trf", "dogbox"])
def test_basic_and_jac_options(method):                    # :59-63, :90-95
    from bounded_lsq import least_squares
    for jac in ['2-point', '3-point', jac_trivial]:
        res = least_squares(fun_trivial, 2.0, jac, method=method)
        np.testing.assert_allclose(res.x, 0, atol=1e-4)
        np.testing.assert_allclose(res.obj_value, 25)


@pytest.mark.parametrize("method", ["trf", "dogbox"])
def test_full_result(method):                              # :141-160
    from bounded_lsq import least_squares
    res = least_squares(fun_trivial, 2.0, method=method)
    assert res.x.shape == (1,) and res.fun.shape == (1,) and res.jac.shape == (1, 1)
    np.testing.assert_allclose(res.x, 0, atol=1e-4)
    np.testing.assert_allclose(res.obj_value, 25)
    np.testing.assert_allclose(res.fun, 5)
    np.testing.assert_allclose(res.jac, 0, atol=1e-4)
    np.testing.assert_allclose(res.optimality, 0, atol=1e-3)
    np.testing.assert_array_equal(res.active_mask, 0)
    assert res.nfev < 10 and res.njev < 10 and res.status > 0 and res.success


@pytest.mark.parametrize("method", ["trf", "dogbox"])
def test_args_kwargs_and_nfev(method):                     # :97-101, :112-126
    from bounded_lsq import least_squares
    a = 3.0
    res = least_squares(fun_trivial, 2.0, jac_trivial, args=(a,), method=method)
    np.testing.assert_allclose(res.x, a, rtol=1e-4)
    res = least_squares(fun_trivial, 2.0, jac_trivial, kwargs={'a': a}, method=method)
    np.testing.assert_allclose(res.x, a, rtol=1e-4)
    with pytest.raises(TypeError):
        least_squares(fun_trivial, 2.0, args=(3, 4), method=method)
    with pytest.raises(TypeError):
        least_squares(fun_trivial, 2.0, kwargs={'kaboom': 3}, method=method)
    with pytest.raises(TypeError):                         # unknown option (:128-132)
        least_squares(fun_trivial, 2.0, method=method, options={'no_such_option': 100})
    res = least_squares(fun_trivial, 2.0, max_nfev=1, method=method)
    assert res.nfev == 1 and res.status == 0


@pytest.mark.parametrize("method", ["trf", "dogbox"])
def test_rosenbrock(method):                               # :162-169
    from bounded_lsq import least_squares
    x0 = [-2, 1]
    for scaling in [1.0, np.array([1.0, 5.0]), 'jac']:
        for jac in ['2-point', '3-point', jac_rosenbrock]:
            res = least_squares(fun_rosenbrock, x0, jac, scaling=scaling, method=method)
            np.testing.assert_allclose(res.x, [1, 1], rtol=1e-7)


@pytest.mark.parametrize("method", ["trf", "dogbox"])
def test_in_bounds_and_shapes(method):                     # :222-249
    from bounded_lsq import least_squares
    for jac in ['2-point', '3-point', jac_trivial]:
        res = least_squares(fun_trivial, 2.0, jac=jac, bounds=(-1.0, 3.0), method=method)
        np.testing.assert_allclose(res.x, 0.0, atol=1e-4)
        np.testing.assert_array_equal(res.active_mask, [0])
        res = least_squares(fun_trivial, 2.0, jac=jac, bounds=(0.5, 3.0), method=method)
        np.testing.assert_allclose(res.x, 0.5, atol=1e-4)
        np.testing.assert_array_equal(res.active_mask, [-1])
        assert 0.5 <= res.x <= 3

    for bounds, expect in [((0.5, [2.0, 2.0]), [0.5, 0.5]), (([0.3, 0.2], 3.0), [0.3, 0.2]),
                           (([-1, 0.5], [1.0, 3.0]), [0.0, 0.5])]:
        res = least_squares(lambda x: np.asarray(x), [1.0, 1.0], bounds=bounds, method=method)
        np.testing.assert_allclose(res.x, expect, atol=1e-5)


@pytest.mark.parametrize("method", ["trf", "dogbox"])
def test_rosenbrock_bounds(method):                        # :251-270
    from bounded_lsq import least_squares
    for x0, lb, ub in ROSEN_SPECS:
        for scaling in [1.0, [1.0, 2.0], 'jac']:
            for jac in ['2-point', '3-point', jac_rosenbrock]:
                res = least_squares(fun_rosenbrock, x0, jac, (lb, ub), scaling=scaling,
                                    method=method)
                np.testing.assert_allclose(res.optimality, 0.0, atol=1e-5)


def test_fun_and_jac_shape_errors():                       # :171-199
    from bounded_lsq import least_squares
    with pytest.raises(RuntimeError):
        least_squares(lambda x: np.ones((2, 2)), 2.0)
    with pytest.raises(RuntimeError):
        least_squares(fun_trivial, 2.0, jac=lambda x: np.ones((1, 1, 1)))
    with pytest.raises(RuntimeError):                       # m mismatch (trf.py:209-211)
        least_squares(lambda x: np.ones(3), [2.0, 1.0], jac=lambda x: np.ones((2, 2)))


# ---- batched front-end: every problem must match its own sequential solve -------
def _expfit_batch(B, m=40):
    t = np.linspace(0, 4, m)
    ys = []
    for b in range(B):
        rng = np.random.default_rng(100 + b)
        truth = np.array([2.0 + 0.1 * b, -0.7 + 0.02 * b, 0.5, 0.3 + 0.01 * b])
        y = truth[0] * np.exp(truth[1] * t) + truth[2] * np.cos(truth[3] * t)
        ys.append(y + 0.01 * rng.standard_normal(m))
    Y = np.array(ys)

    def fun(P):
        P = np.atleast_2d(P)
        return (P[:, 0:1] * np.exp(P[:, 1:2] * t) + P[:, 2:3] * np.cos(P[:, 3:4] * t)) - Y[:P.shape[0]]

    def jac(P):
        P = np.atleast_2d(P)
        e = np.exp(P[:, 1:2] * t)
        return np.stack([e, P[:, 0:1] * t * e, np.cos(P[:, 3:4] * t),
                         -P[:, 2:3] * t * np.sin(P[:, 3:4] * t)], axis=2)
    return fun, jac, Y, t


@pytest.mark.parametrize("method", ["trf", "dogbox"])
@pytest.mark.parametrize("scaling", [1.0, "jac"])
@pytest.mark.parametrize("bounded", [False, True])
def test_least_squares_batch_matches_sequential(method, scaling, bounded):
    from bounded_lsq import least_squares, least_squares_batch
    B = 10
    fun, jac, Y, t = _expfit_batch(B)
    X0 = np.tile(np.array([1.0, -0.1, 1.0, 1.0]), (B, 1))
    X0[:, 0] += 0.05 * np.arange(B)
    bounds = (np.array([0.0, -2.0, 0.0, 0.0]), np.array([1.8, 0.0, 3.0, 2.0])) if bounded \
        else (-np.inf, np.inf)
    res = least_squares_batch(fun, X0, jac, bounds=bounds, method=method, scaling=scaling)
    assert len(res) == B
    for b in range(B):
        def fun_b(p, b=b):
            return p[0] * np.exp(p[1] * t) + p[2] * np.cos(p[3] * t) - Y[b]

        def jac_b(p, b=b):
            e = np.exp(p[1] * t)
            return np.stack([e, p[0] * t * e, np.cos(p[3] * t), -p[2] * t * np.sin(p[3] * t)], 1)
        ref = least_squares(fun_b, X0[b], jac_b, bounds=bounds, method=method, scaling=scaling)
        r = res[b]
        assert (r.nfev, r.njev, r.status) == (ref.nfev, ref.njev, ref.status), b
        np.testing.assert_allclose(r.x, ref.x, rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(r.obj_value, ref.obj_value, rtol=1e-9)
        np.testing.assert_array_equal(r.active_mask, ref.active_mask)
        assert r.success == ref.success and r.message == ref.message


@pytest.mark.parametrize("method", ["trf", "dogbox"])
@pytest.mark.parametrize("scaling", [1.0, "jac"])
@pytest.mark.parametrize("bounded", [False, True])
def test_device_outer_driver_matches_sequential(method, scaling, bounded):
    """The device-resident outer driver (blsq_outer_*: ratio test, Delta / alpha update,
    termination and accept on the GPU, masked re-factorisation) returns, problem by problem,
    what the sequential host driver returns: same nfev / njev / status, same x."""
    from bounded_lsq import least_squares, least_squares_batch
    B = 10
    fun, jac, Y, t = _expfit_batch(B)
    X0 = np.tile(np.array([1.0, -0.1, 1.0, 1.0]), (B, 1))
    X0[:, 0] += 0.05 * np.arange(B)
    bounds = (np.array([0.0, -2.0, 0.0, 0.0]), np.array([1.8, 0.0, 3.0, 2.0])) if bounded \
        else (-np.inf, np.inf)
    res = least_squares_batch(fun, X0, jac, bounds=bounds, method=method, scaling=scaling,
                              driver='device')
    assert len(res) == B
    for b in range(B):
        def fun_b(p, b=b):
            return p[0] * np.exp(p[1] * t) + p[2] * np.cos(p[3] * t) - Y[b]

        def jac_b(p, b=b):
            e = np.exp(p[1] * t)
            return np.stack([e, p[0] * t * e, np.cos(p[3] * t), -p[2] * t * np.sin(p[3] * t)], 1)
        ref = least_squares(fun_b, X0[b], jac_b, bounds=bounds, method=method, scaling=scaling)
        r = res[b]
        assert (r.nfev, r.njev, r.status) == (ref.nfev, ref.njev, ref.status), b
        np.testing.assert_allclose(r.x, ref.x, rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(r.obj_value, ref.obj_value, rtol=1e-9)
        # (the optimality at a converged point is a cancellation residue: |g| ~ 1e-9 with rounding
        # noise ~ 1e-14 from eps * |J| |f|; the two drivers round a few operations differently)
        np.testing.assert_allclose(r.optimality, ref.optimality, rtol=1e-6, atol=1e-12)
        np.testing.assert_array_equal(r.active_mask, ref.active_mask)
        assert r.success == ref.success and r.message == ref.message


def test_device_outer_driver_max_nfev_and_frozen_problems():
    """max_nfev stops a problem with status 0 exactly like the host driver, and problems that
    finish early are frozen while the rest of the batch continues."""
    from bounded_lsq import least_squares_batch
    B = 6
    fun, jac, Y, t = _expfit_batch(B)
    X0 = np.tile(np.array([1.0, -0.1, 1.0, 1.0]), (B, 1))
    X0[:, 0] += 0.3 * np.arange(B)
    for mx in (3, 7):
        host = least_squares_batch(fun, X0, jac, method='trf', max_nfev=mx)
        dev = least_squares_batch(fun, X0, jac, method='trf', max_nfev=mx, driver='device')
        for h, d in zip(host, dev):
            assert (h.nfev, h.njev, h.status) == (d.nfev, d.njev, d.status)
            np.testing.assert_allclose(d.x, h.x, rtol=1e-9, atol=1e-12)


def test_least_squares_batch_validation():
    from bounded_lsq import least_squares_batch
    with pytest.raises(ValueError):
        least_squares_batch(lambda X: X, np.zeros(3), lambda X: X)             # x0 not (B, n)
    with pytest.raises(ValueError):
        least_squares_batch(lambda X: X, np.zeros((2, 2)), lambda X: X, method='lm')
    with pytest.raises(ValueError):
        least_squares_batch(lambda X: X, np.zeros((2, 2)), lambda X: X, bounds=(1.0, 0.0))
    with pytest.raises(ValueError):
        least_squares_batch(lambda X: X, np.full((2, 2), 5.0), lambda X: X, bounds=(0.0, 1.0))


def test_device_outer_driver_torch_callbacks_subprocess():
    """Fully device-resident loop: user callbacks written with torch fill the driver's device
    buffers in place (zero copy).  Runs tools/bench_outer.py on a small batch in a child process
    (torch must initialise the HIP runtime before libblsq_hip.so is loaded)
    and checks that the TRF iteration counts equal the host driver's for every problem."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "bench_outer.py"),
                          "48", "96", "12"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stdout.splitlines()
    trf_i = [i for i, l in enumerate(lines) if l.startswith("trf: device driver, torch callbacks")]
    assert trf_i, out.stdout
    assert "differ host vs device driver: 0 of 48" in out.stdout
    assert "nfev differs from the host driver for 0 problems" in lines[trf_i[0] + 1]
    assert any(l.startswith("dogbox: device driver, torch callbacks") for l in lines)


@pytest.mark.parametrize("method", ["2-point", "3-point"])
@pytest.mark.parametrize("case", ["unbounded", "bounded", "at_bounds", "rel_step"])
def test_fd_jacobian_matches_scipy_approx_derivative(method, case):
    """blsq_fd_points_dev / blsq_fd_assemble_dev restate scipy's approx_derivative (the
    third-party routine behind the reference's jac='2-point'|'3-point', least_squares.py:357-365)
    for a batch: same steps (bounds-aware, one-sided switching), same points, same quotient —
    bit for bit, problem by problem."""
    from scipy.optimize._numdiff import approx_derivative
    from bounded_lsq import _abi
    from bounded_lsq._fd import FdJacobian
    rng = np.random.default_rng(3)
    B, m, n = 5, 37, 9
    A = rng.standard_normal((B, m, n))
    Y = rng.standard_normal((B, m))
    X = rng.uniform(-2.0, 2.0, (B, n))
    X[0, 0] = 0.0                                            # sign convention at 0
    lb = np.full((B, n), -np.inf); ub = np.full((B, n), np.inf)
    rel = None
    if case in ("bounded", "at_bounds"):
        lb = X - rng.uniform(1e-9, 1.0, (B, n)); ub = X + rng.uniform(1e-9, 1.0, (B, n))
    if case == "at_bounds":                                  # x exactly on a bound
        ub[:, ::2] = X[:, ::2]
        lb[:, 1::3] = X[:, 1::3]
    if case in ("bounded", "at_bounds"):
        lb[1] = -np.inf                                      # half-bounded problems
        ub[2] = np.inf
    if case == "rel_step":
        rel = np.full(n, 1e-6); rel[3] = 0.0                 # 0 -> scipy substitutes its default

    def fun_b(x, b):
        return A[b] @ np.tanh(x) + 0.1 * (x @ x) - Y[b]

    def fun_points(Xp):                                      # (B, P, n) -> (B, P, m)
        return np.stack([np.stack([fun_b(Xp[b, p], b) for p in range(Xp.shape[1])]) for b in range(B)])

    F0 = np.stack([fun_b(X[b], b) for b in range(B)])
    ctx = _abi.Context(0)
    fd = FdJacobian(ctx, B, m, n, method, rel)
    try:
        J = fd.jac_host(fun_points, X, F0, lb, ub)
    finally:
        fd.close(); ctx.close()
    for b in range(B):
        Jref = approx_derivative(lambda x: fun_b(x, b), X[b], method=method, rel_step=rel,
                                 f0=F0[b], bounds=(lb[b], ub[b]))
        np.testing.assert_array_equal(J[b], Jref)


@pytest.mark.parametrize("method", ["trf", "dogbox"])
def test_least_squares_batch_fd_jacobian(method):
    """jac='2-point' in the batched entry (host callbacks: scipy's approx_derivative per problem,
    as the reference does) agrees with the sequential least_squares(jac='2-point')."""
    from bounded_lsq import least_squares, least_squares_batch
    B = 4
    fun, jac, Y, t = _expfit_batch(B)
    X0 = np.tile(np.array([1.0, -0.1, 1.0, 1.0]), (B, 1))
    X0[:, 0] += 0.05 * np.arange(B)
    bounds = (np.array([0.0, -2.0, 0.0, 0.0]), np.array([1.8, 0.0, 3.0, 2.0]))
    for driver in ("host", "device"):
        res = least_squares_batch(fun, X0, '2-point', bounds=bounds, method=method, driver=driver)
        for b in range(B):
            def fun_b(p, b=b):
                return p[0] * np.exp(p[1] * t) + p[2] * np.cos(p[3] * t) - Y[b]
            ref = least_squares(fun_b, X0[b], '2-point', bounds=bounds, method=method)
            r = res[b]
            assert (r.nfev, r.njev, r.status) == (ref.nfev, ref.njev, ref.status), (driver, b)
            np.testing.assert_allclose(r.x, ref.x, rtol=1e-9, atol=1e-12)


def test_least_squares_batch_diff_step_args_kwargs():
    """diff_step, args and kwargs reach the callbacks / the FD Jacobian of the batched entry exactly
    as in least_squares (least_squares.py:351-371)."""
    from bounded_lsq import least_squares, least_squares_batch
    B = 3
    _, _, Y, t = _expfit_batch(B)
    X0 = np.tile(np.array([1.0, -0.1, 1.0, 1.0]), (B, 1))
    X0[:, 0] += 0.05 * np.arange(B)
    bounds = (np.array([0.0, -2.0, 0.0, 0.0]), np.array([1.8, 0.0, 3.0, 2.0]))

    def fun(X, data, shift=0.0):
        return X[:, :1] * np.exp(X[:, 1:2] * t) + X[:, 2:3] * np.cos(X[:, 3:4] * t) - data + shift

    for driver in ("host", "device"):
        res = least_squares_batch(fun, X0, '2-point', bounds=bounds, method='trf', diff_step=1e-6,
                                  args=(Y,), kwargs=dict(shift=0.01), driver=driver)
        for b in range(B):
            def fun_b(p, data, shift=0.0, b=b):
                return p[0] * np.exp(p[1] * t) + p[2] * np.cos(p[3] * t) - data[b] + shift
            ref = least_squares(fun_b, X0[b], '2-point', bounds=bounds, method='trf', diff_step=1e-6,
                                args=(Y,), kwargs=dict(shift=0.01))
            assert (res[b].nfev, res[b].njev, res[b].status) == (ref.nfev, ref.njev, ref.status), (driver, b)
            np.testing.assert_allclose(res[b].x, ref.x, rtol=1e-9, atol=1e-12)


# ---- the reference's records replayed DIRECTLY through the device-resident driver ---------------
def _rosen_batch(X):
    return np.stack([rosen(x) for x in X])


def _rosen_jac_batch(X):
    return np.stack([rosen_jac(x) for x in X])


@pytest.mark.parametrize("method", ["trf", "dogbox"])
@pytest.mark.parametrize("scaling", [1.0, "jac", [1.0, 5.0]], ids=["s1", "jac", "s15"])
def test_reference_records_through_the_device_driver(method, scaling):
    """The six bounded-Rosenbrock records of tests/golden/e2e.json (reference's own trf() /
    dogbox()) as ONE batch of six problems with different start points and boxes through
    least_squares_batch(driver='device'): every problem must return the reference's nfev, njev,
    status, x, objective and active mask — the device-resident outer logic pinned to the
    reference directly, not via the host driver."""
    import bounded_lsq
    want = [scaling] if isinstance(scaling, str) else [float(v) for v in np.atleast_1d(scaling)]
    recs = [r for r in E2E["records"] if r["tag"].startswith("rosen_B") and r["method"] == method
            and (r["scaling"] == scaling if isinstance(scaling, str) else r["scaling"] == want)]
    recs.sort(key=lambda r: r["tag"])
    assert len(recs) == 6
    unh = lambda v: np.array([float.fromhex(s) for s in v])                     # noqa: E731
    X0 = np.stack([unh(r["x0"]) for r in recs])
    lb = np.stack([unh(r["lb"]) for r in recs])
    ub = np.stack([unh(r["ub"]) for r in recs])
    tol = float.fromhex(E2E["tol"])
    for driver in ("device", "host"):
        res = bounded_lsq.least_squares_batch(_rosen_batch, X0, _rosen_jac_batch, bounds=(lb, ub),
                                              method=method, ftol=tol, xtol=tol, gtol=tol,
                                              scaling=scaling, driver=driver)
        for r, rec in zip(res, recs):
            assert (r.nfev, r.njev, r.status) == (rec["nfev"], rec["njev"], rec["status"]), (driver, rec["tag"])
            np.testing.assert_allclose(r.x, unh(rec["x"]), rtol=1e-8, atol=1e-11)
            np.testing.assert_allclose(r.obj_value, float.fromhex(rec["obj_value"]), rtol=1e-7, atol=1e-18)
            np.testing.assert_array_equal(r.active_mask, rec["active_mask"])


def test_suite58_family_through_the_device_driver():
    """One family of the reference's benchmark set with several start-point / box variants
    (Rosenbrock_B_0..5 of tests/golden/suite58.json) as one batch on the device driver."""
    import bounded_lsq
    S = load_json("suite58.json")
    probs = [p for p in S["problems"] if p["family"] == "Rosenbrock" and p["bounded"]]
    unh = lambda v: np.array([float.fromhex(s) for s in v])                     # noqa: E731
    X0 = np.stack([unh(p["x0"]) for p in probs])
    lb = np.stack([unh(p["lb"]) for p in probs])
    ub = np.stack([unh(p["ub"]) for p in probs])
    tol = float.fromhex(S["tol"])
    for method in ("trf", "dogbox"):
        res = bounded_lsq.least_squares_batch(_rosen_batch, X0, _rosen_jac_batch, bounds=(lb, ub),
                                              method=method, ftol=tol, xtol=tol, gtol=tol,
                                              scaling=1.0, driver="device")
        for r, p in zip(res, probs):
            rec = [q for q in S["records"] if q["problem"] == p["name"] and q["method"] == method
                   and q["scaling"] == 1.0][0]
            if rec.get("stable"):
                assert (r.nfev, r.njev, r.status) == (rec["nfev"], rec["njev"], rec["status"]), p["name"]
                np.testing.assert_allclose(r.x, unh(rec["x"]), rtol=1e-7, atol=1e-10)


@pytest.mark.parametrize("method", ["trf", "dogbox"])
def test_sequential_drivers_lease_their_plans_from_a_pool(method):
    """The drop-in drivers do not create and destroy a plan per solve: a solve of a shape seen before reuses the plan
    of the last one (`_hip_step.lease_solver / return_solver`).  A plan's history — other problems, other conditioning,
    an exception in the middle of a solve — never changes a number: every result equals the one of a fresh context
    bit for bit; the pool stays bounded and closes with its context."""
    import bounded_lsq as bl
    from bounded_lsq import _abi, _hip_step
    f1, j1 = expfit_problem(E2E["expfit_seed"])
    lo, hi = EXPFIT_BOX
    kw = dict(bounds=(lo, hi), method=method)

    def solve(ctx, fun, jac, x0):
        return bl.least_squares(fun, x0, jac, options={"ctx": ctx}, **kw)

    def solve_rosen(ctx):
        return bl.least_squares(rosen, [-2.0, 1.0], rosen_jac, method=method, options={"ctx": ctx})

    fresh = []
    for which in (0, 1):
        c = _abi.Context(0)
        fresh.append(solve(c, f1, j1, EXPFIT_X0) if which == 0 else solve_rosen(c))
        c.close()
    ctx = _abi.Context(0)
    r1 = solve(ctx, f1, j1, EXPFIT_X0)
    pool = ctx.__dict__["_solver_pool"]
    assert len(pool) == 1
    h1 = pool[0].h.value
    r2 = solve_rosen(ctx)                                                             # another shape: another plan
    assert len(pool) == 2
    with pytest.raises(ZeroDivisionError):                                            # a solve that dies half way
        calls = {"n": 0}

        def bad(x):
            calls["n"] += 1
            if calls["n"] > 2:
                raise ZeroDivisionError
            return f1(x)
        solve(ctx, bad, j1, EXPFIT_X0)
    assert len(pool) == 2
    r3 = solve(ctx, f1, j1, EXPFIT_X0)                                                # the first plan again
    assert len(pool) == 2 and any(s.h.value == h1 for s in pool)
    for a, b in ((r1, fresh[0]), (r3, fresh[0]), (r2, fresh[1])):
        assert np.array_equal(a.x, b.x) and a.nfev == b.nfev and a.status == b.status
        assert a.obj_value == b.obj_value and a.optimality == b.optimality
    keep = _hip_step.PLAN_POOL_KEEP
    rng = np.random.default_rng(3)
    for k in range(keep + 3):                                                         # more shapes than the pool keeps
        A = rng.standard_normal((6 + k, 3)); y = rng.standard_normal(6 + k)
        bl.least_squares(lambda x, A=A, y=y: A @ x - y, np.zeros(3), lambda x, A=A: A, method=method,
                         options={"ctx": ctx})
    assert len(pool) == keep
    held = list(pool)
    ctx.close()
    assert all(not s.h for s in held)
