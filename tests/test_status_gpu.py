"""Error statuses of the TRF step: where the reference raises ValueError out of
`intersect_trust_region` (trust_region.py:28-29 "`s` is zero.", :34-35 "`x` is not within the
trust region."), called from find_reflected_step (trf.py:128).

How the two conditions can be reached through the step path at all:
  * "`s` is zero": r_h is p_h with some signs flipped, so dot(r_h, r_h) == 0 needs an
    UNDERFLOW of the squares: residuals of size 1e-170 and a variable one denormal away from its
    bound (what make_strictly_feasible leaves behind at lb = 0) give a reflective step with
    |p_h| ~ 1e-170.  Deterministic; the oracle (= the reference's arithmetic) raises.
  * "`x` is not within the trust region": solve_lsq_trust_region returns ||p_h|| <= Delta up to
    one rounding (it rescales when phi > 0, trust_region.py:149-150), and the argument is
    p_h * p_stride with p_stride < 1, so c = ||p_h p_stride||^2 - Delta^2 > 0 only when p_stride is
    within an ulp or two of 1 AND the roundings fall the right way: a knife edge in the reference
    itself (shown on the oracle in tests/test_oracle_golden.py: a few per cent of a batch
    engineered to to_bound = 1 - 2^-53 raise).  The GPU test engineers such a batch for the GPU's
    own step (two passes), requires that the condition fires for some problems, that it only ever
    fires on that knife edge, and checks the host-side contract (B == 1 raises the reference's
    exception; batches raise naming the problem; the device driver freezes the problem).
"""
import numpy as np
import pytest

from oracle import blsq_oracle as orc
from _cases import zero_direction_problem, knife_edge_place

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bl():
    import bounded_lsq
    return bounded_lsq


def test_zero_direction_matches_reference_and_raises(bl):
    P = zero_direction_problem()
    with pytest.raises(ValueError, match="`s` is zero"):
        orc.trf_step_solve(P["J"], P["f"], P["x"], P["lb"], P["ub"], P["scale"], 1.0, 0.0)
    m, n = P["J"].shape
    sol = bl.TrfStepSolver(1, m, n)
    sol.factor(P["J"][None], P["f"][None], P["x"][None], P["lb"][None], P["ub"][None],
               P["scale"][None])
    with pytest.raises(ValueError, match="`s` is zero"):          # B == 1: drop-in behaviour
        sol.step(np.array([1.0]), np.array([0.0]))
    sol.close()
    # inside a batch: status[b] = BLSQ_STATUS_ZERO_DIRECTION for that problem only, the
    # neighbours are untouched and still match the oracle
    from bounded_lsq import _synth
    Q = _synth.trf_batch(40, 3, m, n)
    for k in ("J", "f", "x", "lb", "ub", "scale"):
        Q[k][1] = P[k]
    sol = bl.TrfStepSolver(3, m, n)
    sol.factor(Q["J"], Q["f"], Q["x"], Q["lb"], Q["ub"], Q["scale"])
    S = sol.step(np.array([10.0, 1.0, 0.5]), np.zeros(3))
    assert list(S.status) == [0, 1, 0]
    for b, D in ((0, 10.0), (2, 0.5)):
        _, So = orc.trf_step_solve(Q["J"][b], Q["f"][b], Q["x"][b], Q["lb"][b], Q["ub"][b],
                                   Q["scale"][b], D, 0.0)
        assert np.linalg.norm(S.step[b] - So.step) <= 1e-10 * np.linalg.norm(So.step)
        np.testing.assert_array_equal(S.hits[b], So.hits)
    sol.close()


def test_outside_trust_region_only_on_the_knife_edge(bl):
    """8192 problems engineered (from the GPU's OWN trust-region step, two passes) to end
    1 - 2^-53 of the way to a bound.  BLSQ_STATUS_OUTSIDE_TR may fire only there, and only where
    c = ||p_h to_bound||^2 - Delta^2 is rounding noise; where it fires, the same problem alone
    (B = 1) raises the reference's ValueError.  How many problems hit is a property of the build's
    summation orders (typically a few per thousand), so a build with no hit skips the second half
    — the status -> exception mapping itself is covered on the CPU (tests/test_frontend_cpu.py)
    and, for status 1, by test_zero_direction_matches_reference_and_raises."""
    from bounded_lsq import _synth
    B, m, n = 8192, 100, 64
    rng = np.random.default_rng(8192)
    P = dict(J=rng.standard_normal((B, m, n)), f=rng.standard_normal((B, m)),
             x=np.zeros((B, n)), scale=np.ones((B, n)))   # x = 0: see tests/_cases.py (resolution of to_bound)
    w = rng.uniform(5, 10, (B, n))
    P["lb"], P["ub"] = P["x"] - w, P["x"] + w
    sol = bl.TrfStepSolver(B, m, n)
    F = sol.factor(P["J"], P["f"], P["x"], P["lb"], P["ub"], P["scale"])
    # Delta = half the Gauss-Newton step length: pass 0 with a huge radius gives that length
    S0 = sol.step(np.full(B, 1e6), np.zeros(B))
    Delta = 0.5 * S0.step_h_norm
    sol.step(Delta, np.zeros(B))
    D0 = sol.fetch_step()
    v = np.where((F.g < 0), P["ub"] - P["x"], np.where(F.g > 0, P["x"] - P["lb"], 1.0))
    p = np.sqrt(v) * D0.p_h_tr                                    # p = d * p_h, scale = 1
    rows = np.arange(B)
    # the variable that will hit: moving AWAY from the bound that defines its v_j (so that the other
    # bound can be moved without changing the step), the largest such component
    cand = np.where(p * F.g > 0, np.abs(p), -1.0)
    jj = np.argmax(cand, axis=1)
    ok = (D0.branch == 0) & (cand[rows, jj] > 0) & (S0.branch == 0)
    assert ok.sum() > B // 2
    Q = knife_edge_place(dict(P, j=np.where(ok, jj, 0), Delta=Delta), np.where(ok[:, None], p, 1.0))
    for k in ("lb", "ub"):                                        # problems not selected stay as they were
        Q[k][~ok] = P[k][~ok]
    sol.factor(Q["J"], Q["f"], Q["x"], Q["lb"], Q["ub"], Q["scale"])
    S = sol.step(Delta, np.zeros(B))
    D = sol.fetch_step()
    sol.close()
    np.testing.assert_array_equal(D.p_h_tr[ok], D0.p_h_tr[ok])     # same step, now one ulp short
    assert set(np.unique(S.status)) <= {0, 2}
    hit = np.flatnonzero(S.status == 2)
    refl = np.flatnonzero(S.branch == 1)
    assert np.all(ok[refl]) and refl.size > ok.sum() // 2
    print("knife edge: %d of %d reflective, OUTSIDE_TR fired for %d" % (refl.size, B, hit.size))
    eps = np.finfo(float).eps
    for b in hit:                              # it fires only where the reference's test is noise
        assert D.branch[b] == 1 and 1.0 - 4 * eps <= D.to_bound[b] < 1.0
        xx = D.p_h_tr[b] * D.to_bound[b]
        c = np.dot(xx, xx) - Delta[b] ** 2
        assert abs(c) <= 8 * eps * Delta[b] ** 2
    if hit.size == 0:
        pytest.skip("no knife-edge problem of this batch rounds to c > 0 with this build")
    # host contract: alone (B == 1) the same problem raises; results do not depend on the batch
    b = int(hit[0])
    sol = bl.TrfStepSolver(1, m, n)
    sol.factor(Q["J"][b][None], Q["f"][b][None], Q["x"][b][None], Q["lb"][b][None],
               Q["ub"][b][None], Q["scale"][b][None])
    with pytest.raises(ValueError, match="not within the trust region"):
        sol.step(Delta[b:b + 1], np.zeros(1))
    sol.close()


@pytest.mark.parametrize("driver", ["host", "device"])
def test_batched_drivers_abort_like_the_reference(bl, driver, monkeypatch):
    """least_squares_batch: a problem whose step reports a status makes the solve raise
    ValueError (the reference aborts there), naming the problem — on both drivers.

    Through the public front end the underflow case cannot be reached (|g| ~ 1e-170 is below any
    admissible gtol, so the reference — and the batch — stop with status 1 before the step);
    the test lifts that guard (gtol = 0, which the raw blsq_outer_start accepts) to drive the
    status through both drivers' plumbing."""
    P = zero_direction_problem()
    m, n = P["J"].shape
    from bounded_lsq import _synth, _batch
    Q = _synth.trf_batch(40, 3, m, n)
    for k in ("J", "f", "x", "lb", "ub"):
        Q[k][1] = P[k]
    # linear residuals f(x) = f0 + J (x - x0): the first step-solve sees exactly (J, f0)
    X0 = Q["x"].copy()

    def fun(X):
        return Q["f"] + np.einsum("bij,bj->bi", Q["J"], X - X0)

    def jac(X):
        return Q["J"].copy()

    # as shipped: problem 1 terminates on gtol at once, nothing raises (reference behaviour)
    res = bl.least_squares_batch(fun, X0, jac, bounds=(Q["lb"], Q["ub"]), method="trf",
                                 driver=driver)
    assert res[1].status == 1 and res[1].nfev == 1
    # dogbox never calls intersect_trust_region
    res = bl.least_squares_batch(fun, X0, jac, bounds=(Q["lb"], Q["ub"]), method="dogbox",
                                 driver=driver)
    assert len(res) == 3
    clamp = _batch._clamp_tolerances
    monkeypatch.setattr(_batch, "_clamp_tolerances", lambda f, x, g: clamp(f, x, 1.0)[:2] + (0.0,))
    with pytest.raises(ValueError, match=r"problem 1: `s` is zero"):
        bl.least_squares_batch(fun, X0, jac, bounds=(Q["lb"], Q["ub"]), method="trf",
                               driver=driver)
