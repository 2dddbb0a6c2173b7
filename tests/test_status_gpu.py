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
from _cases import zero_direction_problem, knife_edge_base, knife_edge_place

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
    Q0 = knife_edge_base(B=512, m=100, n=64)
    B, m, n = Q0["J"].shape
    sol = bl.TrfStepSolver(B, m, n)
    # pass 1, wide bounds: the GPU's OWN trust-region step (the bound is then placed for it: the
    # oracle's step differs in the last digits, which would smear to_bound over +-1e-15 around 1)
    sol.factor(Q0["J"], Q0["f"], Q0["x"], Q0["lb"], Q0["ub"], Q0["scale"])
    sol.step(Q0["Delta"], np.zeros(B))
    D0 = sol.fetch_step()
    assert np.all(D0.branch == 0)
    Q = knife_edge_place(Q0, Q0["d"] * D0.p_h_tr)
    # pass 2: same J, f, x and the same bound that defines v_j -> the same step, now one ulp short
    sol.factor(Q["J"], Q["f"], Q["x"], Q["lb"], Q["ub"], Q["scale"])
    S = sol.step(Q["Delta"], np.zeros(B))
    D = sol.fetch_step()
    sol.close()
    np.testing.assert_array_equal(D.p_h_tr, D0.p_h_tr)
    assert set(np.unique(S.status)) <= {0, 2}
    hit = np.flatnonzero(S.status == 2)
    refl = np.flatnonzero(S.branch == 1)
    assert refl.size > B // 8, "the engineered batch must reach the reflective branch"
    print("knife edge: %d of %d reflective, OUTSIDE_TR fired for %d" % (refl.size, B, hit.size))
    assert hit.size > 0, "BLSQ_STATUS_OUTSIDE_TR never fired on %d knife-edge cases" % refl.size
    eps = np.finfo(float).eps
    for b in hit:                              # it fires only where the reference's test is noise
        assert D.branch[b] == 1 and D.to_bound[b] < 1.0
        xx = D.p_h_tr[b] * D.to_bound[b]
        c = np.dot(xx, xx) - Q["Delta"][b] ** 2
        assert abs(c) <= 8 * eps * Q["Delta"][b] ** 2
    # (the reference arithmetic on bounds placed for ITS OWN step raises for a few per cent of such
    # a batch too: tests/test_oracle_golden.py::test_outside_trust_region_is_a_knife_edge_in_the_
    # reference; problem by problem the two cannot agree, the condition being rounding noise)
    # host contract: alone (B == 1) the same problem raises; results do not depend on the batch
    b = int(hit[0])
    sol = bl.TrfStepSolver(1, m, n)
    sol.factor(Q["J"][b][None], Q["f"][b][None], Q["x"][b][None], Q["lb"][b][None],
               Q["ub"][b][None], Q["scale"][b][None])
    with pytest.raises(ValueError, match="not within the trust region"):
        sol.step(Q["Delta"][b:b + 1], np.zeros(1))
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
