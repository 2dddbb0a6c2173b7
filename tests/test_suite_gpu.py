"""Convergence regression on the 12-problem suite: the public front end on the
GPU step path must reproduce what the REFERENCE's drivers returned for the same functions, start
points and bounds (tests/golden/suite.json, captured by tests/golden/make_golden.py) — iteration
counts, termination status, solution, objective and active mask.  10 of the 96 records are
unstable in the reference itself (its counts change under a one-ulp move of the start point; the
fixture stores those neighbour runs) and are compared through properties."""
import numpy as np
import pytest

from _golden import load_json
from _suite import SUITE_BY_NAME

pytestmark = pytest.mark.gpu

SUITE_REC = load_json("suite.json")
TOL = float.fromhex(SUITE_REC["tol"])


def unhex(v):
    return np.array([float.fromhex(s) for s in v])


@pytest.mark.parametrize("rec", SUITE_REC["records"],
                         ids=["%s-box%d-%s-%s" % (r["problem"], r["box"], r["method"], r["scaling"])
                              for r in SUITE_REC["records"]])
def test_suite_record(rec):
    import bounded_lsq
    prob = SUITE_BY_NAME[rec["problem"]]
    lb, ub = prob["boxes"][rec["box"]]
    res = bounded_lsq.least_squares(prob["fun"], prob["x0"].copy(), jac=prob["jac"],
                                    bounds=(np.array(lb, float), np.array(ub, float)),
                                    method=rec["method"], ftol=TOL, xtol=TOL, gtol=TOL,
                                    scaling=rec["scaling"])
    if rec["stable"]:
        assert (res.nfev, res.njev, res.status) == (rec["nfev"], rec["njev"], rec["status"])
        np.testing.assert_allclose(res.x, unhex(rec["x"]), rtol=1e-7, atol=1e-10)
        np.testing.assert_allclose(res.obj_value, float.fromhex(rec["obj_value"]), rtol=1e-7,
                                   atol=1e-14)
        np.testing.assert_array_equal(res.active_mask, rec["active_mask"])
        return
    # The REFERENCE's own counts for this record change when the start point moves by one ulp
    # (rec["neighbours"], captured by make_golden.py): the path is decided by rounding noise, so
    # no independent factorisation can be asked to retrace it.  Properties instead: the run
    # terminates successfully and reaches an objective no worse than the reference's runs.
    assert res.status > 0 and res.success
    objs = [float.fromhex(rec["obj_value"])] + [float.fromhex(q["obj_value"])
                                                 for q in rec["neighbours"]]
    assert res.obj_value <= max(objs) * (1 + 1e-6) + 1e-12
    lb = np.array(lb, float); ub = np.array(ub, float)
    assert np.all(res.x >= lb) and np.all(res.x <= ub)


def test_suite_batched_device_driver_from_many_starts():
    """The device-resident batched driver on a suite problem from 16 perturbed start points:
    every problem reproduces the sequential front end (itself pinned to the reference above)."""
    import bounded_lsq
    prob = SUITE_BY_NAME["kowalik_osborne"]
    lb, ub = (np.array(v, float) for v in prob["boxes"][1])
    rng = np.random.default_rng(11)
    B = 16
    X0 = np.clip(prob["x0"] * (1 + 0.05 * rng.standard_normal((B, prob["x0"].size))), lb, ub)

    def fun(X):
        return np.stack([prob["fun"](x) for x in X])

    def jac(X):
        return np.stack([prob["jac"](x) for x in X])
    for method in ("trf", "dogbox"):
        res = bounded_lsq.least_squares_batch(fun, X0, jac, bounds=(lb, ub), method=method,
                                              driver="device")
        for b in range(B):
            ref = bounded_lsq.least_squares(prob["fun"], X0[b], jac=prob["jac"], bounds=(lb, ub),
                                            method=method)
            assert (res[b].nfev, res[b].njev, res[b].status) == (ref.nfev, ref.njev, ref.status)
            np.testing.assert_allclose(res[b].x, ref.x, rtol=1e-8, atol=1e-12)
