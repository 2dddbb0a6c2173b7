"""CSNE tier (csrc/csne_kernels.hip): problems the conditioning certificate keeps off the normal-equations path keep
their Gram-Cholesky factor as a PRECONDITIONER; the cheap Newton iteration on phi(alpha) records its evaluations, ONE
streaming pass over J corrects every recorded solve, the scalar iteration is replayed on the corrected phi / phi'
(trust_region.py:111-150) and the final step corrected once more.  Whatever the tier: step within 1e-10 of the oracle
(the reference's gesdd on the augmented matrix, trf.py:264-274), masks, branch and iteration counts identical."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
RTOL = 1e-10


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / np.linalg.norm(b)


def logspaced(rng, B, m, n, kappa):
    J = np.empty((B, m, n))
    for b in range(B):
        U, _ = np.linalg.qr(rng.standard_normal((m, n)))
        V, _ = np.linalg.qr(rng.standard_normal((n, n)))
        J[b] = (U * np.logspace(0, -np.log10(kappa), n)) @ V.T * np.sqrt(m)
    return J


def run_trf(P, Delta, alpha0=None):
    import bounded_lsq as bl
    from bounded_lsq import _abi
    B, m, n = P["J"].shape
    ctx = _abi.Context(0)
    sol = bl.TrfStepSolver(B, m, n, ctx=ctx)
    ctx.gram_stats(reset=True); ctx.cqr2_stats(reset=True); ctx.csne_stats(reset=True)
    sol.factor(P["J"], P["f"], P["x"], P["lb"], P["ub"], P["scale"])
    S = sol.step(Delta, np.zeros(B) if alpha0 is None else alpha0)
    stats = {"gram": ctx.gram_stats(), "cqr2": ctx.cqr2_stats(), "csne": ctx.csne_stats()}
    sol.close(); ctx.close()
    return stats, S


def check(P, Delta, S, skip=()):
    from oracle import blsq_oracle as orc
    worst = 0.0
    for b in range(P["J"].shape[0]):
        if b in skip:
            continue
        _, So = orc.trf_step_solve(P["J"][b], P["f"][b], P["x"][b], P["lb"][b], P["ub"][b], P["scale"][b],
                                   Delta[b], 0.0)
        e = rel(S.step[b], So.step)
        worst = max(worst, e)
        assert e < RTOL, (b, e)
        np.testing.assert_array_equal(S.hits[b], So.hits)
        assert int(S.n_iter[b]) == So.n_iter and int(S.branch[b]) == So.branch, (b, S.n_iter[b], So.n_iter)
        assert abs(S.predicted_reduction[b] - So.predicted_reduction) <= 1e-9 * abs(So.predicted_reduction), b
        assert abs(S.alpha[b] - So.alpha) <= 1e-9 * abs(So.alpha), (b, S.alpha[b], So.alpha)
    return worst


@pytest.mark.parametrize("m,n,kappa", [(4096, 256, 3e3), (1500, 200, 1e4), (3000, 100, 3e4), (700, 129, 1e3),
                                       (2000, 80, 5e3), (4096, 255, 3e4)])
def test_unbounded_ill_conditioned_problems_take_the_tier(m, n, kappa):
    from bounded_lsq import _synth
    rng = np.random.default_rng(int(kappa) % 1000 + n)
    B = 4
    P = _synth.trf_batch(77, B, m, n, unbounded=True)
    P["J"] = logspaced(rng, B, m, n, kappa)
    Delta = np.array([10.0, 0.5, 0.05, 1e6])                 # (the last one: the Gauss-Newton step itself)
    stats, S = run_trf(P, Delta)
    print(stats, S.n_iter)
    assert stats["gram"] == (0, B)
    assert stats["csne"] == (B, B, 0) and stats["cqr2"] == 0, stats
    w = check(P, Delta, S)
    print("worst step error", w)


def test_problems_beyond_the_tiers_measured_bound_go_on_to_the_next_tier():
    """kappa(J) = 3e5 .. 1e6: the certificate's bound is above CSNE_K2_MAX or the measured first-order correction above
    CSNE_ETA_MAX — the problem is declined (at factor or at step time) and CholeskyQR2 / the tree deliver the step."""
    from bounded_lsq import _synth
    rng = np.random.default_rng(12)
    B, m, n = 3, 2048, 192
    P = _synth.trf_batch(78, B, m, n, unbounded=True)
    P["J"][0] = logspaced(rng, 1, m, n, 1.5e5)[0]
    P["J"][1] = logspaced(rng, 1, m, n, 1e6)[0]
    P["J"][2] = logspaced(rng, 1, m, n, 1e9)[0]
    Delta = np.array([10.0, 0.5, 2.0])
    stats, S = run_trf(P, Delta)
    routed, delivered, declined = stats["csne"]
    assert stats["gram"] == (0, B) and delivered == 0 and routed == declined, stats
    check(P, Delta, S)
