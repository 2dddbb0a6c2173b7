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


@pytest.mark.parametrize("mfma", [1, 0])
@pytest.mark.parametrize("m,n", [(1000, 81), (1237, 130), (531, 193), (2050, 250), (4096, 256)])
def test_both_pass_kernels_on_ragged_shapes(m, n, mfma, blsq_opt):
    """The pass over J exists twice — the MFMA kernel (16-row tiles, four column slices) and the vector-ALU kernel
    (option csne_mfma = 0) — and both handle a last tile of fewer than sixteen rows, column counts that are odd or no
    multiple of the slice width, and chunks of unequal length.  The batch mixes depths of the recording."""
    from bounded_lsq import _synth
    blsq_opt("csne_mfma", mfma)
    rng = np.random.default_rng(m + n)
    B = 3
    P = _synth.trf_batch(n, B, m, n, unbounded=True)
    P["J"] = logspaced(rng, B, m, n, 4e3)
    Delta = np.array([0.02, 1.0, 1e6])
    stats, S = run_trf(P, Delta)
    assert stats["csne"] == (B, B, 0) and stats["cqr2"] == 0, stats
    w = check(P, Delta, S)
    print("mfma", mfma, "worst step error", w, "iterations", S.n_iter)
    assert len(set(int(k) for k in S.n_iter)) > 1             # (different depths in one launch)


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


def test_bounded_problems_on_the_tier_take_the_reflective_branch():
    """Rejected problems WITH bounds (upper bounds only, so that the Coleman-Li block leaves the variables with g > 0
    unregularised): a large radius sends x + p out of the box — find_reflected_step / find_gradient_step
    (trf.py:105-170) and the model comparison run on the corrected p, every product with p taken from the normal
    equations (H p = -(c g_h + alpha p)) instead of the factor.  hits, branch, choice and the step as the oracle's."""
    import bounded_lsq as bl
    from bounded_lsq import _abi, _synth
    from oracle import blsq_oracle as orc
    rng = np.random.default_rng(31)
    B, m, n = 12, 1536, 144
    P = _synth.trf_batch(91, B, m, n)
    P["J"] = logspaced(rng, B, m, n, 2e4)
    P["lb"][:] = -np.inf
    P["ub"] = P["x"] + rng.uniform(0.5, 3.0, (B, n))
    Delta = np.where(np.arange(B) % 3 == 0, 300.0, np.where(np.arange(B) % 3 == 1, 30.0, 3.0))
    ctx = _abi.Context(0)
    sol = bl.TrfStepSolver(B, m, n, ctx=ctx)
    ctx.csne_stats(reset=True)
    sol.factor(P["J"], P["f"], P["x"], P["lb"], P["ub"], P["scale"])
    S = sol.step(Delta, np.zeros(B))
    on, eta = sol.debug_csne()
    D = sol.fetch_step()
    sol.close(); ctx.close()
    refl = 0
    for b in range(B):
        _, So = orc.trf_step_solve(P["J"][b], P["f"][b], P["x"][b], P["lb"][b], P["ub"][b], P["scale"][b],
                                   Delta[b], 0.0)
        assert rel(S.step[b], So.step) < RTOL, (b, on[b], rel(S.step[b], So.step))
        np.testing.assert_array_equal(S.hits[b], So.hits)
        assert int(S.n_iter[b]) == So.n_iter and int(S.branch[b]) == So.branch
        assert int(D.choice[b]) == So.choice
        assert abs(S.predicted_reduction[b] - So.predicted_reduction) <= 1e-9 * abs(So.predicted_reduction), b
        refl += int(on[b] == 1 and So.branch == 1)
    print("bounded: on the tier", int(on.sum()), "of", B, "- reflective among them", refl, "- largest eta", eta.max())
    assert on.sum() >= B // 2 and refl >= 3, (on, refl)


@pytest.mark.parametrize("noise", [1e-9, 1e-3, 1.0])
def test_small_and_large_residual_problems(noise):
    """f = -J x* + noise: from a nearly consistent system (||J p + f|| << ||f||: the pass forms J^T (J p + f) with f
    INSIDE the product, so the small residual is not lost against g = J^T f) to the large-residual one."""
    from bounded_lsq import _synth
    rng = np.random.default_rng(5)
    B, m, n = 4, 2048, 128
    P = _synth.trf_batch(63, B, m, n, unbounded=True)
    P["J"] = logspaced(rng, B, m, n, 1e4)
    xs = rng.standard_normal((B, n))
    P["f"] = -np.einsum("bmn,bn->bm", P["J"], xs) + noise * rng.standard_normal((B, m))
    pg = [np.linalg.norm(np.linalg.lstsq(P["J"][b], -P["f"][b], rcond=None)[0]) for b in range(B)]
    Delta = np.array([2.0 * pg[0], 0.7 * pg[1], 0.1 * pg[2], 1e-3 * pg[3]])
    stats, S = run_trf(P, Delta)
    assert stats["csne"][0] == B and stats["csne"][2] == 0, stats
    check(P, Delta, S)


def test_a_problems_bits_do_not_depend_on_its_batch():
    """The tier is chosen per problem by the problem's own numbers, the pass sums a problem's rows in chunks that are a
    function of m alone and the depth of the recording (the NE of the pass kernel) only adds zero vectors: a problem
    alone, in another order, or beside well-conditioned and deeper-iterating neighbours gives the same bits."""
    from bounded_lsq import _synth
    rng = np.random.default_rng(9)
    B, m, n = 8, 2048, 160
    P = _synth.trf_batch(5, B, m, n)
    for b, kap in ((1, 2e3), (2, 3e4), (5, 8e3), (6, 1e6)):
        P["J"][b] = logspaced(rng, 1, m, n, kap)[0]
        P["lb"][b] = -np.inf; P["ub"][b] = np.inf
    Delta = np.array([10.0, 0.5, 0.05, 2.0, 1.0, 1e4, 0.3, 0.7])
    stats, S = run_trf(P, Delta)
    assert stats["gram"] == (4, 4) and stats["csne"][0] == 3 and stats["csne"][2] == 0, stats
    check(P, Delta, S)
    order = np.array([5, 2, 1])
    Q = {k: v[order].copy() for k, v in P.items()}
    stats2, S2 = run_trf(Q, Delta[order])
    assert stats2["csne"] == (3, 3, 0), stats2
    for i, b in enumerate(order):
        assert np.array_equal(S2.step[i], S.step[b]) and S2.alpha[i] == S.alpha[b]
    one = {k: v[2:3].copy() for k, v in P.items()}
    _, S1 = run_trf(one, Delta[2:3])
    assert np.array_equal(S1.step[0], S.step[2])


def test_inner_iterations_carry_alpha_and_need_no_new_factorisation():
    """trf.py:283-331: the inner loop calls solve_lsq_trust_region again with a smaller Delta and the carried alpha
    (initial_alpha inside the bracket: no restart) — step calls on the SAME factor; the pass reads the caller's J again."""
    import bounded_lsq as bl
    from bounded_lsq import _abi, _synth
    from oracle import blsq_oracle as orc
    rng = np.random.default_rng(77)
    B, m, n = 3, 3000, 200
    P = _synth.trf_batch(13, B, m, n, unbounded=True)
    P["J"] = logspaced(rng, B, m, n, 6e3)
    ctx = _abi.Context(0)
    sol = bl.TrfStepSolver(B, m, n, ctx=ctx)
    ctx.csne_stats(reset=True)
    sol.factor(P["J"], P["f"], P["x"], P["lb"], P["ub"], P["scale"])
    Fo = [orc.trf_factor(P["J"][b], P["f"][b], P["x"][b], P["lb"][b], P["ub"][b], P["scale"][b]) for b in range(B)]
    Delta = np.array([5.0, 1.0, 0.2])
    alpha = np.zeros(B)
    for it in range(4):
        S = sol.step(Delta, alpha)
        for b in range(B):
            So = orc.trf_step(Fo[b], float(Delta[b]), float(alpha[b]))
            assert rel(S.step[b], So.step) < RTOL, (it, b, rel(S.step[b], So.step))
            assert int(S.n_iter[b]) == So.n_iter
            assert abs(S.alpha[b] - So.alpha) <= 1e-9 * abs(So.alpha)
        # the rejected-step update of trf.py:326-328
        Dn = 0.25 * S.step_h_norm
        alpha = S.alpha * Delta / Dn
        Delta = Dn
    assert ctx.csne_stats() == (B, 4 * B, 0)
    sol.close(); ctx.close()


@pytest.mark.parametrize("m,n,kappa,bounded", [(1800, 144, 2e4, False), (4096, 256, 3e3, False), (1500, 100, 5e4, True),
                                               (900, 81, 1e3, True)])
def test_dogbox_newton_step_is_corrected_at_factor_time(m, n, kappa, bounded):
    """dogbox on the tier: lstsq(J_free, -f) (dogbox.py:197) of a rejected free block is ONE solve — the cheap one
    corrected against J at factor time; the dogleg's predicted reduction (dogbox.py:208-209) from the normal equations
    the corrected step satisfies.  Newton / Cauchy steps, step, masks, tr_hit and predicted reduction as the oracle's."""
    import bounded_lsq as bl
    from bounded_lsq import _synth, _abi
    from oracle import blsq_oracle as orc
    rng = np.random.default_rng(21 + n)
    B = 4
    P = _synth.dogbox_batch(31, B, m, n)
    P["J"] = logspaced(rng, B, m, n, kappa)
    if not bounded:
        P["lb"][:] = -np.inf; P["ub"][:] = np.inf; P["on_bound"][:] = 0
    else:                                                       # wide bounds, a tenth of the variables sitting on one
        P["lb"] = P["x"] - rng.uniform(50.0, 500.0, (B, n)); P["ub"] = P["x"] + rng.uniform(50.0, 500.0, (B, n))
        P["on_bound"][:] = 0
        for b in range(B):
            idx = rng.choice(n, n // 10, replace=False)
            P["x"][b, idx] = P["lb"][b, idx]; P["on_bound"][b, idx] = -1
    ctx = _abi.Context(0)
    sol = bl.DogboxStepSolver(B, m, n, ctx=ctx)
    ctx.gram_stats(reset=True); ctx.cqr2_stats(reset=True); ctx.csne_stats(reset=True)
    sol.factor(P["J"], P["f"], P["x"], P["lb"], P["ub"], P["scale"], P["on_bound"])
    stats = (ctx.gram_stats(), ctx.cqr2_stats(), ctx.csne_stats())
    F = sol.fetch_factor(want_steps=True)
    Delta = np.array([0.02, 1.0, 50.0, 1e6])
    S = sol.step(Delta)
    sol.close(); ctx.close()
    print(stats)
    assert stats[0] == (0, B) and stats[1] == 0 and stats[2] == (B, B, 0), stats
    for b in range(B):
        Fo, So = orc.dogbox_step_solve(P["J"][b], P["f"][b], P["x"][b], P["lb"][b], P["ub"][b], P["scale"][b],
                                       P["on_bound"][b], float(Delta[b]))
        assert rel(S.step[b], So.step) < RTOL, (b, rel(S.step[b], So.step))
        np.testing.assert_array_equal(S.on_bound_new[b], So.on_bound_new)
        assert bool(S.tr_hit[b]) == bool(So.tr_hit)
        assert abs(S.predicted_reduction[b] - So.predicted_reduction) <= 1e-9 * abs(So.predicted_reduction), b


def _illcond_fit(B, m, n, kappa, seed):
    """B nonlinear fits  r(x) = tanh(A_b x - y_b)  with ill-conditioned A_b (kappa(A) = kappa): the Jacobian
    diag(1 - r^2) A_b inherits the conditioning, so the certificate rejects it at every outer iteration."""
    rng = np.random.default_rng(seed)
    A = logspaced(rng, B, m, n, kappa) / np.sqrt(m)
    xs = rng.standard_normal((B, n))
    Y = np.einsum("bmn,bn->bm", A, xs) + 0.05 * rng.standard_normal((B, m))

    def fun(X):
        X = np.atleast_2d(X)
        return np.tanh(np.einsum("bmn,bn->bm", A[:X.shape[0]], X) - Y[:X.shape[0]])

    def jac(X):
        X = np.atleast_2d(X)
        r = np.tanh(np.einsum("bmn,bn->bm", A[:X.shape[0]], X) - Y[:X.shape[0]])
        return (1.0 - r * r)[:, :, None] * A[:X.shape[0]]
    return fun, jac, A, Y, xs


@pytest.mark.parametrize("method", ["trf", "dogbox"])
def test_end_to_end_solves_on_the_tier_sequential_batched_and_device_resident(method):
    """Whole solves whose every outer iteration is on the tier: the sequential host driver (inner iterations re-solve on
    the same factor with the carried alpha: the pass reads the staged J again), the batched host driver, and the
    device-resident outer driver (masked re-factorisation: problems leave and re-enter the tier as their Jacobians are
    refreshed while the others keep their flags, recordings and J) — the same nfev / njev / status and the same x."""
    from bounded_lsq import least_squares, least_squares_batch, _abi, _hip_step
    B, m, n = 5, 600, 96
    fun, jac, A, Y, xs = _illcond_fit(B, m, n, 2e3, 17)
    X0 = xs + 0.3 * np.random.default_rng(3).standard_normal((B, n))
    kw = dict(method=method, max_nfev=25)
    ctx = _hip_step.default_context()
    ctx.csne_stats(reset=True)
    seq = []
    for b in range(B):
        def fun_b(p, b=b):
            return np.tanh(A[b] @ p - Y[b])

        def jac_b(p, b=b):
            r = np.tanh(A[b] @ p - Y[b])
            return (1.0 - r * r)[:, None] * A[b]
        seq.append(least_squares(fun_b, X0[b], jac_b, **kw))
    routed, delivered, declined = ctx.csne_stats(reset=True)
    assert routed >= sum(r.njev for r in seq) - B and delivered > 0, (routed, delivered, declined)
    host = least_squares_batch(fun, X0, jac, **kw)
    dev = least_squares_batch(fun, X0, jac, driver='device', **kw)
    r2 = ctx.csne_stats(reset=True)
    assert r2[0] > 0 and r2[1] > 0, r2
    assert len({r.nfev for r in seq}) > 1                   # (different iteration counts: masked factor calls happen)
    for b in range(B):
        for name, res in (("host", host[b]), ("device", dev[b])):
            assert (res.nfev, res.njev, res.status) == (seq[b].nfev, seq[b].njev, seq[b].status), (name, b)
            np.testing.assert_allclose(res.x, seq[b].x, rtol=1e-8, atol=1e-10)
            np.testing.assert_allclose(res.obj_value, seq[b].obj_value, rtol=1e-9)


def test_alternating_conditioning_classes_on_one_plan(blsq_opt):
    """Two input sets alternate on ONE plan through the device API — a well-conditioned bounded batch and one whose
    problems are rejected (CSNE tier) — so the optimistic guess of blsq_trf_factor_dev fails again and again: the plan
    backs off (2, 4, ... factor calls without guessing after each wrong guess).  Whatever it guesses and however often:
    the bits of the synchronous mode at every call, and the same path statistics."""
    import bounded_lsq as bl
    from bounded_lsq import _synth, _abi
    rng = np.random.default_rng(41)
    B, m, n = 6, 1024, 128
    good = _synth.trf_batch(900, B, m, n)
    bad = _synth.trf_batch(901, B, m, n, unbounded=True)
    bad["J"] = logspaced(rng, B, m, n, 5e3)
    Delta = np.array([0.5, 0.05, 5.0, 0.2, 0.3, 1.0])
    runs = []
    for opt in (1, 0):
        blsq_opt("optimistic", opt)
        ctx = _abi.Context(0)
        sol = bl.TrfStepSolver(B, m, n, ctx=ctx)
        dev = [{k: ctx.to_device(P[k]) for k in ("J", "f", "x", "lb", "ub", "scale")} for P in (good, bad)]
        dD, dA = ctx.to_device(Delta), ctx.to_device(np.zeros(B))
        ctx.gram_stats(reset=True); ctx.csne_stats(reset=True)
        outs = []
        for call in range(14):
            d = dev[call & 1]
            sol.factor_dev(d["J"], d["f"], d["x"], d["lb"], d["ub"], d["scale"])
            sol.step_dev(dD, dA)
            S = sol.fetch_step()
            outs.append((S.step.copy(), np.asarray(S.alpha).copy(), np.asarray(S.n_iter).copy(), S.hits.copy()))
        stats = (ctx.gram_stats(), ctx.csne_stats())
        assert stats == ((7 * B, 7 * B), (7 * B, 7 * B, 0)), (opt, stats)
        runs.append(outs)
        sol.close()
        for d in dev:
            for v in d.values():
                ctx.free(v)
        ctx.free(dD); ctx.free(dA)
        ctx.close()
    for call, (a, b) in enumerate(zip(*runs)):
        for x1, x0 in zip(a, b):
            assert np.array_equal(x1, x0), call


@pytest.mark.parametrize("scale_mode", [1, 2])
def test_jac_scaling_on_the_tier(scale_mode):
    """scaling='jac' (trf.py:216-219, 239-242): scale = 1 / ||J_j|| (mode 1) resp. min(scale, 1 / ||J_j||) (mode 2) from
    the Gram's diagonal; the tier's pass multiplies J by d = sqrt(v) scale (J_h = J D).  Badly scaled columns on top of
    an ill-conditioned J."""
    import bounded_lsq as bl
    from bounded_lsq import _abi, _synth
    from oracle import blsq_oracle as orc
    rng = np.random.default_rng(8)
    B, m, n = 4, 1200, 110
    P = _synth.trf_batch(19, B, m, n, unbounded=True)
    P["J"] = logspaced(rng, B, m, n, 4e3) * 10.0 ** rng.uniform(-2, 2, size=(B, 1, n))
    scale0 = np.full((B, n), 1.0) if scale_mode == 1 else rng.uniform(0.5, 2.0, (B, n)) / np.linalg.norm(P["J"], axis=1)
    Delta = np.array([10.0, 0.5, 0.05, 1e9])
    ctx = _abi.Context(0)
    sol = bl.TrfStepSolver(B, m, n, ctx=ctx)
    ctx.csne_stats(reset=True)
    F = sol.factor(P["J"], P["f"], P["x"], P["lb"], P["ub"], scale0.copy(), scale_mode=scale_mode)
    scale = F.scale
    S = sol.step(Delta, np.zeros(B))
    stats = ctx.csne_stats()
    sol.close(); ctx.close()
    assert stats[0] >= B - 1 and stats[2] == 0, stats          # (column scaling changes nothing for the equilibrated system)
    for b in range(B):
        jn = np.linalg.norm(P["J"][b], axis=0)
        sc = 1.0 / jn if scale_mode == 1 else np.minimum(scale0[b], 1.0 / jn)
        _, So = orc.trf_step_solve(P["J"][b], P["f"][b], P["x"][b], P["lb"][b], P["ub"][b], sc, Delta[b], 0.0)
        assert rel(S.step[b], So.step) < RTOL, (b, rel(S.step[b], So.step))
        assert int(S.n_iter[b]) == So.n_iter
        np.testing.assert_allclose(scale[b], sc, rtol=1e-12)
