"""BASELINE config 5 at its per-rank size: one tall problem, 250 000 x 128 rows per rank
(2 000 000 x 128 over 8 GPUs).  A one-GPU box cannot host two RCCL ranks, so the two
row blocks are factored one after the other on the same device and exchanged by the test (the
Householder route: blsq_tsqr_local_dev / blsq_tsqr_combine_dev), and the Gram route runs as ONE
rank on the stacked 500 000 rows through blsq_tsqr_factor_dev; both must match the reference's
own path (oracle: scipy gesdd on the whole 500 128 x 128 augmented matrix)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
RTOL = 1e-10


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / np.linalg.norm(b)


@pytest.fixture(scope="module")
def tall():
    from oracle import blsq_oracle as orc
    rng = np.random.default_rng(2_000_000)
    m, n = 500_000, 128
    J = rng.standard_normal((m, n))
    f = rng.standard_normal(m)
    x = rng.uniform(-1.0, 1.0, n)
    lb = x - rng.uniform(1e-3, 0.05, n)
    ub = x + rng.uniform(1e-3, 0.05, n)
    P = dict(J=J, f=f, x=x, lb=lb, ub=ub, scale=np.ones(n))
    F = orc.trf_factor(J, f, x, lb, ub, P["scale"])           # ~10-20 s of LAPACK on the host
    ref = {D: orc.trf_step(F, D, 0.0) for D in (0.5, 10.0)}
    return P, ref


def test_config5_row_blocks_of_250000_rows(tall):
    import bounded_lsq as bl
    from bounded_lsq import _abi
    from bounded_lsq._multi import TsqrTrfSolver, row_block, tri_ld
    P, ref = tall
    m, n = P["J"].shape
    nranks = 2
    ctx = _abi.Context(0)
    ld = tri_ld(n)
    dstack = ctx.malloc(8 * nranks * ld * ld)
    dvec = {k: ctx.to_device(P[k]) for k in ("x", "lb", "ub", "scale")}
    sols = []
    for r in range(nranks):
        lo, hi = row_block(m, nranks, r)
        assert hi - lo == 250_000
        sol = TsqrTrfSolver(hi - lo, n, nranks, r, ctx=ctx, m_total=m)
        dJ = ctx.to_device(P["J"][lo:hi]); df = ctx.to_device(P["f"][lo:hi])
        sol.local_triangle_dev(dJ, df, _abi.vp(dstack.value + 8 * r * ld * ld))
        ctx.sync()
        ctx.free(dJ); ctx.free(df)
        sols.append(sol)
    for sol in sols:                              # every rank reaches the same step
        sol.combine_dev(dstack, dvec["x"], dvec["lb"], dvec["ub"], dvec["scale"])
        for Delta, So in ref.items():
            S = sol.step(np.array([Delta]), np.array([0.0]))
            assert rel(S.step[0], So.step) < RTOL, (Delta, rel(S.step[0], So.step))
            np.testing.assert_array_equal(S.hits[0], So.hits)
            assert int(S.n_iter[0]) == So.n_iter and int(S.branch[0]) == So.branch
        sol.close()
    ctx.close()


@pytest.mark.parametrize("gram", ["1", "0"])
def test_config5_single_rank_collective_route(tall, gram, monkeypatch, blsq_opt):
    """The same 500 000 rows as ONE rank through blsq_tsqr_factor_dev (real communicator): Gram
    all-reduce route with the front end on, triangle all-gather route with it off."""
    blsq_opt("BLSQ_GRAM", gram)
    from bounded_lsq import _abi
    from bounded_lsq._multi import TsqrTrfSolver
    P, ref = tall
    m, n = P["J"].shape
    ctx = _abi.Context(0)
    sol = TsqrTrfSolver(m, n, 1, 0, ctx=ctx, m_total=m, comm_id=ctx.comm_new_id())
    d = {k: ctx.to_device(P[k]) for k in ("J", "f", "x", "lb", "ub", "scale")}
    ctx.gram_stats(reset=True)
    sol.factor_dev(d["J"], d["f"], d["x"], d["lb"], d["ub"], d["scale"])
    assert ctx.gram_stats() == ((1, 0) if gram == "1" else (0, 0))
    for Delta, So in ref.items():
        S = sol.step(np.array([Delta]), np.array([0.0]))
        assert rel(S.step[0], So.step) < RTOL, (Delta, rel(S.step[0], So.step))
        np.testing.assert_array_equal(S.hits[0], So.hits)
        assert int(S.n_iter[0]) == So.n_iter
    sol.close()
    ctx.comm_destroy()
    ctx.close()


def _k2_max_of(m):
    """gram_k2_max (csrc/chol_kernels.hip)"""
    def acc(mm):
        chunk = 1024.0 if mm > 131072 else 2048.0
        return np.sqrt(min(mm, chunk)) + np.sqrt(np.ceil(mm / chunk))
    return 2.5e5 * min(1.0, acc(4096.0) / acc(float(m)))


def test_tall_problems_at_the_edge_of_the_gate():
    """The error constant of the normal-equations path (step error <= c eps kappa_2, c <= 0.2
    calibrated at m <= 4096) checked where the Gram's accumulation is 60x longer: 250 000 x 128 (the per-rank
    block of BASELINE config 5), unbounded, equicorrelated columns and log-spaced spectra tuned so that the
    TRUE condition number of the equilibrated J^T J sits in the last factor eight below the gate — the sharp
    third stage of the certificate keeps such problems on the fast path.  Step vs the oracle (gesdd on the
    whole matrix) <= 1e-10; the constant c is printed.  Problems just beyond the gate must take the tree."""
    import bounded_lsq as bl
    from bounded_lsq import _abi
    from oracle import blsq_oracle as orc
    m, n = 250_000, 128
    kmax = _k2_max_of(m)
    assert 2.3e5 < kmax < 2.5e5
    rng = np.random.default_rng(123)
    eps = np.finfo(float).eps

    def equicorr(rho):
        Z = rng.standard_normal((m, n)); c = rng.standard_normal((m, 1))
        return np.sqrt(1 - rho) * Z + np.sqrt(rho) * c

    def logspaced(kappa):
        U, _ = np.linalg.qr(rng.standard_normal((m, n)))
        V, _ = np.linalg.qr(rng.standard_normal((n, n)))
        return (U * np.logspace(0, -np.log10(kappa), n)) @ V.T * np.sqrt(m)

    cases = [("equicorr", equicorr(1 - 128 / (0.45 * kmax))), ("equicorr", equicorr(1 - 128 / (0.8 * kmax))),
             ("equicorr", equicorr(1 - 128 / (3.0 * kmax))),
             ("logspaced", logspaced(200.0)), ("logspaced", logspaced(270.0)), ("logspaced", logspaced(900.0))]
    ctx = _abi.Context(0)
    sol = bl.TrfStepSolver(1, m, n, ctx=ctx)
    x = rng.uniform(-1.0, 1.0, n)
    lb = np.full(n, -np.inf); ub = np.full(n, np.inf)
    near, cmax = 0, 0.0
    for tag, J in cases:
        f = rng.standard_normal(m)
        sv = np.linalg.svd(J / np.linalg.norm(J, axis=0), compute_uv=False)
        true_k2 = (sv[0] / sv[-1]) ** 2
        ctx.gram_stats(reset=True)
        sol.factor(J[None], f[None], x[None], lb[None], ub[None], np.ones((1, n)))
        fast = ctx.gram_stats() == (1, 0)
        k2 = float(sol.debug_cond()[0])
        assert k2 == 0 or k2 >= true_k2 * (1 - 1e-6), (tag, k2, true_k2)       # the bound is one
        if true_k2 <= kmax / 4:
            assert fast, (tag, true_k2, k2)                                    # ... and sharp
        if true_k2 > kmax * 1.001:
            assert not fast, (tag, true_k2, k2)
        F = orc.trf_factor(J, f, x, lb, ub, np.ones(n))
        for Delta in (0.5 * np.linalg.norm(orc.trf_step(F, 1e300, 0.0).step), 1e300):
            So = orc.trf_step(F, Delta, 0.0)
            S = sol.step(np.array([Delta]), np.array([0.0]))
            e = rel(S.step[0], So.step)
            assert e < RTOL, (tag, true_k2, fast, Delta, e)
            assert int(S.n_iter[0]) == So.n_iter
            if fast:
                cmax = max(cmax, e / (eps * true_k2))
        near += bool(fast and true_k2 >= kmax / 8)
        print("tall %-9s true kappa_2 %.3e (gate %.3e)  K2 %.3e  %s  step error %.2e"
              % (tag, true_k2, kmax, k2, "fast path" if fast else "tree", e))
    sol.close(); ctx.close()
    print("tall problems near the gate on the fast path: %d, largest c = error / (eps kappa_2) = %.3f" % (near, cmax))
    assert near >= 3, near
    assert cmax < 0.2, cmax


def test_two_million_rows_at_the_edge_of_the_tightened_gate():
    """BASELINE config 5 as written — ONE 2 000 000 x 128 problem — where the gate is tightest: the Gram accumulates
    2 000 000 products per entry (1954 row chunks of 1024), gram_k2_max(m) = 1.5e5 instead of 2.5e5.  Equicorrelated
    columns with the TRUE kappa_2 of the equilibrated J^T J at 0.24 and 0.7 of that gate (unbounded: the system solved
    is J^T J itself): the first must be on the fast path (the certificate is sharp to within its factor four), and
    whatever path either takes, the step is within 1e-10 of the oracle (gesdd on the whole matrix, trf.py:272);
    c = error / (eps kappa_2) of a fast-path problem must stay below the calibrated 0.2."""
    import bounded_lsq as bl
    from bounded_lsq import _abi
    from oracle import blsq_oracle as orc
    m, n = 2_000_000, 128
    kmax = _k2_max_of(m)
    assert 1.4e5 < kmax < 1.6e5, kmax
    rng = np.random.default_rng(2026)
    eps = np.finfo(float).eps
    ctx = _abi.Context(0)
    sol = bl.TrfStepSolver(1, m, n, ctx=ctx)
    x = rng.uniform(-1.0, 1.0, n)
    lb = np.full(n, -np.inf); ub = np.full(n, np.inf)
    cmax = 0.0
    for frac in (0.24, 0.7):
        rho = 1.0 - n / (frac * kmax)
        J = np.sqrt(1 - rho) * rng.standard_normal((m, n))
        J += np.sqrt(rho) * rng.standard_normal((m, 1))
        f = rng.standard_normal(m)
        G = J.T @ J
        dsc = 1.0 / np.sqrt(np.diag(G))
        ev = np.linalg.eigvalsh(G * dsc[:, None] * dsc[None, :])
        true_k2 = ev[-1] / ev[0]                                 # (to ~1e-10 relative: eps kappa_2)
        ctx.gram_stats(reset=True)
        sol.factor(J[None], f[None], x[None], lb[None], ub[None], np.ones((1, n)))
        fast = ctx.gram_stats() == (1, 0)
        k2 = float(sol.debug_cond()[0])
        assert k2 == 0 or not np.isfinite(k2) or k2 >= true_k2 * (1 - 1e-6), (frac, k2, true_k2)
        if true_k2 <= kmax / 4:
            assert fast, (frac, true_k2, k2)
        F = orc.trf_factor(J, f, x, lb, ub, np.ones(n))          # (the host's LAPACK on 2 000 128 x 128: ~20 s)
        pg = np.linalg.norm(orc.trf_step(F, 1e300, 0.0).step)
        for Delta in (0.5 * pg, 1e300):
            So = orc.trf_step(F, Delta, 0.0)
            S = sol.step(np.array([Delta]), np.array([0.0]))
            e = rel(S.step[0], So.step)
            assert e < RTOL, (frac, true_k2, fast, Delta, e)
            assert int(S.n_iter[0]) == So.n_iter
            if fast:
                cmax = max(cmax, e / (eps * true_k2))
        print("2e6 rows: true kappa_2 %.3e = %.2f of the gate %.3e, K2 %.3e, %s, step error %.2e"
              % (true_k2, true_k2 / kmax, kmax, k2, "fast path" if fast else "rejected", e))
        del J, F
    sol.close(); ctx.close()
    print("2e6 rows: largest c = error / (eps kappa_2) on the fast path = %.3f" % cmax)
    assert cmax < 0.2, cmax
