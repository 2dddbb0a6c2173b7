"""Normal-equations fast path of the factorisation (gram_kernels.hip, chol_kernels.hip) and its conditioning gate.

The step must match the CPU oracle to 1e-10 / bit-exact masks WHICHEVER path factors a problem;
the diagnostic counter (blsq_debug_gram_stats) shows which one ran.

TRF: the gate looks at the system the step is solved from — the equilibrated H = D G D + E^2 of the
Coleman-Li augmented Jacobian [J D; E] (trf.py:264-270), factored by Cholesky straight from the
Gram; no triangle of J is formed for a problem that passes.  For an UNBOUNDED problem D = I, E = 0,
H = J^T J: columns that all share a common component with cosine rho give sigma_min(R') =
sqrt(1 - rho), and the gate (pivots and sigma_min estimate >= 0.1) passes up to rho ~ 0.98-0.99 and
must reject beyond.  With bounds close to x the E^2 block dominates H and even a badly conditioned
J is solved accurately through H — the routing tests therefore use unbounded problems, and
`test_bounded_problems_are_gated_on_the_augmented_system` covers the other case.
dogbox: likewise the gate looks at the factor of the free-column system [J[:, free] | f]
(dogbox.py:197), the Cholesky factor of the gathered principal sub-matrix of the Gram."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL = 1e-10


def rel(a, b):
    a = np.asarray(a, float); b = np.asarray(b, float)
    den = np.linalg.norm(b)
    return np.linalg.norm(a - b) / (den if den > 0 else 1.0)


@pytest.fixture(scope="module")
def bl():
    import bounded_lsq
    return bounded_lsq


def _equicorrelated(B, m, n, rho, seed):
    rng = np.random.default_rng(seed)
    Z = rng.standard_normal((B, m, n))
    common = rng.standard_normal((B, m, 1))
    rho = np.broadcast_to(np.asarray(rho, float).reshape(-1, 1, 1), (B, 1, 1))
    return np.sqrt(1 - rho) * Z + np.sqrt(rho) * common


K2_MAX = 2.5e5          # GRAM_K2_MAX of csrc/blsq_kernels.h


def k2_max_of(m):
    """gram_k2_max (csrc/chol_kernels.hip): the gate for a Gram accumulated over m rows"""
    def acc(mm):
        chunk = 1024.0 if mm > 131072 else 2048.0
        return np.sqrt(min(mm, chunk)) + np.sqrt(np.ceil(mm / chunk))
    return K2_MAX * min(1.0, acc(4096.0) / acc(float(m)))


def _certificate_holds(P, k2, stats, sharp=8.0):
    """The gate's verdicts are consistent with its bound, and the bound IS one: for an unbounded TRF
    problem the system solved is the column-equilibrated J^T J, whose true condition number must
    not exceed K2 (k2 == 0: rejected on a Cholesky pivot before the bound was computed).  And the
    certificate is SHARP: its third stage (Cholesky of C - tau I) decides kappa_2 <= k2_max up to the
    overestimate of lambda_max alone, so a problem whose true condition number is below k2_max / `sharp`
    must be on the fast path."""
    B, m = P["J"].shape[0], P["J"].shape[1]
    kmax = k2_max_of(m)
    fast = 0
    for b in range(B):
        Jn = P["J"][b] / np.linalg.norm(P["J"][b], axis=0)
        sv = np.linalg.svd(Jn, compute_uv=False)
        true_k2 = (sv[0] / sv[-1]) ** 2
        if k2[b] > 0:
            assert k2[b] >= true_k2 * (1 - 1e-6), (b, k2[b], true_k2)
        else:
            assert true_k2 > 1e5          # a pivot below 1e-3: sigma_min(R') is below it too
        fast += 0 < k2[b] <= kmax * (1 + 1e-12)
        if true_k2 <= kmax / sharp:
            assert 0 < k2[b] <= kmax * (1 + 1e-12), ("certificate not sharp", b, k2[b], true_k2)
    assert stats == (fast, B - fast), (stats, k2)


def _check(bl, P, Delta, kind="trf"):
    """-> (path counts, worst step error); the proven bounds of the call are left in _check.k2"""
    from oracle import blsq_oracle as orc
    from bounded_lsq import _abi
    B, m, n = P["J"].shape
    ctx = _abi.Context(0)
    worst = 0.0
    if kind == "trf":
        sol = bl.TrfStepSolver(B, m, n, ctx=ctx)
        ctx.gram_stats(reset=True)
        sol.factor(P["J"], P["f"], P["x"], P["lb"], P["ub"], P["scale"])
        stats = ctx.gram_stats()
        _check.k2 = sol.debug_cond()
        S = sol.step(Delta, np.zeros(B))
        for b in range(B):
            _, So = orc.trf_step_solve(P["J"][b], P["f"][b], P["x"][b], P["lb"][b], P["ub"][b],
                                       P["scale"][b], Delta[b], 0.0)
            e = rel(S.step[b], So.step)
            assert e < RTOL, (b, e, stats)
            np.testing.assert_array_equal(S.hits[b], So.hits)
            worst = max(worst, e)
    else:
        sol = bl.DogboxStepSolver(B, m, n, ctx=ctx)
        ctx.gram_stats(reset=True)
        sol.factor(P["J"], P["f"], P["x"], P["lb"], P["ub"], P["scale"], P["on_bound"])
        stats = ctx.gram_stats()
        _check.k2 = sol.debug_cond()
        S = sol.step(Delta)
        for b in range(B):
            _, So = orc.dogbox_step_solve(P["J"][b], P["f"][b], P["x"][b], P["lb"][b], P["ub"][b],
                                          P["scale"][b], P["on_bound"][b], Delta[b])
            e = rel(S.step[b], So.step)
            assert e < RTOL, (b, e, stats)
            np.testing.assert_array_equal(S.on_bound_new[b], So.on_bound_new)
            worst = max(worst, e)
    sol.close(); ctx.close()
    return stats, worst


@pytest.mark.parametrize("B,m,n", [(6, 512, 64), (3, 2000, 256), (4, 1500, 128), (5, 300, 16),
                                   (3, 700, 100), (2, 4096, 200), (3, 900, 271),
                                   # n % 16 == 0 where the rhs column leaves the MFMA tiles
                                   (2, 1000, 112), (2, 2100, 160), (2, 3000, 208), (2, 5000, 256),
                                   # narrow: direct-from-global kernel, 1..4 column tiles
                                   (4, 3000, 7), (4, 5000, 31), (3, 2500, 47), (3, 2200, 62),
                                   # ... and n = 16, 32, 48, 64: rhs column from the operand fragments
                                   (4, 900, 16), (3, 2000, 32), (3, 1000, 48), (5, 700, 64), (2, 5000, 64)])
def test_well_conditioned_batches_take_the_fast_path(bl, B, m, n):
    from bounded_lsq import _synth
    P = _synth.trf_batch(77 + n, B, m, n)
    stats, _ = _check(bl, P, np.where(np.arange(B) % 2 == 0, 10.0, 0.5))
    assert stats == (B, 0)


@pytest.mark.parametrize("n", [48, 256])
@pytest.mark.parametrize("rho", [0.5, 0.9, 0.96, 0.98, 0.99, 0.995, 0.999, 1 - 1e-8])
def test_gate_around_its_threshold(bl, rho, n):
    """Equicorrelated columns: kappa_2 of the equilibrated J^T J is (1 + (n - 1) rho) / (1 - rho).
    Whatever the gate decides the step matches the oracle (inside _check); its bound K2 is a true
    upper bound of that condition number; problems are on the fast path exactly when K2 <= 2.5e5.
    (Measured, tests/gate_calib.py: K2 / kappa_2 = 5 .. 150 on this family; the fast path's step
    error stays below 1e-12 up to the gate.)"""
    from bounded_lsq import _synth
    B, m = 3, 2048
    P = _synth.trf_batch(31, B, m, n, unbounded=True)
    P["J"] = _equicorrelated(B, m, n, rho, 5)
    stats, worst = _check(bl, P, np.array([10.0, 0.5, 2.0]))
    _certificate_holds(P, _check.k2, stats)
    if rho <= 0.5:
        assert stats == (B, 0), stats                   # clearly inside: fast path
    if rho >= 1 - 1e-8:
        assert stats == (0, B), stats                   # clearly outside: Householder tree
    if stats[1] == 0:
        assert worst < 1e-11                            # the fast path has a wide margin


@pytest.mark.parametrize("kappa", [2.0, 10.0, 30.0, 100.0, 300.0, 1e3, 1e5])
def test_logspaced_spectrum_across_the_gate(bl, kappa):
    """J = U diag(s) V^T with singular values log-spaced over [1/kappa, 1]: many small singular
    values at once (the equicorrelated family has only one direction that matters)."""
    from bounded_lsq import _synth
    B, m, n = 3, 1200, 80
    P = _synth.trf_batch(17, B, m, n, unbounded=True)
    rng = np.random.default_rng(11)
    J = np.empty((B, m, n))
    for b in range(B):
        U, _ = np.linalg.qr(rng.standard_normal((m, n)))
        V, _ = np.linalg.qr(rng.standard_normal((n, n)))
        J[b] = (U * np.logspace(0, -np.log10(kappa), n)) @ V.T
    P["J"] = J
    stats, worst = _check(bl, P, np.array([10.0, 0.5, 0.05]))
    _certificate_holds(P, _check.k2, stats)
    if kappa <= 100.0:
        assert stats == (B, 0)                          # kappa_2 = 1e4: inside (the norm bounds alone stop at kappa ~ 200)
    if kappa >= 1e3:
        assert stats == (0, B)
    if stats[1] == 0:
        assert worst < 1e-11


def _logspaced(B, m, n, kappa, rng):
    J = np.empty((B, m, n))
    for b in range(B):
        U, _ = np.linalg.qr(rng.standard_normal((m, n)))
        V, _ = np.linalg.qr(rng.standard_normal((n, n)))
        J[b] = (U * np.logspace(0, -np.log10(kappa), n)) @ V.T
    return J


def test_the_fast_path_at_the_edge_of_the_gate(bl):
    """Problems tuned to land just inside the gate — TRUE kappa_2 of the equilibrated system within a
    factor four of k2_max (the third certificate stage is sharp, so the fast path really runs there) — from
    three families.  The worst step error there gives the empirical constant of the bound
    step error <= c eps kappa_2 of DESIGN.md 3.0: c <= 0.2, i.e. <= 5.6e-12 at the gate, 18x under the bar."""
    from bounded_lsq import _synth
    rng = np.random.default_rng(4)
    cases = []
    for (m, n, rho) in ((2048, 48, 0.9996), (2048, 48, 0.9998), (2048, 256, 0.997), (2048, 256, 0.9985),
                        (4096, 128, 0.9990), (4096, 128, 0.9993)):              # equicorrelated
        P = _synth.trf_batch(61, 2, m, n, unbounded=True)
        P["J"] = _equicorrelated(2, m, n, rho, 9)
        cases.append(P)
    for (m, n, kappa) in ((1200, 80, 300.0), (1200, 80, 420.0), (4096, 256, 350.0), (4096, 256, 450.0)):
        P = _synth.trf_batch(62, 2, m, n, unbounded=True)                        # log-spaced spectrum
        P["J"] = _logspaced(2, m, n, kappa, rng)
        cases.append(P)
    near, worst_all, cmax = 0, 0.0, 0.0
    eps = np.finfo(float).eps
    for P in cases:
        B, m = P["J"].shape[0], P["J"].shape[1]
        stats, worst = _check(bl, P, np.full(B, 0.5))
        _certificate_holds(P, _check.k2, stats)
        k2 = _check.k2
        for b in range(B):
            Jn = P["J"][b] / np.linalg.norm(P["J"][b], axis=0)
            sv = np.linalg.svd(Jn, compute_uv=False)
            true_k2 = (sv[0] / sv[-1]) ** 2
            if 0 < k2[b] <= k2_max_of(m) * (1 + 1e-12) and true_k2 >= k2_max_of(m) / 4:
                near += 1
        if stats[1] == 0:
            worst_all = max(worst_all, worst)
    print("edge of the gate: %d problems on the fast path with true kappa_2 within 4x of the gate, worst "
          "step error %.2e" % (near, worst_all))
    assert near >= 8, "the cases must probe the last factor four below the gate: %d" % near
    assert worst_all < 1e-11, worst_all


@pytest.mark.parametrize("n,s_", [(32, 0.95), (64, 0.97), (128, 0.992)])
def test_kahan_matrix_pivots_pass_but_sigma_min_does_not(bl, n, s_):
    """Adversarial for a pivot-only gate (and for estimates of sigma_min that start from an unlucky
    vector): J = Q K with K the n x n Kahan matrix — unit columns after equilibration, every
    Cholesky pivot moderate, yet sigma_min orders of magnitude smaller.  The certificate (explicit
    inverse) must send it to the Householder tree, and the step must match the oracle."""
    from bounded_lsq import _synth
    B, m = 2, 1024
    c_ = np.sqrt(1 - s_ ** 2)
    K = np.zeros((n, n))
    for i in range(n):
        K[i, i] = s_ ** i
        K[i, i + 1:] = -c_ * s_ ** i
    Kn = K / np.linalg.norm(K, axis=0)
    assert np.min(np.abs(np.diag(np.linalg.qr(Kn)[1]))) > 0.1          # pivots of the unit-column matrix
    assert np.linalg.svd(Kn, compute_uv=False)[-1] < 0.02             # ... but a tiny sigma_min
    rng = np.random.default_rng(3)
    P = _synth.trf_batch(23, B, m, n, unbounded=True)
    J = np.empty((B, m, n))
    for b in range(B):
        Q, _ = np.linalg.qr(rng.standard_normal((m, n)))
        J[b] = Q @ K
    P["J"] = J
    stats, _ = _check(bl, P, np.array([10.0, 0.5]))
    assert stats == (0, B), stats
    assert np.all((_check.k2 == 0) | (_check.k2 > 1e8))     # the proven bound sees what the pivots do not


def test_mixed_batch_splits_between_the_paths(bl):
    from bounded_lsq import _synth
    B, m, n = 8, 1024, 96
    P = _synth.trf_batch(8, B, m, n, unbounded=True)
    rho = np.where(np.arange(B) % 2 == 0, 0.3, 0.9999)
    P["J"] = _equicorrelated(B, m, n, rho, 3)
    stats, _ = _check(bl, P, np.full(B, 1.0))
    assert stats == (B // 2, B // 2)


def test_rank_deficient_zero_and_nonfinite_columns_go_to_the_tree(bl):
    """(The reference's answer for a rank-deficient J is decided by rounding noise in the null
    space — tests/test_hip_parity.py KNIFE_EDGE — so only the routing is asserted here; parity of
    rank-deficient fixtures is covered by the golden tests, which run with the front end on.)"""
    from bounded_lsq import _synth, _abi
    B, m, n = 4, 600, 40
    P = _synth.trf_batch(4, B, m, n, unbounded=True)
    P["J"][0][:, 7] = 0.0                               # zero column
    P["J"][1][:, 9] = P["J"][1][:, 3]                   # duplicate column
    P["J"][2][:, 11] = 2.0 * P["J"][2][:, 0] - P["J"][2][:, 5]   # dependent column
    ctx = _abi.Context(0)
    sol = bl.TrfStepSolver(B, m, n, ctx=ctx)
    ctx.gram_stats(reset=True)
    sol.factor(P["J"], P["f"], P["x"], P["lb"], P["ub"], P["scale"])
    assert ctx.gram_stats() == (1, 3)
    S = sol.step(np.full(B, 0.7), np.zeros(B))
    assert np.all(np.isfinite(S.step))
    sol.close()
    sol = bl.TrfStepSolver(1, m, n, ctx=ctx)
    J = P["J"][3:4].copy(); J[0, 5, 5] = np.nan
    ctx.gram_stats(reset=True)
    sol.factor(J, P["f"][3:4], P["x"][3:4], P["lb"][3:4], P["ub"][3:4], P["scale"][3:4])
    assert ctx.gram_stats() == (0, 1)                   # non-finite input never passes the gate
    sol.close(); ctx.close()


def test_wide_and_square_problems(bl):
    from bounded_lsq import _synth
    for (B, m, n) in [(3, 64, 64), (2, 40, 64), (3, 17, 16)]:
        P = _synth.trf_batch(12, B, m, n)
        _check(bl, P, np.full(B, 0.9))


def test_dogbox_uses_the_same_front_end(bl):
    from bounded_lsq import _synth
    B, m, n = 6, 800, 72
    P = _synth.dogbox_batch(21, B, m, n)
    stats, _ = _check(bl, P, np.full(B, 0.05), kind="dogbox")
    assert stats == (B, 0)
    P["J"] = _equicorrelated(B, m, n, 0.9999, 2)
    stats, _ = _check(bl, P, np.full(B, 0.05), kind="dogbox")
    assert stats == (0, B)


def test_env_switch_disables_the_fast_path(bl, monkeypatch, blsq_opt):
    from bounded_lsq import _synth
    blsq_opt("BLSQ_GRAM", "0")
    P = _synth.trf_batch(5, 3, 400, 32)
    stats, _ = _check(bl, P, np.full(3, 0.9))
    assert stats == (0, 0)                              # the front end never ran


def test_single_tall_problem_row_chunks(bl):
    """B = 1: the Gram is accumulated by many workgroups over row chunks and reduced."""
    from bounded_lsq import _synth
    P = _synth.trf_batch(9, 1, 40000, 64)
    stats, _ = _check(bl, P, np.array([0.8]))
    assert stats == (1, 0)


def test_badly_scaled_columns_are_equilibrated(bl, monkeypatch, blsq_opt):
    """Column norms spread over many orders of magnitude: the equilibrated Gram is as well
    conditioned as the normalised columns, so the fast path is taken and is as accurate as the
    Householder tree.  (At a spread of 1e+-6, cond(J) ~ 1e11, the reference's own SVD answer moves
    by ~1e-10 — both paths then differ from the oracle by the same amount, and from each other by
    far less.)"""
    from bounded_lsq import _synth, _abi
    B, m, n = 3, 1000, 60
    P = _synth.trf_batch(15, B, m, n)
    rng = np.random.default_rng(0)
    J0 = P["J"]
    P["J"] = J0 * 10.0 ** rng.uniform(-3, 3, size=(B, 1, n))
    stats, _ = _check(bl, P, np.full(B, 0.5))
    assert stats == (B, 0)
    P["J"] = J0 * 10.0 ** rng.uniform(-6, 6, size=(B, 1, n))
    steps = {}
    for mode in ("1", "0"):
        blsq_opt("BLSQ_GRAM", mode)
        ctx = _abi.Context(0)
        sol = bl.TrfStepSolver(B, m, n, ctx=ctx)
        ctx.gram_stats(reset=True)
        sol.factor(P["J"], P["f"], P["x"], P["lb"], P["ub"], P["scale"])
        assert ctx.gram_stats() == ((B, 0) if mode == "1" else (0, 0))
        steps[mode] = sol.step(np.full(B, 0.5), np.zeros(B)).step.copy()
        sol.close(); ctx.close()
    for b in range(B):
        assert rel(steps["1"][b], steps["0"][b]) < 1e-11


def test_tile_groups_do_not_change_a_single_bit(bl, monkeypatch, blsq_opt):
    """Small batches split the Gram's tiles over several workgroups; every tile still accumulates
    the same k-steps in the same order, so the whole step is bitwise the same for any split."""
    from bounded_lsq import _synth, _abi
    for (B, m, n) in [(1, 4096, 256), (2, 3000, 200), (3, 2100, 128), (2, 1500, 100)]:
        P = _synth.trf_batch(41 + n, B, m, n)
        Delta = np.full(B, 0.7)
        outs = []
        for tg in ("1", "3", "8"):
            blsq_opt("BLSQ_GRAM_TILE_GROUPS", tg)
            ctx = _abi.Context(0)
            sol = bl.TrfStepSolver(B, m, n, ctx=ctx)
            ctx.gram_stats(reset=True)
            sol.factor(P["J"], P["f"], P["x"], P["lb"], P["ub"], P["scale"])
            assert ctx.gram_stats() == (B, 0)
            outs.append(sol.step(Delta, np.zeros(B)).step.copy())
            sol.close(); ctx.close()
        assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])


def test_one_tile_per_wave_kernel_matches_the_generic_one_bit_for_bit(bl, monkeypatch, blsq_opt):
    """At most four (row chunk, problem) pairs with n a multiple of 16 run gram1_kernel: a wave owns ONE output tile and
    reads its operand fragments straight from global memory, four more workgroups per chunk reproduce the generic
    kernel's sums of the rhs column.  Same k-steps in the same order with the same instruction: the whole step and
    g = J^T f are bitwise those of the tile-group launch (option gram1 = 0) — so a problem solved alone, as the drop-in
    front end does, has the bits it has inside a large batch.  Row counts: one chunk, two chunks, a last k-step of
    fewer than four rows, a last 32-row block that is not full."""
    from bounded_lsq import _synth, _abi
    for (B, m, n) in [(1, 4096, 256), (2, 4096, 256), (1, 2048, 240), (1, 2311, 96), (2, 1001, 80), (1, 3000, 112),
                      (1, 2050, 144), (1, 4096, 160), (3, 900, 256), (1, 700, 256)]:
        P = _synth.trf_batch(61 + n + m, B, m, n)
        Delta = np.full(B, 0.7)
        outs = []
        for g1 in ("1", "0"):
            blsq_opt("BLSQ_GRAM1", g1)
            ctx = _abi.Context(0)
            sol = bl.TrfStepSolver(B, m, n, ctx=ctx)
            ctx.gram_stats(reset=True)
            sol.factor(P["J"], P["f"], P["x"], P["lb"], P["ub"], P["scale"])
            assert ctx.gram_stats() == (B, 0)
            S = sol.step(Delta, np.zeros(B))
            outs.append((S.step.copy(), sol.fetch_factor().g.copy(), S.alpha.copy()))
            sol.close(); ctx.close()
        assert np.array_equal(outs[0][1], outs[1][1]), (B, m, n)     # g = J^T f: the rhs column
        assert np.array_equal(outs[0][0], outs[1][0]), (B, m, n)
        assert np.array_equal(outs[0][2], outs[1][2]), (B, m, n)
    # ... and inside a batch large enough for the static-tile-row kernel: the same bits for problem 0
    B, m, n = 300, 4096, 256
    P = _synth.trf_batch(61 + n + m, 1, m, n)
    Pb = _synth.trf_batch(7, B, m, n)
    for k in P:
        Pb[k][0] = P[k][0]
    res = []
    for Q, nb in ((P, 1), (Pb, B)):
        ctx = _abi.Context(0)
        sol = bl.TrfStepSolver(nb, m, n, ctx=ctx)
        sol.factor(Q["J"], Q["f"], Q["x"], Q["lb"], Q["ub"], Q["scale"])
        res.append(sol.step(np.full(nb, 0.7), np.zeros(nb)).step[0].copy())
        sol.close(); ctx.close()
    assert np.array_equal(res[0], res[1])


def test_static_tile_row_kernel_matches_the_generic_one_bit_for_bit(bl, monkeypatch, blsq_opt):
    """n = 241 .. 256 (16 column tiles) with one workgroup per row chunk runs gram16_kernel (static
    tile rows per wave, shared operand fragments); every tile still accumulates the same k-steps in
    the same order and the rhs column is summed in the same order as in the generic kernel, so the
    whole step is bitwise the same (BLSQ_GRAM16 = 0 forces the generic kernel) — and so a problem's
    bits do not depend on whether its batch was small enough for tile groups."""
    from bounded_lsq import _synth, _abi
    for (B, m, n) in [(3, 4096, 256), (2, 2100, 250), (2, 5000, 241), (2, 1000, 256), (2, 900, 255)]:
        P = _synth.trf_batch(51 + n, B, m, n)
        Delta = np.full(B, 0.7)
        outs = []
        for g16, tg in (("1", "1"), ("0", "1"), ("1", "4")):
            blsq_opt("BLSQ_GRAM16", g16)
            blsq_opt("BLSQ_GRAM_TILE_GROUPS", tg)
            ctx = _abi.Context(0)
            sol = bl.TrfStepSolver(B, m, n, ctx=ctx)
            ctx.gram_stats(reset=True)
            sol.factor(P["J"], P["f"], P["x"], P["lb"], P["ub"], P["scale"])
            assert ctx.gram_stats() == (B, 0)
            S = sol.step(Delta, np.zeros(B))
            outs.append((S.step.copy(), sol.fetch_factor().g.copy()))
            sol.close(); ctx.close()
        for o in outs[1:]:
            assert np.array_equal(outs[0][1], o[1])        # g = J^T f: the rhs column
            assert np.array_equal(outs[0][0], o[0])


@pytest.mark.parametrize("n", [15, 16, 17, 47, 48, 49, 62, 63, 64, 65, 79, 80, 81, 111, 112, 113,
                               127, 128, 129, 239, 240, 241, 255, 256, 257])
def test_every_kernel_dispatch_boundary(bl, n):
    """Widths on both sides of every switch between kernels — narrow (direct from global) / tile
    table / 8 column tiles / 16 column tiles for the Gram, one wave with the matrix in registers
    (N <= 80) / a workgroup per problem for the Cholesky and the certificate, rhs column inside or
    outside the MFMA tiles (n % 16) — both solvers, two row counts (one and several row chunks),
    against the oracle at the 1e-10 bar with bit-exact masks."""
    from bounded_lsq import _synth
    for m in (3 * n + 5, 2048 + 3 * n):
        P = _synth.trf_batch(500 + n, 2, m, n)
        stats, _ = _check(bl, P, np.array([0.7, 0.05]))
        assert stats == (2, 0)
        Pd = _synth.dogbox_batch(900 + n, 2, m, n)
        stats, _ = _check(bl, Pd, np.array([0.7, 0.05]), kind="dogbox")
        assert stats == (2, 0)


@pytest.mark.parametrize("n", [16, 32, 48, 64])
@pytest.mark.parametrize("m", [130, 257, 515, 1000, 1029, 2048, 2090])
def test_narrow_gram_full_rounds_and_ragged_tails(bl, m, n, blsq_opt):
    """gram_direct_kernel with the rhs column outside its tiles (n a multiple of 16): the FULL rounds of a row chunk run
    a loop without clamps or masks (pointers advanced by a scalar, the next round's requests unconditional), the rest —
    a last round that is not full, fewer rows than one round, a second chunk of a few rows — the masked loop; two, four
    or eight waves per workgroup by the row count.  Whatever the split between the two loops and the number of waves:
    within 1e-10 of the oracle, masks bit-exact, and the same bits for 2, 4 and 8 waves wherever the option applies
    (the k-steps a wave takes depend on the wave count, so the sums differ — parity is the bar there)."""
    from bounded_lsq import _synth
    P = _synth.trf_batch(300 + m + n, 3, m, n)
    stats, w1 = _check(bl, P, np.array([0.7, 0.05, 3.0]))
    assert stats == (3, 0)
    Pd = _synth.dogbox_batch(700 + m + n, 3, m, n)
    stats, w2 = _check(bl, Pd, np.array([0.7, 0.05, 3.0]), kind="dogbox")
    assert stats == (3, 0)
    for nw in (2, 4, 8):
        blsq_opt("gram_direct_nw", nw)
        stats, _ = _check(bl, P, np.array([0.7, 0.05, 3.0]))
        assert stats == (3, 0)


@pytest.mark.parametrize("n", [6, 16, 48, 64, 79])
def test_newton_rounds_in_one_launch(bl, monkeypatch, n, blsq_opt):
    """N <= 80: the Gauss-Newton step, the bracket and every Newton round of a normal-equations-path
    problem run inside ONE launch (lm_rounds_reg_kernel).  Against the round-by-round kernels
    (BLSQ_LM_FUSED = 0): the same iteration counts, alpha and step to rounding — and bit for bit the
    same result for a problem whether or not its batch also holds Householder-path problems (which
    keep the round-by-round loop)."""
    from bounded_lsq import _synth, _abi
    B, m = 6, 40 * n + 30
    P = _synth.trf_batch(300 + n, B, m, n)
    Delta = np.array([0.5, 0.05, 5.0, 0.2, 0.01, 1.0])

    def run(PP, DD, fused):
        blsq_opt("BLSQ_LM_FUSED", fused)
        ctx = _abi.Context(0)
        sol = bl.TrfStepSolver(len(DD), m, n, ctx=ctx)
        ctx.gram_stats(reset=True)
        sol.factor(PP["J"], PP["f"], PP["x"], PP["lb"], PP["ub"], PP["scale"])
        stats = ctx.gram_stats()
        S = sol.step(DD, np.zeros(len(DD)))
        out = (S.step.copy(), np.asarray(S.alpha).copy(), np.asarray(S.n_iter).copy(), S.hits.copy())
        sol.close(); ctx.close()
        return out, stats

    (st1, a1, it1, h1), stats = run(P, Delta, "1")
    assert stats == (B, 0)
    (st0, a0, it0, h0), _ = run(P, Delta, "0")
    assert np.array_equal(it1, it0) and it1.max() >= 1
    np.testing.assert_allclose(a1, a0, rtol=1e-10, atol=0)
    for b in range(B):
        assert rel(st1[b], st0[b]) < 1e-11
    np.testing.assert_array_equal(h1, h0)
    # the same problems beside two that fail the gate (no bounds, equicorrelated columns)
    Q = {k: np.concatenate([v, v[:2]]) for k, v in P.items()}
    Q["J"][B:] = _equicorrelated(2, m, n, 1 - 1e-9, 7)
    Q["lb"][B:] = -np.inf; Q["ub"][B:] = np.inf
    (stm, am, itm, hm), statsm = run(Q, np.concatenate([Delta, Delta[:2]]), "1")
    if n >= 2:
        assert statsm == (B, 2)
    assert np.array_equal(stm[:B], st1) and np.array_equal(am[:B], a1) and np.array_equal(itm[:B], it1)


@pytest.mark.parametrize("scale_mode", [0, 1])
@pytest.mark.parametrize("shape", [(700, 64), (1500, 200)])
def test_optimistic_verdict_of_the_device_api(bl, monkeypatch, shape, scale_mode, blsq_opt):
    """blsq_trf_factor_dev does not wait for the gate's counters: it guesses "everybody stays on the
    normal-equations path, nobody needs the SVD", blsq_trf_step_dev enqueues its kernels on that guess
    and only then reads the verdict — a wrong guess runs the fallback stage and the step once more.
    Right guess, wrong guess (problems that fail the certificate; a rank-deficient problem that needs
    the SVD) and the synchronous mode (BLSQ_OPTIMISTIC = 0) must give the same bits, and the path
    statistics must come out the same."""
    from bounded_lsq import _synth, _abi
    m, n = shape
    B = 6
    good = _synth.trf_batch(400 + n, B, m, n)
    mixed = {k: v.copy() for k, v in good.items()}
    mixed["J"][1] = _equicorrelated(1, m, n, 1 - 1e-9, 3)[0]          # fails the certificate
    mixed["lb"][1] = -np.inf; mixed["ub"][1] = np.inf
    mixed["J"][4][:, n - 1] = mixed["J"][4][:, 0]                        # rank deficient: pivot gate, then the SVD
    mixed["lb"][4] = -np.inf; mixed["ub"][4] = np.inf
    Delta = np.array([0.5, 0.05, 5.0, 0.2, 0.3, 1.0])
    for P, expect_fb in ((good, 0), (mixed, 2)):
        outs = []
        for opt in ("1", "0"):
            blsq_opt("BLSQ_OPTIMISTIC", opt)
            ctx = _abi.Context(0)
            sol = bl.TrfStepSolver(B, m, n, ctx=ctx)
            d = {k: ctx.to_device(P[k]) for k in ("J", "f", "x", "lb", "ub", "scale")}
            dD, dA = ctx.to_device(Delta), ctx.to_device(np.zeros(B))
            ctx.gram_stats(reset=True)
            for _ in range(2):                                          # (a second round on the settled plan)
                sol.factor_dev(d["J"], d["f"], d["x"], d["lb"], d["ub"], d["scale"], scale_mode)
                sol.step_dev(dD, dA)
            S = sol.fetch_step()
            stats = ctx.gram_stats()
            assert stats == (2 * (B - expect_fb), 2 * expect_fb), (opt, stats)
            outs.append((S.step.copy(), np.asarray(S.alpha).copy(), np.asarray(S.n_iter).copy(), S.hits.copy(),
                         sol.fetch_factor().g.copy(), ctx.to_host(d["scale"], (B, n), np.float64)))
            sol.close()
            for v in list(d.values()) + [dD, dA]:
                ctx.free(v)
            ctx.close()
        for x1, x0 in zip(*outs):
            assert np.array_equal(x1, x0)


@pytest.mark.parametrize("how", ["sync", "h2d", "free"])
def test_pending_verdict_is_resolved_before_the_callers_jacobian_can_change(bl, how):
    """Lifetime rule of the optimistic device API (include/blsq.h): a wrong guess is repaired from the
    caller's J, so J must stay untouched until the verdict has been read — which blsq_sync does, and so do
    the library's own blsq_memcpy_h2d / blsq_dev_free.  A caller that syncs (or uploads the next Jacobian
    through the library, or frees J) between factor_dev and step_dev gets the step of the ORIGINAL J on
    exactly the batches where the fallback runs."""
    from bounded_lsq import _synth, _abi
    B, m, n = 6, 1500, 200
    P = _synth.trf_batch(4400, B, m, n)
    P["J"][2] = _equicorrelated(1, m, n, 1 - 1e-9, 3)[0]              # fails the certificate
    P["lb"][2] = -np.inf; P["ub"][2] = np.inf
    Delta = np.array([0.5, 0.05, 5.0, 0.2, 0.3, 1.0])
    outs = []
    for clobber in (False, True):
        ctx = _abi.Context(0)
        sol = bl.TrfStepSolver(B, m, n, ctx=ctx)
        d = {k: ctx.to_device(P[k]) for k in ("J", "f", "x", "lb", "ub", "scale")}
        dD, dA = ctx.to_device(Delta), ctx.to_device(np.zeros(B))
        ctx.gram_stats(reset=True)
        sol.factor_dev(d["J"], d["f"], d["x"], d["lb"], d["ub"], d["scale"])
        if clobber:
            junk = np.full((B, m, n), 7.25)
            if how == "sync":
                ctx.sync()                                             # the verdict is read here ...
                ctx.check(ctx.lib.blsq_memcpy_h2d(ctx.h, d["J"], _abi.ptr(junk), junk.nbytes), "h2d")
            elif how == "h2d":                                         # ... or by the upload itself
                ctx.check(ctx.lib.blsq_memcpy_h2d(ctx.h, d["J"], _abi.ptr(junk), junk.nbytes), "h2d")
            else:
                ctx.free(d.pop("J"))                                   # ... or by the free
        sol.step_dev(dD, dA)
        S = sol.fetch_step()
        assert ctx.gram_stats() == (B - 1, 1)
        outs.append((S.step.copy(), S.hits.copy(), np.asarray(S.n_iter).copy()))
        sol.close()
        for v in list(d.values()) + [dD, dA]:
            ctx.free(v)
        ctx.close()
    for a, b in zip(*outs):
        assert np.array_equal(a, b)


def test_a_dropped_verdict_is_still_counted(bl):
    """A second factor_dev before any step drops the first call's pending verdict: no repair (the factor is
    being overwritten), but the path statistics and the decision to guess again still see it."""
    from bounded_lsq import _synth, _abi
    B, m, n = 4, 1500, 200
    P = _synth.trf_batch(4500, B, m, n)
    P["J"][1] = _equicorrelated(1, m, n, 1 - 1e-9, 5)[0]
    P["lb"][1] = -np.inf; P["ub"][1] = np.inf
    ctx = _abi.Context(0)
    sol = bl.TrfStepSolver(B, m, n, ctx=ctx)
    d = {k: ctx.to_device(P[k]) for k in ("J", "f", "x", "lb", "ub", "scale")}
    dD, dA = ctx.to_device(np.full(B, 0.3)), ctx.to_device(np.zeros(B))
    ctx.gram_stats(reset=True)
    for _ in range(3):
        sol.factor_dev(d["J"], d["f"], d["x"], d["lb"], d["ub"], d["scale"])
    sol.step_dev(dD, dA)
    ctx.sync()
    assert ctx.gram_stats() == (3 * (B - 1), 3)
    sol.close()
    for v in list(d.values()) + [dD, dA]:
        ctx.free(v)
    ctx.close()


def test_optimistic_verdict_of_the_dogbox_device_api(bl, monkeypatch, blsq_opt):
    """blsq_dogbox_factor_dev / blsq_dogbox_step_dev: the same optimistic scheme as TRF — right guess,
    wrong guess (a problem that fails the certificate) and the synchronous mode give the same bits."""
    from bounded_lsq import _synth, _abi
    B, m, n = 6, 700, 64
    good = _synth.dogbox_batch(500, B, m, n)
    mixed = {k: v.copy() for k, v in good.items()}
    mixed["J"][2] = _equicorrelated(1, m, n, 1 - 1e-9, 3)[0]
    mixed["lb"][2] = -np.inf; mixed["ub"][2] = np.inf; mixed["on_bound"][2] = 0
    Delta = np.full(B, 0.05)
    for P, expect_fb in ((good, 0), (mixed, 1)):
        outs = []
        for opt in ("1", "0"):
            blsq_opt("BLSQ_OPTIMISTIC", opt)
            ctx = _abi.Context(0)
            sol = bl.DogboxStepSolver(B, m, n, ctx=ctx)
            d = {k: ctx.to_device(P[k]) for k in ("J", "f", "x", "lb", "ub", "scale", "on_bound")}
            dD = ctx.to_device(Delta)
            ctx.gram_stats(reset=True)
            for _ in range(2):
                sol.factor_dev(d["J"], d["f"], d["x"], d["lb"], d["ub"], d["scale"], d["on_bound"])
                sol.step_dev(dD)
            S = sol.fetch_step()
            assert ctx.gram_stats() == (2 * (B - expect_fb), 2 * expect_fb), opt
            outs.append((S.step.copy(), S.on_bound_new.copy(), S.predicted_reduction.copy()))
            sol.close()
            for v in list(d.values()) + [dD]:
                ctx.free(v)
            ctx.close()
        for x1, x0 in zip(*outs):
            assert np.array_equal(x1, x0)


@pytest.mark.parametrize("solver", ["trf", "dogbox"])
def test_counter_and_vector_routes_of_the_device_api_give_the_same_bits(bl, monkeypatch, solver, blsq_opt):
    """Two routes of the device-resident calls that change no arithmetic: the verdict / Newton-round counters reach
    the host through a one-lane kernel and a polled pinned slot (BLSQ_PUBLISH = 0: hipMemcpyAsync + event), and the
    caller's x / lb / ub / scale / on_bound are packed into the state layout by the prep launch (BLSQ_FUSE_PACK = 0:
    by a launch of their own in front of the Gram); the verdict's counters ride on the step kernel of the next step call
    (BLSQ_PUBLISH_RIDE = 0: a publishing launch of their own at the end of the factor call).  Right and wrong guesses,
    three calls on the same plan, several Newton rounds: same bits, same path statistics on every combination."""
    from bounded_lsq import _synth, _abi
    B, m, n = 6, 700, 64
    if solver == "trf":
        good = _synth.trf_batch(464, B, m, n)
        keys = ("J", "f", "x", "lb", "ub", "scale")
        Delta = np.array([0.5, 0.05, 5.0, 0.2, 0.3, 1.0])
    else:
        good = _synth.dogbox_batch(500, B, m, n)
        keys = ("J", "f", "x", "lb", "ub", "scale", "on_bound")
        Delta = np.full(B, 0.05)
    mixed = {k: v.copy() for k, v in good.items()}
    mixed["J"][2] = _equicorrelated(1, m, n, 1 - 1e-9, 3)[0]            # fails the certificate
    mixed["lb"][2] = -np.inf; mixed["ub"][2] = np.inf
    if solver == "dogbox":
        mixed["on_bound"][2] = 0
    for P, expect_fb in ((good, 0), (mixed, 1)):
        outs = []
        for publish, fuse, ride in (("1", "1", "1"), ("0", "1", "1"), ("1", "0", "1"), ("0", "0", "1"), ("1", "1", "0"),
                                    ("1", "0", "0")):
            blsq_opt("BLSQ_PUBLISH", publish)
            blsq_opt("BLSQ_FUSE_PACK", fuse)
            blsq_opt("BLSQ_PUBLISH_RIDE", ride)
            ctx = _abi.Context(0)
            sol = (bl.TrfStepSolver if solver == "trf" else bl.DogboxStepSolver)(B, m, n, ctx=ctx)
            d = {k: ctx.to_device(P[k]) for k in keys}
            extra = [ctx.to_device(Delta)] + ([ctx.to_device(np.zeros(B))] if solver == "trf" else [])
            ctx.gram_stats(reset=True)
            for _ in range(3):
                sol.factor_dev(*[d[k] for k in keys])
                sol.step_dev(*extra)
            S = sol.fetch_step()
            assert ctx.gram_stats() == (3 * (B - expect_fb), 3 * expect_fb), (publish, fuse, ride)
            if solver == "trf":
                outs.append((S.step.copy(), np.asarray(S.alpha).copy(), np.asarray(S.n_iter).copy(), S.hits.copy(),
                             sol.fetch_factor().g.copy()))
            else:
                outs.append((S.step.copy(), S.on_bound_new.copy(), S.predicted_reduction.copy()))
            sol.close()
            for v in list(d.values()) + extra:
                ctx.free(v)
            ctx.close()
        for o in outs[1:]:
            for x1, x0 in zip(outs[0], o):
                assert np.array_equal(x1, x0)


def test_direct_kernel_wave_counts(bl, monkeypatch, blsq_opt):
    """Narrow problems (at most four column tiles): the direct Gram kernel runs two, four or eight
    waves per workgroup by the row count (about 256 rows per wave; BLSQ_GRAM_DIRECT_NW forces one).
    The wave count fixes the summation order, nothing else: every choice gives g = J^T f to rounding and
    the same step to far below the parity bar — ragged row counts, one or several row chunks."""
    from bounded_lsq import _synth, _abi
    for (B, m, n) in [(5, 512, 64), (3, 300, 40), (4, 1000, 30), (2, 129, 64), (3, 2500, 17), (2, 33, 5)]:
        P = _synth.trf_batch(800 + m, B, m, n)
        Delta = np.full(B, 0.3)
        outs = {}
        for nw in ("default", "2", "4", "8"):
            if nw == "default":
                monkeypatch.delenv("BLSQ_GRAM_DIRECT_NW", raising=False)
            else:
                blsq_opt("BLSQ_GRAM_DIRECT_NW", nw)
            ctx = _abi.Context(0)
            sol = bl.TrfStepSolver(B, m, n, ctx=ctx)
            sol.factor(P["J"], P["f"], P["x"], P["lb"], P["ub"], P["scale"])
            S = sol.step(Delta, np.zeros(B))
            outs[nw] = (sol.fetch_factor().g.copy(), S.step.copy(), S.hits.copy())
            sol.close(); ctx.close()
        want = 8 if m <= 128 else 2 if m <= 512 else 4 if m <= 1024 else 8
        for k in range(3):
            assert np.array_equal(outs["default"][k], outs[str(want)][k]), (m, n, k)
        for nw in ("2", "4", "8"):
            for b in range(B):
                gx = (P["J"][b].astype(np.longdouble).T @ P["f"][b].astype(np.longdouble)).astype(np.float64)
                assert rel(outs[nw][0][b], gx) < 1e-13, (m, n, nw)
                assert rel(outs[nw][1][b], outs["8"][1][b]) < 1e-11, (m, n, nw)
            assert np.array_equal(outs[nw][2], outs["8"][2]), (m, n, nw)


def test_zeros_outside_the_factor_are_stored_only_when_needed(bl):
    """N > 80: the Cholesky of the augmented system does not store the zeros outside the triangle while
    the plan knows they are still there (only the stacked QR and the Jacobi SVD write into that part of
    the slots); the Newton-round factors never store them.  A plan that goes clean -> dirty (a problem for
    the Householder tree, a rank-deficient one for the SVD) -> clean again gives, call by call, the bits of
    a fresh plan."""
    from bounded_lsq import _synth
    B, m, n = 5, 900, 130
    good = _synth.trf_batch(530, B, m, n)
    other = _synth.trf_batch(531, B, m, n)
    hard = {k: v.copy() for k, v in good.items()}
    hard["J"][1] = _equicorrelated(1, m, n, 1 - 1e-9, 3)[0]
    hard["lb"][1] = -np.inf; hard["ub"][1] = np.inf
    hard["J"][3][:, n - 1] = hard["J"][3][:, 0]
    hard["lb"][3] = -np.inf; hard["ub"][3] = np.inf
    Delta = np.array([0.5, 0.05, 5.0, 0.2, 0.3])
    sol = bl.TrfStepSolver(B, m, n)
    for call, P in enumerate([good, other, hard, good, other, hard, hard, good, good]):
        outs = []
        for s_ in (sol, bl.TrfStepSolver(B, m, n)):
            s_.factor(P["J"], P["f"], P["x"], P["lb"], P["ub"], P["scale"])
            S = s_.step(Delta, np.zeros(B))
            outs.append((S.step.copy(), np.asarray(S.alpha).copy(), np.asarray(S.n_iter).copy(), S.hits.copy()))
            if s_ is not sol:
                s_.close()
        for x1, x0 in zip(*outs):
            assert np.array_equal(x1, x0, equal_nan=True), call
    sol.close()


def test_round_loop_follows_the_round_count_of_the_last_call(bl):
    """N > 80: the Newton rounds are enqueued ahead of their counters only as far as the plan's LAST step
    call had work; beyond that the host looks first.  Step calls whose round counts go 0 -> several ->
    1 -> 0 -> several on one plan (trust radii from 'Gauss-Newton step inside' to 'far outside') give the
    bits of a fresh plan, which runs ahead through every round."""
    from bounded_lsq import _synth
    B, m, n = 6, 900, 130
    P = _synth.trf_batch(540, B, m, n)
    sol = bl.TrfStepSolver(B, m, n)
    sol.factor(P["J"], P["f"], P["x"], P["lb"], P["ub"], P["scale"])
    seen = set()
    for call, dl in enumerate([1e6, 1e-3, 0.3, 1e6, 1e-6, 0.05, 1e6, 1e6, 1e-2]):
        Delta = np.full(B, dl)
        S = sol.step(Delta, np.zeros(B))
        ref = bl.TrfStepSolver(B, m, n)
        ref.factor(P["J"], P["f"], P["x"], P["lb"], P["ub"], P["scale"])
        R = ref.step(Delta, np.zeros(B))
        ref.close()
        for x1, x0 in ((S.step, R.step), (np.asarray(S.alpha), np.asarray(R.alpha)),
                       (np.asarray(S.n_iter), np.asarray(R.n_iter)), (S.hits, R.hits)):
            assert np.array_equal(x1, x0, equal_nan=True), call
        seen.add(int(np.max(np.asarray(S.n_iter))))
    sol.close()
    assert 0 in seen and max(seen) >= 2, seen          # (the sequence really had calls without and with rounds)


@pytest.mark.parametrize("shape", [(700, 64), (900, 150)])
def test_second_guess_of_the_trf_device_api(bl, monkeypatch, shape, blsq_opt):
    """TRF: after a call in which the Cholesky kernel (N <= 80; N > 80: stage 0 of the certificate, which is still
    launched) settled every problem (first certificate bound + the rank gate's column-norm bound), the remaining
    certificate and gate launches of the next call are not enqueued.  Guess holds / fails softly (a column of norm
    1e-12) / fails hard (certificate; rank deficiency) / holds again: the bits of the synchronous mode at every call
    (and, N > 80, of BLSQ_SETTLE0 = 0, which always enqueues the whole tail)."""
    from bounded_lsq import _synth, _abi
    m, n = shape
    B = 6
    good = _synth.trf_batch(520, B, m, n)
    other = _synth.trf_batch(521, B, m, n)
    tiny = {k: v.copy() for k, v in good.items()}
    tiny["J"][4][:, 7] *= 1e-12
    tiny["lb"][4] = -np.inf; tiny["ub"][4] = np.inf
    hard = {k: v.copy() for k, v in good.items()}
    hard["J"][1] = _equicorrelated(1, m, n, 1 - 1e-9, 3)[0]
    hard["lb"][1] = -np.inf; hard["ub"][1] = np.inf
    hard["J"][3][:, n - 1] = hard["J"][3][:, 0]
    hard["lb"][3] = -np.inf; hard["ub"][3] = np.inf
    seq = [good, other, good, tiny, other, good, hard, good, other, good]
    Delta = np.array([0.5, 0.05, 5.0, 0.2, 0.3, 1.0])
    for scale_mode in (0, 1):
        runs = []
        for opt, settle in (("1", "1"), ("0", "1"), ("1", "0")):
            blsq_opt("BLSQ_OPTIMISTIC", opt)
            blsq_opt("BLSQ_SETTLE0", settle)
            ctx = _abi.Context(0)
            sol = bl.TrfStepSolver(B, m, n, ctx=ctx)
            dD, dA = ctx.to_device(Delta), ctx.to_device(np.zeros(B))
            outs = []
            for P in seq:
                d = {k: ctx.to_device(P[k]) for k in ("J", "f", "x", "lb", "ub", "scale")}
                sol.factor_dev(d["J"], d["f"], d["x"], d["lb"], d["ub"], d["scale"], scale_mode)
                sol.step_dev(dD, dA)
                S = sol.fetch_step()
                outs.append((S.step.copy(), np.asarray(S.alpha).copy(), np.asarray(S.n_iter).copy(), S.hits.copy(),
                             sol.fetch_factor().g.copy(), ctx.to_host(d["scale"], (B, n), np.float64)))
                for v in d.values():
                    ctx.free(v)
            runs.append(outs)
            sol.close(); ctx.free(dD); ctx.free(dA); ctx.close()
        for other_run in runs[1:]:
            for call, (o1, o0) in enumerate(zip(runs[0], other_run)):
                for x1, x0 in zip(o1, o0):
                    assert np.array_equal(x1, x0, equal_nan=True), (scale_mode, call)


def test_second_guess_of_the_dogbox_device_api(bl, monkeypatch, blsq_opt):
    """N <= 80: once a call has seen EVERY problem settled inside the Cholesky kernel (certified by the
    first bound, Cauchy and Newton steps written there), the next call does not enqueue the certificate /
    gate / solve launches at all and checks the settled counter when it resolves.  A sequence of calls
    in which that guess holds, fails softly (a column of norm 1e-12: the rank gate must look at the
    problem, nothing leaves the path), fails hard (a problem for the Householder tree) and holds again
    gives the bits of the synchronous mode at every call."""
    from bounded_lsq import _synth, _abi
    B, m, n = 6, 700, 64
    good = _synth.dogbox_batch(510, B, m, n)
    other = _synth.dogbox_batch(511, B, m, n)
    tiny = {k: v.copy() for k, v in good.items()}
    tiny["J"][4][:, 7] *= 1e-12
    tiny["lb"][4] = -np.inf; tiny["ub"][4] = np.inf; tiny["on_bound"][4] = 0
    hard = {k: v.copy() for k, v in good.items()}
    hard["J"][2] = _equicorrelated(1, m, n, 1 - 1e-9, 3)[0]
    hard["lb"][2] = -np.inf; hard["ub"][2] = np.inf; hard["on_bound"][2] = 0
    seq = [good, other, good, tiny, other, good, hard, good, other, good]
    Delta = np.full(B, 0.05)
    runs = []
    for opt in ("1", "0"):
        blsq_opt("BLSQ_OPTIMISTIC", opt)
        ctx = _abi.Context(0)
        sol = bl.DogboxStepSolver(B, m, n, ctx=ctx)
        dD = ctx.to_device(Delta)
        outs = []
        for P in seq:
            d = {k: ctx.to_device(P[k]) for k in ("J", "f", "x", "lb", "ub", "scale", "on_bound")}
            sol.factor_dev(d["J"], d["f"], d["x"], d["lb"], d["ub"], d["scale"], d["on_bound"])
            sol.step_dev(dD)
            S = sol.fetch_step()
            _, nw, ca = sol.fetch_factor(want_steps=True)
            outs.append((S.step.copy(), S.on_bound_new.copy(), S.predicted_reduction.copy(), nw.copy(), ca.copy()))
            for v in d.values():
                ctx.free(v)
        runs.append(outs)
        sol.close(); ctx.free(dD); ctx.close()
    for call, (o1, o0) in enumerate(zip(*runs)):
        for x1, x0 in zip(o1, o0):
            assert np.array_equal(x1, x0, equal_nan=True), call


def test_chunk_pairs_summed_in_the_kernel_match_the_reduction_pass(bl, monkeypatch, blsq_opt):
    """Two row chunks (2048 < m <= 4096), 16 column tiles and at least 256 problems: one workgroup
    takes both chunks and adds them in the kernel, (0 + P0) + P1 — what the separate reduction pass
    computes from the two partial Grams (BLSQ_GRAM_PAIR = 0 keeps that pass).  Bit for bit, so a
    problem's result still does not depend on the size of its batch."""
    from bounded_lsq import _synth, _abi
    for (B, m, n) in [(256, 2100, 256), (257, 2049, 250), (256, 4096, 241)]:
        base = _synth.trf_batch(91 + n, 8, m, n)
        P = {k: np.concatenate([v] * ((B + 7) // 8))[:B].copy() for k, v in base.items()}
        P["f"] = P["f"] * (1.0 + 0.01 * np.arange(B))[:, None]        # (not 32 copies of 8 problems)
        Delta = np.full(B, 0.7)
        outs = []
        for pair in ("1", "0"):
            blsq_opt("BLSQ_GRAM_PAIR", pair)
            ctx = _abi.Context(0)
            sol = bl.TrfStepSolver(B, m, n, ctx=ctx)
            ctx.gram_stats(reset=True)
            sol.factor(P["J"], P["f"], P["x"], P["lb"], P["ub"], P["scale"])
            assert ctx.gram_stats() == (B, 0)
            S = sol.step(Delta, np.zeros(B))
            outs.append((S.step.copy(), sol.fetch_factor().g.copy()))
            sol.close(); ctx.close()
        assert np.array_equal(outs[0][1], outs[1][1])
        assert np.array_equal(outs[0][0], outs[1][0])
        for b in (0, B - 1):
            assert rel(outs[0][1][b], P["J"][b].T @ P["f"][b]) < 1e-13


def test_both_cholesky_kernels_agree_bit_for_bit(bl, monkeypatch, blsq_opt):
    """N > 80: launches of at most 256 problems use the right-looking register kernel (flag-driven schedule;
    BLSQ_CHOL_RL2 = 0: the barrier-synchronous one), larger ones the
    left-looking one (BLSQ_CHOL_RL forces either).  All apply the same operands in the same order —
    the right-looking kernel publishes the STORED entry times its equilibration, exactly what the
    left-looking one reads back — so which one ran (i.e. how many problems shared the launch, or were
    still active in a Newton round) never shows in the results."""
    from bounded_lsq import _synth, _abi
    for (B, m, n, kind) in [(4, 1000, 200, "trf"), (3, 2100, 128, "trf"), (2, 4096, 256, "trf"),
                            (4, 600, 100, "trf"), (4, 900, 120, "dogbox")]:
        P = _synth.trf_batch(70 + n, B, m, n) if kind == "trf" else _synth.dogbox_batch(70 + n, B, m, n)
        outs = []
        for rl, rl2 in (("0", "1"), ("1", "1"), ("1", "0")):       # left-looking, flag-driven, barrier-synchronous
            blsq_opt("BLSQ_CHOL_RL", rl)
            blsq_opt("BLSQ_CHOL_RL2", rl2)
            ctx = _abi.Context(0)
            got = []
            if kind == "trf":
                sol = bl.TrfStepSolver(B, m, n, ctx=ctx)
                sol.factor(P["J"], P["f"], P["x"], P["lb"], P["ub"], P["scale"])
                for D in (0.7, 0.05, 5.0):
                    S = sol.step(np.full(B, D), np.zeros(B))
                    got += [S.step.copy(), np.asarray(S.alpha).copy(), S.predicted_reduction.copy()]
            else:
                sol = bl.DogboxStepSolver(B, m, n, ctx=ctx)
                sol.factor(P["J"], P["f"], P["x"], P["lb"], P["ub"], P["scale"], P["on_bound"])
                for D in (0.7, 0.05):
                    S = sol.step(np.full(B, D))
                    got += [S.step.copy(), S.predicted_reduction.copy()]
            sol.close(); ctx.close()
            outs.append(got)
        for x0, x1, x2 in zip(*outs):
            assert np.array_equal(x0, x1) and np.array_equal(x0, x2)


def test_k_split_kernel_for_eight_column_tiles(bl, monkeypatch, blsq_opt):
    """n = 113 .. 128 (8 column tiles) runs gram8_kernel for EVERY batch size: static tile rows per
    wave, the k-steps of a row chunk split between two wave groups whose partial tiles are added in
    a fixed order.  That order differs from the generic kernel's (BLSQ_GRAM8 = 0), so the two agree
    to rounding only; what must hold bit for bit is that a problem's result does not depend on the
    batch it is in, nor on a tile-group request."""
    from bounded_lsq import _synth, _abi
    for (B, m, n) in [(5, 2100, 128), (4, 1000, 113), (3, 4096, 127), (3, 700, 120), (2, 33, 128)]:
        P = _synth.trf_batch(61 + n, B, m, n, unbounded=(m < 200))
        Delta = np.full(B, 0.7)

        def run(idx, g8="1", tg=None):
            blsq_opt("BLSQ_GRAM8", g8)
            if tg: blsq_opt("BLSQ_GRAM_TILE_GROUPS", tg)
            else: monkeypatch.delenv("BLSQ_GRAM_TILE_GROUPS", raising=False)
            ctx = _abi.Context(0)
            sol = bl.TrfStepSolver(len(idx), m, n, ctx=ctx)
            sol.factor(*(P[k][idx] for k in ("J", "f", "x", "lb", "ub", "scale")))
            S = sol.step(Delta[idx], np.zeros(len(idx)))
            out = (S.step.copy(), sol.fetch_factor().g.copy())
            sol.close(); ctx.close()
            return out

        full = run(np.arange(B))
        alone = run(np.array([B - 1]))
        assert np.array_equal(full[0][B - 1], alone[0][0]) and np.array_equal(full[1][B - 1], alone[1][0])
        grouped = run(np.arange(B), tg="4")
        assert np.array_equal(full[0], grouped[0]) and np.array_equal(full[1], grouped[1])
        generic = run(np.arange(B), g8="0")
        for b in range(B):
            gref = P["J"][b].T @ P["f"][b]
            assert rel(full[1][b], gref) < 1e-13
            assert rel(full[0][b], generic[0][b]) < 1e-11


@pytest.mark.parametrize("rho", [0.99, 0.9999, 1 - 1e-8])
def test_bounded_problems_are_gated_on_the_augmented_system(bl, rho):
    """Bounds close to x: the Coleman-Li block E^2 = diag(g jv scale^2) dominates H = D G D + E^2
    (D = diag(sqrt(v) scale), v = distance to the bound ~ 0.02), so H is well conditioned whatever
    J is — the reference's own SVD sees the same augmented matrix.  Such problems stay on the
    normal-equations path and still match the oracle to 1e-10 (asserted inside _check)."""
    from bounded_lsq import _synth
    B, m, n = 3, 2048, 64
    P = _synth.trf_batch(33, B, m, n)
    P["J"] = _equicorrelated(B, m, n, rho, 6)
    stats, worst = _check(bl, P, np.array([10.0, 0.5, 2.0]))
    assert stats == (B, 0), stats
    assert worst < 1e-11


def test_stage_zero_of_the_certificate_changes_no_verdict(bl, monkeypatch, blsq_opt):
    """Stage 0 (comparison-matrix bound: two triangular solves instead of the explicit inverse) only SETTLES
    well-conditioned problems early; with it switched off (BLSQ_CERT0 = 0) the later stages reach the same
    verdicts, and every step comes out bit for bit the same.  Its bound is a bound: K2 >= the true kappa_2."""
    from bounded_lsq import _synth, _abi
    B, m, n = 8, 2048, 160
    P = _synth.trf_batch(91, B, m, n, unbounded=True)
    rho = np.array([0.0, 0.0, 0.3, 0.9, 0.99, 0.9995, 0.0, 1 - 1e-9])
    Jc = _equicorrelated(B, m, n, rho, 13)
    for b in range(B):
        if rho[b] > 0:
            P["J"][b] = Jc[b]
    Delta = np.full(B, 0.7)
    outs = []
    for flag in ("1", "0"):
        blsq_opt("BLSQ_CERT0", flag)
        ctx = _abi.Context(0)
        sol = bl.TrfStepSolver(B, m, n, ctx=ctx)
        ctx.gram_stats(reset=True)
        sol.factor(P["J"], P["f"], P["x"], P["lb"], P["ub"], P["scale"])
        stats = ctx.gram_stats()
        k2 = sol.debug_cond()
        S = sol.step(Delta, np.zeros(B))
        outs.append((stats, k2.copy(), S.step.copy(), np.asarray(S.n_iter).copy()))
        sol.close(); ctx.close()
    (s1, k1, st1, it1), (s0, k0, st0, it0) = outs
    assert s1 == s0 and s1[1] >= 1 and s1[0] >= 5
    assert np.array_equal(st1, st0) and np.array_equal(it1, it0)
    assert np.any(k1 != k0)                                   # (stage 0 settled some problems with its own bound)
    _certificate_holds(P, k1, s1)
    _certificate_holds(P, k0, s0)


def test_flag_driven_cholesky_is_race_free_under_repetition(bl, monkeypatch, blsq_opt):
    """The right-looking Cholesky kernel hands tiles between its waves through LDS flags and counters, without
    workgroup barriers.  A missed hand-over would show as different bits: every shape (6 ... 17 tile columns, one
    problem ... more problems than CUs, gathered sub-matrices of dogbox) is factored 25 times and must reproduce
    the left-looking kernel's result bit for bit every time."""
    from bounded_lsq import _synth, _abi
    for (B, m, n, kind) in [(1, 300, 81, "trf"), (300, 300, 96, "trf"), (7, 500, 150, "trf"), (520, 260, 256, "trf"),
                            (3, 400, 271, "trf"), (40, 400, 130, "dogbox")]:
        P = _synth.trf_batch(170 + n, B, m, n) if kind == "trf" else _synth.dogbox_batch(170 + n, B, m, n)
        ref = None
        for rep, rl in enumerate(["0"] + ["1"] * 25):
            blsq_opt("BLSQ_CHOL_RL", rl)
            ctx = _abi.Context(0)
            if kind == "trf":
                sol = bl.TrfStepSolver(B, m, n, ctx=ctx)
                sol.factor(P["J"], P["f"], P["x"], P["lb"], P["ub"], P["scale"])
                S = sol.step(np.full(B, 0.3), np.zeros(B))
                got = (S.step.copy(), np.asarray(S.alpha).copy())
            else:
                sol = bl.DogboxStepSolver(B, m, n, ctx=ctx)
                sol.factor(P["J"], P["f"], P["x"], P["lb"], P["ub"], P["scale"], P["on_bound"])
                S = sol.step(np.full(B, 0.3))
                got = (S.step.copy(), S.predicted_reduction.copy())
            sol.close(); ctx.close()
            if ref is None:
                ref = got
            else:
                assert all(np.array_equal(a, b) for a, b in zip(ref, got)), (B, m, n, kind, rep)


def test_open_jacobian_systems_go_to_the_third_stage_directly(bl, monkeypatch, blsq_opt):
    """A problem stage 0 cannot settle whose system is J^T J itself (no Coleman-Li block) skips the explicit inverse
    of the norm stage and is decided by the shifted factorisation (GramCholArgs::cert_open; BLSQ_CERT_DIRECT = 0:
    everybody through the norm stage).  Both routes PROVE what they decide, so paths and steps agree wherever both
    settle a problem the same way — here: true kappa_2 well inside / well beyond the gate — and every reported bound
    is one; a hopeless problem (a pivot beyond the gate) is rejected without any further stage."""
    from bounded_lsq import _synth, _abi
    B, m, n = 10, 1500, 130
    P = _synth.trf_batch(141, B, m, n, unbounded=True)
    rng = np.random.default_rng(5)
    kap = [1.0, 30.0, 100.0, 200.0, 3e3, 1e5, 1e7, 150.0, 60.0, 2e4]      # kappa(J): gate near 480
    for b in range(B):
        U, _ = np.linalg.qr(rng.standard_normal((m, n)))
        V, _ = np.linalg.qr(rng.standard_normal((n, n)))
        P["J"][b] = (U * np.logspace(0, -np.log10(kap[b]), n)) @ V.T * np.sqrt(m)
    P["J"][6][:, 7] = P["J"][6][:, 3] * (1 + 1e-9)                          # two columns nearly equal: a tiny pivot
    outs = []
    for flag in ("1", "0"):
        blsq_opt("BLSQ_CERT_DIRECT", flag)
        ctx = _abi.Context(0)
        sol = bl.TrfStepSolver(B, m, n, ctx=ctx)
        ctx.gram_stats(reset=True)
        sol.factor(P["J"], P["f"], P["x"], P["lb"], P["ub"], P["scale"])
        stats = ctx.gram_stats()
        k2 = sol.debug_cond().copy()
        S = sol.step(np.full(B, 0.7), np.zeros(B))
        outs.append((stats, k2, S.step.copy()))
        sol.close(); ctx.close()
    (s1, k1, st1), (s0, k0, st0) = outs
    assert s1 == s0 and s1[0] >= 5 and s1[1] >= 3, (s1, s0)               # the same paths either way
    for b in range(B):
        assert np.linalg.norm(st1[b] - st0[b]) <= 1e-10 * np.linalg.norm(st0[b])
    _certificate_holds(P, k1, s1)
    _certificate_holds(P, k0, s0)
