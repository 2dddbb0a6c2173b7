"""GPU parity: the HIP path, called through the C-ABI, against
  (1) golden vectors captured from the reference (tests/golden/),
  (2) the CPU oracle on seeded batches,
  (3) size-independent properties at BASELINE.json's full sizes.
Tolerances (north_star): step vector within 1e-10 relative of the reference
CPU path; masks bit-exact."""
import numpy as np
import pytest

from _golden import load_npz, trf_inputs, dog_inputs

pytestmark = pytest.mark.gpu

RTOL = 1e-10

# Fixtures whose REFERENCE answer is set by rounding noise (see
# tests/test_oracle_golden.py::test_rankdef_fixture_is_noise_determined): a
# rank-deficient unbounded J with Delta beyond the min-norm Gauss-Newton step
# drives alpha -> 1e-17, where the null-space outputs of LAPACK's SVD (s ~ 1e-16,
# arbitrary uf) decide p.  No independent factorisation can match those to
# 1e-10; they are checked through properties instead.
KNIFE_EDGE = {"rankdef_32x8"}


def rel(a, b):
    a = np.asarray(a, float); b = np.asarray(b, float)
    den = np.linalg.norm(b)
    return np.linalg.norm(a - b) / (den if den > 0 else 1.0)


@pytest.fixture(scope="module")
def bl():
    import bounded_lsq
    return bounded_lsq


TRF_CASES = (load_npz("trf_small.npz") + load_npz("trf_large.npz") +
             load_npz("trf_choice2.npz"))    # find_gradient_step wins (trf.py:159-170, choice 2)


@pytest.fixture(params=["gram_front_end", "qr_tree_only"], autouse=True)
def fact_path(request, monkeypatch, blsq_opt):
    """Every test of this file runs twice: with the normal-equations front end of the
    factorisation (Gram + gated Cholesky; ill-conditioned problems still reach the Householder
    tree through its gate) and with the front end switched off, so the TSQR tree keeps its full
    coverage."""
    blsq_opt("BLSQ_GRAM", "1" if request.param == "gram_front_end" else "0")
    return request.param


@pytest.fixture(params=["svd_free", "svd_only"])
def tr_path(request, monkeypatch, blsq_opt):
    """Both trust-region paths: the SVD-free one (QR of [R; sqrt(alpha) I], taken when the
    full-rank gate passes) and the Jacobi-SVD one forced for every problem."""
    blsq_opt("BLSQ_NO_SVDFREE", "1" if request.param == "svd_only" else "0")
    blsq_opt("BLSQ_SVDFREE_MIN_N", "0")      # let small fixtures take the SVD-free path too
    return request.param


@pytest.mark.parametrize("name,ins,out", TRF_CASES, ids=[c[0] for c in TRF_CASES])
def test_trf_golden(bl, tr_path, name, ins, out):
    P = trf_inputs(ins)
    m, n = P["J"].shape
    sol = bl.TrfStepSolver(1, m, n)
    sol.factor_dev  # noqa: B018 (API exists)
    F = sol.factor(P["J"][None], P["f"][None], P["x"][None], P["lb"][None], P["ub"][None],
                   P["scale"][None])
    assert rel(F.g[0], out["g"]) < 1e-12
    assert abs(F.g_norm[0] - float(out["g_norm"])) <= 1e-12 * max(1.0, float(out["g_norm"]))
    assert abs(F.theta[0] - float(out["theta"])) <= 1e-12
    sol.lib.blsq_trf_step_dev  # noqa: B018
    S = sol.step(np.array([P["Delta"]]), np.array([P["alpha0"]]))
    D = sol.fetch_step()
    _, sing = sol.fetch_factor(want_singular=True)
    sref = np.asarray(out["s"], float)
    used_fast = int(sol.debug_fast()[0])
    if tr_path == "svd_only":
        assert used_fast == 0
    if not used_fast:                       # singular values exist only on the SVD path
        assert rel(np.sort(sing[0])[::-1], sref) < 1e-12
    else:                                   # the gate must never pass a rank-deficient problem
        assert m >= n and sref[-1] > 1e3 * np.finfo(float).eps * m * sref[0]
    if name in KNIFE_EDGE:
        Delta = P["Delta"]
        assert abs(np.linalg.norm(D.p_h_tr[0]) - Delta) <= 0.011 * Delta
        pr = float(out["predicted_reduction"])
        assert abs(S.predicted_reduction[0] - pr) <= 1e-3 * abs(pr)
        assert int(S.branch[0]) == int(out["branch"]) and int(S.status[0]) == 0
        sol.close()
        return
    assert int(S.n_iter[0]) == int(out["n_iter"]), "More' iteration count"
    assert int(S.branch[0]) == int(out["branch"])
    assert int(D.choice[0]) == int(out["choice"])
    assert rel(D.p_h_tr[0], out["p_h_tr"]) < RTOL
    a_ref = float(out["alpha"])
    assert abs(S.alpha[0] - a_ref) <= 1e-9 * max(abs(a_ref), 1e-300)
    np.testing.assert_array_equal(S.hits[0], out["hits"])           # bit-exact mask
    assert rel(S.step_h[0], out["step_h"]) < RTOL
    assert rel(S.step[0], out["step"]) < RTOL
    assert rel(S.x_new[0], out["x_new"]) < RTOL
    np.testing.assert_array_equal(S.active_new[0], out["active_new"])  # bit-exact mask
    pr = float(out["predicted_reduction"])
    assert abs(S.predicted_reduction[0] - pr) <= 1e-10 * abs(pr)
    assert abs(S.step_h_norm[0] - float(out["step_h_norm"])) <= 1e-10 * float(out["step_h_norm"])
    assert abs(S.correction[0] - float(out["correction"])) <= 1e-10 * max(
        abs(float(out["correction"])), 1e-300)
    # x_new must be strictly feasible (make_strictly_feasible)
    assert np.all(S.x_new[0] > P["lb"]) and np.all(S.x_new[0] < P["ub"])
    sol.close()


DOG_CASES = (load_npz("dog_small.npz") + load_npz("dog_large.npz") +
             load_npz("dog_fallback.npz"))   # dogbox.py:211-216 taken (fallback = 1) + near misses


@pytest.mark.parametrize("name,ins,out", DOG_CASES, ids=[c[0] for c in DOG_CASES])
def test_dogbox_golden(bl, tr_path, name, ins, out):
    P = dog_inputs(ins)
    m, n = P["J"].shape
    sol = bl.DogboxStepSolver(1, m, n)
    F = sol.factor(P["J"][None], P["f"][None], P["x"][None], P["lb"][None], P["ub"][None],
                   P["scale"][None], P["on_bound"][None])
    assert rel(F.g[0], out["g"]) < 1e-12
    np.testing.assert_array_equal(F.active_set[0], out["active_set"])
    assert abs(F.g_norm[0] - float(out["g_norm"])) <= 1e-12 * max(1.0, float(out["g_norm"]))
    _, nw, ca = sol.fetch_factor(want_steps=True)
    assert rel(nw[0], out["newton_full"]) < RTOL
    assert rel(ca[0], out["cauchy_full"]) < RTOL
    S = sol.step(np.array([P["Delta"]]))
    assert rel(S.step[0], out["step"]) < RTOL
    assert rel(S.x_new[0], out["x_new"]) < RTOL
    np.testing.assert_array_equal(S.on_bound_new[0], out["on_bound_new"])  # bit-exact mask
    assert int(S.tr_hit[0]) == int(out["tr_hit"])
    assert int(S.fallback[0]) == int(out["fallback"])
    pr = float(out["predicted_reduction"])
    assert abs(S.predicted_reduction[0] - pr) <= 1e-10 * abs(pr)
    ssn = float(out["step_scaled_norm"])
    assert abs(S.step_scaled_norm[0] - ssn) <= 1e-10 * ssn
    sol.close()


@pytest.mark.parametrize("B,m,n", [(16, 512, 64), (5, 200, 37), (3, 1500, 20), (2, 4096, 256),
                                   (2, 2500, 50), (3, 1100, 17), (4, 700, 100), (2, 5000, 33),
                                   (2, 3000, 200), (6, 90, 5), (2, 1089, 16),
                                   # few live rows in the late panels (Gram of < 16 rows: the
                                   # Cholesky-QR panel path must fall back), square and tall-thin
                                   (3, 64, 64), (3, 70, 64), (2, 300, 256), (2, 1025, 32),
                                   (2, 2049, 48)])
def test_trf_batch_vs_oracle(bl, tr_path, B, m, n):
    from oracle import blsq_oracle as orc
    from bounded_lsq import _synth
    P = _synth.trf_batch(1000 + n, B, m, n)
    Delta = np.where(np.arange(B) % 2 == 0, 10.0, 0.5)
    sol = bl.TrfStepSolver(B, m, n)
    sol.factor(P["J"], P["f"], P["x"], P["lb"], P["ub"], P["scale"])
    S = sol.step(Delta, np.zeros(B))
    nb = 0
    for b in range(B):
        _, So = orc.trf_step_solve(P["J"][b], P["f"][b], P["x"][b], P["lb"][b], P["ub"][b],
                                   P["scale"][b], Delta[b], 0.0)
        assert int(S.n_iter[b]) == So.n_iter and int(S.branch[b]) == So.branch
        assert rel(S.step[b], So.step) < RTOL, (b, rel(S.step[b], So.step))
        np.testing.assert_array_equal(S.hits[b], So.hits)
        nb += So.branch
    sol.close()


@pytest.mark.parametrize("B,m,n", [(16, 512, 64), (5, 200, 37), (2, 4096, 256), (3, 1500, 20),
                                   (2, 2500, 50), (4, 700, 100), (6, 90, 5)])
def test_dogbox_batch_vs_oracle(bl, tr_path, B, m, n):
    from oracle import blsq_oracle as orc
    from bounded_lsq import _synth
    P = _synth.dogbox_batch(2000 + n, B, m, n)
    Delta = np.where(np.arange(B) % 2 == 0, 0.02, 0.005)
    sol = bl.DogboxStepSolver(B, m, n)
    sol.factor(P["J"], P["f"], P["x"], P["lb"], P["ub"], P["scale"], P["on_bound"])
    S = sol.step(Delta)
    for b in range(B):
        _, So = orc.dogbox_step_solve(P["J"][b], P["f"][b], P["x"][b], P["lb"][b], P["ub"][b],
                                      P["scale"][b], P["on_bound"][b], Delta[b])
        assert rel(S.step[b], So.step) < RTOL, (b, rel(S.step[b], So.step))
        np.testing.assert_array_equal(S.on_bound_new[b], So.on_bound_new)
        assert int(S.tr_hit[b]) == int(So.tr_hit)
    sol.close()


def test_jac_scaling_modes(bl):
    """'jac' scaling (trf.py:216-219,239-242): column norms come from R."""
    from bounded_lsq import _synth, SCALE_JAC_INIT, SCALE_JAC_UPDATE
    B, m, n = 3, 120, 10
    P = _synth.trf_batch(77, B, m, n)
    P["J"][1, :, 3] = 0.0                      # zero column -> scale 1 at init
    sol = bl.TrfStepSolver(B, m, n)
    F = sol.factor(P["J"], P["f"], P["x"], P["lb"], P["ub"], np.ones((B, n)), SCALE_JAC_INIT)
    ref = np.linalg.norm(P["J"], axis=1)
    ref[ref == 0] = 1.0
    np.testing.assert_allclose(F.scale, 1.0 / ref, rtol=1e-13)
    big = np.full((B, n), 1e-3)                # min(scale, 1/norm) keeps the smaller
    F2 = sol.factor(P["J"], P["f"], P["x"], P["lb"], P["ub"], big, SCALE_JAC_UPDATE)
    with np.errstate(divide="ignore"):
        np.testing.assert_allclose(F2.scale, np.minimum(big, 1.0 / np.linalg.norm(P["J"], axis=1)),
                                   rtol=1e-13)
    sol.close()


def test_step_is_repeatable_without_refactor(bl):
    """step() with a new Delta must not disturb the factor state (trf.py:283-285)."""
    from bounded_lsq import _synth
    B, m, n = 4, 300, 24
    P = _synth.trf_batch(5, B, m, n)
    sol = bl.TrfStepSolver(B, m, n)
    sol.factor(P["J"], P["f"], P["x"], P["lb"], P["ub"], P["scale"])
    S1 = sol.step(np.full(B, 0.7), np.zeros(B))
    sol.step(np.full(B, 0.05), S1.alpha.copy())
    S3 = sol.step(np.full(B, 0.7), np.zeros(B))
    np.testing.assert_array_equal(S1.step, S3.step)
    np.testing.assert_array_equal(S1.x_new, S3.x_new)
    sol.close()


def test_invalid_arguments_are_reported(bl):
    from bounded_lsq import _abi
    with pytest.raises(_abi.BlsqError):
        bl.TrfStepSolver(0, 10, 2)
    with pytest.raises(_abi.BlsqError):
        bl.TrfStepSolver(1, 10, 2000)          # n + 1 > 1088
    with pytest.raises(_abi.BlsqError):
        bl.TrfStepSolver(1, 5000, 600)         # tall needs n + 1 <= 544


@pytest.mark.parametrize("m,n,nranks", [(6000, 40, 4), (3000, 128, 3), (20000, 16, 8),
                                        (6001, 40, 4), (1003, 24, 3)])      # unequal row blocks
def test_tsqr_row_blocks_single_gpu(bl, m, n, nranks):
    """Row-block TSQR, Householder route, rehearsed on ONE GPU: every 'rank'
    factors its row block, the triangles are stacked in rank order (what the all-gather inside
    blsq_tsqr_factor_dev produces) and merged; the step must match the oracle on the full
    problem.  Every plan gets the GLOBAL row count (it enters the reference's rank test)."""
    from oracle import blsq_oracle as orc
    from bounded_lsq import _synth, _abi
    from bounded_lsq._multi import TsqrTrfSolver, row_block, tri_ld
    P = _synth.trf_problem(4242 + n, m, n)
    ctx = bl._hip_step.default_context()
    ld = tri_ld(n)
    dstack = ctx.malloc(8 * nranks * ld * ld)
    dvec = {k: ctx.to_device(P[k]) for k in ("x", "lb", "ub", "scale")}
    sols = []
    for r in range(nranks):
        lo, hi = row_block(m, nranks, r)
        sol = TsqrTrfSolver(hi - lo, n, nranks, r, ctx=ctx, m_total=m)
        dJ = ctx.to_device(P["J"][lo:hi]); df = ctx.to_device(P["f"][lo:hi])
        sol.local_triangle_dev(dJ, df, _abi.vp(dstack.value + 8 * r * ld * ld))
        ctx.sync()
        ctx.free(dJ); ctx.free(df)
        sols.append(sol)
    Delta = 0.7
    _, So = orc.trf_step_solve(P["J"], P["f"], P["x"], P["lb"], P["ub"], P["scale"], Delta, 0.0)
    for sol in sols[:2]:                        # every rank gets the same answer
        sol.combine_dev(dstack, dvec["x"], dvec["lb"], dvec["ub"], dvec["scale"])
        S = sol.step(np.array([Delta]), np.array([0.0]))
        assert rel(S.step[0], So.step) < RTOL
        np.testing.assert_array_equal(S.hits[0], So.hits)
        assert int(S.n_iter[0]) == So.n_iter
    for sol in sols:
        sol.close()
    for d in list(dvec.values()) + [dstack]:
        ctx.free(d)


@pytest.mark.parametrize("m,n", [(6000, 40), (3000, 128), (20000, 16), (5000, 256)])
def test_tall_problem_factor_through_the_librarys_own_collective(bl, m, n, fact_path):
    """blsq_tsqr_factor_dev on a real RCCL communicator (one rank: all a one-GPU box allows — RCCL
    refuses two ranks on one device): rendezvous id, ncclCommInitRank, the Gram all-reduce (or, with
    the front end off / the gate failing, the triangle all-gather) enqueued on the library's stream,
    replicated factorisation; the step must match the oracle."""
    from oracle import blsq_oracle as orc
    from bounded_lsq import _synth, _abi
    from bounded_lsq._multi import TsqrTrfSolver, exchange_id_tcp
    P = _synth.trf_problem(777 + n, m, n)
    ctx = _abi.Context(0)
    comm_id = exchange_id_tcp(0, 1, "127.0.0.1", 0, ctx.comm_new_id)
    assert len(comm_id) == ctx.lib.blsq_comm_id_bytes() == 128
    sol = TsqrTrfSolver(m, n, 1, 0, ctx=ctx, m_total=m, comm_id=comm_id)
    assert ctx.lib.blsq_comm_size(ctx.h) == 1 and ctx.lib.blsq_comm_rank(ctx.h) == 0
    d = {k: ctx.to_device(P[k]) for k in ("J", "f", "x", "lb", "ub", "scale")}
    ctx.gram_stats(reset=True)
    sol.factor_dev(d["J"], d["f"], d["x"], d["lb"], d["ub"], d["scale"])
    stats = ctx.gram_stats()
    assert stats == ((1, 0) if fact_path == "gram_front_end" else (0, 0))
    for Delta in (0.7, 10.0):
        _, So = orc.trf_step_solve(P["J"], P["f"], P["x"], P["lb"], P["ub"], P["scale"], Delta, 0.0)
        S = sol.step(np.array([Delta]), np.array([0.0]))
        assert rel(S.step[0], So.step) < RTOL
        np.testing.assert_array_equal(S.hits[0], So.hits)
        assert int(S.n_iter[0]) == So.n_iter
    assert np.allclose(ctx.comm_max([1.5, -2.0]), [1.5, -2.0])       # (single rank: identity)
    sol.close()
    ctx.comm_destroy()
    ctx.close()


def test_mixed_rank_batch_takes_both_paths(bl, blsq_opt):
    """A batch mixing well-conditioned, rank-deficient and badly conditioned problems:
    the gate must send only the clearly full-rank ones down the SVD-free path, and
    every problem must still match the oracle."""
    from oracle import blsq_oracle as orc
    from bounded_lsq import _synth
    blsq_opt("BLSQ_SVDFREE_MIN_N", "0")
    blsq_opt("BLSQ_NO_SVDFREE", "0")
    B, m, n = 6, 120, 12
    P = _synth.trf_batch(31337, B, m, n, unbounded=True)
    P["J"][1, :, 5] = P["J"][1, :, 2]                      # exactly rank deficient
    P["J"][3, :, 7] = 0.0                                  # zero column
    P["J"][4, :, 9] = P["J"][4, :, 1] * (1 + 1e-13)        # cond ~ 1e13: beyond the gate
    # rank-deficient problems get a Delta inside the min-norm step (beyond it the reference's
    # answer is decided by LAPACK null-space noise, see KNIFE_EDGE)
    Delta = np.array([0.3, 0.1, 5.0, 0.1, 0.1, 0.05])
    sol = bl.TrfStepSolver(B, m, n)
    sol.factor(P["J"], P["f"], P["x"], P["lb"], P["ub"], P["scale"])
    fast = sol.debug_fast()
    assert list(fast) == [1, 0, 1, 0, 0, 1]
    S = sol.step(Delta, np.zeros(B))
    for b in range(B):
        _, So = orc.trf_step_solve(P["J"][b], P["f"][b], P["x"][b], P["lb"][b], P["ub"][b],
                                   P["scale"][b], Delta[b], 0.0)
        if b == 4:      # cond 1e13: the reference's own answer is only good to ~cond*eps
            assert rel(S.step[b], So.step) < 1e-2
            continue
        assert int(S.n_iter[b]) == So.n_iter, b
        assert rel(S.step[b], So.step) < RTOL, (b, rel(S.step[b], So.step))
    sol.close()


# ---- size-independent properties at BASELINE.json's full size (4096 x 256) -----------------
FULL = (4096, 256)


def _full_batch(kind, B, seed):
    from bounded_lsq import _synth
    m, n = FULL
    return (_synth.trf_batch if kind == "trf" else _synth.dogbox_batch)(seed, B, m, n)


def test_full_size_batch_independence_and_determinism(bl):
    """A problem's result does not depend on its batch neighbours or on the run: solving a
    batch of 24, the same batch again, and two of its problems alone (B = 1 plans) gives
    bit-identical steps and masks."""
    m, n = FULL
    B = 24
    P = _full_batch("trf", B, 77)
    Delta = np.where(np.arange(B) % 2 == 0, 10.0, 0.5)
    sol = bl.TrfStepSolver(B, m, n)
    sol.factor(P["J"], P["f"], P["x"], P["lb"], P["ub"], P["scale"])
    S1 = sol.step(Delta, np.zeros(B))
    step1, hits1, xnew1 = S1.step.copy(), S1.hits.copy(), S1.x_new.copy()
    sol.factor(P["J"], P["f"], P["x"], P["lb"], P["ub"], P["scale"])
    S2 = sol.step(Delta, np.zeros(B))
    np.testing.assert_array_equal(step1, S2.step)
    np.testing.assert_array_equal(hits1, S2.hits)
    sol.close()
    one = bl.TrfStepSolver(1, m, n)
    for b in (3, 20):
        sl = slice(b, b + 1)
        one.factor(P["J"][sl], P["f"][sl], P["x"][sl], P["lb"][sl], P["ub"][sl], P["scale"][sl])
        S = one.step(Delta[sl], np.zeros(1))
        np.testing.assert_array_equal(S.step[0], step1[b])
        np.testing.assert_array_equal(S.hits[0], hits1[b])
        np.testing.assert_array_equal(S.x_new[0], xnew1[b])
    one.close()


def test_full_size_row_permutation_invariance(bl):
    """J^T J and J^T f do not change when the residuals are reordered, so neither does the
    step: permuting the m rows of (J, f) moves the TRF and dogbox steps by rounding only and
    leaves every mask unchanged."""
    m, n = FULL
    B = 4
    rng = np.random.default_rng(5)
    perm = rng.permutation(m)
    P = _full_batch("trf", B, 78)
    Delta = np.where(np.arange(B) % 2 == 0, 10.0, 0.5)
    sol = bl.TrfStepSolver(B, m, n)
    sol.factor(P["J"], P["f"], P["x"], P["lb"], P["ub"], P["scale"])
    S = sol.step(Delta, np.zeros(B))
    step, hits, branch = S.step.copy(), S.hits.copy(), S.branch.copy()
    sol.factor(np.ascontiguousarray(P["J"][:, perm]), np.ascontiguousarray(P["f"][:, perm]),
               P["x"], P["lb"], P["ub"], P["scale"])
    Sp = sol.step(Delta, np.zeros(B))
    for b in range(B):
        assert rel(Sp.step[b], step[b]) < RTOL
    np.testing.assert_array_equal(Sp.hits, hits)
    np.testing.assert_array_equal(Sp.branch, branch)
    sol.close()
    Pd = _full_batch("dogbox", B, 79)
    Dd = np.where(np.arange(B) % 2 == 0, 0.02, 0.005)
    dog = bl.DogboxStepSolver(B, m, n)
    dog.factor(Pd["J"], Pd["f"], Pd["x"], Pd["lb"], Pd["ub"], Pd["scale"], Pd["on_bound"])
    D = dog.step(Dd)
    dstep, donb = D.step.copy(), D.on_bound_new.copy()
    dog.factor(np.ascontiguousarray(Pd["J"][:, perm]), np.ascontiguousarray(Pd["f"][:, perm]),
               Pd["x"], Pd["lb"], Pd["ub"], Pd["scale"], Pd["on_bound"])
    Dp = dog.step(Dd)
    for b in range(B):
        assert rel(Dp.step[b], dstep[b]) < RTOL
    np.testing.assert_array_equal(Dp.on_bound_new, donb)
    dog.close()


def test_full_size_residual_scaling(bl):
    """Scaling f and Delta by 4 (a power of two: exact in binary floating point) scales the
    unconstrained trust-region step exactly — same Newton iterates, same masks."""
    m, n = FULL
    B = 4
    P = _full_batch("trf", B, 80)
    lb = np.full_like(P["lb"], -np.inf); ub = np.full_like(P["ub"], np.inf)
    Delta = np.full(B, 0.5)
    sol = bl.TrfStepSolver(B, m, n)
    sol.factor(P["J"], P["f"], P["x"], lb, ub, P["scale"])
    S = sol.step(Delta, np.zeros(B))
    step, niter = S.step.copy(), S.n_iter.copy()
    sol.factor(P["J"], 4.0 * P["f"], P["x"], lb, ub, P["scale"])
    S4 = sol.step(4.0 * Delta, np.zeros(B))
    np.testing.assert_array_equal(S4.n_iter, niter)
    for b in range(B):
        assert rel(S4.step[b], 4.0 * step[b]) < RTOL
    sol.close()


# ---- Cholesky-QR panel fast path and its fallback ------------------------------------------
def _correlated_batch(B, m, n, rho, seed):
    """J whose columns inside every 16-column panel are strongly correlated (cosine ~ rho)."""
    rng = np.random.default_rng(seed)
    J = rng.standard_normal((B, m, n))
    for p0 in range(0, n, 16):
        common = rng.standard_normal((B, m, 1))
        J[:, :, p0:p0 + 16] = np.sqrt(1 - rho) * J[:, :, p0:p0 + 16] + np.sqrt(rho) * common
    return J


@pytest.mark.parametrize("rho", [0.0, 0.5, 0.9, 0.99, 1 - 1e-6, 1 - 1e-12])
def test_correlated_panels_fast_path_and_fallback(bl, rho, fact_path):
    """Panels from well conditioned (Cholesky-QR + Householder reconstruction) to numerically
    dependent (exact Householder column loop): the step matches the oracle either way, and the
    diagnostic counters show which path ran."""
    from oracle import blsq_oracle as orc
    from bounded_lsq import _synth, _abi
    B, m, n = 3, 1500, 48
    P = _synth.trf_batch(4242, B, m, n)
    P["J"] = _correlated_batch(B, m, n, rho, 7)
    Delta = np.array([10.0, 0.5, 2.0])
    ctx = _abi.Context(0)
    sol = bl.TrfStepSolver(B, m, n, ctx=ctx)
    ctx.cqr_stats(reset=True)
    sol.factor(P["J"], P["f"], P["x"], P["lb"], P["ub"], P["scale"])
    fast, slow = ctx.cqr_stats()
    S = sol.step(Delta, np.zeros(B))
    if fact_path == "qr_tree_only":          # (with the Gram front end a well conditioned
        assert fast + slow > 0               #  problem may never reach the QR kernel at all)
        if rho <= 0.5:
            assert fast > 0                  # well conditioned panels take the fast path
    if rho >= 1 - 1e-6 and (fact_path == "qr_tree_only" or fast + slow > 0):
        assert slow > 0                      # numerically dependent columns must not
    for b in range(B):
        _, So = orc.trf_step_solve(P["J"][b], P["f"][b], P["x"][b], P["lb"][b], P["ub"][b],
                                   P["scale"][b], Delta[b], 0.0)
        assert rel(S.step[b], So.step) < RTOL, (rho, b, rel(S.step[b], So.step))
        np.testing.assert_array_equal(S.hits[b], So.hits)
    sol.close(); ctx.close()


@pytest.mark.parametrize("s_", [0.97, 0.94, 0.92, 0.85])
def test_kahan_structured_panels(bl, s_):
    """Adversarial for a Gram-based panel factorisation: every 16-column panel is Q_p K with K a
    Kahan matrix (geometric diagonal, cond(K) 4e1 ... 1e4 while every Cholesky pivot stays
    moderate).  The pivot threshold of the fast path must hand the bad ones to the exact
    Householder loop; the step matches the oracle throughout."""
    from oracle import blsq_oracle as orc
    from bounded_lsq import _synth
    B, m, n = 2, 1200, 48
    rng = np.random.default_rng(1)
    c_ = np.sqrt(1 - s_ ** 2)
    K = np.zeros((16, 16))
    for i in range(16):
        K[i, i] = s_ ** i
        K[i, i + 1:] = -c_ * s_ ** i
    P = _synth.trf_batch(99, B, m, n)
    J = np.empty((B, m, n))
    for b in range(B):
        Q, _ = np.linalg.qr(rng.standard_normal((m, n)))
        for p0 in range(0, n, 16):
            J[b][:, p0:p0 + 16] = Q[:, p0:p0 + 16] @ K
    P["J"] = J
    Delta = np.array([10.0, 0.5])
    sol = bl.TrfStepSolver(B, m, n)
    sol.factor(P["J"], P["f"], P["x"], P["lb"], P["ub"], P["scale"])
    S = sol.step(Delta, np.zeros(B))
    for b in range(B):
        _, So = orc.trf_step_solve(P["J"][b], P["f"][b], P["x"][b], P["lb"][b], P["ub"][b],
                                   P["scale"][b], Delta[b], 0.0)
        assert rel(S.step[b], So.step) < 1e-11, (s_, b, rel(S.step[b], So.step))
        np.testing.assert_array_equal(S.hits[b], So.hits)
    sol.close()


def test_randomised_shapes_against_oracle(bl):
    """40 seeded random shapes around every structural boundary of the QR (panel width 16, leaf
    height 1024, pair mode, stacked merges, partial panels, one-column problems): TRF and dogbox
    steps against the oracle."""
    from oracle import blsq_oracle as orc
    from bounded_lsq import _synth
    rng = np.random.default_rng(20240)
    ns = [1, 2, 15, 16, 17, 31, 32, 33, 47, 48, 49, 63, 64, 65, 80, 96, 97, 128]
    worst = 0.0
    for it in range(40):
        n = int(rng.choice(ns))
        m = int(rng.choice([n, n + 1, n + 7, 2 * n + 3, 511, 512, 513, 1023, 1024, 1025, 2047,
                            2049, 3000]))
        m = max(m, n)
        B = int(rng.integers(1, 4))
        P = _synth.trf_batch(7000 + it, B, m, n)
        Delta = rng.choice([10.0, 0.5, 0.05], size=B)
        sol = bl.TrfStepSolver(B, m, n)
        sol.factor(P["J"], P["f"], P["x"], P["lb"], P["ub"], P["scale"])
        S = sol.step(Delta, np.zeros(B))
        for b in range(B):
            _, So = orc.trf_step_solve(P["J"][b], P["f"][b], P["x"][b], P["lb"][b], P["ub"][b],
                                       P["scale"][b], Delta[b], 0.0)
            e = rel(S.step[b], So.step)
            worst = max(worst, e)
            assert e < RTOL, ("trf", it, B, m, n, b, e)
            np.testing.assert_array_equal(S.hits[b], So.hits)
        sol.close()
        Pd = _synth.dogbox_batch(8000 + it, B, m, n)
        Dd = rng.choice([0.02, 0.005, 1.0], size=B)
        dog = bl.DogboxStepSolver(B, m, n)
        dog.factor(Pd["J"], Pd["f"], Pd["x"], Pd["lb"], Pd["ub"], Pd["scale"], Pd["on_bound"])
        D = dog.step(Dd)
        for b in range(B):
            _, So = orc.dogbox_step_solve(Pd["J"][b], Pd["f"][b], Pd["x"][b], Pd["lb"][b],
                                          Pd["ub"][b], Pd["scale"][b], Pd["on_bound"][b], Dd[b])
            e = rel(D.step[b], So.step)
            assert e < RTOL, ("dogbox", it, B, m, n, b, e)
            np.testing.assert_array_equal(D.on_bound_new[b], So.on_bound_new)
        dog.close()
    assert worst < RTOL


# ---- BASELINE configs 3 / 4 at their batch sizes: B = 1024, 512 x 64 -------------------------
@pytest.mark.parametrize("kind", ["dogbox", "trf"])
def test_batch_of_1024_problems_512x64(bl, kind):
    """Config 3 (dogbox) / the per-GPU share of config 4 (TRF): 1024 problems of 512 x 64 in one
    call.  16 sampled problems against the oracle (1e-10 / bit-exact masks); all 1024 through
    size-independent properties: feasibility, model decrease, the trust-region bound, and — the
    batch being 64 copies of the same 16 problems — bit-identical results for every copy (a
    problem's arithmetic does not depend on its position in the launch)."""
    from oracle import blsq_oracle as orc
    from bounded_lsq import _synth
    B, m, n, K = 1024, 512, 64, 16
    gen = _synth.dogbox_batch if kind == "dogbox" else _synth.trf_batch
    P0 = gen(31000, K, m, n)
    P = {k: np.tile(v, (B // K,) + (1,) * (v.ndim - 1)) for k, v in P0.items()}
    if kind == "trf":
        Delta = np.tile(np.where(np.arange(K) % 2 == 0, 10.0, 0.5), B // K)
        sol = bl.TrfStepSolver(B, m, n)
        sol.factor(P["J"], P["f"], P["x"], P["lb"], P["ub"], P["scale"])
        S = sol.step(Delta, np.zeros(B))
        mask = S.hits
    else:
        Delta = np.tile(np.where(np.arange(K) % 2 == 0, 0.02, 0.005), B // K)
        sol = bl.DogboxStepSolver(B, m, n)
        sol.factor(P["J"], P["f"], P["x"], P["lb"], P["ub"], P["scale"], P["on_bound"])
        S = sol.step(Delta)
        mask = S.on_bound_new
    sol.close()
    for b in range(K):                                   # the oracle on the 16 distinct problems
        if kind == "trf":
            _, So = orc.trf_step_solve(P0["J"][b], P0["f"][b], P0["x"][b], P0["lb"][b], P0["ub"][b],
                                       P0["scale"][b], Delta[b], 0.0)
            np.testing.assert_array_equal(S.hits[b], So.hits)
            assert int(S.n_iter[b]) == So.n_iter and int(S.branch[b]) == So.branch
        else:
            _, So = orc.dogbox_step_solve(P0["J"][b], P0["f"][b], P0["x"][b], P0["lb"][b],
                                          P0["ub"][b], P0["scale"][b], P0["on_bound"][b], Delta[b])
            np.testing.assert_array_equal(S.on_bound_new[b], So.on_bound_new)
            assert int(S.tr_hit[b]) == int(So.tr_hit)
        assert rel(S.step[b], So.step) < RTOL, (b, rel(S.step[b], So.step))
        assert abs(S.predicted_reduction[b] - So.predicted_reduction) <= 1e-10 * abs(So.predicted_reduction)
    # every copy of a problem: the same bits
    for arr in (S.step, S.x_new, mask, S.predicted_reduction):
        a = np.asarray(arr).reshape((B // K, K) + np.asarray(arr).shape[1:])
        assert np.array_equal(a, np.broadcast_to(a[0], a.shape))
    # properties over the whole batch
    assert np.all(S.status == 0)
    assert np.all(S.predicted_reduction > 0)
    if kind == "trf":
        assert np.all(S.x_new > P["lb"]) and np.all(S.x_new < P["ub"])      # strictly feasible
        assert np.all(S.step_h_norm <= Delta * (1 + 1e-12))
    else:
        assert np.all(S.x_new >= P["lb"]) and np.all(S.x_new <= P["ub"])
        assert np.all(np.abs(S.step) <= Delta[:, None] * P["scale"] * (1 + 1e-15))
