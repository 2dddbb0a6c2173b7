"""Pin the CPU oracle (oracle/blsq_oracle.py) against golden vectors captured
from the reference itself (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from _golden import load_npz, load_json, unhex, trf_inputs, dog_inputs
from oracle import blsq_oracle as orc

H = load_json("helpers.json")


def _eq(a, b):
    np.testing.assert_array_equal(np.asarray(a), np.asarray(b))


def test_helper_step_to_bound():
    for c in H["step_size_to_bound"]:
        t, hits = orc.step_to_bound(unhex(c["x"]), unhex(c["d"]), unhex(c["lb"]),
                                    unhex(c["ub"]))
        assert t == float.fromhex(c["step"])
        _eq(hits, c["hits"])


def test_helper_active_constraints():
    for c in H["find_active_constraints"]:
        a = orc.active_constraints(unhex(c["x"]), unhex(c["lb"]), unhex(c["ub"]),
                                   rtol=float.fromhex(c["rtol"]))
        _eq(a, c["active"])


def test_helper_nudge_inside():
    for c in H["make_strictly_feasible"]:
        r = orc.nudge_inside(unhex(c["x"]), unhex(c["lb"]), unhex(c["ub"]),
                             rstep=float.fromhex(c["rstep"]))
        _eq(r, unhex(c["out"]))


def test_helper_cl_scaling():
    for c in H["scaling_vector"]:
        v, jv = orc.cl_scaling(unhex(c["x"]), unhex(c["g"]), unhex(c["lb"]),
                               unhex(c["ub"]))
        _eq(v, unhex(c["v"]))
        _eq(jv, unhex(c["jv"]))


def test_helper_sphere_and_quadratic():
    for c in H["intersect_trust_region"]:
        tn, tp = orc.sphere_intersections(unhex(c["x"]), unhex(c["s"]),
                                          float.fromhex(c["Delta"]))
        assert tn == float.fromhex(c["t_neg"]) and tp == float.fromhex(c["t_pos"])
    for c in H["minimize_quadratic"]:
        t, y = orc.quad_1d_min(*(float.fromhex(c[k]) for k in "ablu"))
        assert t == float.fromhex(c["t"]) and y == float.fromhex(c["y"])
    with pytest.raises(ValueError):
        orc.sphere_intersections(np.zeros(2), np.zeros(2), 1.0)
    with pytest.raises(ValueError):
        orc.sphere_intersections(np.array([2.0, 0]), np.ones(2), 1.0)


TRF_CASES = (load_npz("trf_small.npz") + load_npz("trf_large.npz") +
             load_npz("trf_choice2.npz"))    # find_gradient_step wins (trf.py:159-170, choice 2)


@pytest.mark.parametrize("name,ins,out", TRF_CASES, ids=[c[0] for c in TRF_CASES])
def test_trf_tuple_matches_reference(name, ins, out):
    P = trf_inputs(ins)
    F, S = orc.trf_step_solve(P["J"], P["f"], P["x"], P["lb"], P["ub"],
                              P["scale"], P["Delta"], P["alpha0"])
    # the oracle makes the same third-party calls on the same data: exact
    for k in ("g", "v", "jv", "d", "g_h", "diag_h", "s"):
        _eq(getattr(F, k), out[k])
    assert F.g_norm == float(out["g_norm"]) and F.theta == float(out["theta"])
    _eq(np.abs(F.uf), out["abs_uf"])
    _eq(S.p_h_tr, out["p_h_tr"])
    assert S.alpha == float(out["alpha"]) and S.n_iter == int(out["n_iter"])
    assert S.to_bound == float(out["to_bound"])
    _eq(S.hits, out["hits"])
    assert S.branch == int(out["branch"]) and S.choice == int(out["choice"])
    k = S.steps_h.shape[0]
    _eq(S.steps_h, out["steps_h"][:k])
    _eq(S.qp, out["qp"][:k])
    _eq(S.step_h, out["step_h"])
    assert S.predicted_reduction == float(out["predicted_reduction"])
    _eq(S.step, out["step"])
    _eq(S.x_new, out["x_new"])
    assert S.step_h_norm == float(out["step_h_norm"])
    assert S.correction == float(out["correction"])
    _eq(orc.active_constraints(S.x_new, P["lb"], P["ub"], rtol=1e-8),
        out["active_new"])


DOG_CASES = (load_npz("dog_small.npz") + load_npz("dog_large.npz") +
             load_npz("dog_fallback.npz"))   # dogbox.py:211-216 taken (fallback = 1) + near misses


@pytest.mark.parametrize("name,ins,out", DOG_CASES, ids=[c[0] for c in DOG_CASES])
def test_dogbox_tuple_matches_reference(name, ins, out):
    P = dog_inputs(ins)
    F, S = orc.dogbox_step_solve(P["J"], P["f"], P["x"], P["lb"], P["ub"],
                                 P["scale"], P["on_bound"], P["Delta"])
    _eq(F.g, out["g"])
    _eq(F.active.astype(np.uint8), out["active_set"])
    assert F.g_norm == float(out["g_norm"])
    full = np.zeros(F.n)
    full[F.free] = F.newton
    _eq(full, out["newton_full"])
    full[F.free] = F.cauchy
    _eq(full, out["cauchy_full"])
    _eq(S.step, out["step"])
    _eq(S.x_new, out["x_new"])
    _eq(S.on_bound_new, out["on_bound_new"])
    assert int(S.tr_hit) == int(out["tr_hit"])
    assert int(S.fallback) == int(out["fallback"])
    assert S.predicted_reduction == float(out["predicted_reduction"])
    assert S.step_scaled_norm == float(out["step_scaled_norm"])


def test_rankdef_fixture_is_noise_determined():
    """Evidence for tests/test_hip_parity.py::KNIFE_EDGE: replacing the two
    null-space singular values of the reference's SVD (8e-16 and 0) by other
    values of the same rounding-noise size moves the reference's own step by
    O(1) -- so that fixture pins properties, not digits."""
    case = [c for c in load_npz("trf_small.npz") if c[0] == "rankdef_32x8"][0]
    P = trf_inputs(case[1])
    F = orc.trf_factor(P["J"], P["f"], P["x"], P["lb"], P["ub"], P["scale"])
    assert F.s[-1] < 1e-15 and F.s[-2] < 1e-14
    S = orc.trf_step(F, P["Delta"], P["alpha0"])
    rng = np.random.default_rng(0)
    worst = 0.0
    for _ in range(5):
        s2 = F.s.copy()
        uf2 = F.uf.copy()
        s2[-2:] = np.abs(rng.standard_normal(2)) * 1e-16
        uf2[-2:] = rng.standard_normal(2)
        S2 = orc.trf_step(F._replace(s=s2, uf=uf2), P["Delta"], P["alpha0"])
        worst = max(worst, np.linalg.norm(S2.p_h_tr - S.p_h_tr) / np.linalg.norm(S.p_h_tr))
        assert abs(S2.predicted_reduction - S.predicted_reduction) < 1e-3 * S.predicted_reduction
    assert worst > 1e-2


def test_outside_trust_region_is_a_knife_edge_in_the_reference():
    """trust_region.py:34-35 as reached from trf.py:128: solve_lsq_trust_region returns
    ||p_h|| <= Delta up to ONE rounding, and the argument of intersect_trust_region is that
    vector times p_stride < 1 — so "`x` is not within the trust region." needs p_stride within an
    ulp of 1 and lucky roundings.  Evidence on the oracle (= the reference's arithmetic):
    (i) over random steps ||p_h|| / Delta - 1 never exceeds 2 ulp; (ii) on a batch engineered to
    to_bound = 1 - 2^-53 (the hitting variable at x = 0, where (bound - x) / p has that
    resolution) the exception is raised for a few problems and not for the others."""
    from bounded_lsq import _synth
    from _cases import knife_edge_batch
    eps = np.finfo(float).eps
    worst = 0.0
    for seed in range(2000, 2060):
        m, n = [(24, 6), (40, 8), (64, 16)][seed % 3]
        P = _synth.trf_problem(seed, m, n)
        F = orc.trf_factor(P["J"], P["f"], P["x"], P["lb"], P["ub"], P["scale"])
        pg = np.linalg.norm(F.V.dot(F.uf / F.s))
        for frac in (0.9, 0.5, 0.2, 0.05):
            Delta = frac * pg
            p, _, _ = orc.tr_subproblem(F.n, F.m, F.uf, F.s, F.V, Delta, initial_alpha=0.0)
            worst = max(worst, np.linalg.norm(p) / Delta - 1.0)
    assert worst <= 2 * eps
    Q = knife_edge_batch(B=128, m=100, n=64)      # (a few per cent of them raise; none at n = 6)
    raised = 0
    for b in range(128):
        try:
            orc.trf_step_solve(Q["J"][b], Q["f"][b], Q["x"][b], Q["lb"][b], Q["ub"][b],
                               Q["scale"][b], Q["Delta"][b], 0.0)
        except ValueError as exc:
            assert "not within the trust region" in str(exc)
            raised += 1
    assert 0 < raised < 16


def test_zero_direction_needs_underflow():
    """trust_region.py:28-29 from trf.py:128: dot(r_h, r_h) == 0 with r_h a sign-flipped p_h is
    an underflow of the squares (|p_h| ~ 1e-170)."""
    from _cases import zero_direction_problem
    P = zero_direction_problem()
    with pytest.raises(ValueError, match="`s` is zero"):
        orc.trf_step_solve(P["J"], P["f"], P["x"], P["lb"], P["ub"], P["scale"], 1.0, 0.0)
