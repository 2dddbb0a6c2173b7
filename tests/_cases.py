"""Engineered inputs shared by CPU (oracle) and GPU tests: the two conditions under which the
reference's intersect_trust_region raises ValueError (trust_region.py:28-29, 34-35)."""
import numpy as np

from oracle import blsq_oracle as orc


def zero_direction_problem(seed=3000, m=24, n=6):
    from bounded_lsq import _synth
    P = _synth.trf_problem(seed, m, n, unbounded=True)
    P["f"] = P["f"] * 1e-170
    F = orc.trf_factor(P["J"], P["f"], P["x"], P["lb"], P["ub"], P["scale"])
    S = orc.trf_step(F, 1.0, 0.0)
    p = F.d * S.p_h_tr
    j = int(np.flatnonzero((p < 0) & (F.g < 0))[0])
    P["x"][j] = 5e-324                 # one denormal above the bound (bounds.py:91-94 leaves this)
    P["lb"][j] = 0.0
    return P


def knife_edge_batch(B=256, m=24, n=6, base_seed=5000):
    """Problems whose trust-region step (||p_h|| = Delta after the rescale) ends the largest
    representable fraction below 1 of the way to a bound: to_bound = 1 - 2^-53, so that
    c = ||p_h to_bound||^2 - Delta^2 is a matter of roundings."""
    from bounded_lsq import _synth
    out = dict(J=np.empty((B, m, n)), f=np.empty((B, m)), x=np.empty((B, n)), lb=np.empty((B, n)),
               ub=np.empty((B, n)), scale=np.ones((B, n)), Delta=np.empty(B))
    b, seed = 0, base_seed
    target = np.nextafter(1.0, 0.0)
    while b < B:
        seed += 1
        P = _synth.trf_problem(seed, m, n)
        rng = np.random.default_rng(seed + 5)
        w = rng.uniform(5, 10, n)
        j = int(rng.integers(n))
        # the variable that will hit sits at exactly 0: only there is (bound - x) / p fine-grained
        # enough to land on 1 - 2^-53 (elsewhere ulp(bound) / |p| ~ 1e-15 is the resolution)
        P["x"][j] = 0.0
        P["lb"] = P["x"] - w
        P["ub"] = P["x"] + w
        F = orc.trf_factor(P["J"], P["f"], P["x"], P["lb"], P["ub"], P["scale"])
        Delta = 0.5 * np.linalg.norm(F.V.dot(F.uf / F.s))
        S = orc.trf_step(F, Delta, 0.0)
        p = F.d * S.p_h_tr
        # moving AWAY from the bound that defines v_j, so that editing the other one changes nothing
        if S.branch != 0 or not (p[j] * F.g[j] > 0):
            continue
        key = "ub" if p[j] > 0 else "lb"
        bound = p[j] * target
        for _ in range(8):                     # walk the bound ulp by ulp until to_bound == target
            t = bound / p[j]
            if t == target:
                break
            bound = np.nextafter(bound, 0.0 if t > target else 2 * p[j])
        P[key][j] = bound
        for k in ("J", "f", "x", "lb", "ub"):
            out[k][b] = P[k]
        out["Delta"][b] = Delta
        b += 1
    return out
