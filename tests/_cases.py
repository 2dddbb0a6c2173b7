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


def knife_edge_base(B=256, m=24, n=6, base_seed=5000):
    """Problems whose trust-region step (||p_h|| = Delta after the rescale) is feasible with wide
    bounds, one variable `j[b]` sitting at exactly x = 0 and moving AWAY from the bound that
    defines its Coleman-Li v_j — so that the other bound of that variable can be placed anywhere
    without changing the step.  Returns the batch, j, and the oracle's p = d * p_h."""
    from bounded_lsq import _synth
    out = dict(J=np.empty((B, m, n)), f=np.empty((B, m)), x=np.empty((B, n)), lb=np.empty((B, n)),
               ub=np.empty((B, n)), scale=np.ones((B, n)), Delta=np.empty(B),
               j=np.empty(B, dtype=int), p=np.empty((B, n)), d=np.empty((B, n)))
    b, seed = 0, base_seed
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
        if S.branch != 0 or not (p[j] * F.g[j] > 0):
            continue
        for k in ("J", "f", "x", "lb", "ub"):
            out[k][b] = P[k]
        out["Delta"][b] = Delta
        out["j"][b] = j
        out["p"][b] = p
        out["d"][b] = F.d
        b += 1
    return out


def knife_edge_place(Q, p):
    """Move the free bound of variable j[b] so that the step p[b] (B x n, the step of WHOEVER is
    going to be tested: the oracle's or the GPU's own) ends the largest representable fraction
    below 1 of the way to it: to_bound = 1 - 2^-53, where c = ||p_h to_bound||^2 - Delta^2 is a
    matter of roundings.  Returns an edited copy."""
    R = {k: np.array(v, copy=True) for k, v in Q.items()}
    target = np.nextafter(1.0, 0.0)
    for b in range(R["J"].shape[0]):
        j = int(R["j"][b])
        pj = p[b, j]
        bound = pj * target
        for _ in range(8):                     # walk the bound ulp by ulp until to_bound == target
            t = bound / pj
            if t == target:
                break
            bound = np.nextafter(bound, 0.0 if t > target else 2 * pj)
        R["ub" if pj > 0 else "lb"][b, j] = bound
    return R


def knife_edge_batch(B=256, m=24, n=6, base_seed=5000):
    """knife_edge_base with the bounds placed for the oracle's own step."""
    Q = knife_edge_base(B, m, n, base_seed)
    return knife_edge_place(Q, Q["p"])
