"""Randomised parity sweep of the step-solve against the CPU oracle across shapes, conditionings and
column scalings (both trust-region solvers).  Not part of the test suite (runtime grows with the
count); prints the worst cases and exits non-zero on any violation of the 1e-10 / bit-exact bar.

usage: python tests/fuzz_parity.py [count] [seed]   (lives under tests/: it uses the CPU oracle)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bounded-lsq_amd")); sys.path.insert(0, ROOT)
import bounded_lsq as bl
from bounded_lsq import _synth, _abi
from oracle import blsq_oracle as orc

count = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ctx = _abi.Context(0)
worst, fails, paths = [], 0, [0, 0]
for case in range(count):
    n = int(rng.choice([rng.integers(1, 17), rng.integers(17, 80), rng.integers(80, 271)]))
    m = int(n + rng.integers(0, 40)) if rng.random() < 0.2 else int(rng.integers(n, 3000))
    B = int(rng.integers(1, 4))
    kind = "trf" if rng.random() < 0.6 else "dogbox"
    P = _synth.dogbox_batch(int(rng.integers(1 << 30)), B, m, n) if kind == "dogbox" else \
        _synth.trf_batch(int(rng.integers(1 << 30)), B, m, n)
    kappa = 10.0 ** rng.uniform(0, 4)
    if rng.random() < 0.7 and m >= n:                       # prescribed spectrum
        for b in range(B):
            U, _ = np.linalg.qr(rng.standard_normal((m, n)))
            V, _ = np.linalg.qr(rng.standard_normal((n, n)))
            P["J"][b] = (U * np.logspace(0, -np.log10(kappa), n)) @ V.T * np.sqrt(m)
    if rng.random() < 0.3:                                  # badly scaled columns
        P["J"] = P["J"] * 10.0 ** rng.uniform(-3, 3, size=(B, 1, n))
    Delta = 10.0 ** rng.uniform(-2, 1.5, size=B)
    ctx.gram_stats(reset=True)
    if kind == "trf":
        sol = bl.TrfStepSolver(B, m, n, ctx=ctx)
        sol.factor(P["J"], P["f"], P["x"], P["lb"], P["ub"], P["scale"])
        S = sol.step(Delta, np.zeros(B))
    else:
        sol = bl.DogboxStepSolver(B, m, n, ctx=ctx)
        sol.factor(P["J"], P["f"], P["x"], P["lb"], P["ub"], P["scale"], P["on_bound"])
        S = sol.step(Delta)
    gs = ctx.gram_stats(); paths[0] += gs[0]; paths[1] += gs[1]
    for b in range(B):
        if kind == "trf":
            _, So = orc.trf_step_solve(P["J"][b], P["f"][b], P["x"][b], P["lb"][b], P["ub"][b],
                                       P["scale"][b], Delta[b], 0.0)
            mask_ok = np.array_equal(S.hits[b], So.hits)
        else:
            _, So = orc.dogbox_step_solve(P["J"][b], P["f"][b], P["x"][b], P["lb"][b], P["ub"][b],
                                          P["scale"][b], P["on_bound"][b], Delta[b])
            if So is None:                                  # every variable active: no step
                continue
            mask_ok = np.array_equal(S.on_bound_new[b], So.on_bound_new)
        den = np.linalg.norm(So.step)
        e = np.linalg.norm(S.step[b] - So.step) / (den if den > 0 else 1.0)
        cond = np.linalg.cond(P["J"][b])
        worst.append((e, kind, B, m, n, "%.1e" % cond, gs))
        # the reference's own SVD answer moves by ~cond * eps: beyond cond ~ 1e5 that, not the bar
        # of 1e-10, is what two correct implementations can agree to
        tol = max(1e-10, 100 * np.finfo(float).eps * cond)
        if not (e < tol) or not mask_ok:
            fails += 1
            print("VIOLATION", kind, (B, m, n), "err %.2e" % e, "tol %.1e" % tol, "mask", mask_ok, "cond %.1e" % cond, gs)
    sol.close()
worst.sort(key=lambda t: -t[0])
print("cases", count, "problems", len(worst), "on (Gram, tree) paths:", paths, "violations:", fails)
for w in worst[:8]:
    print("  err %.2e  %s B=%d %dx%d cond %s paths %s" % w)
ctx.close()
sys.exit(1 if fails else 0)
