"""Stress run (GPU box): the flag-driven Cholesky against the left-looking kernel's bits, hundreds of repetitions per shape
(python tools/stress_rl2.py; exit code 1 on any mismatch).  Round 3: 520 launches over four shapes, no mismatch."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bounded-lsq_amd")); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bounded_lsq as bl
from bounded_lsq import _synth, _abi
bad = 0
for (B, m, n, reps) in [(520, 260, 256, 120), (300, 300, 96, 150), (7, 500, 150, 150), (64, 400, 271, 100)]:
    P = _synth.trf_batch(170 + n, B, m, n)
    ref = None
    for rep in range(reps + 1):
        os.environ["BLSQ_CHOL_RL"] = "0" if rep == 0 else "1"
        ctx = _abi.Context(0)
        sol = bl.TrfStepSolver(B, m, n, ctx=ctx)
        sol.factor(P["J"], P["f"], P["x"], P["lb"], P["ub"], P["scale"])
        k2 = sol.debug_cond().copy()
        S = sol.step(np.full(B, 0.3), np.zeros(B))
        got = (S.step.copy(), np.asarray(S.alpha).copy())
        sol.close(); ctx.close()
        if ref is None: ref = got
        elif not all(np.array_equal(a, b) for a, b in zip(ref, got)):
            bad += 1; print("MISMATCH", B, m, n, rep, flush=True)
    print("shape", (B, m, n), "reps", reps, "ok" if not bad else "BAD", flush=True)
sys.exit(1 if bad else 0)
