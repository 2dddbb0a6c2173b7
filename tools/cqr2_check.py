"""CholeskyQR2 middle tier vs the Householder tree vs the oracle on prescribed spectra (GPU box).
usage: python tools/cqr2_check.py"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bounded-lsq_amd")); sys.path.insert(0, ROOT)
import bounded_lsq as bl
from bounded_lsq import _synth, _abi
from oracle import blsq_oracle as orc
rng = np.random.default_rng(5)
for (B, m, n, kap) in ((4, 4096, 256, 3e3), (3, 1500, 200, 1e5), (3, 3000, 100, 1e6), (3, 3000, 100, 3e6), (3, 3000, 100, 1e7),
                       (2, 700, 129, 1e3), (3, 2100, 255, 3e7), (3, 2100, 255, 1e8), (2, 900, 240, 1e10)):
    P = _synth.trf_batch(77, B, m, n, unbounded=True)
    for b in range(B):
        U, _ = np.linalg.qr(rng.standard_normal((m, n))); V, _ = np.linalg.qr(rng.standard_normal((n, n)))
        P["J"][b] = (U * np.logspace(0, -np.log10(kap), n)) @ V.T * np.sqrt(m)
    Delta = np.array([10.0, 0.5, 0.05, 100.0])[:B]
    ref = [orc.trf_step_solve(P["J"][b], P["f"][b], P["x"][b], P["lb"][b], P["ub"][b], P["scale"][b], Delta[b], 0.0)[1] for b in range(B)]
    # the oracle's own movement under a one-ulp change of J
    move = []
    for b in range(B):
        Jp = np.nextafter(P["J"][b], np.inf * (rng.integers(0, 2, size=P["J"][b].shape) * 2 - 1))
        Sp = orc.trf_step_solve(Jp, P["f"][b], P["x"][b], P["lb"][b], P["ub"][b], P["scale"][b], Delta[b], 0.0)[1]
        move.append(np.linalg.norm(Sp.step - ref[b].step) / np.linalg.norm(ref[b].step))
    out = {}
    for cq in ("1", "0"):
        os.environ["BLSQ_CQR2"] = cq
        ctx = _abi.Context(0)
        sol = bl.TrfStepSolver(B, m, n, ctx=ctx)
        ctx.gram_stats(reset=True); ctx.cqr2_stats(reset=True)
        sol.factor(P["J"], P["f"], P["x"], P["lb"], P["ub"], P["scale"])
        gs, nq = ctx.gram_stats(), ctx.cqr2_stats()
        S = sol.step(Delta, np.zeros(B))
        errs = [np.linalg.norm(S.step[b] - ref[b].step) / np.linalg.norm(ref[b].step) for b in range(B)]
        out[cq] = (gs, nq, errs, [int(S.n_iter[b]) == ref[b].n_iter for b in range(B)])
        sol.close(); ctx.close()
    print("%dx%d kappa %.0e: cqr2 %d of %d | errors cqr2 %s | tree %s | oracle 1-ulp move %s | n_iter ok %s %s"
          % (m, n, kap, out["1"][1], out["1"][0][1], ["%.1e" % e for e in out["1"][2]], ["%.1e" % e for e in out["0"][2]],
             ["%.1e" % e for e in move], out["1"][3], out["0"][3]))
