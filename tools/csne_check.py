"""Calibration of the CSNE tier: step error against the oracle, measured first-order correction eta and the proven bound
K2 over a sweep of kappa(J) (log-spaced spectra, unbounded problems, three trust-region radii per kappa).
  python tools/csne_check.py [m n]          (product build: problems beyond CSNE_ETA_MAX show eta = -1 and the next tier's error)
A calibration build (make EXTRA_DEFS=-DBLSQ_CSNE_ETA_MAX=1.0) shows what the tier itself would deliver there."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "bounded-lsq_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bounded_lsq as bl
from bounded_lsq import _abi, _synth
from oracle import blsq_oracle as orc

m, n = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4096, 256)
rng = np.random.default_rng(5)
print("kappa      K2          eta        err(step)  err(alpha) n_iter(ours/ref)  tier")
for kappa in (3e2, 1e3, 3e3, 1e4, 3e4, 1e5, 3e5, 1e6, 3e6):
    B = 3
    P = _synth.trf_batch(77, B, m, n, unbounded=True)
    for b in range(B):
        U, _ = np.linalg.qr(rng.standard_normal((m, n)))
        V, _ = np.linalg.qr(rng.standard_normal((n, n)))
        P["J"][b] = (U * np.logspace(0, -np.log10(kappa), n)) @ V.T * np.sqrt(m)
    Delta = np.array([10.0, 0.5, 0.05])
    if os.environ.get("CSNE_CHECK_HARD"):                    # the hard regime: alpha small against sigma_min^2, or no alpha at all
        pg = [np.linalg.norm(np.linalg.lstsq(P["J"][b], -P["f"][b], rcond=None)[0]) for b in range(B)]
        Delta = np.array([2.0 * pg[0], 0.9 * pg[1], 0.3 * pg[2]])
    ctx = _abi.Context(0)
    sol = bl.TrfStepSolver(B, m, n, ctx=ctx)
    ctx.csne_stats(reset=True); ctx.cqr2_stats(reset=True)
    sol.factor(P["J"], P["f"], P["x"], P["lb"], P["ub"], P["scale"])
    k2 = sol.debug_cond()
    S = sol.step(Delta, np.zeros(B))
    on, eta = sol.debug_csne()
    cs = ctx.csne_stats(); cq = ctx.cqr2_stats()
    for b in range(B):
        _, So = orc.trf_step_solve(P["J"][b], P["f"][b], P["x"][b], P["lb"][b], P["ub"][b], P["scale"][b], Delta[b], 0.0)
        e = np.linalg.norm(S.step[b] - So.step) / np.linalg.norm(So.step)
        ea = abs(S.alpha[b] - So.alpha) / max(abs(So.alpha), 1e-300)
        print(f"{kappa:8.1e}  {k2[b]:10.3e}  {eta[b]:10.3e} {e:10.3e} {ea:10.3e}   {int(S.n_iter[b])}/{So.n_iter}"
              f"   csne={cs} cqr2={cq}", flush=True)
    sol.close(); ctx.close()
