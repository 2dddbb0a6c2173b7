"""Throughput of the other configured shapes (BASELINE.json configs[2..3]: 512x64) for TRF and
dogbox step-solves, device-resident inputs.  Not the headline bench; numbers go to DESIGN.md."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bounded-lsq_amd")); sys.path.insert(0, ROOT)
from bounded_lsq import TrfStepSolver, DogboxStepSolver, _abi, _synth

ctx = _abi.Context(0)
SHAPES = [("trf", 8192, 512, 64), ("dogbox", 8192, 512, 64), ("trf", 1024, 2048, 128)]
if len(sys.argv) > 1:                      # kind:B:m:n ...
    SHAPES = [(a.split(":")[0],) + tuple(int(v) for v in a.split(":")[1:]) for a in sys.argv[1:]]
for kind, B, m, n in SHAPES:
    P = _synth.dogbox_batch(5, B, m, n) if kind == "dogbox" else _synth.trf_batch(5, B, m, n)
    d = {k: ctx.to_device(P[k]) for k in P}
    if kind == "trf":
        sol = TrfStepSolver(B, m, n, ctx=ctx)
        dD = ctx.to_device(np.where(np.arange(B) % 2 == 0, 10.0, 0.5)); dA = ctx.to_device(np.zeros(B))
        def step():
            sol.factor_dev(d["J"], d["f"], d["x"], d["lb"], d["ub"], d["scale"]); sol.step_dev(dD, dA)
    else:
        sol = DogboxStepSolver(B, m, n, ctx=ctx)
        dD = ctx.to_device(np.full(B, 0.02))
        def step():
            sol.factor_dev(d["J"], d["f"], d["x"], d["lb"], d["ub"], d["scale"], d["on_bound"]); sol.step_dev(dD)
    ctx.cqr_stats(reset=True); ctx.gram_stats(reset=True)
    step(); ctx.sync()
    print("   one step-solve batch: problems (normal-equations path, QR tree) =", ctx.gram_stats(),
          " QR panels (Cholesky-QR, Householder column loop) =", ctx.cqr_stats())
    ctx.timing(True); ctx.timing_reset()
    t0 = time.perf_counter()
    K = 5
    for _ in range(K): step()
    ctx.sync(); el = time.perf_counter() - t0
    tm = ctx.timing_read(); ctx.timing(False)
    gbs = B * m * (n + 1) * 8 * K / el / 1e9
    tfs = B * (2.0 * m * n * n - 2.0 * n ** 3 / 3) * K / el / 1e12
    print("%-6s B=%d %dx%d: %.0f step-solves/s  (%.2f ms/step; J read at %.0f GB/s, QR %.1f TF/s)  kernels ms/step: %s" % (
        kind, B, m, n, B * K / el, 1e3 * el / K, gbs, tfs,
        {k: round(v[0] / K, 3) for k, v in tm.items() if v[1]}))
    sol.close()
    for p in d.values(): ctx.free(p)
