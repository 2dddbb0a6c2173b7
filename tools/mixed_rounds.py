"""Diagnostic: the bench's mixed-conditioning batch (bounded), per-slot kernel times under the kernel choices of
the Cholesky launches (BLSQ_CHOL_RL / BLSQ_CHOL_RL2).  python tools/mixed_rounds.py [steps]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bounded-lsq_amd")); sys.path.insert(0, ROOT)
import bench
from bounded_lsq import _abi

def mixed(P):
    rng = np.random.default_rng(4242)
    nn = P["J"].shape[2]
    V, _ = np.linalg.qr(rng.standard_normal((nn, nn)))
    K = min(64, P["J"].shape[0])
    kap = 10.0 ** rng.uniform(0.0, 4.0, K)
    for b in range(K):
        sv = np.logspace(0.0, -np.log10(kap[b]), nn)
        P["J"][b] = (P["J"][b] @ (V * sv)) @ V.T
    for b in range(K, P["J"].shape[0]):
        P["J"][b] = P["J"][b % K]

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
ctx = _abi.Context(0)
bm = bench.Bench("c2", ctx, 0, 1, mutate=mixed)
for tag, env in (("default", {}), ("left-looking", {"BLSQ_CHOL_RL": "0"}), ("right-looking, all", {"BLSQ_CHOL_RL": "1"}),
                 ("default again", {})):
    for k in ("BLSQ_CHOL_RL", "BLSQ_CHOL_RL2"):
        os.environ.pop(k, None)
    os.environ.update(env)
    bm.step(); ctx.sync()
    ctx.timing(True); ctx.timing_reset()
    t0 = time.perf_counter()
    for _ in range(steps):
        bm.step()
    ctx.sync()
    e = time.perf_counter() - t0
    tm = ctx.timing_read(); ctx.timing(False)
    print("%-20s %.3f ms per step  %s" % (tag, 1e3 * e / steps,
          {k: (round(v[0] / steps, 4), v[1] // steps if len(v) > 1 else None) for k, v in tm.items() if v[0] > 0}), flush=True)
bm.close(); ctx.close()
