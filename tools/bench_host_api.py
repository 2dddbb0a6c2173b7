"""PCIe-inclusive rate of the host-pointer API (never the bench `value`): the same TRF step-solve
workload as bench.py, but J / f / x / bounds are numpy arrays handed over on every factor call
(blsq_trf_factor copies them to the device) and the step outputs come back to numpy."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bounded-lsq_amd"))
from bounded_lsq import TrfStepSolver, _synth

B, m, n = 256, 4096, 256
P = _synth.trf_batch(10_000, B, m, n)
Delta = np.where(np.arange(B) % 2 == 0, 10.0, 0.5)
sol = TrfStepSolver(B, m, n)
for it in range(4):
    t0 = time.perf_counter()
    sol.factor(P["J"], P["f"], P["x"], P["lb"], P["ub"], P["scale"])
    t1 = time.perf_counter()
    S = sol.step(Delta, np.zeros(B))
    t2 = time.perf_counter()
    if it:
        print("host API: factor %.1f ms (J = %.2f GB over PCIe -> %.1f GB/s incl. compute), step %.1f ms"
              " -> %.0f step-solves/s" % (1e3 * (t1 - t0), P["J"].nbytes / 1e9,
                                         P["J"].nbytes / 1e9 / (t1 - t0), 1e3 * (t2 - t1), B / (t2 - t0)))
sol.close()
