"""Where the drop-in front end spends its time on ONE 4096 x 256 problem (GPU box): python tools/prof_frontend.py
cProfile of bounded_lsq.least_squares with numpy callbacks, callbacks' own time listed beside."""
import cProfile, os, pstats, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bounded-lsq_amd")); sys.path.insert(0, ROOT)
import bounded_lsq
from bounded_lsq import _synth

m, n = 4096, 256
P = _synth.trf_batch(2024, 1, m, n)
J1, x1 = np.ascontiguousarray(P["J"][0]), P["x"][0].copy()
y1 = J1 @ x1 + 0.1 * P["f"][0]
cb = {"t": 0.0}


def fun1(xx):
    t_ = time.perf_counter(); r_ = np.tanh(J1 @ xx - y1); cb["t"] += time.perf_counter() - t_
    return r_


def jac1(xx):
    t_ = time.perf_counter(); r_ = (1.0 - np.tanh(J1 @ xx - y1) ** 2)[:, None] * J1; cb["t"] += time.perf_counter() - t_
    return r_


import threadpoolctl
threadpoolctl.threadpool_limits(limits=int(os.environ.get("CB_THREADS", "1")))
kw = dict(jac=jac1, bounds=(P["lb"][0] - 1.0, P["ub"][0] + 1.0), method="trf", max_nfev=12)
bounded_lsq.least_squares(fun1, x1 + 0.01, **kw)
for rep in range(6):
    cb["t"] = 0.0
    t0 = time.perf_counter(); r = bounded_lsq.least_squares(fun1, x1 + 0.01, **kw); e = time.perf_counter() - t0
    print("total %.3f ms, callbacks %.3f ms, nfev %d njev %d -> %.3f ms per iteration beside the callbacks"
          % (1e3 * e, 1e3 * cb["t"], r.nfev, r.njev, 1e3 * (e - cb["t"]) / r.njev))
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    bounded_lsq.least_squares(fun1, x1 + 0.01, **kw)
pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(22)
