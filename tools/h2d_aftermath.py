"""Diagnostic: does a host-pointer (H2D) factor call leave something behind that slows later device-API calls?"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bounded-lsq_amd")); sys.path.insert(0, ROOT)
import bench
from bounded_lsq import _abi

ctx = _abi.Context(0)
def single(tag):
    b1 = bench.Bench("c2-single", ctx, 0, 1)
    for _ in range(5): b1.step()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(100): b1.step()
    ctx.sync()
    e = time.perf_counter() - t0
    b1.close()
    print("%-40s c2-single %.3f ms per step" % (tag, 10 * e), flush=True)
single("fresh")
big = bench.Bench("c2", ctx, 0, 1, batch=int(sys.argv[1]) if len(sys.argv) > 1 else 128)
big.step(); ctx.sync()
single("after a device-API batch")
big.step_host(); ctx.sync()
single("after one host-pointer (pageable) call")
if hasattr(ctx, "pinned_empty"):
    Jp = ctx.pinned_empty(big.P["J"].shape); fp = ctx.pinned_empty(big.P["f"].shape)
    Jp[...] = big.P["J"]; fp[...] = big.P["f"]
    keep = big.P["J"], big.P["f"]
    big.P["J"], big.P["f"] = Jp, fp
    big.step_host(); big.step_host(); ctx.sync()
    big.P["J"], big.P["f"] = keep
    buf = Jp
    single("after host-pointer calls from page-locked buffers (still held)")
    ctx.pinned_free(Jp); ctx.pinned_free(fp)
    single("after freeing it")
big.close()
single("after closing the batch solver")
import bounded_lsq
from bounded_lsq import _synth
P1 = _synth.trf_batch(10_000, 1, 4096, 256)
J1, x1 = np.ascontiguousarray(P1["J"][0]), P1["x"][0].copy()
y1 = J1 @ x1 + 0.1 * P1["f"][0]
def fun1(xx): return np.tanh(J1 @ xx - y1)
def jac1(xx): return (1.0 - np.tanh(J1 @ xx - y1) ** 2)[:, None] * J1
kw1 = dict(jac=jac1, bounds=(P1["lb"][0] - 1.0, P1["ub"][0] + 1.0), method="trf", max_nfev=12)
bounded_lsq.least_squares(fun1, x1 + 0.01, **kw1)
single("right after least_squares() with numpy callbacks")
time.sleep(1.0)
single("one second later")
sol1 = bounded_lsq.TrfStepSolver(1, 4096, 256); sol1.close()
single("after a plan on the default context")
ctx.close()
