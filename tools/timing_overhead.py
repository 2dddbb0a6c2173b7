"""Diagnostic: what the per-launch timing events cost the step they measure (headline workload)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bounded-lsq_amd")); sys.path.insert(0, ROOT)
import bench
from bounded_lsq import _abi
ctx = _abi.Context(0)
b = bench.Bench("c2", ctx, 0, 1)
for _ in range(5): b.step()
ctx.sync()
for rep in range(2):
    for tag, on in (("events off", False), ("events on", True)):
        ctx.timing(on); ctx.timing_reset()
        t0 = time.perf_counter()
        for _ in range(60): b.step()
        ctx.sync()
        e = time.perf_counter() - t0
        ctx.timing(False)
        print("%-10s %.4f ms per step  (%.0f step-solves/s)" % (tag, 1e3 * e / 60, 512 * 60 / e), flush=True)
b.close(); ctx.close()
