"""Is a side config's step bound by the host's enqueue rate?  Issue K steps without a sync, then sync:
   python tools/host_bound.py c3|c4 [K]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bounded-lsq_amd")); sys.path.insert(0, ROOT)
import bench
from bounded_lsq import _abi
kind = sys.argv[1] if len(sys.argv) > 1 else "c3"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 400
ctx = _abi.Context(0)
b = bench.Bench(kind, ctx, 0, 1)
for _ in range(20): b.step()
ctx.sync()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(K): b.step()
    t1 = time.perf_counter()
    ctx.sync()
    t2 = time.perf_counter()
    print("%s: host issue %.1f us/step, until drained %.1f us/step (the stream was %.1f us behind at the end)"
          % (kind, 1e6 * (t1 - t0) / K, 1e6 * (t2 - t0) / K, 1e6 * (t2 - t1)))
b.close(); ctx.close()
