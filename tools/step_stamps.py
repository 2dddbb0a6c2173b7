"""Diagnostic: phases of trf_step_kernel for problem 0 (diagnostic build: make -C bounded-lsq_amd/csrc diag,
   BLSQ_LIB=.../libblsq_hip_diag.so):  python tools/step_stamps.py c2-single|c4|c2"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bounded-lsq_amd")); sys.path.insert(0, ROOT)
import bench
from bounded_lsq import _abi
kind = sys.argv[1] if len(sys.argv) > 1 else "c2-single"
ctx = _abi.Context(0)
b = bench.Bench(kind, ctx, 0, 1)
for _ in range(3): b.step()
ctx.sync()
st = np.zeros(32, dtype=np.int64)
fn = ctx.lib.blsq_debug_step_stamps; fn.argtypes = [C.c_void_p]; fn.restype = C.c_int
assert fn(st.ctypes.data) == 0
names = ["p (solve / load)", "p_h out, p = d p_h", "step_to_bound, hits", "reflected direction", "intersect sums",
         "step_to_bound 2", "model products (one pass)", "sums along r_h", "r_h, p_h; step_to_bound 3",
         "sums along -g_h, c_h", "nine sums, choice", "step, x_new, active set", "last two sums, scalars"]
print(kind, "(problem 0; zero-length phases belong to the other branch)")
for i, nm in enumerate(names):
    if st[i + 1] and st[i]: print("  %-32s %6.2f us" % (nm, 0.01 * (st[i + 1] - st[i])))
print("  total %.2f us" % (0.01 * (st[12] - st[0])))
b.close(); ctx.close()
