"""Diagnostic: phases of dog_step_kernel for problem 0 (diagnostic build, BLSQ_LIB=.../libblsq_hip_diag.so): python tools/dog_stamps.py [c3]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bounded-lsq_amd")); sys.path.insert(0, ROOT)
import bench
from bounded_lsq import _abi
kind = sys.argv[1] if len(sys.argv) > 1 else "c3"
ctx = _abi.Context(0)
b = bench.Bench(kind, ctx, 0, 1)
for _ in range(3): b.step()
ctx.sync()
st = np.zeros(32, dtype=np.int64)
fn = ctx.lib.blsq_debug_dog_stamps; fn.argtypes = [C.c_void_p]; fn.restype = C.c_int
assert fn(st.ctypes.data) == 0
us = lambda a, c: 0.01 * (st[c] - st[a])
print("%s: publish %.2f | entry -> box %.2f | box setup %.2f | dogleg %.2f | hits %.2f | predicted reduction (triangle product) %.2f | scatter, outputs %.2f | total %.2f us"
      % (kind, us(6, 7), us(7, 0), us(0, 1), us(1, 2), us(2, 3), us(3, 4), us(4, 5), us(6, 5)))
b.close(); ctx.close()
