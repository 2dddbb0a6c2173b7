"""Diagnostic: phases of gram_chol_reg_kernel (N <= 80, one wave per problem) for one problem of a 1024-problem launch."""
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
st = np.zeros((4, 20, 8), dtype=np.int64)
fn = ctx.lib.blsq_debug_chol_stamps; fn.argtypes = [C.c_void_p]; fn.restype = C.c_int
assert fn(st.ctypes.data) == 0
m = st[0][19]
us = lambda x: 0.01 * x
names = ["scales", "tile loads", "factorisation (5 row blocks)", "certificate bound", "TRF finish", "dogbox finish"]
print(kind, " ".join("%s %.2f" % (nm, us(m[i + 1] - m[i])) for i, nm in enumerate(names)), "| total %.2f us" % us(m[6] - m[0]))
b.close(); ctx.close()
