"""Diagnostic: per-chunk timeline of one workgroup of cqr2_apply_kernel (512 problems of 4096 x 256, all through the tier;
diagnostic build)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bounded-lsq_amd")); sys.path.insert(0, ROOT)
import bench
from bounded_lsq import _abi
def all_rejected(P):
    rng = np.random.default_rng(4243)
    nn = P["J"].shape[2]
    V, _ = np.linalg.qr(rng.standard_normal((nn, nn)))
    K = min(64, P["J"].shape[0])
    sv = np.logspace(0.0, -np.log10(3e3), nn)
    for b in range(K):
        P["J"][b] = (P["J"][b] @ (V * sv)) @ V.T
    for b in range(K, P["J"].shape[0]):
        P["J"][b] = P["J"][b % K]
    P["lb"][:] = -np.inf; P["ub"][:] = np.inf
ctx = _abi.Context(0)
bm = bench.Bench("c2", ctx, 0, 1, mutate=all_rejected)
for _ in range(2): bm.step()
ctx.sync()
st = np.zeros((8, 130, 8), dtype=np.int64)
fn = ctx.lib.blsq_debug_cqr2_stamps; fn.argtypes = [C.c_void_p]; fn.restype = C.c_int
assert fn(st.ctypes.data) == 0
us = lambda x: 0.01 * x
nc = 16                                   # chunks of 32 rows per workgroup of 512 rows
for w in range(8):
    a = st[w, 2:nc - 1]
    ph = [us(a[:, i + 1] - a[:, i]).mean() for i in range(7)]
    per = us(st[w, 3:nc - 1, 0] - st[w, 2:nc - 2, 0]).mean()
    print("wave %d: issue + late stores %.2f | first k-tiles %.2f | commit + issue %.2f | last k-tiles %.2f | early stores %.2f | w_f %.2f | commit + barrier %.2f || chunk %.2f us"
          % (w, *ph, per))
bm.close(); ctx.close()
