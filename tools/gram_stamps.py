"""Diagnostic: wall-clock timeline (100 MHz) of the row chunks of gram16_kernel for ONE workgroup of a full launch.
Needs `make -C bounded-lsq_amd/csrc diag` and BLSQ_LIB=.../libblsq_hip_diag.so."""
import ctypes as C
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bounded-lsq_amd"))
from bounded_lsq import TrfStepSolver, _abi, _synth
B, m, n = 512, 4096, 256
P = _synth.trf_batch(1, 8, m, n)
P = {k: np.ascontiguousarray(np.tile(v, (B // 8,) + (1,) * (v.ndim - 1))) for k, v in P.items()}
ctx = _abi.Context(0)
sol = TrfStepSolver(B, m, n, ctx=ctx)
d = {k: ctx.to_device(P[k]) for k in ("J", "f", "x", "lb", "ub", "scale")}
for _ in range(3):
    sol.factor_dev(d["J"], d["f"], d["x"], d["lb"], d["ub"], d["scale"]); ctx.sync()
st = np.zeros((8, 130, 4), dtype=np.int64)
fn = ctx.lib.blsq_debug_gram_stamps
fn.argtypes = [C.c_void_p]; fn.restype = C.c_int
assert fn(st.ctypes.data) == 0
nc = 128
us = lambda x: 0.01 * x
print("workgroup of problem 300: %d chunks of 32 rows; ideal per chunk: 17408 MFMA cycles per SIMD = 8.29 us at 2.1 GHz" % nc)
tot = us(st[0, nc - 1, 3] - st[0, 0, 0])
print("whole loop %.1f us = %.3f us per chunk" % (tot, tot / nc))
for w in range(8):
    a = st[w, :nc]
    half = us(a[:, 1] - a[:, 0]); rest = us(a[:, 2] - a[:, 1]); bar = us(a[:, 3] - a[:, 2])
    print("wave %d: k-steps 0-3 %.2f  k-steps 4-7 (+ first commit) %.2f  wait at the barrier %.2f  (means; barrier max %.2f)"
          % (w, half[2:].mean(), rest[2:].mean(), bar[2:].mean(), bar[2:].max()))
ch = us(st[0, 1:nc, 0] - st[0, :nc - 1, 0])
print("chunk period of wave 0: mean %.2f min %.2f max %.2f; first 4: %s; last 4: %s" % (ch.mean(), ch.min(), ch.max(), ch[:4].round(2), ch[-4:].round(2)))
