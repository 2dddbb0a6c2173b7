"""Diagnostic: timeline of stage 0 of the certificate (one-pass form) for ONE problem (diagnostic build)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bounded-lsq_amd"))
from bounded_lsq import TrfStepSolver, _abi, _synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
P = _synth.trf_batch(1, B, 1024, 256)
ctx = _abi.Context(0)
sol = TrfStepSolver(B, 1024, 256, ctx=ctx)
d = {k: ctx.to_device(P[k]) for k in ("J", "f", "x", "lb", "ub", "scale")}
for _ in range(3):
    sol.factor_dev(d["J"], d["f"], d["x"], d["lb"], d["ub"], d["scale"]); ctx.sync()
st = np.zeros((4, 20, 8), dtype=np.int64)
fn = ctx.lib.blsq_debug_chol_stamps; fn.argtypes = [C.c_void_p]; fn.restype = C.c_int
assert fn(st.ctypes.data) == 0
us = lambda x: 0.01 * x
m = st[0][18]
print("B = %d: entry -> scales + share read %.2f, invdiag + init %.2f, 16 block steps %.2f, reductions + verdict %.2f; total %.2f us"
      % (B, us(m[1] - m[0]), us(m[2] - m[1]), us(m[3] - m[2]), us(m[4] - m[3]), us(m[4] - m[0])))
steps = [us(st[1][kb][0]) for kb in range(15, -1, -1)]
print("block steps (us):", np.round(np.diff([us(m[2])] + steps), 2))
for wv, tag in ((1, "wave 0"), (2, "wave 2")):
    print(tag + ": per step — wait for the panel, barrier, issue of the next, [chain], barrier, update, barrier")
    for kb in (15, 12, 8, 4, 1):
        r = st[wv][kb]
        print("  kb %2d: vmcnt %.2f  barrier %.2f  issue %.2f  chain %.2f  barrier %.2f  update %.2f  barrier %.2f"
              % (kb, us(r[2] - r[1]), us(r[3] - r[2]), us(r[4] - r[3]), us(r[5] - r[4]), us(r[6] - r[5]), us(r[7] - r[6]), us(r[0] - r[7])))

for kb in (15, 8, 1):
    print("  kb %2d chain of wave 0: operands from LDS %.2f us, 16-step substitution %.2f us, store %.2f us"
          % (kb, us(st[3][kb][0] - st[1][kb][4]), us(st[3][kb][1] - st[3][kb][0]), us(st[1][kb][5] - st[3][kb][1])))
