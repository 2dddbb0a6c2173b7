"""Diagnostic: wall-clock timeline of the row blocks of the Cholesky kernels (one problem of a full launch).
Needs `make -C bounded-lsq_amd/csrc diag` and BLSQ_LIB=.../libblsq_hip_diag.so.
  python tools/chol_stamps.py [B]      B > 256: left-looking kernel, B <= 256: right-looking kernel"""
import ctypes as C
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bounded-lsq_amd"))
from bounded_lsq import TrfStepSolver, _abi, _synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
m, n = 1024, 256
P = _synth.trf_batch(1, B, m, n)
ctx = _abi.Context(0)
sol = TrfStepSolver(B, m, n, ctx=ctx)
d = {k: ctx.to_device(P[k]) for k in ("J", "f", "x", "lb", "ub", "scale")}
for _ in range(3):
    sol.factor_dev(d["J"], d["f"], d["x"], d["lb"], d["ub"], d["scale"]); ctx.sync()
st = np.zeros((4, 20, 8), dtype=np.int64)
fn = ctx.lib.blsq_debug_chol_stamps
fn.argtypes = [C.c_void_p]; fn.restype = C.c_int
assert fn(st.ctypes.data) == 0
NT = 17
def us(x): return 0.01 * x                   # 100 MHz
if B > 256:
    a, w3 = st[0], st[1]
    print("left-looking, B = %d: per row block (us): A = Schur complements, chain, sync, C = row solve + store, sync" % B)
    tot = np.zeros(5)
    for kb in range(NT):
        r = a[kb]
        ph = [us(r[1] - r[0]), us(r[2] - r[1]), us(r[3] - r[2]), us(r[4] - r[3]), us(r[5] - r[4])]
        tot += ph
        q = w3[kb]
        print("kb %2d  A %6.2f chain %5.2f sync %5.2f C %5.2f sync %5.2f | total %6.2f   wave 3: A %6.2f wait %6.2f C %5.2f"
              % (kb, *ph, us(r[5] - r[0]), us(q[1] - q[0]), us(q[3] - q[1]), us(q[4] - q[3])))
    print("sum   A %6.2f chain %5.2f sync %5.2f C %5.2f sync %5.2f | total %.2f us" % (*tot, us(a[NT - 1][5] - a[0][0])))
elif os.environ.get("BLSQ_CHOL_RL2", "1") != "0":
    c, wk, cr = st[2], st[3], st[1]
    print("right-looking, flag-driven, B = %d.  wave 0: wait for the diagonal tile, chain.  critical tiles: RINV flag -> "
          "(kb,kb+1) published -> its flag seen by the diagonal owner -> diagonal tile handed over -> seen by wave 0.  "
          "worker 2: critical duties, other row tiles + publish, wait for all published, trailing update, wait for RINV" % B)
    for kb in range(NT):
        r, q, x = c[kb], wk[kb], cr[kb]
        wait = us(r[0] - c[kb - 1][1]) if kb else 0.0
        nxt = c[kb + 1][0] if kb + 1 < NT else x[3]
        print("kb %2d  wave0: wait %5.2f chain %5.2f | critical: solve %5.2f handoff %5.2f diag %5.2f wake %5.2f | worker: duties %5.2f row %5.2f pub-wait %5.2f trail %6.2f rinv-wait %6.2f"
              % (kb, wait, us(r[1] - r[0]), us(x[1] - r[1]), us(x[2] - x[1]), us(x[3] - x[2]), us(nxt - x[3]),
                 us(q[1] - q[0]), us(q[2] - q[1]), us(q[3] - q[2]), us(q[4] - q[3]),
                 us(wk[kb + 1][0] - q[4]) if kb + 1 < NT else 0.0))
    print("loop %.2f us" % us(c[NT - 1][1] - c[0][0]))
    z, y = wk[17], c[17]
    print("worker: entry -> scales done %.2f  scaling of the tiles %.2f  barrier X %.2f  zero fill %.2f  | loop %.2f | final barrier %.2f  | whole body %.2f us"
          % (us(z[1] - z[0]), us(z[2] - z[1]), us(z[3] - z[2]), us(z[4] - z[3]), us(wk[18][0] - z[4]),
             us(wk[18][1] - wk[18][0]), us(wk[18][1] - z[0])))
    print("wave 0: entry -> scales done %.2f  to barrier X %.2f  zero fill %.2f | loop %.2f | whole body %.2f us"
          % (us(y[1] - y[0]), us(y[3] - y[1]), us(y[4] - y[3]), us(c[18][0] - y[4]), us(c[18][1] - y[0])))
else:
    c, wk = st[2], st[3]
    print("right-looking, B = %d: wave 0: wait for the diagonal tile, chain, barrier B, barrier C; worker 2: row solve, "
          "barrier C, diagonal update, trailing update, wait at B" % B)
    for kb in range(NT):
        r, q = c[kb], wk[kb]
        wait = us(r[0] - c[kb - 1][3]) if kb else 0.0
        print("kb %2d  wave0: wait %5.2f chain %5.2f B %5.2f C %5.2f | worker: solve %5.2f C-wait %5.2f diag %5.2f trail %6.2f B-wait %6.2f"
              % (kb, wait, us(r[1] - r[0]), us(r[2] - r[1]), us(r[3] - r[2]),
                 us(q[1] - q[0]), us(q[2] - q[1]), us(q[3] - q[2]), us(q[4] - q[3]),
                 us(wk[kb + 1][0] - q[4]) if kb + 1 < NT else 0.0))
    print("total %.2f us" % us(c[NT - 1][3] - c[0][0]))
    z, y = wk[17], c[17]
    print("worker: scales %.2f  tile loads %.2f  barrier X %.2f  zero fill %.2f  | loop %.2f | final barrier %.2f  | whole body %.2f us"
          % (us(z[1] - z[0]), us(z[2] - z[1]), us(z[3] - z[2]), us(z[4] - z[3]), us(wk[18][0] - z[4]),
             us(wk[18][1] - wk[18][0]), us(wk[18][1] - z[0])))
    print("wave 0: scales %.2f  to barrier X %.2f  zero fill %.2f | loop %.2f | whole body %.2f us"
          % (us(y[1] - y[0]), us(y[3] - y[1]), us(y[4] - y[3]), us(c[18][0] - y[4]), us(c[18][1] - y[0])))
