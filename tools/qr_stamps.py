"""Diagnostic: per-phase cycle shares of the QR kernel (needs `make -C bounded-lsq_amd/csrc diag`
and BLSQ_LIB=.../libblsq_hip_diag.so)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bounded-lsq_amd"))
from bounded_lsq import TrfStepSolver, _abi, _synth

B, m, n = int(sys.argv[1]) if len(sys.argv) > 1 else 64, 4096, 256
P = _synth.trf_batch(1, B, m, n)
ctx = _abi.Context(0)
sol = TrfStepSolver(B, m, n, ctx=ctx)
d = {k: ctx.to_device(P[k]) for k in ("J", "f", "x", "lb", "ub", "scale")}
nslot = B * 6
dbg = ctx.malloc(8 * 8 * nslot)
ctx.lib.blsq_debug_qr_stamps(dbg)
names = ["load", "apply", "fac:update", "emitR", "G+T", "spillV", "fac:dots+reduce", "fac:barrier"]
sol.factor_dev(d["J"], d["f"], d["x"], d["lb"], d["ub"], d["scale"]); ctx.sync()
a = ctx.to_host(dbg, (nslot, 8), np.float64)
off = 0
for label, slots in (("leaf", 4 * B), ("merge", B), ("aug", B)):
    x = a[off:off + slots]; off += slots
    tot = x.sum(1).mean()
    print("%-6s wgs %4d  mean cycles/wg %.4g  " % (label, slots, tot) +
          "  ".join("%s %.1f%%" % (nm, 100 * x[:, i].mean() / tot) for i, nm in enumerate(names) if nm != "-"))
