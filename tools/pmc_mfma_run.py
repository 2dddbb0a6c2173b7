"""Workload for the SQ MFMA-counter passes (run under rocprofv3 --pmc ... --kernel-trace):
  1. the FP64 MFMA probe, one and two waves per SIMD  (mfma_f64_probe_kernel: a launch that
     executes a KNOWN number of v_mfma_f64_16x16x4_f64 back to back — the calibration point)
  2. one TRF step-solve of the bench workload (4096 x 256, 512 problems; PMC_B / PMC_M / PMC_N
     override the shape) per factorisation path
Prints what the probe launches executed so that tools/pmc_mfma.py can price the counters."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bounded-lsq_amd"))
import numpy as np  # noqa: E402
from bounded_lsq import TrfStepSolver, _abi, _synth  # noqa: E402

B, m, n = (int(os.environ.get(k, d)) for k, d in (("PMC_B", "512"), ("PMC_M", "4096"), ("PMC_N", "256")))
ctx = _abi.Context(0)
info = {}
for w in (1, 2):
    tf, nm, ms = ctx.probe("mfma_f64", w)
    info["probe_%dw" % w] = {"tflops": tf, "mfma_wave_insts_timed_launch": nm, "ms": ms,
                             "warmup_launch_insts": nm / 100.0}
P = _synth.trf_batch(10_000, B, m, n)
d = {k: ctx.to_device(P[k]) for k in ("J", "f", "x", "lb", "ub", "scale")}
dD = ctx.to_device(np.where(np.arange(B) % 2 == 0, 10.0, 0.5))
dA = ctx.to_device(np.zeros(B))
for gram in ("1", "0"):
    os.environ["BLSQ_GRAM"] = gram
    sol = TrfStepSolver(B, m, n, ctx=ctx)
    # three step-solves on the normal-equations path: the first launch of a kernel is cold (code
    # object load, clocks ramping, first touch of its buffers) and tools/pmc_mfma.py leaves it out
    for _ in range(3 if gram == "1" else 1):
        sol.factor_dev(d["J"], d["f"], d["x"], d["lb"], d["ub"], d["scale"])
        sol.step_dev(dD, dA)
        ctx.sync()
    sol.close()
print(json.dumps(info))
