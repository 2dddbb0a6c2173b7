"""Calibration of the conditioning certificate (gram_cond_kernel): for families of Jacobians with
known spectra print the PROVEN bound K2 >= kappa_2(C), the true kappa_2 of the equilibrated system,
which path the problem took, and the step error of BOTH paths against the oracle.
usage: python tests/gate_calib.py   (GPU box; lives under tests/: it drives the CPU oracle)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bounded-lsq_amd")); sys.path.insert(0, ROOT)
import bounded_lsq as bl  # noqa: E402
from bounded_lsq import _synth, _abi  # noqa: E402
from oracle import blsq_oracle as orc  # noqa: E402


def run(P, Delta, gram):
    os.environ["BLSQ_GRAM"] = gram
    B, m, n = P["J"].shape
    ctx = _abi.Context(0)
    sol = bl.TrfStepSolver(B, m, n, ctx=ctx)
    ctx.gram_stats(reset=True)
    sol.factor(P["J"], P["f"], P["x"], P["lb"], P["ub"], P["scale"])
    st = ctx.gram_stats()
    k2 = sol.debug_cond()
    S = sol.step(Delta, np.zeros(B))
    sol.close(); ctx.close()
    return S.step.copy(), k2, st


def report(tag, P, Delta):
    B = P["J"].shape[0]
    assert B == 1
    s1, k2, st = run(P, Delta, "1")
    s0, _, _ = run(P, Delta, "0")
    for b in range(B):
        _, So = orc.trf_step_solve(P["J"][b], P["f"][b], P["x"][b], P["lb"][b], P["ub"][b],
                                   P["scale"][b], Delta[b], 0.0)
        Jn = P["J"][b] / np.linalg.norm(P["J"][b], axis=0)
        sv = np.linalg.svd(Jn, compute_uv=False)
        den = np.linalg.norm(So.step)
        e1 = np.linalg.norm(s1[b] - So.step) / den
        e0 = np.linalg.norm(s0[b] - So.step) / den
        print("%-28s K2 %.3e  true kappa2(C) %.3e  ratio %6.1f  %s  err(gram-path build) %.1e  err(tree) %.1e"
              % (tag, k2[b], (sv[0] / sv[-1]) ** 2, k2[b] / (sv[0] / sv[-1]) ** 2,
                 "GRAM" if st == (1, 0) else "tree", e1, e0)
              + "  c = err / (eps kappa2) = %.3f" % (e1 / (np.finfo(float).eps * (sv[0] / sv[-1]) ** 2)))


def equicorr(B, m, n, rho, seed):
    rng = np.random.default_rng(seed)
    Z = rng.standard_normal((B, m, n)); c = rng.standard_normal((B, m, 1))
    return np.sqrt(1 - rho) * Z + np.sqrt(rho) * c


rng = np.random.default_rng(0)
for (m, n) in ((4096, 256), (512, 64), (2048, 128), (300, 255), (600, 256)):
    P = _synth.trf_batch(1, 1, m, n, unbounded=True)
    report("gaussian %dx%d" % (m, n), P, np.array([0.5]))
for rho in (0.5, 0.9, 0.96, 0.98, 0.99, 0.995, 0.999):
    for n in (48, 256):
        P = _synth.trf_batch(2, 1, 2048, n, unbounded=True)
        P["J"] = equicorr(1, 2048, n, rho, 5)
        report("equicorr rho=%g n=%d" % (rho, n), P, np.array([0.5]))
for kappa in (2, 10, 30, 100, 300, 1e3, 1e4):
    m, n = 1200, 80
    P = _synth.trf_batch(3, 1, m, n, unbounded=True)
    U, _ = np.linalg.qr(rng.standard_normal((m, n))); V, _ = np.linalg.qr(rng.standard_normal((n, n)))
    P["J"][0] = (U * np.logspace(0, -np.log10(kappa), n)) @ V.T
    report("logspaced kappa=%g" % kappa, P, np.array([0.5]))
for n, s_ in ((32, 0.95), (64, 0.97), (128, 0.992)):
    c_ = np.sqrt(1 - s_ ** 2)
    K = np.zeros((n, n))
    for i in range(n):
        K[i, i] = s_ ** i
        K[i, i + 1:] = -c_ * s_ ** i
    P = _synth.trf_batch(4, 1, 1024, n, unbounded=True)
    Q, _ = np.linalg.qr(rng.standard_normal((1024, n)))
    P["J"][0] = Q @ K
    report("kahan n=%d s=%g" % (n, s_), P, np.array([0.5]))
