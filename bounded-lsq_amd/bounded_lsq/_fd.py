"""Finite-difference Jacobians for B problems at once, on the device (blsq_fd_* of
include/blsq.h).

The reference's ``jac='2-point' | '3-point'`` is scipy's ``approx_derivative(fun, x,
rel_step=diff_step, method=jac, f0=f, bounds=bounds)`` (least_squares.py:357-365), one
problem and one column at a time.  `FdJacobian` restates it for a batch: one kernel chooses
the steps (bounds-aware, one-sided switching) and writes the perturbed points
``X [B][P][n]`` (P = n or 2n), the caller's ``fun`` evaluates them in ONE batched call, a
second kernel assembles ``J [B][m][n]``.
"""
import ctypes as C

import numpy as np

from ._abi import vp, ptr

FD_METHODS = {"2-point": 2, "3-point": 3}


class FdJacobian:
    def __init__(self, ctx, B, m, n, method="2-point", rel_step=None):
        if method not in FD_METHODS:
            raise ValueError("`jac` must be '2-point', '3-point' or callable.")
        self.ctx, self.B, self.m, self.n = ctx, int(B), int(m), int(n)
        self.method = FD_METHODS[method]
        self.P = self.n if self.method == 2 else 2 * self.n
        self.d_X = ctx.malloc(8 * self.B * self.P * self.n)
        self.d_F = ctx.malloc(8 * self.B * self.P * self.m)
        self.d_h = ctx.malloc(8 * self.B * self.n)
        self.d_os = ctx.malloc(self.B * self.n)
        self.d_rel = None
        if rel_step is not None:
            rs = np.ascontiguousarray(np.broadcast_to(np.asarray(rel_step, dtype=float), (self.n,)))
            self.d_rel = ctx.to_device(rs)

    def close(self):
        for name in ("d_X", "d_F", "d_h", "d_os", "d_rel"):
            p = getattr(self, name, None)
            if p is not None:
                self.ctx.free(p)
                setattr(self, name, None)

    def points(self, d_x, d_lb, d_ub):
        """-> device pointer of the perturbed points X [B][P][n] (x, lb, ub: device, [B][n])."""
        self.ctx.check(self.ctx.lib.blsq_fd_points_dev(
            self.ctx.h, self.B, self.n, self.method, d_x, d_lb, d_ub, self.d_rel, self.d_X,
            self.d_h, self.d_os), "blsq_fd_points_dev")
        return self.d_X

    def assemble(self, d_x, d_f0, d_J, d_mask=None):
        """J [B][m][n] <- from f0 [B][m] and self.d_F [B][P][m] (filled by the caller's fun)."""
        self.ctx.check(self.ctx.lib.blsq_fd_assemble_dev(
            self.ctx.h, self.B, self.m, self.n, self.method, d_x, self.d_h, self.d_os, d_f0,
            self.d_F, d_J, d_mask), "blsq_fd_assemble_dev")

    # ---- host convenience (tests, numpy callbacks) ---------------------------------------
    def jac_host(self, fun_points, X, F0, lb, ub):
        """numpy in / numpy out: fun_points(Xp (B, P, n)) -> (B, P, m)."""
        ctx = self.ctx
        B, n, m, P = self.B, self.n, self.m, self.P
        d = [ctx.to_device(np.ascontiguousarray(np.broadcast_to(a, (B, n)), dtype=float))
             for a in (X, lb, ub)]
        d_f0 = ctx.to_device(np.ascontiguousarray(F0, dtype=float))
        d_J = ctx.malloc(8 * B * m * n)
        try:
            self.points(*d)
            Xp = ctx.to_host(self.d_X, (B, P, n), np.float64)
            Fp = np.ascontiguousarray(fun_points(Xp), dtype=float)
            if Fp.shape != (B, P, m):
                raise RuntimeError("`fun` must return an array of shape (B, P, m).")
            ctx.check(ctx.lib.blsq_memcpy_h2d(ctx.h, self.d_F, ptr(Fp), Fp.nbytes), "h2d(F)")
            self.assemble(d[0], d_f0, d_J)
            return ctx.to_host(d_J, (B, m, n), np.float64)
        finally:
            for p in d + [d_f0, d_J]:
                ctx.free(p)
