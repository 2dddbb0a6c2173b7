"""Device-resident batched outer drivers (blsq_outer_* of include/blsq.h).

`OuterDriver` keeps x, f, J and every per-problem scalar of B lock-step problems on the
GPU; the ratio test, the Delta / alpha updates, the termination tests and the accept step
(trf.py:238-261,309-358; dogbox.py:164-194,221-272) run there.  Between two user callbacks
only one integer crosses the boundary.

Two ways to feed the callbacks:
  * host callbacks (``run_host``): ``fun(X) -> (B, m)``, ``jac(X) -> (B, m, n)`` on numpy
    arrays; the driver copies x_trial down and f / J up (J only for accepted problems);
  * device callbacks (``run_device``): ``fun(x_ptr, f_ptr)``, ``jac(x_ptr, J_ptr, mask_ptr)``
    receive raw device pointers (e.g. wrapped as torch tensors by the caller) and fill the
    driver's buffers in place — nothing but the counters leaves the GPU.
"""
import ctypes as C

import numpy as np

from . import _abi
from ._abi import vp, ptr

METHODS = {"trf": 0, "dogbox": 1}


def raise_step_errors(status):
    """The reference raises ValueError out of the step (trust_region.py:28-29,34-35), which
    aborts its solve.  The device driver freezes such a problem with status = -BLSQ_STATUS_*;
    raise the same exception here, naming the first offending problem."""
    from ._hip_step import STATUS_MESSAGES
    bad = np.nonzero(np.asarray(status) < 0)[0]
    if bad.size:
        b = int(bad[0])
        code = -int(status[b])
        raise ValueError("problem %d: %s" % (b, STATUS_MESSAGES.get(code, "step status %d" % code)))


class OuterDriver:
    def __init__(self, method, B, m, n, ctx=None):
        if method not in METHODS:
            raise ValueError("`method` must be 'trf' or 'dogbox'.")
        self.ctx = ctx if ctx is not None else _abi.Context(0)
        self._own_ctx = ctx is None
        self.method, self.B, self.m, self.n = method, int(B), int(m), int(n)
        h = vp()
        self.ctx.check(self.ctx.lib.blsq_outer_create(self.ctx.h, METHODS[method], self.B, self.m,
                                                      self.n, C.byref(h)), "blsq_outer_create")
        self.h = h
        bufs = [vp() for _ in range(6)]
        self.ctx.check(self.ctx.lib.blsq_outer_buffers(self.h, *[C.byref(b) for b in bufs]),
                       "blsq_outer_buffers")
        (self.d_x, self.d_x_trial, self.d_f, self.d_f_trial, self.d_J, self.d_accepted) = bufs
        self.ctx.adopt(self)

    def close(self):
        h, self.h = getattr(self, "h", None), None
        ctx = self.ctx
        if h and ctx is not None and getattr(ctx, "h", None):
            ctx.lib.blsq_outer_destroy(h)
        if self._own_ctx and ctx is not None:
            self.ctx = None                      # (before ctx.close(): it walks its plans, this one included)
            ctx.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- raw protocol ------------------------------------------------------------------
    def start(self, x0, x_start, lb, ub, scale, jac_scaling, ftol, xtol, gtol, max_nfev):
        arrs = [np.ascontiguousarray(np.broadcast_to(a, (self.B, self.n)), dtype=np.float64)
                for a in (x0, x_start, lb, ub, scale)]
        self.ctx.check(self.ctx.lib.blsq_outer_start(
            self.h, *[ptr(a) for a in arrs], 1 if jac_scaling else 0, float(ftol), float(xtol),
            float(gtol), int(max_nfev)), "blsq_outer_start")

    def begin(self):
        self.ctx.check(self.ctx.lib.blsq_outer_begin(self.h), "blsq_outer_begin")

    def propose(self):
        c = C.c_int32(0)
        self.ctx.check(self.ctx.lib.blsq_outer_propose(self.h, C.byref(c)), "blsq_outer_propose")
        return c.value

    def judge(self):
        c = C.c_int32(0)
        self.ctx.check(self.ctx.lib.blsq_outer_judge(self.h, C.byref(c)), "blsq_outer_judge")
        return c.value

    def fetch(self):
        B, m, n = self.B, self.m, self.n
        out = dict(x=np.empty((B, n)), f=np.empty((B, m)), obj=np.empty(B), optimality=np.empty(B),
                   on_bound=np.empty((B, n), dtype=np.int64), nfev=np.empty(B, dtype=np.int32),
                   njev=np.empty(B, dtype=np.int32), status=np.empty(B, dtype=np.int32))
        self.ctx.check(self.ctx.lib.blsq_outer_fetch(
            self.h, *[ptr(out[k]) for k in ("x", "f", "obj", "optimality", "on_bound", "nfev",
                                           "njev", "status")]), "blsq_outer_fetch")
        return out

    # ---- helpers for host-side callbacks -----------------------------------------------
    def _up(self, dptr, arr, shape=None, what="callback"):
        arr = np.ascontiguousarray(arr, dtype=np.float64)
        if shape is not None and arr.shape != tuple(shape):      # never write past the buffer
            raise RuntimeError("`%s` must return an array of shape %s." % (what, tuple(shape)))
        self.ctx.check(self.ctx.lib.blsq_memcpy_h2d(self.ctx.h, dptr, ptr(arr), arr.nbytes), "h2d")

    def _down(self, dptr, shape, dtype=np.float64):
        return self.ctx.to_host(dptr, shape, dtype)

    def run_host(self, fun, jac):
        """Lock-step loop with numpy callbacks.  Requires start()."""
        B, m, n = self.B, self.m, self.n
        X = self._down(self.d_x, (B, n))
        F = np.ascontiguousarray(fun(X), dtype=np.float64)
        if F.shape != (B, m):
            raise RuntimeError("`fun` must return an array of shape (B, m).")
        J = np.ascontiguousarray(jac(X), dtype=np.float64)
        if J.shape != (B, m, n):
            raise RuntimeError("`jac` must return an array of shape (B, m, n).")
        self._up(self.d_f, F, (B, m), "fun")
        self._up(self.d_J, J, (B, m, n), "jac")
        self.begin()
        itemJ = m * n * 8
        while self.propose() > 0:
            Xt = self._down(self.d_x_trial, (B, n))
            self._up(self.d_f_trial, fun(Xt), (B, m), "fun")
            if self.judge() > 0:
                acc = self._down(self.d_accepted, (B,), np.int32)
                X = self._down(self.d_x, (B, n))
                Jn = np.ascontiguousarray(jac(X), dtype=np.float64)
                if Jn.shape != (B, m, n):
                    raise RuntimeError("`jac` must return an array of shape (B, m, n).")
                idx = np.nonzero(acc)[0]
                # only fresh Jacobians travel: contiguous runs of accepted problems, one copy each
                runs = np.split(idx, np.nonzero(np.diff(idx) > 1)[0] + 1)
                for r in runs:
                    b0, nb = int(r[0]), len(r)
                    dst = vp(self.d_J.value + b0 * itemJ)
                    self.ctx.check(self.ctx.lib.blsq_memcpy_h2d(self.ctx.h, dst, ptr(Jn[b0:b0 + nb]),
                                                                nb * itemJ), "h2d(J)")
        R = self.fetch()
        raise_step_errors(R["status"])
        return R

    def run_device(self, fun_dev, jac_dev, sync=None, rel_step=None, bounds_dev=None):
        """Lock-step loop with device callbacks:
        ``fun_dev(x_ptr, f_ptr, reps)`` fills f [B * reps][m] from x [B * reps][n]; the points of
        problem b are rows b * reps .. b * reps + reps - 1 (reps == 1 except for the
        finite-difference evaluations);
        ``jac_dev(x_ptr, J_ptr, accepted_ptr)`` fills J[b] for every b (accepted_ptr None: all
        problems) or at least those with accepted[b] != 0 — or one of the strings '2-point' /
        '3-point': the Jacobian is then estimated on the device from `fun_dev` (``FdJacobian``:
        scipy's approx_derivative restated for a batch; `rel_step` = diff_step, `bounds_dev` =
        (lb_ptr, ub_ptr) device arrays [B][n], needed for the bounds-aware steps).
        `sync()` (optional) must wait for the callbacks' own stream; the library's stream is idle
        whenever a callback runs."""
        sync = sync or (lambda: None)
        fd = None
        if isinstance(jac_dev, str):
            from ._fd import FdJacobian
            if bounds_dev is None:
                raise ValueError("finite differences need `bounds_dev` = (lb_ptr, ub_ptr).")
            fd = FdJacobian(self.ctx, self.B, self.m, self.n, jac_dev, rel_step)
            d_lb, d_ub = bounds_dev

            def jac_dev(x_ptr, J_ptr, mask_ptr, fd=fd):           # noqa: F811
                Xp = fd.points(x_ptr, d_lb, d_ub)
                self.ctx.sync()
                fun_dev(Xp, fd.d_F, fd.P)
                sync()
                fd.assemble(x_ptr, self.d_f, J_ptr, mask_ptr)
                self.ctx.sync()
        try:
            fun_dev(self.d_x, self.d_f, 1)
            sync()
            jac_dev(self.d_x, self.d_J, None)
            sync()
            self.begin()
            while self.propose() > 0:
                fun_dev(self.d_x_trial, self.d_f_trial, 1)
                sync()
                if self.judge() > 0:
                    jac_dev(self.d_x, self.d_J, self.d_accepted)
                    sync()
            R = self.fetch()
            raise_step_errors(R["status"])
            return R
        finally:
            if fd is not None:
                fd.close()
