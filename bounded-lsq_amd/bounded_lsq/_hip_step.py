"""Python face of the HIP step solvers (thin: argument marshalling only).

``TrfStepSolver`` / ``DogboxStepSolver`` own one plan each (fixed B, m, n) and
expose the two-call seam of the reference drivers:

    factor(...)   once per outer iteration   (trf.py:244-277 / dogbox.py:165-199)
    step(...)     once per inner iteration   (trf.py:284-308 / dogbox.py:203-220)

All arithmetic happens in libblsq_hip.so on the GPU.
"""
import ctypes as C
import threading
from collections import namedtuple

import numpy as np

from . import _abi
from ._abi import ptr, f64

SCALE_GIVEN, SCALE_JAC_INIT, SCALE_JAC_UPDATE = 0, 1, 2

TrfFactorOut = namedtuple("TrfFactorOut", "g g_norm theta scale")
TrfStepOut = namedtuple(
    "TrfStepOut", "alpha step_h step x_new hits active_new predicted_reduction "
    "step_h_norm correction n_iter branch status")
TrfStepDetail = namedtuple("TrfStepDetail", TrfStepOut._fields + ("p_h_tr", "to_bound", "choice"))
DogFactorOut = namedtuple("DogFactorOut", "g active_set g_norm all_active scale")
DogStepOut = namedtuple(
    "DogStepOut", "step x_new on_bound_new tr_hit predicted_reduction "
    "step_scaled_norm fallback status")

_default_ctx = {}


def default_context(device_id=0):
    c = _default_ctx.get(device_id)
    if c is None:
        c = _default_ctx[device_id] = _abi.Context(device_id)
    return c


# ---- plans of the sequential drivers: leased, not created per solve ---------------------------------------
# A plan (device buffers for one (B, m, n)) costs ~0.9 ms to create and ~0.3 ms to destroy — beside a 4096 x 256 solve
# of five iterations that is a fifth of the library's own time.  The drop-in drivers (`_drivers.trf / dogbox`) lease
# theirs from a small per-context pool: a solve of a shape seen before reuses the plan of the last one.  A plan's
# history decides which launches a call enqueues, never a number it returns (tests/test_drivers_gpu.py).
PLAN_POOL_KEEP = 4                     # plans a context keeps between solves (0: create / destroy per solve)
PLAN_POOL_MAX_BYTES = 64 << 20         # ... of problems up to this size of [J] only
_pool_lock = threading.Lock()


def lease_solver(cls, B, m, n, ctx=None):
    """A solver of class `cls` for (B, m, n) on `ctx`: one the pool holds, or a new one."""
    ctx = ctx or default_context()
    with _pool_lock:
        pool = ctx.__dict__.setdefault("_solver_pool", [])
        for i in range(len(pool) - 1, -1, -1):
            s = pool[i]
            if not s.h:                                        # (closed with its context, or by hand)
                del pool[i]
            elif type(s) is cls and (s.B, s.m, s.n) == (int(B), int(m), int(n)):
                return pool.pop(i)
    return cls(B, m, n, ctx=ctx)


def return_solver(s):
    """Back to its context's pool (the least recently returned plan leaves when the pool is full)."""
    ctx = s.ctx
    if (not s.h or not getattr(ctx, "h", None) or PLAN_POOL_KEEP <= 0
            or 8 * s.B * s.m * s.n > PLAN_POOL_MAX_BYTES):
        s.close()
        return
    with _pool_lock:
        pool = ctx.__dict__.setdefault("_solver_pool", [])
        pool.append(s)
        out, pool[:] = pool[:-PLAN_POOL_KEEP], pool[-PLAN_POOL_KEEP:]
    for o in out:
        o.close()


STATUS_MESSAGES = {1: "`s` is zero.",                            # trust_region.py:28-29
                   2: "`x` is not within the trust region."}     # trust_region.py:34-35


def _raise_status(status):
    """B == 1 drop-in behaviour: the reference raises ValueError here
    (trust_region.py:28-29,34-35)."""
    if status in STATUS_MESSAGES:
        raise ValueError(STATUS_MESSAGES[status])


def raise_batch_status(status, active=None):
    """Batched drivers: the reference would abort the solve of that problem with ValueError;
    raise it naming the first offending (still active) problem."""
    st = np.asarray(status)
    bad = np.nonzero((st != 0) & (True if active is None else np.asarray(active)))[0]
    if bad.size:
        b = int(bad[0])
        raise ValueError("problem %d: %s" % (b, STATUS_MESSAGES.get(int(st[b]), "step status %d" % st[b])))


class TrfStepSolver:
    def __init__(self, B, m, n, ctx=None):
        self.ctx = ctx or default_context()
        self.lib = self.ctx.lib
        self.B, self.m, self.n = int(B), int(m), int(n)
        h = _abi.vp()
        self.ctx.check(self.lib.blsq_trf_plan_create(self.ctx.h, self.B, self.m, self.n,
                                                     C.byref(h)), "blsq_trf_plan_create")
        self.h = h
        self.ctx.adopt(self)

    def close(self):
        if getattr(self, "h", None):
            if getattr(self.ctx, "h", None):         # (a closed ctx has closed its plans already)
                self.lib.blsq_trf_plan_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- host-pointer API ---------------------------------------------------
    def factor(self, J, f, x, lb, ub, scale, scale_mode=SCALE_GIVEN):
        B, m, n = self.B, self.m, self.n
        J = f64(J, (B, m, n)); f = f64(f, (B, m))
        x = f64(x, (B, n)); lb = f64(lb, (B, n)); ub = f64(ub, (B, n))
        scale = np.array(scale, dtype=np.float64).reshape(B, n).copy()
        g = np.empty((B, n)); g_norm = np.empty(B); theta = np.empty(B)
        self.ctx.check(self.lib.blsq_trf_factor(
            self.h, ptr(J), ptr(f), ptr(x), ptr(lb), ptr(ub), ptr(scale), int(scale_mode),
            ptr(g), ptr(g_norm), ptr(theta)), "blsq_trf_factor")
        return TrfFactorOut(g, g_norm, theta, scale)

    def step(self, Delta, alpha, active_rtol=1e-8):
        B, n = self.B, self.n
        Delta = f64(np.broadcast_to(np.asarray(Delta, float), (B,)))
        alpha = np.array(np.broadcast_to(np.asarray(alpha, float), (B,)), dtype=np.float64)
        step_h = np.empty((B, n)); step = np.empty((B, n)); x_new = np.empty((B, n))
        hits = np.empty((B, n), np.int64); act = np.empty((B, n), np.int64)
        pred = np.empty(B); shn = np.empty(B); corr = np.empty(B)
        n_iter = np.empty(B, np.int32); branch = np.empty(B, np.int32)
        status = np.empty(B, np.int32)
        self.ctx.check(self.lib.blsq_trf_step(
            self.h, ptr(Delta), ptr(alpha), float(active_rtol), ptr(step_h), ptr(step),
            ptr(x_new), ptr(hits), ptr(act), ptr(pred), ptr(shn), ptr(corr), ptr(n_iter),
            ptr(branch), ptr(status)), "blsq_trf_step")
        if B == 1:
            _raise_status(int(status[0]))
        return TrfStepOut(alpha, step_h, step, x_new, hits, act, pred, shn, corr, n_iter,
                          branch, status)

    # ---- device-resident API (inputs already in HBM) ------------------------
    def factor_dev(self, dJ, df, dx, dlb, dub, dscale, scale_mode=SCALE_GIVEN):
        self.ctx.check(self.lib.blsq_trf_factor_dev(self.h, dJ, df, dx, dlb, dub, dscale,
                                                    int(scale_mode)), "blsq_trf_factor_dev")

    def step_dev(self, dDelta, dalpha, active_rtol=1e-8):
        self.ctx.check(self.lib.blsq_trf_step_dev(self.h, dDelta, dalpha, float(active_rtol)),
                       "blsq_trf_step_dev")

    def fetch_factor(self, want_singular=False):
        B, n = self.B, self.n
        g = np.empty((B, n)); g_norm = np.empty(B); theta = np.empty(B)
        scale = np.empty((B, n))
        sing = np.empty((B, n)) if want_singular else None
        self.ctx.check(self.lib.blsq_trf_fetch_factor(self.h, ptr(g), ptr(g_norm), ptr(theta),
                                                      ptr(scale), ptr(sing)),
                       "blsq_trf_fetch_factor")
        out = TrfFactorOut(g, g_norm, theta, scale)
        return (out, sing) if want_singular else out

    def debug_fast(self):
        fl = np.empty(self.B, np.int32)
        self.ctx.check(self.lib.blsq_trf_debug_fast(self.h, ptr(fl)), "blsq_trf_debug_fast")
        return fl

    def debug_cond(self):
        """Proven bound K2 >= kappa_2 of the equilibrated system of the last factor call (0: none)."""
        k2 = np.empty(self.B)
        self.ctx.check(self.lib.blsq_trf_debug_cond(self.h, ptr(k2)), "blsq_trf_debug_cond")
        return k2

    def debug_sweeps(self):
        sw = np.empty(self.B, np.int32)
        self.ctx.check(self.lib.blsq_trf_debug_sweeps(self.h, ptr(sw)), "blsq_trf_debug_sweeps")
        return sw

    def debug_csne(self):
        """-> (on_tier [B] int32, eta [B]): the CSNE tier's problems and the largest first-order correction the last
        step call measured for each (-1: the tier declined the problem in that call)."""
        on = np.empty(self.B, np.int32)
        eta = np.empty(self.B)
        self.ctx.check(self.lib.blsq_trf_debug_csne(self.h, ptr(on), ptr(eta)), "blsq_trf_debug_csne")
        return on, eta

    def fetch_step(self):
        B, n = self.B, self.n
        alpha = np.empty(B)
        step_h = np.empty((B, n)); step = np.empty((B, n)); x_new = np.empty((B, n))
        hits = np.empty((B, n), np.int64); act = np.empty((B, n), np.int64)
        pred = np.empty(B); shn = np.empty(B); corr = np.empty(B)
        n_iter = np.empty(B, np.int32); branch = np.empty(B, np.int32)
        status = np.empty(B, np.int32); p_h_tr = np.empty((B, n)); to_bound = np.empty(B)
        choice = np.empty(B, np.int32)
        self.ctx.check(self.lib.blsq_trf_fetch_step(
            self.h, ptr(alpha), ptr(step_h), ptr(step), ptr(x_new), ptr(hits), ptr(act),
            ptr(pred), ptr(shn), ptr(corr), ptr(n_iter), ptr(branch), ptr(status),
            ptr(p_h_tr), ptr(to_bound), ptr(choice)), "blsq_trf_fetch_step")
        return TrfStepDetail(alpha, step_h, step, x_new, hits, act, pred, shn, corr, n_iter,
                             branch, status, p_h_tr, to_bound, choice)


class DogboxStepSolver:
    def __init__(self, B, m, n, ctx=None):
        self.ctx = ctx or default_context()
        self.lib = self.ctx.lib
        self.B, self.m, self.n = int(B), int(m), int(n)
        h = _abi.vp()
        self.ctx.check(self.lib.blsq_dogbox_plan_create(self.ctx.h, self.B, self.m, self.n,
                                                        C.byref(h)), "blsq_dogbox_plan_create")
        self.h = h
        self.ctx.adopt(self)

    def close(self):
        if getattr(self, "h", None):
            if getattr(self.ctx, "h", None):
                self.lib.blsq_dogbox_plan_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def factor(self, J, f, x, lb, ub, scale, on_bound, scale_mode=SCALE_GIVEN):
        B, m, n = self.B, self.m, self.n
        J = f64(J, (B, m, n)); f = f64(f, (B, m))
        x = f64(x, (B, n)); lb = f64(lb, (B, n)); ub = f64(ub, (B, n))
        scale = np.array(scale, dtype=np.float64).reshape(B, n).copy()
        ob = np.ascontiguousarray(on_bound, dtype=np.int64).reshape(B, n)
        g = np.empty((B, n)); act = np.empty((B, n), np.uint8); g_norm = np.empty(B)
        alla = np.empty(B, np.int32)
        self.ctx.check(self.lib.blsq_dogbox_factor(
            self.h, ptr(J), ptr(f), ptr(x), ptr(lb), ptr(ub), ptr(scale), int(scale_mode),
            ptr(ob), ptr(g), ptr(act), ptr(g_norm), ptr(alla)), "blsq_dogbox_factor")
        return DogFactorOut(g, act, g_norm, alla, scale)

    def step(self, Delta):
        B, n = self.B, self.n
        Delta = f64(np.broadcast_to(np.asarray(Delta, float), (B,)))
        step = np.empty((B, n)); x_new = np.empty((B, n)); obn = np.empty((B, n), np.int64)
        tr_hit = np.empty(B, np.uint8); pred = np.empty(B); ssn = np.empty(B)
        fb = np.empty(B, np.uint8); status = np.empty(B, np.int32)
        self.ctx.check(self.lib.blsq_dogbox_step(
            self.h, ptr(Delta), ptr(step), ptr(x_new), ptr(obn), ptr(tr_hit), ptr(pred),
            ptr(ssn), ptr(fb), ptr(status)), "blsq_dogbox_step")
        return DogStepOut(step, x_new, obn, tr_hit, pred, ssn, fb, status)

    def debug_cond(self):
        k2 = np.empty(self.B)
        self.ctx.check(self.lib.blsq_dogbox_debug_cond(self.h, ptr(k2)), "blsq_dogbox_debug_cond")
        return k2

    def factor_dev(self, dJ, df, dx, dlb, dub, dscale, don_bound, scale_mode=SCALE_GIVEN):
        self.ctx.check(self.lib.blsq_dogbox_factor_dev(self.h, dJ, df, dx, dlb, dub, dscale,
                                                       int(scale_mode), don_bound),
                       "blsq_dogbox_factor_dev")

    def step_dev(self, dDelta):
        self.ctx.check(self.lib.blsq_dogbox_step_dev(self.h, dDelta), "blsq_dogbox_step_dev")

    def fetch_factor(self, want_steps=False):
        B, n = self.B, self.n
        g = np.empty((B, n)); act = np.empty((B, n), np.uint8); g_norm = np.empty(B)
        alla = np.empty(B, np.int32); scale = np.empty((B, n))
        nw = np.empty((B, n)) if want_steps else None
        ca = np.empty((B, n)) if want_steps else None
        self.ctx.check(self.lib.blsq_dogbox_fetch_factor(
            self.h, ptr(g), ptr(act), ptr(g_norm), ptr(alla), ptr(scale), ptr(nw), ptr(ca)),
            "blsq_dogbox_fetch_factor")
        out = DogFactorOut(g, act, g_norm, alla, scale)
        return (out, nw, ca) if want_steps else out

    def fetch_step(self):
        B, n = self.B, self.n
        step = np.empty((B, n)); x_new = np.empty((B, n)); obn = np.empty((B, n), np.int64)
        tr_hit = np.empty(B, np.uint8); pred = np.empty(B); ssn = np.empty(B)
        fb = np.empty(B, np.uint8); status = np.empty(B, np.int32)
        self.ctx.check(self.lib.blsq_dogbox_fetch_step(
            self.h, ptr(step), ptr(x_new), ptr(obn), ptr(tr_hit), ptr(pred), ptr(ssn),
            ptr(fb), ptr(status)), "blsq_dogbox_fetch_step")
        return DogStepOut(step, x_new, obn, tr_hit, pred, ssn, fb, status)
