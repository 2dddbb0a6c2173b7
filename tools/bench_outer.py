"""End-to-end batched solves: host lock-step driver vs device-resident outer driver
(SURVEY.md 8f-1).  B bounded problems  min ||A_b tanh(x) - y_b||^2,  lb <= x <= ub  of one shape;
callbacks are USER code — numpy on the host, or torch on the GPU writing straight into the driver's
device buffers (zero copy).  Prints solves/s for each combination.

usage: python tools/bench_outer.py [B m n]
"""
import os, sys, time
import numpy as np
import torch                      # user-side (callbacks); imported first so ONE HIP runtime is loaded
torch.cuda.init()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bounded-lsq_amd"))
from bounded_lsq import least_squares_batch, OuterDriver, _abi          # noqa: E402
from bounded_lsq._hostmath import shift_into_interior                    # noqa: E402

B, m, n = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (2048, 512, 64)
rng = np.random.default_rng(0)
A = rng.standard_normal((B, m, n)) / np.sqrt(n)
xt = rng.uniform(-1.0, 1.0, (B, n))
Y = np.einsum('bmn,bn->bm', A, np.tanh(xt)) + 1e-3 * rng.standard_normal((B, m))
X0 = np.zeros((B, n))
lb, ub = np.full(n, -0.8), np.full(n, 0.8)


def fun(X):
    return np.einsum('bmn,bn->bm', A, np.tanh(X)) - Y


def jac(X):
    return A * (1.0 - np.tanh(X) ** 2)[:, None, :]


def report(tag, res_or_R, dt):
    if isinstance(res_or_R, dict):
        nfev, st = res_or_R["nfev"], res_or_R["status"]
    else:
        nfev = np.array([r.nfev for r in res_or_R]); st = np.array([r.status for r in res_or_R])
    print("%-34s %8.1f solves/s  (%.3f s; mean nfev %.1f, statuses %s)" % (
        tag, B / dt, dt, nfev.mean(), dict(zip(*np.unique(st, return_counts=True)))), flush=True)


ctx = _abi.Context(0)
for method in ("trf", "dogbox"):
    t0 = time.perf_counter()
    r_host = least_squares_batch(fun, X0, jac, bounds=(lb, ub), method=method, ctx=ctx)
    report(method + ": host driver, numpy callbacks", r_host, time.perf_counter() - t0)
    t0 = time.perf_counter()
    r_dev = least_squares_batch(fun, X0, jac, bounds=(lb, ub), method=method, ctx=ctx, driver='device')
    report(method + ": device driver, numpy callbacks", r_dev, time.perf_counter() - t0)
    bad = [i for i, (a, b) in enumerate(zip(r_host, r_dev))
           if (a.nfev, a.njev, a.status) != (b.nfev, b.njev, b.status)]
    print("   problems whose (nfev, njev, status) differ host vs device driver: %d of %d %s" % (
        len(bad), B, [((r_host[i].nfev, r_host[i].njev, r_host[i].status),
                       (r_dev[i].nfev, r_dev[i].njev, r_dev[i].status),
                       float(np.abs(r_host[i].x - r_dev[i].x).max())) for i in bad[:4]]))

    class _Dev:                                   # raw device pointer -> torch tensor, zero copy
        def __init__(self, p, shape):
            self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": "<f8",
                                             "data": (int(p.value), False), "version": 2}

    def wrap(p, shape):
        return torch.as_tensor(_Dev(p, shape), device="cuda")

    At = torch.as_tensor(A, device="cuda"); Yt = torch.as_tensor(Y, device="cuda")

    def fun_dev(xp, fp, reps):                   # reps points per problem (FD evaluations)
        x = wrap(xp, (B, reps, n)); f = wrap(fp, (B, reps, m))
        torch.baddbmm(-Yt.unsqueeze(1).expand(B, reps, m), torch.tanh(x), At.transpose(1, 2), out=f)

    def jac_dev(xp, Jp, mask):
        x = wrap(xp, (B, n)); J = wrap(Jp, (B, m, n))
        torch.mul(At, (1.0 - torch.tanh(x) ** 2).unsqueeze(1), out=J)

    xs = np.stack([shift_into_interior(X0[b], lb, ub, rstep=1e-10) for b in range(B)]) \
        if method == "trf" else X0
    with OuterDriver(method, B, m, n, ctx=ctx) as drv:
        t0 = time.perf_counter()
        drv.start(X0, xs, lb, ub, np.ones(n), False, 1.49e-8, 1.49e-8, 1.49e-8, 100 * n)
        R = drv.run_device(fun_dev, jac_dev, sync=torch.cuda.synchronize)
        report(method + ": device driver, torch callbacks", R, time.perf_counter() - t0)
    print("   nfev differs from the host driver for %d problems" % int(
        (R["nfev"] != np.array([r.nfev for r in r_host])).sum()))
    print("   max |x_dev - x_host| = %.2e" % np.abs(R["x"] - np.array([r.x for r in r_host])).max())
    lbt = torch.as_tensor(np.broadcast_to(lb, (B, n)).copy(), device="cuda")
    ubt = torch.as_tensor(np.broadcast_to(ub, (B, n)).copy(), device="cuda")
    import ctypes
    with OuterDriver(method, B, m, n, ctx=ctx) as drv:
        t0 = time.perf_counter()
        drv.start(X0, xs, lb, ub, np.ones(n), False, 1.49e-8, 1.49e-8, 1.49e-8, 100 * n)
        R2 = drv.run_device(fun_dev, '2-point', sync=torch.cuda.synchronize,
                            bounds_dev=(ctypes.c_void_p(lbt.data_ptr()), ctypes.c_void_p(ubt.data_ptr())))
        report(method + ": device driver, torch fun, FD jac", R2, time.perf_counter() - t0)
    print("   max |x_fd - x_analytic| = %.2e" % np.abs(R2["x"] - R["x"]).max())
