#!/usr/bin/env python3
"""bench.py — trust-region step-solves/second on batched dense Jacobians (MI355X).

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N > 1 is launched by torch.distributed.run, one rank per GPU.  Batches are sharded by problem
  (independent problems, NO data-path collective, weak scaling: per-GPU batch fixed); the tall
  problem of config 5 is split by rows and uses the library's own RCCL collective.  torch is
  used only for the barrier and the max-over-ranks of the elapsed time; the compute path is
  libblsq_hip.so through ctypes.

One "step" = one step-solve (trf.py:244-308 resp. dogbox.py:170-220 = factor + one
inner step) for every problem of the per-GPU batch, inputs resident in HBM.

--config selects the BASELINE.json workload (default c2, the one the metric is quoted on):
    c2         TRF, 4096 x 256, 512 problems per GPU ("batched 4096x256 dense Jacobians at 1 GPU")
    c2-single  TRF, 4096 x 256, one problem          (configs[1] as written)
    c3         dogbox, 512 x 64, 1024 problems per GPU (configs[2])
    c4         TRF, 512 x 64, 1024 problems per GPU  (configs[3]: 8192 over 8 GPUs)
    c5         TRF, one tall problem, 250 000 x 128 rows PER RANK (configs[4]: 2 000 000 x 128 over
               8 GPUs), blsq_tsqr_factor_dev: local Gram + ncclAllReduce over RCCL
`python bench.py --gpus N` with N > 1 and no launcher around it (WORLD_SIZE unset) starts the N ranks itself — as a
child `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...` before this process
has touched the GPU — and relays rank 0's line and a non-zero exit code of any rank.
`roofline` prices a dominant MFMA / streaming kernel (gram, qr_leaf, csne_pass) with the work of its own algorithm; where
the dominant slot is a latency-bound n-space kernel (the 512 x 64 shapes, one problem) it carries the WHOLE STEP's
fractions instead (`scope: whole_step`): algorithmic bytes / ms_per_step against HBM, the Gram's flops / ms_per_step
against the FP64 MFMA peak — figures the committed kernel stats reproduce.
Every config prints the same JSON shape with its own SURVEY 8(d) bytes / flops.  The default (c2)
line also carries, at N = 1: `side_configs` (the other configs' rates, same process, short runs),
`householder_only` (the same workload with the normal-equations front end off), `h2d_inclusive`
(the host-pointer API: numpy in, numpy out) and `cpu_baseline`; at N > 1 a `c5_tsqr` side figure
(the tall problem over all ranks).  Side figures are never `value`.
"""
import argparse
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "bounded-lsq_amd"))

import numpy as np  # noqa: E402

PEAK_FP64_TFLOPS = 78.6     # MI355X FP64 vector = matrix peak (nominal; AMD spec)
PEAK_HBM_GBS = 8000.0       # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s measured copy)

CONFIGS = {
    "c2": dict(kind="trf", m=4096, n=256, batch=512,
               label="C2 batched: TRF exact step-solve, m=4096 n=256"),
    "c2-single": dict(kind="trf", m=4096, n=256, batch=1,
                      label="C2 single problem: TRF exact step-solve, m=4096 n=256"),
    "c3": dict(kind="dogbox", m=512, n=64, batch=1024,
               label="C3: dogbox dogleg step-solve, m=512 n=64"),
    "c4": dict(kind="trf", m=512, n=64, batch=1024,
               label="C4 (per-GPU share of 8192): TRF exact step-solve, m=512 n=64"),
    # (the headline shape through dogbox: a side leg only — `certificate_rejected_dogbox`)
    "c2-dogbox": dict(kind="dogbox", m=4096, n=256, batch=512,
                      label="C2 shape through dogbox: dogleg step-solve, m=4096 n=256"),
    "c5": dict(kind="tsqr", m=250_000, n=128, batch=1,
               label="C5: one tall TRF problem split by rows, 250000 x 128 per rank"),
}


# ---- algorithmic work per step-solve ------------------------------------------
def alg_bytes(kind, m, n):
    b = 8 * (m * n + m + 4 * n) + 8 * 2 * n + 8 * n
    return b + (8 * n if kind == "dogbox" else 0)


def alg_flops(kind, m, n):
    if kind == "dogbox":
        return 2 * m * n * n - 2.0 * n ** 3 / 3.0 + 10 * m * n + n * n
    return 2 * (m + n) * n * n + 11 * n ** 3 + 6 * (m + n) * n


def leaf_flops(m, n, rows_per_leaf=1024):
    """Householder QR of the [J f] row blocks (2 r N^2 - 2/3 N^3 per r x N leaf, N = n + 1)."""
    N = n + 1
    nleaf = max(1, -(-m // rows_per_leaf))
    r = -(-m // nleaf)
    return nleaf * (2.0 * r * N * N - 2.0 * N ** 3 / 3.0)


def gram_flops(m, n):
    """The symmetric product [J f]^T [J f]: one multiply-add per row and entry of the upper triangle."""
    N = n + 1
    return float(m) * N * (N + 1)


def gram_bytes(m, n):
    return 8.0 * (m * (n + 1) + (n + 1) * (n + 2) / 2)


def measured_traffic(cfg_name, kernel, m, n, B):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC summary, when it was
    collected for this exact workload (bench.py itself cannot collect PMC counters).
    -> (bytes or None, source string or None)"""
    fname = "hbm_traffic_latest.json" if cfg_name == "c2" else "hbm_traffic_%s.json" % cfg_name
    path = os.path.join(ROOT, "profiles", fname)
    try:
        with open(path) as fh:
            t = json.load(fh)
        cfg = t.get("config", {})
        if (cfg.get("m"), cfg.get("n"), cfg.get("batch")) != (m, n, B):
            return None, None
        dom = t.get("dominant")
        if dom and dom.get("slot") == kernel:
            return dom["hbm_bytes_per_launch"], "profiles/%s (%s; separate rocprofv3 --pmc FETCH_SIZE / " \
                "WRITE_SIZE passes of the same workload, gfx950 FETCH x2 correction)" % (fname, t.get("tag", "committed"))
    except Exception:
        pass
    return None, None


def make_deltas(kind, B):
    """TRF: half 'reflective' (Delta=10: Gauss-Newton step, reflection branch) and half 'feasible'
    (Delta=0.5: More' iterations) as SURVEY.md 8(d) asks; dogbox: Delta = 0.02."""
    if kind == "dogbox":
        return np.full(B, 0.02)
    return np.where(np.arange(B) % 2 == 0, 10.0, 0.5)


# ----------------------------------------------------------------- CPU leg --
_CPU = {}


def _cpu_one(b):
    from oracle import blsq_oracle as orc
    P = _CPU["P"]
    orc.trf_step_solve(P["J"][b], P["f"][b], P["x"][b], P["lb"][b], P["ub"][b], P["scale"][b],
                       float(_CPU["Delta"][b]), 0.0)
    return b


def _cpu_worker_init():
    try:
        from threadpoolctl import threadpool_limits
        _CPU["limit"] = threadpool_limits(limits=1)
    except Exception:
        pass


def _host_info():
    info = {"cpu": "", "logical_cores": os.cpu_count() or 1}
    try:
        info["affinity"] = len(os.sched_getaffinity(0))
    except Exception:
        info["affinity"] = info["logical_cores"]
    try:
        phys = set()
        pid = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and not info["cpu"]:
                info["cpu"] = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                pid = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
                phys.add((pid, core))
        info["physical_cores"] = len(phys) or info["logical_cores"]
    except Exception:
        info["physical_cores"] = info["logical_cores"]
    try:                                        # cgroup v2 CPU quota of this container, if any
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        info["cgroup_cpu_quota"] = None if q == "max" else float(q) / float(per)
    except Exception:
        info["cgroup_cpu_quota"] = None
    return info


def cpu_baseline(P, Delta, budget_s=24.0):
    """The reference's CPU path (numpy/scipy restatement, oracle/) on the host cores, four
    threading configurations; the best is `value`.  Runs BEFORE the GPU is
    initialised (fork).  Process pools: the box's one-GPU share of the host (16 workers) AND the
    host's physical cores (capped by what this process may run on)."""
    import multiprocessing as mp
    from threadpoolctl import threadpool_limits, threadpool_info
    host = _host_info()
    share = min(host["affinity"], int(os.environ.get("BLSQ_CPU_WORKERS", "16")))
    allc = min(host["affinity"], host["physical_cores"],
               int(os.environ.get("BLSQ_CPU_WORKERS_ALL", "128")))
    _CPU["P"] = P
    _CPU["Delta"] = Delta
    B = P["J"].shape[0]
    _cpu_one(0)                                   # warm-up (imports, page faults)
    t0 = time.perf_counter()
    with threadpool_limits(limits=1):
        _cpu_one(0)
    t1 = time.perf_counter() - t0                 # single-thread cost of one solve
    per_mode = budget_s / 4.0
    res = {}
    k = int(max(2, min(B, per_mode / max(t1, 1e-6))))
    with threadpool_limits(limits=1):             # (ii) one BLAS thread, sequential loop
        t0 = time.perf_counter()
        for b in range(k):
            _cpu_one(b % B)
        res["blas1_loop"] = (k / (time.perf_counter() - t0), 1, k)
    t0 = time.perf_counter()                      # (i) default BLAS threads, sequential loop
    for b in range(k):
        _cpu_one(b % B)
    nthr = max([d.get("num_threads", 1) for d in threadpool_info()] + [1])
    res["blas_default_loop"] = (k / (time.perf_counter() - t0), nthr, k)
    ctx = mp.get_context("fork")
    for label, nw in (("procs_x_blas1_gpu_share", share), ("procs_x_blas1_physical_cores", allc)):
        if nw <= 1 or (label.endswith("physical_cores") and nw <= share):
            continue
        k = int(max(nw, min(8 * B, nw * per_mode / max(t1, 1e-6))))
        with ctx.Pool(nw, initializer=_cpu_worker_init) as pool:
            pool.map(_cpu_one, [b % B for b in range(nw)])       # warm the workers
            t0 = time.perf_counter()
            pool.map(_cpu_one, [b % B for b in range(k)], chunksize=1)
            res[label] = (k / (time.perf_counter() - t0), nw, k)
    best = max(res, key=lambda kk: res[kk][0])
    return {
        "value": res[best][0], "unit": "step-solves/s", "cores": res[best][1],
        "kind": "port",
        "sample": "%d TRF step-solves of the same seeded batch (mode %s; oracle/blsq_oracle.py "
                  "= scipy.linalg.svd(gesdd) path of trf.py:244-308)" % (res[best][2], best),
        "modes": {kk: {"value": v[0], "threads": v[1], "solves": v[2]} for kk, v in res.items()},
        "host": host,
    }


# ---- mutations of the headline batch for the conditioning legs -------------------------------
def mut_mixed(P):
    """kappa(J) log-uniform over [1, 1e4] (J = Z V diag(s) V^T, Z Gaussian, s log-spaced over [1/kappa_b, 1]; 64
    distinct problems tiled to the batch)"""
    rng = np.random.default_rng(4242)
    nn = P["J"].shape[2]
    V, _ = np.linalg.qr(rng.standard_normal((nn, nn)))
    K = min(64, P["J"].shape[0])
    kap = 10.0 ** rng.uniform(0.0, 4.0, K)
    for b in range(K):
        sv = np.logspace(0.0, -np.log10(kap[b]), nn)
        P["J"][b] = (P["J"][b] @ (V * sv)) @ V.T
    for b in range(K, P["J"].shape[0]):
        P["J"][b] = P["J"][b % K]
    P["kappa"] = kap


def mut_mixed_unbounded(P):
    mut_mixed(P)
    P["lb"][:] = -np.inf
    P["ub"][:] = np.inf


def mut_all_rejected(P):
    """every problem beyond the gate: kappa(J) = 3e3, no bounds — what a batch gets whose problems ALL fail the
    certificate (`householder_only` is the tree alone)"""
    rng = np.random.default_rng(4243)
    nn = P["J"].shape[2]
    V, _ = np.linalg.qr(rng.standard_normal((nn, nn)))
    K = min(64, P["J"].shape[0])
    sv = np.logspace(0.0, -np.log10(3e3), nn)
    for b in range(K):
        P["J"][b] = (P["J"][b] @ (V * sv)) @ V.T
    for b in range(K, P["J"].shape[0]):
        P["J"][b] = P["J"][b % K]
    P["lb"][:] = -np.inf
    P["ub"][:] = np.inf


def conditioning_leg(key, mut, ctx, name, batch, steps, check):
    """one of the conditioning legs -> its record (a side figure, never `value`)"""
    bm = Bench(name, ctx, 0, 1, batch=batch, mutate=mut)
    B = bm.B
    try:
        km = max(2, min(steps, 20))
        ctx.csne_stats(reset=True)
        em, km = time_steps(bm, km, 1, ctx.sync)
        gsm = ctx.gram_stats()
        cq2 = ctx.cqr2_stats(reset=True)
        cs = ctx.csne_stats(reset=True)
        ncs = cs[0] // (km + 1 + bm.profile[1])          # (routed per factor call: warm-up + profile + timed regions)
        rec = {
            "value": B * km / em, "unit": "step-solves/s", "ms_per_step": 1e3 * em / km, "steps": km,
            "factorisation_paths": {"normal_equations": gsm[0] // km, "csne": ncs, "choleskyqr2": cq2 // km,
                                    "householder_tree": gsm[1] // km - cq2 // km - ncs,
                                    "csne_step_solves_delivered_declined": [cs[1], cs[2]]},
            "parity": bm.parity(min(16, B)) if check > 0 else None,
            "kernels_ms_per_step": {k: round(v, 4) for k, v in profile_table(bm).items() if v > 0},
            "note": ("kappa(J) = 3e3 for every problem, no bounds: all of them beyond the gate"
                     if key.startswith("certificate_rejected") else
                     "kappa(J) log-uniform over [1, 1e4]; each problem on the path its certificate allows. "
                     + ("Bounds as in the headline workload: the Coleman-Li block E^2 of the augmented "
                        "system [J D; E] (trf.py:264-270) keeps the SOLVED system well conditioned."
                        if key == "mixed_conditioning" else
                        "No bounds: the solved system is J^T J itself."))}
    finally:
        bm.close()
    return rec


# ------------------------------------------------------------- one config --
class Bench:
    """Device-resident inputs + the step closure of one config on one ctx."""

    def __init__(self, name, ctx, rank, world, batch=None, m=None, n=None, comm_ready=False,
                 mutate=None, alt_mutate=None):
        from bounded_lsq import TrfStepSolver, DogboxStepSolver, _synth
        cfg = dict(CONFIGS[name])
        if batch:
            cfg["batch"] = batch
        if m:
            cfg["m"] = m
        if n:
            cfg["n"] = n
        self.name, self.cfg, self.ctx = name, cfg, ctx
        self.rank, self.world = rank, world
        kind, B, m, n = cfg["kind"], cfg["batch"], cfg["m"], cfg["n"]
        self.kind, self.B, self.m, self.n = kind, B, m, n
        self.Delta = make_deltas(kind, B)
        if kind == "tsqr":
            from bounded_lsq._multi import TsqrTrfSolver
            rng = np.random.default_rng(555 + rank)            # this rank's row block
            J = rng.standard_normal((m, n))
            f = rng.standard_normal(m)
            r0 = np.random.default_rng(554)                     # x / bounds: the same on every rank
            x = r0.uniform(-1.0, 1.0, n)
            self.P = dict(J=J[None], f=f[None], x=x[None], lb=(x - r0.uniform(1e-3, 0.05, n))[None],
                          ub=(x + r0.uniform(1e-3, 0.05, n))[None], scale=np.ones((1, n)))
            assert world == 1 or comm_ready, "c5 at N > 1 needs ctx.comm_init first"
            self.sol = TsqrTrfSolver(m, n, world, rank, ctx=ctx, m_total=m * world)
            self.Delta = np.array([0.5])
        elif kind == "dogbox":
            self.P = _synth.dogbox_batch(20_000 + rank * B, B, m, n)
            if mutate is not None:
                mutate(self.P)
                if not np.isfinite(self.P["lb"]).any() and not np.isfinite(self.P["ub"]).any():
                    self.P["on_bound"][:] = 0                   # (no bounds: nothing to sit on)
            self.sol = DogboxStepSolver(B, m, n, ctx=ctx)
        else:
            self.P = _synth.trf_batch(10_000 + rank * B, B, m, n)   # each rank its own problems
            if mutate is not None:
                mutate(self.P)
            self.sol = TrfStepSolver(B, m, n, ctx=ctx)
        keys = ("J", "f", "x", "lb", "ub", "scale") + (("on_bound",) if kind == "dogbox" else ())
        self.d = {k: ctx.to_device(self.P[k]) for k in keys}
        self.dDelta = ctx.to_device(self.Delta)
        self.dAlpha = ctx.to_device(np.zeros(B))
        # a SECOND input set on the same plan (TRF only): step() then alternates between the two, so what the
        # optimistic device API guessed from the last call ("every problem on the fast path") fails every other call
        self.d2 = None
        self.calls = 0
        if alt_mutate is not None and kind == "trf":
            self.P2 = _synth.trf_batch(30_000 + rank * B, B, m, n)
            alt_mutate(self.P2)
            self.d2 = {k: ctx.to_device(self.P2[k]) for k in keys}

    def step(self):
        d, s = self.d, self.sol
        if self.d2 is not None:
            d = self.d2 if (self.calls & 1) else self.d
            self.calls += 1
        if self.kind == "dogbox":
            s.factor_dev(d["J"], d["f"], d["x"], d["lb"], d["ub"], d["scale"], d["on_bound"])
            s.step_dev(self.dDelta)
        else:
            s.factor_dev(d["J"], d["f"], d["x"], d["lb"], d["ub"], d["scale"])
            s.step_dev(self.dDelta, self.dAlpha)

    def step_host(self):
        """The host-pointer API: numpy in, numpy out (H2D of J inside the call)."""
        P, s = self.P, self.sol
        if self.kind == "dogbox":
            s.factor(P["J"], P["f"], P["x"], P["lb"], P["ub"], P["scale"], P["on_bound"])
            return s.step(self.Delta)
        s.factor(P["J"], P["f"], P["x"], P["lb"], P["ub"], P["scale"])
        return s.step(self.Delta, np.zeros(self.B))

    def close(self):
        self.sol.close()
        for p in list(self.d.values()) + list((self.d2 or {}).values()) + [self.dDelta, self.dAlpha]:
            self.ctx.free(p)

    # ---- parity spot check against the oracle (rank 0) ----
    def parity(self, nprob):
        from oracle import blsq_oracle as orc
        P, B = self.P, self.B
        S = self.sol.fetch_step()
        worst, masks_ok = 0.0, True
        if self.kind == "tsqr" and self.world > 1:
            return {"problems": 0, "note": "row blocks of other ranks are not on this host: "
                                           "parity of the split problem is covered by tests/"}
        for b in range(min(nprob, B)):
            if self.kind == "dogbox":
                _, So = orc.dogbox_step_solve(P["J"][b], P["f"][b], P["x"][b], P["lb"][b], P["ub"][b],
                                              P["scale"][b], P["on_bound"][b], float(self.Delta[b]))
                masks_ok = masks_ok and bool(np.array_equal(S.on_bound_new[b], So.on_bound_new))
            else:
                _, So = orc.trf_step_solve(P["J"][b], P["f"][b], P["x"][b], P["lb"][b], P["ub"][b],
                                           P["scale"][b], float(self.Delta[b]), 0.0)
                masks_ok = masks_ok and bool(np.array_equal(S.hits[b], So.hits))
            worst = max(worst, float(np.linalg.norm(S.step[b] - So.step) / np.linalg.norm(So.step)))
        out = {"problems": min(nprob, B), "max_rel_step_err": worst, "masks_bit_exact": masks_ok}
        if self.kind != "dogbox":
            out["svd_free_fraction"] = float(self.sol.debug_fast().mean())
        return out


def time_steps(bench, steps, warmup, fence, min_time=0.0, reduce_max=None):
    """W warm-up steps, a short PROFILE region (two HIP events around every launch -> the per-kernel table; untimed:
    those events cost a step about 2 %), then `steps` timed steps between two fences (barrier + device sync) with
    events around the launches of the dominant kernel only (its live average duration: the roofline).
    min_time > 0: if the timed region was shorter, it is repeated as ONE region of r x steps steps with
    r = ceil(min_time / elapsed) — every rank derives r from the same max-over-ranks figure — and that
    longer region is what is reported.  -> (elapsed of this rank, steps actually timed);
    bench.profile = (slot -> (ms, launches), profile steps), bench.dom = (slot name, (ms, launches) in the timed region)"""
    ctx = bench.ctx
    for _ in range(warmup):
        bench.step()

    def region(k, only):
        fence()
        ctx.timing(True, only=only)
        ctx.timing_reset()
        ctx.gram_stats(reset=True)
        ctx.cqr2_stats(reset=True)
        t0 = time.perf_counter()
        for _ in range(k):
            bench.step()
        ctx.sync()
        el = time.perf_counter() - t0
        fence()
        tm = ctx.timing_read()
        ctx.timing(False)
        return el, tm
    kp = max(2, min(steps, 10))
    _, prof = region(kp, None)
    dom = max(prof, key=lambda k: prof[k][0])
    elapsed, tm = region(steps, dom)
    if min_time > 0.0:
        seen = reduce_max(elapsed) if reduce_max else elapsed
        if seen < min_time:
            reps = int(min(400, max(2, -(-min_time // max(seen, 1e-6)))))
            steps = steps * reps
            elapsed, tm = region(steps, dom)
    bench.profile = (prof, kp)
    bench.dom = (dom, tm[dom])
    return elapsed, steps


def profile_table(bench):
    """ms per step of every kernel slot, from the profile region of time_steps"""
    prof, kp = bench.profile
    return {k: v[0] / kp for k, v in prof.items()}


def record(bench, cfg_name, elapsed, steps, warmup, world, probe):
    """The JSON record of one timed config (rank 0)."""
    ctx = bench.ctx
    kind, B, m, n = bench.kind, bench.B, bench.m, bench.n
    akind = "dogbox" if kind == "dogbox" else "trf"
    units = (1 if kind == "tsqr" else B * world) * steps         # step-solves in the timed region
    value = units / elapsed
    ms_per_step = 1e3 * elapsed / steps
    per_step_ms = profile_table(bench)                           # (every slot: the profile region)
    dom, (dom_total_ms, dom_launches) = bench.dom                # (the dominant kernel: live, over the timed region)
    dom_ms = dom_total_ms / steps                                # dominant kernel, per step
    per_step_ms[dom] = dom_ms
    # The dominant launch is priced with the work of ITS OWN algorithm: the Gram
    # kernel m N (N + 1) flops over 8 m N bytes, the Householder leaf 2 r N^2 - 2/3 N^3.  SURVEY
    # 8(d)'s per-solve figure (an SVD-based count of the whole step-solve) is larger than what
    # either kernel executes and is reported beside it, never as `achieved`.
    nprob = 1 if kind == "tsqr" else B                           # problems per launch on this GPU
    ws_gbs = alg_bytes(akind, m, n) * nprob / (ms_per_step * 1e-3) / 1e9          # whole step: SURVEY 8(d) bytes
    ws_tf = gram_flops(m, n) * nprob / (ms_per_step * 1e-3) / 1e12                # ... and the Gram's own flops
    traffic, tsrc = measured_traffic(cfg_name, dom, m, n, B)
    common = {
        "traffic": traffic, "traffic_source": tsrc,
        "peak_source": "nominal (AMD spec: FP64 matrix = vector 78.6 TFLOP/s; HBM3E 8 TB/s)",
        "kernel": dom, "kernel_ms_per_step": per_step_ms[dom],
        "survey_flops_per_solve": alg_flops(akind, m, n),
        "survey_bytes_per_solve": alg_bytes(akind, m, n),
        "whole_step_hbm_gbs": ws_gbs, "whole_step_hbm_frac_of_8TBs": ws_gbs / PEAK_HBM_GBS,
        "whole_step_mfma_tflops_gram_flops": ws_tf, "whole_step_mfma_frac_gram_flops": ws_tf / PEAK_FP64_TFLOPS,
    }
    own = {"gram": (gram_flops(m, n), gram_bytes(m, n)),
           "qr_leaf": (leaf_flops(m, n), 8.0 * m * (n + 1)),
           # the CSNE pass streams J once: 8 m n bytes, 2 m n flops per recorded vector (not priced: memory-bound)
           "csne_pass": (4.0 * m * n, 8.0 * m * (n + 1))}.get(dom)
    if own is not None:
        own_f, own_b = own
        balance = PEAK_FP64_TFLOPS * 1e12 / (PEAK_HBM_GBS * 1e9)
        bound = "mfma" if own_f / own_b > balance else "hbm"
        ach_tf = own_f * nprob / (dom_ms * 1e-3) / 1e12
        ach_gbs = own_b * nprob / (dom_ms * 1e-3) / 1e9
        roof = dict(common, **{
            "scope": "kernel", "bound": bound,
            "achieved": ach_tf if bound == "mfma" else ach_gbs,
            "peak": PEAK_FP64_TFLOPS if bound == "mfma" else PEAK_HBM_GBS,
            "unit": "TFLOP/s" if bound == "mfma" else "GB/s",
            "frac": (ach_tf / PEAK_FP64_TFLOPS) if bound == "mfma" else (ach_gbs / PEAK_HBM_GBS),
            "kernel_flops_per_solve": own_f, "kernel_bytes_per_solve": own_b,
            "kernel_tflops": ach_tf, "kernel_hbm_gbs": ach_gbs})
        if probe:
            roof["peak_measured"] = probe
            if bound == "mfma" and probe.get("mfma_f64_tflops"):
                roof["frac_of_measured_peak"] = ach_tf / probe["mfma_f64_tflops"]
            if bound == "hbm" and probe.get("hbm_copy_gbs"):
                roof["frac_of_measured_peak"] = ach_gbs / probe["hbm_copy_gbs"]
    else:
        # The dominant slot is a latency-bound n-space kernel (one wave or workgroup per problem): no roofline of
        # its own is claimed.  The line carries the WHOLE STEP against both rooflines — algorithmic bytes and the
        # Gram's own flops over ms_per_step — and reports the one it is closer to; the kernel is named, not priced.
        hb, mf = ws_gbs / PEAK_HBM_GBS, ws_tf / PEAK_FP64_TFLOPS
        bound = "hbm" if hb >= mf else "mfma"
        roof = dict(common, **{
            "scope": "whole_step", "bound": bound,
            "achieved": ws_gbs if bound == "hbm" else ws_tf,
            "peak": PEAK_HBM_GBS if bound == "hbm" else PEAK_FP64_TFLOPS,
            "unit": "GB/s" if bound == "hbm" else "TFLOP/s",
            "frac": hb if bound == "hbm" else mf,
            "note": "dominant slot `%s` is latency-bound (one problem's dependent chain): whole-step fractions, "
                    "SURVEY 8(d) bytes x problems / ms_per_step and Gram flops x problems / ms_per_step" % dom})
        if probe:
            roof["peak_measured"] = probe
    gs = ctx.gram_stats()
    metric = ("dogbox" if kind == "dogbox" else "TRF") + " step-solves/sec (batched m x n dense Jacobian)"
    return {
        "metric": metric, "value": value, "unit": "step-solves/s", "n_gpus": world, "steps": steps,
        "warmup": warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "%s, %d problem%s per GPU, %s, inputs resident in HBM" % (
                       CONFIGS[cfg_name]["label"] if (m, n) == (CONFIGS[cfg_name]["m"], CONFIGS[cfg_name]["n"])
                       else "%s step-solve, m=%d n=%d" % (akind, m, n), B, "" if B == 1 else "s",
                       ("Delta = 0.02" if kind == "dogbox" else "Delta mix 10/0.5 (reflective/feasible)")
                       + ("; two input sets alternate between calls" if bench.d2 is not None else
                          "; the same inputs every step (best case of the optimistic device API: its guess always holds)")),
                   "name": cfg_name, "m": m, "n": n, "batch_per_gpu": B,
                   "sharding": ("by rows: %d rows per rank, one ncclAllReduce of the Gram per factor call"
                                % m) if kind == "tsqr" else "by problem, no collective"},
        "roofline": roof,
        "factorisation_paths": {"normal_equations": gs[0], "householder_tree": gs[1]},
        "kernels_ms_per_step": per_step_ms,
        "kernels_note": "the dominant slot: HIP events over the timed region; the others: a profile region of %d "
                        "steps before it (events around every launch cost a step about 2 %%, so the timed region "
                        "carries them for the dominant kernel only)" % bench.profile[1],
    }


def side_run(name, ctx, steps, warmup, probe, **kw):
    """A short run of another config on the same ctx -> compact record (never `value`)."""
    b = Bench(name, ctx, 0, 1, **kw)
    try:
        el, steps = time_steps(b, steps, warmup, ctx.sync)
        r = record(b, name, el, steps, warmup, 1, probe)
        par = b.parity(2)
    finally:
        b.close()
    rf = r["roofline"]
    return {"workload": r["config"]["workload"], "value": r["value"], "unit": r["unit"],
            "ms_per_step": r["ms_per_step"], "steps": steps,
            "roofline": {k: rf[k] for k in ("scope", "bound", "achieved", "peak", "unit", "frac", "kernel",
                                            "kernel_ms_per_step", "whole_step_hbm_frac_of_8TBs",
                                            "whole_step_mfma_frac_gram_flops")},
            "mfma_ceiling_frac_survey": r["value"] * rf["survey_flops_per_solve"] / (PEAK_FP64_TFLOPS * 1e12),
            "kernels_ms_per_step": {k: round(v, 4) for k, v in r["kernels_ms_per_step"].items() if v > 0},
            "parity": par}


def with_timeout(fn, seconds, what):
    """Run fn() in a thread; a side figure that hangs must not take the bench line with it."""
    box = {}

    def run():
        try:
            box["out"] = fn()
        except Exception as exc:                              # noqa: BLE001
            box["out"] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:300])}
    th = threading.Thread(target=run, daemon=True)
    th.start()
    th.join(seconds)
    if th.is_alive():
        return {"error": "%s did not finish within %d s" % (what, seconds)}, True
    return box["out"], False


# --------------------------------------------------------------------- main --
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--config", default="c2", choices=sorted(k for k in CONFIGS if k != "c2-dogbox"))
    ap.add_argument("--steps", type=int, default=None,
                    help="timed steps (default: enough for a timed region of about a second)")
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--batch", type=int, default=None, help="problems per GPU (default: the config's)")
    ap.add_argument("--m", type=int, default=None)
    ap.add_argument("--n", type=int, default=None)
    ap.add_argument("--no-householder", action="store_true",
                    help="skip the side run with the normal-equations front end switched off")
    ap.add_argument("--no-side", action="store_true", help="skip the other configs' side figures")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-h2d", action="store_true", help="skip the host-pointer (PCIe-inclusive) leg")
    ap.add_argument("--no-probe", action="store_true", help="skip the measured-peak probes")
    ap.add_argument("--check", type=int, default=2, help="problems checked against the oracle")
    ap.add_argument("--min-time", type=float, default=None,
                    help="shortest timed region in seconds (default: 5 for the headline config, 1 otherwise): a shorter "
                         "--steps region is repeated as one region of r x steps steps (reported as `steps`, with "
                         "`steps_requested`); 0: exactly --steps")
    ap.add_argument("--cpu-child", type=int, default=None, help=argparse.SUPPRESS)   # (internal: the CPU leg's process)
    args = ap.parse_args()
    if args.cpu_child is not None:
        # the CPU leg, in a process of its own that never touches the GPU (its pools fork freely)
        from bounded_lsq import _synth
        Bc = max(1, args.cpu_child)
        Pc = _synth.trf_batch(10_000, min(Bc, 64), 4096, 256)       # a sample of the same seeded batch
        print(json.dumps(cpu_baseline(Pc, make_deltas("trf", min(Bc, 64)))), flush=True)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # No launcher around us: start the N ranks ourselves, as a CHILD (this process has not touched the GPU and
        # never will), relay rank 0's line and the launcher's exit code (non-zero if any rank failed).
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        sys.exit(subprocess.run(cmd, env=env).returncode)
    name = args.config
    default_steps = {"c2": 250, "c2-single": 500, "c3": 400, "c4": 300, "c5": 400}[name]
    steps = args.steps if args.steps is not None else default_steps
    warmup = args.warmup if args.warmup is not None else 10

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    main_line = name == "c2" and not (args.m or args.n)
    if args.min_time is None:
        args.min_time = 5.0 if main_line else 1.0
    if args.gpus != world and rank == 0:
        print("bench.py: --gpus %d but the launcher started %d rank(s): reporting n_gpus = %d"
              % (args.gpus, world, world), file=sys.stderr)
    cpu = None                                               # (the CPU leg runs LAST, in a child: see below)

    dist = None
    backend = os.environ.get("BLSQ_DIST_BACKEND", "nccl")     # "gloo": rehearsal on one GPU
    if world > 1:
        import torch
        import torch.distributed as dist
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    from bounded_lsq import _abi
    ndev = max(1, _abi.load().blsq_device_count())
    ctx = _abi.Context(local_rank % ndev)

    def tdev():
        """explicit torch device of this rank (a helper thread's current device is 0, not ours)"""
        import torch
        return torch.device("cuda", local_rank) if backend == "nccl" else torch.device("cpu")

    def bcast_comm_id():
        """rank 0's RCCL rendezvous id to every rank, through the launcher's process group"""
        import torch
        dev = tdev()
        nb = ctx.lib.blsq_comm_id_bytes()
        buf = torch.zeros(nb, dtype=torch.uint8, device=dev)
        if rank == 0:
            buf.copy_(torch.frombuffer(bytearray(ctx.comm_new_id()), dtype=torch.uint8))
        dist.broadcast(buf, src=0)
        return bytes(buf.cpu().numpy().tobytes())

    def fence():
        ctx.sync()
        if dist is not None:
            import torch
            if backend == "nccl":
                torch.cuda.synchronize(tdev())
            dist.barrier()
            if backend == "nccl":
                torch.cuda.synchronize(tdev())

    def max_over_ranks(v):
        if dist is None:
            return v
        import torch
        t = torch.tensor([v], dtype=torch.float64, device=tdev())
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def all_ranks(v):
        """every rank's value, rank order (rank 0 reports them: the driver can check N = 1 against BENCH)"""
        if dist is None:
            return [v]
        import torch
        t = torch.tensor([v], dtype=torch.float64, device=tdev())
        g = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(g, t)
        return [float(x.item()) for x in g]

    probe = None
    if not args.no_probe:
        try:
            probe = {"mfma_f64_tflops": ctx.probe("mfma_f64", 2)[0],
                     "hbm_copy_gbs": ctx.probe("copy", 1024)[0],
                     "source": "blsq_debug_probe in this run: back-to-back v_mfma_f64_16x16x4_f64 from "
                               "every SIMD (2 waves); 1 GiB device-to-device copy, read + written bytes"}
        except Exception as exc:                               # noqa: BLE001
            probe = {"error": str(exc)[:200]}

    comm_ready = False
    comm_error = None
    comm_hung = False

    def bring_up_comm():
        """the library's own RCCL communicator over all ranks, guarded: an exception is reported, a bring-up that does
        not return within two minutes is reported and left alone (its helper thread may still sit in librccl)"""
        nonlocal comm_ready, comm_error, comm_hung
        def go():
            ctx.comm_init(world, rank, bcast_comm_id())
            return {"ok": True}
        res, hung = with_timeout(go, 120, "blsq_comm_init over %d ranks" % world)
        comm_hung = hung
        if isinstance(res, dict) and res.get("ok"):
            comm_ready = True
        else:
            comm_error = (res or {}).get("error", "unknown") if isinstance(res, dict) else str(res)
    if world > 1 and name == "c5":
        bring_up_comm()                                        # (the tall problem needs it before its first factor call)
        if not comm_ready:
            raise RuntimeError("c5 at N > 1 needs the RCCL communicator: %s" % comm_error)
    bench = Bench(name, ctx, rank, world, batch=args.batch, m=args.m, n=args.n, comm_ready=comm_ready)
    steps_requested = steps
    own_elapsed, steps = time_steps(bench, steps, warmup, fence, args.min_time, max_over_ranks)
    elapsed = max_over_ranks(own_elapsed)
    out = record(bench, name, elapsed, steps, warmup, world, probe) if rank == 0 else None
    per_rank = all_ranks(own_elapsed)
    if out is not None:
        out["steps_requested"] = steps_requested
        units_rank = (1 if bench.kind == "tsqr" else bench.B) * steps
        out["per_rank"] = {"elapsed_s": per_rank, "step_solves_per_s": [units_rank / e for e in per_rank],
                           "note": "each rank's own clock over the same fenced region; `value` uses the max"}
    if world > 1 and not comm_ready:
        # The batch configs have no data-path collective and their `value` is in hand; NOW — never in front of the
        # timed region — every N > 1 run also brings up the library's RCCL communicator, so that the line shows which
        # librccl the ranks resolved and that it spans them all (the `c5_tsqr` leg below then reuses it).
        bring_up_comm()
    if out is not None:
        if comm_ready:
            out["rccl"] = ctx.comm_info()
        elif comm_error:
            out["rccl"] = {"error": comm_error}
    parity = bench.parity(args.check) if (rank == 0 and args.check > 0) else None

    extras = {}
    side_hung = False
    if world == 1 and main_line:
        B, m, n = bench.B, bench.m, bench.n
        if not args.no_householder:
            # the same workload with the normal-equations front end off: every problem through the
            # Householder TSQR tree — what a batch gets whose problems fail the conditioning gate
            gram_was = ctx.get_option("gram")
            ctx.set_option("gram", 0)                          # (a route switch: read when a plan is created)
            try:
                bh = Bench(name, ctx, 0, 1, batch=args.batch)
            finally:
                ctx.set_option("gram", gram_was)
            kh = max(2, min(steps_requested, 20))
            eh, kh = time_steps(bh, kh, 1, ctx.sync)
            extras["householder_only"] = {
                "value": B * kh / eh, "unit": "step-solves/s", "ms_per_step": 1e3 * eh / kh, "steps": kh,
                "kernels_ms_per_step": {k: round(v, 4) for k, v in profile_table(bh).items() if v > 0},
                "note": "option gram = 0: Householder TSQR tree for every problem"}
            bh.close()
        if not args.no_householder:
            # A batch of mixed conditioning: kappa(J) log-uniform over [1, 1e4] (J = Z V diag(s) V^T, Z
            # Gaussian, s log-spaced over [1/kappa_b, 1]; 64 distinct problems tiled to the batch).  The
            # certificate sends each problem down its own path; `factorisation_paths` shows the split.
            for key, mut in (("mixed_conditioning", mut_mixed), ("mixed_conditioning_unbounded", mut_mixed_unbounded),
                             ("certificate_rejected", mut_all_rejected)):
                extras[key] = conditioning_leg(key, mut, ctx, name, args.batch, steps_requested, args.check)
            # ... and the all-rejected batch through dogbox (its Newton step corrected at factor time: CSNE)
            extras["certificate_rejected_dogbox"] = conditioning_leg("certificate_rejected_dogbox", mut_all_rejected, ctx,
                                                                     "c2-dogbox", args.batch, steps_requested, args.check)
            # The headline's inputs repeat: what the optimistic device API guesses from the last call always holds.
            # Here two input sets ALTERNATE on one plan — the headline batch and the unbounded mixed-conditioning one —
            # so the guess "every problem on the fast path" fails every other call (repair + the step once more).
            ba = Bench(name, ctx, 0, 1, batch=args.batch, alt_mutate=mut_mixed_unbounded)
            try:
                ka = 2 * max(2, min(steps_requested, 20) // 2)
                ea, ka = time_steps(ba, ka, 2, ctx.sync)
                extras["alternating_conditioning"] = {
                    "value": B * ka / ea, "unit": "step-solves/s", "ms_per_step": 1e3 * ea / ka, "steps": ka,
                    "kernels_ms_per_step": {k: round(v, 4) for k, v in profile_table(ba).items() if v > 0},
                    "note": "even calls: the headline batch; odd calls: kappa(J) log-uniform over [1, 1e4], no bounds "
                            "(its rejected problems on the CSNE tier) — same plan, the optimistic verdict of the device "
                            "API wrong at every second call; the mean of the two batches' costs plus the repairs"}
            finally:
                ba.close()
        if not args.no_h2d:
            # numpy in, numpy out through blsq_trf_factor / blsq_trf_step: the 8 MiB Jacobian of every
            # problem crosses PCIe inside the call ("including and excluding H2D of J")
            bench.step_host()
            kh = 3
            # (a NEW array per call, as a `jac` callback returns one: the runtime's pin-on-the-fly path is fast only
            #  for a buffer it has seen before — the copy of J is made outside the clock)
            eh = 0.0
            keepJ = bench.P["J"]
            for _ in range(kh):
                bench.P["J"] = keepJ.copy()
                t0 = time.perf_counter()
                bench.step_host()
                eh += time.perf_counter() - t0
                bench.P["J"] = keepJ
            t0 = time.perf_counter()
            for _ in range(kh):
                bench.step_host()
            er = time.perf_counter() - t0
            extras["h2d_inclusive"] = {
                "value": B * kh / eh, "unit": "step-solves/s", "ms_per_step": 1e3 * eh / kh, "steps": kh,
                "host_to_device_GBps": 8.0 * B * m * (n + 1) * kh / eh / 1e9,
                "same_buffer_again": {"value": B * kh / er, "host_to_device_GBps": 8.0 * B * m * (n + 1) * kh / er / 1e9},
                "note": "host-pointer API, pageable numpy buffers, a NEW array of J per call (one copy, the runtime "
                        "pins on the fly; `same_buffer_again`: the same array in every call); `pinned`: the same from "
                        "page-locked buffers (blsq_host_alloc), copied in sub-batches under the Grams of the previous "
                        "ones; the PCIe link bounds all of them, never `value`"}
            try:
                # the same with J / f in page-locked memory (Context.pinned_empty -> blsq_host_alloc): straight DMA
                Jp = ctx.pinned_empty(bench.P["J"].shape)
                fp = ctx.pinned_empty(bench.P["f"].shape)
                Jp[...] = bench.P["J"]
                fp[...] = bench.P["f"]
                keep = bench.P["J"], bench.P["f"]
                bench.P["J"], bench.P["f"] = Jp, fp
                bench.step_host()
                t0 = time.perf_counter()
                for _ in range(kh):
                    bench.step_host()
                ep = time.perf_counter() - t0
                bench.P["J"], bench.P["f"] = keep
                ctx.pinned_free(Jp)
                ctx.pinned_free(fp)
                extras["h2d_inclusive"]["pinned"] = {
                    "value": B * kh / ep, "ms_per_step": 1e3 * ep / kh,
                    "host_to_device_GBps": 8.0 * B * m * (n + 1) * kh / ep / 1e9}
            except Exception as exc:                           # noqa: BLE001
                extras["h2d_inclusive"]["pinned"] = {"error": str(exc)[:200]}
        def front_end_leg():
            try:
                # per-iteration cost of the drop-in front end on ONE 4096 x 256 problem with host callbacks
                # (least_squares.py:351-371: fun and jac return numpy arrays every iteration)
                import bounded_lsq
                P1 = bench.P
                J1, x1 = np.ascontiguousarray(P1["J"][0]), P1["x"][0].copy()
                y1 = J1 @ x1 + 0.1 * P1["f"][0]

                cb = {"t": 0.0}                                            # time spent inside the callbacks, measured in place

                def fun1(xx):
                    t_ = time.perf_counter()
                    r_ = np.tanh(J1 @ xx - y1)
                    cb["t"] += time.perf_counter() - t_
                    return r_

                def jac1(xx):
                    t_ = time.perf_counter()
                    r_ = (1.0 - np.tanh(J1 @ xx - y1) ** 2)[:, None] * J1
                    cb["t"] += time.perf_counter() - t_
                    return r_
                kw1 = dict(jac=jac1, bounds=(P1["lb"][0] - 1.0, P1["ub"][0] + 1.0), method="trf", max_nfev=12)
                # (the callbacks' BLAS on ONE thread: a 4096 x 256 matvec gains nothing from 64, and a pool of spinning
                #  BLAS threads on the job's 16-CPU share delays the library's own host thread by milliseconds)
                try:
                    import threadpoolctl
                    blas1 = threadpoolctl.threadpool_limits(limits=1)
                except Exception:                              # noqa: BLE001
                    blas1 = None
                bounded_lsq.least_squares(fun1, x1 + 0.01, **kw1)          # (warm: code objects, allocator)
                t0 = time.perf_counter()
                sol1 = bounded_lsq.TrfStepSolver(1, m, n)                  # what every solve pays once: its plan
                sol1.close()
                eplan = time.perf_counter() - t0
                cb["t"] = 0.0
                t0 = time.perf_counter()
                r1 = bounded_lsq.least_squares(fun1, x1 + 0.01, **kw1)
                e1 = time.perf_counter() - t0
                ecb = cb["t"]
                best = (e1 - ecb, e1, ecb, r1)
                for _ in range(4):                              # (best of five solves: the box's host share is noisy)
                    cb["t"] = 0.0
                    t0 = time.perf_counter()
                    r1 = bounded_lsq.least_squares(fun1, x1 + 0.01, **kw1)
                    e1 = time.perf_counter() - t0
                    if e1 - cb["t"] < best[0]:
                        best = (e1 - cb["t"], e1, cb["t"], r1)
                _, e1, ecb, r1 = best
                if blas1 is not None:
                    blas1.restore_original_limits()
                extras["least_squares_single_4096x256"] = {
                    "nfev": int(r1.nfev), "njev": int(r1.njev), "status": int(r1.status), "total_ms": 1e3 * e1,
                    "callbacks_ms": 1e3 * ecb, "plan_create_ms": 1e3 * eplan,
                    "plan_pooled": True, "solves": 5, "callbacks_blas_threads": 1 if blas1 is not None else None,
                    "ms_per_iteration_excluding_callbacks_and_plan": 1e3 * (e1 - ecb) / max(1, int(r1.njev)),
                    "note": "bounded_lsq.least_squares (sequential host driver, numpy callbacks): wall time per outer "
                            "iteration = H2D of the 8 MiB Jacobian + factor + inner steps + result fetches; the plan "
                            "is leased from the context's pool (created by the warm call: plan_create_ms is what a "
                            "first solve of a shape pays once, nothing is subtracted for it here)"}
            except Exception as exc:                           # noqa: BLE001
                extras["least_squares_single_4096x256"] = {"error": str(exc)[:200]}
        if not args.no_side:
            side = {}
            for sn, ks in (("c2-single", 100), ("c3", 40), ("c4", 40), ("c5", 100)):
                side[sn], hung = with_timeout(lambda sn=sn, ks=ks: side_run(sn, ctx, ks, 3, probe), 180,
                                              "side config " + sn)
                if hung:
                    side_hung = True
                    break
            extras["side_configs"] = side
    if world > 1 and main_line and not args.no_side and not comm_hung:
        # the tall problem of config 5 over ALL ranks (250 000 rows each): the library's own RCCL
        # communicator, Gram all-reduce inside blsq_tsqr_factor_dev.  A side figure: guarded, timed out.
        def c5_leg():
            if not comm_ready:
                ctx.comm_init(world, rank, bcast_comm_id())
            b5 = Bench("c5", ctx, rank, world, comm_ready=True)
            try:
                k5 = 100
                e5, k5 = time_steps(b5, k5, 5, fence)
                e5 = max_over_ranks(e5)
                r5 = record(b5, "c5", e5, k5, 5, world, probe) if rank == 0 else None
                info5 = ctx.comm_info()
            finally:
                b5.close()
            if r5 is None:
                return None
            return {"workload": "one TRF problem of %d x 128 split by rows over %d ranks" % (250_000 * world, world),
                    "value": r5["value"], "unit": "step-solves/s", "ms_per_step": r5["ms_per_step"],
                    "steps": k5, "kernels_ms_per_step": r5["kernels_ms_per_step"],
                    "mfma_ceiling_frac_survey": r5["value"] * alg_flops("trf", 250_000 * world, 128)
                    / (PEAK_FP64_TFLOPS * 1e12 * world),
                    "collective": "ncclAllReduce(sum) of the 144 x 144 Gram on the library's stream",
                    "rccl": info5}
        c5, hung = with_timeout(c5_leg, 240, "c5 over RCCL")
        if rank == 0:
            extras["c5_tsqr"] = c5
        if hung or (isinstance(c5, dict) and "error" in c5):
            # a hung or failed collective on SOME rank: report and leave without any further
            # collective (teardown included) — the other ranks may be anywhere
            if rank == 0:
                out.update(extras)
                out["cpu_baseline"] = None
                out["speedup_vs_cpu"] = None
                out["parity"] = parity
                print(json.dumps(out), flush=True)
            sys.stdout.flush()
            # non-zero: the launcher must see a failed multi-GPU leg as a failure.  No further GPU call on
            # this ctx (a helper thread may still sit in the stuck collective), no teardown, no re-exec.
            os._exit(3)

    if world == 1 and main_line and not args.no_h2d and not side_hung:
        # (the last GPU leg: its numpy callbacks wake the BLAS thread pool, whose workers then spin for tens of
        #  milliseconds on the box's CPU share — measured: a latency-bound side leg that followed at once ran at HALF
        #  speed; the CPU leg below waits a moment for them to go back to sleep)
        front_end_leg()
        time.sleep(1.0)
    if rank == 0 and world == 1 and not args.no_cpu and main_line and not side_hung:
        # The CPU leg LAST, after every GPU leg (so that the GPU work of this run is contiguous), and in a child
        # process of its own: this one has initialised the GPU and must not fork worker pools.
        try:
            Bc = args.batch or CONFIGS["c2"]["batch"]
            cp = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-child", str(Bc)],
                                capture_output=True, text=True, timeout=600)
            cpu = json.loads(cp.stdout.strip().splitlines()[-1]) if cp.returncode == 0 else \
                {"error": (cp.stderr or cp.stdout)[-300:]}
            if cpu is not None and "value" not in cpu:
                extras["cpu_baseline_error"] = cpu
                cpu = None
        except Exception as exc:                               # noqa: BLE001
            extras["cpu_baseline_error"] = str(exc)[:300]
            cpu = None
    if rank == 0:
        out.update(extras)
        out["cpu_baseline"] = cpu
        out["speedup_vs_cpu"] = (out["value"] / cpu["value"]) if cpu else None
        out["parity"] = parity
        # compact digest of the side figures as the LAST key (a log tail of 2000 characters keeps it):
        # name -> [step-solves/s, ms per step, roofline fraction of the dominant kernel or null]
        digest = {}
        for k, v in (extras.get("side_configs") or {}).items():
            if isinstance(v, dict) and "value" in v:
                # (third entry: the roofline fraction — of the dominant MFMA / streaming kernel, or of the whole step
                #  where the dominant slot is latency-bound: `roofline.scope` says which)
                digest[k] = [round(v["value"], 1), round(v["ms_per_step"], 4), round(v["roofline"]["frac"], 3)]
            else:
                digest[k] = v
        for k in ("householder_only", "certificate_rejected", "certificate_rejected_dogbox", "mixed_conditioning",
                  "mixed_conditioning_unbounded", "alternating_conditioning", "h2d_inclusive"):
            if k in extras:
                digest[k] = [round(extras[k]["value"], 1), round(extras[k]["ms_per_step"], 4), None]
        if "c5_tsqr" in extras and isinstance(extras["c5_tsqr"], dict) and "value" in extras["c5_tsqr"]:
            digest["c5_tsqr"] = [round(extras["c5_tsqr"]["value"], 1), round(extras["c5_tsqr"]["ms_per_step"], 4), None]
        out["side_summary"] = digest
        print(json.dumps(out), flush=True)
    if comm_hung:
        # the communicator's bring-up never came back (its helper thread still sits in librccl): the line is out and
        # says so (`rccl.error`); the measurement itself needed no collective — leave without teardown
        sys.stdout.flush()
        os._exit(0)
    if side_hung:
        # a side leg never came back: its helper thread still drives this ctx and stream, so nothing more
        # is enqueued behind it (close() would wait on the stuck work) — the line is out, leave, non-zero
        sys.stdout.flush()
        os._exit(4)
    bench.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
