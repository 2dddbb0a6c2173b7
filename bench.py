#!/usr/bin/env python3
"""bench.py — TRF step-solves/second on batched dense Jacobians (MI355X).

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N > 1 is launched by torch.distributed.run, one rank per GPU; the batch is
  sharded by problem (independent problems, NO data-path collective, weak
  scaling: per-GPU batch fixed).  torch is used only for the barrier and the
  max-over-ranks of the elapsed time; the compute path is libblsq_hip.so
  through ctypes.

One "step" = one TRF step-solve (trf.py:244-308 = factor + one
inner step) for every problem of the per-GPU batch, inputs resident in HBM.
Workload at N = 1: BASELINE.json configs[1] shape (m=4096, n=256, TRF exact
step) batched, which is the configuration the metric / north_star target is
quoted on ("batched 4096x256 dense Jacobians at 1 GPU").

Prints ONE JSON line (rank 0) with `roofline` and `cpu_baseline` objects.  `factorisation_paths`
counts how many problems the normal-equations front end factored and how many its conditioning
gate handed to the Householder tree; `householder_only` (N = 1) is the same workload re-timed with
the front end switched off — a side figure, never `value`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "bounded-lsq_amd"))

import numpy as np  # noqa: E402

PEAK_FP64_TFLOPS = 78.6     # MI355X FP64 vector = matrix peak (nominal)
PEAK_HBM_GBS = 8000.0       # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)


def alg_bytes_trf(m, n):    # single pass over inputs, outputs once
    return 8 * (m * n + m + 4 * n) + 8 * 2 * n + 8 * n


def alg_flops_trf(m, n):    # R-SVD count + GEMV terms
    return 2 * (m + n) * n * n + 11 * n ** 3 + 6 * (m + n) * n


def leaf_flops_trf(m, n, rows_per_leaf=1024):
    """Flops of the dominant kernel's OWN work: Householder QR of the [J f] row blocks
    (2 r N^2 - 2/3 N^3 per r x N leaf, N = n + 1), nothing else of the step-solve."""
    N = n + 1
    nleaf = max(1, -(-m // rows_per_leaf))
    r = -(-m // nleaf)
    return nleaf * (2.0 * r * N * N - 2.0 * N ** 3 / 3.0)


def gram_flops_trf(m, n):
    """Flops of the normal-equations front end's Gram kernel: the symmetric product
    [J f]^T [J f], one multiply-add per row and per entry of the upper triangle."""
    N = n + 1
    return float(m) * N * (N + 1)


def gram_bytes_trf(m, n):
    return 8.0 * (m * (n + 1) + (n + 1) * (n + 2) / 2)


def measured_traffic(kernel, m, n, B):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC summary
    (profiles/hbm_traffic_latest.json: separate --pmc FETCH_SIZE / WRITE_SIZE passes,
    gfx950 correction per MI355X_MICROARCH.md) when it was collected for this exact
    workload; None otherwise (bench.py itself cannot collect PMC counters)."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic_latest.json")
    try:
        with open(path) as fh:
            t = json.load(fh)
        cfg = t.get("config", {})
        if (cfg.get("m"), cfg.get("n"), cfg.get("batch")) != (m, n, B):
            return None
        dom = t.get("dominant")
        if dom and dom.get("slot") == kernel:
            return dom["hbm_bytes_per_launch"]
        for name, v in t["kernels"].items():          # older summaries: first QR entry = leaf
            if kernel.split("_")[0] in name:
                return v["hbm_bytes"]
    except Exception:
        pass
    return None


def make_deltas(B):
    """Half 'reflective' (Delta=10: Gauss-Newton step, reflection branch) and
    half 'feasible' (Delta=0.5: More' iterations) as SURVEY.md 8(d) asks."""
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


def cpu_baseline(P, Delta, budget_s=24.0):
    """The reference's CPU path (numpy/scipy restatement, oracle/) on the host
    cores, three threading configurations; best is reported.
    Must run BEFORE the GPU is initialised (uses fork)."""
    import multiprocessing as mp
    from threadpoolctl import threadpool_limits, threadpool_info
    ncpu = os.cpu_count() or 1
    try:
        ncpu = min(ncpu, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    host_logical = ncpu
    ncpu = min(ncpu, int(os.environ.get("BLSQ_CPU_WORKERS", "16")))  # a 1-GPU box's CPU share
    _CPU["P"] = P
    _CPU["Delta"] = Delta
    B = P["J"].shape[0]
    _cpu_one(0)                                   # warm-up (imports, page faults)
    t0 = time.perf_counter()
    with threadpool_limits(limits=1):
        _cpu_one(0)
    t1 = time.perf_counter() - t0                 # single-thread cost of one solve
    per_mode = budget_s / 3.0
    res = {}
    # (ii) one BLAS thread, sequential loop
    k = int(max(2, min(B, per_mode / max(t1, 1e-6))))
    with threadpool_limits(limits=1):
        t0 = time.perf_counter()
        for b in range(k):
            _cpu_one(b % B)
        res["blas1_loop"] = (k / (time.perf_counter() - t0), 1, k)
    # (i) default BLAS threads, sequential loop
    k = int(max(2, min(B, per_mode / max(t1, 1e-6))))
    t0 = time.perf_counter()
    for b in range(k):
        _cpu_one(b % B)
    nthr = max([d.get("num_threads", 1) for d in threadpool_info()] + [1])
    res["blas_default_loop"] = (k / (time.perf_counter() - t0), nthr, k)
    # (iii) ncpu processes x 1 BLAS thread
    if ncpu > 1:
        k = int(max(ncpu, min(4 * B, ncpu * per_mode / max(t1, 1e-6))))
        ctx = mp.get_context("fork")
        with ctx.Pool(ncpu, initializer=_cpu_worker_init) as pool:
            pool.map(_cpu_one, [b % B for b in range(ncpu)])   # warm the workers
            t0 = time.perf_counter()
            pool.map(_cpu_one, [b % B for b in range(k)], chunksize=1)
            res["procs_x_blas1"] = (k / (time.perf_counter() - t0), ncpu, k)
    best = max(res, key=lambda kk: res[kk][0])
    cpu_model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except Exception:
        pass
    return {
        "value": res[best][0], "unit": "step-solves/s", "cores": res[best][1],
        "kind": "port",
        "sample": "%d TRF step-solves of the same seeded batch (mode %s; oracle/blsq_oracle.py "
                  "= scipy.linalg.svd(gesdd) path of trf.py:244-308)" % (res[best][2], best),
        "modes": {kk: {"value": v[0], "threads": v[1], "solves": v[2]} for kk, v in res.items()},
        "host": {"cpu": cpu_model, "logical_cores": host_logical, "workers_cap": ncpu},
    }


# --------------------------------------------------------------------- main --
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=512,
                    help="problems per GPU (512: two waves of 256 CUs; the n-space kernels run one workgroup per problem)")
    ap.add_argument("--no-householder", action="store_true",
                    help="skip the side run with the normal-equations front end switched off")
    ap.add_argument("--m", type=int, default=4096)
    ap.add_argument("--n", type=int, default=256)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--check", type=int, default=2, help="problems checked against the oracle")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    B, m, n = args.batch, args.m, args.n

    from bounded_lsq import _synth
    P = _synth.trf_batch(10_000 + rank * B, B, m, n)   # each rank its own problems
    Delta = make_deltas(B)
    alpha0 = np.zeros(B)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:
        cpu = cpu_baseline(P, Delta)              # before any GPU initialisation (fork)

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

    from bounded_lsq import TrfStepSolver, _abi
    ndev = max(1, _abi.load().blsq_device_count())
    ctx = _abi.Context(local_rank % ndev)
    sol = TrfStepSolver(B, m, n, ctx=ctx)
    d = {k: ctx.to_device(P[k]) for k in ("J", "f", "x", "lb", "ub", "scale")}
    dDelta = ctx.to_device(Delta)
    dAlpha = ctx.to_device(alpha0)

    def one_step():
        sol.factor_dev(d["J"], d["f"], d["x"], d["lb"], d["ub"], d["scale"])
        sol.step_dev(dDelta, dAlpha)

    def fence():
        ctx.sync()
        if dist is not None:
            import torch
            if backend == "nccl":
                torch.cuda.synchronize()
            dist.barrier()
            if backend == "nccl":
                torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    fence()
    ctx.timing(True)
    ctx.timing_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    ctx.sync()
    elapsed = time.perf_counter() - t0
    fence()
    ctx.timing(False)
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64,
                         device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # Same workload with the normal-equations front end switched off (every problem through the
    # Householder TSQR tree — what a batch gets whose problems fail the conditioning gate).  A
    # reported side figure at N = 1 only, never `value`.
    householder = None
    if world == 1 and not args.no_householder:
        os.environ["BLSQ_GRAM"] = "0"                  # read when a plan is created
        sol_h = TrfStepSolver(B, m, n, ctx=ctx)
        os.environ.pop("BLSQ_GRAM", None)

        def step_h():
            sol_h.factor_dev(d["J"], d["f"], d["x"], d["lb"], d["ub"], d["scale"])
            sol_h.step_dev(dDelta, dAlpha)
        step_h(); ctx.sync()
        kh = max(2, min(args.steps, 4))
        th = time.perf_counter()
        for _ in range(kh):
            step_h()
        ctx.sync()
        eh = time.perf_counter() - th
        householder = {"value": B * kh / eh, "unit": "step-solves/s", "ms_per_step": 1e3 * eh / kh,
                       "steps": kh, "note": "BLSQ_GRAM=0: Householder TSQR tree for every problem"}
        sol_h.close()

    # parity spot-check of the timed configuration (rank 0, a few problems)
    parity = None
    if rank == 0 and args.check > 0:
        from oracle import blsq_oracle as orc
        S = sol.fetch_step()
        worst = 0.0
        masks_ok = True
        for b in range(min(args.check, B)):
            _, So = orc.trf_step_solve(P["J"][b], P["f"][b], P["x"][b], P["lb"][b], P["ub"][b],
                                       P["scale"][b], float(Delta[b]), 0.0)
            worst = max(worst, float(np.linalg.norm(S.step[b] - So.step) /
                                     np.linalg.norm(So.step)))
            masks_ok = masks_ok and bool(np.array_equal(S.hits[b], So.hits))
        sw = sol.debug_sweeps()
        parity = {"problems": min(args.check, B), "max_rel_step_err": worst,
                  "masks_bit_exact": masks_ok,
                  "jacobi_sweeps": [int(sw.min()), float(sw.mean()), int(sw.max())],
                  "svd_free_fraction": float(sol.debug_fast().mean())}

    if rank == 0:
        timing = ctx.timing_read()
        total_solves = B * world * args.steps
        value = total_solves / elapsed
        ms_per_step = 1e3 * elapsed / args.steps
        kern = {k: {"ms_total": v[0], "launches": v[1],
                    "avg_ms": (v[0] / v[1] if v[1] else 0.0)} for k, v in timing.items()}
        per_step_ms = {k: v["ms_total"] / args.steps for k, v in kern.items()}
        dom = max(per_step_ms, key=lambda k: per_step_ms[k])
        dom_launches_per_step = max(1, kern[dom]["launches"] // max(1, args.steps))
        dom_avg_ms = kern[dom]["avg_ms"]
        dom_ms = dom_avg_ms * dom_launches_per_step        # dominant kernel, per step
        survey_flops = alg_flops_trf(m, n) * B             # R-SVD based step-solve count
        byts = alg_bytes_trf(m, n) * B
        # The roofline prices the dominant launch with the flops of ITS OWN algorithm (DESIGN.md
        # 6): the Gram kernel m N (N + 1), the Householder leaf 2 r N^2 - 2/3 N^3 per row block.
        # SURVEY 8(d)'s per-solve figure (an SVD-based count of the whole step-solve) is larger
        # than what either kernel executes and is reported beside it, never as `achieved`.
        own = {"gram": gram_flops_trf, "qr_leaf": leaf_flops_trf}.get(dom, alg_flops_trf)(m, n)
        achieved_tf = own * B / (dom_ms * 1e-3) / 1e12
        gs = ctx.gram_stats()
        out = {
            "metric": "TRF step-solves/sec (batched m x n dense Jacobian)",
            "value": value, "unit": "step-solves/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "C2 batched: TRF exact step-solve, m=%d n=%d, %d problems "
                                   "per GPU, Delta mix 10/0.5 (reflective/feasible), inputs "
                                   "resident in HBM" % (m, n, B),
                       "m": m, "n": n, "batch_per_gpu": B, "sharding": "by problem, no collective"},
            "roofline": {
                "bound": "mfma", "achieved": achieved_tf, "peak": PEAK_FP64_TFLOPS,
                "unit": "TFLOP/s", "frac": achieved_tf / PEAK_FP64_TFLOPS,
                "traffic": measured_traffic(dom, m, n, B),
                "kernel": dom, "kernel_ms_per_step": per_step_ms[dom],
                "kernel_flops_per_solve": own,
                "kernel_bytes_per_solve": gram_bytes_trf(m, n) if dom == "gram" else alg_bytes_trf(m, n),
                "kernel_hbm_gbs": (gram_bytes_trf(m, n) if dom == "gram" else alg_bytes_trf(m, n))
                * B / (dom_ms * 1e-3) / 1e9,
                "survey_flops_per_solve": alg_flops_trf(m, n),
                "survey_flops_over_kernel_time_tflops": survey_flops / (dom_ms * 1e-3) / 1e12,
                "alg_bytes_per_solve": alg_bytes_trf(m, n),
                "whole_step_survey_tflops": survey_flops / (ms_per_step * 1e-3) / 1e12,
                "whole_step_hbm_gbs": byts / (ms_per_step * 1e-3) / 1e9,
                "hbm_frac_of_8TBs": byts / (ms_per_step * 1e-3) / 1e9 / PEAK_HBM_GBS,
            },
            "factorisation_paths": {"normal_equations": gs[0], "householder_tree": gs[1]},
            "householder_only": householder,
            "kernels_ms_per_step": per_step_ms,
            "cpu_baseline": cpu,
            "speedup_vs_cpu": (value / cpu["value"]) if cpu else None,
            "parity": parity,
        }
        print(json.dumps(out))
    sol.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
