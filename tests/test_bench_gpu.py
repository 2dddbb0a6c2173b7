"""bench.py as the driver runs it for N > 1 — but self-sufficient: `python bench.py --gpus 2 --config c5` with no launcher
around it starts its own two ranks (torch.distributed.run as a child), the ranks bring up the library's RCCL
communicator (here the socket stand-in, two processes on the one GPU; gloo for the bench's own barrier) and rank 0's
line reports what the first real multi-GPU run needs: n_gpus, per-rank clocks, the collective library and its span."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = os.path.join(ROOT, "tests", "stub_ccl", "libblsq_stub_ccl.so")


def _bench(*flags, world=2):
    if not os.path.exists(STUB):
        out = subprocess.run(["make", "-C", os.path.dirname(STUB)], capture_output=True, text=True)
        assert out.returncode == 0 and os.path.exists(STUB), out.stdout[-1500:] + out.stderr[-1500:]
    env = dict(os.environ, BLSQ_RCCL_PATH=STUB, BLSQ_DIST_BACKEND="gloo", OMP_NUM_THREADS="4")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "5", "--warmup", "1",
           "--min-time", "0", "--no-probe", "--no-side", "--check", "0"] + list(flags)
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_starts_its_own_ranks_for_the_tall_problem():
    line = _bench("--config", "c5")
    assert line["n_gpus"] == 2 and line["steps"] == 5
    assert line["rccl"]["ranks"] == 2 and line["rccl"]["library"].endswith("libblsq_stub_ccl.so")
    assert len(line["per_rank"]["elapsed_s"]) == 2
    assert "250000 rows per rank" in line["config"]["sharding"]
    assert line["roofline"]["frac"] > 0 and line["value"] > 0


def test_batch_configs_at_two_ranks_report_the_communicator_too():
    line = _bench("--config", "c4")
    assert line["n_gpus"] == 2 and line["scaling"] == "weak"
    assert line["rccl"]["ranks"] == 2
    assert len(line["per_rank"]["step_solves_per_s"]) == 2
    # (a latency-bound dominant slot claims no kernel roofline; with two ranks' launches interleaving on ONE card the
    #  slot that dominates rank 0's five steps is the Gram in some runs, the Newton rounds' Cholesky in most)
    roof = line["roofline"]
    assert roof["scope"] == ("kernel" if roof["kernel"] in ("gram", "qr_leaf") else "whole_step")
    # whole-job rate = what both ranks did over the slowest rank's clock
    assert line["value"] == pytest.approx(2 * 1024 * 5 / max(line["per_rank"]["elapsed_s"]), rel=1e-6)
