"""N > 1 host logic on CPU: sharding arithmetic and the triangle exchange over
torch.distributed (gloo, world_size 2).  The triangles here come from the
oracle-side QR (numpy) — the point is the plumbing: rank order of the
all-gather and that merging the gathered stack reproduces the full problem."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_and_row_block_arithmetic():
    from bounded_lsq._multi import shard_range, row_block, tri_ld
    for total, world in [(8192, 8), (1024, 3), (5, 8), (1, 1)]:
        spans = [shard_range(total, world, r) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
            assert a1 == b0 and a0 <= a1
        assert max(hi - lo for lo, hi in spans) == -(-total // world)
    blocks = [row_block(2_000_000, 8, r) for r in range(8)]
    assert blocks[0] == (0, 250_000) and blocks[-1] == (1_750_000, 2_000_000)
    assert tri_ld(128) == 144 and tri_ld(256) == 272 and tri_ld(15) == 16


WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(%(root)r, "bounded-lsq_amd"))
sys.path.insert(0, %(root)r)
import torch, torch.distributed as dist
from bounded_lsq._multi import allgather_triangles, row_block, tri_ld
from bounded_lsq import _synth
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
m, n = 600, 12
P = _synth.trf_problem(99, m, n)
lo, hi = row_block(m, world, rank)
A = np.column_stack([P["J"][lo:hi], P["f"][lo:hi]])
R = np.linalg.qr(A, mode="r")                      # (n+1) x (n+1) triangle of the row block
ld = tri_ld(n)
tri = torch.zeros((ld, ld), dtype=torch.float64)
tri[:n + 1, :n + 1] = torch.from_numpy(R)
stack = allgather_triangles(tri, world)
assert stack.shape == (world, ld, ld)
# rank order: slot r must hold rank r's triangle
mine = stack[rank].numpy()
assert np.array_equal(mine, tri.numpy())
# merging the gathered stack == QR of the whole problem (up to row signs)
S = np.vstack([stack[r].numpy()[:n + 1, :n + 1] for r in range(world)])
Rm = np.linalg.qr(S, mode="r")
Rf = np.linalg.qr(np.column_stack([P["J"], P["f"]]), mode="r")
assert np.allclose(np.abs(Rm), np.abs(Rf), rtol=1e-11, atol=1e-12)
g = Rm[:n, :n].T @ Rm[:n, n]
assert np.allclose(g, P["J"].T @ P["f"], rtol=1e-11, atol=1e-11)
dist.barrier()
dist.destroy_process_group()
open(os.path.join(%(out)r, "rank%%d.ok" %% rank), "w").write("ok")
'''


def test_triangle_allgather_gloo_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT, "out": str(tmp_path)})
    import socket
    with socket.socket() as sk:                    # a free rendezvous port
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert (tmp_path / "rank0.ok").exists() and (tmp_path / "rank1.ok").exists()
