"""Multi-GPU host logic: one process per GPU (torch.distributed over RCCL).

Two cases:

* batches of independent problems -> contiguous shards, NO data-path
  collective (``shard_range``);
* ONE very tall problem -> row blocks; each rank factors its block of [J f] to
  an (n+1)x(n+1) triangle on its GPU, the triangles are exchanged with ONE
  all-gather (RCCL over xGMI: n=128 -> 162 KB per rank, latency-bound), every
  rank merges the stack and runs the ordinary n-space path redundantly, so
  g, the SVD factor and the step come out replicated (``TsqrTrfSolver``).

torch is plumbing here (exchange buffers + the collective); all arithmetic is
in libblsq_hip.so.
"""
import ctypes as C

import numpy as np

from . import _abi
from ._hip_step import TrfStepSolver, SCALE_GIVEN


def shard_range(total, world, rank):
    """Contiguous block of ceil(total/world) items for `rank` (last may be short)."""
    per = -(-int(total) // int(world))
    lo = min(rank * per, total)
    return lo, min(lo + per, total)


def row_block(m, world, rank):
    """Rows [rank*m/world, (rank+1)*m/world) of a tall problem."""
    return (rank * m) // world, ((rank + 1) * m) // world


def tri_ld(n):
    return (n + 1 + 15) // 16 * 16


def allgather_triangles(local_tri, world):
    """All-gather of the per-rank triangles (torch tensor, any device) ->
    tensor [world, ld, ld] in rank order.  backend nccl == RCCL on ROCm."""
    import torch
    import torch.distributed as dist
    out = torch.empty((world,) + tuple(local_tri.shape), dtype=local_tri.dtype,
                      device=local_tri.device)
    if world == 1:
        out[0].copy_(local_tri)
        return out
    dist.all_gather_into_tensor(out.view(-1), local_tri.contiguous().view(-1))
    return out


class TsqrTrfSolver(TrfStepSolver):
    """TRF step solver for one tall problem whose rows are split over ranks.

    ``factor_dev`` takes THIS rank's row block (device pointers); the state
    after it (and every ``step``) is identical on all ranks.
    """

    def __init__(self, m_local, n, world, rank, ctx=None):  # noqa: D401 (no super().__init__)
        from ._hip_step import default_context
        self.ctx = ctx or default_context()
        self.lib = self.ctx.lib
        self.B, self.m, self.n = 1, int(m_local), int(n)
        self.world, self.rank = int(world), int(rank)
        self.ld = tri_ld(n)
        h = _abi.vp()
        self.ctx.check(self.lib.blsq_tsqr_plan_create(self.ctx.h, self.m, self.n, self.world,
                                                      C.byref(h)), "blsq_tsqr_plan_create")
        self.h = h

    def local_triangle_dev(self, dJ_block, df_block, dtri_out):
        self.ctx.check(self.lib.blsq_tsqr_local_dev(self.h, dJ_block, df_block, dtri_out),
                       "blsq_tsqr_local_dev")

    def combine_dev(self, dtri_stack, dx, dlb, dub, dscale, scale_mode=SCALE_GIVEN):
        self.ctx.check(self.lib.blsq_tsqr_combine_dev(self.h, dtri_stack, dx, dlb, dub, dscale,
                                                      int(scale_mode)),
                       "blsq_tsqr_combine_dev")

    def factor_dev(self, dJ_block, df_block, dx, dlb, dub, dscale, scale_mode=SCALE_GIVEN):
        """local QR -> all-gather (torch.distributed, RCCL) -> merge + n-space."""
        import torch
        dev = torch.device("cuda", self.ctx.device_id)
        tri = torch.empty((self.ld, self.ld), dtype=torch.float64, device=dev)
        self.local_triangle_dev(dJ_block, df_block, _abi.vp(tri.data_ptr()))
        self.ctx.sync()                                   # library stream -> torch stream
        stack = allgather_triangles(tri, self.world)
        torch.cuda.synchronize(dev)
        self.combine_dev(_abi.vp(stack.data_ptr()), dx, dlb, dub, dscale, scale_mode)
        self.ctx.sync()
        return stack
