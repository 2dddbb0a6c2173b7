"""Multi-GPU host logic: one process per GPU, no PyTorch.

Two cases:

* batches of independent problems -> contiguous shards, NO data-path collective
  (``shard_range``);
* ONE very tall problem -> row blocks (``row_block``); ``TsqrTrfSolver.factor_dev`` takes this
  rank's block and the library does the rest on its own stream: local Gram, ONE
  ``ncclAllReduce`` of the (n+1)^2 Gram over RCCL / xGMI, replicated Cholesky + gate — or, for a
  problem the gate rejects, local Householder TSQR, ONE ``ncclAllGather`` of the triangles and a
  replicated merge (``blsq_tsqr_factor_dev``).  Every rank ends with the same factor state and
  computes the same step.

The only thing the host has to move between the ranks is the communicator's rendezvous id (128
bytes, once): ``exchange_id_tcp`` does that over a plain TCP socket; any other channel works
(``comm_id=`` takes the bytes).
"""
import ctypes as C
import socket
import struct
import time

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


def exchange_id_tcp(rank, world, addr, port, make_id, timeout=120.0):
    """Rank 0 calls ``make_id()`` (-> bytes) and serves it on (addr, port); every other rank connects
    (retrying while the server is not up yet), names itself by its rank and receives the id.  Each
    rank is served exactly once, all of them within ONE deadline of `timeout` seconds; a connection
    that does not name a rank still missing is dropped without consuming a slot.  Returns the id on
    every rank.  Plain sockets: no dependency beyond the standard library.  (The id is a rendezvous
    token for processes of one job on one host or fabric, not a secret.)"""
    rank, world = int(rank), int(world)
    deadline = time.time() + timeout
    if rank == 0:
        payload = bytes(make_id())
        if world == 1:
            return payload
        missing = set(range(1, world))
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as srv:
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr, int(port)))
            srv.listen(world)
            while missing:
                left = deadline - time.time()
                if left <= 0:
                    raise TimeoutError("ranks %s did not fetch the communicator id within %.0f s"
                                       % (sorted(missing), timeout))
                srv.settimeout(left)
                try:
                    conn, _peer = srv.accept()
                except socket.timeout:
                    continue
                with conn:
                    try:
                        conn.settimeout(min(5.0, max(0.1, deadline - time.time())))
                        who = struct.unpack("<I", _recv_exact(conn, 4))[0]
                        if who not in missing:
                            continue                       # stranger or duplicate: no slot consumed
                        conn.sendall(struct.pack("<I", len(payload)) + payload)
                        missing.discard(who)
                    except (ConnectionError, socket.timeout, struct.error):
                        continue                           # that rank will come again
        return payload
    while True:
        try:
            with socket.create_connection((addr, int(port)), timeout=5.0) as sk:
                sk.settimeout(max(0.1, deadline - time.time()))
                sk.sendall(struct.pack("<I", rank))
                head = _recv_exact(sk, 4)
                return _recv_exact(sk, struct.unpack("<I", head)[0])
        except (ConnectionError, socket.timeout):          # server not up yet / slow, or it dropped this connection
            # (rank 0 drops a connection on purpose when its read times out, on a duplicate and when its send
            #  fails — "that rank will come again": refused, reset, aborted, broken pipe and the ConnectionError
            #  of _recv_exact are all retried until the deadline)
            if time.time() > deadline:
                raise
            time.sleep(0.05)


def _recv_exact(sk, n):
    buf = b""
    while len(buf) < n:
        chunk = sk.recv(n - len(buf))
        if not chunk:
            raise ConnectionError("peer closed while sending the communicator id")
        buf += chunk
    return buf


class TsqrTrfSolver(TrfStepSolver):
    """TRF step solver for one tall problem whose rows are split over ranks.

    ``factor_dev`` takes THIS rank's row block (device pointers); the state after it (and every
    ``step``) is identical on all ranks.  `m_total` = rows of the whole problem (it enters the
    reference's rank test, trust_region.py:109, and must be the same number everywhere).
    `comm_id`: the rendezvous id every rank got from rank 0 (``Context.comm_new_id`` +
    ``exchange_id_tcp``); None when the ctx already carries a communicator, when world == 1, or
    when the caller exchanges the triangles itself (``local_triangle_dev`` / ``combine_dev``).
    """

    def __init__(self, m_local, n, world, rank, ctx=None, m_total=None, comm_id=None):  # noqa: D401
        from ._hip_step import default_context
        self.ctx = ctx or default_context()
        self.lib = self.ctx.lib
        self.B, self.m, self.n = 1, int(m_local), int(n)
        self.world, self.rank = int(world), int(rank)
        self.m_total = int(m_total) if m_total is not None else self.m * self.world
        self.ld = tri_ld(n)
        if comm_id is not None:
            self.ctx.comm_init(self.world, self.rank, comm_id)
        h = _abi.vp()
        self.ctx.check(self.lib.blsq_tsqr_plan_create(self.ctx.h, self.m, self.m_total, self.n,
                                                      self.world, C.byref(h)),
                       "blsq_tsqr_plan_create")
        self.h = h
        self.ctx.adopt(self)

    def factor_dev(self, dJ_block, df_block, dx, dlb, dub, dscale, scale_mode=SCALE_GIVEN):
        """local Gram -> all-reduce -> Cholesky + gate (or: local TSQR -> all-gather -> merge),
        then the n-space path; asynchronous on the ctx stream apart from the gate read-back."""
        self.ctx.check(self.lib.blsq_tsqr_factor_dev(self.h, dJ_block, df_block, dx, dlb, dub,
                                                     dscale, int(scale_mode)),
                       "blsq_tsqr_factor_dev")

    # the two halves of the Householder route, for hosts that exchange the triangles themselves
    def local_triangle_dev(self, dJ_block, df_block, dtri_out):
        self.ctx.check(self.lib.blsq_tsqr_local_dev(self.h, dJ_block, df_block, dtri_out),
                       "blsq_tsqr_local_dev")

    def combine_dev(self, dtri_stack, dx, dlb, dub, dscale, scale_mode=SCALE_GIVEN):
        self.ctx.check(self.lib.blsq_tsqr_combine_dev(self.h, dtri_stack, dx, dlb, dub, dscale,
                                                      int(scale_mode)),
                       "blsq_tsqr_combine_dev")
