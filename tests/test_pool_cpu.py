"""The plan pool of the sequential drivers (`_hip_step.lease_solver / return_solver`) without a GPU: stand-in solver
and context objects with the attributes the pool looks at (B, m, n, h, ctx, close)."""
import threading

import pytest


class FakeCtx:
    def __init__(self):
        self.h = object()


class FakeSolver:
    created = 0

    def __init__(self, B, m, n, ctx=None):
        self.B, self.m, self.n, self.ctx, self.h = int(B), int(m), int(n), ctx, object()
        type(self).created += 1

    def close(self):
        self.h = None


class OtherSolver(FakeSolver):
    pass


@pytest.fixture
def hs():
    from bounded_lsq import _hip_step
    keep = _hip_step.PLAN_POOL_KEEP
    FakeSolver.created = 0
    yield _hip_step
    _hip_step.PLAN_POOL_KEEP = keep


def test_a_returned_plan_is_leased_again_for_the_same_class_and_shape(hs):
    ctx = FakeCtx()
    a = hs.lease_solver(FakeSolver, 1, 100, 7, ctx=ctx)
    hs.return_solver(a)
    assert hs.lease_solver(FakeSolver, 1, 100, 7, ctx=ctx) is a           # same class, same shape
    hs.return_solver(a)
    b = hs.lease_solver(FakeSolver, 1, 100, 8, ctx=ctx)                   # another shape
    c = hs.lease_solver(OtherSolver, 1, 100, 7, ctx=ctx)                  # another class
    assert b is not a and c is not a and type(c) is OtherSolver
    other = FakeCtx()
    assert hs.lease_solver(FakeSolver, 1, 100, 7, ctx=other) is not a     # another context: its own pool
    assert hs.lease_solver(FakeSolver, 1, 100, 7, ctx=ctx) is a
    assert hs.lease_solver(FakeSolver, 1, 100, 7, ctx=ctx) is not a       # (a is out: a second solve in flight gets its own)


def test_the_pool_is_bounded_and_closes_what_it_drops(hs):
    ctx = FakeCtx()
    made = [hs.lease_solver(FakeSolver, 1, 10 + k, 3, ctx=ctx) for k in range(hs.PLAN_POOL_KEEP + 3)]
    for s in made:
        hs.return_solver(s)
    pool = ctx._solver_pool
    assert len(pool) == hs.PLAN_POOL_KEEP and pool == made[-hs.PLAN_POOL_KEEP:]
    assert all(s.h is None for s in made[:3]) and all(s.h is not None for s in pool)
    big = hs.lease_solver(FakeSolver, 512, 4096, 256, ctx=ctx)            # 4.3 GB of J: never kept
    hs.return_solver(big)
    assert big.h is None and len(pool) == hs.PLAN_POOL_KEEP
    hs.PLAN_POOL_KEEP = 0                                                 # pool off: create / destroy per solve
    s = hs.lease_solver(FakeSolver, 1, 99, 3, ctx=ctx)
    hs.return_solver(s)
    assert s.h is None


def test_closed_plans_and_closed_contexts_are_never_handed_out(hs):
    ctx = FakeCtx()
    a = hs.lease_solver(FakeSolver, 1, 50, 5, ctx=ctx)
    hs.return_solver(a)
    a.close()                                                             # (closed with its context, or by hand)
    b = hs.lease_solver(FakeSolver, 1, 50, 5, ctx=ctx)
    assert b is not a and a not in ctx._solver_pool
    ctx.h = None                                                          # the context is gone: the plan is closed, not kept
    hs.return_solver(b)
    assert b.h is None


def test_concurrent_solves_never_share_a_plan(hs):
    ctx = FakeCtx()
    seen, lock = [], threading.Lock()

    def work():
        for _ in range(200):
            s = hs.lease_solver(FakeSolver, 1, 64, 4, ctx=ctx)
            with lock:
                assert s not in seen
                seen.append(s)
            with lock:
                seen.remove(s)
            hs.return_solver(s)

    ts = [threading.Thread(target=work) for _ in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert FakeSolver.created <= 4 + hs.PLAN_POOL_KEEP
