"""Longer fuzz sweeps than the test suite runs (GPU box): python tools/fuzz_long.py [count per sweep]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bounded-lsq_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_fuzz_gpu as fz
from bounded_lsq import _abi
count = int(sys.argv[1]) if len(sys.argv) > 1 else 150
ctx = _abi.Context(0)
tot_bad = 0
for seed, lk in ((11, (0.0, 4.0)), (12, (2.0, 6.0)), (13, (4.0, 8.0)), (14, (0.0, 8.0)), (15, (2.5, 5.5))):
    recs, bad, paths = fz.run_sweep(count, seed, ctx, budget_s=150, verbose=False, log_kappa=lk)
    exc = sum(r[7] for r in recs)
    worst = max([r[0] for r in recs if not r[7]] + [0.0])
    print("seed %d log10 kappa %s: %d problems, paths (Gram, rejected) %s, cqr2 so far %d, csne (routed, delivered, declined) so far %s, "
          "violations %d, excused %d, worst accepted %.2e"
          % (seed, lk, len(recs), paths, ctx.cqr2_stats(), ctx.csne_stats(), len(bad), exc, worst), flush=True)
    for v in bad:
        print("  VIOLATION case %d %s %s b=%d err %.2e oracle-move %.2e mask_ok %s mask_stable %s" % v)
    tot_bad += len(bad)
ctx.close()
sys.exit(1 if tot_bad else 0)
