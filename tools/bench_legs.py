"""The conditioning legs of bench.py alone (certificate_rejected, mixed_conditioning[_unbounded]): quick A/B of the tiers
a rejected problem takes.  python tools/bench_legs.py [leg ...] [--batch B] [--steps K]"""
import argparse, json, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "bounded-lsq_amd"))
import bench
from bounded_lsq import _abi

ap = argparse.ArgumentParser()
ap.add_argument("legs", nargs="*", default=["certificate_rejected", "mixed_conditioning_unbounded", "mixed_conditioning"])
ap.add_argument("--batch", type=int, default=None)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--check", type=int, default=2)
a = ap.parse_args()
ctx = _abi.Context(0)
muts = {"certificate_rejected": bench.mut_all_rejected, "mixed_conditioning_unbounded": bench.mut_mixed_unbounded,
        "mixed_conditioning": bench.mut_mixed}
muts["certificate_rejected_dogbox"] = bench.mut_all_rejected
muts["householder_only"] = None                        # the headline batch with option gram = 0: the Householder tree for all
for leg in a.legs:
    if leg == "householder_only":
        ctx.set_option("gram", 0)
    r = bench.conditioning_leg(leg, muts[leg], ctx, "c2-dogbox" if leg.endswith("dogbox") else "c2", a.batch, a.steps, a.check)
    print(leg, json.dumps({k: r[k] for k in ("value", "ms_per_step", "factorisation_paths", "parity", "kernels_ms_per_step")}), flush=True)
    if leg == "householder_only":
        ctx.set_option("gram", 1)
