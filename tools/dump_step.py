"""Step outputs of a bench config as an .npz (to compare two builds bit for bit): python tools/dump_step.py <config> <out.npz>"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bounded-lsq_amd")); sys.path.insert(0, ROOT)
import bench
from bounded_lsq import _abi
ctx = _abi.Context(0)
b = bench.Bench(sys.argv[1], ctx, 0, 1)
b.step(); b.step()
S = b.sol.fetch_step()
out = {k: np.asarray(getattr(S, k)) for k in dir(S) if not k.startswith("_") and isinstance(getattr(S, k), np.ndarray)}
np.savez(sys.argv[2], **out)
print(sys.argv[1], sorted(out))
b.close(); ctx.close()
