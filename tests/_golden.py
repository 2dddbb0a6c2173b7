"""Readers for the committed golden fixtures (tests/golden/)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class Case(dict):
    pass


def load_npz(fname):
    """-> list of (name, inputs dict, outputs dict)."""
    z = np.load(os.path.join(GOLDEN, fname))
    out = []
    for name in z["names"]:
        name = str(name)
        ins, outs = {}, {}
        for key in z.files:
            if key.startswith(name + "/in/"):
                ins[key[len(name) + 4:]] = z[key]
            elif key.startswith(name + "/out/"):
                outs[key[len(name) + 5:]] = z[key]
        out.append((name, ins, outs))
    return out


def load_json(fname):
    with open(os.path.join(GOLDEN, fname)) as fh:
        return json.load(fh)


def unhex(lst):
    return np.array([float.fromhex(v) for v in lst], dtype=float)


def trf_inputs(ins):
    """Full inputs, regenerated from the seed when the fixture stores one."""
    from bounded_lsq import _synth
    if "seed" in ins:
        P = _synth.trf_problem(int(ins["seed"]), int(ins["m"]), int(ins["n"]))
    else:
        P = {k: np.array(ins[k]) for k in ("J", "f", "x", "lb", "ub", "scale")}
    P["Delta"] = float(ins["Delta"])
    P["alpha0"] = float(ins["alpha0"])
    return P


def dog_inputs(ins):
    from bounded_lsq import _synth
    if "seed" in ins:
        P = _synth.dogbox_problem(int(ins["seed"]), int(ins["m"]), int(ins["n"]))
        for k in ("x", "lb", "ub", "scale", "on_bound"):   # edited vectors stored in full
            if k in ins:
                P[k] = np.array(ins[k])
    else:
        P = {k: np.array(ins[k]) for k in ("J", "f", "x", "lb", "ub", "scale",
                                            "on_bound")}
    P["Delta"] = float(ins["Delta"])
    return P
