"""Convergence regression on the REFERENCE's OWN benchmark problem set (SURVEY.md 8f-3;
benchmarks/lsq_problems.py:1003-1018: 58 problems, all restated in tests/_suite58.py — the one defined by
a measurement table, CoatingThickness, from the table captured into the fixture as data).

tests/golden/suite58.json holds, per problem, the start point and box of the reference's factory and
what the reference's public drivers returned (both methods x numeric / 'jac' scaling = 232 records:
nfev, njev, status, x, objective, optimality, active mask), plus the same runs from start points
moved by one ulp.  The public front end on the GPU step path must reproduce every record that is
stable in the reference itself exactly in iteration counts, status and mask; records whose counts
change in the reference under the one-ulp move are compared through properties; where the
reference itself raises (NaN Jacobians) nothing is compared.

`test_runner_table` prints the table the reference's runner prints (run_benchmarks.py:116-154:
problem, n, m, solver, nfev, g norm, value, active, status) for the GPU path, next to the
reference's numbers, and writes it to gpurun_out/suite58_table.txt."""
import os

import numpy as np
import pytest

from _golden import load_json
import _suite58

pytestmark = pytest.mark.gpu

S58 = load_json("suite58.json")
TOL = float.fromhex(S58["tol"])
PROBLEMS = {p["name"]: p for p in S58["problems"]}


def unhex(v):
    return np.array([float.fromhex(s) for s in v])


def solve(rec):
    import bounded_lsq
    p = PROBLEMS[rec["problem"]]
    fun, jac = _suite58.functions(p)
    with np.errstate(all="ignore"):
        return bounded_lsq.least_squares(fun, unhex(p["x0"]), jac=jac,
                                         bounds=(unhex(p["lb"]), unhex(p["ub"])),
                                         method=rec["method"], ftol=TOL, xtol=TOL, gtol=TOL,
                                         scaling=rec["scaling"])


def test_the_fixture_covers_the_references_problem_set():
    assert S58["reference_problem_count"] == 58
    assert len(S58["problems"]) == 58 and S58["not_restated"] == []
    assert sum(1 for p in S58["problems"] if p["bounded"]) == 26
    assert len(S58["records"]) == 4 * 58
    ct = PROBLEMS["CoatingThickness"]
    assert (ct["n"], ct["m"]) == (134, 252) and len(ct["data"]["y"]) == 126


@pytest.mark.parametrize("rec", S58["records"],
                         ids=["%s-%s-%s" % (r["problem"], r["method"], r["scaling"])
                              for r in S58["records"]])
def test_suite58_record(rec):
    if "error" in rec:                       # the reference itself fails on this one
        try:
            solve(rec)
        except Exception:                    # noqa: BLE001  (any outcome but a hang is acceptable)
            pass
        return
    res = solve(rec)
    p = PROBLEMS[rec["problem"]]
    lb, ub = unhex(p["lb"]), unhex(p["ub"])
    if rec["stable"]:
        assert (res.nfev, res.njev, res.status) == (rec["nfev"], rec["njev"], rec["status"])
        obj_ref, x_ref = float.fromhex(rec["obj_value"]), unhex(rec["x"])
        fun0 = _suite58.functions(p)[0](unhex(p["x0"]))
        # (absolute floor relative to the objective at the start: Watson12 ends at 6e-10 from 30,
        #  cond(J) ~ 1e13 — its last digits are not determined by the data)
        np.testing.assert_allclose(res.obj_value, obj_ref, rtol=1e-6,
                                   atol=1e-12 * max(1.0, float(fun0.dot(fun0))))
        np.testing.assert_array_equal(res.active_mask, rec["active_mask"])
        if not np.allclose(res.x, x_ref, rtol=1e-6, atol=1e-9):
            # same counts, same status, same objective, different digits of x: legitimate only where
            # the data do not determine x to that accuracy — an ill-conditioned Jacobian at the
            # solution (Watson: cond ~ 1e8 .. 1e13) or a second minimiser of equal value (Biggs
            # EXP6 is symmetric under exchanging its exponential terms)
            _, jac = _suite58.functions(p)
            with np.errstate(all="ignore"):
                cond = np.linalg.cond(jac(x_ref))
            twin = abs(res.obj_value - obj_ref) <= 1e-9 * max(1.0, obj_ref) and res.optimality <= 1e-6
            assert cond > 1e6 or twin, (cond, res.x, x_ref)
            if cond > 1e6 and not twin:
                np.testing.assert_allclose(res.x, x_ref, rtol=1e-10 * cond, atol=1e-10 * cond)
        return
    # unstable in the reference itself (its counts change when x0 moves by one ulp): properties
    objs = [float.fromhex(rec["obj_value"])] + [float.fromhex(q["obj_value"])
                                                 for q in rec["neighbours"] if q["status"] != -99]
    worst = max(objs)
    statuses = {rec["status"]} | {q["status"] for q in rec["neighbours"]}
    if min(statuses) > 0:
        assert res.status > 0
    assert np.all(res.x >= lb) and np.all(res.x <= ub)
    assert res.obj_value <= worst * (1 + 1e-4) + 1e-12, (res.obj_value, objs)


def test_runner_table(capsys):
    """The runner's report (run_benchmarks.py:116-154) for the GPU path beside the reference."""
    header = "{:<25} {:<5} {:<5} {:<15} {:<5} {:<10} {:<10} {:<8} {:<8} | reference: nfev  value      status".format(
        "problem", "n", "m", "solver", "nfev", "g norm", "value", "active", "status")
    lines = [header, "-" * len(header)]
    names = {("dogbox", "1.0"): "dogbox", ("dogbox", "jac"): "dogbox-s", ("trf", "1.0"): "trf",
             ("trf", "jac"): "trf-s"}
    agree = total = 0
    last = None
    for rec in S58["records"]:
        if "error" in rec:
            continue
        res = solve(rec)
        p = PROBLEMS[rec["problem"]]
        first = rec["problem"] != last
        last = rec["problem"]
        lines.append("{:<25} {:<5} {:<5} {:<15} {:<5} {:<10.2e} {:<10.2e} {:<8} {:<8} | {:>15}  {:<10.2e} {}{}".format(
            rec["problem"] if first else "", p["n"] if first else "", p["m"] if first else "",
            names[(rec["method"], str(rec["scaling"]))], res.nfev, res.optimality, res.obj_value,
            int(np.sum(res.active_mask != 0)), res.status, rec["nfev"],
            float.fromhex(rec["obj_value"]), rec["status"], "" if rec["stable"] else "  (unstable in the reference)"))
        total += 1
        agree += (res.nfev, res.status) == (rec["nfev"], rec["status"])
    lines.append("%d of %d runs reproduce the reference's nfev and status exactly" % (agree, total))
    text = "\n".join(lines)
    with capsys.disabled():
        print("\n" + text)
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "suite58_table.txt"), "w") as fh:
            fh.write(text + "\n")
    stable = sum(1 for r in S58["records"] if r.get("stable"))
    assert agree >= stable
