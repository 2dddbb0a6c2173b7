"""Generic reader of rocprofv3 --pmc passes: per kernel, the counters summed over its launches
(the first launch of each kernel left out when there are at least three) and the launch durations.
usage: python tools/pmc_counters.py <out.json> <pass_dir> [<pass_dir> ...]   (one directory per --pmc pass)"""
import csv
import glob
import json
import os
import sys


def read_pass(d):
    dur, order = {}, {}
    for path in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        with open(path) as fh:
            rows = sorted(csv.DictReader(fh), key=lambda r: int(r["Start_Timestamp"]))
        for r in rows:
            name = r["Kernel_Name"].split("(")[0]
            k = order.get(name, 0)
            order[name] = k + 1
            dur[(name, k)] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    cnt, seen = {}, {}
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as fh:
            rows = list(csv.DictReader(fh))
        rows.sort(key=lambda r: int(r.get("Dispatch_Id", 0)))
        disp = {}
        for r in rows:
            name = r["Kernel_Name"].split("(")[0]
            did = r.get("Dispatch_Id")
            if (name, did) not in disp:
                k = seen.get(name, 0)
                seen[name] = k + 1
                disp[(name, did)] = k
            e = cnt.setdefault((name, disp[(name, did)]), {})
            e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    out = {}
    for name in sorted({k[0] for k in cnt}):
        keys = sorted(k for k in cnt if k[0] == name)
        first = 1 if len(keys) >= 3 else 0
        tot, t, nl = {}, 0, 0
        for k in keys[first:]:
            for c, v in cnt[k].items():
                tot[c] = tot.get(c, 0.0) + v
            t += dur.get(k, 0)
            nl += 1
        out[name] = {"launches": nl, "avg_duration_us": t / max(nl, 1) / 1e3,
                     "per_launch": {c: v / max(nl, 1) for c, v in tot.items()}}
    return out


def main():
    outp, dirs = sys.argv[1], sys.argv[2:]
    merged = {}
    for d in dirs:
        for name, rec in read_pass(d).items():
            if "blsq::" not in name:
                continue
            m = merged.setdefault(name, {"launches": rec["launches"], "avg_duration_us": {}, "per_launch": {}})
            m["avg_duration_us"][os.path.basename(d.rstrip("/"))] = rec["avg_duration_us"]
            m["per_launch"].update(rec["per_launch"])
    for name, m in merged.items():
        c = m["per_launch"]
        d = {}
        if c.get("SQ_WAVE_CYCLES"):
            for k in ("SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_LDS",
                      "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_MISC", "SQ_ACTIVE_INST_SCA", "SQ_WAIT_ANY", "SQ_INST_CYCLES_VMEM"):
                if k in c:
                    d[k + "/SQ_WAVE_CYCLES"] = c[k] / c["SQ_WAVE_CYCLES"]
        if c.get("SQ_INSTS_LDS"):
            for k in ("SQ_LDS_BANK_CONFLICT", "SQ_LDS_ADDR_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_UNALIGNED_STALL"):
                if k in c:
                    d[k + "/SQ_INSTS_LDS"] = c[k] / c["SQ_INSTS_LDS"]
        if c.get("SQ_INSTS_VALU_MFMA_F64") or c.get("SQ_INSTS_VALU_MFMA_MOPS_F64"):
            mf = c.get("SQ_INSTS_VALU_MFMA_F64") or 0.0
            for k in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_SMEM"):
                if k in c and mf:
                    d[k + "_per_MFMA"] = c[k] / mf
        m["derived"] = d
    with open(outp, "w") as fh:
        json.dump({"note": __doc__, "kernels": merged}, fh, indent=1)
    for name in merged:
        if "gram16" in name or "gram8" in name:
            print(name, json.dumps(merged[name]["derived"], indent=1))


if __name__ == "__main__":
    main()
