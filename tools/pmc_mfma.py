"""Turn SQ counter passes of tools/pmc_mfma_run.py into MFMA-pipe utilisation per kernel.

usage: python tools/pmc_mfma.py <pmc_dir> <out.json>

<pmc_dir> holds the *_counter_collection.csv and *_kernel_trace.csv of
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 \
              SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE \
              --kernel-trace -- python3 tools/pmc_mfma_run.py

Calibration: the probe launch issues FP64 MFMAs back to back from every SIMD, so by construction
its MFMA pipes are busy for the whole launch.  With c = counter value, t = launch duration:
    busy_fraction(kernel) = (c_busy(kernel) / t(kernel)) / (c_busy(probe) / t(probe))
which needs no assumption about the counter's unit or the clock; the probe also gives
counter-per-instruction (c / known MFMA count), reported so the raw numbers can be checked."""
import csv
import glob
import json
import os
import sys


def main():
    d, outp = sys.argv[1], sys.argv[2]
    dur = {}                                     # (kernel short name, nth launch) -> ns
    order = {}
    for path in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        with open(path) as fh:
            rows = sorted(csv.DictReader(fh), key=lambda r: int(r["Start_Timestamp"]))
        for r in rows:
            name = r["Kernel_Name"].split("(")[0]
            k = order.get(name, 0)
            order[name] = k + 1
            dur[(name, k)] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    cnt = {}                                     # (name, nth) -> {counter: value}
    seen = {}
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
    def agg(pred):
        tot, t = {}, 0
        for key, c in cnt.items():
            if pred(key):
                for kk, v in c.items():
                    tot[kk] = tot.get(kk, 0.0) + v
                t += dur.get(key, 0)
        return tot, t
    # probe: the timed launch of each flavour is the 2nd and 4th probe launch (warm-ups 1st, 3rd)
    out = {"note": __doc__.split("Calibration:")[1].strip(), "kernels": {}}
    pn = [k for k in cnt if "mfma_f64_probe_kernel" in k[0]]
    pn.sort(key=lambda k: k[1])
    probes = {}
    for label, idx in (("probe_1_wave_per_simd", 1), ("probe_2_waves_per_simd", 3)):
        key = next((k for k in pn if k[1] == idx), None)
        if key is None:
            continue
        c, t = cnt[key], dur.get(key, 0)
        waves = 256 * (4 if idx == 1 else 8)
        nm = waves * 20000 * 8
        probes[label] = {"duration_ns": t, "mfma_wave_insts": nm, "counters": c,
                         "busy_cycles_per_mfma": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / nm,
                         "insts_f64_per_mfma": c.get("SQ_INSTS_VALU_MFMA_F64", 0) / nm,
                         "mops_f64_per_mfma": c.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0) / nm,
                         "busy_per_ns": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(t, 1),
                         "tflops": nm * 2048.0 / max(t, 1) * 1e-3}
    out["probes"] = probes
    ref = probes.get("probe_2_waves_per_simd") or probes.get("probe_1_wave_per_simd")
    names = sorted({k[0] for k in cnt if "blsq::" in k[0] and "probe" not in k[0]})
    for name in names:
        nl = sum(1 for k in cnt if k[0] == name)
        first = 1 if nl >= 3 else 0           # (the cold first launch is left out when there are warm ones)
        c, t = agg(lambda key, name=name, first=first: key[0] == name and key[1] >= first)
        if t == 0:
            continue
        e = {"launches": nl, "launches_used": nl - first, "duration_ns": t, "counters": c}
        if ref and ref["busy_per_ns"] > 0:
            e["mfma_busy_fraction"] = (c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / t) / ref["busy_per_ns"]
            if ref["insts_f64_per_mfma"] > 0:
                nm = c.get("SQ_INSTS_VALU_MFMA_F64", 0) / ref["insts_f64_per_mfma"]
                e["mfma_f64_wave_insts"] = nm
                e["mfma_tflops"] = nm * 2048.0 / t * 1e-3
                e["mfma_frac_of_probe_rate"] = e["mfma_tflops"] / ref["tflops"]
        out["kernels"][name] = e
    with open(outp, "w") as fh:
        json.dump(out, fh, indent=1)
    for label, p in probes.items():
        print(label, "%.1f TF/s" % p["tflops"], "busy/mfma %.2f" % p["busy_cycles_per_mfma"],
              "insts/mfma %.3f" % p["insts_f64_per_mfma"], "mops/mfma %.3f" % p["mops_f64_per_mfma"])
    for name, e in out["kernels"].items():
        if "mfma_busy_fraction" in e and e["mfma_busy_fraction"] > 0.01:
            print("%-60s busy %.3f  %s" % (name[-60:], e["mfma_busy_fraction"],
                                            "mfma-rate/probe %.3f" % e.get("mfma_frac_of_probe_rate", 0)))


if __name__ == "__main__":
    main()
