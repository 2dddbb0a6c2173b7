#!/bin/bash
# One rocprofv3 counter pass over tools/pmc_mfma_run.py for a shape; prints per-kernel counters.
#   tools/pmc_pass.sh <tag> <B> <m> <n> COUNTER [COUNTER ...]
set -o pipefail
TAG=$1; export PMC_B=$2 PMC_M=$3 PMC_N=$4; shift 4
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
timeout -k 10 400 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/${TAG} -o run -- python3 tools/pmc_mfma_run.py > $O/${TAG}.log 2>&1 || exit 1
python3 - "$O/${TAG}" <<'PY'
import csv, glob, sys
acc = {}
for p in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"].split("(")[0].replace("void blsq::", "")
        if k.startswith("blsq::"): k = k[6:]
        acc.setdefault((int(r["Dispatch_Id"]), k), {})[r["Counter_Name"]] = float(r["Counter_Value"])
seen = {}
for (d, k), v in sorted(acc.items()):
    if seen.get(k, 0) >= 1 or "probe" in k or "rocclr" in k: continue
    seen[k] = seen.get(k, 0) + 1
    print("%-34s" % k[:34], "  ".join("%s=%.4g" % (a.replace("SQ_", ""), b) for a, b in sorted(v.items())))
PY
