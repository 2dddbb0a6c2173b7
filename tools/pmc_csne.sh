#!/bin/bash
# Instruction-mix and wait counters of the CSNE leg's kernels (csne_pass_kernel in particular):
#   tools/pmc_csne.sh <tag>   -> gpurun_out/<tag>_pmc_csne.txt.  Separate --pmc passes with --kernel-trace only.
set -o pipefail
TAG=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
RUN="python3 tools/bench_legs.py certificate_rejected --steps 3 --check 0"
pass() {
  local name=$1; shift
  timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/${TAG}_pc_$name -o run -- $RUN > $O/${TAG}_pc_$name.log 2>&1 || { echo "pass $name failed"; tail -5 $O/${TAG}_pc_$name.log; return 1; }
}
pass mix SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES || exit 1
pass act SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES || exit 1
pass mfma SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES || echo "(no mfma pass)"
python3 - $O/${TAG}_pc_mix $O/${TAG}_pc_act $O/${TAG}_pc_mfma <<'PY' | tee $O/${TAG}_pmc_csne.txt
import csv, glob, sys
for d in sys.argv[1:]:
    acc = {}
    for p in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            k = r["Kernel_Name"].split("(")[0].replace("void blsq::", "").replace("blsq::", "")
            e = acc.setdefault(k, {})
            e.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        if not any(t in k for t in ("csne_pass", "gram16", "lm_update", "csne_fix")): continue
        n = len(next(iter(v.values())))
        print("%-28s launches %3d  " % (k[:28], n) + "  ".join("%s=%.4g" % (c.replace("SQ_", ""), sum(x) / len(x)) for c, x in sorted(v.items())))
PY
