#!/bin/bash
# Instruction-mix / LDS / wait counters of the bench workload's kernels (the Gram kernel in particular):
#   tools/pmc_gram.sh <tag>   -> gpurun_out/<tag>_pmc_gram.json (+ the list of SQ counters of this box)
# Separate --pmc passes with --kernel-trace only (never combined with other trace domains).
set -o pipefail
TAG=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
rocprofv3 --list-avail > $O/${TAG}_avail.txt 2>&1 || true
grep -o "SQ_[A-Z0-9_]*" $O/${TAG}_avail.txt | sort -u > $O/${TAG}_sq_counters.txt
SHORT="--no-cpu --no-householder --no-side --no-h2d --no-probe --check 0 --min-time 0"
have() { grep -qx "$1" $O/${TAG}_sq_counters.txt; }
pass() {   # name, counters...
  local name=$1; shift
  local list=""
  for c in "$@"; do if have $c; then list="$list $c"; else echo "(counter $c not on this box)"; fi; done
  [ -z "$list" ] && return 0
  timeout -k 10 300 rocprofv3 --pmc $list --kernel-trace --output-format csv -d $O/${TAG}_$name -o run -- python3 bench.py --steps 4 --warmup 1 $SHORT > $O/${TAG}_$name.log 2>&1 || { echo "pass $name failed"; tail -5 $O/${TAG}_$name.log; return 1; }
}
pass mix SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_VALU_MFMA_F64 SQ_WAVE_CYCLES SQ_BUSY_CYCLES || exit 1
pass lds SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES || exit 1
pass act SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES || exit 1
python tools/pmc_counters.py $O/${TAG}_pmc_gram.json $O/${TAG}_mix $O/${TAG}_lds $O/${TAG}_act
