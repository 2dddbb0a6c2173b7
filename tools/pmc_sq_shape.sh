#!/bin/bash
# SQ counter pass for another shape:  tools/pmc_sq_shape.sh <tag> <B> <m> <n>
set -o pipefail
TAG=$1; export PMC_B=$2 PMC_M=$3 PMC_N=$4
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/${TAG}_sq -o run -- python3 tools/pmc_mfma_run.py > $O/${TAG}_sq.log 2>&1 || exit 1
python tools/pmc_mfma.py $O/${TAG}_sq $O/${TAG}_pmc_sq.json
