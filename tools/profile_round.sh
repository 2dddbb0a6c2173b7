#!/bin/bash
# Collect the judged artefacts of one build on the GPU box (run through gpurun):
#   tools/profile_round.sh <tag> [nobench]     e.g. r02f   -> gpurun_out/<tag>_*
# 1) default bench line (cpu_baseline, side configs, h2d)   2) rocprofv3 kernel stats of a short run
# 3) two separate PMC passes (FETCH_SIZE, WRITE_SIZE) -> traffic summary json
# 4) SQ pass: MFMA-pipe utilisation per kernel, calibrated on the FP64 MFMA probe (tools/pmc_mfma.py)
set -o pipefail
TAG=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
SHORT="--no-cpu --no-householder --no-side --no-h2d --no-probe --check 0 --min-time 0"
if [ "$2" != "nobench" ]; then
timeout -k 10 600 python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err || exit 1
tail -c 400 $O/${TAG}_bench.json; echo
fi
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats -o run -- python3 bench.py --steps 20 --warmup 2 $SHORT > $O/${TAG}_stats.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${TAG}_fetch -o run -- python3 bench.py --steps 1 --warmup 0 $SHORT > $O/${TAG}_fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${TAG}_write -o run -- python3 bench.py --steps 1 --warmup 0 $SHORT > $O/${TAG}_write.log 2>&1 || exit 1
python tools/pmc_traffic.py $O/${TAG}_fetch $O/${TAG}_write $O/${TAG}_hbm_traffic.json 4096 256 512 $TAG
timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/${TAG}_sq -o run -- python3 tools/pmc_mfma_run.py > $O/${TAG}_sq.log 2>&1 || exit 1
python tools/pmc_mfma.py $O/${TAG}_sq $O/${TAG}_pmc_sq.json
find $O/${TAG}_stats -name "*kernel_stats.csv" -exec cp {} $O/${TAG}_kernel_stats.csv \;
head -14 $O/${TAG}_kernel_stats.csv
