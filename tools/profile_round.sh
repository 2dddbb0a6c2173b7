#!/bin/bash
# Collect the judged artefacts of one build on the GPU box (run through gpurun):
#   tools/profile_round.sh <tag>      e.g. r01d   -> gpurun_out/<tag>_*
# 1) default bench line (with cpu_baseline)  2) rocprofv3 kernel stats of a short run
# 3) two separate PMC passes (FETCH_SIZE, WRITE_SIZE)  4) traffic summary json
set -o pipefail
TAG=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
if [ "$2" != "noprof-bench" ]; then
timeout -k 10 500 python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err || exit 1
tail -c 600 $O/${TAG}_bench.json
fi
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats -o run -- python3 bench.py --steps 4 --warmup 1 --no-cpu --no-householder --check 0 > $O/${TAG}_stats.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${TAG}_fetch -o run -- python3 bench.py --steps 1 --warmup 0 --no-cpu --no-householder --check 0 > $O/${TAG}_fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${TAG}_write -o run -- python3 bench.py --steps 1 --warmup 0 --no-cpu --no-householder --check 0 > $O/${TAG}_write.log 2>&1 || exit 1
python tools/pmc_traffic.py $O/${TAG}_fetch $O/${TAG}_write $O/${TAG}_hbm_traffic.json 4096 256 512
find $O/${TAG}_stats -name "*kernel_stats.csv" -exec cp {} $O/${TAG}_kernel_stats.csv \;
head -12 $O/${TAG}_kernel_stats.csv
