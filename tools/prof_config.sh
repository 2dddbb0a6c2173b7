#!/bin/bash
# rocprofv3 kernel stats of one bench config:  tools/prof_config.sh <config> [extra bench flags]
set -o pipefail
C=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
SHORT="--no-cpu --no-householder --no-side --no-h2d --no-probe --check 0 --min-time 0"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$C -o run -- python3 bench.py --config $C --steps 30 --warmup 3 $SHORT "$@" > $O/prof_$C.log 2>&1 || exit 1
find $O/prof_$C -name "*kernel_stats.csv" -exec head -16 {} \; | cut -c1-160
