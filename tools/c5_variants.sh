#!/bin/bash
# Kernel stats of the C5 row block (250000 x 128) for a few row-chunk heights (experiments only).
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
SHORT="--no-cpu --no-householder --no-side --no-h2d --no-probe --check 0"
for R in "$@"; do
  export BLSQ_GRAM_TALL_ROWS=$R
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5v_$R -o run -- python3 bench.py --config c5 --steps 30 --warmup 3 $SHORT > $O/c5v_$R.log 2>&1 || exit 1
  echo "== rows $R"; tail -c 200 $O/c5v_$R.log | grep -o '"ms_per_step": [0-9.]*'
  find $O/c5v_$R -name "*kernel_stats.csv" -exec head -8 {} \; | cut -c1-150
done
