#!/bin/bash
# A/B environment settings in one GPU call: tools/ab_env.sh "VAR=a VAR=b ..." [bench args]
SETS=$1; shift 1
for rep in 1 2; do
for S in $SETS; do
  env $S timeout -k 10 200 python bench.py --steps 4 --warmup 1 --no-cpu "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$S', round(d['value']), round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['kernels_ms_per_step'].items()})" || exit 1
done; done
