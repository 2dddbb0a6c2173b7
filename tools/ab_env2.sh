#!/bin/bash
# A/B environment settings in one GPU call, default step counts: tools/ab_env2.sh "VAR=a VAR=b ..." [bench args]
SETS=$1; shift 1
for rep in 1 2; do
for S in $SETS; do
  env $S timeout -k 10 200 python bench.py --no-cpu --no-householder --no-side --no-h2d --no-probe --check 0 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$S', round(d['value']), round(d['ms_per_step'],4))" || exit 1
done; done
