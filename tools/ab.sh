#!/bin/bash
# A/B builds of the library in one GPU call: tools/ab.sh "libA.so libB.so ..." [bench args]
LIBS=$1; shift 1
for rep in 1 2; do
for L in $LIBS; do
  BLSQ_LIB=$PWD/bounded-lsq_amd/bounded_lsq/$L timeout -k 10 200 python bench.py --steps 4 --warmup 1 --no-cpu --no-householder "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', round(d['value']), round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['kernels_ms_per_step'].items()})" || exit 1
done; done
