#!/bin/bash
# A/B builds with the per-kernel slots: tools/ab_slots.sh "libA.so libB.so" [bench args]
LIBS=$1; shift 1
for rep in 1 2; do
for L in $LIBS; do
  BLSQ_LIB=$PWD/bounded-lsq_amd/bounded_lsq/$L timeout -k 10 200 python bench.py --no-cpu --no-householder --no-side --no-h2d --no-probe --check 0 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', round(d['value']), round(d['ms_per_step'],4), {k: round(v,3) for k,v in d['kernels_ms_per_step'].items() if v > 0})" || exit 1
done; done
