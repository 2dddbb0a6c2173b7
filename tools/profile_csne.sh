#!/bin/bash
# Profile round of the CSNE tier on the GPU box:  tools/profile_csne.sh <tag> [leg]   -> gpurun_out/<tag>_csne_*
#   kernel stats (rocprofv3 --kernel-trace --stats) and the HBM traffic (separate FETCH_SIZE / WRITE_SIZE passes) of the
#   bench's `certificate_rejected` leg (512 problems of 4096 x 256, kappa(J) = 3e3, no bounds: all on the tier)
set -o pipefail
TAG=$1; LEG=${2:-certificate_rejected}
# (BLSQ_CSNE=0 in the environment profiles the CholeskyQR2 tier on the same leg; leg householder_only: the tree)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
RUN="python3 tools/bench_legs.py $LEG --steps 4 --check 0"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_csne_ks -o run -- $RUN > $O/${TAG}_csne_ks.log 2>&1 || exit 1
find $O/${TAG}_csne_ks -name "*kernel_stats.csv" -exec cp {} $O/${TAG}_csne_${LEG}_kernel_stats.csv \;
head -20 $O/${TAG}_csne_${LEG}_kernel_stats.csv | cut -c1-170
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/${TAG}_csne_$C -o run -- $RUN > $O/${TAG}_csne_$C.log 2>&1 || exit 1
done
python3 tools/pmc_traffic.py $O/${TAG}_csne_FETCH_SIZE $O/${TAG}_csne_WRITE_SIZE $O/${TAG}_csne_${LEG}_hbm_traffic.json 4096 256 512 $TAG
python3 - $O/${TAG}_csne_${LEG}_hbm_traffic.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["hbm_bytes"])[:12]:
    print("%-60s launches %4d  %8.3f GB per launch" % (k[:60], v["launches"], v["hbm_bytes_per_launch"] / 1e9))
PY
