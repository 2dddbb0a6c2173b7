#!/bin/bash
# Round-3 full profile round on the GPU box:  tools/profile_round3.sh <tag>   -> gpurun_out/<tag>_*
#   tools/profile_round.sh (bench line, kernel stats, FETCH / WRITE passes, SQ MFMA pass) + kernel stats of the side
#   configs + the instruction-mix / LDS / wait counter passes of tools/pmc_gram.sh
TAG=$1
bash tools/profile_round.sh $TAG || exit 1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for C in c2-single c3 c4 c5; do
  bash tools/prof_config.sh $C > gpurun_out/${TAG}_prof_$C.txt 2>&1 || exit 1
  find gpurun_out/prof_$C -name "*kernel_stats.csv" -exec cp {} gpurun_out/${TAG}_${C}_kernel_stats.csv \;
done
bash tools/pmc_gram.sh $TAG || exit 1
echo "profile round $TAG done"
