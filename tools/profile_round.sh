#!/bin/bash
# The round's profile records in one GPU call (each pass a rocprofv3 run of its own, under its own time limit):
# kernel stats of the bench command, PMC HBM traffic of the three step kernels, SQ issue counters of the looping kernel.
# usage (GPU box, repo root): bash tools/profile_round.sh <tag>     -> gpurun_out/r03/prof_<tag>/
set -o pipefail
TAG=$1
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r03/prof_$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -o bench -- python3 $R/bench.py > $O/bench.out 2>&1 || { tail -5 $O/bench.out; exit 1; }
cd $R
for m in rollout step mlp; do
  bash tools/pmc_traffic.sh $O/traffic $m 600 > $O/traffic_$m.txt 2>&1 || { tail -5 $O/traffic_$m.txt; exit 1; }
  tail -3 $O/traffic_$m.txt
done
bash tools/pmc_sq.sh $O/sq 200 4096 10 rollout > $O/sq_rollout.txt 2>&1 || { tail -5 $O/sq_rollout.txt; exit 1; }
cat $O/sq_rollout.txt
find $O -name "*.csv" -size +2M -delete   # (per-dispatch counter dumps: the summaries above are what is kept)
