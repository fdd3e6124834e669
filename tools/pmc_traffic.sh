#!/bin/bash
# HBM traffic of the step kernels by PMC counters: one rocprofv3 --pmc run per counter (FETCH_SIZE and WRITE_SIZE do not
# fit one pass), over tools/prof_step.py.  usage (GPU box, repo root): bash tools/pmc_traffic.sh <outdir> <mode> [steps]
#   mode: rollout (k_step looping) | step (k_step one step per launch) | mlp (k_rollout_mlp)
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$1; MODE=$2; STEPS=${3:-600}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  if [ "$MODE" = step ]; then ARGS="$STEPS 4096 10"; else ARGS="$STEPS 4096 10 $MODE"; fi
  timeout -s KILL 150 rocprofv3 --pmc $c --output-format csv -d $O/${MODE}_$c -o p -- python3 $R/tools/prof_step.py $ARGS > $O/${MODE}_$c.log 2>&1 || { echo FAIL $c; tail -5 $O/${MODE}_$c.log; exit 1; }
done
python3 $R/tools/pmc_traffic_summary.py $O $MODE
