#!/bin/bash
# MFMA-busy counter pass over the fused update kernels (k_ppo, k_diff) at 4.096 M rows.
# usage (on the GPU box, from the repo root): bash tools/pmc_mfma.sh <outdir>
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc -o p -- python3 $R/tools/time_ppo_minibatch.py 4096000 > $O/pmc.log 2>&1 || { tail -5 $O/pmc.log; exit 1; }
timeout -k 10 100 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 $R/tools/time_ppo_minibatch.py 4096000 > $O/trace.log 2>&1 || { tail -5 $O/trace.log; exit 1; }
python3 $R/tools/pmc_summary.py $O/pmc | grep -E "k_ppo|k_diff"
grep -E "k_ppo|k_diff" $O/trace/t_kernel_stats.csv | cut -d, -f1-4 | cut -c1-160
