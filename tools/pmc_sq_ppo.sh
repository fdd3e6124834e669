#!/bin/bash
# SQ counter passes over the fused update kernels (k_ppo, k_diff) at 4.096 M rows: instruction mix, issue and wait cycles.
# usage (on the GPU box, from the repo root): bash tools/pmc_sq_ppo.sh <outdir>
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for c in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM" \
         "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_VALU_MFMA_BUSY_CYCLES" \
         "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL"; do
  i=$((i+1))
  timeout -s KILL 150 rocprofv3 --pmc $c --output-format csv -d $O/set$i -o p -- python3 $R/tools/time_ppo_minibatch.py 4096000 > $O/set$i.log 2>&1 || { echo FAIL set$i; tail -5 $O/set$i.log; exit 1; }
  python3 $R/tools/pmc_summary.py $O/set$i | grep -E "k_ppo<|k_diff" || true
done
