#!/bin/bash
# SQ counter passes over ec_linear2_fwd_bn (one rocprofv3 run per set).  usage: bash tools/pmc_linear2.sh <outdir>
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for c in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
         "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_VMEM"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $O/set$i -o p -- python3 $R/tools/time_linear2.py > $O/set$i.log 2>&1 || { echo FAIL set$i; tail -5 $O/set$i.log; exit 1; }
  python3 $R/tools/pmc_summary.py $O/set$i | grep -E "k_linear" || true
done
