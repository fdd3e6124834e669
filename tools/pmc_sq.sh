#!/bin/bash
# SQ / SQC counter passes over the looping step kernel (one rocprofv3 run per counter set, as the pool requires).
# usage (on the GPU box, from the repo root): bash tools/pmc_sq.sh <outdir> [prof_step.py args...]
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$1; shift
ARGS=${@:-200 4096 10 rollout}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for c in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM" \
         "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" \
         "SQ_IFETCH SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_FLAT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" \
         "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_INST_REQ"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $O/set$i -o p -- python3 $R/tools/prof_step.py $ARGS > $O/set$i.log 2>&1 || { echo FAIL set$i; tail -5 $O/set$i.log; exit 1; }
  python3 $R/tools/pmc_summary.py $O/set$i | grep -E "k_step|k_rollout" || true
done
