#!/bin/bash
# Per-step time of the in-kernel-Actor rollout and of the per-step launches under every launch geometry ec_create can take
# (MARL_AMD_EC_EPW x MARL_AMD_EC_MW x MARL_AMD_EC_PRODUCERS), for one env shape.
# usage: bash tools/sweep_geometry.sh <envs> <nodes> <outfile> "<epw list>" "<mw list>"
B=$1; N=$2; OUT=$3; EPWS=${4:-1}; MWS=${5:-"1 2 3 4"}
: > $OUT
echo "default geometry" >> $OUT
timeout -k 10 120 python tools/time_rollout_mlp.py 2 $B $N 2>&1 | grep "us/step" >> $OUT || exit 1
for epw in $EPWS; do for mw in $MWS; do for pr in 0 1; do
  echo "epw=$epw mw=$mw producers=$pr" >> $OUT
  MARL_AMD_EC_EPW=$epw MARL_AMD_EC_MW=$mw MARL_AMD_EC_PRODUCERS=$pr timeout -k 10 120 python tools/time_rollout_mlp.py 2 $B $N 2>&1 | grep -E "us/step|Error|error" | head -4 >> $OUT
  rc=${PIPESTATUS[0]}; [ $rc = 124 ] || [ $rc = 137 ] || [ $rc = 134 ] || [ $rc = 139 ] && { echo "stopped: rc $rc" >> $OUT; exit 1; }
done; done; done
exit 0
