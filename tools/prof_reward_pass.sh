#!/bin/bash
# per-kernel times of the GAIL reward pass (tools/bench_reward_pass.py) under rocprofv3: usage prof_reward_pass.sh <rows> <nodes> <tag>
set -o pipefail
O=$PWD/gpurun_out/r03/rp_$3; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o rp -- python3 $GRAFT_REPO_ROOT/tools/bench_reward_pass.py $1 $2 > $O/out.txt 2>&1 || { tail -5 $O/out.txt; exit 1; }
grep "^rows" $O/out.txt
f=$(find $O -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("%-90s calls %5s avg %10.1f us  total %8.2f ms" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
