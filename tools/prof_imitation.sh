#!/bin/bash
# per-kernel times of GAIL / DGAIL iterations (tools/bench_imitation.py) under rocprofv3.
# usage: prof_imitation.sh <tag> <bench_imitation args...>
set -o pipefail
TAG=$1; shift
O=$PWD/gpurun_out/r03/imit_$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o p -- python3 $GRAFT_REPO_ROOT/tools/bench_imitation.py "$@" > $O/out.txt 2>&1 || { tail -5 $O/out.txt; exit 1; }
tail -1 $O/out.txt | cut -c1-200
f=$(find $O -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.1f ms" % (tot / 1e6))
for r in rows[:32]:
    print("%-100s calls %5s avg %9.1f us  total %8.2f ms" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
