#!/bin/bash
# The round's record in one GPU call: full parity suite -> bench line -> rocprofv3 kernel stats of the same bench command
# -> GAIL / DGAIL iteration times.  Every GPU step under its own time limit; stops at the first failure.
# usage: bash tools/measure_round.sh <tag>
set -o pipefail
TAG=$1
O=gpurun_out/r02; mkdir -p $O
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $O/gpu_$TAG.log 2>&1; rc=$?
tail -3 $O/gpu_$TAG.log; [ $rc = 0 ] || { grep -E "^(FAILED|ERROR|E  )" $O/gpu_$TAG.log | head -20; exit 1; }
timeout -k 10 300 python bench.py > $O/bench_$TAG.json 2> $O/bench_$TAG.err || { tail -5 $O/bench_$TAG.err; exit 1; }
cat $O/bench_$TAG.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_bench_$TAG -o bench -- python3 $GRAFT_REPO_ROOT/bench.py > $GRAFT_REPO_ROOT/$O/prof_bench_$TAG.out 2>&1 || { tail -5 $GRAFT_REPO_ROOT/$O/prof_bench_$TAG.out; exit 1; }
cd $GRAFT_REPO_ROOT
: > $O/imitation_$TAG.jsonl
timeout -k 10 200 python tools/bench_imitation.py --algo gail --envs 4096 >> $O/imitation_$TAG.jsonl 2>$O/imit.err || { tail -5 $O/imit.err; exit 1; }
timeout -k 10 200 python tools/bench_imitation.py --algo dgail --envs 4096 >> $O/imitation_$TAG.jsonl 2>$O/imit.err || { tail -5 $O/imit.err; exit 1; }
timeout -k 10 200 python tools/bench_imitation.py --algo dgail --envs 4096 --diffusion-reward >> $O/imitation_$TAG.jsonl 2>$O/imit.err || { tail -5 $O/imit.err; exit 1; }
timeout -k 10 200 python tools/bench_imitation.py --algo dgail --envs 1024 >> $O/imitation_$TAG.jsonl 2>$O/imit.err || { tail -5 $O/imit.err; exit 1; }
timeout -k 10 250 python tools/bench_imitation.py --algo gail --envs 2048 --nodes 40 >> $O/imitation_$TAG.jsonl 2>$O/imit.err || { tail -5 $O/imit.err; exit 1; }
timeout -k 10 400 python tools/bench_imitation.py --algo gail --envs 16384 --nodes 40 --iters 4 >> $O/imitation_$TAG.jsonl 2>$O/imit.err || { tail -5 $O/imit.err; exit 1; }
cat $O/imitation_$TAG.jsonl
