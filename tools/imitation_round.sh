#!/bin/bash
# GAIL / DGAIL iteration times of the round (profiles/rNN/imitation_iterations_final.jsonl).  usage: imitation_round.sh <tag>
set -o pipefail
O=gpurun_out/r03; mkdir -p $O
F=$O/imitation_$1.jsonl
: > $F
run() { timeout -k 10 $1 python tools/bench_imitation.py "${@:2}" >> $F 2>$O/imit.err || { tail -5 $O/imit.err; exit 1; }; }
run 200 --algo gail --envs 4096 --iters 6
run 200 --algo dgail --envs 4096 --iters 6
run 200 --algo dgail --envs 4096 --iters 6 --diffusion-reward
run 200 --algo dgail --envs 1024
run 250 --algo gail --envs 2048 --nodes 40
run 400 --algo gail --envs 16384 --nodes 40 --iters 4
cut -c1-260 $F
