#!/bin/bash
# A/B of one env-kernel variant against another on one box: usage: bash tools/ab_env.sh "<ENV=.. for A>" "<ENV=.. for B>"
# (each side: bench.py without the PPO / CPU legs; prints us per env step of the episode launches)
mkdir -p gpurun_out/r03
i=0
for side in "$@"; do
  env $side timeout -k 10 200 python bench.py --ppo-iters 0 --cpu-seconds 0 --roofline-steps 100 > gpurun_out/r03/ab_$i.json 2> gpurun_out/r03/ab_$i.err || { tail -5 gpurun_out/r03/ab_$i.err; exit 1; }
  python -c "
import json
d=json.load(open('gpurun_out/r03/ab_$i.json')); r=d['roofline']
print('$side', 'us/env-step %.2f' % d['config']['us_per_env_step'], 'kernel us/step %.2f' % (r['kernel_avg_us']/100), 'L %.1f D %.1f' % (r['live_tasks_per_env'], r['rng_doubles_per_env_step']), 'one-step median %.1f' % d['per_step_launch']['kernel_median_us'])"
  i=$((i+1))
done
