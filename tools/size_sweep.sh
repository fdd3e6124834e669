#!/bin/bash
# us per env step of the looping kernel at several env counts (bench.py without the PPO / CPU legs)
mkdir -p gpurun_out/r03
for B in "$@"; do
  timeout -k 10 200 python bench.py --envs $B --ppo-iters 0 --cpu-seconds 0 --roofline-steps 100 > gpurun_out/r03/sweep_$B.json 2> gpurun_out/r03/sweep_$B.err || { tail -5 gpurun_out/r03/sweep_$B.err; exit 1; }
  python -c "
import json
d=json.load(open('gpurun_out/r03/sweep_$B.json')); r=d['roofline']
print('$B envs: us/env-step %.2f' % d['config']['us_per_env_step'], 'kernel us/step %.2f' % (r['kernel_avg_us']/100), 'value %.4g' % d['value'], 'one-step median %.1f' % d['per_step_launch']['kernel_median_us'])"
done
