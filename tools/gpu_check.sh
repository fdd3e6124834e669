#!/bin/bash
# One GPU round trip while tuning the env kernels: parity suite -> bench line -> phase stamps (each under its own
# time limit; stops at the first failure).  usage: bash tools/gpu_check.sh <tag> [tests|bench|stamps ...]
set -o pipefail
TAG=$1; shift
WHAT=${@:-tests bench stamps}
O=gpurun_out/r03; mkdir -p $O
for w in $WHAT; do
  case $w in
    tests) timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $O/gpu_$TAG.log 2>&1; rc=$?; tail -3 $O/gpu_$TAG.log; [ $rc = 0 ] || { grep -E "^(FAILED|ERROR|E  )" $O/gpu_$TAG.log | head -20; exit 1; };;
    envtests) timeout -k 10 600 python -u -m pytest tests/test_env_gpu.py tests/test_runner_gpu.py -m gpu -x -q --timeout 120 --timeout-method thread > $O/gpu_$TAG.log 2>&1; rc=$?; tail -3 $O/gpu_$TAG.log; [ $rc = 0 ] || { grep -E "^(FAILED|ERROR|E  )" $O/gpu_$TAG.log | head -20; exit 1; };;
    bench) timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_$TAG.json 2> $O/bench_$TAG.err || { tail -5 $O/bench_$TAG.err; exit 1; }
       python -c "
import json; d=json.load(open('$O/bench_$TAG.json'))
print('value %.4g  us/env-step %.2f  frac %.4f  one-step kernel median %.1f us  ppo %s' % (d['value'], d['config']['us_per_env_step'], d['roofline']['frac'], d['per_step_launch']['kernel_median_us'], {k: round(v, 2) for k, v in d['ppo_iteration'].items() if k.endswith('_ms')}))";;
    rehearse) timeout -k 10 300 python bench.py --gpus 2 --rehearse --steps 5 --warmup 2 --cpu-seconds 0 > $O/rehearse_$TAG.json 2> $O/rehearse_$TAG.err || { tail -5 $O/rehearse_$TAG.err; exit 1; }; python -c "import json; d=json.loads([l for l in open('$O/rehearse_$TAG.json') if l.startswith('{')][-1]); print('rehearse n_gpus', d['n_gpus'], 'value %.4g' % d['value'], 'ppo', d['ppo_iteration']['total_ms'])";;
    stamps) MARL_AMD_LIB=$PWD/tools/abl/${STAMPLIB:-lib_dev_stamps.so} timeout -k 10 200 python tools/stamps.py 4096 2>&1 | tail -26 || exit 1;;
    stampsroll) MARL_AMD_LIB=$PWD/tools/abl/${STAMPLIB:-lib_dev_stamps.so} STAMPS_ROLLOUT=20 timeout -k 10 200 python tools/stamps.py 4096 2>&1 | tail -14 | head -12 || exit 1;;
  esac
done
