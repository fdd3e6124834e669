#!/usr/bin/env python3
"""Small driver for rocprofv3: N eager steps of the bench workload (policy kernel + ec_step), or -- 4th argument
"rollout" -- the same steps through ec_rollout_random_policy, one launch per episode."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch as th  # noqa: E402

from marl_amd.envs.edge_computing.multihop.edge_computing_env import BatchedEdgeComputingEnv  # noqa: E402
from tests.util import make_args  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
N = int(sys.argv[3]) if len(sys.argv) > 3 else 10
ns = make_args(["--task_size_max=6000", "--edge_node_num=%d" % N])
env = BatchedEdgeComputingEnv(ns, n_envs=B)
act = th.empty(B, N, dtype=th.int64, device=env.device)
if len(sys.argv) > 4 and sys.argv[4] == "rollout":
    for ep in range(max(1, steps // env.episode_limit)):
        env.reset()
        env.observe()
        env.random_actions(seed=0, counter=ep, out=act)
        env.rollout_random_policy(act, env.episode_limit, seed=0)
    th.cuda.synchronize()
    print("status", env.status(), "live/env", env.live_tasks() / B)
    sys.exit(0)
env.reset()
env.observe()
for t in range(steps):
    if t and t % env.episode_limit == 0:
        env.reset()
        env.observe()
    env.random_actions(seed=0, counter=t, out=act)
    env.step(act)
th.cuda.synchronize()
print("status", env.status(), "live/env", env.live_tasks() / B)
