#!/usr/bin/env python3
"""Small driver for rocprofv3: N eager steps of the bench workload (policy kernel + ec_step), or -- 4th argument
"rollout" -- the same steps through ec_rollout_random_policy, one launch per episode; "mlp" -- through
ec_rollout_mlp_policy (random Actor weights)."""
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
if len(sys.argv) > 4 and sys.argv[4] == "mlp":
    gen = th.Generator().manual_seed(1)
    dims = [(64, 9 + N), (64,), (64, 64), (64,), (N + 1, 64), (N + 1,)]
    params = tuple((th.randn(*d, generator=gen) * (0.3 if len(d) == 2 else 0.1)).to(env.device) for d in dims)
    hist = None
    for ep in range(max(1, steps // env.episode_limit)):
        env.reset()
        env.observe()
        u = th.rand(env.episode_limit, B * N, device=env.device)
        hist = env.rollout_mlp_policy(params, env.episode_limit, uniforms=u, history=hist)
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
