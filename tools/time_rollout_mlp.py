#!/usr/bin/env python3
"""Per-step time of the Actor-driven rollout: ec_rollout_mlp_policy (policy inside the looping env kernel) next to
per-step ec_policy_forward + ec_step launches.  usage: time_rollout_mlp.py [episodes] [B] [N]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch as th  # noqa: E402

from marl_amd.envs.edge_computing.multihop.edge_computing_env import BatchedEdgeComputingEnv  # noqa: E402
from marl_amd.utils import mlp_device  # noqa: E402
from tests.util import make_args  # noqa: E402

episodes = int(sys.argv[1]) if len(sys.argv) > 1 else 4
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
N = int(sys.argv[3]) if len(sys.argv) > 3 else 10
ns = make_args(["--task_size_max=6000", "--edge_node_num=%d" % N])


def run(fused):
    env = BatchedEdgeComputingEnv(ns, n_envs=B)
    T = env.episode_limit
    gen = th.Generator().manual_seed(1)
    dims = [(64, 9 + N), (64,), (64, 64), (64,), (N + 1, 64), (N + 1,)]
    params = tuple((th.randn(*d, generator=gen) * (0.3 if len(d) == 2 else 0.1)).to(env.device) for d in dims)
    hist = None
    times = []
    for ep in range(episodes):
        env.reset()
        env.observe()
        u = th.rand(T, B * N, device=env.device)
        a, b = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
        a.record()
        if fused:
            hist = env.rollout_mlp_policy(params, T, uniforms=u, history=hist)
        else:
            for t in range(T):
                _, _, act = mlp_device.policy_forward(params, env.obs, N, env.avail, u[t], want_probs=False)
                env.step(act.view(B, N))
        b.record()
        th.cuda.synchronize()
        times.append(a.elapsed_time(b) * 1e3 / T)
    st = env.status()
    live = env.live_tasks() / B
    env.close()
    return times, st, live


for fused in (False, True, False, True):
    times, st, live = run(fused)
    print("%-34s us/step per episode: %s  status=%d live/env=%.1f" % (
        "policy inside the rollout kernel" if fused else "policy + step launches (eager)",
        " ".join("%.2f" % x for x in times), st, live))
