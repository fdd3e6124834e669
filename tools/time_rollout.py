#!/usr/bin/env python3
"""Per-step time of the env loop through ec_rollout_random_policy (one launch per `chunk` steps) next to the
per-step launches of ec_step_random_policy.  usage: time_rollout.py [episodes] [B] [N] [chunk]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch as th  # noqa: E402

from marl_amd.envs.edge_computing.multihop.edge_computing_env import BatchedEdgeComputingEnv  # noqa: E402
from tests.util import make_args  # noqa: E402

episodes = int(sys.argv[1]) if len(sys.argv) > 1 else 6
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
N = int(sys.argv[3]) if len(sys.argv) > 3 else 10
chunk = int(sys.argv[4]) if len(sys.argv) > 4 else 100
ns = make_args(["--task_size_max=6000", "--edge_node_num=%d" % N])


def run(fused):
    env = BatchedEdgeComputingEnv(ns, n_envs=B)
    act = th.empty(B, N, dtype=th.int64, device=env.device)
    T = env.episode_limit
    times = []
    for ep in range(episodes):
        env.reset()
        env.observe()
        env.random_actions(seed=0, counter=ep, out=act)
        a, b = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
        a.record()
        if fused:
            for t in range(0, T, chunk):
                env.rollout_random_policy(act, min(chunk, T - t), seed=0, counter=0)
        else:
            for t in range(T):
                env.step_random_policy(act, seed=0, counter=0)
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
        "rollout kernel, %d steps/launch" % chunk if fused else "one launch per step (eager)",
        " ".join("%.2f" % x for x in times), st, live))
