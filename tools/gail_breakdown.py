#!/usr/bin/env python3
"""Where one GAIL training call spends its time (wall clock with synchronisation around each stage)."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch as th  # noqa: E402

from marl_amd.agent_register import agent_config_register, agent_register  # noqa: E402
from marl_amd.default_config import add_default_args  # noqa: E402
from marl_amd.envs.edge_computing.multihop.edge_computing_config import add_edge_computing_env_args  # noqa: E402
from marl_amd.runners.runner_register import runner_register  # noqa: E402

envs = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nodes = int(sys.argv[2]) if len(sys.argv) > 2 else 10
p = argparse.ArgumentParser(conflict_handler="resolve")
p = agent_config_register["gail"](add_edge_computing_env_args(add_default_args(p)))
ns = p.parse_args(["--task_size_max=6000", "--edge_node_num=%d" % nodes])
ns.n_threads, ns.obs_shape, ns.n_agents, ns.n_actions = envs, 9, nodes, nodes + 1
ns.state_shape, ns.device = 9 * nodes, "cuda"
ns.ec_graph_rollout = 1
th.manual_seed(100)
agent = agent_register["gail"](ns)
runner = runner_register[ns.runner](ns, agent)
acc = {}


def wrap(name):
    fn = getattr(agent, name)

    def timed(*a, **k):
        th.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn(*a, **k)
        th.cuda.synchronize()
        acc.setdefault(name, []).append(time.perf_counter() - t0)
        return r
    setattr(agent, name, timed)


for n in ("_discriminator_step", "_train_fused", "compute_bc_loss", "train"):
    wrap(n)
for _ in range(5):
    runner.run(test_mode=False)
for k, v in acc.items():
    print("%-24s calls/iter %.1f  last-iteration total %.1f ms" % (k, len(v) / 5.0, 1e3 * sum(v[-max(1, len(v) // 5):])))
