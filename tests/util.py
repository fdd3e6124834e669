"""Shared helpers for the parity tests."""
import argparse
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def make_args(extra=(), **over):
    """Namespace built the way the reference's main.py:23-27 stacks its parsers."""
    from marl_amd.default_config import add_default_args
    from marl_amd.envs.edge_computing.multihop.edge_computing_config import add_edge_computing_env_args
    from marl_amd.rl.policy_gradient_rl.ppo.ppo_config import add_ppo_args
    p = argparse.ArgumentParser(conflict_handler="resolve")
    p = add_default_args(p)
    p = add_edge_computing_env_args(p)
    p = add_ppo_args(p)
    args = p.parse_args(list(extra))
    for k, v in over.items():
        setattr(args, k, v)
    return args


def args_from_golden(g):
    """Namespace carrying the env constants recorded in an env*.npz fixture."""
    args = make_args()
    ints = {"episode_limit", "mini_time_slot_num", "edge_node_num", "task_size_min", "task_size_max",
            "task_complexity_min", "task_complexity_max", "single_core_cpu_capacity", "transmission_rate_min",
            "transmission_rate_max", "edge_node_seed", "task_seed", "failure_seed", "link_seed", "k"}
    for k, v in zip(g["cfg_keys"], g["cfg_vals"]):
        k = str(k)
        setattr(args, k, int(v) if k in ints else float(v))
    args.cpu_core_list = [int(x) for x in g["cpu_core_list"]]
    return args


def golden_cases(g):
    out = []
    for s in g["cases"]:
        name, k, pol, eps = str(s).split(":")
        out.append((name, int(k), pol, int(eps)))
    return out
