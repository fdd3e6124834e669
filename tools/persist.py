#!/usr/bin/env python3
"""Diagnostic (EC_STAMPS build): are the slow blocks of a looping launch the same blocks (= the same envs) in the next
launch, and do they sit on the same CUs?"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch as th  # noqa: E402

from marl_amd import _abi  # noqa: E402
from marl_amd.envs.edge_computing.multihop.edge_computing_env import BatchedEdgeComputingEnv  # noqa: E402
from tests.util import make_args  # noqa: E402

B = 4096
env = BatchedEdgeComputingEnv(make_args(["--task_size_max=6000"]), n_envs=B)
L = _abi.lib()
L.ec_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]
act = th.empty(B, 10, dtype=th.int64, device=env.device)
env.reset()
env.observe()
env.random_actions(seed=0, counter=0, out=act)
env.rollout_random_policy(act, 40, seed=0)  # fill the queues


def launch(n):
    env.rollout_random_policy(act, n, seed=0)
    nb = B // env.info_geometry()["envs_per_block"] + 1
    buf = np.zeros((nb + 8) * 16, dtype=np.uint64)
    k = C.c_int64()
    L.ec_debug_stamps(env._h, buf.ctypes.data_as(C.c_void_p), C.byref(k))
    s = buf[: k.value * 16].reshape(k.value, 16)
    end = s[:, 8].astype(np.int64)
    cu = (s[:, 14] >> np.uint64(8)) & np.uint64(0xff) | ((s[:, 14] >> np.uint64(32)) << np.uint64(8))
    return (end - end.min()) / 100.0, cu


cur0 = env.debug_rng_cursors().astype(np.int64)
e1, c1 = launch(30)
cur1 = env.debug_rng_cursors().astype(np.int64)
e2, c2 = launch(30)
epb = env.info_geometry()["envs_per_block"]
draws = (cur1 - cur0)[:, :, 1].reshape(-1, epb * 10)          # failure-stream words per lane in launch 1
print("corr(block finishing time, failure draws of the block: sum %.2f, max lane %.2f)" % (
    np.corrcoef(e1, draws.sum(1))[0, 1], np.corrcoef(e1, draws.max(1))[0, 1]))
print("finishing-time spread: launch 1 mean %.1f max %.1f us; launch 2 mean %.1f max %.1f us" % (e1.mean(), e1.max(), e2.mean(), e2.max()))
print("corr(launch 1, launch 2) over blocks = %.2f;  blocks on the same CU in both launches: %.0f %%" % (
    np.corrcoef(e1, e2)[0, 1], 100.0 * (c1 == c2).mean()))
