#!/usr/bin/env python3
"""Phase breakdown of ec_step from the EC_STAMPS diagnostic build (MARL_AMD_LIB must point to it)."""
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

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ns = make_args(["--task_size_max=6000"])
env = BatchedEdgeComputingEnv(ns, n_envs=B)
L = _abi.lib()
L.ec_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]
act = th.empty(B, 10, dtype=th.int64, device=env.device)
env.reset()
env.observe()
names = ["entry->staged", "offload", "fail window", "mini-slots", "reduce", "generate+writeback",
         "(unused)", "observe", "producer: window+doubles (from entry)", "producer: regeneration (from offload end)",
         "passes of wave 0 through the per-mini-slot code", "per-mini-slot passes of lane 0"]
acc = []
subs = []
ROLL = int(os.environ.get("STAMPS_ROLLOUT", "0"))  # >1: steps per launch through the looping kernel (last one is stamped)
if ROLL:
    env.random_actions(seed=0, counter=0, out=act)
for t in range(160 if not ROLL else 160 // ROLL):
    if ROLL:
        env.rollout_random_policy(act, ROLL, seed=0)
    else:
        env.random_actions(seed=0, counter=t, out=act)
        env.step(act)
    if (t >= 60 and t % 10 == 0) if not ROLL else (t * ROLL >= 60):
        nb = env.info_geometry()["B"] // env.info_geometry()["envs_per_block"] + 1
        buf = np.zeros((nb + 8) * 32, dtype=np.uint64)
        n = C.c_int64()
        L.ec_debug_stamps(env._h, buf.ctypes.data_as(C.c_void_p), C.byref(n))
        sub = buf[n.value * 16: n.value * 32].reshape(n.value, 16).astype(np.int64)
        s = buf[: n.value * 16].reshape(n.value, 16).astype(np.int64)
        subs.append(sub.mean(0))
        seg = np.stack([s[:, 1] - s[:, 0], s[:, 2] - s[:, 1], s[:, 3] - s[:, 2], s[:, 4] - s[:, 3],
                        s[:, 5] - s[:, 4], s[:, 6] - s[:, 5], s[:, 7] - s[:, 6], s[:, 8] - s[:, 7]], 1)
        seg = np.concatenate([seg, np.stack([s[:, 11] - s[:, 0], s[:, 12] - s[:, 2], s[:, 9] * 100, s[:, 10] * 100], 1)], 1)
        restages = int((s[:, 13] > 0).sum()), int(s[:, 13].sum())
        q = len(s) // 4
        quart = [float((s[i * q:(i + 1) * q, 8] - s[i * q:(i + 1) * q, 0]).mean()) / 100.0 for i in range(4)]
        span = (s[:, 8].max() - s[:, 0].min())
        acc.append((seg.mean(0), seg.max(0), span, (s[:, 8] - s[:, 0]).mean(), s[:, 0].max() - s[:, 0].min()))
m = np.mean([a[0] for a in acc], 0) / 100.0  # wall_clock64 ticks at 100 MHz -> us
mx = np.mean([a[1] for a in acc], 0) / 100.0
print("epb", env.info_geometry()["envs_per_block"], "blocks span %.1f us, mean block %.1f us, start skew %.1f us" % (
    np.mean([a[2] for a in acc]) / 100.0, np.mean([a[3] for a in acc]) / 100.0, np.mean([a[4] for a in acc]) / 100.0))
print("mean block duration by quarter of the block index range: %s us" % " ".join("%.1f" % x for x in quart))
print("window restages, cumulative over the %d steps so far: %d blocks with any, %d lane-restages" % ((t + 1,) + restages))
for nm, a, b in zip(names, m, mx):
    print("  %-42s mean %6.2f us   max-block %6.2f us" % (nm, a, b))
if not ROLL:
    sn = ["next-event search + jump", "admission", "window check + draw fetch", "executing slots: classify", "executing slots: events",
          "executing slots: compaction", "waiting-queue sweep", "receiving queues: classify", "receiving queues: events",
          "loop exit"]
    sm = np.mean(subs, 0) / 100.0
    print("mini-slot loop of wave 0 by section (sum over its passes):")
    for nm, a in zip(sn, sm):
        print("  %-42s %6.2f us" % (nm, a))

