#!/usr/bin/env python3
"""Diagnostic (EC_STAMPS build, MARL_AMD_LIB): per-block duration of the last step of a looping launch against where
the block's main and producer waves run (XCC, SE, CU, SIMD from HW_ID)."""
import ctypes as C
import collections
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
ns = make_args(["--task_size_max=6000"])
env = BatchedEdgeComputingEnv(ns, n_envs=B)
L = _abi.lib()
L.ec_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]
act = th.empty(B, 10, dtype=th.int64, device=env.device)
env.reset()
env.observe()
env.random_actions(seed=0, counter=0, out=act)
env.rollout_random_policy(act, 60, seed=0)
nb = B // env.info_geometry()["envs_per_block"] + 1
buf = np.zeros((nb + 8) * 32, dtype=np.uint64)  # (the library copies 32 words per block)
n = C.c_int64()
L.ec_debug_stamps(env._h, buf.ctypes.data_as(C.c_void_p), C.byref(n))
s = buf[: n.value * 16].reshape(n.value, 16)
end = s[:, 8].astype(np.int64)
dur = (s[:, 8].astype(np.int64) - s[:, 0].astype(np.int64)) / 100.0


def place(w):
    hw, xcc = int(w) & 0xffffffff, (int(w) >> 32) & 0xf
    return xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15, (hw >> 4) & 3  # xcc, se, sh, cu, simd


cus = collections.defaultdict(list)
for b in range(n.value):
    m, p = place(s[b, 14]), place(s[b, 15])
    cus[m[:4]].append((m[4], p[4], end[b], dur[b]))
print("blocks", n.value, "CUs used", len(cus), "blocks/CU", collections.Counter(len(v) for v in cus.values()))
# finishing time of the launch's last step relative to the earliest block, by whether the main wave shares its SIMD
t0 = end.min()
shared, alone = [], []
for cu, v in cus.items():
    simds = collections.Counter(x[0] for x in v)
    for msimd, psimd, e, du in v:
        (shared if simds[msimd] > 1 else alone).append(((e - t0) / 100.0, du))
for name, arr in (("main wave shares its SIMD with another main", shared), ("main wave alone on its SIMD", alone)):
    if arr:
        a = np.array(arr)
        print("%-46s n=%4d  end of step 60 after the first block: mean %6.1f us  max %6.1f us   last-step duration mean %5.1f us"
              % (name, len(a), a[:, 0].mean(), a[:, 0].max(), a[:, 1].mean()))
print("main SIMD histogram", collections.Counter(place(s[b, 14])[4] for b in range(n.value)))
print("producer SIMD histogram", collections.Counter(place(s[b, 15])[4] for b in range(n.value)))
# per XCC, and against the block's load (tasks alive in its envs at the end)
byx = collections.defaultdict(list)
for b in range(n.value):
    byx[place(s[b, 14])[0]].append((end[b] - t0) / 100.0)
print("end of last step after the first block, by XCC:", {k: round(float(np.mean(v)), 1) for k, v in sorted(byx.items())})
epb = env.info_geometry()["envs_per_block"]
th.cuda.synchronize()
import ctypes
P = B * 10
def dl(name):
    return None
live = (env.debug_dump_counts() if hasattr(env, "debug_dump_counts") else None)
if live is None:
    # tasks per env from the public dump (slow path: sample 256 envs)
    idx = np.linspace(0, B - 1, 256).astype(int)
    load = np.array([len(env.debug_dump(int(e))) for e in idx])
    blk = idx // epb
    e_rel = (end[blk] - t0) / 100.0
    print("corr(block finishing time, tasks alive in a sampled env of the block) = %.2f" % np.corrcoef(e_rel, load)[0, 1])
cu_mean = {cu: float(np.mean([(x[2] - t0) / 100.0 for x in v])) for cu, v in cus.items()}
within = float(np.mean([np.std([(x[2] - t0) / 100.0 for x in v]) for v in cus.values()]))
vals = np.array(list(cu_mean.values()))
print("per-CU mean finishing time: std across CUs %.1f us, mean std within a CU %.1f us; slowest CUs:" % (vals.std(), within),
      sorted(((round(v, 1), cu) for cu, v in cu_mean.items()), reverse=True)[:6])
bycu = collections.defaultdict(list)
for cu, v in cu_mean.items():
    bycu[cu[3]].append(v)
print("by cu_id within the SE/SH:", {k: round(float(np.mean(v)), 1) for k, v in sorted(bycu.items())})
grp = collections.defaultdict(list)
for b in range(n.value):
    m, p = place(s[b, 14]), place(s[b, 15])
    grp[("main simd", m[4])].append((end[b] - t0) / 100.0)
    grp[("prod simd", p[4])].append((end[b] - t0) / 100.0)
    grp[("main wave slot", int(s[b, 14]) & 15)].append((end[b] - t0) / 100.0)
    grp[("main-prod simd distance", (p[4] - m[4]) % 4)].append((end[b] - t0) / 100.0)
for k in sorted(grp):
    print(k, "n=%d mean %.1f" % (len(grp[k]), np.mean(grp[k])))
# does a block's rank within its CU (by block index) matter?
rank = collections.defaultdict(list)
for cu, v in cus.items():
    order = np.argsort([x[2] for x in v])
blk_by_cu = collections.defaultdict(list)
for b in range(n.value):
    blk_by_cu[place(s[b, 14])[:4]].append(b)
for cu, bl in blk_by_cu.items():
    for r, b in enumerate(sorted(bl)):
        rank[r].append((end[b] - t0) / 100.0)
print("by order of block index within the CU:", {r: round(float(np.mean(v)), 1) for r, v in sorted(rank.items())})
rk = collections.defaultdict(list)
for cu, bl in blk_by_cu.items():
    for r, b in enumerate(sorted(bl)):
        rk[r].append(dur[b])
print("last-step duration by order of block index within the CU:", {r: round(float(np.mean(v)), 2) for r, v in sorted(rk.items())})
entry = s[:, 13].astype(np.int64)
print("kernel entry of the block after the first block to enter: mean %.1f us, max %.1f us; by order of block index within the CU:"
      % ((entry - entry.min()).mean() / 100.0, (entry - entry.min()).max() / 100.0),
      {r: round(float(np.mean([(entry[b] - entry.min()) / 100.0 for b in bl_r])), 1)
       for r, bl_r in sorted({r: [sorted(bl)[r] for bl in blk_by_cu.values() if len(bl) > r] for r in range(4)}.items())})
gap, body = s[:, 9].astype(np.int64) / 100.0, s[:, 10].astype(np.int64) / 100.0
for name, arr in (("between step bodies (end barrier + geometry reload)", gap), ("inside step bodies", body)):
    print("%s, us per step by order of block index within the CU:" % name,
          {r: round(float(np.mean([arr[sorted(bl)[r]] for bl in blk_by_cu.values() if len(bl) > r])) / 60.0, 2) for r in range(4)})
for cu in list(blk_by_cu)[:4]:
    print("CU", cu, [(r, "main simd %d slot %d" % (place(s[b, 14])[4], int(s[b, 14]) & 15), "prod simd %d slot %d" % (place(s[b, 15])[4], int(s[b, 15]) & 15),
                     round(body[b] / 60.0, 1)) for r, b in enumerate(sorted(blk_by_cu[cu]))])
if os.environ.get("STAMPS_BARRIER"):  # -DEC_STAMPS_BARRIER build: slot 13 holds the end-of-step barrier's total instead
    print("end-of-step barrier alone, us per step by order of block index within the CU:",
          {r: round(float(np.mean([entry[sorted(bl)[r]] for bl in blk_by_cu.values() if len(bl) > r])) / 100.0 / 60.0, 2) for r in range(4)})
