import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch as th
from marl_amd import _abi
from marl_amd.envs.edge_computing.multihop.edge_computing_env import BatchedEdgeComputingEnv
from tests.util import make_args
L = _abi.lib()
L.ec_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]
B, N = 4096, 10
ns = make_args(["--task_size_max=6000"])
env = BatchedEdgeComputingEnv(ns, n_envs=B)
gen = th.Generator().manual_seed(1)
dims = [(64, 9 + N), (64,), (64, 64), (64,), (N + 1, 64), (N + 1,)]
scale = float(os.environ.get("STAMPS_WEIGHT_SCALE", "1"))  # 0: a uniform policy over the available actions (the random policy's load)
params = tuple((th.randn(*d, generator=gen) * (0.3 if len(d) == 2 else 0.1) * scale).to(env.device) for d in dims)
T = 100
for ep in range(3):
    env.reset(); env.observe()
    u = th.rand(T, B * N, device=env.device)
    env.rollout_mlp_policy(params, T, uniforms=u)
th.cuda.synchronize()
buf = np.zeros((B, 16), dtype=np.uint64)
n = C.c_int64(0)
L.ec_debug_stamps(env._h, buf.ctypes.data_as(C.c_void_p), C.byref(n))
buf = buf[: n.value].astype(np.float64) / 100.0 / T  # wall_clock64 ticks at 100 MHz -> us per step
for name, i in (("policy tile (wave 0)", 9), ("wait others + action", 10), ("step body", 11), ("loop gap", 12)):
    print("%-24s mean %.2f us  p10 %.2f  p90 %.2f" % (name, buf[:, i].mean(), np.percentile(buf[:, i], 10), np.percentile(buf[:, i], 90)))
if os.environ.get("STAMPS_POLICY"):  # EC_STAMPS_POLICY build: slots 0..4 hold the policy's sections instead
    for i, nm in enumerate(["first round trip + layer 1", "layer 2", "layer 3", "logits exchange", "mask + softmax + draw"]):
        print("  policy: %-28s mean %6.2f us per step" % (nm, buf[:, i].mean()))
    sys.exit(0)
# phases of the LAST step body of the launch (EC_STAMP points 0..8 of step_body; 9..12 were overwritten by the totals above)
raw = np.zeros((B, 16), dtype=np.uint64)
L.ec_debug_stamps(env._h, raw.ctypes.data_as(C.c_void_p), C.byref(n))
s = raw[: n.value].astype(np.int64)
names = ["entry->staged", "offload", "fail window", "mini-slots", "reduce", "generate+writeback", "(unused)", "observe"]
for i, nm in enumerate(names):
    print("  %-24s mean %6.2f us" % (nm, (s[:, i + 1] - s[:, i]).mean() / 100.0))
