#!/usr/bin/env python3
"""Launch time of the fused actor / critic passes at DGAIL's minibatch size (rows of a 4.096 M-row table picked through
row_index) and at full batch; with MARL_AMD_LIB=tools/abl/lib_ppo_noflush.so the same without the gradient flush.

    python tools/time_ppo_minibatch.py [rows_per_pass ...]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch as th  # noqa: E402

from marl_amd.rl.policy_gradient_rl.dgail.dgail_network import PolicyNet, ValueNet  # noqa: E402
from marl_amd.utils import ppo_device  # noqa: E402

dev = "cuda"
N, A, R = 10, 11, 4096000
sizes = [int(x) for x in sys.argv[1:]] or [32768, 262144, 1048576, 4096000]
th.manual_seed(0)
pol, val = PolicyNet(9 + N, A, 64).to(dev), ValueNet(9, 64).to(dev)
obs = th.rand(R, 9, device=dev)
act = th.randint(0, A, (R,), device=dev)
adv, ret, lp_old = th.randn(R, device=dev), th.randn(R, device=dev), -th.rand(R, device=dev) * 2
fp, fv = ppo_device.FlatGrads(pol), ppo_device.FlatGrads(val)
losses = th.zeros(2, device=dev)


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    th.cuda.synchronize()
    e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    th.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print("lib:", os.environ.get("MARL_AMD_LIB", "default"))
for n in sizes:
    idx = th.randperm(R, device=dev)[:n].contiguous() if n < R else None
    ta = timed(lambda: ppo_device.actor_pass(pol, obs, None, act, None, N, old_logp=lp_old, adv=adv, eps_clip=0.2,
                                             entropy_coef=8e-4, inv_denom=1.0 / n, grads=fp.flat, loss=losses[0:1],
                                             want_entropy=True, row_index=idx, logits=True))
    tc = timed(lambda: ppo_device.critic_pass(val, obs, None, 0, target=ret, inv_denom=0.2 / n, grads=fv.flat,
                                              loss=losses[1:2], row_index=idx))
    tiles = (n + 31) // 32
    print("rows %8d  tiles/wave %6.1f  actor %8.1f us  critic %8.1f us" % (n, tiles / 1024.0, ta, tc))

# the diffusion discriminator's compute_loss call at the largest size (ec_diffusion_grad: forward only / forward + backward)
from marl_amd.rl.policy_gradient_rl.dgail.dgail_network import DiffusionDiscriminator  # noqa: E402
from marl_amd.utils import mlp_device  # noqa: E402

n = max(sizes)
disc = DiffusionDiscriminator(9 + A, 64).to(dev)
t = th.randint(0, 2, (n,), device=dev)
noise = th.randn(n, 9 + A, device=dev)
coef = th.randn(n, device=dev) / n
call = mlp_device.DiffusionLossCall(disc, 1, obs[:n], act[:n], A, t, noise)
tf = timed(lambda: call.loss(), reps=10)
tb = timed(lambda: call.backward(coef), reps=10)
print("rows %8d  k_diff forward %8.1f us  forward+backward %8.1f us" % (n, tf, tb))
