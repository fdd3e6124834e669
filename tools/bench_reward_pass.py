#!/usr/bin/env python3
"""The GAIL reward pass (Discriminator.forward_rows under no_grad, training mode) four ways on one box: layer by layer,
tails folded into the next layer (ec_linear_fwd_bn), with the next BatchNorm's statistics from the same kernel, and with
layers 1 and 2 in one kernel (ec_linear2_fwd_bn, the product's path)."""
import os
import sys
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch as th  # noqa: E402

import marl_amd.rl.policy_gradient_rl.gail.gail_network as gn  # noqa: E402
from marl_amd.utils import linear_device  # noqa: E402

linear_device.FUSE2_MAX_K = int(os.environ.get("FUSE2_MAX_K", linear_device.FUSE2_MAX_K))
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 4096000
nodes = int(sys.argv[2]) if len(sys.argv) > 2 else 10
args = SimpleNamespace(obs_shape=9, n_agents=nodes, n_actions=nodes + 1, hidden_dim=64, activate_fun="relu")
disc = gn.Discriminator(args).cuda()
x = th.randn(rows, 9 + 2 * nodes + 1, device="cuda")


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    th.cuda.synchronize()
    a, b = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    th.cuda.synchronize()
    return a.elapsed_time(b) / reps


with th.no_grad():
    t_two = timed(lambda: disc.forward_rows(x))
    gn.Discriminator.FUSE_FIRST_TWO = False
    t_full = timed(lambda: disc.forward_rows(x))
    real = linear_device.fwd_bn
    linear_device.fwd_bn = lambda *a, want_stats=False, **k: real(*a, **k)
    orig = gn.Discriminator._forward_rows_fused_tails

    def no_stats(self, x_, sync=False):
        x_ = self.fc1(x_)
        for i in (1, 2, 3):
            bn, drop, nxt = getattr(self, "bn%d" % i), getattr(self, "dropout%d" % i), getattr(self, "fc%d" % (i + 1))
            running = (bn.running_mean, bn.running_var, bn.num_batches_tracked, bn.momentum)
            _, _, _, sc, sh = gn._bn_batch_stats(x_.contiguous(), bn.weight, bn.bias, bn.eps, running)
            x_ = real(x_, sc, sh, gn._draw_seed(), float(drop.p), nxt.weight, nxt.bias)
        return x_
    gn.Discriminator._forward_rows_fused_tails = no_stats
    t_tails = timed(lambda: disc.forward_rows(x))
    gn.Discriminator._forward_rows_fused_tails = orig
    linear_device.fwd_bn = real
    gn.Discriminator._fused_tails_ok = lambda self, x_: False
    t_layers = timed(lambda: disc.forward_rows(x))
print("rows %d, %d nodes: layer by layer %.2f ms, tails folded %.2f ms, + statistics from the same kernel %.2f ms, "
      "layers 1 and 2 in one kernel %.2f ms" % (rows, nodes, t_layers, t_tails, t_full, t_two))
