#!/usr/bin/env python3
"""ec_linear2_fwd_bn and the statistics-only pass of layer 1 at the reward pass's shape, a few launches each (for
rocprofv3 --kernel-trace / --pmc).  usage: time_linear2.py [rows] [k_in]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch as th  # noqa: E402

from marl_amd.utils import linear_device  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 4096000
k_in = int(sys.argv[2]) if len(sys.argv) > 2 else 30
gen = th.Generator().manual_seed(1)
x = th.randn(rows, k_in, generator=gen).cuda()
w1, b1 = (th.randn(128, k_in, generator=gen) * 0.3).cuda(), th.randn(128, generator=gen).cuda()
w2, b2 = (th.randn(64, 128, generator=gen) * 0.2).cuda(), th.randn(64, generator=gen).cuda()
scale, shift = (th.rand(128, generator=gen) + 0.5).cuda(), (th.randn(128, generator=gen) * 0.5).cuda()
with th.no_grad():
    for _ in range(3):
        linear_device.fwd_stats_only(x, w1, b1)
        linear_device.fwd2_bn(x, w1, b1, scale, shift, 7, 0.3, w2, b2, want_stats=True)
    th.cuda.synchronize()
    a, b = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
    for name, fn in (("stats-only layer 1", lambda: linear_device.fwd_stats_only(x, w1, b1)),
                     ("ec_linear2_fwd_bn", lambda: linear_device.fwd2_bn(x, w1, b1, scale, shift, 7, 0.3, w2, b2, want_stats=True))):
        a.record()
        for _ in range(5):
            fn()
        b.record()
        th.cuda.synchronize()
        print("%s: %.3f ms" % (name, a.elapsed_time(b) / 5))
    obs = th.randn(rows, 9, device="cuda")
    acts = th.randint(0, 11, (rows,), device="cuda")
    for _ in range(2):
        linear_device.disc_rows(obs.view(-1, 10, 9), acts.view(-1, 10), 10, 11)
    a.record()
    for _ in range(5):
        linear_device.disc_rows(obs.view(-1, 10, 9), acts.view(-1, 10), 10, 11)
    b.record()
    th.cuda.synchronize()
    print("ec_disc_rows (10 nodes): %.3f ms" % (a.elapsed_time(b) / 5))
