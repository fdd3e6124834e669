#!/usr/bin/env python3
"""ec_linear_wgrad against torch's weight-gradient products at the GAIL discriminator's shapes."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch as th  # noqa: E402

from marl_amd.utils import linear_device  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 4096000


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
    return a.elapsed_time(b) / reps * 1e-3


th.set_grad_enabled(False)
for c_out, k_in in ((128, 30), (64, 128), (32, 64)):
    gy, x = th.randn(rows, c_out, device="cuda"), th.randn(rows, k_in, device="cuda")
    gb = rows * (c_out + k_in) * 4 / 1e9
    c = 1024
    full = (rows // c) * c
    t_hip = timed(lambda: linear_device.wgrad(gy, x))
    t_bmm = timed(lambda: (th.bmm(gy[:full].reshape(-1, c, c_out).transpose(1, 2), x[:full].reshape(-1, c, k_in)).sum(0),
                           gy[:full].reshape(-1, c, c_out).sum(1).sum(0)))
    t_mm = timed(lambda: (gy.t() @ x, gy.sum(0)), reps=3)
    print("rows %d  dW [%3d x %3d]  ec_linear_wgrad %7.1f us (%.2f TB/s)   chunked bmm + sums %7.1f us   gy.t() @ x + sum %8.1f us"
          % (rows, c_out, k_in, t_hip * 1e6, gb / t_hip / 1e3, t_bmm * 1e6, t_mm * 1e6))
