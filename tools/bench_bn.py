#!/usr/bin/env python3
"""Achieved HBM rate of the GAIL-discriminator layer-tail kernels (csrc/ec_bn.hip) at [rows, C] fp32.
usage: python tools/bench_bn.py [rows] [C ...]"""
import ctypes as C_
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch as th  # noqa: E402

from marl_amd import _abi  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 4096000
cols = [int(c) for c in sys.argv[2:]] or [128, 64, 32]
L = _abi.lib()
vp = lambda t: C_.c_void_p(t.data_ptr()) if t is not None else C_.c_void_p(0)
stream = C_.c_void_p(th.cuda.current_stream().cuda_stream)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    th.cuda.synchronize()
    a, b = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    th.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3


for Cc in cols:
    x = th.randn(rows, Cc, device="cuda")
    dout = th.randn(rows, Cc, device="cuda")
    out = th.empty_like(x)
    scale, shift = th.rand(Cc, device="cuda") + 0.5, th.randn(Cc, device="cuda")
    a_, b_, c_ = (th.randn(Cc, device="cuda") for _ in range(3))
    n_part = 1024
    partials = th.empty(n_part, 2, Cc, device="cuda")
    gb = rows * Cc * 4 / 1e9
    seed, p = C_.c_uint64(123), C_.c_float(0.3)
    res = {
        "col_stats (1R)": (timed(lambda: L.ec_bn_col_stats(vp(x), rows, Cc, n_part, vp(partials), stream)), 1),
        "fwd (1R+1W)": (timed(lambda: L.ec_bn_relu_dropout_fwd(vp(x), vp(scale), vp(shift), None, seed, p, rows, Cc, vp(out), stream)), 2),
        "bwd_sums (2R)": (timed(lambda: L.ec_bn_relu_dropout_bwd_sums(vp(dout), vp(x), vp(scale), vp(shift), None, seed, p, rows, Cc, n_part, vp(partials), stream)), 2),
        "bwd_dx (2R+1W)": (timed(lambda: L.ec_bn_relu_dropout_bwd_dx(vp(dout), vp(x), vp(scale), vp(shift), None, seed, p, vp(a_), vp(b_), vp(c_), rows, Cc, vp(out), stream)), 3),
        "torch var_mean": (timed(lambda: th.var_mean(x, dim=0, unbiased=False)), 1),
        "torch copy (1R+1W)": (timed(lambda: out.copy_(x)), 2),
    }
    for k, (t, passes) in res.items():
        print("rows %d C %3d  %-20s %8.1f us  %6.2f TB/s" % (rows, Cc, k, t * 1e6, passes * gb / t / 1e3))
