#!/usr/bin/env python3
"""ec_linear_fwd against F.linear at the GAIL discriminator's forward and input-gradient shapes; checks the values too."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch as th  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from marl_amd import _abi  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 4096000
L = _abi.lib()
vp = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
stream = C.c_void_p(th.cuda.current_stream().cuda_stream)


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
for k_in, c_out in ((30, 128), (128, 64), (64, 32), (32, 64), (64, 128), (128, 30), (90, 128), (96, 128)):
    x = th.randn(rows, k_in, device="cuda")
    w, b = th.randn(c_out, k_in, device="cuda") * 0.2, th.randn(c_out, device="cuda")
    y = th.empty(rows, c_out, device="cuda")
    t0 = timed(lambda: _abi.check(L.ec_linear_fwd(vp(x), vp(w), vp(b), rows, k_in, c_out, vp(y), stream), "ec_linear_fwd"))
    want = F.linear(x, w, b)
    err = float((y - want).abs().max())
    t2 = timed(lambda: F.linear(x, w, b))
    print("rows %d  %3d -> %3d  ec_linear_fwd %7.1f us (%.2f TB/s)   F.linear %7.1f us   max |diff| %.2e" % (
        rows, k_in, c_out, t0 * 1e6, rows * (k_in + c_out) * 4 / t0 / 1e12, t2 * 1e6, err))

# the tail folded into the next layer (ec_linear_fwd_bn) against ec_bn_relu_dropout_fwd followed by the faster of the two layers
for k_in, c_out, p in ((128, 64, 0.3), (64, 32, 0.3), (32, 1, 0.2)):
    x = th.randn(rows, k_in, device="cuda")
    sc, sh = th.rand(k_in, device="cuda") + 0.5, th.randn(k_in, device="cuda") * 0.3
    w, b = th.randn(c_out, k_in, device="cuda") * 0.2, th.randn(c_out, device="cuda")
    y, act = th.empty(rows, c_out, device="cuda"), th.empty(rows, k_in, device="cuda")
    t0 = timed(lambda: _abi.check(L.ec_linear_fwd_bn(vp(x), vp(sc), vp(sh), C.c_uint64(7), C.c_float(p), vp(w), vp(b), rows, k_in,
                                                     c_out, vp(y), 0, vp(None), stream), "ec_linear_fwd_bn"))
    part = th.empty(1024, 2, c_out, device="cuda")
    t0s = timed(lambda: _abi.check(L.ec_linear_fwd_bn(vp(x), vp(sc), vp(sh), C.c_uint64(7), C.c_float(p), vp(w), vp(b), rows, k_in,
                                                      c_out, vp(y), 1024, vp(part), stream), "ec_linear_fwd_bn"))
    t1 = timed(lambda: _abi.check(L.ec_bn_relu_dropout_fwd(vp(x), vp(sc), vp(sh), vp(None), C.c_uint64(7), C.c_float(p), rows, k_in,
                                                           vp(act), stream), "ec_bn_relu_dropout_fwd"))
    t2 = timed(lambda: _abi.check(L.ec_linear_fwd(vp(act), vp(w), vp(b), rows, k_in, c_out, vp(y), stream), "ec_linear_fwd"))
    t3 = timed(lambda: F.linear(act, w, b))
    print("rows %d  %3d -> %3d  with the column sums of y: %7.1f us" % (rows, k_in, c_out, t0s * 1e6))
    print("rows %d  %3d -> %3d  ec_linear_fwd_bn %7.1f us (%.2f TB/s)   tail %7.1f us + layer %7.1f us (F.linear %7.1f us)" % (
        rows, k_in, c_out, t0 * 1e6, rows * (k_in + c_out) * 4 / t0 / 1e12, t1 * 1e6, t2 * 1e6, t3 * 1e6))
