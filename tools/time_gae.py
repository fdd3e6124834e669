#!/usr/bin/env python3
"""ec_gae at the bench size (4096 x 100 x 10: 82 MB of algorithmic traffic): mean launch time over launches queued back
to back through the C ABI (gae_device's own host work, about 20 us a call, would otherwise be what the events see)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as th  # noqa: E402

from marl_amd import _abi  # noqa: E402

B, T, N = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (4096, 100, 10)))
dev = th.device("cuda:0")
r, v, nv = (th.randn(B, T, N, 1, device=dev) for _ in range(3))
m = th.ones(B, T, N, 1, device=dev)
adv = th.empty_like(r)
L = _abi.lib()
ptrs = [C.c_void_p(t.data_ptr()) for t in (r, v, nv, m, adv)]
stream = C.c_void_p(th.cuda.current_stream(dev).cuda_stream)


def launch(k):
    for _ in range(k):
        _abi.check(L.ec_gae(*ptrs, B, T, N, C.c_float(0.99), C.c_float(0.99 * 0.95), stream), "ec_gae")


launch(20)
K = 200
a, b = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
th.cuda.synchronize()
a.record()
launch(K)
b.record()
th.cuda.synchronize()
us = a.elapsed_time(b) * 1e3 / K
print("ec_gae %d x %d x %d: %.1f us per launch (%d back to back) -> %.2f TB/s of 8 (algorithmic 20 B per element)" % (
    B, T, N, us, K, 20.0 * B * T * N / (us * 1e-6) / 1e12))
