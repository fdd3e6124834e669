#!/usr/bin/env python3
"""Time the three fp32 MFMA entry points at rollout (B*N rows) and reward (B*T*N rows) sizes and print achieved
TFLOP/s against the fp32 matrix peak (157.3 TFLOP/s, MI355X_MICROARCH.md).  Flops = the useful multiply-adds of the
reference networks, not the zero-padded MFMA issue count.

    python tools/bench_mlp.py [--rows 40960,4096000] [--agents 10]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PEAK = 157.3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", default="40960,4096000")
    ap.add_argument("--agents", type=int, default=10)
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    import torch as th
    from marl_amd.rl.policy_gradient_rl.dgail.dgail_network import DiffusionDiscriminator, GAILDiscriminator, PolicyNet
    from marl_amd.utils import mlp_device
    dev = th.device("cuda:0")
    N, A = a.agents, a.agents + 1
    SA = 9 + A
    th.manual_seed(0)
    pol = PolicyNet(9 + N, A, 64).to(dev)
    gd = GAILDiscriminator(SA, 64).to(dev)
    dd = DiffusionDiscriminator(SA, 64).to(dev).eval()
    pp, gp, dp = mlp_device.linear_params(pol), mlp_device.linear_params(gd), mlp_device.linear_params(dd.mlp)
    te, ce = dd.time_embed.weight.detach(), dd.cond_embed.weight.detach()
    flops = {
        "policy": 2 * ((9 + N) * 64 + 64 * 64 + 64 * A),
        "gail": 2 * (SA * 64 + 64 * 64 + 64),
        # per condition: forward 148->64->64->SA (embedding part folded: SA*64 + 64*64 + 64*SA) + gradient (64*64 + 64*SA)
        "diffusion": 2 * 2 * (SA * 64 + 64 * 64 + 64 * SA + 64 * 64 + 64 * SA),
    }
    out = []
    for rows in [int(r) for r in a.rows.split(",")]:
        obs = th.rand(rows, 9, device=dev)
        avail = (th.rand(rows, A, device=dev) < 0.7).int()
        avail[:, 0] = 1
        u = th.rand(rows, device=dev)
        act = th.randint(0, A, (rows,), device=dev)
        t1 = th.randint(0, 2, (rows,), device=dev, dtype=th.uint8)
        n1 = th.randn(rows, SA, device=dev)
        n2 = th.randn(rows, SA, device=dev)
        fns = {
            "policy": lambda: mlp_device.policy_forward(pp, obs, N, avail, u, want_probs=False),
            "gail": lambda: mlp_device.gail_reward(gp, obs, act, A, 0.9, None, 0.0),
            "diffusion": lambda: mlp_device.diffusion_reward(dp, te, ce, obs, act, A, t1, n1, t1, n2),
        }
        for name, fn in fns.items():
            for _ in range(3):
                fn()
            th.cuda.synchronize()
            ev = [(th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)) for _ in range(a.reps)]
            for s, e in ev:
                s.record()
                fn()
                e.record()
            th.cuda.synchronize()
            ms = sorted(s.elapsed_time(e) for s, e in ev)[a.reps // 2]
            tf = flops[name] * rows / (ms * 1e-3) / 1e12
            out.append({"kernel": name, "rows": rows, "median_ms": ms, "useful_TFLOPs": tf, "frac_of_fp32_mfma_peak": tf / PEAK,
                        "rows_per_s": rows / (ms * 1e-3)})
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
