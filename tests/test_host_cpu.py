"""Host-side logic that needs no GPU: registries and flag defaults, the rollout buffer, and the data-parallel form of
the PPO learner on two `gloo` ranks (world_size 2) against one rank holding the whole batch.

The learner's GAE runs on the HIP library in the product; here `get_gae` is replaced by the reference's T-step torch
loop (utils/advantage_utils.py:4-9 restated) so that `PPOAgent.train()` can run on CPU tensors — this file tests the
sharding / all-reduce arithmetic, not the kernel (tests/test_gae_gpu.py does that).
"""
import argparse
import os
import socket

import pytest
import torch as th
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.util import make_args


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _torch_gae(args, rewards, v, nv, masks):
    adv = th.zeros_like(rewards)
    run = th.zeros_like(rewards[:, 0])
    for t in reversed(range(rewards.shape[1])):
        delta = rewards[:, t] + args.gamma * nv[:, t] * masks[:, t] - v[:, t]
        run = delta + args.gamma * args.lambda_ * run * masks[:, t]
        adv[:, t] = run
    return adv


def test_registries_and_defaults():
    from marl_amd.agent_register import agent_config_register, agent_register
    from marl_amd.envs.env_register import env_config_register, env_register
    from marl_amd.runners.runner_register import runner_register
    assert {"edge_computing", "edge_computing_batched"} <= set(env_register) == set(env_config_register)
    assert "parallel" in runner_register and "ppo" in agent_register and "ppo" in agent_config_register
    with pytest.raises(KeyError):
        env_register["no_such_env"]
    a = make_args()
    # envs/edge_computing/multihop/edge_computing_config.py:1-49, ppo_config.py:1-30 (reference defaults)
    assert (a.edge_node_num, a.episode_limit, a.mini_time_slot_num, a.k) == (10, 100, 10, 4)
    assert (a.mini_time_slot, a.deadline) == (0.05, 3.5)
    assert (a.task_seed, a.failure_seed, a.edge_node_seed, a.link_seed) == (100, 200, 500, 601)
    assert (a.gamma, a.lambda_, a.lr, a.epochs, a.eps_clip, a.hidden_dim) == (0.99, 0.95, 0.0003, 4, 0.2, 64)
    assert a.normalize_rewards is True and a.normalize_advantages is False and a.use_entropy is True
    a2 = make_args(["--task_size_max", "6000", "--n_threads", "8"])
    assert a2.task_size_max == 6000 and a2.n_threads == 8


def _small_args(B):
    return make_args(n_threads=B, episode_limit=6, n_agents=3, n_actions=4, obs_shape=9, hidden_dim=64, epochs=2,
                     normalize_advantages=True, ec_force_device="cpu")


def test_episode_buffer_cpu():
    from marl_amd.buffer.episode_buffer import EpisodeBuffer
    a = _small_args(5)
    buf = EpisodeBuffer(a, device="cpu")
    assert not buf.can_sample()
    g = th.Generator().manual_seed(0)
    frames = []
    for t in range(a.episode_limit):
        f = (th.rand(5, 3, 9, generator=g), th.randint(0, 2, (5, 3, 4), generator=g, dtype=th.int),
             th.randint(0, 4, (5, 3, 1), generator=g), th.rand(5, 1, 1, generator=g).expand(5, 3, 1),
             th.ones(5, 3, 1) * (t < a.episode_limit - 1), th.rand(5, 3, 9, generator=g))
        frames.append(f)
        buf.insert(*f)
    assert buf.can_sample()
    out = buf.sample()
    # same field order, shapes and dtypes as buffer/episode_buffer.py:6-12,30-37
    assert [tuple(x.shape) for x in out] == [(5, 6, 3, 9), (5, 6, 3, 4), (5, 6, 3, 1), (5, 6, 3, 1), (5, 6, 3, 1), (5, 6, 3, 9)]
    assert [x.dtype for x in out] == [th.float32, th.int32, th.int64, th.float32, th.float32, th.float32]
    for t, f in enumerate(frames):
        for got, want in zip(out, f):
            assert th.equal(got[:, t], want.to(got.dtype))
    buf.reset()
    assert buf.t == 0 and all(float(x.abs().sum()) == 0 for x in buf.sample())


def _fill(agent, a, lo, hi, seed=7):
    """Deterministic synthetic rollout for envs [lo, hi) of a global batch."""
    g = th.Generator().manual_seed(seed)
    Bg, T, N, A = 8, a.episode_limit, a.n_agents, a.n_actions
    obs = th.rand(Bg, T + 1, N, 9, generator=g)
    avail = (th.rand(Bg, T, N, A, generator=g) < 0.7).int()
    avail[..., 0] = 1
    actions = th.zeros(Bg, T, N, 1, dtype=th.long)  # action 0 is always available
    actions[avail[..., 1:2] == 1] = 1
    rewards = th.randn(Bg, T, 1, 1, generator=g).expand(Bg, T, N, 1)
    masks = th.ones(Bg, T, N, 1)
    masks[:, -1] = 0
    agent.buffer.reset()
    for t in range(T):
        agent.buffer.insert(obs[lo:hi, t], avail[lo:hi, t], actions[lo:hi, t], rewards[lo:hi, t], masks[lo:hi, t],
                            obs[lo:hi, t + 1])


def _train_once(a, lo, hi):
    import marl_amd.rl.policy_gradient_rl.ppo.ppo_agent as pa
    pa.get_gae = _torch_gae
    th.manual_seed(123)
    agent = pa.PPOAgent(a)
    assert agent.device.type == "cpu"
    _fill(agent, a, lo, hi)
    agent.train()
    flat = th.cat([p.detach().reshape(-1) for p in list(agent.actor.parameters()) + list(agent.critic.parameters())])
    return flat, {k: float(v) for k, v in agent.last_losses.items()}


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    th.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from marl_amd.rl.policy_gradient_rl.ppo import ppo_agent as pa
        # scalar helpers against the concatenation
        g = th.Generator().manual_seed(1)
        x = th.randn(2, 50, generator=g) * 3 + 1
        mean, std = pa.global_mean_std(x[rank])
        # flat gradient all-reduce
        lin = th.nn.Linear(3, 2)
        for p in lin.parameters():
            p.grad = th.full_like(p, float(rank + 1))
        pa.all_reduce_grads_(list(lin.parameters()))
        gsum = [float(p.grad.flatten()[0]) for p in lin.parameters()]
        a = _small_args(4)
        flat, losses = _train_once(a, 4 * rank, 4 * rank + 4)
        q.put((rank, float(mean), float(std), float(x.mean()), float(x.std()), gsum, flat, losses))
    finally:
        dist.destroy_process_group()


def test_ppo_data_parallel_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=180) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, mean, std, want_mean, want_std, gsum, _, _ in got:
        assert abs(mean - want_mean) < 1e-6 and abs(std - want_std) < 1e-6   # unbiased std of the whole batch
        assert gsum == [3.0, 3.0]
    # both ranks end with identical parameters ...
    assert th.equal(got[0][6], got[1][6])
    # ... equal (to fp32 summation-order noise) to one rank training on the whole batch
    flat1, losses1 = _train_once(_small_args(8), 0, 8)
    assert th.allclose(got[0][6], flat1, rtol=1e-4, atol=2e-6), float((got[0][6] - flat1).abs().max())
    for k in losses1:
        assert abs(got[0][7][k] - losses1[k]) < 1e-5 * max(1.0, abs(losses1[k]))
