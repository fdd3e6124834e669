"""The learner's RCCL leg on real GPUs: 2 ranks x B/2 environments == 1 rank x B environments for the fused PPO update
(the assertion of tests/test_host_cpu.py::test_ppo_data_parallel_gloo_world2, on the `nccl` backend and the HIP
kernels).  Needs two visible GPUs: skipped on a one-GPU box (the driver's multi-GPU node runs it)."""
import os
import socket

import numpy as np
import pytest
import torch as th
import torch.multiprocessing as mp

from tests.util import make_args

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _args(B):
    return make_args(n_threads=B, episode_limit=12, n_agents=10, n_actions=11, obs_shape=9, hidden_dim=64, epochs=3,
                     normalize_advantages=True)


def _fill(agent, a, lo, hi, seed=7):
    """Deterministic synthetic rollout for envs [lo, hi) of a global batch of 16."""
    g = th.Generator().manual_seed(seed)
    Bg, T, N, A = 16, a.episode_limit, a.n_agents, a.n_actions
    obs = th.rand(Bg, T + 1, N, 9, generator=g)
    avail = (th.rand(Bg, T, N, A, generator=g) < 0.7).int()
    avail[..., 0] = 1
    actions = th.zeros(Bg, T, N, 1, dtype=th.long)
    actions[avail[..., 1:2] == 1] = 1
    rewards = th.randn(Bg, T, 1, 1, generator=g).expand(Bg, T, N, 1)
    masks = th.ones(Bg, T, N, 1)
    masks[:, -1] = 0
    agent.buffer.reset()
    for t in range(T):
        agent.buffer.insert(obs[lo:hi, t].cuda(), avail[lo:hi, t].cuda(), actions[lo:hi, t].cuda(), rewards[lo:hi, t].cuda(),
                            masks[lo:hi, t].cuda(), obs[lo:hi, t + 1].cuda())


def _train_once(a, lo, hi):
    import marl_amd.rl.policy_gradient_rl.ppo.ppo_agent as pa
    th.manual_seed(123)
    agent = pa.PPOAgent(a)
    assert agent.device.type == "cuda" and agent._fg_actor is not None  # the fused HIP update, not the torch path
    _fill(agent, a, lo, hi)
    agent.train()
    flat = th.cat([p.detach().reshape(-1) for p in list(agent.actor.parameters()) + list(agent.critic.parameters())])
    return flat.cpu(), {k: float(v) for k, v in agent.last_losses.items()}


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    th.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=th.device("cuda:%d" % rank))
    try:
        flat, losses = _train_once(_args(8), 8 * rank, 8 * rank + 8)
        q.put((rank, flat.numpy().copy(), losses))
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(th.cuda.device_count() < 2, reason="needs two GPUs (RCCL over xGMI)")
def test_ppo_data_parallel_nccl_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=240) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # both ranks end with identical parameters (same reduced gradients, same optimiser arithmetic) ...
    assert (got[0][1] == got[1][1]).all()
    # ... equal, to fp32 summation-order noise through Adam, to one rank training on the whole batch
    a = _args(16)
    flat1, losses1 = _train_once(a, 0, 16)
    diff = np.abs(got[0][1] - flat1.numpy())
    budget = a.epochs * a.lr
    assert (diff <= 0.02 * budget + 1e-7).mean() >= 0.99 and diff.max() <= 1.05 * a.lr, (float(diff.max()), budget)
    for k in losses1:
        assert abs(got[0][2][k] - losses1[k]) < 1e-4 * max(1.0, abs(losses1[k]))


# ---------------------------------------------------------------------------------------------- GAIL / DGAIL (product path)
# The assertion of tests/test_dp_imitation_cpu.py on the HIP kernels: 2 ranks x B/2 environments == 1 rank x B environments
# for GAILAgent.train() (whole-job BatchNorm statistics through _bn_batch_stats(sync) / _BnReluDropHipFn) and
# DGAILAgent.train() (fused diffusion / GAIL-discriminator / minibatch policy steps run eagerly under a process group).
# Two forms: `gloo` with both ranks on cuda:0 (runs on the one-GPU box: same product code, host-staged reductions) and
# `nccl` on two GPUs (skipped below two devices).
def _imitation_worker(rank, world, port, q, kind, backend):
    import torch.distributed as dist
    from tests.test_dp_imitation_cpu import train_once
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = rank if backend == "nccl" else 0
    th.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=th.device("cuda:%d" % dev))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        q.put((rank, train_once(kind, rank, world, device="cuda")))
    finally:
        dist.destroy_process_group()


def _imitation_world2(kind, backend):
    from tests.test_dp_imitation_cpu import compare, imitation_args, train_once, BG
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_imitation_worker, args=(r, 2, port, q, kind, backend)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    single = train_once(kind, 0, 1, device="cuda")
    a = imitation_args(kind, BG, "cuda")
    lrs = ({"actor": a.lr, "critic": a.lr, "discriminator": a.lr} if kind == "gail" else
           {"policy": a.policy_lr, "value": a.value_lr, "discriminator": a.discriminator_lr,
            "gail_discriminator": a.gail_discriminator_lr})
    compare(got[0], got[1], single, lrs)


@pytest.mark.parametrize("kind", ["gail", "dgail"])
def test_imitation_data_parallel_two_ranks_on_one_gpu(kind):
    _imitation_world2(kind, "gloo")


@pytest.mark.skipif(th.cuda.device_count() < 2, reason="needs two GPUs (RCCL over xGMI)")
@pytest.mark.parametrize("kind", ["gail", "dgail"])
def test_imitation_data_parallel_nccl_world2(kind):
    _imitation_world2(kind, "nccl")
