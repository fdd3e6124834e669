"""Data-parallel equivalence of the imitation learners (BASELINE configs[3] / [4] run them on 8 GPUs):
2 ranks x B/2 environments make the same GAILAgent.train() / DGAILAgent.train() update as 1 rank x B environments.

world_size-2 `gloo` on the CPU, on the learners' torch formulation (ec_force_device=cpu), with every random draw of
the update replayed so that the two runs see the same numbers: expert samples from one seeded source (rank 0's are
broadcast to the others), row-wise draws (dropout masks, expert rows, diffusion (t, noise), minibatch permutations) drawn
for the WHOLE batch and cut into the ranks' shards.  What makes the equivalence hold is in the product: whole-job
BatchNorm statistics in the GAIL discriminator (gail_network._SyncBatchNormFn / _bn_batch_stats(sync=True)), global
normalisation statistics, gradient averaging, shared expert samples.
Reference: rl/policy_gradient_rl/gail/gail_agent.py:128-231, gail_network.py:97-125, dgail/dgail_agent.py:203-381.
tests/test_dist_gpu.py holds the `nccl` twins (skipped below two GPUs)."""
import os
import socket

import numpy as np
import pytest
import torch as th
import torch.distributed as dist
import torch.multiprocessing as mp

from marl_amd.utils.random_source import RandomSource

BG, T, N, A = 8, 6, 3, 4          # envs of the whole job, episode length, agents, actions


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def torch_gae(args, rewards, v, nv, masks):
    adv = th.zeros_like(rewards)
    last = th.zeros_like(rewards[:, 0])
    for t in reversed(range(rewards.shape[1])):
        delta = rewards[:, t] + args.gamma * nv[:, t] * masks[:, t] - v[:, t]
        last = delta + args.gamma * args.lambda_ * last * masks[:, t]
        adv[:, t] = last
    return adv


def imitation_args(kind, B, device="cpu"):
    import argparse
    from marl_amd.agent_register import agent_config_register
    from marl_amd.default_config import add_default_args
    from marl_amd.envs.edge_computing.multihop.edge_computing_config import add_edge_computing_env_args
    p = argparse.ArgumentParser(conflict_handler="resolve")
    p = agent_config_register[kind](add_edge_computing_env_args(add_default_args(p)))
    a = p.parse_args([])
    a.n_threads, a.episode_limit, a.n_agents, a.n_actions, a.obs_shape = B, T, N, A, 9
    a.state_shape, a.device = 9 * N, device
    if device == "cpu":
        a.ec_force_device = "cpu"
    a.ppo_batch_size = 6        # DGAIL: minibatches of 6 * n_threads rows -> three per epoch at any shard size
    a.batch_size_run = 20
    a.batch_expert_transitions = 5
    return a


def fill(agent, lo, hi, device, n_expert=40):
    """rows [lo, hi) of one synthetic global rollout; the same expert store on every rank"""
    g = th.Generator().manual_seed(11)
    obs = th.rand(BG, T + 1, N, 9, generator=g)
    act = th.randint(0, A, (BG, T, N, 1), generator=g)
    rew = th.randn(BG, T, 1, 1, generator=g).expand(BG, T, N, 1)
    avail = th.ones(BG, T, N, A, dtype=th.int)
    agent.buffer.reset()
    for t in range(T):
        m = th.ones(hi - lo, N, 1) * (t < T - 1)
        agent.buffer.insert(obs[lo:hi, t].to(device), avail[lo:hi, t].to(device), act[lo:hi, t].to(device),
                            rew[lo:hi, t].to(device), m.to(device), obs[lo:hi, t + 1].to(device))
    e_obs = th.rand(n_expert, N, 9, generator=g)
    e_act = th.randint(0, A, (n_expert, N), generator=g)
    agent.store_expert_data(e_obs.to(device), e_act.to(device))


class ShardedSource(RandomSource):
    """The learner's random source for rank `rank` of `world` (world = 1: the single-process run of the same job): every
    draw comes from one seeded CPU generator in call order; draws that are per ROW of the batch are made for the whole
    job's rows and cut to the shard, so that the ranks' shards together see what the single process sees."""

    def __init__(self, device, rank, world, shards, mb_local):
        super().__init__(device)
        self.g = th.Generator().manual_seed(77)
        self.rank, self.world, self.shards, self.mb_local = rank, world, shards, mb_local

    def _cut(self, full, n_local):
        if self.world == 1:
            return full
        return full[self.rank * n_local:(self.rank + 1) * n_local]

    def _local(self, n):
        """rows of one shard, from the row count a call is made with"""
        return n if self.world > 1 else n // self.shards

    def choice_without_replacement(self, size, n):
        return th.randperm(size, generator=self.g)[:n].to(self.device)

    def choice_without_replacement_many(self, size, n, count):
        return th.stack([self.choice_without_replacement(size, n) for _ in range(count)])

    def choice_with_replacement(self, size, n):
        nl = self._local(n)
        return self._cut(th.randint(0, size, (nl * self.shards,), generator=self.g), nl).to(self.device)

    def permutation(self, n):
        nl = self._local(n)
        perms = [th.randperm(nl, generator=self.g) for _ in range(self.shards)]
        if self.world > 1:
            return perms[self.rank].to(self.device)
        # one process: minibatch i = the shards' minibatches i side by side (shard s holds rows [s * nl, (s + 1) * nl))
        out = []
        for i in range(0, nl, self.mb_local):
            out += [perms[s][i:i + self.mb_local] + s * nl for s in range(self.shards)]
        return th.cat(out).to(self.device)

    def diffusion_draws(self, rows, dim):
        nl = self._local(rows)
        t = th.randint(0, 2, (nl * self.shards,), generator=self.g)
        noise = th.randn(nl * self.shards, dim, generator=self.g)
        return self._cut(t, nl).to(self.device), self._cut(noise, nl).to(self.device)


class ShardedDropout:
    """F.dropout with keep masks drawn for the whole job's agent rows and cut to the shard (inputs of any other row count
    -- the 64 expert transitions -- get the same mask on every rank); also serves gail_network._explicit_mask (HIP tails)."""

    def __init__(self, rank, world, shards, rows_local):
        self.rank, self.world, self.shards, self.rows_local, self.calls = rank, world, shards, rows_local, 0

    def mask(self, x, p):
        self.calls += 1
        g = th.Generator().manual_seed(1000 + self.calls)
        if x.shape[0] == self.rows_local * (self.shards if self.world == 1 else 1):
            full = th.rand(self.rows_local * self.shards, x.shape[1], generator=g) >= p
            m = full if self.world == 1 else full[self.rank * self.rows_local:(self.rank + 1) * self.rows_local]
        else:
            m = th.rand(x.shape[0], x.shape[1], generator=g) >= p
        return m.to(x.device)

    def __enter__(self):
        import torch.nn.functional as F
        import marl_amd.rl.policy_gradient_rl.gail.gail_network as gn
        self._orig, self._orig_hook = F.dropout, gn._explicit_mask

        def dropout(x, p=0.5, training=True, inplace=False):
            if not training or p == 0.0:
                return x
            return x * self.mask(x, p).to(x.dtype) / (1.0 - p)
        F.dropout = dropout
        gn._explicit_mask = lambda x, p: self.mask(x, p)
        return self

    def __exit__(self, *exc):
        import torch.nn.functional as F
        import marl_amd.rl.policy_gradient_rl.gail.gail_network as gn
        F.dropout, gn._explicit_mask = self._orig, self._orig_hook


def train_once(kind, rank, world, device="cpu", shards=2):
    """One train() of the learner `kind` on shard `rank` of `world` (world = 1: all BG envs in one process).  Returns
    {module name: flat parameters}."""
    B = BG // shards if world > 1 else BG
    lo = rank * B if world > 1 else 0
    a = imitation_args(kind, B, device)
    rows_local = (BG // shards) * T * N
    th.manual_seed(123)
    if kind == "gail":
        import marl_amd.rl.policy_gradient_rl.gail.gail_agent as mod
        agent = mod.GAILAgent(a)
        nets = ("actor", "critic", "discriminator")
    else:
        import marl_amd.rl.policy_gradient_rl.dgail.dgail_agent as mod
        agent = mod.DGAILAgent(a)
        agent._bc_pretrain_steps = 3
        nets = ("policy", "value", "discriminator", "gail_discriminator")
    if device == "cpu":
        mod.get_gae = torch_gae
    else:   # the product path: fused HIP updates, not the torch formulation
        assert agent.device.type == "cuda" and (agent._fg_actor if kind == "gail" else agent._fg_policy) is not None
    fill(agent, lo, lo + B, agent.device)
    agent.rng = agent.expert_transitions.rng = ShardedSource(agent.device, rank, world, shards,
                                                             a.ppo_batch_size * (BG // shards))
    with ShardedDropout(rank, world, shards, rows_local):
        if kind == "gail":
            agent.train()
        else:
            assert agent.train(1000) is not None
    out = {}
    for name in nets:
        out[name] = {k: v.detach().cpu().numpy().copy() for k, v in getattr(agent, name).state_dict().items()}
    return out


def _worker(rank, world, port, q, kind):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    th.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        q.put((rank, train_once(kind, rank, world)))
    finally:
        dist.destroy_process_group()


def compare(got0, got1, single, lrs):
    """both ranks identical; equal to the single process to fp32 summation-order noise.  Every optimiser is Adam, whose
    step is ~lr whatever the gradient's size: the bound is a small fraction of one step."""
    # pre-BatchNorm biases: their exact gradient is zero, what Adam steps on is rounding noise (see test_learners_ref)
    noise = ("fc1.bias", "fc2.bias", "fc3.bias")
    for net, sd in single.items():
        for k, want in sd.items():
            assert np.array_equal(got0[net][k], got1[net][k]), (net, k)
            if net == "discriminator" and k in noise and "bn1.weight" in sd:
                continue
            if "num_batches_tracked" in k:
                assert got0[net][k] == want, (net, k)
                continue
            err = np.abs(got0[net][k] - want).max()
            # (running means: the reward pass runs after the step, on the noise-stepped biases: +-lr * momentum)
            tol = 0.25 * lrs[net] if "running_mean" in k else 0.02 * lrs[net]
            assert err <= tol + 1e-5 * np.abs(want).max(), (net, k, err)


def _run_world2(kind):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, kind)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return got


def test_gail_data_parallel_gloo_world2():
    got = _run_world2("gail")
    single = train_once("gail", 0, 1)
    a = imitation_args("gail", BG)
    compare(got[0], got[1], single, {"actor": a.lr, "critic": a.lr, "discriminator": a.lr})
    # the update moved things, and per-rank statistics would not have passed: the discriminator's running means are the
    # whole batch's
    assert np.abs(single["discriminator"]["bn1.running_mean"]).max() > 0


def test_dgail_data_parallel_gloo_world2():
    got = _run_world2("dgail")
    single = train_once("dgail", 0, 1)
    a = imitation_args("dgail", BG)
    compare(got[0], got[1], single, {"policy": a.policy_lr, "value": a.value_lr, "discriminator": a.discriminator_lr,
                                     "gail_discriminator": a.gail_discriminator_lr})


def test_sync_batchnorm_function_matches_whole_batch_autograd():
    """_SyncBatchNormFn with no process group degenerates to the whole-batch BatchNorm: values and gradients against
    torch's own batch_norm in float64 (the two-rank case is what the gloo tests above run)."""
    from marl_amd.rl.policy_gradient_rl.gail.gail_network import _SyncBatchNormFn
    g = th.Generator().manual_seed(0)
    x = (th.randn(50, 8, generator=g) * 3 + 1).double().requires_grad_(True)
    w = th.rand(8, generator=g).double().requires_grad_(True)
    b = th.rand(8, generator=g).double().requires_grad_(True)
    dy = th.randn(50, 8, generator=g).double()
    y, mean, var = _SyncBatchNormFn.apply(x, w, b, 1e-5)
    y.backward(dy)
    got = (y.detach(), x.grad.clone(), w.grad.clone(), b.grad.clone())
    x.grad = w.grad = b.grad = None
    y2 = th.nn.functional.batch_norm(x, None, None, w, b, training=True, eps=1e-5)
    y2.backward(dy)
    for a_, b_ in zip(got, (y2.detach(), x.grad, w.grad, b.grad)):
        assert th.allclose(a_, b_, rtol=1e-9, atol=1e-10)
    assert th.allclose(mean, x.detach().mean(0)) and th.allclose(var, x.detach().var(0, unbiased=False))
