"""The learners against the REFERENCE's own PPOAgent / GAILAgent / DGAILAgent (SURVEY 8c fixture 8).

tests/golden/learn.npz holds, for each learner, what tests/golden/make_golden.py recorded by importing the reference:
the initial state_dicts, the rollout buffer, the expert transitions, every random draw the update made (expert sample
indices, the (t, noise) of DiffusionDiscriminator.compute_loss, the minibatch shuffles, the GAIL discriminator's
dropout masks) and the state_dicts after `train()`.  Here the package's agents start from the same parameters and
data, replay the same draws (marl_amd/utils/random_source.py) and must land on the same parameters.

  device "cpu":  the learners' torch-op formulation (host logic; `get_gae` replaced by the reference's loop, since the
                 HIP scan needs a GPU) -- runs in the CPU suite.
  device "cuda": the product path -- fused ec_ppo_* kernels, ec_gae, ec_gail_reward / ec_diffusion_reward, hipGraph
                 minibatch steps -- marked gpu.

Tolerances.  Every optimiser is Adam: a step moves a parameter by ~lr * g / (|g| + eps), i.e. by about lr whatever the
gradient's size, so a parameter's total movement over an update is at most (steps * lr) and an fp32-rounding-sized
change of a gradient only matters where that gradient is itself rounding-sized (the step then is anything in
[-lr, lr]).  The assertions therefore bound the mismatch by a small FRACTION of the movement: at least 99 % of the
elements within 2 % of (steps * lr) and none beyond one full step... see _check.
"""
import argparse

import numpy as np
import pytest
import torch as th

from tests.util import load_golden

DEVICES = [pytest.param("cpu", id="torch-cpu"), pytest.param("cuda", id="hip", marks=pytest.mark.gpu)]


def _torch_gae(args, rewards, v, nv, masks):
    """utils/advantage_utils.py:4-9"""
    adv = th.zeros_like(rewards)
    run = th.zeros_like(rewards[:, 0])
    for t in reversed(range(rewards.shape[1])):
        delta = rewards[:, t] + args.gamma * nv[:, t] * masks[:, t] - v[:, t]
        run = delta + args.gamma * args.lambda_ * run * masks[:, t]
        adv[:, t] = run
    return adv


def _args(kind, cfg, device, extra=()):
    from marl_amd.agent_register import agent_config_register
    from marl_amd.default_config import add_default_args
    from marl_amd.envs.edge_computing.multihop.edge_computing_config import add_edge_computing_env_args
    p = argparse.ArgumentParser(conflict_handler="resolve")
    p = agent_config_register[kind](add_edge_computing_env_args(add_default_args(p)))
    a = p.parse_args(["--task_size_max=6000"] + list(extra))
    a.n_threads, a.episode_limit, a.n_agents = (int(x) for x in cfg)
    a.n_actions, a.obs_shape, a.state_shape = a.n_agents + 1, 9, 9 * a.n_agents
    a.device = device
    a.ec_force_device = device
    return a


def _sd(g, prefix):
    return {k[len(prefix):]: th.from_numpy(g[k]) for k in g.files if k.startswith(prefix)}


def _fill(buffer, g, tag, device):
    for k in ("obs", "avail_actions", "actions", "rewards", "masks", "next_obs"):
        getattr(buffer, k).copy_(th.from_numpy(g[tag + "buf_" + k]).to(device))
    buffer.t = buffer.obs.shape[1]


def _draws(g, tag, drop_kinds=()):
    kinds = [str(k) for k in g[tag + "draw_kinds"]]
    out = []
    for i, k in enumerate(kinds):
        if k not in drop_kinds:
            out.append((k, g[tag + "draw_%04d" % i]))
    return out


def _check(got_module, want_sd, start_sd, lr, steps, what, skip=()):
    """post-update parameters against the reference's: see the module docstring for the shape of the bound"""
    got_sd = got_module.state_dict()
    params = {k for k, _ in got_module.named_parameters()}  # buffers (BatchNorm statistics) are compared by the caller
    worst = 0.0
    for k, want in want_sd.items():
        if k not in params or k in skip:
            continue
        got = got_sd[k].detach().cpu().double()
        diff = (got - want.double()).abs()
        moved = (want.double() - start_sd[k].double()).abs().max().item() if k in start_sd else 0.0
        budget = steps * lr
        frac_ok = (diff <= 0.02 * budget + 1e-7).double().mean().item()
        assert frac_ok >= 0.99, "%s %s: only %.4f of the elements within 2%% of steps*lr (moved up to %.3g)" % (what, k, frac_ok, moved)
        assert diff.max().item() <= 1.05 * lr + 1e-7, "%s %s: max |diff| %.3g exceeds one Adam step %.3g" % (what, k, diff.max().item(), lr)
        worst = max(worst, diff.max().item())
    return worst


def _moved(want_sd, start_sd):
    return max((want_sd[k].double() - start_sd[k].double()).abs().max().item() for k in want_sd
               if want_sd[k].dtype.is_floating_point and k in start_sd)


@pytest.mark.parametrize("device", DEVICES)
def test_ppo_train_matches_reference(device):
    """rl/policy_gradient_rl/ppo/ppo_agent.py:85-134, two consecutive train() calls on one buffer"""
    import marl_amd.rl.policy_gradient_rl.ppo.ppo_agent as pa
    g = load_golden("learn")
    a = _args("ppo", g["ppo_cfg"], device)
    old_gae = pa.get_gae
    if device == "cpu":
        pa.get_gae = _torch_gae
    try:
        agent = pa.PPOAgent(a)
        assert (agent._fg_actor is not None) == (device == "cuda"), "the fused update must be the GPU path"
        a0, c0 = _sd(g, "ppo_actor0_"), _sd(g, "ppo_critic0_")
        agent.actor.load_state_dict(a0)
        agent.critic.load_state_dict(c0)
        _fill(agent.buffer, g, "ppo_", agent.device)
        assert _moved(_sd(g, "ppo_actor1_"), a0) > 2 * a.lr  # the fixture's update is not a no-op
        agent.train()
        _check(agent.actor, _sd(g, "ppo_actor1_"), a0, a.lr, a.epochs, "ppo actor after train 1")
        _check(agent.critic, _sd(g, "ppo_critic1_"), c0, a.lr, a.epochs, "ppo critic after train 1")
        agent.train()
        _check(agent.actor, _sd(g, "ppo_actor2_"), a0, a.lr, 2 * a.epochs, "ppo actor after train 2")
        _check(agent.critic, _sd(g, "ppo_critic2_"), c0, a.lr, 2 * a.epochs, "ppo critic after train 2")
    finally:
        pa.get_gae = old_gae


class _DropoutReplay:
    """F.dropout replaced by the masks the reference drew (the GAIL discriminator is in train mode throughout,
    gail_network.py:59-132)"""

    def __init__(self, masks, device):
        self.masks = [th.from_numpy(m).to(device) for m in masks]
        self.pos = 0

    def __enter__(self):
        import torch.nn.functional as F
        self._orig = F.dropout

        def dropout(x, p=0.5, training=True, inplace=False):
            if not training or p == 0.0:
                return x
            m = self.masks[self.pos]
            self.pos += 1
            assert m.shape == x.shape, (m.shape, x.shape)
            return x * m.to(x.dtype) / (1.0 - p)

        F.dropout = dropout
        # the HIP layer tail (BatchNorm -> ReLU -> Dropout in one node, csrc/ec_bn.hip) takes its keep mask from this hook
        import marl_amd.rl.policy_gradient_rl.gail.gail_network as gn
        self._orig_hook = gn._explicit_mask

        def explicit(x, p):
            m = self.masks[self.pos]
            self.pos += 1
            assert m.shape == x.shape, (m.shape, x.shape)
            return m
        gn._explicit_mask = explicit
        return self

    def __exit__(self, *exc):
        import torch.nn.functional as F
        import marl_amd.rl.policy_gradient_rl.gail.gail_network as gn
        F.dropout = self._orig
        gn._explicit_mask = self._orig_hook


@pytest.mark.parametrize("device", DEVICES)
def test_gail_train_matches_reference(device):
    """rl/policy_gradient_rl/gail/gail_agent.py:128-231: discriminator step (BatchNorm + Dropout), GAIL rewards, PPO
    with the behaviour-cloning term and its weight schedule"""
    import marl_amd.rl.policy_gradient_rl.gail.gail_agent as ga
    from marl_amd.utils.random_source import ReplaySource
    g = load_golden("learn")
    a = _args("gail", g["gail_cfg"], device)
    old_gae = ga.get_gae
    if device == "cpu":
        ga.get_gae = _torch_gae
    try:
        agent = ga.GAILAgent(a)
        assert (agent._fg_actor is not None) == (device == "cuda")
        start = {}
        for name in ("actor", "critic", "discriminator"):
            start[name] = _sd(g, "gail_%s0_" % name)
            getattr(agent, name).load_state_dict(start[name])
        _fill(agent.buffer, g, "gail_", agent.device)
        e_obs, e_act = th.from_numpy(g["gail_expert_obs"]), th.from_numpy(g["gail_expert_actions"])
        for i in range(e_obs.shape[0]):  # one store_expert_data call per expert-driven step (gail_runner.py:59-70)
            agent.store_expert_data(e_obs[i:i + 1].to(agent.device), e_act[i:i + 1].to(agent.device))
        assert len(agent.expert_transitions) == e_obs.shape[0] >= 10
        agent.performance_history = [0.5, 0.6, 0.9, 0.4, 0.6]
        draws = _draws(g, "gail_")
        replay = ReplaySource(agent.device, [d for d in draws if d[0] != "dropout"])
        agent.rng = agent.expert_transitions.rng = replay
        with _DropoutReplay([v for k, v in draws if k == "dropout"], agent.device) as dr:
            agent.train()
        assert replay.exhausted() and dr.pos == 9
        assert abs(agent.bc_loss_weight - float(g["gail_bc_weight1"])) < 1e-12
        assert _moved(_sd(g, "gail_actor1_"), start["actor"]) > 2 * a.lr
        # A bias in front of a BatchNorm in train mode has a gradient of exactly zero in exact arithmetic (the batch
        # mean removes it): what autograd returns is rounding noise, and Adam turns noise of any size into steps of
        # about +-lr.  Those three vectors are noise in the reference too and are not compared.
        noise = ("fc1.bias", "fc2.bias", "fc3.bias")
        for name, steps in (("discriminator", 1), ("critic", a.epochs), ("actor", a.epochs)):
            _check(getattr(agent, name), _sd(g, "gail_%s1_" % name), start[name], a.lr, steps, "gail " + name,
                   skip=noise if name == "discriminator" else ())
        # BatchNorm running statistics travelled too (three forward passes in train mode; the last one sees the
        # updated, partly noise-stepped fc biases: +-lr * momentum on the running means)
        want = _sd(g, "gail_discriminator1_")
        got = agent.discriminator.state_dict()
        for k in ("bn1.running_mean", "bn2.running_var", "bn3.running_mean"):
            assert th.allclose(got[k].cpu(), want[k], rtol=1e-3, atol=1e-4), k
        assert int(got["bn1.num_batches_tracked"]) == int(want["bn1.num_batches_tracked"]) == 3
    finally:
        ga.get_gae = old_gae


@pytest.mark.parametrize("simplified", [True, False], ids=["gail-disc-reward", "diffusion-reward"])
@pytest.mark.parametrize("device", DEVICES)
def test_dgail_train_matches_reference(device, simplified):
    """rl/policy_gradient_rl/dgail/dgail_agent.py:335-381: BC pretrain, diffusion-discriminator step (spectral norm,
    R1 double backward), GAIL-discriminator step, rewards (default: GAIL discriminator; else both compute_loss passes
    with one power iteration each), minibatch PPO with the joint loss"""
    import marl_amd.rl.policy_gradient_rl.dgail.dgail_agent as da
    from marl_amd.utils.random_source import ReplaySource
    g = load_golden("learn")
    tag = "dgail_" if simplified else "dgaild_"
    assert bool(g[tag + "use_simplified_reward"]) == simplified
    a = _args("dgail", g[tag + "cfg"], device, ["--batch_size_run=10"] + ([] if simplified else ["--use_simplified_reward="]))
    assert a.use_simplified_reward is simplified
    old_gae = da.get_gae
    if device == "cpu":
        da.get_gae = _torch_gae
    try:
        agent = da.DGAILAgent(a)
        nets = ("policy", "value", "discriminator", "gail_discriminator")
        start = {}
        for name in nets:
            start[name] = _sd(g, tag + name + "0_")
            getattr(agent, name).load_state_dict(start[name])
        _fill(agent.buffer, g, tag, agent.device)
        e_obs, e_act = th.from_numpy(g[tag + "expert_obs"]), th.from_numpy(g[tag + "expert_actions"])
        for i in range(e_obs.shape[0]):
            agent.store_expert_data(e_obs[i:i + 1].to(agent.device), e_act[i:i + 1].to(agent.device))
        agent._bc_pretrain_steps = 6
        # The DataLoader's RandomSampler calls randperm twice per epoch: the permutation, and an empty tail
        # (`randperm(n)[:num_samples % n]`) when the permutation runs out.  Only the first is a draw of the algorithm.
        draws, seen_perm = [], 0
        for k, v in _draws(g, tag):
            if k == "perm":
                seen_perm += 1
                if seen_perm % 2 == 0:
                    continue
            draws.append((k, v))
        replay = ReplaySource(agent.device, draws)
        agent.rng = agent.expert_transitions.rng = replay
        stats = agent.train(5000)
        assert replay.exhausted(), "%d of %d recorded draws used" % (replay.pos, len(draws))
        want_stats = dict(zip((str(k) for k in g[tag + "stat_keys"]), g[tag + "stat_vals"]))
        for k in ("discriminator_loss", "gail_discriminator_loss", "prob_expert", "prob_agent", "avg_imitation_reward",
                  "entropy", "value_loss"):
            assert abs(stats[k] - want_stats[k]) <= 2e-4 * max(1.0, abs(want_stats[k])), (k, stats[k], want_stats[k])
        assert abs(stats["policy_loss"] - want_stats["policy_loss"]) <= 2e-4, (stats["policy_loss"], want_stats["policy_loss"])
        rows = int(np.prod(g[tag + "cfg"]))
        n_mb = a.ppo_epochs * ((rows + a.ppo_batch_size - 1) // a.ppo_batch_size)
        assert _moved(_sd(g, tag + "policy1_"), start["policy"]) > 2 * 1e-3  # the BC pretrain's Adam(1e-3) steps
        _check(agent.discriminator, _sd(g, tag + "discriminator1_"), start["discriminator"], a.discriminator_lr, 1, "dgail discriminator")
        _check(agent.gail_discriminator, _sd(g, tag + "gail_discriminator1_"), start["gail_discriminator"],
               a.gail_discriminator_lr, 1, "dgail gail_discriminator")
        _check(agent.value, _sd(g, tag + "value1_"), start["value"], a.value_lr, n_mb, "dgail value")
        # policy: 6 pretrain steps at 1e-3, then n_mb steps at policy_lr
        _check(agent.policy, _sd(g, tag + "policy1_"), start["policy"], 1e-3, 6, "dgail policy")
    finally:
        da.get_gae = old_gae
