"""ParallelRunner + PPOAgent on the batched HIP environment.

  * the runner's episode_info against the same rollout stepped on the CPU oracle with the same (deterministic,
    state-dependent) policy: counters exact, float aggregates to fp32 rounding of the per-step outputs;
  * two PPO iterations end to end (rollout -> device buffer -> reward normalisation -> GAE kernel -> 4 epochs):
    buffer contents consistent with the env's outputs, parameters move, losses finite, masked actions never taken.
"""
import numpy as np
import pytest
import torch as th

from tests.util import args_from_golden, load_golden

pytestmark = pytest.mark.gpu


class _FirstNeighbourAgent:
    """Stub with the agent interface: forwards to the lowest-numbered available action (local execution only when no
    neighbour is available) - keeps multi-hop traffic high."""

    class _Buf:
        def reset(self):
            pass

        def insert(self, *a):
            pass

    def __init__(self):
        self.buffer = self._Buf()
        self.trained = 0

    @staticmethod
    def pick(avail):
        return avail.to(th.int64).argmax(dim=-1)

    def select_actions(self, obs, avail, test_mode=False):
        return self.pick(avail)

    def train(self):
        self.trained += 1


def _args(B, **over):
    g = load_golden("env10")
    a = args_from_golden(g)
    a.n_threads = B
    a.algo_type = "rl"
    a.obs_shape = 9
    a.n_agents = a.edge_node_num
    a.n_actions = a.edge_node_num + 1
    for k, v in over.items():
        setattr(a, k, v)
    return a


def test_runner_episode_info_matches_oracle(oracle_lib):
    from marl_amd import _abi
    from marl_amd.runners.runner_register import runner_register
    from oracle.ec_oracle import OracleEnv
    B, base = 96, 11
    args = _args(B, ec_env_id_base=base, ec_hop_hist_len=32)
    agent = _FirstNeighbourAgent()
    runner = runner_register["parallel"](args, agent)
    infos = [runner.run(test_mode=False), runner.run(test_mode=True)]
    assert agent.trained == 1 and runner.t_env == B * args.episode_limit

    orc = OracleEnv(_abi.config_from_args(args, hop_hist_len=32), runner.env.tables, B, env_id_base=base)
    for ep in range(2):
        orc.reset()
        ret = ctime = 0.0
        cnt = np.zeros(4, dtype=np.int64)
        for t in range(args.episode_limit):
            _, avail = orc.observe()
            out = orc.step(avail.argmax(-1))
            assert out["bad"] == 0
            ret += float(out["reward"].astype(np.float32).astype(np.float64).sum())
            ctime += float(out["completion_time"].astype(np.float32).astype(np.float64).sum())
            cnt += out["counters"].sum(0)
        fin, succ, drop, fail = [int(x) for x in cnt]
        info = infos[ep]
        assert fin > 0
        assert info["success_rate"] == succ / fin and info["drop_rate"] == drop / fin and info["failure_rate"] == fail / fin
        assert abs(info["episode_return"] - ret / B) <= 1e-9 * max(1.0, abs(ret / B))
        assert abs(info["task_completion_time"] - ctime / (succ + drop)) <= 1e-9 * (ctime / (succ + drop))
        hops = out["hop_hist"].sum(0)  # cumulative over episodes, as edge_node.py:82
        assert info["max_hop_dict"] == {h: int(c) for h, c in enumerate(hops) if c}
    runner.close_env()


def test_ppo_two_iterations():
    from marl_amd.agent_register import agent_register
    from marl_amd.runners.runner_register import runner_register
    B = 64
    args = _args(B, episode_limit=40)
    th.manual_seed(100)
    agent = agent_register["ppo"](args)
    assert agent.device.type == "cuda" and agent._fused_policy_ok()
    runner = runner_register["parallel"](args, agent)
    before = th.cat([p.detach().reshape(-1).clone() for p in agent.actor.parameters()])
    info = runner.run(test_mode=False)
    buf = agent.buffer
    obs, avail, actions, rewards, masks, next_obs = buf.sample()
    assert buf.can_sample() and runner.t_env == B * args.episode_limit
    # transitions chain: next_obs[t] is obs[t+1]; the last mask is 0, all others 1 (parallel_episode_runner.py:102)
    assert th.equal(next_obs[:, :-1], obs[:, 1:])
    assert float(masks[:, -1].sum()) == 0 and float(masks[:, :-1].min()) == 1
    # team reward broadcast over agents (:101)
    assert th.equal(rewards, rewards[:, :, :1].expand_as(rewards))
    # sampled actions respect the availability mask
    assert bool((avail.gather(-1, actions) == 1).all())
    after = th.cat([p.detach().reshape(-1) for p in agent.actor.parameters()])
    assert not th.equal(before, after)
    assert all(bool(th.isfinite(v)) for v in agent.last_losses.values())
    for k in ("episode_return", "success_rate", "drop_rate", "failure_rate", "task_completion_time"):
        assert np.isfinite(info[k])
    assert abs(info["success_rate"] + info["drop_rate"] + info["failure_rate"] - 1.0) < 1e-12
    info2 = runner.run(test_mode=False)
    assert runner.t_env == 2 * B * args.episode_limit and np.isfinite(info2["episode_return"])
    runner.close_env()


def test_fused_select_actions_matches_torch_distribution():
    """select_actions (HIP kernel, inverse CDF) draws from the same masked softmax as the reference's
    Categorical path: compare empirical frequencies on a fixed state."""
    from marl_amd.agent_register import agent_register
    args = _args(4096)
    th.manual_seed(5)
    agent = agent_register["ppo"](args)
    with th.no_grad():
        agent.actor.fc3.weight.mul_(30.0)
    N, A = args.n_agents, args.n_actions
    obs1 = th.rand(1, N, 9, device=agent.device)
    avail1 = (th.rand(1, N, A, device=agent.device) < 0.6).int()
    avail1[..., 0] = 1
    obs, avail = obs1.expand(4096, N, 9).contiguous(), avail1.expand(4096, N, A).contiguous()
    acts = th.stack([agent.select_actions(obs, avail) for _ in range(8)])  # 32768 draws per agent
    with th.no_grad():
        logits = agent.actor(agent._with_ids(obs1))
        logits[avail1 == 0] = -1e10
        probs = th.softmax(logits, -1)[0]
    freq = th.stack([(acts[..., n].reshape(-1, 1) == th.arange(A, device=acts.device)).float().mean(0) for n in range(N)])
    assert float((freq - probs).abs().max()) < 0.012   # ~4 sigma at p=0.5, 32768 draws
    assert float(freq[avail1[0] == 0].sum()) == 0.0


def _imitation_args(kind, B, **over):
    import argparse
    from marl_amd.agent_register import agent_config_register
    from marl_amd.default_config import add_default_args
    from marl_amd.envs.edge_computing.multihop.edge_computing_config import add_edge_computing_env_args
    p = argparse.ArgumentParser(conflict_handler="resolve")
    p = agent_config_register[kind](add_edge_computing_env_args(add_default_args(p)))
    a = p.parse_args(["--task_size_max=6000"])
    a.n_threads, a.obs_shape, a.n_agents, a.n_actions = B, 9, a.edge_node_num, a.edge_node_num + 1
    a.state_shape, a.device = 9 * a.edge_node_num, "cuda"
    for k, v in over.items():
        setattr(a, k, v)
    return a


def test_dgail_runner_iterations():
    """DGAILRunner + DGAILAgent end to end on the device: expert-guided rollout (ec_expert_policy), expert store,
    both discriminator updates, rewards through ec_gail_reward / ec_diffusion_reward, minibatch PPO."""
    from marl_amd.agent_register import agent_register
    from marl_amd.runners.runner_register import runner_register
    B = 32
    a = _imitation_args("dgail", B, episode_limit=50)
    th.manual_seed(1)
    agent = agent_register["dgail"](a)
    agent._bc_pretrain_steps = 20
    runner = runner_register["dgail_runner"](a, agent)
    assert runner._expert_ratio() == 1.0
    info = runner.run(test_mode=False)
    T = a.episode_limit
    assert runner.t_env == B * T and info["episode_steps"] == T
    # t_env < bc_guidance_steps (2000) during the whole first episode: every step is expert-driven (dgail_runner.py:31-32)
    assert runner.expert_steps == B * T
    assert info["expert_buffer_size"] == min(runner.expert_steps, a.max_expert_buffer_size)
    assert {"discriminator_loss", "gail_discriminator_loss", "policy_loss", "value_loss", "entropy",
            "avg_imitation_reward"} <= set(info)
    assert all(np.isfinite(info[k]) for k in ("discriminator_loss", "policy_loss", "avg_imitation_reward", "episode_return"))
    obs, avail, actions, rewards, masks, next_obs = agent.buffer.sample()
    assert bool((avail.gather(-1, actions) == 1).all())          # the buffer holds the AGENT's (masked) actions
    # the expert store holds states in which the expert acted, with the expert's decision for them
    n = len(agent.expert_transitions)
    eo, ea = agent.expert_transitions.obs[:n], agent.expert_transitions.actions[:n]
    no_task = eo[..., 5] < 0
    assert bool((ea[no_task] == a.n_agents).all()) and bool((ea[~no_task] < a.n_agents).all())
    # rewards: HIP entry points against the torch formulas, simplified (GAIL discriminator) and diffusion branch
    env_r = rewards
    got = agent._compute_rewards(obs, actions, env_r, runner.t_env)
    want = agent._compute_rewards_torch(obs, actions, env_r, False, agent._imitation_weight(runner.t_env))
    assert th.allclose(got, want, rtol=2e-4, atol=2e-5)
    agent._use_simplified_reward = False
    r_diff = agent._compute_rewards(obs, actions, env_r, runner.t_env)
    assert r_diff.shape == got.shape and bool(th.isfinite(r_diff).all()) and float(r_diff.std()) > 0
    agent._use_simplified_reward = True
    info2 = runner.run(test_mode=False)
    assert runner.t_env == 2 * B * T and agent.update_count == 2
    info3 = runner.run(test_mode=True)
    assert runner.t_env == 2 * B * T and "policy_loss" not in info3 and np.isfinite(info3["success_rate"])
    runner.close_env()


@pytest.mark.parametrize("kind", ["dgail", "gail"])
def test_imitation_runner_graph_rollout(kind):
    """--ec_graph_rollout for the expert-guided runners: eager first episode, recorded second, replayed afterwards;
    every episode fills the buffer with masked agent actions, counts env steps and feeds the expert store."""
    from marl_amd.agent_register import agent_register
    from marl_amd.runners.runner_register import runner_register
    B = 24
    a = _imitation_args(kind, B, episode_limit=30, ec_graph_rollout=1, ec_kernel_rollout=0)  # (the kernel rollout has nothing to replay)
    th.manual_seed(4)
    agent = agent_register[kind](a)
    if kind == "dgail":
        agent._bc_pretrain_steps = 5
    runner = runner_register[kind + "_runner"](a, agent)
    returns = []
    for ep in range(4):
        before = int(agent.expert_transitions._written.item())
        info = runner.run(test_mode=False)
        assert runner.t_env == (ep + 1) * B * a.episode_limit
        assert agent.buffer.t == a.episode_limit
        obs, avail, actions, rewards, masks, next_obs = agent.buffer.sample()
        assert bool((avail.gather(-1, actions) == 1).all()), ep
        assert th.equal(next_obs[:, :-1], obs[:, 1:])
        assert int(agent.expert_transitions._written.item()) > before          # guidance ratio is > 0 this early
        assert np.isfinite(info["episode_return"]) and 0 < info["success_rate"] <= 1
        returns.append(info["episode_return"])
    assert any(g is not None for g in runner._graphs.values())                 # episodes 3 and 4 were replays
    assert len(set(returns)) == 4                                             # fresh random numbers in every replay
    info = runner.run(test_mode=True)
    assert runner.t_env == 4 * B * a.episode_limit and np.isfinite(info["success_rate"])
    runner.close_env()


def test_dgail_graphed_minibatch_step_equals_eager():
    """The hipGraph of DGAIL's minibatch step (policy + value forward, joint loss, backward, two clipped Adam steps)
    leaves the networks where the same step run eagerly leaves them, and capturing it does not move them."""
    import copy
    from marl_amd.agent_register import agent_register
    B = 16
    a = _imitation_args("dgail", B, episode_limit=20)
    th.manual_seed(3)
    g_agent = agent_register["dgail"](a)
    a2 = copy.copy(a)
    a2.ec_graph_update = 0
    e_agent = agent_register["dgail"](a2)
    assert g_agent._graph_update and not e_agent._graph_update
    for m in ("policy", "value"):
        getattr(e_agent, m).load_state_dict(getattr(g_agent, m).state_dict())
    dev = g_agent.device
    N, O = a.n_agents, a.obs_shape
    rows, mb = B * 20 * N, 64 * B
    for ag in (g_agent, e_agent):
        ag.expert_transitions.append(th.rand(300, N, O, device=dev), th.randint(0, N + 1, (300, N), device=dev))
    e_agent.expert_transitions.obs.copy_(g_agent.expert_transitions.obs)
    e_agent.expert_transitions.actions.copy_(g_agent.expert_transitions.actions)
    gen = th.Generator(device=dev).manual_seed(5)
    pin = th.rand(rows, g_agent.policy_input_dim, device=dev, generator=gen)
    act = th.randint(0, N + 1, (rows,), device=dev, generator=gen)
    logp = -th.rand(rows, device=dev, generator=gen) * 3
    adv = th.randn(rows, device=dev, generator=gen)
    ret = th.randn(rows, 1, device=dev, generator=gen)
    bc_w = th.full((), 0.3, device=dev)
    before = copy.deepcopy(g_agent.policy.state_dict())
    g = g_agent._minibatch_graph(pin, act, logp, adv, ret, mb, bc_w)
    for k, v in g_agent.policy.state_dict().items():
        assert th.equal(v, before[k]), k                       # warm-up and capture left the networks untouched
    totals_e = th.zeros(3, device=dev)
    for step in range(3):
        idx = th.randperm(rows, device=dev, generator=gen)[:mb]
        bc = g_agent.expert_transitions.sample_transitions(a.batch_expert_transitions)
        g["idx"].copy_(idx)
        g["bc_obs"].copy_(bc[0])
        g["bc_act"].copy_(bc[1])
        g["graph"].replay()
        e_agent._minibatch_step(pin, act, logp, adv, ret, idx, bc, bc_w, totals_e)
    assert th.allclose(g["totals"], totals_e, rtol=1e-4, atol=1e-6)
    for m in ("policy", "value"):
        for (k, p), (_, q) in zip(getattr(g_agent, m).state_dict().items(), getattr(e_agent, m).state_dict().items()):
            assert th.allclose(p, q, rtol=1e-4, atol=1e-6), (m, k)
    moved = sum(float((v - before[k]).abs().max()) for k, v in g_agent.policy.state_dict().items())
    assert moved > 0


def test_gail_runner_iterations():
    from marl_amd.agent_register import agent_register
    from marl_amd.runners.runner_register import runner_register
    B = 32
    a = _imitation_args("gail", B, episode_limit=40)
    th.manual_seed(2)
    agent = agent_register["gail"](a)
    runner = runner_register["gail_runner"](a, agent)
    info = runner.run(test_mode=False, t_expert=0)
    assert runner.t_env == B * a.episode_limit and agent.update_count == 1
    assert len(agent.expert_transitions) == min(B * a.episode_limit, a.max_expert_buffer_size)   # ratio 1.0 at t_expert 0
    assert "discriminator_loss" in agent.last_losses and all(bool(th.isfinite(v)) for v in agent.last_losses.values())
    assert abs(info["success_rate"] + info["drop_rate"] + info["failure_rate"] - 1.0) < 1e-12
    # past the guidance window nothing is expert driven and the store stops growing
    stored = int(agent.expert_transitions._written.item())
    runner.run(test_mode=False, t_expert=a.expert_guidance_steps)
    assert int(agent.expert_transitions._written.item()) == stored and agent.update_count == 2
    runner.run(test_mode=True)
    runner.close_env()


def test_ppo_fused_update_matches_torch_update():
    """One PPOAgent.train() on the same rollout with the fused HIP update and with the torch-autograd update
    (args.ec_torch_update): same losses and the same parameters after 4 epochs of Adam, to fp32 tolerance."""
    from marl_amd.agent_register import agent_register
    from marl_amd.runners.runner_register import runner_register
    B = 48
    args = _args(B, episode_limit=30, normalize_advantages=True)
    th.manual_seed(9)
    fused = agent_register["ppo"](args)
    runner = runner_register["parallel"](args, fused)
    fused.train = lambda: None          # roll out only
    runner.run(test_mode=False)
    del fused.train
    args_t = _args(B, episode_limit=30, normalize_advantages=True, ec_torch_update=True)
    plain = agent_register["ppo"](args_t)
    assert fused._fg_actor is not None and plain._fg_actor is None
    plain.actor.load_state_dict(fused.actor.state_dict())
    plain.critic.load_state_dict(fused.critic.state_dict())
    for dst, src in zip(plain.buffer.sample(), fused.buffer.sample()):
        dst.copy_(src)
    fused.train()
    plain.train()
    for k in ("critic_loss", "actor_loss"):
        a_, b_ = float(fused.last_losses[k]), float(plain.last_losses[k])
        assert abs(a_ - b_) <= 1e-4 * max(1.0, abs(b_)), (k, a_, b_)
    for (n1, p1), (_, p2) in zip(list(fused.actor.named_parameters()) + list(fused.critic.named_parameters()),
                                 list(plain.actor.named_parameters()) + list(plain.critic.named_parameters())):
        # Adam's first steps move every weight by ~lr regardless of gradient scale: compare against the step size
        assert float((p1 - p2).abs().max()) <= 0.05 * 4 * args.lr + 1e-6, (n1, float((p1 - p2).abs().max()))
    runner.close_env()


def test_graph_rollout_equals_eager_rollout():
    """args.ec_graph_rollout: the captured-episode replay produces the same trajectories, buffer and metrics as the
    eager loop (deterministic stub policy; env seeds identical)."""
    from marl_amd.buffer.episode_buffer import EpisodeBuffer
    from marl_amd.runners.runner_register import runner_register

    class Agent(_FirstNeighbourAgent):
        def __init__(self, args):
            super().__init__()
            self.buffer = EpisodeBuffer(args, device="cuda")

    infos, bufs = [], []
    for graph in (False, True):
        args = _args(40, ec_env_id_base=5, episode_limit=30, ec_graph_rollout=graph)
        agent = Agent(args)
        runner = runner_register["parallel"](args, agent)
        infos.append([runner.run(test_mode=False) for _ in range(3)])
        assert runner.t_env == 3 * 40 * 30 and agent.trained == 3 and agent.buffer.can_sample()
        bufs.append([t.clone() for t in agent.buffer.sample()])
        runner.close_env()
    assert infos[0] == infos[1]
    for a, b in zip(*bufs):
        assert th.equal(a, b)


def test_kernel_rollout_equals_per_step_rollout():
    """args.ec_kernel_rollout (Actor + env, the whole episode in one launch, then ec_store_episode) against the
    per-step launches (ec_policy_forward, ec_step, ec_store_transition) fed the same uniforms: the agent's buffer and
    episode_info are identical, bit for bit, over two episodes (the second continues the envs' RNG streams); test
    mode leaves the buffer alone."""
    from marl_amd.agent_register import agent_register
    from marl_amd.runners.runner_register import runner_register
    from marl_amd.utils import mlp_device
    B, T = 70, 30

    def make(kernel):
        args = _args(B, ec_env_id_base=3, episode_limit=T, ec_kernel_rollout=kernel, ec_hop_hist_len=32)
        th.manual_seed(7)
        agent = agent_register["ppo"](args)
        with th.no_grad():
            for p in agent.actor.parameters():   # away from the near-uniform initial policy
                p.mul_(4.0)
        agent.train = lambda: None
        return args, agent, runner_register["parallel"](args, agent)

    args, agent, runner = make(1)
    assert runner._kernel_rollout_ok()
    infos, bufs, us = [], [], []
    for ep in range(2):
        infos.append(runner.run(test_mode=False))
        bufs.append([t.clone() for t in agent.buffer.sample()])
        us.append(runner._uniforms.clone())
    before = [t.clone() for t in agent.buffer.sample()]
    info_test = runner.run(test_mode=True)
    assert np.isfinite(info_test["episode_return"])
    runner.close_env()

    args2, agent2, runner2 = make(0)
    assert not runner2._kernel_rollout_ok()
    state = {"ep": 0, "t": 0}

    def select(obs, avail, test_mode=False):
        u = us[state["ep"]][state["t"]]
        state["t"] += 1
        _, _, a = mlp_device.policy_forward(mlp_device.linear_params(agent2.actor), obs, args2.n_agents, avail, u,
                                            want_probs=False)
        return a.view(obs.shape[:-1])

    agent2.select_actions = select
    for ep in range(2):
        state["ep"], state["t"] = ep, 0
        info = runner2.run(test_mode=False)
        assert info == infos[ep], (ep, info, infos[ep])
        for name, x, y in zip(("obs", "avail", "actions", "rewards", "masks", "next_obs"), bufs[ep], agent2.buffer.sample()):
            assert th.equal(x, y), (ep, name)
    runner2.close_env()
    assert len(th.unique(bufs[0][2])) > 3 and float(bufs[0][4][:, -1].sum()) == 0
    # the test-mode episode did not touch the buffer... except that the runner resets it first, as the per-step path does
    assert all(t.shape == b.shape for t, b in zip(agent.buffer.sample(), before))


def test_buffer_contents_three_writers_and_oracle(oracle_lib):
    """EpisodeBuffer contents (buffer/episode_buffer.py:21-28, parallel_episode_runner.py:101-113), independently of
    the store kernels: the same two episodes filed by (i) the runner's torch `insert` branch, (ii) ec_store_transition
    per step and (iii) ec_store_episode after the one-launch rollout give identical tensors, and what they hold is what
    the CPU oracle observes / returns when it replays the stored actions (obs[:, t], avail[:, t], team reward, mask,
    next_obs[:, t])."""
    from marl_amd import _abi
    from marl_amd.agent_register import agent_register
    from marl_amd.runners.runner_register import runner_register
    from marl_amd.utils import mlp_device
    from oracle.ec_oracle import OracleEnv
    B, T, base = 70, 25, 9

    def make(kernel):
        args = _args(B, ec_env_id_base=base, episode_limit=T, ec_kernel_rollout=kernel, ec_hop_hist_len=32)
        th.manual_seed(11)
        agent = agent_register["ppo"](args)
        with th.no_grad():
            for p in agent.actor.parameters():
                p.mul_(4.0)
        agent.train = lambda: None
        return args, agent, runner_register["parallel"](args, agent)

    # (iii) one launch per episode + ec_store_episode; its uniforms drive the other two writers
    args, agent, runner = make(1)
    assert runner._kernel_rollout_ok()
    bufs3, us = [], []
    for ep in range(2):
        runner.run(test_mode=False)
        bufs3.append([t.clone() for t in agent.buffer.sample()])
        us.append(runner._uniforms.clone())
        assert agent.buffer.next_obs_is_shifted_obs and agent.buffer.known_mask_sum == B * args.n_agents * (T - 1)
    tables = runner.env.tables
    runner.close_env()

    def replay(torch_insert):
        args2, agent2, runner2 = make(0)
        if torch_insert:
            runner2._store.usable = lambda buf: False   # the runner's own torch-op branch (:143-154)
        state = {"ep": 0, "t": 0}

        def select(obs, avail, test_mode=False):
            u = us[state["ep"]][state["t"]]
            state["t"] += 1
            _, _, a = mlp_device.policy_forward(mlp_device.linear_params(agent2.actor), obs.contiguous(), args2.n_agents,
                                                avail.contiguous(), u, want_probs=False)
            return a.view(obs.shape[:-1])

        agent2.select_actions = select
        out = []
        for ep in range(2):
            state["ep"], state["t"] = ep, 0
            runner2.run(test_mode=False)
            out.append([t.clone() for t in agent2.buffer.sample()])
            assert agent2.buffer.next_obs_is_shifted_obs == (not torch_insert)
        runner2.close_env()
        return out

    bufs1 = replay(True)    # (i)
    bufs2 = replay(False)   # (ii)
    names = ("obs", "avail_actions", "actions", "rewards", "masks", "next_obs")
    for ep in range(2):
        for name, x, y, z in zip(names, bufs1[ep], bufs2[ep], bufs3[ep]):
            assert th.equal(x, y), (ep, name, "torch insert vs ec_store_transition")
            assert th.equal(x, z), (ep, name, "torch insert vs ec_store_episode")

    # against the oracle: replay the stored actions
    orc = OracleEnv(_abi.config_from_args(args, hop_hist_len=32), tables, B, env_id_base=base, n_threads=4)
    for ep in range(2):
        obs, avail, actions, rewards, masks, next_obs = [t.cpu().numpy() for t in bufs3[ep]]
        orc.reset()
        for t in range(T):
            o, av = orc.observe()
            assert np.array_equal(obs[:, t], o.astype(np.float32)), (ep, t, "obs")
            assert np.array_equal(avail[:, t], av), (ep, t, "avail")
            out = orc.step(actions[:, t, :, 0])
            assert out["bad"] == 0
            r = out["reward"].astype(np.float32)
            assert np.array_equal(rewards[:, t, :, 0], np.repeat(r[:, None], args.n_agents, 1)), (ep, t, "reward")
            assert np.array_equal(masks[:, t, :, 0], np.full((B, args.n_agents), 0.0 if t == T - 1 else 1.0, np.float32))
            o2, _ = orc.observe()
            assert np.array_equal(next_obs[:, t], o2.astype(np.float32)), (ep, t, "next_obs")
    orc.close()


@pytest.mark.parametrize("kernel_rollout", [0, 1])
def test_gail_runner_n40_expert_in_lockstep_with_oracle(oracle_lib, kernel_rollout):
    """BASELINE configs[4] shape (edge_node_num=40, GAIL): GAILRunner + GAILAgent for two iterations on the scaled
    topology.  During expert guidance (ratio 1.0 below bc_guidance_steps, gail_agent.py:68-75) every env executes the
    expert's action; an oracle env batch is driven by the same actions in lock step and its expert
    (oc_expert_policy = optimal_agent.py:102-120,166-375 restated) must agree with the device's at every step, as
    must observation, mask and reward.  kernel_rollout=0: the per-step launches (ec_expert_policy), checked as they
    happen; 1: the whole episode in one launch (ec_rollout_mlp_expert_policy), checked from its histories."""
    from marl_amd import _abi
    from marl_amd.agent_register import agent_register
    from marl_amd.runners.runner_register import runner_register
    from oracle.ec_oracle import OracleEnv
    B, N, T, base = 40, 40, 20, 2
    a = _imitation_args("gail", B, edge_node_num=N, n_agents=N, n_actions=N + 1, state_shape=9 * N, episode_limit=T,
                        ec_env_id_base=base, ec_hop_hist_len=32, ec_kernel_rollout=kernel_rollout)
    th.manual_seed(9)
    agent = agent_register["gail"](a)
    assert agent._fused_ok() and agent._fg_actor is not None     # N + 1 = 41 actions: still the HIP policy / update path
    runner = runner_register["gail_runner"](a, agent)
    assert runner._kernel_rollout_ok(True) == bool(kernel_rollout)
    orc = OracleEnv(_abi.config_from_args(a, hop_hist_len=32), runner.env.tables, B, env_id_base=base, n_threads=4)
    state = {"t": 0, "checked": 0}
    inner = agent.get_expert_actions

    def get_expert_actions(env):
        acts = inner(env)
        o, av = orc.observe()
        assert np.array_equal(env.obs.cpu().numpy(), o.astype(np.float32)), state
        assert np.array_equal(env.avail.cpu().numpy(), av), state
        want = orc.expert_policy(av)
        assert np.array_equal(acts.cpu().numpy(), want), state
        out = orc.step(want)                                     # ratio 1.0: the expert's action is what env.step gets
        assert out["bad"] == 0
        state["reward"] = out["reward"].astype(np.float32)
        state["t"] += 1
        state["checked"] += 1
        return acts

    agent.get_expert_actions = get_expert_actions
    step = runner.env.step

    def checked_step(actions, **kw):
        r = step(actions, **kw)
        assert np.array_equal(r[0].cpu().numpy(), state["reward"]), state["t"]
        return r

    runner.env.step = checked_step
    for it in range(2):
        orc.reset()
        info = runner.run(test_mode=False, t_expert=it * B * T)
        assert it * B * T + B * T <= a.bc_guidance_steps        # still ratio 1.0
        assert np.isfinite(info["episode_return"]) and all(bool(th.isfinite(v)) for v in agent.last_losses.values())
        assert "discriminator_loss" in agent.last_losses        # >= 10 expert transitions: the discriminator trained
        if kernel_rollout:
            hist = {k: v.cpu().numpy() for k, v in runner._hist.items()}
            obs0, avail0 = runner.env.obs.cpu().numpy(), runner.env.avail.cpu().numpy()   # still from before the first step
            for t in range(T):
                o, av = orc.observe()
                assert np.array_equal(obs0 if t == 0 else hist["obs"][t - 1], o.astype(np.float32)), (it, t)
                assert np.array_equal(avail0 if t == 0 else hist["avail"][t - 1], av), (it, t)
                want = orc.expert_policy(av)
                assert np.array_equal(hist["expert_actions"][t], want), (it, t)
                out = orc.step(want)
                assert out["bad"] == 0
                assert np.array_equal(hist["reward"][t], out["reward"].astype(np.float32)), (it, t)
                state["checked"] += 1
            # the PPO buffer holds the AGENT's masked draws, not what the envs executed
            _, avail_b, actions_b, _, _, _ = agent.buffer.sample()
            assert bool((avail_b.gather(-1, actions_b) == 1).all())
    assert state["checked"] == 2 * T and len(agent.expert_transitions) == min(2 * T * B, a.max_expert_buffer_size)
    n = len(agent.expert_transitions)
    ea = agent.expert_transitions.actions[:n]
    assert int(ea.max()) <= N and int((ea < N).sum()) > 0      # neighbours / self, or idle (N) without a task
    runner.close_env()
    orc.close()


def test_gail_fused_update_matches_torch_update():
    """GAILAgent PPO part: fused kernels + autograd BC term accumulated into the same flat buffer vs pure autograd."""
    from marl_amd.agent_register import agent_register
    B = 24
    a1 = _imitation_args("gail", B, episode_limit=20)
    a2 = _imitation_args("gail", B, episode_limit=20, ec_torch_update=1)
    th.manual_seed(4)
    fused, plain = agent_register["gail"](a1), agent_register["gail"](a2)
    assert fused._fg_actor is not None and plain._fg_actor is None
    for m1, m2 in ((fused.actor, plain.actor), (fused.critic, plain.critic)):
        m2.load_state_dict(m1.state_dict())
    g = th.Generator(device="cuda").manual_seed(0)
    N, A, T = a1.n_agents, a1.n_actions, 20
    batch = [th.rand(B, T, N, 9, device="cuda", generator=g), th.ones(B, T, N, A, dtype=th.int, device="cuda"),
             th.randint(0, A, (B, T, N, 1), device="cuda", generator=g), th.randn(B, T, 1, 1, device="cuda", generator=g).expand(B, T, N, 1),
             th.ones(B, T, N, 1, device="cuda"), th.rand(B, T, N, 9, device="cuda", generator=g)]
    batch[4][:, -1] = 0
    for ag in (fused, plain):
        for dst, src in zip(ag.buffer.sample(), batch):
            dst.copy_(src)
        ag.buffer.t = T
    # fewer than 10 expert transitions: no discriminator step, env rewards are used, BC loss on the same 5 transitions
    eo, ea = th.rand(5, N, 9, device="cuda", generator=g), th.randint(0, A, (5, N), device="cuda", generator=g)
    for ag in (fused, plain):
        ag.store_expert_data(eo, ea)
    ptrs = [p.grad.data_ptr() for p in fused.actor.parameters()]
    fused.train()
    plain.train()
    assert ptrs == [p.grad.data_ptr() for p in fused.actor.parameters()]      # autograd accumulated in place
    for k in ("critic_loss", "actor_loss"):
        x, y = float(fused.last_losses[k]), float(plain.last_losses[k])
        assert abs(x - y) <= 1e-4 * max(1.0, abs(y)), (k, x, y)
    for (n1, p1), (_, p2) in zip(list(fused.actor.named_parameters()) + list(fused.critic.named_parameters()),
                                 list(plain.actor.named_parameters()) + list(plain.critic.named_parameters())):
        assert float((p1 - p2).abs().max()) <= 0.05 * 4 * a1.lr + 1e-6, (n1, float((p1 - p2).abs().max()))


def test_main_driver_trains_tests_and_checkpoints(tmp_path, monkeypatch):
    """python -m marl_amd.main --name=ppo ... : train loop with periodic evaluation (utils/run_utils.py:15-140),
    scalar tags, checkpoint files under results/models/<folder>/<run>/ that load back."""
    import json
    from marl_amd import main as driver
    monkeypatch.chdir(tmp_path)
    argv = ["--name=ppo", "--task_size_max=6000", "--seed=100", "--runner=parallel", "--n_threads=16", "--episode_limit=20",
            "--t_max=900", "--test_interval=320", "--test_nepisode=32", "--folder=t", "--log_tag=unit"]
    out = driver.main(argv)
    assert out["episodes"] == 3                       # t_env 320, 640, 960 > t_max
    run_dir = tmp_path / "results" / "models" / "t" / out["run"]
    assert (run_dir / "actor.th").exists() and (run_dir / "critic.th").exists()
    tags = {json.loads(l)["tag"] for l in open(tmp_path / "results" / "tensorboard" / "t" / out["run"] / "scalars.jsonl")}
    assert {"Training/Episode_Reward", "Training/Cumulative_Avg_Reward", "Training/Moving_Avg_Reward_100",
            "success_rate_avg", "success_rate_std", "Test/Episode_Return_Avg"} <= tags
    res = driver.main(["--name=ppo", "--task_size_max=6000", "--n_threads=16", "--episode_limit=20", "--test_nepisode=32",
                       "--test_models=1", "--load_models=1", "--folder=t", "--checkpoint_path=" + out["run"]])
    assert 0.0 <= res["success_rate"] <= 1.0 and abs(res["success_rate"] + res["drop_rate"] + res["failure_rate"] - 1) < 1e-9


@pytest.mark.parametrize("B,cap,ratio", [(64, 1000, 0.5), (700, 100, 0.9), (4096, 1000, 1.0), (33, 50, 0.0), (1, 5, 1.0)])
def test_expert_mix_equals_torch_append(B, cap, ratio):
    """ec_expert_mix (coin, action select, FIFO append of the guided runners: gail_runner.py:72-98, gail_agent.py:77-83)
    against the torch formulation (rand < ratio, th.where, ExpertTransitions.append), over several steps so that the
    ring wraps; B > capacity keeps the newest rows.  Bit-exact: ring contents, counters, executed actions."""
    import torch as th
    from marl_amd.buffer.expert_transitions import ExpertTransitions
    dev = th.device("cuda")
    N, O, A = 10, 9, 11
    g = th.Generator().manual_seed(B + cap)
    ref, got = ExpertTransitions(cap, N, O, dev), ExpertTransitions(cap, N, O, dev)
    n_ref, n_got = th.zeros((), dtype=th.int64, device=dev), th.zeros((), dtype=th.int64, device=dev)
    r = th.tensor([ratio], dtype=th.float32, device=dev)
    for step in range(7):
        coin = th.rand(B, generator=g).to(dev)
        obs = th.rand(B, N, O, generator=g).to(dev)
        ea = th.randint(0, A, (B, N), generator=g).to(dev)
        aa = th.randint(0, A, (B, N), generator=g).to(dev)
        use = coin < r
        want_exec = th.where(use.unsqueeze(-1), ea, aa)
        ref.append(obs, ea, use)
        n_ref += use.sum()
        got_exec = got.mix_append(coin, r, ea, aa, obs, n_got)
        assert th.equal(got_exec, want_exec), step
        assert int(got._written) == int(ref._written) and int(n_got) == int(n_ref), step
        k = len(ref)
        assert len(got) == k
        assert th.equal(got.obs[:cap], ref.obs[:cap]) and th.equal(got.actions[:cap], ref.actions[:cap]), step


@pytest.mark.gpu
@pytest.mark.parametrize("B,T,cap,ratio", [(37, 5, 1000, 0.5), (300, 7, 64, 0.6), (700, 3, 10_000, 1.0), (513, 4, 100, 0.02),
                                           (64, 6, 50, 0.0)])
def test_expert_append_episode_equals_per_step_appends(B, T, cap, ratio):
    """ec_expert_append_episode (the FIFO appends of a whole kernel rollout) leaves the ring as T per-step append() calls
    do: wrap-around, more selected rows than the ring holds, several 256-env chunks, nothing selected; twice in a row."""
    import torch as th
    from marl_amd.buffer.expert_transitions import ExpertTransitions
    dev = th.device("cuda:0")
    N, O = 10, 9
    gen = th.Generator().manual_seed(B * 31 + T)
    a, b = ExpertTransitions(cap, N, O, dev), ExpertTransitions(cap, N, O, dev)
    n_a = th.zeros((), dtype=th.int64, device=dev)
    total = 0
    for rep in range(2):
        coin = th.rand(T, B, generator=gen).to(dev)
        ratios = (th.full((T,), ratio) * th.linspace(1.0, 0.5, T)).to(dev)
        obs0 = th.randn(B, N, O, generator=gen).to(dev)
        obs_hist = th.randn(T, B, N, O, generator=gen).to(dev)
        ex = th.randint(0, N + 1, (T, B, N), generator=gen).to(dev)
        for t in range(T):
            use = coin[t] < ratios[t]
            b.append(obs0 if t == 0 else obs_hist[t - 1], ex[t], use)
            total += int(use.sum())
        a.append_episode(coin, ratios, ex, obs0, obs_hist, n_a)
        assert int(a._written) == int(b._written) == total == int(n_a)
        n = len(b)
        assert len(a) == n
        assert th.equal(a.obs[:n], b.obs[:n]) and th.equal(a.actions[:n], b.actions[:n]), rep


@pytest.mark.gpu
def test_guided_rollout_entry_points_reject_bad_arguments():
    """ec_rollout_mlp_expert_policy / ec_expert_append_episode / ec_adam_clip_step say what is wrong instead of launching:
    coin without ratios, wrong shapes, a missing workspace, too many networks."""
    import ctypes as C
    import torch as th
    from marl_amd import _abi
    from marl_amd.buffer.expert_transitions import ExpertTransitions
    from marl_amd.envs.env_register import env_register
    a = _imitation_args("gail", 4, episode_limit=10)
    env = env_register["edge_computing_batched"](a, n_envs=4)
    env.reset()
    N = a.n_agents
    dev = env.device
    params = tuple(th.zeros(*d, device=dev) for d in [(64, 9 + N), (64,), (64, 64), (64,), (N + 1, 64), (N + 1,)])
    coin = th.rand(3, 4, device=dev)
    with pytest.raises(ValueError, match="go together"):
        env.rollout_mlp_policy(params, 3, coin=coin)
    with pytest.raises(ValueError, match="ratios"):
        env.rollout_mlp_policy(params, 3, coin=coin, ratios=th.ones(4, device=dev))
    with pytest.raises(ValueError, match="coin"):
        env.rollout_mlp_policy(params, 3, coin=coin.double(), ratios=th.ones(3, device=dev))
    hist = env.rollout_mlp_policy(params, 3, coin=coin, ratios=th.ones(3, device=dev))       # the well-formed call works
    assert hist["expert_actions"].shape == (3, 4, N)
    L = _abi.lib()
    vp = lambda t: C.c_void_p(t.data_ptr())
    null = C.c_void_p(0)
    ring = ExpertTransitions(8, N, 9, dev)
    rc = L.ec_expert_append_episode(vp(coin), vp(th.ones(3, device=dev)), vp(hist["expert_actions"]), vp(env.obs), vp(hist["obs"]),
                                    3, 4, N, 9, 8, vp(ring.obs), vp(ring.actions), vp(ring._written), null, null, null)
    assert rc != 0 and b"null" in L.ec_last_error()
    with pytest.raises(ValueError, match="shapes"):
        ring.append_episode(coin, th.ones(2, device=dev), hist["expert_actions"], env.obs, hist["obs"])
    assert int(ring._written) == 0
    ring.append_episode(coin, th.ones(3, device=dev), hist["expert_actions"], env.obs, hist["obs"])   # ratio 1: every env, every step
    assert int(ring._written) == 12 and len(ring) == 8
    assert th.equal(ring.actions[(12 - 1) % 8], hist["expert_actions"][2, 3])                # the newest entry is (t=2, b=3)
    n5 = (C.c_int32 * 5)(1, 1, 1, 1, 1)
    assert L.ec_adam_clip_step(5, n5, null, null, null, null, null, null, null, null, null, null, null, null, null) != 0
    assert b"networks" in L.ec_last_error()
    env.close()


@pytest.mark.parametrize("kind", ["ppo", "gail", "dgail"])
def test_unfused_shapes_raise_unless_torch_update_is_asked_for(kind):
    """A GPU learner whose shape the fused update kernels do not cover (hidden_dim != 64) must not slide onto the
    torch-autograd formulation silently (10x slower): it raises, naming --ec_torch_update, and runs when that is set."""
    from marl_amd.agent_register import agent_register
    from marl_amd.utils.rl_utils import UnfusedShape
    make = (lambda **kw: _args(4, **kw)) if kind == "ppo" else (lambda **kw: _imitation_args(kind, 4, **kw))
    with pytest.raises(UnfusedShape, match="ec_torch_update"):
        agent_register[kind](make(hidden_dim=128))
    agent = agent_register[kind](make(hidden_dim=128, ec_torch_update=1))
    assert agent.device.type == "cuda"
