"""Host-side logic that needs no GPU: registries and flag defaults, the rollout buffer, and the data-parallel form of
the PPO learner on two `gloo` ranks (world_size 2) against one rank holding the whole batch.

The learner's GAE runs on the HIP library in the product; here `get_gae` is replaced by the reference's T-step torch
loop (utils/advantage_utils.py:4-9 restated) so that `PPOAgent.train()` can run on CPU tensors — this file tests the
sharding / all-reduce arithmetic, not the kernel (tests/test_gae_gpu.py does that).
"""
import argparse
import os
import socket

import numpy as np
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


def test_graphml_topology_matches_networkx(tmp_path):
    """generate_graph2 (network_graph.py:267-307): the stdlib GraphML reader yields the edge list, in the order,
    that `nx.read_graphml(path).edges()` yields (the order assigns the link draws), and build_static_tables turns it
    into the tables the reference's Graph / EdgeNode would hold (two Link passes over RandomState(link_seed))."""
    nx = pytest.importorskip("networkx")
    from marl_amd.envs.edge_computing.multihop import network_graph as ng
    g = nx.Graph()
    g.add_nodes_from(str(i) for i in (0, 3, 1, 2, 4, 5, 6))     # document order != numeric order
    for u, v in ((3, 0), (0, 1), (2, 1), (5, 6), (4, 0), (6, 2), (3, 5), (1, 4)):
        g.add_edge(str(u), str(v), LinkLabel="x")
    path = str(tmp_path / "topo.graphml")
    nx.write_graphml(g, path)
    want = [(int(u), int(v)) for u, v in nx.read_graphml(path).edges()]
    n, edges = ng.graphml_edges(path)
    assert n == 7 and edges == want
    a = make_args(["--edge_node_num=7", "--topology_file=" + path])
    t = ng.build_static_tables(a)
    assert t.edges == want
    # link parameters: pass 1 keeps the rates, pass 2 the failure rates, two uniforms per Link (links.py:7-9)
    rs = np.random.RandomState(a.link_seed)
    rates = []
    for _ in want:
        rates.append(rs.uniform(a.transmission_rate_min, a.transmission_rate_max) * 1000000 * 8)
        rs.uniform(a.transmission_failure_rate_min, a.transmission_failure_rate_max)
    fails = []
    for _ in want:
        rs.uniform(a.transmission_rate_min, a.transmission_rate_max)
        fails.append(rs.uniform(a.transmission_failure_rate_min, a.transmission_failure_rate_max))
    for (u, v), r, f in zip(want, rates, fails):
        assert t.rate[u, v] == t.rate[v, u] == r and t.tfr[u, v] == t.tfr[v, u] == f
        assert t.adj[u, v] == t.adj[v, u] == 1
    assert int(t.adj[:, :7].sum()) == 2 * len(want) + 7 and not t.adj[:, 7].any()
    with pytest.raises(ValueError):
        ng.build_static_tables(make_args(["--edge_node_num=9", "--topology_file=" + path]))


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
        # plain numpy through the queue: torch tensors travel as shared-memory handles that die with the child
        q.put((rank, float(mean), float(std), float(x.mean()), float(x.std()), gsum, flat.numpy().copy(), losses))
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
    assert (got[0][6] == got[1][6]).all()
    # ... equal (to fp32 summation-order noise) to one rank training on the whole batch
    flat1, losses1 = _train_once(_small_args(8), 0, 8)
    assert th.allclose(th.from_numpy(got[0][6]), flat1, rtol=1e-4, atol=2e-6), float((th.from_numpy(got[0][6]) - flat1).abs().max())
    for k in losses1:
        assert abs(got[0][7][k] - losses1[k]) < 1e-5 * max(1.0, abs(losses1[k]))


# ---------------------------------------------------------------------------------------------- GAIL / DGAIL host logic
def test_expert_transitions_fifo():
    from marl_amd.buffer.expert_transitions import ExpertTransitions
    st = ExpertTransitions(5, n_agents=2, obs_shape=3, device="cpu")
    assert len(st) == 0
    mk = lambda lo, n: (th.arange(lo, lo + n).float().view(n, 1, 1).expand(n, 2, 3).contiguous(),
                        th.arange(lo, lo + n).view(n, 1).expand(n, 2).contiguous())
    o, a = mk(0, 4)
    st.append(o, a, th.tensor([True, False, True, True]))        # stores 0, 2, 3
    assert len(st) == 3 and st.actions[:3, 0].tolist() == [0, 2, 3]
    o, a = mk(10, 4)
    st.append(o, a)                                               # 10..13: ring of 5 wraps, oldest (0, 2) dropped
    assert len(st) == 5
    assert sorted(st.actions[:5, 0].tolist()) == [3, 10, 11, 12, 13]
    assert th.equal(st.obs[:5, 0, 0].long(), st.actions[:5, 0])   # obs and actions stay paired
    so, sa = st.sample_transitions(64)
    assert so.shape == (5, 2, 3) and sorted(sa[:, 0].tolist()) == [3, 10, 11, 12, 13]   # without replacement
    ro, ra = st.sample_rows(1000)
    assert ro.shape == (1000, 3) and set(ra.tolist()) <= {3, 10, 11, 12, 13}
    # the checkpoint is the reference's expert_data.th layout (gail_agent.py:244-248): lists of per-step tensors,
    # oldest first, although the ring has wrapped
    sd = st.state_dict()
    assert isinstance(sd["expert_obs"], list) and sd["expert_obs"][0].shape == (2, 3) and sd["expert_actions"][0].shape == (2,)
    assert [int(x[0]) for x in sd["expert_actions"]] == [3, 10, 11, 12, 13]
    assert all(int(o_[0, 0]) == int(a_[0]) for o_, a_ in zip(sd["expert_obs"], sd["expert_actions"]))
    st2 = ExpertTransitions(5, 2, 3, "cpu")
    st2.load_state_dict(sd)
    assert len(st2) == 5 and st2.actions[:5, 0].tolist() == [3, 10, 11, 12, 13]
    st3 = ExpertTransitions(5, 2, 3, "cpu")                       # the reference's list-of-tensors checkpoint layout
    st3.load_state_dict({"expert_obs": [o[i] for i in range(4)], "expert_actions": [a[i] for i in range(4)]})
    assert len(st3) == 4


def test_expert_transitions_more_rows_than_capacity():
    """One append can select more envs than the ring holds (n_threads > max_expert_buffer_size): the FIFO keeps the
    newest `capacity` of them, every slot from exactly one source row, (obs, action) pairs intact -- what appending
    the rows one by one (gail_agent.py:77-83) would leave."""
    from marl_amd.buffer.expert_transitions import ExpertTransitions
    cap, B, N = 5, 23, 3
    st = ExpertTransitions(cap, N, 9, "cpu")
    for rnd in range(3):
        ids = th.arange(B) + 100 * rnd
        obs = ids.view(B, 1, 1).expand(B, N, 9).float().contiguous()   # every element of a row carries the row id
        act = ids.view(B, 1).expand(B, N).contiguous()
        select = th.ones(B, dtype=th.bool) if rnd != 1 else (th.arange(B) % 2 == 0)
        st.append(obs, act, select)
        assert len(st) == cap
        chosen = ids[select][-cap:]                                    # newest `cap` selected rows
        got_obs, got_act = st.obs[:cap], st.actions[:cap]
        assert th.equal(got_obs[:, 0, 0].long(), got_act[:, 0])        # pairs stay matched
        assert (got_obs == got_obs[:, :1, :1]).all() and (got_act == got_act[:, :1]).all()  # no element-wise mixing
        assert sorted(got_act[:, 0].tolist()) == sorted(chosen.tolist())


def _imitation_args(kind, B):
    from marl_amd.default_config import add_default_args
    from marl_amd.envs.edge_computing.multihop.edge_computing_config import add_edge_computing_env_args
    from marl_amd.agent_register import agent_config_register
    p = argparse.ArgumentParser(conflict_handler="resolve")
    p = agent_config_register[kind](add_edge_computing_env_args(add_default_args(p)))
    a = p.parse_args([])
    a.n_threads, a.episode_limit, a.n_agents, a.n_actions, a.obs_shape = B, 6, 3, 4, 9
    a.state_shape, a.device, a.ec_force_device = 27, "cpu", "cpu"
    return a


def _fill_imitation(agent, a, seed=3):
    g = th.Generator().manual_seed(seed)
    B, T, N, A = a.n_threads, a.episode_limit, a.n_agents, a.n_actions
    agent.buffer.reset()
    for t in range(T):
        obs = th.rand(B, N, 9, generator=g)
        avail = th.ones(B, N, A, dtype=th.int)
        act = th.randint(0, A, (B, N, 1), generator=g)
        agent.buffer.insert(obs, avail, act, th.randn(B, 1, 1, generator=g).expand(B, N, 1),
                            th.ones(B, N, 1) * (t < T - 1), th.rand(B, N, 9, generator=g))
        agent.store_expert_data(obs, th.randint(0, A, (B, N), generator=g), th.rand(B, generator=g) < 0.8)


def test_dgail_agent_train_cpu():
    """reference defaults (dgail_config.py) + one full DGAILAgent.train on CPU tensors with the torch GAE loop."""
    import marl_amd.rl.policy_gradient_rl.dgail.dgail_agent as da
    da.get_gae = _torch_gae
    a = _imitation_args("dgail", 40)
    assert (a.ppo_batch_size, a.policy_lr, a.entropy_coef, a.value_loss_coef, a.grad_norm_clip) == (64, 2e-5, 8e-4, 0.2, 1.0)
    assert a.max_expert_buffer_size == 20000 and a.batch_size_run == 100 and a.use_simplified_reward is True
    th.manual_seed(0)
    agent = da.DGAILAgent(a)
    agent._bc_pretrain_steps = 5
    _fill_imitation(agent, a)
    assert agent.train(0) is not None or len(agent.expert_transitions) < a.batch_size_run
    assert len(agent.expert_transitions) >= a.batch_size_run
    before = [p.detach().clone() for p in agent.policy.parameters()]
    stats = agent.train(1000)
    assert set(stats) >= {"discriminator_loss", "gail_discriminator_loss", "prob_expert", "prob_agent",
                          "avg_imitation_reward", "policy_loss", "value_loss", "entropy"}
    assert all(np.isfinite(v) for v in stats.values())
    assert any(not th.equal(b, p) for b, p in zip(before, agent.policy.parameters()))
    assert agent.update_count >= 1 and agent._bc_pretrain_steps == 0
    # reward arithmetic of both branches against the formulas of dgail_agent.py:482-518
    obs, _, actions, env_r, _, _ = agent.buffer.sample()
    sa = agent._prepare_batch(obs, actions)
    with th.no_grad():
        p = agent.gail_discriminator(sa).clamp(1e-8, 1 - 1e-8).squeeze(-1)
        want = -th.log(1 - p) * (1.0 - 0.2 * min(1.0, 40000 / 80000.0)) + 0.02 * env_r.reshape(-1)
    got = agent._compute_rewards(obs, actions, env_r, 40000)
    assert got.shape == (40, 6, 3, 1) and th.allclose(got.reshape(-1), want, rtol=1e-6, atol=1e-7)
    agent._use_simplified_reward = False
    r2 = agent._compute_rewards(obs, actions, env_r, 40000)
    assert r2.shape == got.shape and bool(th.isfinite(r2).all())
    # schedules, dgail_agent.py:383-401,170-178
    for _ in range(6):
        agent.update_performance_monitoring(0.5)
    assert agent._w_factor > 1.0 and len(agent.performance_history) == 6
    w = agent.bc_loss_weight
    agent._update_bc_weight()
    assert agent.bc_loss_weight == min(2.0, w * 1.1)


def test_gail_agent_train_cpu():
    import marl_amd.rl.policy_gradient_rl.gail.gail_agent as ga
    ga.get_gae = _torch_gae
    a = _imitation_args("gail", 16)
    assert (a.eps_clip, a.lr, a.bc_loss_weight, a.max_expert_buffer_size, a.expert_guidance_steps) == (0.1, 3e-4, 2, 1000, 40000)
    th.manual_seed(0)
    agent = ga.GAILAgent(a)
    assert agent.get_expert_ratio(0) == 1.0 and agent.get_expert_ratio(10 ** 6) == 0.2
    assert abs(agent.get_expert_ratio(21000) - 0.45) < 1e-12
    _fill_imitation(agent, a)
    before = [p.detach().clone() for p in list(agent.actor.parameters()) + list(agent.discriminator.parameters())]
    agent.train()
    after = list(agent.actor.parameters()) + list(agent.discriminator.parameters())
    assert any(not th.equal(b, p) for b, p in zip(before, after))
    assert all(bool(th.isfinite(v)) for v in agent.last_losses.values()) and "discriminator_loss" in agent.last_losses
    assert agent.bc_loss_weight == max(a.min_bc_weight, 2 * a.bc_decay_rate)
    # discriminator layout of gail_network.py:59-132
    names = [n for n, _ in agent.discriminator.named_parameters()]
    assert names[:4] == ["fc1.weight", "fc1.bias", "bn1.weight", "bn1.bias"] and agent.discriminator.fc1.in_features == 9 + 3 + 4


@pytest.mark.parametrize("N", [10, 40])
def test_network_mirrors_match_reference_fixtures(N):
    """The package's torch modules load the REFERENCE's state_dicts (same parameter names) and reproduce the outputs
    the reference's own modules produced (tests/golden/mlp.npz, made by importing rl/policy_gradient_rl/...)."""
    from marl_amd.rl.policy_gradient_rl.dgail.dgail_network import DiffusionDiscriminator, GAILDiscriminator, PolicyNet
    from marl_amd.rl.policy_gradient_rl.ppo.ppo_network import Actor
    from tests.util import load_golden
    g = load_golden("mlp")
    tag = "n%d_" % N
    A, SA = N + 1, 10 + N
    obs, avail, actions = (th.from_numpy(g[tag + k]) for k in ("obs", "avail", "actions"))
    rows = obs.shape[0]
    ids = th.eye(N)[th.arange(rows) % N]
    sd = lambda name: {k[len(tag + name + "_sd_"):]: th.from_numpy(g[k]) for k in g.files if k.startswith(tag + name + "_sd_")}
    ns = argparse.Namespace(obs_shape=9, n_agents=N, n_actions=A, hidden_dim=64, activate_fun="relu", use_orthogonal_init=False)
    for name, net in (("actor", Actor(ns)), ("policy", PolicyNet(9 + N, A, 64))):
        net.load_state_dict(sd(name))
        with th.no_grad():
            logits = net(th.cat([obs, ids], -1))
        logits[avail == 0] = -1e10
        assert th.allclose(logits, th.from_numpy(g[tag + name + "_logits"]), rtol=1e-6, atol=1e-6)
    sa = th.cat([obs, th.nn.functional.one_hot(actions, A).float()], -1)
    gd = GAILDiscriminator(SA, 64)
    gd.load_state_dict(sd("gd"))
    with th.no_grad():
        assert th.allclose(gd(sa).squeeze(-1), th.from_numpy(g[tag + "gd_prob"]), rtol=1e-6, atol=1e-7)
    dd = DiffusionDiscriminator(SA, 64)
    dd.load_state_dict(sd("dd"))
    dd.eval()
    assert th.allclose(dd.mlp[0].weight, th.from_numpy(g[tag + "dd_w1"]), rtol=1e-6, atol=1e-7)   # effective weights
    for cname, cval in (("expert", 1), ("agent", 0)):
        L = dd.compute_loss(sa, th.full((rows,), cval, dtype=th.long), th.from_numpy(g[tag + "dd_t_" + cname]).long(),
                            th.from_numpy(g[tag + "dd_noise_" + cname])).detach()
        assert th.allclose(L, th.from_numpy(g[tag + "dd_loss_" + cname]), rtol=1e-5, atol=1e-6)


def test_splitk_linear_and_select_embedding_match_stock_autograd():
    """SplitKLinear / _embed2 (dgail_network.py mirrors) give the stock modules' values and gradients, including the
    double backward of DiffusionDiscriminator.compute_loss (create_graph=True)."""
    import copy
    from marl_amd.rl.policy_gradient_rl.dgail import dgail_network as dn
    th.manual_seed(0)
    rows = 9000                                            # above SplitKLinear.MIN_ROWS, not a multiple of the chunk
    a = dn.SplitKLinear(20, 64).double()
    b = th.nn.Linear(20, 64).double()
    b.load_state_dict(a.state_dict())
    x1 = th.randn(rows, 20, dtype=th.float64, requires_grad=True)
    x2 = x1.detach().clone().requires_grad_(True)
    (a(x1).tanh().sum()).backward()
    (b(x2).tanh().sum()).backward()
    assert th.allclose(x1.grad, x2.grad, rtol=1e-12, atol=1e-14)
    assert th.allclose(a.weight.grad, b.weight.grad, rtol=1e-10, atol=1e-12) and th.allclose(a.bias.grad, b.bias.grad, rtol=1e-10)

    net = dn.DiffusionDiscriminator(20, 64).double()
    ref = copy.deepcopy(net)
    ref.forward = lambda x, t, c: ref.mlp(th.cat([x, ref.time_embed(t), ref.cond_embed(c)], dim=-1))   # stock embedding
    for m in ref.mlp:
        if isinstance(m, dn.SplitKLinear):
            m.MIN_ROWS = 10 ** 9                                                                       # stock linear
    sa = th.rand(rows, 20, dtype=th.float64)
    t = th.randint(0, 2, (rows,))
    noise = th.randn(rows, 20, dtype=th.float64)
    c = th.randint(0, 2, (rows,))
    for n in (net, ref):
        n.train()
    la = net.compute_loss(sa, c, t, noise)
    lb = ref.compute_loss(sa, c, t, noise)
    assert th.allclose(la, lb, rtol=1e-10, atol=1e-12)
    la.mean().backward()
    lb.mean().backward()
    for (k, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        assert th.allclose(p.grad, q.grad, rtol=1e-8, atol=1e-11), k


def test_tall_batchnorm_matches_stock():
    """gail_network.TallBatchNorm1d's large-input path (column reductions + fused multiply-adds) against nn.BatchNorm1d:
    outputs, input / weight / bias gradients and running statistics."""
    import torch as th
    from marl_amd.rl.policy_gradient_rl.gail.gail_network import TallBatchNorm1d, _TallBatchNormFn
    th.manual_seed(5)
    rows, C = 1000, 24
    x = (th.randn(rows, C) * th.linspace(0.5, 3.0, C) + th.linspace(-2, 2, C)).double()
    g = th.randn(rows, C).double()
    stock, tall = th.nn.BatchNorm1d(C).double(), TallBatchNorm1d(C).double()
    with th.no_grad():
        for m in (stock, tall):
            m.weight.copy_(th.linspace(0.5, 1.5, C))
            m.bias.copy_(th.linspace(-1, 1, C))
    old_numel, old_chunk = TallBatchNorm1d.TALL_NUMEL, _TallBatchNormFn.CHUNK
    TallBatchNorm1d.TALL_NUMEL, _TallBatchNormFn.CHUNK = 0, 300   # force the path, several reduction chunks
    try:
        outs = []
        for m in (stock, tall):
            xi = x.clone().requires_grad_(True)
            for _ in range(2):  # two calls: running statistics and num_batches_tracked advance alike
                y = m(xi)
            (y * g).sum().backward()
            outs.append((y.detach(), xi.grad, m.weight.grad, m.bias.grad, m.running_mean, m.running_var,
                         m.num_batches_tracked))
    finally:
        TallBatchNorm1d.TALL_NUMEL, _TallBatchNormFn.CHUNK = old_numel, old_chunk
    for a, b in zip(*outs):
        assert th.allclose(a.double(), b.double(), rtol=1e-10, atol=1e-11), (a - b).abs().max()
    # below the threshold it IS the stock module
    assert th.equal(tall.eval()(x), th.nn.functional.batch_norm(x, tall.running_mean, tall.running_var, tall.weight, tall.bias,
                                                               False, 0.1, tall.eps))


def test_tall_bn_relu_dropout_node_matches_modules(monkeypatch):
    """The one-node BatchNorm -> ReLU -> Dropout of the GAIL discriminator's tall-input path against the three stock
    modules on the same keep mask: output, input / weight / bias gradients, running statistics."""
    import torch as th
    import marl_amd.rl.policy_gradient_rl.gail.gail_network as gn
    th.manual_seed(6)
    rows, C, p = 900, 16, 0.3
    x = (th.randn(rows, C) * 2 + th.linspace(-1, 1, C)).double()
    g = th.randn(rows, C).double()
    mask = th.rand(rows, C) >= p
    monkeypatch.setattr(gn, "_keep_mask", lambda x_, p_: mask)
    monkeypatch.setattr(gn._TallBatchNormFn, "CHUNK", 256)
    stock, tall = th.nn.BatchNorm1d(C).double(), gn.TallBatchNorm1d(C).double()
    with th.no_grad():
        for m in (stock, tall):
            m.weight.copy_(th.linspace(0.5, 1.5, C))
            m.bias.copy_(th.linspace(-0.5, 0.5, C))
    xa = x.clone().requires_grad_(True)
    out_a = th.relu(stock(xa)) * mask / (1 - p)
    (out_a * g).sum().backward()
    xb = x.clone().requires_grad_(True)
    out_b = tall.relu_dropout(xb, p)
    (out_b * g).sum().backward()
    for a, b in ((out_a, out_b), (xa.grad, xb.grad), (stock.weight.grad, tall.weight.grad), (stock.bias.grad, tall.bias.grad),
                 (stock.running_mean, tall.running_mean), (stock.running_var, tall.running_var)):
        assert th.allclose(a.detach(), b.detach(), rtol=1e-10, atol=1e-11), (a - b).abs().max()


def test_clip_adam_step_declines_on_cpu_and_leaves_everything_alone():
    """optim_device.clip_adam_step is a device path only: for CPU parameters (and for anything that is not a plain fp32
    Adam) it returns False without touching parameters, gradients or optimiser state, and the caller's torch path runs."""
    import torch as th
    from marl_amd.utils import optim_device
    net = th.nn.Linear(4, 3)
    for p in net.parameters():
        p.grad = th.ones_like(p)
    before = [p.detach().clone() for p in net.parameters()]
    for opt in (th.optim.Adam(net.parameters(), lr=1e-2), th.optim.Adam(net.parameters(), lr=1e-2, weight_decay=0.1),
                th.optim.SGD(net.parameters(), lr=1e-2)):
        assert optim_device.clip_adam_step([opt], 1.0) is False
        assert len(opt.state) == 0
    assert optim_device.clip_adam_step([], 1.0) is False
    assert all(th.equal(a, b) for a, b in zip(before, net.parameters()))
    assert all(bool((p.grad == 1).all()) for p in net.parameters())
