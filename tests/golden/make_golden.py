#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by IMPORTING the reference.

Run in the build container only (the reference never travels to the GPU box):

    python tests/golden/make_golden.py            # all fixtures
    python tests/golden/make_golden.py env10      # one group

Every fixture is data only: inputs (seeds, actions) and the outputs the
reference produced for them.  Each group runs in a fresh interpreter because
the reference caches its topology at class level (edge_node.py:19-20,87-91) and
seeds stdlib ``random`` at import (network_graph.py:13).

Groups
  rng      RandomState stream prefixes (edge_node.py:28-29, task.py:10-12)
  env10    default topology (N=10, task_size_max=6000): static tables + action replay
  env40    scaled topology (N=40): static tables + short action replay
  gae      utils/advantage_utils.py:4-9 on random inputs
  expert10 / expert40   OptimalAgent.get_current_optimal_action (optimal/optimal_agent.py:102-120,166-375) on the
           states of replayed trajectories (expert-driven and expert/random mixtures)
  mlp      the reference's torch modules on fixed inputs (logits, probabilities, discriminator rewards)
  learn    the reference's PPOAgent / GAILAgent / DGAILAgent: one train() on a fixed buffer with every random draw
           recorded => post-update state_dicts
"""
import argparse
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("MARL_REFERENCE", "/root/reference")


# --------------------------------------------------------------------------
def _ref_args(extra):
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    from default_config import add_default_args
    from envs.edge_computing.multihop.edge_computing_config import add_edge_computing_env_args
    from rl.policy_gradient_rl.ppo.ppo_config import add_ppo_args
    p = argparse.ArgumentParser(conflict_handler='resolve')
    p = add_default_args(p)
    p = add_edge_computing_env_args(p)
    p = add_ppo_args(p)
    return p.parse_args(extra)


def _ticks(x):
    """timers are sums of k additions of mini_time_slot: recover k."""
    return int(round(x / 0.05))


def _snapshot_rows(env, step_idx):
    """One row per live task: [step,node,qtype,pos,size,cycles,hop,ewt,cet,twt,ctt,bwt]
    qtype: 0 new_task, 1 buffer_queue, 2 execution_queue, 3 executing_queue, 4+src receiving_queues[src]."""
    rows = []

    def put(node, qtype, pos, t):
        rows.append([step_idx, node, qtype, pos, t.task_size, t.task_cpu_cycle, t.hop,
                     _ticks(t.execute_waiting_time), _ticks(t.current_execute_time),
                     _ticks(t.transmission_waiting_time), _ticks(t.current_transmission_time),
                     _ticks(t.buffer_waiting_time)])

    for n in env.edge_nodes:
        if n.new_task is not None:
            put(n.id, 0, 0, n.new_task)
        for j, t in enumerate(n.buffer_queue):
            put(n.id, 1, j, t)
        for j, t in enumerate(n.execution_queue):
            put(n.id, 2, j, t)
        for j, t in enumerate(n.executing_queue):
            put(n.id, 3, j, t)
        for s, q in enumerate(n.receiving_queues):
            for j, t in enumerate(q):
                put(n.id, 4 + s, j, t)
    return rows


def _policy(name, rng, env, avail):
    """Deterministic test policies (ours, not the reference's)."""
    import numpy as np
    N = len(avail)
    acts = np.zeros(N, dtype=np.int64)
    for i in range(N):
        a = np.asarray(avail[i])
        node = env.edge_nodes[i]
        if name == "random_valid":
            idx = np.flatnonzero(a)
            acts[i] = idx[rng.randint(len(idx))]
        elif name == "local_first":
            acts[i] = i if a[i] else int(np.flatnonzero(a)[0])
        elif name == "max_forward":
            if a[N]:
                acts[i] = N
            else:
                nb = [j for j in np.flatnonzero(a) if j != i and j != N]
                acts[i] = nb[(env.current_step + i) % len(nb)] if nb else i
        elif name == "unmasked":
            # ignores the queue-length mask (still adjacency-legal, so no ZeroDivisionError in
            # edge_node.py:227) and sometimes idles with a pending task (edge_node.py:208-210)
            u = rng.random_sample()
            if u < 0.1:
                acts[i] = N
            else:
                cand = list(node.actions)
                acts[i] = cand[rng.randint(len(cand))]
        elif name in ("expert", "mixed"):
            pass  # whole-env decision, see _replay
        else:
            raise KeyError(name)
    return acts


def _replay(args, k, policy, episodes, snap_every, pol_seed, expert=None):
    """ParallelRunner semantics for env k (parallel_episode_runner.py:17-21,40-113)."""
    import copy
    import numpy as np
    from envs.env_register import env_register
    a = copy.deepcopy(args)
    a.task_seed += k
    a.failure_seed += k
    env = env_register[a.env_name](a)
    N = a.edge_node_num
    rng = np.random.RandomState(pol_seed)
    out = {key: [] for key in ("actions", "obs", "avail", "reward", "terminated", "info",
                               "next_obs", "qlen", "hop_keys")}
    snaps = []
    hop_hist = []
    gstep = 0
    for ep in range(episodes):
        env.reset()
        while True:
            obs = env.get_obs()
            avail = env.get_avail_actions()
            if expert is not None:
                # the expert's decision for this state is recorded at EVERY step, whoever acts
                ex = np.asarray(expert.get_current_optimal_action(env), dtype=np.int64)
                direct = np.asarray(expert._conservative_action_selection(env, conservativeness=1.0), dtype=np.int64)
                assert np.array_equal(ex, direct), "the reference fell back to its greedy path"
                out.setdefault("expert_actions", []).append(ex)
                if policy == "expert" or (policy == "mixed" and rng.random_sample() < 0.5):
                    acts = ex
                else:
                    acts = _policy("random_valid", rng, env, avail)
            else:
                acts = _policy(policy, rng, env, avail)
            reward, terminated, info = env.step(acts)
            out["actions"].append(acts)
            out["obs"].append(np.asarray(obs, dtype=np.float64))
            out["avail"].append(np.asarray(avail, dtype=np.int32))
            out["reward"].append(float(reward))
            out["terminated"].append(bool(terminated))
            out["info"].append([info["finish_task_number"], info["success_finish_task_number"],
                                info["drop_task_number"], info["failure_task_number"],
                                info["task_completion_time"]])
            hh = np.zeros(64, dtype=np.int64)
            for h, c in info["max_hop_dict"].items():
                hh[h] += c
            hop_hist.append(hh)
            out["next_obs"].append(np.asarray(env.get_obs(), dtype=np.float64))
            # queue lengths after the step: [node, (new_task, buffer, execution, executing, recv[0..N-1])]
            ql = np.zeros((N, 4 + N), dtype=np.int32)
            for n in env.edge_nodes:
                ql[n.id, 0] = 0 if n.new_task is None else 1
                ql[n.id, 1] = len(n.buffer_queue)
                ql[n.id, 2] = len(n.execution_queue)
                ql[n.id, 3] = len(n.executing_queue)
                for s, q in enumerate(n.receiving_queues):
                    ql[n.id, 4 + s] = len(q)
            out["qlen"].append(ql)
            if gstep % snap_every == snap_every - 1:
                snaps.extend(_snapshot_rows(env, gstep))
            gstep += 1
            if terminated:
                break
    res = {
        "actions": np.asarray(out["actions"], dtype=np.int64),
        "obs": np.asarray(out["obs"]),
        "avail": np.asarray(out["avail"]),
        "reward": np.asarray(out["reward"], dtype=np.float64),
        "terminated": np.asarray(out["terminated"], dtype=np.uint8),
        "info": np.asarray(out["info"], dtype=np.float64),
        "next_obs": np.asarray(out["next_obs"]),
        "qlen": np.asarray(out["qlen"]),
        "hop_hist": np.asarray(hop_hist, dtype=np.int64),
        "snap": np.asarray(snaps, dtype=np.float64).reshape(-1, 12),
    }
    if expert is not None:
        res["expert_actions"] = np.asarray(out["expert_actions"], dtype=np.int64)
    return res


def _static_tables(args):
    """Static per-node tables as the reference builds them (edge_node.py:22-85)."""
    import numpy as np
    from envs.env_register import env_register
    import copy
    env = env_register[args.env_name](copy.deepcopy(args))
    N = args.edge_node_num
    nodes = env.edge_nodes
    adj = np.zeros((N, N + 1), dtype=np.uint8)
    for n in nodes:
        for a in n.actions:
            adj[n.id, a] = 1
    return {
        "edges": np.asarray(env.get_edges(), dtype=np.int64),
        "adj_order": np.asarray([n.actions + [-1] * (N + 1 - len(n.actions)) for n in nodes], dtype=np.int64),
        "adj": adj,
        "cores": np.asarray([int(n.cpu_core_num) for n in nodes], dtype=np.int64),
        "p": np.asarray([n.task_probability for n in nodes], dtype=np.float64),
        "efr": np.asarray([n.execution_failure_rate for n in nodes], dtype=np.float64),
        "cap": np.asarray([float(n.cpu_capacity) for n in nodes], dtype=np.float64),
        "qlen_max": np.asarray([n.execution_queue_len for n in nodes], dtype=np.float64),
        "rate": np.asarray([n.transmission_rates for n in nodes], dtype=np.float64),
        "tfr": np.asarray([n.transmission_failure_rates for n in nodes], dtype=np.float64),
    }


# --------------------------------------------------------------------------
def gen_rng():
    import numpy as np
    out = {}
    seeds = [0, 1, 100, 109, 200, 500, 601, 4295, 2 ** 32 - 1]
    out["seeds"] = np.asarray(seeds, dtype=np.int64)
    out["doubles"] = np.stack([np.random.RandomState(s).random_sample(1500) for s in seeds])
    # generate_task draw pattern (edge_node.py:159-166 + task.py:10-12): binomial(1,p) then 3 doubles on arrival
    ps = [0.0, 0.07474158005205533, 0.5, 0.5592089368350069, 0.9174961169661047, 1.0]
    rows = []
    for s in (100, 4195):
        for p in ps:
            r = np.random.RandomState(s)
            for _ in range(300):
                arr = r.binomial(1, p)
                if arr == 1:
                    tid = r.random()
                    size = r.uniform(1000, 6000) * 1024 * 8
                    cyc = r.uniform(800, 2400) * size
                else:
                    tid = size = cyc = -1.0
                rows.append([s, p, arr, tid, size, cyc])
    out["task_draws"] = np.asarray(rows, dtype=np.float64)
    # node-parameter stream (edge_node.py:31-34,57): choice, uniform, uniform per node
    r = np.random.RandomState(500)
    rows = []
    for _ in range(40):
        c = r.choice([4, 8, 12, 16, 20, 24, 28, 32])
        rows.append([c, r.uniform(0, 1.0), r.uniform(0, 0.3)])
    out["node_param_draws"] = np.asarray(rows, dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "rng.npz"), **out)


def gen_env(tag):
    import numpy as np
    if tag == "env10":
        args = _ref_args(["--task_size_max=6000", "--seed=100"])
        cases = [(0, "random_valid", 3), (1, "random_valid", 3), (2, "random_valid", 2), (7, "random_valid", 2),
                 (4095, "random_valid", 2), (0, "local_first", 2), (3, "local_first", 2),
                 (0, "max_forward", 2), (5, "max_forward", 2), (0, "unmasked", 2), (11, "unmasked", 2)]
    elif tag == "env10d":
        # reference defaults (task_size_max=2000, edge_computing_config.py:14)
        args = _ref_args(["--seed=100"])
        cases = [(0, "random_valid", 2), (9, "unmasked", 2)]
    elif tag == "env40":
        args = _ref_args(["--task_size_max=6000", "--seed=100", "--edge_node_num=40"])
        cases = [(0, "random_valid", 2), (16383, "random_valid", 1), (3, "max_forward", 1), (8, "unmasked", 1)]
    else:
        raise KeyError(tag)
    out = {}
    cfg_keys = ["episode_limit", "mini_time_slot", "mini_time_slot_num", "edge_node_num", "task_probability_min",
                "task_probability_max", "task_size_min", "task_size_max", "task_complexity_min",
                "task_complexity_max", "deadline", "single_core_cpu_capacity", "transmission_rate_min",
                "transmission_rate_max", "execution_failure_rate_min", "execution_failure_rate_max",
                "transmission_failure_rate_min", "transmission_failure_rate_max", "task_drop_penalty",
                "task_failure_penalty", "task_success_reward", "task_hop_penalty", "edge_node_seed", "task_seed",
                "failure_seed", "rl_queue_coeff", "link_seed", "k", "beta"]
    out["cfg_keys"] = np.asarray(cfg_keys)
    out["cfg_vals"] = np.asarray([float(getattr(args, k)) for k in cfg_keys], dtype=np.float64)
    out["cpu_core_list"] = np.asarray(args.cpu_core_list, dtype=np.int64)
    # NOTE: the topology is generated by the FIRST env constructed in the process
    for k, v in _static_tables(args).items():
        out["static_" + k] = v
    names = []
    for ci, (k, pol, eps) in enumerate(cases):
        res = _replay(args, k, pol, eps, snap_every=10, pol_seed=1234 + ci)
        name = "c%02d" % ci
        names.append("%s:%d:%s:%d" % (name, k, pol, eps))
        for kk, vv in res.items():
            if kk in ("obs", "next_obs"):
                pass
            out[name + "_" + kk] = vv
    out["cases"] = np.asarray(names)
    np.savez_compressed(os.path.join(HERE, tag + ".npz"), **out)


def gen_expert(tag):
    import numpy as np
    if tag == "expert10":
        args = _ref_args(["--task_size_max=6000", "--seed=100"])
        cases = [(0, "expert", 2), (5, "expert", 1), (0, "mixed", 2), (17, "mixed", 2)]
    else:
        args = _ref_args(["--task_size_max=6000", "--seed=100", "--edge_node_num=40"])
        cases = [(0, "expert", 1), (2, "mixed", 1)]
    out = {}
    cfg_keys = ["episode_limit", "mini_time_slot", "mini_time_slot_num", "edge_node_num", "task_probability_min",
                "task_probability_max", "task_size_min", "task_size_max", "task_complexity_min",
                "task_complexity_max", "deadline", "single_core_cpu_capacity", "transmission_rate_min",
                "transmission_rate_max", "execution_failure_rate_min", "execution_failure_rate_max",
                "transmission_failure_rate_min", "transmission_failure_rate_max", "task_drop_penalty",
                "task_failure_penalty", "task_success_reward", "task_hop_penalty", "edge_node_seed", "task_seed",
                "failure_seed", "rl_queue_coeff", "link_seed", "k", "beta"]
    out["cfg_keys"] = np.asarray(cfg_keys)
    out["cfg_vals"] = np.asarray([float(getattr(args, k)) for k in cfg_keys], dtype=np.float64)
    out["cpu_core_list"] = np.asarray(args.cpu_core_list, dtype=np.int64)
    for k, v in _static_tables(args).items():
        out["static_" + k] = v
    from optimal.optimal_agent import OptimalAgent
    args.n_agents, args.n_actions, args.obs_shape = args.edge_node_num, args.edge_node_num + 1, 9
    args.state_shape = 9 * args.edge_node_num
    expert = OptimalAgent(args)
    names = []
    for ci, (k, pol, eps) in enumerate(cases):
        res = _replay(args, k, pol, eps, snap_every=10, pol_seed=4321 + ci, expert=expert)
        name = "c%02d" % ci
        names.append("%s:%d:%s:%d" % (name, k, pol, eps))
        for kk in ("actions", "expert_actions", "avail", "reward", "info", "qlen"):
            out[name + "_" + kk] = res[kk]
    out["cases"] = np.asarray(names)
    np.savez_compressed(os.path.join(HERE, tag + ".npz"), **out)


def gen_gae():
    import numpy as np
    import torch as th
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    from utils.advantage_utils import get_gae
    out = {}
    for name, (B, T, N) in {"a": (8, 100, 10), "b": (3, 17, 5)}.items():
        g = th.Generator().manual_seed(7 + B)
        ns = argparse.Namespace(n_threads=B, episode_limit=T, n_agents=N, gamma=0.99, lambda_=0.95, device="cpu")
        r = th.randn(B, T, N, 1, generator=g)
        v = th.randn(B, T, N, 1, generator=g)
        nv = th.randn(B, T, N, 1, generator=g)
        m = (th.rand(B, T, N, 1, generator=g) > 0.1).float()
        m[:, -1] = 0
        adv = get_gae(ns, r, v, nv, m)
        for k, t in dict(r=r, v=v, nv=nv, m=m, adv=adv).items():
            out["%s_%s" % (name, k)] = t.numpy()
    np.savez_compressed(os.path.join(HERE, "gae.npz"), **out)


GROUPS = {"rng": gen_rng, "env10": lambda: gen_env("env10"), "env10d": lambda: gen_env("env10d"),
          "env40": lambda: gen_env("env40"), "gae": gen_gae, "expert10": lambda: gen_expert("expert10"),
          "expert40": lambda: gen_expert("expert40")}



# --------------------------------------------------------------------------
def gen_mlp():
    """Fixtures (5)/(6) of SURVEY 8c from the reference's OWN torch modules: state_dicts + inputs => masked logits /
    softmax (Actor, PolicyNet), GAIL-discriminator probabilities and rewards, diffusion compute_loss for both
    conditions with the (t, noise) the reference drew (replayed from the same torch seed), D_phi and reward."""
    import numpy as np
    import torch as th
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    from rl.policy_gradient_rl.ppo.ppo_network import Actor
    from rl.policy_gradient_rl.dgail.dgail_network import PolicyNet, GAILDiscriminator, DiffusionDiscriminator
    out = {}
    for N in (10, 40):
        A, SA, rows = N + 1, 9 + N + 1, 640 if N == 10 else 333
        tag = "n%d_" % N
        g = th.Generator().manual_seed(1000 + N)
        obs = th.rand(rows, 9, generator=g)
        obs[:, 3] *= 6.0
        obs[:, 8] = th.randint(0, 4, (rows,), generator=g).float()
        avail = (th.rand(rows, A, generator=g) < 0.6).int()
        avail[:, 0] = 1
        actions = th.randint(0, A, (rows,), generator=g)
        ids = th.eye(N)[th.arange(rows) % N]
        out[tag + "obs"], out[tag + "avail"], out[tag + "actions"] = obs.numpy(), avail.numpy(), actions.numpy()
        # --- PPO Actor (ppo_network.py:6-25) and DGAIL PolicyNet (dgail_network.py:6-19)
        th.manual_seed(7 + N)
        ns = argparse.Namespace(obs_shape=9, n_agents=N, n_actions=A, hidden_dim=64, activate_fun="relu",
                                use_orthogonal_init=False)
        for name, net in (("actor", Actor(ns)), ("policy", PolicyNet(9 + N, A, 64))):
            with th.no_grad():
                for p in net.parameters():
                    p.mul_(3.0)
                logits = net(th.cat([obs, ids], dim=-1))
                logits[avail == 0] = -1e10                       # ppo_agent.py:69, dgail_agent.py:122
                probs = th.softmax(logits, dim=-1)
            for k, v in net.state_dict().items():
                out[tag + name + "_sd_" + k] = v.numpy()
            out[tag + name + "_logits"], out[tag + name + "_probs"] = logits.numpy(), probs.numpy()
        # --- GAIL discriminator reward (dgail_network.py:37-50, dgail_agent.py:482-485,501-518)
        sa = th.cat([obs, th.nn.functional.one_hot(actions, A).float()], dim=-1)
        gd = GAILDiscriminator(SA, 64)
        with th.no_grad():
            for p in gd.parameters():
                p.mul_(2.0)
            p_ = gd(sa)
            base = -th.log(1.0 - p_.clamp(1e-8, 1 - 1e-8)).squeeze(-1)
        for k, v in gd.state_dict().items():
            out[tag + "gd_sd_" + k] = v.numpy()
        out[tag + "gd_prob"], out[tag + "gd_base_reward"] = p_.squeeze(-1).numpy(), base.numpy()
        # --- diffusion discriminator (dgail_network.py:53-108, dgail_agent.py:487-499)
        dd = DiffusionDiscriminator(SA, 64)
        with th.no_grad():
            dd.mlp[4].weight.mul_(3.0)
        dd.eval()                                                 # spectral norm: stored u/v, no power iteration
        losses = {}
        for cname, cval, seed in (("expert", 1, 11), ("agent", 0, 12)):
            c = th.full((rows,), cval, dtype=th.long)
            th.manual_seed(seed)                                  # compute_loss draws t then noise (:90-91)
            L = dd.compute_loss(sa, c).detach()
            th.manual_seed(seed)
            t = th.randint(0, 2, (rows,))
            noise = th.randn_like(sa)
            losses[cname] = L
            out[tag + "dd_t_" + cname], out[tag + "dd_noise_" + cname] = t.numpy().astype(np.uint8), noise.numpy()
            out[tag + "dd_loss_" + cname] = L.numpy()
        pe, pa = th.exp(-losses["expert"]), th.exp(-losses["agent"])
        dphi = pe / (pe + pa + 1e-8)
        out[tag + "dd_dphi"], out[tag + "dd_reward"] = dphi.numpy(), (-th.log(1 - dphi + 1e-8)).numpy()
        # effective (spectrally normalised) weights as the module computes them
        out[tag + "dd_w1"], out[tag + "dd_b1"] = dd.mlp[0].weight.detach().numpy(), dd.mlp[0].bias.detach().numpy()
        out[tag + "dd_w2"], out[tag + "dd_b2"] = dd.mlp[2].weight.detach().numpy(), dd.mlp[2].bias.detach().numpy()
        out[tag + "dd_w3"], out[tag + "dd_b3"] = dd.mlp[4].weight.detach().numpy(), dd.mlp[4].bias.detach().numpy()
        out[tag + "dd_time_embed"] = dd.time_embed.weight.detach().numpy()
        out[tag + "dd_cond_embed"] = dd.cond_embed.weight.detach().numpy()
        for k, v in dd.state_dict().items():
            out[tag + "dd_sd_" + k] = v.numpy()
    np.savez_compressed(os.path.join(HERE, "mlp.npz"), **out)


GROUPS["mlp"] = gen_mlp


# --------------------------------------------------------------------------
class _Recorder:
    """Records, in call order, every random draw the reference's learners make inside an update, by wrapping the
    library functions they call: np.random.choice (expert samples), th.randint / th.randn_like (compute_loss,
    dgail_network.py:90-91), th.randperm (the DataLoader's shuffle) and F.dropout (GAIL discriminator)."""

    def __init__(self):
        self.draws = []

    def __enter__(self):
        import numpy as np
        import torch as th
        import torch.nn.functional as F
        self._np_choice, self._randint, self._randn_like = np.random.choice, th.randint, th.randn_like
        self._randperm, self._dropout = th.randperm, F.dropout
        rec = self

        def choice(a, size=None, replace=True, p=None):
            r = rec._np_choice(a, size=size, replace=replace, p=p)
            rec.draws.append(("choice", np.atleast_1d(np.asarray(r)).astype(np.int64)))
            return r

        def randint(*a, **k):
            r = rec._randint(*a, **k)
            rec.draws.append(("t", r.detach().cpu().numpy().astype(np.int64)))
            return r

        def randn_like(*a, **k):
            r = rec._randn_like(*a, **k)
            rec.draws.append(("noise", r.detach().cpu().numpy().copy()))
            return r

        def randperm(*a, **k):
            r = rec._randperm(*a, **k)
            rec.draws.append(("perm", r.detach().cpu().numpy().astype(np.int64)))
            return r

        def dropout(x, p=0.5, training=True, inplace=False):
            if not training or p == 0.0:
                return x
            keep = (th.rand_like(x) >= p)
            rec.draws.append(("dropout", keep.cpu().numpy()))
            return x * keep.to(x.dtype) / (1.0 - p)

        np.random.choice, th.randint, th.randn_like, th.randperm, F.dropout = choice, randint, randn_like, randperm, dropout
        return self

    def __exit__(self, *exc):
        import numpy as np
        import torch as th
        import torch.nn.functional as F
        np.random.choice, th.randint, th.randn_like = self._np_choice, self._randint, self._randn_like
        th.randperm, F.dropout = self._randperm, self._dropout

    def dump(self, out, tag):
        import numpy as np
        out[tag + "draw_kinds"] = np.asarray([k for k, _ in self.draws])
        for i, (_, v) in enumerate(self.draws):
            out[tag + "draw_%04d" % i] = v


def _learn_args(algo, extra):
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    from default_config import add_default_args
    from envs.edge_computing.multihop.edge_computing_config import add_edge_computing_env_args
    from envs.env_register import env_register
    adders = {}
    from rl.policy_gradient_rl.ppo.ppo_config import add_ppo_args
    from rl.policy_gradient_rl.gail.gail_config import add_gail_args
    from rl.policy_gradient_rl.dgail.dgail_config import add_dgail_args
    adders = {"ppo": add_ppo_args, "gail": add_gail_args, "dgail": add_dgail_args}
    p = argparse.ArgumentParser(conflict_handler='resolve')
    p = adders[algo](add_edge_computing_env_args(add_default_args(p)))
    args = p.parse_args(extra)
    args.device = "cpu"
    env = env_register[args.env_name](args)          # main.py:37-43
    info = env.get_env_info()
    args.n_agents, args.n_actions = info["n_agents"], info["n_actions"]
    args.obs_shape, args.state_shape = info["obs_shape"], info["state_shape"]
    return args


def _learn_rollout(args, agent, expert_every=0):
    """One episode of args.n_threads reference envs (env k seeded as parallel_episode_runner.py:19-20) filed into
    agent.buffer; the agent's own select_actions acts.  expert_every > 0: every expert_every-th step of env 0 also
    stores the expert's (obs, action) pair with the agent (gail_runner.py:59-70 / dgail_runner.py:66-80)."""
    import copy
    import numpy as np
    import torch as th
    from envs.env_register import env_register
    B, N = args.n_threads, args.n_agents
    envs = []
    for k in range(B):
        a = copy.deepcopy(args)
        a.task_seed += k
        a.failure_seed += k
        envs.append(env_register[a.env_name](a))
    for e in envs:
        e.reset()
    for e in envs:  # reset leaves no task (edge_node.py:171-177): one idle step gives the episode something to do
        e.step([N] * N)
        e.current_step = 0
    agent.buffer.reset()
    for t in range(args.episode_limit):
        obs = th.as_tensor(np.array([e.get_obs() for e in envs]), dtype=th.float)
        avail = th.as_tensor(np.array([e.get_avail_actions() for e in envs]), dtype=th.int)
        actions = agent.select_actions(obs, avail, False)
        if expert_every and t % expert_every == 0:
            ex = agent.get_expert_actions(envs[0])
            agent.store_expert_data(obs[0], th.as_tensor(np.asarray(ex), dtype=th.long))
        rewards, terms = [], []
        for k, e in enumerate(envs):
            r, term, _ = e.step(actions[k].cpu().numpy())
            rewards.append(r)
            terms.append(term)
        next_obs = th.as_tensor(np.array([e.get_obs() for e in envs]), dtype=th.float)
        rew = th.as_tensor(rewards, dtype=th.float).view(B, 1, 1).expand(B, N, 1)
        masks = (1 - th.as_tensor(terms, dtype=th.float)).view(B, 1, 1).expand(B, N, 1)
        agent.buffer.insert(obs, avail, actions.unsqueeze(-1), rew, masks, next_obs)


def _dump_sd(out, tag, sd):
    for k, v in sd.items():
        out[tag + k] = v.detach().cpu().numpy().copy()


def gen_learn():
    """SURVEY 8c fixture (8): the reference's OWN learners on a fixed buffer.  Initial state_dicts + buffer + expert
    transitions + every random draw made inside the update  =>  post-update state_dicts (and the returned stats)."""
    import numpy as np
    import torch as th
    out = {}
    th.set_num_threads(1)

    # ---- PPOAgent.train (ppo_agent.py:85-134)
    args = _learn_args("ppo", ["--task_size_max=6000", "--n_threads=4", "--episode_limit=20"])
    from rl.policy_gradient_rl.ppo.ppo_agent import PPOAgent
    np.random.seed(5)
    th.manual_seed(5)
    agent = PPOAgent(args)
    _dump_sd(out, "ppo_actor0_", agent.actor.state_dict())
    _dump_sd(out, "ppo_critic0_", agent.critic.state_dict())
    _learn_rollout(args, agent)
    for k in ("obs", "avail_actions", "actions", "rewards", "masks", "next_obs"):
        out["ppo_buf_" + k] = getattr(agent.buffer, k).numpy().copy()
    agent.train()
    _dump_sd(out, "ppo_actor1_", agent.actor.state_dict())
    _dump_sd(out, "ppo_critic1_", agent.critic.state_dict())
    agent.train()  # a second update on the same buffer: Adam moments and the StepLR count carry over
    _dump_sd(out, "ppo_actor2_", agent.actor.state_dict())
    _dump_sd(out, "ppo_critic2_", agent.critic.state_dict())
    out["ppo_cfg"] = np.asarray([args.n_threads, args.episode_limit, args.n_agents], dtype=np.int64)

    # ---- GAILAgent.train (gail_agent.py:128-231)
    args = _learn_args("gail", ["--task_size_max=6000", "--n_threads=2", "--episode_limit=20"])
    from rl.policy_gradient_rl.gail.gail_agent import GAILAgent
    np.random.seed(6)
    th.manual_seed(6)
    agent = GAILAgent(args)
    for name in ("actor", "critic", "discriminator"):
        _dump_sd(out, "gail_%s0_" % name, getattr(agent, name).state_dict())
    _learn_rollout(args, agent, expert_every=1)       # 20 expert transitions (>= 10: the discriminator trains)
    for k in ("obs", "avail_actions", "actions", "rewards", "masks", "next_obs"):
        out["gail_buf_" + k] = getattr(agent.buffer, k).numpy().copy()
    out["gail_expert_obs"] = th.stack(agent.expert_obs_buffer).numpy()
    out["gail_expert_actions"] = th.stack(agent.expert_actions_buffer).numpy()
    agent.performance_history = [0.5, 0.6, 0.9, 0.4, 0.6]   # mean < 0.7: the BC weight grows (:118-126)
    with _Recorder() as rec:
        agent.train()
    rec.dump(out, "gail_")
    for name in ("actor", "critic", "discriminator"):
        _dump_sd(out, "gail_%s1_" % name, getattr(agent, name).state_dict())
    out["gail_bc_weight1"] = np.float64(agent.bc_loss_weight)
    out["gail_cfg"] = np.asarray([args.n_threads, args.episode_limit, args.n_agents], dtype=np.int64)

    # ---- DGAILAgent.train (dgail_agent.py:335-381): BC pretrain, both discriminators, rewards, minibatch PPO
    for tag, extra in (("dgail_", []), ("dgaild_", ["--use_simplified_reward="])):
        args = _learn_args("dgail", ["--task_size_max=6000", "--n_threads=1", "--episode_limit=25", "--batch_size_run=10"]
                           + extra)
        from rl.policy_gradient_rl.dgail.dgail_agent import DGAILAgent
        np.random.seed(7)
        th.manual_seed(7)
        agent = DGAILAgent(args)
        nets = ("policy", "value", "discriminator", "gail_discriminator")
        for name in nets:
            _dump_sd(out, tag + name + "0_", getattr(agent, name).state_dict())
        _learn_rollout(args, agent, expert_every=1)   # 25 expert transitions
        for k in ("obs", "avail_actions", "actions", "rewards", "masks", "next_obs"):
            out[tag + "buf_" + k] = getattr(agent.buffer, k).numpy().copy()
        out[tag + "expert_obs"] = th.stack([t[0] for t in agent.expert_transitions]).numpy()
        out[tag + "expert_actions"] = th.stack([t[1] for t in agent.expert_transitions]).numpy()
        agent._bc_pretrain_steps = 6                  # 700 in the reference (:92): same code, fewer steps
        with _Recorder() as rec:
            stats = agent.train(5000)
        rec.dump(out, tag)
        for name in nets:
            _dump_sd(out, tag + name + "1_", getattr(agent, name).state_dict())
        out[tag + "stat_keys"] = np.asarray(sorted(stats))
        out[tag + "stat_vals"] = np.asarray([float(stats[k]) for k in sorted(stats)], dtype=np.float64)
        out[tag + "use_simplified_reward"] = np.int64(bool(args.use_simplified_reward))
        out[tag + "cfg"] = np.asarray([args.n_threads, args.episode_limit, args.n_agents], dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "learn.npz"), **out)


GROUPS["learn"] = gen_learn


if __name__ == "__main__":
    if len(sys.argv) > 1:
        GROUPS[sys.argv[1]]()
    else:
        for g in GROUPS:
            print("generating", g, flush=True)
            subprocess.check_call([sys.executable, os.path.abspath(__file__), g])
