"""Pins the CPU oracle (oracle/ec_oracle.c) and the host-side table builder against the golden
vectors produced by importing the reference (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from marl_amd._abi import config_from_args
from marl_amd.envs.edge_computing.multihop.network_graph import build_static_tables
from tests.util import args_from_golden, golden_cases, load_golden


def test_rng_doubles_match_randomstate(oracle_lib):
    g = load_golden("rng")
    for seed, want in zip(g["seeds"], g["doubles"]):
        got = oracle_lib.rng_doubles(int(seed), want.shape[0])
        assert np.array_equal(got, want), seed


def test_rng_generate_task_pattern(oracle_lib):
    """binomial(1,p) + id/size/cycles draws, edge_node.py:159-166, task.py:10-12"""
    g = load_golden("rng")["task_draws"]
    keys = sorted({(int(r[0]), float(r[1])) for r in g})
    for seed, p in keys:
        want = g[(g[:, 0] == seed) & (g[:, 1] == p)][:, 2:]
        got = oracle_lib.rng_task_draws(seed, p, want.shape[0], 1000, 6000, 800, 2400)
        assert np.array_equal(got, want), (seed, p)


@pytest.mark.parametrize("name", ["env10", "env10d", "env40"])
def test_static_tables(name):
    """network_graph.py:219-264, links.py:7-9, edge_node.py:22-64"""
    g = load_golden(name)
    args = args_from_golden(g)
    t = build_static_tables(args)
    assert np.array_equal(np.asarray(t.unique_edges()), g["static_edges"])
    assert np.array_equal(t.cores, g["static_cores"])
    for a, b in (("p", "p"), ("efr", "efr"), ("cap", "cap"), ("qlen_max", "qlen_max"), ("rate", "rate"),
                 ("tfr", "tfr"), ("adj", "adj")):
        assert np.array_equal(getattr(t, a), g["static_" + b]), a
    N = t.n_nodes
    order = [nb + [i] + [-1] * (N - len(nb)) for i, nb in enumerate(t.neighbours)]
    assert np.array_equal(np.asarray(order), g["static_adj_order"])


def _replay_case(oracle_lib, g, args, tables, name, k):
    cfg = config_from_args(args, hop_hist_len=64)
    env = oracle_lib.OracleEnv(cfg, tables, 1, env_id_base=k)
    actions = g[name + "_actions"]
    T = actions.shape[0]
    snap = g[name + "_snap"]
    N = cfg.n_nodes
    for t in range(T):
        if t % cfg.episode_limit == 0:
            env.reset()
        obs, avail = env.observe()
        assert np.array_equal(obs[0], g[name + "_obs"][t]), (name, t, "obs")
        assert np.array_equal(avail[0], g[name + "_avail"][t]), (name, t, "avail")
        out = env.step(actions[t])
        assert out["bad"] == 0
        info = g[name + "_info"][t]
        assert out["reward"][0] == g[name + "_reward"][t], (name, t, "reward")
        assert out["terminated"][0] == g[name + "_terminated"][t]
        assert np.array_equal(out["counters"][0], info[:4].astype(np.int32)), (name, t, "counters")
        assert out["completion_time"][0] == info[4], (name, t, "completion_time")
        assert np.array_equal(out["hop_hist"][0], g[name + "_hop_hist"][t]), (name, t, "hop_hist")
        nobs, _ = env.observe()
        assert np.array_equal(nobs[0], g[name + "_next_obs"][t]), (name, t, "next_obs")
        rows = env.dump(0)
        ql = np.zeros((N, 4 + N), dtype=np.int32)
        for r in rows:
            ql[int(r[0]), int(r[1])] += 1
        assert np.array_equal(ql, g[name + "_qlen"][t]), (name, t, "qlen")
        want = snap[snap[:, 0] == t][:, 1:]
        if want.shape[0]:
            assert np.array_equal(rows, want), (name, t, "snapshot")
    env.close()


@pytest.mark.parametrize("name", ["env10", "env10d", "env40"])
def test_action_replay_matches_reference(oracle_lib, name):
    """Whole trajectories: obs, avail, reward, info, termination, queue lengths, per-task snapshots."""
    g = load_golden(name)
    args = args_from_golden(g)
    tables = build_static_tables(args)
    for case, k, pol, eps in golden_cases(g):
        _replay_case(oracle_lib, g, args, tables, case, k)


def test_gae_matches_reference(oracle_lib):
    """utils/advantage_utils.py:4-9 (fp32, bit-exact)"""
    g = load_golden("gae")
    for tag in ("a", "b"):
        adv = oracle_lib.gae(g[tag + "_r"], g[tag + "_v"], g[tag + "_nv"], g[tag + "_m"], 0.99, 0.95)
        assert np.array_equal(adv, g[tag + "_adv"][..., 0]), tag


@pytest.mark.parametrize("name", ["expert10", "expert40"])
def test_expert_policy_matches_optimal_agent(oracle_lib, name):
    """oc_expert_policy vs OptimalAgent.get_current_optimal_action (optimal_agent.py:102-120,166-375) recorded at
    every step of expert-driven and expert/random-mixed trajectories."""
    g = load_golden(name)
    args = args_from_golden(g)
    tables = build_static_tables(args)
    cfg = config_from_args(args, hop_hist_len=64)
    for case, k, pol, eps in golden_cases(g):
        env = oracle_lib.OracleEnv(cfg, tables, 1, env_id_base=k)
        actions, want = g[case + "_actions"], g[case + "_expert_actions"]
        n_remote = 0
        for t in range(actions.shape[0]):
            if t % cfg.episode_limit == 0:
                env.reset()
            _, avail = env.observe()
            assert np.array_equal(avail[0], g[case + "_avail"][t]), (case, t, "avail")
            got = env.expert_policy(avail)[0]
            assert np.array_equal(got, want[t]), (case, pol, t, got, want[t])
            n_remote += int(((got != np.arange(cfg.n_nodes)) & (got != cfg.n_nodes)).sum())
            out = env.step(actions[t])
            assert out["reward"][0] == g[case + "_reward"][t], (case, t, "reward")
        assert n_remote > 50  # the fixture exercises the remote-scoring branch
        env.close()
