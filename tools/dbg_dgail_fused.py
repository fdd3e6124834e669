import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch as th
from tests.test_learners_ref import _args, _sd, _fill, _draws
from tests.util import load_golden
import marl_amd.rl.policy_gradient_rl.dgail.dgail_agent as da
from marl_amd.utils.random_source import ReplaySource
g = load_golden("learn")
tag = "dgail_"
res = {}
for mode in ("fused", "torch"):
    a = _args("dgail", g[tag + "cfg"], "cuda", ["--batch_size_run=10"])
    a.ec_torch_update = 1 if mode == "torch" else 0
    a.ec_graph_update = 0
    agent = da.DGAILAgent(a)
    for name in ("policy", "value", "discriminator", "gail_discriminator"):
        getattr(agent, name).load_state_dict(_sd(g, tag + name + "0_"))
    _fill(agent.buffer, g, tag, agent.device)
    e_obs, e_act = th.from_numpy(g[tag + "expert_obs"]), th.from_numpy(g[tag + "expert_actions"])
    for i in range(e_obs.shape[0]):
        agent.store_expert_data(e_obs[i:i + 1].cuda(), e_act[i:i + 1].cuda())
    agent._bc_pretrain_steps = 6
    draws, seen = [], 0
    for k, v in _draws(g, tag):
        if k == "perm":
            seen += 1
            if seen % 2 == 0:
                continue
        draws.append((k, v))
    agent.rng = agent.expert_transitions.rng = ReplaySource(agent.device, draws)
    log = []
    if mode == "fused":
        orig = agent._fused_step
        def step(tables, idx, bc, bc_w, totals):
            before = totals.clone()
            orig(tables, idx, bc, bc_w, totals)
            log.append((totals - before).cpu().numpy().copy())
        agent._fused_step = step
    else:
        orig = agent._minibatch_step
        def step(pi, act_f, lp, adv, ret, idx, bc, bc_w, totals):
            before = totals.clone()
            orig(pi, act_f, lp, adv, ret, idx, bc, bc_w, totals)
            log.append((totals - before).cpu().numpy().copy())
        agent._minibatch_step = step
    stats = agent.train(5000)
    res[mode] = (np.array(log), stats)
f, t = res["fused"][0], res["torch"][0]
for i in range(len(f)):
    print(i, f[i], t[i], f[i] - t[i])
print(res["fused"][1]["policy_loss"], res["torch"][1]["policy_loss"])
