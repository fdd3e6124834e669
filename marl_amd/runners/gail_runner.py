"""`GAILRunner` on the batched HIP environment: the reference's runners/gail_runner.py:5-116 for args.n_threads
environments in lock step.  While `t_expert < expert_guidance_steps` each env flips a coin with the agent's scheduled
expert ratio (:32-34) and executes the expert's action (ec_expert_policy) or the agent's; the PPO buffer always stores
an action sampled from the agent (:73-77), expert-driven envs also feed the expert store (:97-98); every training
episode ends with `agent.update_performance_monitoring` and `agent.train()` (:109-111).

`args.ec_kernel_rollout` (default on): the whole episode — Actor, expert, coin, env step and next observation,
episode_limit times — is ONE launch (ec_rollout_mlp_expert_policy; ec_rollout_mlp_policy when the episode is not guided),
followed by ec_store_episode and, for guided episodes, ec_expert_append_episode.  The draws of an episode then come from
one torch.rand call each for the categorical draws and the coins."""
import torch as th
import torch.distributed as dist

from marl_amd.envs.env_register import env_register
from marl_amd.runners._transition_store import TransitionStore
from marl_amd.utils.dist_utils import all_reduce_, dist_on


class GAILRunner:
    def __init__(self, args, agent):
        self.args = args
        self.agent = agent
        self.env = env_register["edge_computing_batched"](args, n_envs=args.n_threads,
                                                          env_id_base=getattr(args, "ec_env_id_base", 0))
        self.t_env = 0
        B, N = args.n_threads, args.n_agents
        dev = self.env.device
        self._obs = th.empty(B, N, args.obs_shape, dtype=th.float, device=dev)
        self._avail = th.empty(B, N, args.n_actions, dtype=th.int, device=dev)
        self._totals = th.zeros(6, dtype=th.float64, device=dev)
        self._store = TransitionStore(self.env, self._totals)
        self._ratio = th.zeros((), dtype=th.float32, device=dev)   # the episode's expert ratio, on the device
        self._graphs = {}
        self._hist = None
        self._uniforms = self._coin = self._ratios = None
        self._geometry = None

    def _kernel_rollout_ok(self, guided):
        ag = self.agent
        if not getattr(self.args, "ec_kernel_rollout", 1) or self.env.device.type != "cuda":
            return False
        if not (self._store.usable(ag.buffer) and hasattr(ag, "_fused_ok") and ag._fused_ok()):
            return False
        if self._geometry is None:
            self._geometry = self.env.info_geometry()
        return self._geometry["rollout_mlp_expert_policy" if guided else "rollout_mlp_policy"]

    def _episode_kernel(self, test_mode, guided):
        """the episode as two or three launches: ec_rollout_mlp[_expert]_policy, ec_store_episode[, ec_expert_append_episode]"""
        from marl_amd.utils import mlp_device
        args, env = self.args, self.env
        B, N, T = args.n_threads, args.n_agents, args.episode_limit
        if self._uniforms is None:
            self._uniforms = th.empty(T, B * N, dtype=th.float, device=env.device)
            self._coin = th.empty(T, B, dtype=th.float, device=env.device)
            self._ratios = th.empty(T, dtype=th.float, device=env.device)
        if not test_mode:
            self._uniforms.uniform_()
        if guided:
            self._coin.uniform_()
            self._ratios.copy_(self._ratio.expand(T))
        self._hist = env.rollout_mlp_policy(mlp_device.linear_params(self.agent.actor), T,
                                            uniforms=None if test_mode else self._uniforms,   # greedy in test mode (:46-59)
                                            history=self._hist, want_hops=True, coin=self._coin if guided else None,
                                            ratios=self._ratios if guided else None)
        # env.obs / env.avail still hold the observation from before the first step
        self._store.episode(self._hist, env.obs, env.avail, None if test_mode else self.agent.buffer)
        if guided:
            self.agent.expert_transitions.append_episode(self._coin, self._ratios, self._hist["expert_actions"], env.obs,
                                                         self._hist["obs"])

    def run(self, test_mode=False, t_expert=0):
        args, env = self.args, self.env
        B, T = args.n_threads, args.episode_limit
        guided = (not test_mode) and t_expert < args.expert_guidance_steps
        if guided:
            self._ratio.fill_(float(self.agent.get_expert_ratio(t_expert)))
        # --ec_graph_rollout: the first episode of each kind runs eagerly (it warms every kernel in place), the second
        # is recorded, and all from the second on replay the hipGraph
        key = (bool(test_mode), bool(guided))
        graphed = (getattr(args, "ec_graph_rollout", False) and env.device.type == "cuda"
                   and self._store.usable(self.agent.buffer) and not self._kernel_rollout_ok(guided))
        if graphed and key in self._graphs:
            if self._graphs[key] is None:
                g = th.cuda.CUDAGraph()
                th.cuda.synchronize(env.device)
                with th.cuda.graph(g):
                    self._episode(test_mode, guided)
                self._graphs[key] = g
            self._graphs[key].replay()
            if not test_mode:
                self.agent.buffer.t = T
        else:
            self._episode(test_mode, guided)
            if graphed:
                self._graphs[key] = None           # warmed: record at the next call
        if not test_mode:
            self.t_env += B * T
        return self._finish(test_mode)

    def _episode(self, test_mode, guided):
        args, env = self.args, self.env
        B, N = args.n_threads, args.n_agents
        kernel = self._kernel_rollout_ok(guided)
        if kernel and not test_mode:
            self.agent.buffer.begin_episode()  # ec_store_episode overwrites every element
        else:
            self.agent.buffer.reset()
        env.reset()
        env.observe()
        self._totals.zero_()
        if kernel:
            return self._episode_kernel(test_mode, guided)
        fast = self._store.usable(self.agent.buffer)
        buf = None if test_mode else self.agent.buffer
        if fast and buf is not None:
            self._store.initial(buf)
        for t in range(args.episode_limit):
            if fast:
                obs, avail = env.obs, env.avail          # read in place; ec_step replaces them with the next observation
            else:
                self._obs.copy_(env.obs)
                self._avail.copy_(env.avail)
                obs, avail = self._obs, self._avail
            agent_actions = self.agent.select_actions(obs, avail, test_mode)
            if guided and env.device.type == "cuda":
                # coin, select and FIFO append in two launches (ec_expert_mix) instead of ~17
                exec_actions = self.agent.expert_transitions.mix_append(
                    th.rand(B, device=env.device), self._ratio.reshape(1), self.agent.get_expert_actions(env), agent_actions,
                    obs)
            elif guided:
                use_expert = th.rand(B, device=env.device) < self._ratio
                expert_actions = self.agent.get_expert_actions(env)
                exec_actions = th.where(use_expert.unsqueeze(-1), expert_actions, agent_actions)
                self.agent.store_expert_data(obs, expert_actions, use_expert)     # :97-98 (obs is the pre-step one)
            else:
                exec_actions = agent_actions
            reward, terminated, info = env.step(exec_actions, want_hops=(t == args.episode_limit - 1))
            if fast:
                self._store.step(t, agent_actions, reward, terminated, info, buf)
                continue
            self._totals[0] += reward.double().sum()
            self._totals[1] += info["task_completion_time"].double().sum()
            self._totals[2:6] += info["counters"].double().sum(0)
            if not test_mode:
                rewards = reward.view(B, 1, 1).expand(B, N, 1)
                masks = (1 - terminated.float()).view(B, 1, 1).expand(B, N, 1)
                self.agent.buffer.insert(obs, avail, agent_actions.unsqueeze(-1), rewards, masks, env.obs)

    def _finish(self, test_mode):
        args, env = self.args, self.env
        B = args.n_threads
        env.check_status()
        totals = self._totals.clone()
        n_envs = th.tensor([float(B)], dtype=th.float64, device=totals.device)
        hops = env.hop_hist.double().sum(0)
        if dist_on():
            all_reduce_(totals)
            all_reduce_(n_envs)
            all_reduce_(hops)
        ret, ctime, fin, succ, drop, fail = totals.tolist()
        episode_info = {
            "episode_return": ret / n_envs.item(),
            "success_rate": succ / max(fin, 1),
            "drop_rate": drop / max(fin, 1),
            "failure_rate": fail / max(fin, 1),
            "task_completion_time": ctime / max(succ + drop, 1),
            "max_hop_dict": {h: int(c) for h, c in enumerate(hops.tolist()) if c},
        }
        if not test_mode:
            self.agent.update_performance_monitoring(episode_info["success_rate"])
            self.agent.train()
        return episode_info

    def close_env(self):
        self.env.close()
