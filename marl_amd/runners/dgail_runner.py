"""`DGAILRunner` on the batched HIP environment.

Same contract as the reference's runners/dgail_runner.py:5-156 — `DGAILRunner(args, agent)`,
`.run(test_mode) -> ep_info`, `.t_env`, `.close_env()` — for args.n_threads environments in lock step (the reference
runs one).  Per step and per env a coin with the scheduled expert ratio (:30-38,46) decides whether the env executes the
expert's action (ec_expert_policy on the device state) or the agent's; the agent's action is what the PPO buffer
stores either way (:81), and expert-driven envs append (obs, expert action) to the agent's expert store (:93-94).
`t_env` counts env steps summed over the batch, as ParallelRunner does (parallel_episode_runner.py:103-104).

`args.ec_kernel_rollout` (default on): the whole episode — policy, expert, coin, env step and next observation,
episode_limit times — is ONE launch (ec_rollout_mlp_expert_policy; ec_rollout_mlp_policy, greedy, in test mode), followed
by ec_store_episode and ec_expert_append_episode.  The draws of an episode then come from one torch.rand call each for
the categorical draws and the coins.
"""
import torch as th
import torch.distributed as dist

from marl_amd.envs.env_register import env_register
from marl_amd.runners._transition_store import TransitionStore
from marl_amd.utils.dist_utils import all_reduce_, dist_on


class DGAILRunner:
    def __init__(self, args, agent):
        self.args = args
        self.agent = agent
        self.env = env_register["edge_computing_batched"](args, n_envs=args.n_threads,
                                                          env_id_base=getattr(args, "ec_env_id_base", 0))
        self.t_env = 0
        self.episode_count = 0
        self.bc_guidance_steps = getattr(args, "bc_guidance_steps", 2_000)
        self.expert_guidance_steps = getattr(args, "expert_guidance_steps", 6_000)
        B, N = args.n_threads, args.n_agents
        dev = self.env.device
        self._obs = th.empty(B, N, args.obs_shape, dtype=th.float, device=dev)
        self._avail = th.empty(B, N, args.n_actions, dtype=th.int, device=dev)
        self._totals = th.zeros(6, dtype=th.float64, device=dev)
        self.expert_steps = 0  # expert-driven env steps of the last episode
        self._store = TransitionStore(self.env, self._totals)
        # per-step expert ratios of the running episode and the count of expert-driven env steps, on the device: the
        # episode below has no host-dependent values in it, so that it can be replayed from a hipGraph
        self._ratios = th.ones(args.episode_limit, dtype=th.float32, device=dev)
        self._n_expert = th.zeros((), dtype=th.int64, device=dev)
        self._graphs = {}
        self._hist = None
        self._uniforms = self._coin = None
        self._geometry = None

    def _kernel_rollout_ok(self, test_mode):
        ag = self.agent
        if not getattr(self.args, "ec_kernel_rollout", 1) or self.env.device.type != "cuda":
            return False
        if not (self._store.usable(ag.buffer) and hasattr(ag, "_fused_ok") and ag._fused_ok()):
            return False
        if self._geometry is None:
            self._geometry = self.env.info_geometry()
        return self._geometry["rollout_mlp_policy" if test_mode else "rollout_mlp_expert_policy"]

    def _episode_kernel(self, test_mode):
        """the episode as two or three launches: ec_rollout_mlp[_expert]_policy, ec_store_episode[, ec_expert_append_episode]"""
        from marl_amd.utils import mlp_device
        args, env = self.args, self.env
        B, N, T = args.n_threads, args.n_agents, args.episode_limit
        if self._uniforms is None:
            self._uniforms = th.empty(T, B * N, dtype=th.float, device=env.device)
            self._coin = th.empty(T, B, dtype=th.float, device=env.device)
        if not test_mode:
            self._uniforms.uniform_()
            self._coin.uniform_()
        guided = not test_mode
        self._hist = env.rollout_mlp_policy(mlp_device.linear_params(self.agent.policy), T,
                                            uniforms=None if test_mode else self._uniforms,   # greedy in test mode, :114-126
                                            history=self._hist, want_hops=True, coin=self._coin if guided else None,
                                            ratios=self._ratios if guided else None)
        # env.obs / env.avail still hold the observation from before the first step
        self._store.episode(self._hist, env.obs, env.avail, None if test_mode else self.agent.buffer)
        if guided:
            self.agent.expert_transitions.append_episode(self._coin, self._ratios, self._hist["expert_actions"], env.obs,
                                                         self._hist["obs"], self._n_expert)

    def _ratio_at(self, t_env):
        """dgail_runner.py:30-38"""
        if t_env < self.bc_guidance_steps:
            return 1.0
        if t_env < self.expert_guidance_steps:
            progress = (t_env - self.bc_guidance_steps) / (self.expert_guidance_steps - self.bc_guidance_steps)
            return max(0.2, 0.9 * (1 - progress))
        return 0.2

    def _expert_ratio(self):
        return self._ratio_at(self.t_env)

    def run(self, test_mode=False):
        args, env = self.args, self.env
        B, T = args.n_threads, args.episode_limit
        if not test_mode:   # t_env advances by B per step (dgail_runner.py:46 reads it before each step)
            self._ratios.copy_(th.tensor([self._ratio_at(self.t_env + t * B) for t in range(T)], dtype=th.float32))
        # --ec_graph_rollout: the first episode of each kind runs eagerly (it warms every kernel in place), the second
        # is recorded, and all from the second on replay the hipGraph
        key = bool(test_mode)
        graphed = (getattr(args, "ec_graph_rollout", False) and env.device.type == "cuda"
                   and self._store.usable(self.agent.buffer) and not self._kernel_rollout_ok(test_mode))
        if graphed and key in self._graphs:
            if self._graphs[key] is None:
                g = th.cuda.CUDAGraph()
                th.cuda.synchronize(env.device)
                with th.cuda.graph(g):
                    self._episode(test_mode)
                self._graphs[key] = g
            self._graphs[key].replay()
            if not test_mode:
                self.agent.buffer.t = T
        else:
            self._episode(test_mode)
            if graphed:
                self._graphs[key] = None           # warmed: record at the next call
        if not test_mode:
            self.t_env += B * T
        return self._finish(test_mode, T)

    def _episode(self, test_mode):
        args, env = self.args, self.env
        B, N = args.n_threads, args.n_agents
        kernel = self._kernel_rollout_ok(test_mode)
        if kernel and not test_mode:
            self.agent.buffer.begin_episode()  # ec_store_episode overwrites every element
        else:
            self.agent.buffer.reset()
        env.reset()
        env.observe()
        self._totals.zero_()
        self._n_expert.zero_()
        if kernel:
            return self._episode_kernel(test_mode)
        n_expert = self._n_expert
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
            if test_mode:
                exec_actions = agent_actions
            elif env.device.type == "cuda":
                # the coin (dgail_runner.py:46), the action select and the FIFO append (:93-94, before the step: obs is
                # the pre-step one) in two launches (ec_expert_mix) instead of ~17
                exec_actions = self.agent.expert_transitions.mix_append(
                    th.rand(B, device=env.device), self._ratios[t:t + 1], self.agent.get_expert_actions(env), agent_actions,
                    obs, n_expert)
            else:
                use_expert = th.rand(B, device=env.device) < self._ratios[t]   # bool [B]
                expert_actions = self.agent.get_expert_actions(env)
                exec_actions = th.where(use_expert.unsqueeze(-1), expert_actions, agent_actions)
                n_expert += use_expert.sum()
                self.agent.store_expert_data(obs, expert_actions, use_expert)
            reward, terminated, info = env.step(exec_actions, want_hops=(t == args.episode_limit - 1))
            if fast:
                self._store.step(t, agent_actions, reward, terminated, info, buf)   # the buffer keeps the AGENT's action, :81
                continue
            self._totals[0] += reward.double().sum()
            self._totals[1] += info["task_completion_time"].double().sum()
            self._totals[2:6] += info["counters"].double().sum(0)
            if not test_mode:
                rewards = reward.view(B, 1, 1).expand(B, N, 1)
                masks = (1 - terminated.float()).view(B, 1, 1).expand(B, N, 1)
                self.agent.buffer.insert(obs, avail, agent_actions.unsqueeze(-1), rewards, masks, env.obs)

    def _finish(self, test_mode, steps):
        args, env = self.args, self.env
        B = args.n_threads
        n_expert = self._n_expert
        env.check_status()
        totals = self._totals.clone()
        n_envs = th.tensor([float(B)], dtype=th.float64, device=totals.device)
        hops = env.hop_hist.double().sum(0)
        if dist_on():
            all_reduce_(totals)
            all_reduce_(n_envs)
            all_reduce_(hops)
        ret, ctime, fin, succ, drop, fail = totals.tolist()
        self.expert_steps = int(n_expert.item())
        finish_cnt = max(fin, 1)
        success_base = max(succ + drop, 1)
        ep_info = {
            "episode_return": ret / n_envs.item(),        # per env, :116
            "success_rate": succ / finish_cnt,
            "drop_rate": drop / finish_cnt,
            "failure_rate": fail / finish_cnt,
            "task_completion_time": ctime / success_base,
            "max_hop_dict": {h: int(c) for h, c in enumerate(hops.tolist()) if c},
            "episode_steps": steps,
            # (the ranks' minimum under torch.distributed: the train-or-skip decision below leads into collectives)
            "expert_buffer_size": self.agent.expert_transitions.global_len(),
        }
        if not test_mode and ep_info["expert_buffer_size"] > args.batch_size_run:   # :127-131
            train_stats = self.agent.train(self.t_env)
            if train_stats:
                ep_info.update(train_stats)
            self.agent.update_performance_monitoring(ep_info["success_rate"])
        self.episode_count += 1
        if not test_mode and self.episode_count % 10 == 0:
            self._print_detailed_stats(ep_info)
        return ep_info

    def _print_detailed_stats(self, info):
        if dist_on() and dist.get_rank() != 0:
            return
        print("\n--- Episode %d | Env Steps %d ---" % (self.episode_count, self.t_env))
        print("  success rate: %.4f | Episode return: %.2f" % (info.get("success_rate", 0), info.get("episode_return", 0)))
        print("  expert buffer size: %d" % info.get("expert_buffer_size", 0))
        print("  loss of discriminator: %.4f | discriminator output -> expert: %.4f, agent: %.4f"
              % (info.get("discriminator_loss", -1), info.get("prob_expert", -1), info.get("prob_agent", -1)))
        print("  imitation_reward: %.4f" % info.get("avg_imitation_reward", 0))
        print("-" * 50)

    def close_env(self):
        self.env.close()
