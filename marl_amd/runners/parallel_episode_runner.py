"""`ParallelRunner` on the batched HIP environment.

Same contract as the reference's runners/parallel_episode_runner.py:9-127 —
`ParallelRunner(args, agent)`, `.run(test_mode) -> episode_info`, `.t_env`, `.close_env()` — but the
n_threads worker processes, their pipes and the four pickle round-trips per env per step are replaced by ONE
device-resident batch: per step the policy acts on device tensors, `ec_step` advances all environments and
writes the next observation in the same launch, and the transition is copied device-to-device into the
agent's buffer.  Env e of the batch is the reference's worker e (seeds task_seed+e, failure_seed+e, :19-20).

Multi-GPU: each rank constructs the runner with its own `args.ec_env_id_base` shard; episode metrics are
summed over ranks so that `episode_info` is the whole job's.

`args.ec_kernel_rollout` (default on): with the PPO agent the whole episode — Actor forward, mask, softmax, draw, env
step and next observation, episode_limit times — is ONE launch of the looping env kernel (ec_rollout_mlp_policy),
followed by one launch that files the episode into the agent's buffer (ec_store_episode).  The uniforms of the
categorical draws come from one torch.rand call per episode.  Falls back to the per-step launches when the env shape
is outside what that kernel is compiled for.

`args.ec_graph_rollout = True` captures one whole episode (reset, episode_limit x [policy, step, statistics, buffer
insert]) in a hipGraph on first use and replays it afterwards: the rollout is launch-bound (~20 small launches per
step), and replaying removes the host from the loop.  Kernels are warmed on a throw-away one-env instance first, so
the captured environments' RNG streams are untouched and trajectories equal the eager ones.
"""
import torch as th
import torch.distributed as dist

from marl_amd.utils.dist_utils import all_reduce_

from marl_amd.envs.env_register import env_register
from marl_amd.runners._transition_store import TransitionStore


def _dist_on():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


class ParallelRunner:
    def __init__(self, args, agent):
        self.args = args
        self.agent = agent
        env_cls = env_register["edge_computing_batched"]
        self.env = env_cls(args, n_envs=args.n_threads, env_id_base=getattr(args, "ec_env_id_base", 0))
        self.t_env = 0
        B, N = args.n_threads, args.n_agents
        dev = self.env.device
        self._obs = th.empty(B, N, args.obs_shape, dtype=th.float, device=dev)
        self._avail = th.empty(B, N, args.n_actions, dtype=th.int, device=dev)
        self._totals = th.zeros(6, dtype=th.float64, device=dev)
        self._graphs = {}
        self._store = TransitionStore(self.env, self._totals)
        self._hist = None
        self._uniforms = None

    def _select(self, env, obs, avail, test_mode):
        if self.args.algo_type == "rl":
            return self.agent.select_actions(obs, avail, test_mode)
        return self.agent.select_actions(env)

    def _capture(self, test_mode):
        """Warm every kernel of the episode on a one-env scratch instance, then capture the episode."""
        args = self.args
        scratch = type(self.env)(args, n_envs=1, env_id_base=0, device=self.env.device)
        side = th.cuda.Stream(device=self.env.device)
        side.wait_stream(th.cuda.current_stream(self.env.device))
        with th.cuda.stream(side):
            scratch.reset()
            scratch.observe()
            for hops in (False, True):
                a = self._select(scratch, scratch.obs.clone(), scratch.avail.clone(), test_mode)
                scratch.step(a, want_hops=hops)
        th.cuda.current_stream(self.env.device).wait_stream(side)
        th.cuda.synchronize(self.env.device)
        scratch.close()
        g = th.cuda.CUDAGraph()
        with th.cuda.graph(g):
            self._episode(test_mode)
        return g

    def run(self, test_mode=False):
        args, env = self.args, self.env
        B = args.n_threads
        kernel = self._store.usable(self.agent.buffer) and self._kernel_rollout_ok()  # two launches: nothing to replay
        if getattr(args, "ec_graph_rollout", False) and not kernel:
            g = self._graphs.get(bool(test_mode))
            if g is None:
                g = self._graphs[bool(test_mode)] = self._capture(test_mode)
            g.replay()
            if not test_mode:
                self.agent.buffer.t = args.episode_limit
        else:
            self._episode(test_mode)
        if not test_mode:
            self.t_env += B * args.episode_limit
        return self._finish(test_mode)

    def _kernel_rollout_ok(self):
        ag = self.agent
        if not getattr(self.args, "ec_kernel_rollout", 1) or self.args.algo_type != "rl":
            return False
        if not (hasattr(ag, "actor") and hasattr(ag, "_fused_policy_ok") and ag._fused_policy_ok()):
            return False
        if getattr(self, "_kernel_ok", None) is None:
            self._kernel_ok = self.env.info_geometry()["rollout_mlp_policy"]
        return self._kernel_ok

    def _episode_kernel(self, test_mode):
        """the episode as two launches: ec_rollout_mlp_policy, ec_store_episode"""
        from marl_amd.utils import mlp_device
        args, env = self.args, self.env
        T = args.episode_limit
        if self._uniforms is None:
            self._uniforms = th.empty(T, args.n_threads * args.n_agents, dtype=th.float, device=env.device)
        self._uniforms.uniform_()  # the reference's PPO always samples (ppo_agent.py:62); torch's device generator
        self._hist = env.rollout_mlp_policy(mlp_device.linear_params(self.agent.actor), T, uniforms=self._uniforms,
                                            history=self._hist, want_hops=True)
        # env.obs / env.avail still hold the observation from before the first step
        self._store.episode(self._hist, env.obs, env.avail, None if test_mode else self.agent.buffer)

    def _episode(self, test_mode):
        args, env = self.args, self.env
        B, N = args.n_threads, args.n_agents
        fast = self._store.usable(self.agent.buffer)
        kernel = fast and self._kernel_rollout_ok()
        if kernel and not test_mode:
            self.agent.buffer.begin_episode()  # ec_store_episode overwrites every element: no need to zero 540 MB first
        else:
            self.agent.buffer.reset()
        env.reset()
        env.observe()
        self._totals.zero_()
        buf = None if test_mode else self.agent.buffer
        if kernel:
            return self._episode_kernel(test_mode)
        if fast and buf is not None:
            self._store.initial(buf)
        # all environments of the batch terminate together at episode_limit (edge_computing_env.py:139-142)
        for t in range(args.episode_limit):
            if fast:
                # the policy reads the env's observation in place; ec_step overwrites it with the next one, which
                # one launch then files as next_obs[t] / obs[t+1] together with the rest of the transition
                actions = self._select(env, env.obs, env.avail, test_mode)
                reward, terminated, info = env.step(actions, want_hops=(t == args.episode_limit - 1))
                self._store.step(t, actions, reward, terminated, info, buf)
                continue
            self._obs.copy_(env.obs)
            self._avail.copy_(env.avail)
            actions = self._select(env, self._obs, self._avail, test_mode)
            reward, terminated, info = env.step(actions, want_hops=(t == args.episode_limit - 1))
            # accumulate the episode statistics on device (parallel_episode_runner.py:94-99)
            self._totals[0] += reward.double().sum()
            self._totals[1] += info["task_completion_time"].double().sum()
            self._totals[2:6] += info["counters"].double().sum(0)
            if not test_mode:
                rewards = reward.view(B, 1, 1).expand(B, N, 1)                     # team reward broadcast, :101
                masks = (1 - terminated.float()).view(B, 1, 1).expand(B, N, 1)    # :102
                self.agent.buffer.insert(self._obs, self._avail, actions.unsqueeze(-1), rewards, masks, env.obs)

    def _finish(self, test_mode):
        args, env = self.args, self.env
        B = args.n_threads
        env.check_status()
        totals = self._totals.clone()
        n_envs = th.tensor([float(B)], dtype=th.float64, device=totals.device)
        hops = env.hop_hist.double().sum(0)
        if _dist_on():
            all_reduce_(totals)
            all_reduce_(n_envs)
            all_reduce_(hops)
        ret, ctime, fin, succ, drop, fail = totals.tolist()
        episode_info = {
            "episode_return": ret / n_envs.item(),
            "success_rate": succ / fin if fin else 0.0,
            "drop_rate": drop / fin if fin else 0.0,
            "failure_rate": fail / fin if fin else 0.0,
            "task_completion_time": ctime / (succ + drop) if (succ + drop) else 0.0,
            # the reference's parallel runner omits this key although run_utils.test_models reads it
            # (run_utils.py:308); cumulative, as edge_node.py:82 never clears it
            "max_hop_dict": {h: int(c) for h, c in enumerate(hops.tolist()) if c},
        }
        if not test_mode:
            self.agent.train()
        return episode_info

    def close_env(self):
        self.env.close()
