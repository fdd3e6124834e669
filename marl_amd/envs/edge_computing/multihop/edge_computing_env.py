"""The edge_computing environment on the HIP library.

`BatchedEdgeComputingEnv`  B = args.n_threads environments stepped by one kernel launch; every output is
                           a device tensor.  Env e of the batch is the reference env that
                           ParallelRunner builds for worker (ec_env_id_base + e)
                           (runners/parallel_episode_runner.py:17-21): seeds task_seed+e, failure_seed+e.
`EdgeComputingEnv`         the reference's single-env contract (envs/edge_computing/multihop/
                           edge_computing_env.py:7-160): Python float reward, bool terminated, info dict,
                           numpy observations — a batch of one.

State lives in HBM inside the library handle; this file only owns output tensors and the handle.
"""
import ctypes as C

import numpy as np
import torch as th

from marl_amd import _abi
from marl_amd.envs.edge_computing.multihop.network_graph import get_static_tables
from marl_amd.envs.multiagentenv import MultiAgentEnv

OBS_SHAPE = 9  # edge_node.py:74

INFO_KEYS = ("finish_task_number", "success_finish_task_number", "drop_task_number", "failure_task_number")


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class BatchedEdgeComputingEnv(MultiAgentEnv):
    def __init__(self, args, n_envs=None, env_id_base=None, device=None):
        if not th.cuda.is_available():
            raise RuntimeError("BatchedEdgeComputingEnv needs a ROCm device: marl_amd has no CPU fallback")
        self.lib = _abi.lib()
        self.n_envs = int(n_envs if n_envs is not None else getattr(args, "n_threads", 1))
        self.env_id_base = int(env_id_base if env_id_base is not None else getattr(args, "ec_env_id_base", 0))
        self.device = th.device(device if device is not None else "cuda:%d" % th.cuda.current_device())
        self.tables = get_static_tables(args)
        self.hop_hist_len = int(getattr(args, "ec_hop_hist_len", 32))
        self.cfg = _abi.config_from_args(args, hop_hist_len=self.hop_hist_len)
        topo, self._keep = _abi.topology_from_tables(self.tables)
        self._h = C.c_void_p()
        dev_index = self.device.index if self.device.index is not None else th.cuda.current_device()
        _abi.check(self.lib.ec_create(C.byref(self.cfg), C.byref(topo), self.n_envs, self.env_id_base, dev_index,
                                      C.byref(self._h)), "ec_create")

        # edge_computing_env.py:24-30
        args.n_actions = args.edge_node_num + 1
        args.n_agents = args.edge_node_num
        self.args = args
        self.n_agents = args.edge_node_num
        self.n_actions = args.edge_node_num + 1
        self.episode_limit = args.episode_limit
        self.obs_shape = OBS_SHAPE
        self.state_shape = OBS_SHAPE * self.n_agents

        B, N, A = self.n_envs, self.n_agents, self.n_actions
        dev = self.device
        self.obs = th.empty(B, N, OBS_SHAPE, dtype=th.float32, device=dev)
        self.avail = th.empty(B, N, A, dtype=th.int32, device=dev)
        self.reward = th.empty(B, dtype=th.float32, device=dev)
        self.terminated = th.empty(B, dtype=th.uint8, device=dev)
        self.counters = th.empty(B, 4, dtype=th.int32, device=dev)
        self.completion_time = th.empty(B, dtype=th.float32, device=dev)
        self.hop_hist = th.zeros(B, self.hop_hist_len, dtype=th.int32, device=dev)
        self._obs_valid = False

    # ------------------------------------------------------------------ lifetime
    def close(self):
        if getattr(self, "_h", None):
            self.lib.ec_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return C.c_void_p(th.cuda.current_stream(self.device).cuda_stream)

    # ------------------------------------------------------------------ batched API (device tensors)
    def reset(self):
        _abi.check(self.lib.ec_reset(self._h, self._stream()), "ec_reset")
        self._obs_valid = False

    def observe(self, obs=None, avail=None):
        """Fill (and return) obs [B,N,9] f32 and avail [B,N,A] i32 for the current state."""
        obs = self.obs if obs is None else obs
        avail = self.avail if avail is None else avail
        _abi.check(self.lib.ec_observe(self._h, _ptr(obs), _ptr(avail), self._stream()), "ec_observe")
        if obs is self.obs and avail is self.avail:
            self._obs_valid = True
        return obs, avail

    def get_obs(self):
        if not self._obs_valid:
            self.observe()
        return self.obs

    def get_avail_actions(self):
        if not self._obs_valid:
            self.observe()
        return self.avail

    def step_random_policy(self, actions, seed, counter=0, next_actions=None, want_hops=False):
        """step() + the uniform-random policy for the next step in the same launch (ec_step_random_policy): the draw
        equals random_actions(seed, counter + steps taken by this env object so far, this one included).
        next_actions defaults to `actions` itself (in place).  Returns (reward, terminated, info, next_actions)."""
        if actions.dtype != th.int64 or not actions.is_contiguous() or actions.device != self.device:
            raise ValueError("actions must be a contiguous int64 tensor on the env's device")
        nxt = actions if next_actions is None else next_actions
        _abi.check(self.lib.ec_step_random_policy(self._h, _ptr(actions), _ptr(self.reward), _ptr(self.terminated),
                                                  _ptr(self.counters), _ptr(self.completion_time),
                                                  _ptr(self.hop_hist) if want_hops else None, _ptr(self.obs),
                                                  _ptr(self.avail), int(seed), int(counter), _ptr(nxt),
                                                  self._stream()), "ec_step_random_policy")
        self._obs_valid = True
        info = {"counters": self.counters, "task_completion_time": self.completion_time}
        if want_hops:
            info["hop_hist"] = self.hop_hist
        return self.reward, self.terminated, info, nxt

    def rollout_random_policy(self, actions, n_steps, seed, counter=0, history=None, want_hops=False):
        """n_steps calls of step_random_policy in one launch (ec_rollout_random_policy), same results.
        history=None: in place (self.reward, self.obs, ... and `actions` hold what the last step left).
        history=dict: per-step slices are written to its tensors, any subset of reward [K,B] f32, terminated [K,B]
        u8, counters [K,B,4] i32, completion_time [K,B] f32, obs [K,B,N,9] f32 plus the required avail [K,B,N,A]
        i32 and actions [K,B,N] i64 (actions[t] is the action taken at step t+1).  Returns `history` or, in place,
        (reward, terminated, info, actions)."""
        if actions.dtype != th.int64 or not actions.is_contiguous() or actions.device != self.device:
            raise ValueError("actions must be a contiguous int64 tensor on the env's device")
        n_steps = int(n_steps)
        hops = _ptr(self.hop_hist) if want_hops else None
        if history is None:
            _abi.check(self.lib.ec_rollout_random_policy(
                self._h, n_steps, 0, _ptr(actions), _ptr(self.reward), _ptr(self.terminated), _ptr(self.counters),
                _ptr(self.completion_time), hops, _ptr(self.obs), _ptr(self.avail), int(seed), int(counter),
                _ptr(actions), self._stream()), "ec_rollout_random_policy")
            self._obs_valid = True
            info = {"counters": self.counters, "task_completion_time": self.completion_time}
            if want_hops:
                info["hop_hist"] = self.hop_hist
            return self.reward, self.terminated, info, actions
        B, N = self.n_envs, self.n_agents
        shapes = {"reward": ((n_steps, B), th.float32), "terminated": ((n_steps, B), th.uint8),
                  "counters": ((n_steps, B, 4), th.int32), "completion_time": ((n_steps, B), th.float32),
                  "obs": ((n_steps, B, N, 9), th.float32), "avail": ((n_steps, B, N, N + 1), th.int32),
                  "actions": ((n_steps, B, N), th.int64)}
        ptr = {}
        for key, (shape, dtype) in shapes.items():
            t = history.get(key)
            if t is None:
                if key in ("avail", "actions"):
                    raise ValueError("history needs '%s'" % key)
                ptr[key] = None
                continue
            if tuple(t.shape) != shape or t.dtype != dtype or not t.is_contiguous() or t.device != self.device:
                raise ValueError("history['%s'] must be a contiguous %s tensor of shape %s on the env's device" % (key, dtype, shape))
            ptr[key] = _ptr(t)
        _abi.check(self.lib.ec_rollout_random_policy(
            self._h, n_steps, 1, _ptr(actions), ptr["reward"], ptr["terminated"], ptr["counters"],
            ptr["completion_time"], hops, ptr["obs"], ptr["avail"], int(seed), int(counter), ptr["actions"],
            self._stream()), "ec_rollout_random_policy")
        self._obs_valid = False  # self.obs / self.avail were not written: observe() again before reading them
        return history

    def rollout_mlp_policy(self, params, n_steps, uniforms=None, history=None, want_hops=False, coin=None, ratios=None):
        """n_steps x [Actor forward + mask + softmax + draw on the current observation, env step, next observation] in
        one launch (ec_rollout_mlp_policy): what n_steps rounds of mlp_device.policy_forward + step leave behind,
        bit for bit.  params = mlp_device.linear_params(actor); uniforms [n_steps, B*N] f32 (None: greedy).  The
        current observation is taken from observe() if self.obs is stale.  history: dict of [n_steps, ...] tensors
        as in rollout_random_policy (missing ones are allocated; actions[t] is the action taken AT step t).
        coin [n_steps, B] f32 + ratios [n_steps] f32 (device): the expert-guided rollout of the GAIL / DGAIL runners
        (ec_rollout_mlp_expert_policy) — env b executes the expert's decision at step t iff coin[t, b] < ratios[t];
        history["actions"] then holds the agent's draws and history["expert_actions"] the expert's decisions.
        Returns history."""
        n_steps = int(n_steps)
        B, N = self.n_envs, self.n_agents
        if not self._obs_valid:
            self.observe()
        w = [t if (t.dtype == th.float32 and t.is_contiguous()) else t.detach().to(th.float32).contiguous() for t in params]
        if any(t.device != self.device for t in w):
            raise ValueError("policy parameters must be on the env's device")
        if tuple(w[0].shape) != (64, 9 + N) or tuple(w[2].shape) != (64, 64) or tuple(w[4].shape) != (N + 1, 64):
            raise ValueError("rollout_mlp_policy expects the 9+N -> 64 -> 64 -> N+1 Actor")
        if uniforms is not None:
            if (tuple(uniforms.shape) != (n_steps, B * N) or uniforms.dtype != th.float32 or not uniforms.is_contiguous()
                    or uniforms.device != self.device):
                raise ValueError("uniforms must be a contiguous float32 [n_steps, B*N] tensor on the env's device")
        shapes = {"reward": ((n_steps, B), th.float32), "terminated": ((n_steps, B), th.uint8),
                  "counters": ((n_steps, B, 4), th.int32), "completion_time": ((n_steps, B), th.float32),
                  "obs": ((n_steps, B, N, 9), th.float32), "avail": ((n_steps, B, N, N + 1), th.int32),
                  "actions": ((n_steps, B, N), th.int64)}
        if (coin is None) != (ratios is None):
            raise ValueError("coin and ratios go together")
        if coin is not None:
            for name, t, shape in (("coin", coin, (n_steps, B)), ("ratios", ratios, (n_steps,))):
                if tuple(t.shape) != shape or t.dtype != th.float32 or not t.is_contiguous() or t.device != self.device:
                    raise ValueError("%s must be a contiguous float32 %s tensor on the env's device" % (name, shape))
            shapes["expert_actions"] = ((n_steps, B, N), th.int64)
        history = {} if history is None else history
        for key, (shape, dtype) in shapes.items():
            t = history.get(key)
            if t is None:
                history[key] = th.empty(shape, dtype=dtype, device=self.device)
            elif tuple(t.shape) != shape or t.dtype != dtype or not t.is_contiguous() or t.device != self.device:
                raise ValueError("history['%s'] must be a contiguous %s tensor of shape %s on the env's device" % (key, dtype, shape))
        head = [self._h, n_steps, *[_ptr(t) for t in w], _ptr(uniforms) if uniforms is not None else None, _ptr(self.obs),
                _ptr(self.avail)]
        tail = [_ptr(history["reward"]), _ptr(history["terminated"]), _ptr(history["counters"]),
                _ptr(history["completion_time"]), _ptr(self.hop_hist) if want_hops else None, _ptr(history["obs"]),
                _ptr(history["avail"]), self._stream()]
        if coin is None:
            _abi.check(self.lib.ec_rollout_mlp_policy(*head, _ptr(history["actions"]), *tail), "ec_rollout_mlp_policy")
        else:
            _abi.check(self.lib.ec_rollout_mlp_expert_policy(*head, _ptr(coin), _ptr(ratios), _ptr(history["actions"]),
                                                             _ptr(history["expert_actions"]), *tail),
                       "ec_rollout_mlp_expert_policy")
        self._obs_valid = False  # self.obs / self.avail still hold the observation from before the rollout
        return history

    def step(self, actions, want_hops=False, fuse_observe=True):
        """actions: int64 [B,N] device tensor.  Returns (reward[B] f32, terminated[B] u8, info) where info holds
        device tensors: the four counters of edge_computing_env.py:130-135 as columns of `counters`,
        `task_completion_time`, and (if want_hops) the cumulative `max_hop_dict` histogram.  With
        fuse_observe the post-step observation lands in self.obs / self.avail in the same launch."""
        if actions.dtype != th.int64 or not actions.is_contiguous() or actions.device != self.device:
            actions = actions.to(device=self.device, dtype=th.int64).contiguous()
        if actions.numel() != self.n_envs * self.n_agents:
            raise ValueError("actions must have %d x %d elements" % (self.n_envs, self.n_agents))
        _abi.check(self.lib.ec_step(self._h, _ptr(actions), _ptr(self.reward), _ptr(self.terminated),
                                    _ptr(self.counters), _ptr(self.completion_time),
                                    _ptr(self.hop_hist) if want_hops else None,
                                    _ptr(self.obs) if fuse_observe else None,
                                    _ptr(self.avail) if fuse_observe else None, self._stream()), "ec_step")
        self._obs_valid = bool(fuse_observe)
        info = {"counters": self.counters, "task_completion_time": self.completion_time}
        if want_hops:
            info["hop_hist"] = self.hop_hist
        return self.reward, self.terminated, info

    def random_actions(self, seed, counter, out=None, counter_dev=None):
        """Uniform over the current availability mask (config 2 of BASELINE.json).  counter_dev: optional
        zero-initialised int64[2] device tensor that advances once per launch (for hipGraph replays)."""
        avail = self.get_avail_actions()
        out = out if out is not None else th.empty(self.n_envs, self.n_agents, dtype=th.int64, device=self.device)
        _abi.check(self.lib.ec_random_policy(_ptr(avail), self.n_envs * self.n_agents, self.n_actions, int(seed),
                                             int(counter), _ptr(counter_dev), _ptr(out), self._stream()),
                   "ec_random_policy")
        return out

    def expert_actions(self, out=None):
        """OptimalAgent.get_current_optimal_action (optimal/optimal_agent.py:102-120) for every env of the batch,
        on the current device state; [B, N] int64."""
        avail = self.get_avail_actions()
        out = out if out is not None else th.empty(self.n_envs, self.n_agents, dtype=th.int64, device=self.device)
        _abi.check(self.lib.ec_expert_policy(self._h, _ptr(avail), _ptr(out), self._stream()), "ec_expert_policy")
        return out

    def live_tasks(self):
        n = C.c_int64()
        _abi.check(self.lib.ec_debug_live_tasks(self._h, C.byref(n)), "ec_debug_live_tasks")
        return int(n.value)

    def status(self, clear=True):
        st = C.c_uint32(0)
        _abi.check(self.lib.ec_status(self._h, C.byref(st), int(clear), self._stream()), "ec_status")
        return st.value

    def check_status(self):
        """Raise what the reference would have raised (or report a capacity overflow)."""
        st = self.status()
        if st & _abi.EC_ST_BAD_ACTION:
            raise ZeroDivisionError("float division by zero (action targets a non-adjacent node, edge_node.py:227)")
        if st & ~_abi.EC_ST_HOP_OVERFLOW:
            raise _abi.EcError("device status 0x%x: queue/tick capacity exceeded (raise --ec_cap_* / --ec_max_ticks)" % st)
        return st

    def info_geometry(self):
        v = (C.c_int64 * 19)()
        _abi.check(self.lib.ec_info(self._h, v, 19), "ec_info")
        keys = ("B", "N", "k", "D", "cap_buffer", "cap_recv", "cap_wait", "max_ticks", "threads_per_block",
                "envs_per_block")
        out = {k: int(v[i]) for i, k in enumerate(keys)}
        out["state_bytes"] = int(v[10]) | (int(v[11]) << 32)
        out["lds_bytes_per_block"] = int(v[12])
        out["envs_per_wave"], out["main_waves"], out["rng_producer_waves"] = int(v[13]), int(v[14]), int(v[15])
        out["rollout_kernel_resident"] = bool(v[16])
        out["rollout_mlp_policy"] = bool(v[17])  # ec_rollout_mlp_policy is compiled for this k / in-degree / n_actions
        out["rollout_mlp_expert_policy"] = bool(v[18])
        return out

    def debug_dump(self, env, max_rows=8192):
        rows = np.empty((max_rows, 11), dtype=np.float64)
        n = C.c_int64()
        _abi.check(self.lib.ec_debug_dump(self._h, int(env), rows.ctypes.data_as(C.c_void_p), max_rows, C.byref(n)),
                   "ec_debug_dump")
        if n.value > max_rows:
            return self.debug_dump(env, max_rows=int(n.value))
        return rows[: n.value].copy()

    def debug_rng_cursors(self):
        cur = np.empty((self.n_envs, self.n_agents, 2), dtype=np.uint64)
        _abi.check(self.lib.ec_debug_rng_cursors(self._h, cur.ctypes.data_as(C.c_void_p)), "ec_debug_rng_cursors")
        return cur

    # ------------------------------------------------------------------ shape queries (multiagentenv.py)
    def get_obs_size(self):
        return self.obs_shape

    def get_state_size(self):
        return self.state_shape

    def get_total_actions(self):
        return self.n_actions

    def get_state(self):
        return self.get_obs().reshape(self.n_envs, -1)

    def get_edges(self):
        return np.array(self.tables.unique_edges())

    def get_obs_llm(self, env=0):
        """EdgeComputingEnv.get_obs_llm (edge_computing_env.py:43-48) for one env of the batch: per node the dict of
        EdgeNode.get_observation_llm (edge_node.py:138-157).  Host-side and synchronous (it serialises the env through
        ec_debug_dump): this is the LLM-prompt path of step_runner / optimal_runner, not the rollout."""
        self.check_status()
        rows = self.debug_dump(env)
        t, a = self.tables, self.args
        out = []
        for i in range(self.n_agents):
            mine = rows[rows[:, 0] == i]
            waiting = mine[mine[:, 1] == 2]
            waiting = waiting[np.argsort(waiting[:, 2])]
            w = 0.0
            for et in waiting[:, 3]:  # get_waiting_time (edge_node.py:99-104): execute_time - 0, queue order
                w += float(et)
            new = mine[mine[:, 1] == 0]
            rates = [float(v) for v in t.rate[i]]
            o = {"node_id": i, "cpu_capacity": int(t.cores[i]),
                 "cpu_utilization": round(len(waiting) / float(t.qlen_max[i]), 4),
                 "execution_failure_rate": round(float(t.efr[i]), 4), "waiting_time": round(w, 4)}
            if len(new):
                o["task_size"] = float(new[0, 3]) / (1024 * 8)
                o["task_cpu_cycle"] = float(new[0, 4]) / (a.task_size_max * 1024 * 8)
            else:
                o["task_size"] = 0
                o["task_cpu_cycle"] = 0
            o["transmission_rate"] = [round(v / max(rates), 4) for v in rates]
            o["actions_space"] = list(t.neighbours[i]) + [i]
            out.append(o)
        return out

    def get_env_info(self):
        return {"state_shape": self.state_shape, "obs_shape": self.obs_shape, "n_actions": self.n_actions,
                "n_agents": self.n_agents, "episode_limit": self.episode_limit}


class EdgeComputingEnv(BatchedEdgeComputingEnv):
    """A batch of one with the reference's Python-level return types."""

    def __init__(self, args):
        super().__init__(args, n_envs=1, env_id_base=getattr(args, "ec_env_id_base", 0))

    def get_obs(self):
        return super().get_obs()[0].double().cpu().numpy()

    def get_obs_agent(self, agent_id):
        return self.get_obs()[agent_id]

    def get_state(self):
        return self.get_obs().reshape(-1)

    def get_avail_actions(self):
        return super().get_avail_actions()[0].cpu().tolist()

    def get_avail_agent_actions(self, agent_id):
        return self.get_avail_actions()[agent_id]

    def step(self, actions):
        a = th.as_tensor(np.asarray(actions), dtype=th.int64, device=self.device).reshape(1, self.n_agents)
        super().step(a, want_hops=True)
        self.check_status()
        c = self.counters[0].cpu().tolist()
        hops = self.hop_hist[0].cpu().tolist()
        info = {k: int(v) for k, v in zip(INFO_KEYS, c)}
        info["task_completion_time"] = float(self.completion_time[0])
        info["max_hop_dict"] = {h: n for h, n in enumerate(hops) if n}
        return float(self.reward[0]), bool(self.terminated[0]), info
