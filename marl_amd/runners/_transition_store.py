"""Per-step rollout bookkeeping of the runners on ec_store_transition (include/marl_amd.h): the six buffer slice
writes, the team-reward broadcast, the mask and the running episode statistics in one launch.  `usable` says whether
the agent's buffer is the device-resident EpisodeBuffer this needs; otherwise the runner keeps its torch-op path."""
import ctypes as C

import torch as th

from marl_amd import _abi


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class TransitionStore:
    def __init__(self, env, totals):
        self.env = env
        self.totals = totals  # float64 [6] on the env's device

    def usable(self, buf):
        need = ("obs", "avail_actions", "actions", "rewards", "masks", "next_obs")
        if not all(hasattr(buf, k) for k in need):
            return False
        e = self.env
        shapes = {"obs": (e.obs_shape,), "avail_actions": (e.n_actions,), "actions": (1,), "rewards": (1,), "masks": (1,),
                  "next_obs": (e.obs_shape,)}
        dtypes = {"obs": th.float32, "avail_actions": th.int32, "actions": th.int64, "rewards": th.float32,
                  "masks": th.float32, "next_obs": th.float32}
        for k in need:
            t = getattr(buf, k)
            if (not t.is_cuda or t.device != e.device or not t.is_contiguous() or t.dtype != dtypes[k]
                    or tuple(t.shape) != (e.n_envs, e.episode_limit, e.n_agents) + shapes[k]):
                return False
        return True

    def _call(self, t, actions, reward, terminated, info, buf):
        e = self.env
        b = [getattr(buf, k) for k in ("obs", "avail_actions", "actions", "rewards", "masks", "next_obs")] if buf is not None else [None] * 6
        if actions is not None and (actions.dtype != th.int64 or not actions.is_contiguous()):
            actions = actions.to(th.int64).contiguous()
        with th.cuda.device(e.device):
            _abi.check(_abi.lib().ec_store_transition(
                _p(e.obs), _p(e.avail), _p(actions), _p(reward), _p(terminated),
                _p(info["counters"]) if info else _p(None), _p(info["task_completion_time"]) if info else _p(None),
                e.n_envs, e.n_agents, e.obs_shape, e.n_actions, int(t), e.episode_limit, *[_p(x) for x in b],
                _p(self.totals), C.c_void_p(th.cuda.current_stream(e.device).cuda_stream)), "ec_store_transition")

    def episode(self, hist, obs0, avail0, buf):
        """a whole episode from the time-major histories of env.rollout_mlp_policy (ec_store_episode): the buffer
        (buf None: statistics only) and the totals, as episode_limit step() calls would leave them"""
        e = self.env
        b = [getattr(buf, k) for k in ("obs", "avail_actions", "actions", "rewards", "masks", "next_obs")] if buf is not None else [None] * 6
        if getattr(self, "_partial", None) is None:
            self._partial = th.zeros(e.episode_limit, 6, dtype=th.float64, device=e.device)
            self._done = th.zeros(1, dtype=th.int32, device=e.device)
        with th.cuda.device(e.device):
            _abi.check(_abi.lib().ec_store_episode(
                _p(obs0), _p(avail0), _p(hist["obs"]), _p(hist["avail"]), _p(hist["actions"]), _p(hist["reward"]),
                _p(hist["terminated"]), _p(hist["counters"]), _p(hist["completion_time"]), e.n_envs, e.n_agents,
                e.obs_shape, e.n_actions, e.episode_limit, *[_p(x) for x in b], _p(self.totals), _p(self._partial),
                _p(self._done), C.c_void_p(th.cuda.current_stream(e.device).cuda_stream)), "ec_store_episode")
        if buf is not None:
            buf.t = e.episode_limit
            self._mark_whole_episode(buf)

    def _mark_whole_episode(self, buf):
        """the buffer holds one whole episode from reset, filed by the kernels: obs[:, t+1] and next_obs[:, t] are
        copies of one observation, and only the last step terminates (edge_computing_env.py:139-142)"""
        e = self.env
        if hasattr(buf, "next_obs_is_shifted_obs"):
            buf.next_obs_is_shifted_obs = True
            buf.known_mask_sum = float(e.n_envs * e.n_agents * (e.episode_limit - 1))

    def initial(self, buf):
        """after reset + observe: obs[:, 0] / avail_actions[:, 0]"""
        self._call(-1, None, None, None, None, buf)
        self._next_t = 0 if buf is not None else None  # steps filed in sequence since this call

    def step(self, t, actions, reward, terminated, info, buf):
        """after env.step(..., fuse_observe=True): transition t (+ the next step's observation) and the statistics;
        buf None = statistics only (test mode)"""
        self._call(t, actions, reward, terminated, info, buf)
        if buf is not None:
            buf.t = t + 1
            self._next_t = t + 1 if getattr(self, "_next_t", None) == t else None
            if self._next_t == self.env.episode_limit:
                self._mark_whole_episode(buf)
