"""Device-resident FIFO of expert (observation, action) transitions.

The reference keeps a Python list of per-step CPU tensor pairs (obs [N,9], actions [N]) capped at
`max_expert_buffer_size`, dropping the oldest (dgail_agent.py:135-149, gail_agent.py:77-83).  Here the same FIFO is a
ring of two device tensors; a rollout step appends the transitions of every env whose step was expert-driven without a
host round trip (rows that are not selected are scattered to a spare slot past the ring).
"""
import torch as th

from marl_amd.utils.random_source import RandomSource


class ExpertTransitions:
    def __init__(self, capacity, n_agents, obs_shape, device):
        self.capacity = int(capacity)
        self.device = th.device(device)
        self.obs = th.zeros(self.capacity + 1, n_agents, obs_shape, dtype=th.float32, device=self.device)
        self.actions = th.zeros(self.capacity + 1, n_agents, dtype=th.int64, device=self.device)
        self._written = th.zeros((), dtype=th.int64, device=self.device)  # transitions ever appended
        self.rng = RandomSource(self.device)  # the learners share it (tests replace it to replay recorded draws)

    def __len__(self):
        """len(self.expert_transitions) of the reference (synchronises)."""
        return int(min(int(self._written.item()), self.capacity))

    def global_len(self):
        """len() as the ranks agree on it: the minimum over ranks.  Branches that lead into collective code (train or
        skip, discriminator step or not) must be taken by every rank alike, whatever its own expert coin flips stored;
        with no process group this is len()."""
        import torch.distributed as dist
        n = len(self)
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            from marl_amd.utils.dist_utils import all_reduce_
            n = int(all_reduce_(th.tensor([n], dtype=th.int64, device=self.device), dist.ReduceOp.MIN).item())
        return n

    def append(self, obs, actions, select=None):
        """obs [B,N,O], actions [B,N]; select: optional bool [B] (envs whose transition is stored)."""
        B = obs.shape[0]
        if select is None:
            select = th.ones(B, dtype=th.bool, device=self.device)
        sel = select.to(th.int64)
        rank = th.cumsum(sel, 0) - 1
        n_sel = sel.sum()
        # One call can select more rows than the ring holds (B > max_expert_buffer_size): the FIFO then keeps the
        # newest `capacity` of them, as appending them one by one would.  Older selected rows go to the spare slot
        # with the unselected ones, so every ring slot has exactly one source row and (obs, action) stay paired.
        keep = select & (rank >= n_sel - self.capacity)
        slot = th.where(keep, (self._written + rank) % self.capacity, th.full_like(rank, self.capacity))
        self.obs.index_copy_(0, slot, obs.to(self.device, th.float32))
        self.actions.index_copy_(0, slot, actions.to(self.device, th.int64))
        self._written += n_sel

    def mix_append(self, coin, ratio, expert_actions, agent_actions, obs, n_expert=None):
        """One step of the guided runners on the device (ec_expert_mix): env b executes the expert's action iff
        coin[b] < ratio (a device scalar) and then appends (obs[b], expert_actions[b]) as append() would; returns the
        actions to execute [B,N]; n_expert (device int64 scalar, optional) += the number of expert-driven envs."""
        import ctypes as C

        from marl_amd import _abi
        B, N, O = obs.shape
        expert_actions = expert_actions.to(th.int64).contiguous()
        agent_actions = agent_actions.to(th.int64).contiguous()
        obs = obs.to(th.float32).contiguous()
        coin = coin.to(th.float32).contiguous()
        out = th.empty_like(agent_actions)
        vp = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
        with th.cuda.device(self.device):
            _abi.check(_abi.lib().ec_expert_mix(vp(coin), vp(ratio), vp(expert_actions), vp(agent_actions), vp(obs), B, N, O,
                                                self.capacity, vp(out), vp(self.obs), vp(self.actions), vp(self._written),
                                                vp(n_expert), C.c_void_p(th.cuda.current_stream(self.device).cuda_stream)),
                       "ec_expert_mix")
        return out

    def append_episode(self, coin, ratios, expert_actions, obs0, obs_hist, n_expert=None):
        """The appends of a whole episode of env.rollout_mlp_policy(coin=, ratios=) (ec_expert_append_episode): for every
        step t and env b with coin[t, b] < ratios[t], in (t, b) order, (the observation step t acted on, expert_actions[t, b])
        — what mix_append / append leave after being called once per step.  coin [T,B], ratios [T], expert_actions [T,B,N],
        obs0 [B,N,O], obs_hist [T,B,N,O] (observation after step t)."""
        import ctypes as C

        from marl_amd import _abi
        T, B, N = expert_actions.shape
        O = obs0.shape[-1]
        lib = _abi.lib()
        need = int(lib.ec_expert_append_episode_workspace(T, B))
        if getattr(self, "_ws", None) is None or self._ws.numel() < need:
            self._ws = th.empty(max(need, 4), dtype=th.uint8, device=self.device)
        for name, t, dtype in (("coin", coin, th.float32), ("ratios", ratios, th.float32), ("obs0", obs0, th.float32),
                               ("expert_actions", expert_actions, th.int64), ("obs_hist", obs_hist, th.float32)):
            if t.dtype != dtype or not t.is_contiguous() or t.device != self.obs.device:
                raise ValueError("%s must be a contiguous %s tensor on %s" % (name, dtype, self.obs.device))
        if tuple(coin.shape) != (T, B) or tuple(ratios.shape) != (T,) or tuple(obs0.shape) != (B, N, O) \
                or tuple(obs_hist.shape) != (T, B, N, O) or tuple(self.obs.shape[1:]) != (N, O):
            raise ValueError("append_episode: shapes do not fit together")
        vp = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
        with th.cuda.device(self.device):
            _abi.check(lib.ec_expert_append_episode(vp(coin), vp(ratios), vp(expert_actions), vp(obs0), vp(obs_hist), T, B, N, O,
                                                    self.capacity, vp(self.obs), vp(self.actions), vp(self._written),
                                                    vp(n_expert), vp(self._ws),
                                                    C.c_void_p(th.cuda.current_stream(self.device).cuda_stream)),
                       "ec_expert_append_episode")

    def _share(self, obs, actions):
        """data parallel: every rank takes rank 0's sample (the update is then the one-process update whatever
        the number of ranks; the stores differ per rank, so the rows travel, not their indices)"""
        from marl_amd.utils.dist_utils import broadcast_, dist_on
        if dist_on():
            obs, actions = obs.contiguous(), actions.contiguous()
            broadcast_(obs, 0)
            broadcast_(actions, 0)
        return obs, actions

    def sample_transitions(self, n, size=None, shared=False):
        """n distinct transitions, uniformly (np.random.choice(len, n, replace=False) of the reference):
        returns obs [n,N,O], actions [n,N].  `size`: len(self) when the caller already has it (len() synchronises).
        shared: under torch.distributed every rank gets rank 0's sample (n is cut to the smallest store)."""
        from marl_amd.utils.dist_utils import dist_on
        size = len(self) if size is None else int(size)
        n = min(int(n), self.global_len() if (shared and dist_on()) else size)
        idx = self.rng.choice_without_replacement(size, n)
        out = self.obs[idx], self.actions[idx]
        return self._share(*out) if shared else out

    def sample_transitions_many(self, n, count, shared=False):
        """`count` consecutive sample_transitions(n) draws in one go: obs [count,n,N,O], actions [count,n,N] (one size
        query, one batched draw, one gather)."""
        from marl_amd.utils.dist_utils import dist_on
        size = len(self)
        n = min(int(n), self.global_len() if (shared and dist_on()) else size)
        idx = self.rng.choice_without_replacement_many(size, n, int(count))
        out = self.obs[idx], self.actions[idx]
        return self._share(*out) if shared else out

    def sample_rows(self, n):
        """n agent-rows with replacement over the concatenation of all stored transitions
        (dgail_agent.py:209-216): returns obs [n,O], actions [n]."""
        size = len(self)
        N = self.obs.shape[1]
        idx = self.rng.choice_with_replacement(size * N, int(n))
        return self.obs[:size].reshape(size * N, -1)[idx], self.actions[:size].reshape(-1)[idx]

    def state_dict(self):
        """The reference's expert_data.th layout (gail_agent.py:244-248): two Python lists of per-step CPU tensors
        (obs [N,O], actions [N]), oldest first, so that the reference's GAILAgent.load_models can take the file and go on
        appending / popping (:77-83).  The ring is unrolled from its oldest slot."""
        size = len(self)
        start = int(self._written.item()) % self.capacity if size == self.capacity else 0
        order = (th.arange(size, device=self.device) + start) % max(size, 1)
        obs, act = self.obs[order].cpu(), self.actions[order].cpu()
        return {"expert_obs": list(obs.unbind(0)), "expert_actions": list(act.unbind(0))}

    def load_state_dict(self, sd):
        obs, act = sd["expert_obs"], sd["expert_actions"]
        if isinstance(obs, (list, tuple)):  # the reference's list-of-tensors checkpoint format
            obs, act = th.stack(list(obs)), th.stack(list(act))
        n = min(obs.shape[0], self.capacity)
        self.obs[:n].copy_(obs[-n:])
        self.actions[:n].copy_(act[-n:])
        self._written.fill_(n)
