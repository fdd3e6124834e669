"""Device-resident FIFO of expert (observation, action) transitions.

The reference keeps a Python list of per-step CPU tensor pairs (obs [N,9], actions [N]) capped at
`max_expert_buffer_size`, dropping the oldest (dgail_agent.py:135-149, gail_agent.py:77-83).  Here the same FIFO is a
ring of two device tensors; a rollout step appends the transitions of every env whose step was expert-driven without a
host round trip (rows that are not selected are scattered to a spare slot past the ring).
"""
import torch as th


class ExpertTransitions:
    def __init__(self, capacity, n_agents, obs_shape, device):
        self.capacity = int(capacity)
        self.device = th.device(device)
        self.obs = th.zeros(self.capacity + 1, n_agents, obs_shape, dtype=th.float32, device=self.device)
        self.actions = th.zeros(self.capacity + 1, n_agents, dtype=th.int64, device=self.device)
        self._written = th.zeros((), dtype=th.int64, device=self.device)  # transitions ever appended

    def __len__(self):
        """len(self.expert_transitions) of the reference (synchronises)."""
        return int(min(int(self._written.item()), self.capacity))

    def append(self, obs, actions, select=None):
        """obs [B,N,O], actions [B,N]; select: optional bool [B] (envs whose transition is stored)."""
        B = obs.shape[0]
        if select is None:
            select = th.ones(B, dtype=th.bool, device=self.device)
        rank = th.cumsum(select.to(th.int64), 0) - 1
        slot = th.where(select, (self._written + rank) % self.capacity, th.full_like(rank, self.capacity))
        self.obs.index_copy_(0, slot, obs.to(self.device, th.float32))
        self.actions.index_copy_(0, slot, actions.to(self.device, th.int64))
        self._written += select.sum()

    def sample_transitions(self, n, generator=None):
        """n distinct transitions, uniformly (np.random.choice(len, n, replace=False) of the reference):
        returns obs [n,N,O], actions [n,N]."""
        size = len(self)
        n = min(int(n), size)
        if n * 4 <= size:
            # a uniformly random n-subset in random order = the n largest of `size` i.i.d. keys: two small kernels
            # instead of the sort behind randperm (a dozen launches, once per minibatch step of the learners)
            idx = th.rand(size, device=self.device, generator=generator).topk(n).indices
        else:
            idx = th.randperm(size, device=self.device, generator=generator)[:n]
        return self.obs[idx], self.actions[idx]

    def sample_rows(self, n, generator=None):
        """n agent-rows with replacement over the concatenation of all stored transitions
        (dgail_agent.py:209-216): returns obs [n,O], actions [n]."""
        size = len(self)
        N = self.obs.shape[1]
        idx = th.randint(0, size * N, (int(n),), device=self.device, generator=generator)
        return self.obs[:size].reshape(size * N, -1)[idx], self.actions[:size].reshape(-1)[idx]

    def state_dict(self):
        size = len(self)
        return {"expert_obs": self.obs[:size].cpu(), "expert_actions": self.actions[:size].cpu()}

    def load_state_dict(self, sd):
        obs, act = sd["expert_obs"], sd["expert_actions"]
        if isinstance(obs, (list, tuple)):  # the reference's list-of-tensors checkpoint format
            obs, act = th.stack(list(obs)), th.stack(list(act))
        n = min(obs.shape[0], self.capacity)
        self.obs[:n].copy_(obs[-n:])
        self.actions[:n].copy_(act[-n:])
        self._written.fill_(n)
