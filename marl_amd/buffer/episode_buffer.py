"""On-policy rollout storage with the reference's EpisodeBuffer interface (buffer/episode_buffer.py:4-46),
kept resident in HBM: `insert` is a device-to-device copy of one time slice and `sample` returns the
tensors themselves, so the ~540 MB/iteration host-to-device copy of the reference disappears."""
import torch as th


class EpisodeBuffer:
    def __init__(self, args, device=None):
        self.args = args
        self.device = th.device(device if device is not None else ("cuda" if th.cuda.is_available() else "cpu"))
        B, T, N = args.n_threads, args.episode_limit, args.n_agents
        kw = dict(device=self.device)
        self.obs = th.zeros(B, T, N, args.obs_shape, dtype=th.float, **kw)
        self.avail_actions = th.zeros(B, T, N, args.n_actions, dtype=th.int, **kw)
        self.actions = th.zeros(B, T, N, 1, dtype=th.long, **kw)
        self.rewards = th.zeros(B, T, N, 1, dtype=th.float, **kw)
        self.masks = th.zeros(B, T, N, 1, dtype=th.float, **kw)
        self.next_obs = th.zeros(B, T, N, args.obs_shape, dtype=th.float, **kw)
        self.t = 0
        # Set by the runners' transition store when it has filed a whole episode from reset: next_obs[:, t] is then
        # obs[:, t+1] bit for bit and masks.sum() is known on the host, which lets the learners skip a 147 MB
        # comparison, one critic pass and a device->host sync per iteration.  Any other writer clears them.
        self.next_obs_is_shifted_obs = False
        self.known_mask_sum = None

    def insert(self, obs, avail_actions, actions, rewards, masks, next_obs):
        t = self.t
        self.obs[:, t].copy_(obs, non_blocking=True)
        self.avail_actions[:, t].copy_(avail_actions, non_blocking=True)
        self.actions[:, t].copy_(actions, non_blocking=True)
        self.rewards[:, t].copy_(rewards, non_blocking=True)
        self.masks[:, t].copy_(masks, non_blocking=True)
        self.next_obs[:, t].copy_(next_obs, non_blocking=True)
        self.t += 1
        self.next_obs_is_shifted_obs = False
        self.known_mask_sum = None

    def sample(self):
        return self.obs, self.avail_actions, self.actions, self.rewards, self.masks, self.next_obs

    def reset(self):
        for t in (self.obs, self.avail_actions, self.actions, self.rewards, self.masks, self.next_obs):
            t.zero_()
        self.t = 0
        self.next_obs_is_shifted_obs = False
        self.known_mask_sum = None

    def begin_episode(self):
        """t = 0 without zeroing the tensors (a writer that overwrites every element, ec_store_episode, follows); the
        whole-episode flags are cleared like in reset(): whoever files the episode sets them again"""
        self.t = 0
        self.next_obs_is_shifted_obs = False
        self.known_mask_sum = None

    def check_whole_episode_flags(self):
        """the claims behind next_obs_is_shifted_obs / known_mask_sum, verified on the tensors (a 147 MB comparison and a
        device->host sync at 4096 envs: the learners run it on their first update and under args.ec_check_buffer_flags)"""
        if self.next_obs_is_shifted_obs and self.obs.shape[1] > 1:
            if not th.equal(self.next_obs[:, :-1], self.obs[:, 1:]):
                raise RuntimeError("EpisodeBuffer.next_obs_is_shifted_obs is set but next_obs[:, t] != obs[:, t+1]: the buffer "
                                   "was written to after the runner filed the episode (clear the flag, or call reset())")
        if self.known_mask_sum is not None:
            got = float(self.masks.sum())
            if got != float(self.known_mask_sum):
                raise RuntimeError("EpisodeBuffer.known_mask_sum = %r but masks.sum() = %r" % (self.known_mask_sum, got))

    def can_sample(self) -> bool:
        return self.t >= self.args.episode_limit
