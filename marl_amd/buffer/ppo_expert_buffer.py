"""PPOExpertBuffer with the reference's interface (buffer/ppo_expert_buffer.py:4-50): the on-policy rollout store of
the GAIL / DGAIL agents.  Same fields as EpisodeBuffer plus `insertAll`; device-resident here."""
from marl_amd.buffer.episode_buffer import EpisodeBuffer


class PPOExpertBuffer(EpisodeBuffer):
    def insertAll(self, obs, avail_actions, actions, rewards, masks, next_obs):
        for dst, src in ((self.obs, obs), (self.avail_actions, avail_actions), (self.actions, actions),
                         (self.rewards, rewards), (self.masks, masks), (self.next_obs, next_obs)):
            dst.copy_(src)
        self.next_obs_is_shifted_obs = False  # (written in place: the whole-episode claims no longer hold)
        self.known_mask_sum = None
