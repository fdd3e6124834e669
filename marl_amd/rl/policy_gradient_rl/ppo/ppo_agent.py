"""PPO learner with the reference's agent interface (rl/policy_gradient_rl/ppo/ppo_agent.py:15-145):
.buffer, .select_actions(obs, avail, test_mode), .train(), .save_models/.load_models.

Same mathematics, step for step (:85-134): whole-batch reward normalisation with the unbiased std,
old log-probs, critic values of obs/next_obs, GAE (the HIP reverse scan instead of the T-step torch loop),
then `epochs` full-batch updates of critic (masked MSE) and actor (clipped surrogate + entropy bonus) with
grad-norm clipping, Adam and a StepLR stepped once per epoch.

Data-parallel form: with torch.distributed initialised, every rank holds its own shard of environments;
loss numerators and `masks.sum()` are reduced over ranks (all-reduce of a few scalars), gradients are
all-reduced as ONE flat buffer per optimiser step (RCCL over xGMI: <=25 KB, latency bound), and the
clipping/optimiser step then runs identically on every rank.
"""
import torch as th
import torch.distributed as dist
import torch.nn.functional as F
from torch.distributions import Categorical
from torch.optim import Adam

from marl_amd.buffer.episode_buffer import EpisodeBuffer
from marl_amd.rl.policy_gradient_rl.ppo.ppo_network import Actor, Critic
from marl_amd.utils.action_selectors import greedy_selector, soft_selector
from marl_amd.utils.advantage_utils import get_gae
from marl_amd.utils import mlp_device, optim_device, ppo_device


from marl_amd.utils.dist_utils import (all_reduce_grads_, all_reduce_sum_, broadcast_params_, dist_on as _dist_on,
                                       global_mean_std)  # noqa: E402,F401
from marl_amd.utils.rl_utils import UnfusedShape  # noqa: E402


class PPOAgent:
    def __init__(self, args):
        # ppo_agent.py:17 picks cuda when present; `ec_force_device` lets the host-logic tests pin the CPU
        self.device = th.device(getattr(args, "ec_force_device", None) or ("cuda" if th.cuda.is_available() else "cpu"))
        self.args = args
        self.actor = Actor(args).to(self.device)
        self.critic = Critic(args).to(self.device)
        if _dist_on():  # identical initial parameters on every rank
            broadcast_params_([self.actor, self.critic])
        kw = dict(lr=args.lr)
        if args.use_adm_epsilon:
            kw["eps"] = args.adam_epsilon
        self.actor_optimizer = Adam(params=self.actor.parameters(), **kw)
        self.critic_optimizer = Adam(params=self.critic.parameters(), **kw)
        self.actor_lr_scheduler = th.optim.lr_scheduler.StepLR(self.actor_optimizer, step_size=args.lr_decay_step,
                                                               gamma=args.lr_decay_gamma)
        self.critic_lr_scheduler = th.optim.lr_scheduler.StepLR(self.critic_optimizer, step_size=args.lr_decay_step,
                                                                gamma=args.lr_decay_gamma)
        self.buffer = EpisodeBuffer(args, device=self.device)
        self._eye = th.eye(args.n_agents, device=self.device)
        self.last_losses = {}
        self._flags_checked = False
        # fused HIP update: every p.grad is a view into one flat buffer that the kernels accumulate into
        self._fg_actor = self._fg_critic = None
        if self._fused_policy_ok() and not getattr(args, "ec_torch_update", False):
            self._fg_actor = ppo_device.FlatGrads(self.actor)
            self._fg_critic = ppo_device.FlatGrads(self.critic)
        elif self.device.type == "cuda" and not getattr(args, "ec_torch_update", False):
            raise UnfusedShape("PPOAgent", args)

    def _with_ids(self, obs):
        ids = self._eye.expand(*obs.shape[:-2], -1, -1)
        return th.cat([obs, ids], dim=-1)

    def _fused_policy_ok(self):
        a = self.args
        return (self.device.type == "cuda" and a.activate_fun == "relu" and a.hidden_dim == 64 and a.obs_shape == 9
                and a.n_actions <= 64)

    @th.no_grad()
    def select_actions(self, obs, avail_actions, test_mode=False):
        test_mode = False  # the reference's PPO always samples (ppo_agent.py:62)
        if self._fused_policy_ok():
            # one launch: one-hot agent id, 3-layer MLP, mask, softmax and the categorical draw
            # (ec_policy_forward; the uniforms come from torch's device generator)
            lead = obs.shape[:-1]
            u = th.rand(obs.numel() // obs.shape[-1], device=self.device)
            _, _, actions = mlp_device.policy_forward(mlp_device.linear_params(self.actor), obs.to(self.device),
                                                      self.args.n_agents, avail_actions.to(self.device), u,
                                                      want_probs=False)
            return actions.view(lead)
        obs = self._with_ids(obs.to(self.device))
        avail_actions = avail_actions.to(self.device)
        actor_outs = self.actor(obs)
        actor_outs[avail_actions == 0] = -1e10
        return greedy_selector(actor_outs) if test_mode else soft_selector(actor_outs)

    def evaluate_actions(self, inputs, avail_actions, actions):
        actor_outs = self.actor(inputs)
        actor_outs = actor_outs.masked_fill(avail_actions == 0, -1e10)
        dist_ = Categorical(F.softmax(actor_outs, dim=-1))
        return dist_.log_prob(actions.squeeze(-1)).unsqueeze(-1), dist_.entropy().unsqueeze(-1)

    def _train_fused(self):
        """ppo_agent.py:85-134 with every network pass on the fused HIP kernels (ec_ppo_actor_grad /
        ec_ppo_critic_grad): no autograd graph, no per-row intermediates in HBM."""
        a = self.args
        N = a.n_agents
        self._fg_actor.bind()
        self._fg_critic.bind()
        obs, avail_actions, actions, rewards, masks, next_obs = self.buffer.sample()
        shape = rewards.shape
        if not self._flags_checked or getattr(a, "ec_check_buffer_flags", False):
            # the shortcuts below trust two claims the runner's transition store leaves on the buffer: verified on the
            # first update (and on every one under --ec_check_buffer_flags=1)
            self._flags_checked = True
            if hasattr(self.buffer, "check_whole_episode_flags"):
                self.buffer.check_whole_episode_flags()
        if a.normalize_rewards:
            mean, std = global_mean_std(rewards)
            rewards = (rewards - mean) / (std + 1e-5)
        # (old log-probs, ppo_agent.py:96: left by the actor's pass of epoch 0, whose parameters are still the rollout's --
        # no forward-only launch for them)
        old_logp = None
        old_v, old_next_v = ppo_device.values_and_next_values(self.critic, obs, next_obs, masks, N, self.buffer)
        advantages = get_gae(a, rewards, old_v, old_next_v, masks)
        if a.normalize_advantages:
            mean, std = global_mean_std(advantages)
            advantages = (advantages - mean) / (std + 1e-5)
        targets = advantages + old_v
        known = getattr(self.buffer, "known_mask_sum", None)
        if known is not None and not _dist_on():
            inv_denom = 1.0 / known  # the runner filed a whole episode: no device->host sync for masks.sum()
        else:
            inv_denom = 1.0 / float(all_reduce_sum_(masks.sum().reshape(1).clone()))
        losses = th.zeros(2, device=self.device)
        for _ in range(a.epochs):
            losses.zero_()
            self._fg_critic.zero_()
            ppo_device.critic_pass(self.critic, obs, masks, N, target=targets, inv_denom=inv_denom,
                                   grads=self._fg_critic.flat, loss=losses[0:1])
            all_reduce_sum_(self._fg_critic.flat)
            self._clip_and_step(self.critic, self.critic_optimizer)
            self.critic_lr_scheduler.step()

            self._fg_actor.zero_()
            logp, _ = ppo_device.actor_pass(self.actor, obs, avail_actions, actions, masks, N, old_logp=old_logp,
                                            adv=advantages, eps_clip=a.eps_clip,
                                            entropy_coef=a.entropy_coef if a.use_entropy else 0.0, inv_denom=inv_denom,
                                            grads=self._fg_actor.flat, loss=losses[1:2], first_epoch=old_logp is None)
            if old_logp is None:
                old_logp = logp
            all_reduce_sum_(self._fg_actor.flat)
            self._clip_and_step(self.actor, self.actor_optimizer)
            self.actor_lr_scheduler.step()
        all_reduce_sum_(losses)
        self.last_losses = {"critic_loss": losses[0].clone(), "actor_loss": losses[1].clone()}

    def _clip_and_step(self, net, optimizer):
        """clip_grad_norm_ + Adam.step() (ppo_agent.py:128-141): one launch (ec_adam_clip_step), or torch's ~20"""
        a = self.args
        if optim_device.clip_adam_step([optimizer], a.grad_norm_clip if a.use_grad_clip else None):
            return
        if a.use_grad_clip:
            th.nn.utils.clip_grad_norm_(net.parameters(), a.grad_norm_clip)
        optimizer.step()

    def train(self):
        if self._fg_actor is not None:
            return self._train_fused()
        a = self.args
        obs, avail_actions, actions, rewards, masks, next_obs = self.buffer.sample()
        obs = self._with_ids(obs)
        next_obs = self._with_ids(next_obs)
        if a.normalize_rewards:
            mean, std = global_mean_std(rewards)
            rewards = (rewards - mean) / (std + 1e-5)
        with th.no_grad():
            old_log_pi_taken, _ = self.evaluate_actions(obs, avail_actions, actions)
            old_v = self.critic(obs)
            old_next_v = self.critic(next_obs)
            advantages = get_gae(a, rewards, old_v, old_next_v, masks)
            if a.normalize_advantages:
                mean, std = global_mean_std(advantages)
                advantages = (advantages - mean) / (std + 1e-5)
            targets = advantages + old_v
            denom = all_reduce_sum_(masks.sum().reshape(1).clone())
        for _ in range(a.epochs):
            log_pi_taken, entropy = self.evaluate_actions(obs, avail_actions, actions)
            v = self.critic(obs)
            critic_loss = (((v - targets) * masks) ** 2).sum() / denom

            self.critic_optimizer.zero_grad()
            critic_loss.backward()
            all_reduce_grads_(list(self.critic.parameters()))
            if a.use_grad_clip:
                th.nn.utils.clip_grad_norm_(self.critic.parameters(), a.grad_norm_clip)
            self.critic_optimizer.step()
            self.critic_lr_scheduler.step()

            ratios = th.exp(log_pi_taken - old_log_pi_taken)
            surr1 = ratios * advantages
            surr2 = th.clamp(ratios, 1 - a.eps_clip, 1 + a.eps_clip) * advantages
            objective = th.min(surr1, surr2)
            if a.use_entropy:
                objective = objective + a.entropy_coef * entropy
            actor_loss = -(objective * masks).sum() / denom

            self.actor_optimizer.zero_grad()
            actor_loss.backward()
            all_reduce_grads_(list(self.actor.parameters()))
            if a.use_grad_clip:
                th.nn.utils.clip_grad_norm_(self.actor.parameters(), a.grad_norm_clip)
            self.actor_optimizer.step()
            self.actor_lr_scheduler.step()
        # each rank holds its shard's share of the (globally normalised) losses: sum them for reporting
        self.last_losses = {"critic_loss": all_reduce_sum_(critic_loss.detach().clone().reshape(1))[0],
                            "actor_loss": all_reduce_sum_(actor_loss.detach().clone().reshape(1))[0]}

    def save_models(self, path):
        th.save(self.actor.state_dict(), "{}/actor.th".format(path))
        th.save(self.critic.state_dict(), "{}/critic.th".format(path))

    def load_models(self, path):
        self.actor.load_state_dict(th.load("{}/actor.th".format(path), map_location=self.device, weights_only=True))
        self.critic.load_state_dict(th.load("{}/critic.th".format(path), map_location=self.device, weights_only=True))

    def load_offline_models(self, path):
        self.load_models(path)
