"""Host side of ec_ppo_actor_grad / ec_ppo_critic_grad (include/marl_amd.h): the fused forward + loss + backward of
the PPO actor and critic over the whole rollout batch.

`FlatGrads` gives a module one flat fp32 gradient buffer in parameter order and points every `p.grad` into it, so
that the kernel's output IS the optimizer's input and the data-parallel all-reduce is a single call on one tensor."""
import ctypes as C

import torch as th

from marl_amd import _abi
from marl_amd.utils.mlp_device import OBS_DIM, _f32, _p, _stream, linear_params


class FlatGrads:
    def __init__(self, module, storage=None):
        """storage: optional zeroed fp32 tensor of exactly the module's parameter count to use as the flat buffer (a slice
        of a larger one, so that several networks' gradients are cleared by one fill)"""
        self.params = [p for p in module.parameters()]
        n = sum(p.numel() for p in self.params)
        if storage is not None and (storage.numel() != n or storage.dtype != th.float32 or not storage.is_contiguous()):
            raise ValueError("FlatGrads storage must be a contiguous float32 tensor of %d elements" % n)
        self.flat = storage if storage is not None else th.zeros(n, dtype=th.float32, device=self.params[0].device)
        self.bind()

    def bind(self):
        """(re)point every p.grad into the flat buffer: an optimizer's zero_grad() (set_to_none) or a checkpoint load
        in between replaces the .grad tensors, after which the kernels and the optimizer would use different memory"""
        off = 0
        for p in self.params:
            view = self.flat[off:off + p.numel()].view_as(p)
            if p.grad is None or p.grad.data_ptr() != view.data_ptr():
                p.grad = view
            off += p.numel()

    def zero_(self):
        self.flat.zero_()


def _rows(obs):
    obs = _f32(obs, "obs").reshape(-1, OBS_DIM)
    return obs, obs.shape[0]


EC_PPO_LOGITS = 1
EC_PPO_GATHERED = 2
EC_PPO_NLL = 4
EC_PPO_FIRST_EPOCH = 8


def _index(row_index, dev):
    if row_index is None:
        return None
    if row_index.dtype != th.int64 or not row_index.is_contiguous() or row_index.device != dev:
        row_index = row_index.to(device=dev, dtype=th.int64).contiguous()
    return row_index


def actor_pass(actor, obs, avail, actions, mask, n_agents, old_logp=None, adv=None, eps_clip=0.2, entropy_coef=0.0,
               inv_denom=1.0, grads=None, loss=None, want_logp=False, want_entropy=False, row_index=None, logits=False,
               gathered=False, nll=False, first_epoch=False):
    """One launch.  Forward only when old_logp is None (returns logp / entropy); with old_logp and adv it accumulates
    the loss into `loss` [1] and, if `grads` (flat, parameter order) is given, the gradient into it.
    row_index [r] (optional): the pass takes rows row_index[0..r) of the per-row tables (a shuffled minibatch), outputs
    are in pass order; avail / mask may be None (no action mask / all ones); logits=True: Categorical(logits=...).
    gathered=True: the tables ARE the minibatch (gathered by the caller); row_index only names each row's agent.
    nll=True: the objective is the log-prob of `actions` (loss = -inv_denom * sum mask * logp: a cross-entropy term);
    old_logp / adv are not needed.
    first_epoch=True (with adv, no old_logp): the pass of the update's epoch 0 -- the old log-probs are this pass's own
    (ratio 1); returns them as logp for the later epochs (EC_PPO_FIRST_EPOCH)."""
    w1, b1, w2, b2, w3, b3 = linear_params(actor)
    obs, table_rows = _rows(obs)
    dev = obs.device
    A = w3.shape[0]
    if avail is not None:
        avail = avail.to(th.int32).contiguous().reshape(-1, A)
    actions = actions.to(th.int64).contiguous().reshape(-1)
    if mask is not None:
        mask = _f32(mask, "mask").reshape(-1)
    if old_logp is not None:
        old_logp = _f32(old_logp, "old_logp").reshape(-1)
    if adv is not None:
        adv = _f32(adv, "adv").reshape(-1)
    if first_epoch:
        want_logp = True
    for t in (avail, actions, mask, old_logp, adv):
        if t is not None and t.shape[0] != table_rows:
            raise ValueError("rows mismatch")
    row_index = _index(row_index, dev)
    rows = table_rows if (row_index is None or gathered) else int(row_index.shape[0])
    if gathered and (row_index is None or row_index.shape[0] != table_rows):
        raise ValueError("gathered tables need one index per row")
    logp = th.empty(rows, dtype=th.float32, device=dev) if want_logp else None
    ent = th.empty(rows, dtype=th.float32, device=dev) if want_entropy else None
    with th.cuda.device(dev):
        _abi.check(_abi.lib().ec_ppo_actor_grad_ex(_p(w1), _p(b1), _p(w2), _p(b2), _p(w3), _p(b3), _p(obs), _p(avail),
                                                   _p(actions), _p(old_logp), _p(adv), _p(mask), rows, int(n_agents),
                                                   int(A), int(w2.shape[0]), C.c_float(eps_clip),
                                                   C.c_float(entropy_coef), C.c_float(inv_denom), _p(grads), _p(loss),
                                                   _p(logp), _p(ent), _p(row_index),
                                                   (EC_PPO_LOGITS if logits else 0) | (EC_PPO_GATHERED if gathered else 0)
                                                   | (EC_PPO_NLL if nll else 0) | (EC_PPO_FIRST_EPOCH if first_epoch else 0),
                                                   _stream(dev)), "ec_ppo_actor_grad_ex")
    return logp, ent


def critic_pass(critic, obs, mask, n_agents, target=None, inv_denom=1.0, grads=None, loss=None, want_values=False,
                row_index=None):
    """n_agents = 0: the network takes the 9 observation inputs only (DGAIL's ValueNet); row_index / mask as actor_pass."""
    w1, b1, w2, b2, w3, b3 = linear_params(critic)
    obs, table_rows = _rows(obs)
    dev = obs.device
    if w1.shape[1] != OBS_DIM + n_agents:
        raise ValueError("critic input width %d is not 9 + %d" % (w1.shape[1], n_agents))
    if mask is not None:
        mask = _f32(mask, "mask").reshape(-1)
    if target is not None:
        target = _f32(target, "target").reshape(-1)
    row_index = _index(row_index, dev)
    rows = table_rows if row_index is None else int(row_index.shape[0])
    values = th.empty(rows, dtype=th.float32, device=dev) if want_values else None
    with th.cuda.device(dev):
        _abi.check(_abi.lib().ec_ppo_critic_grad_ex(_p(w1), _p(b1), _p(w2), _p(b2), _p(w3), _p(b3), _p(obs), _p(target),
                                                    _p(mask), rows, int(n_agents), int(w2.shape[0]),
                                                    C.c_float(inv_denom), _p(grads), _p(loss), _p(values),
                                                    _p(row_index), _stream(dev)), "ec_ppo_critic_grad_ex")
    return values


def gail_disc_pass(disc, obs, actions, expert, inv_denom=1.0, grads=None, loss=None, want_prob=False):
    """One data set of DGAIL's GAIL-discriminator step (ec_gail_disc_grad): p = sigmoid(MLP(obs | one-hot action)),
    loss += -inv_denom * sum log(p + 1e-8) (expert) or -inv_denom * sum log(1 - p + 1e-8) (agent), gradient into `grads`
    (flat, parameter order).  Returns p [rows] when want_prob."""
    w1, b1, w2, b2, w3, b3 = linear_params(disc)
    obs, rows = _rows(obs)
    dev = obs.device
    n_actions = w1.shape[1] - OBS_DIM
    actions = actions.to(th.int64).contiguous().reshape(-1)
    if actions.shape[0] != rows:
        raise ValueError("rows mismatch")
    prob = th.empty(rows, dtype=th.float32, device=dev) if want_prob else None
    with th.cuda.device(dev):
        _abi.check(_abi.lib().ec_gail_disc_grad(_p(w1), _p(b1), _p(w2), _p(b2), _p(w3), _p(b3), _p(obs), _p(actions), rows,
                                                int(n_actions), int(w2.shape[0]), 1 if expert else 0,
                                                C.c_float(inv_denom), _p(grads), _p(loss), _p(prob), _stream(dev)),
                   "ec_gail_disc_grad")
    return prob


def values_and_next_values(critic, obs, next_obs, masks, n_agents, buffer=None):
    """V(obs) and V(next_obs), both shaped like obs[..., :1] ([B, T, N, 1]; ppo_agent.py:98-100, gail_agent.py:186-188,
    dgail_agent.py:258-262).  When the runner's transition store filed one observation as next_obs[:, t] and obs[:, t+1]
    (it says so on the buffer: EpisodeBuffer.next_obs_is_shifted_obs), V(next_obs[:, t]) IS V(obs[:, t+1]) -- the kernel
    evaluates rows independently, so the values are the same bits -- and only the last step needs a pass of its own."""
    shape = tuple(obs.shape[:-1]) + (1,)
    v = critic_pass(critic, obs, masks, n_agents, want_values=True).view(shape)
    if shape[1] > 1 and buffer is not None and getattr(buffer, "next_obs_is_shifted_obs", False):
        last_mask = masks[:, -1:].contiguous() if masks is not None else None
        last = critic_pass(critic, next_obs[:, -1:].contiguous(), last_mask, n_agents, want_values=True)
        nv = th.cat([v[:, 1:], last.view(shape[0], 1, *shape[2:])], dim=1)
    else:
        nv = critic_pass(critic, next_obs, masks, n_agents, want_values=True).view(shape)
    return v, nv
