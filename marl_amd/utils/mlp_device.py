"""Host side of the three fp32 MLP entry points of libmarl_amd.so (include/marl_amd.h):
policy forward (+mask, softmax, sampling), GAIL discriminator reward and DGAIL diffusion-discriminator reward.

Each function takes the torch modules' parameters as they are (nn.Linear layout, on the device) and the flat
row-major batches the reference builds with .reshape(-1, dim); nothing is copied or re-laid-out on the host.
There is no CPU fallback.
"""
import ctypes as C

import torch as th

from marl_amd import _abi

OBS_DIM = 9  # edge_node.py:74


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _f32(t, what):
    if not t.is_cuda:
        raise RuntimeError("%s must live on the ROCm device: marl_amd has no CPU fallback" % what)
    return t.detach().to(th.float32).contiguous()


def linear_params(seq):
    """(w1, b1, w2, b2, w3, b3) of a module holding three Linear layers in forward order (ppo_network.py:8-19
    fc1..fc3, dgail_network.py:9-15 model.0/2/4); reading `.weight` resolves parametrizations (spectral norm)
    to the effective weight."""
    lins = [m for m in seq.modules() if isinstance(m, th.nn.Linear)]
    if len(lins) != 3:
        raise ValueError("expected a module with three Linear layers")
    out = []
    for m in lins:
        out.append(_f32(m.weight, "weight"))
        out.append(_f32(m.bias, "bias"))
    return tuple(out)


def _stream(dev):
    return C.c_void_p(th.cuda.current_stream(dev).cuda_stream)


def policy_forward(params, obs, n_agents, avail=None, uniforms=None, want_logits=False, want_probs=True,
                   want_actions=True):
    """Actor forward over obs [rows, 9] (row r belongs to agent r % n_agents).
    Returns (logits|None, probs|None, actions|None); actions are drawn by inverse CDF from `uniforms` [rows],
    or greedy when uniforms is None."""
    w1, b1, w2, b2, w3, b3 = params
    obs = _f32(obs, "obs").reshape(-1, OBS_DIM)
    rows = obs.shape[0]
    A = w3.shape[0]
    if w1.shape[1] != OBS_DIM + n_agents:
        raise ValueError("first layer expects %d inputs, got obs(9)+agents(%d)" % (w1.shape[1], n_agents))
    dev = obs.device
    if avail is not None:
        avail = avail.to(th.int32).contiguous().reshape(-1, A)
        if avail.shape[0] != rows:
            raise ValueError("avail rows mismatch")
    if uniforms is not None:
        uniforms = _f32(uniforms, "uniforms").reshape(-1)
        if uniforms.shape[0] != rows:
            raise ValueError("uniforms rows mismatch")
    logits = th.empty((rows, A), dtype=th.float32, device=dev) if want_logits else None
    probs = th.empty((rows, A), dtype=th.float32, device=dev) if want_probs else None
    actions = th.empty((rows,), dtype=th.int64, device=dev) if want_actions else None
    with th.cuda.device(dev):
        _abi.check(_abi.lib().ec_policy_forward(_p(w1), _p(b1), _p(w2), _p(b2), _p(w3), _p(b3), _p(obs), _p(avail),
                                                _p(uniforms), rows, int(n_agents), int(A), int(w2.shape[0]),
                                                _p(logits), _p(probs), _p(actions), _stream(dev)),
                   "ec_policy_forward")
    return logits, probs, actions


def gail_reward(params, obs, actions, n_actions, weight=1.0, env_reward=None, env_coef=0.0, want_prob=False):
    """-log(1 - D(s,a)) * weight (+ env_coef * env_reward): dgail_agent.py:482-485,501-518. Returns (reward, prob|None),
    both [rows]."""
    w1, b1, w2, b2, w3, b3 = params
    obs = _f32(obs, "obs").reshape(-1, OBS_DIM)
    rows = obs.shape[0]
    dev = obs.device
    actions = actions.to(th.int64).contiguous().reshape(-1)
    if actions.shape[0] != rows:
        raise ValueError("actions rows mismatch")
    if w1.shape[1] != OBS_DIM + n_actions:
        raise ValueError("first layer expects %d inputs, got obs(9)+actions(%d)" % (w1.shape[1], n_actions))
    if env_reward is not None:
        env_reward = _f32(env_reward, "env_reward").reshape(-1)
        if env_reward.shape[0] != rows:
            raise ValueError("env_reward rows mismatch")
    out = th.empty((rows,), dtype=th.float32, device=dev)
    prob = th.empty((rows,), dtype=th.float32, device=dev) if want_prob else None
    with th.cuda.device(dev):
        _abi.check(_abi.lib().ec_gail_reward(_p(w1), _p(b1), _p(w2), _p(b2), _p(w3), _p(b3), _p(obs), _p(actions),
                                             _p(env_reward), rows, int(n_actions), int(w2.shape[0]),
                                             C.c_float(weight), C.c_float(env_coef), _p(prob), _p(out), _stream(dev)),
                   "ec_gail_reward")
    return out, prob


def diffusion_reward(params, time_embed, cond_embed, obs, actions, n_actions, t_expert, noise_expert, t_agent,
                     noise_agent, r1_lambda=1e-4):
    """Both compute_loss passes + D_phi + reward of dgail_agent.py:487-499 with the (t, noise) draws passed in.
    Returns (loss_expert, loss_agent, d_phi, reward), each [rows]."""
    w1, b1, w2, b2, w3, b3 = params
    obs = _f32(obs, "obs").reshape(-1, OBS_DIM)
    rows = obs.shape[0]
    dev = obs.device
    SA = OBS_DIM + n_actions
    actions = actions.to(th.int64).contiguous().reshape(-1)
    te = _f32(time_embed, "time_embed")
    ce = _f32(cond_embed, "cond_embed")
    if w1.shape[1] != SA + 2 * w2.shape[0] or w3.shape[0] != SA:
        raise ValueError("diffusion discriminator shapes do not match obs(9)+actions(%d)" % n_actions)
    tx = t_expert.to(th.uint8).contiguous().reshape(-1)
    ta = t_agent.to(th.uint8).contiguous().reshape(-1)
    nx = _f32(noise_expert, "noise_expert").reshape(-1, SA)
    na = _f32(noise_agent, "noise_agent").reshape(-1, SA)
    for t in (actions, tx, ta, nx, na):
        if t.shape[0] != rows:
            raise ValueError("rows mismatch")
    outs = [th.empty((rows,), dtype=th.float32, device=dev) for _ in range(4)]
    with th.cuda.device(dev):
        _abi.check(_abi.lib().ec_diffusion_reward(_p(w1), _p(b1), _p(w2), _p(b2), _p(w3), _p(b3), _p(te), _p(ce), _p(obs),
                                                  _p(actions), _p(tx), _p(nx), _p(ta), _p(na), rows, int(n_actions),
                                                  int(w2.shape[0]), C.c_float(r1_lambda), *[_p(o) for o in outs],
                                                  _stream(dev)),
                   "ec_diffusion_reward")
    return tuple(outs)


class DiffusionLossCall:
    """One DiffusionDiscriminator.compute_loss call (dgail_network.py:82-108) for condition `c` on the fused kernel
    (ec_diffusion_grad), with its backward.

    Construction reads the parametrised weights ONCE with autograd on: in training mode that is one spectral-norm power
    iteration per hidden layer, what the reference's forward does for the call.  The kernel differentiates with respect
    to its direct inputs (effective W1x / W2, the per-t first-layer bias, W3, its column sums, b2, b3); `backward`
    returns those tensors with their gradients, and one torch.autograd.backward over them carries the chain rule to
    the module's parameters (sigma of the spectral norm, the two embeddings) through a graph of a few 64x64 operations.
    """

    def __init__(self, disc, c, obs, actions, n_actions, t, noise):
        lin1, lin2, lin3 = [m for m in disc.mlp.modules() if isinstance(m, th.nn.Linear)]
        self.SA = SA = OBS_DIM + n_actions
        self.n_actions = n_actions
        w1, w2 = lin1.weight, lin2.weight  # effective weights, graph to the originals
        H = w2.shape[0]
        if w1.shape[1] != SA + 2 * H or lin3.weight.shape[0] != SA:
            raise ValueError("diffusion discriminator shapes do not match obs(9)+actions(%d)" % n_actions)
        bias_t = (lin1.bias + disc.time_embed.weight @ w1[:, SA:SA + H].t()
                  + disc.cond_embed.weight[int(c)] @ w1[:, SA + H:].t())  # [2, H]
        w3sum = lin3.weight.sum(0)
        self.inputs = (w1, bias_t, w2, lin2.bias, lin3.weight, lin3.bias, w3sum)
        self._dev = tuple(_f32(x, "parameter") for x in self.inputs)
        self.obs = _f32(obs, "obs").reshape(-1, OBS_DIM)
        self.rows = rows = self.obs.shape[0]
        self.actions = actions.to(th.int64).contiguous().reshape(-1)
        self.t = t.to(th.uint8).contiguous().reshape(-1)
        self.noise = _f32(noise, "noise").reshape(-1, SA)
        for x in (self.actions, self.t, self.noise):
            if x.shape[0] != rows:
                raise ValueError("rows mismatch")
        self.H = H

    def _launch(self, coef, loss, grads, r1_lambda):
        w1, bias_t, w2, b2, w3, b3, w3sum = self._dev
        dev = self.obs.device
        with th.cuda.device(dev):
            _abi.check(_abi.lib().ec_diffusion_grad(_p(w1), int(w1.shape[1]), _p(bias_t), _p(w2), _p(b2), _p(w3), _p(b3),
                                                    _p(w3sum), _p(self.obs), _p(self.actions), _p(self.t), _p(self.noise),
                                                    _p(coef), self.rows, int(self.n_actions), int(self.H),
                                                    C.c_float(r1_lambda), _p(loss), _p(grads), _stream(dev)),
                       "ec_diffusion_grad")

    def loss(self, r1_lambda=1e-4):
        """per-row loss [rows] (no gradient)"""
        out = th.empty((self.rows,), dtype=th.float32, device=self.obs.device)
        self._launch(None, out, None, r1_lambda)
        return out

    def reward(self, loss_expert, weight=1.0, env_reward=None, env_coef=0.0, r1_lambda=1e-4, want_loss=False):
        """This call being the agent-condition loss ln of dgail_agent.py:487-499 and `loss_expert` [rows] the
        expert-condition loss lp of the same rows: -log(1 - D_phi + 1e-8) * weight + env_coef * env_reward, formed in the
        kernel that evaluates ln (ec_diffusion_reward_from_loss).  Returns (reward [rows], ln [rows] | None)."""
        loss_expert = _f32(loss_expert, "loss_expert").reshape(-1)
        if loss_expert.shape[0] != self.rows:
            raise ValueError("loss_expert rows mismatch")
        if env_reward is not None:
            env_reward = _f32(env_reward, "env_reward").reshape(-1)
            if env_reward.shape[0] != self.rows:
                raise ValueError("env_reward rows mismatch")
        dev = self.obs.device
        out = th.empty((self.rows,), dtype=th.float32, device=dev)
        ln = th.empty((self.rows,), dtype=th.float32, device=dev) if want_loss else None
        w1, bias_t, w2, b2, w3, b3, w3sum = self._dev
        with th.cuda.device(dev):
            _abi.check(_abi.lib().ec_diffusion_reward_from_loss(
                _p(w1), int(w1.shape[1]), _p(bias_t), _p(w2), _p(b2), _p(w3), _p(b3), _p(w3sum), _p(self.obs),
                _p(self.actions), _p(self.t), _p(self.noise), _p(loss_expert), _p(env_reward), C.c_float(weight),
                C.c_float(env_coef), self.rows, int(self.n_actions), int(self.H), C.c_float(r1_lambda), _p(ln), _p(out),
                _stream(dev)), "ec_diffusion_reward_from_loss")
        return out, ln

    def backward(self, coef, r1_lambda=1e-4):
        """coef [rows] = d objective / d loss[r].  Returns (tensors, gradients) for torch.autograd.backward."""
        SA, H = self.SA, self.H
        coef = _f32(coef, "coef").reshape(-1)
        if coef.shape[0] != self.rows:
            raise ValueError("coef rows mismatch")
        flat = th.zeros(H * SA + 2 * H + H * H + H + SA * H + SA + H, dtype=th.float32, device=self.obs.device)
        self._launch(coef, None, flat, r1_lambda)
        return self._unpack(flat)

    def backward_bce(self, other_loss, side, label, scale, r1_lambda=1e-4):
        """The backward with the coefficient taken in the kernel (ec_diffusion_grad_bce): d BCE(D_phi, label) / d loss[r]
        with D_phi = exp(-lp) / (exp(-lp) + exp(-ln) + 1e-8) (dgail_agent.py:209-221), this call being lp (side 0) or ln
        (side 1) and `other_loss` [rows] the other one; scale = 1 / rows of the BCE's mean.  Returns (tensors, gradients,
        this call's per-row loss)."""
        SA, H = self.SA, self.H
        other_loss = _f32(other_loss, "other_loss").reshape(-1)
        if other_loss.shape[0] != self.rows:
            raise ValueError("other_loss rows mismatch")
        dev = self.obs.device
        flat = th.zeros(H * SA + 2 * H + H * H + H + SA * H + SA + H, dtype=th.float32, device=dev)
        loss = th.empty((self.rows,), dtype=th.float32, device=dev)
        w1, bias_t, w2, b2, w3, b3, w3sum = self._dev
        with th.cuda.device(dev):
            _abi.check(_abi.lib().ec_diffusion_grad_bce(
                _p(w1), int(w1.shape[1]), _p(bias_t), _p(w2), _p(b2), _p(w3), _p(b3), _p(w3sum), _p(self.obs),
                _p(self.actions), _p(self.t), _p(self.noise), _p(other_loss), int(side), int(label), C.c_float(scale),
                self.rows, int(self.n_actions), int(self.H), C.c_float(r1_lambda), _p(loss), _p(flat), _stream(dev)),
                "ec_diffusion_grad_bce")
        tensors, grads = self._unpack(flat)
        return tensors, grads, loss

    def _unpack(self, flat):
        SA, H = self.SA, self.H
        o = 0

        def take(n, shape):
            nonlocal o
            v = flat[o:o + n].view(shape)
            o += n
            return v
        g_w1x, g_bsum, g_b1 = take(H * SA, (H, SA)), take(H, (H,)), take(H, (H,))
        g_w2, g_b2, g_w3, g_b3, g_ws = take(H * H, (H, H)), take(H, (H,)), take(SA * H, (SA, H)), take(SA, (SA,)), take(H, (H,))
        g_w1 = th.zeros_like(self._dev[0])
        g_w1[:, :SA] = g_w1x
        g_bias = th.stack([g_bsum - g_b1, g_b1])
        return list(self.inputs), [g_w1, g_bias, g_w2, g_b2, g_w3, g_b3, g_ws]
