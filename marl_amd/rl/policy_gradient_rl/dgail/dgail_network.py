"""Networks of the reference's DGAIL agent (rl/policy_gradient_rl/dgail/dgail_network.py:6-108), kept as torch
modules so that training (autograd) and the reference's checkpoints work unchanged (`model.N.*` parameter
names).  The no-grad evaluations of the rollout / reward path do not run these modules: they hand the
parameters to the HIP library (marl_amd/utils/mlp_device.py).
"""
import torch as th
import torch.nn as nn
import torch.nn.functional as F


def _mlp3(n_in, n_hidden, n_out, wrap=lambda m: m, tail=()):
    """Linear-ReLU-Linear-ReLU-Linear(+tail); `wrap` decorates the two hidden Linears."""
    return nn.Sequential(wrap(nn.Linear(n_in, n_hidden)), nn.ReLU(), wrap(nn.Linear(n_hidden, n_hidden)), nn.ReLU(),
                         nn.Linear(n_hidden, n_out), *tail)


class PolicyNet(nn.Module):
    """dgail_network.py:6-19"""

    def __init__(self, input_dim, output_dim, hidden_dim):
        super().__init__()
        self.model = _mlp3(input_dim, hidden_dim, output_dim)

    def forward(self, x):
        return self.model(x)


class ValueNet(nn.Module):
    """dgail_network.py:22-34"""

    def __init__(self, input_dim, hidden_dim):
        super().__init__()
        self.model = _mlp3(input_dim, hidden_dim, 1)

    def forward(self, x):
        return self.model(x)


class GAILDiscriminator(nn.Module):
    """dgail_network.py:37-50: probability that (s, a) came from the expert."""

    def __init__(self, input_dim, hidden_dim):
        super().__init__()
        self.model = _mlp3(input_dim, hidden_dim, 1, tail=(nn.Sigmoid(),))

    def forward(self, x):
        return self.model(x)


class DiffusionDiscriminator(nn.Module):
    """dgail_network.py:53-108: two-step conditional denoiser eps(x_t, t, c) with spectral norm on the hidden
    layers; compute_loss = per-row MSE to the injected noise + 1e-4 * |d sum(eps)/d x_t|^2."""

    R1_LAMBDA = 1e-4  # dgail_network.py:106

    def __init__(self, input_dim, hidden_dim):
        super().__init__()
        self.input_dim = input_dim
        self.hidden = hidden_dim
        self.time_embed = nn.Embedding(2, hidden_dim)
        self.cond_embed = nn.Embedding(2, hidden_dim)
        self.mlp = _mlp3(input_dim + 2 * hidden_dim, hidden_dim, input_dim, wrap=nn.utils.parametrizations.spectral_norm)

    def forward(self, x, t, c):
        return self.mlp(th.cat([x, self.time_embed(t), self.cond_embed(c)], dim=-1))

    def compute_loss(self, sa, c, t=None, noise=None):
        """sa [B, input_dim], c [B] (0 agent / 1 expert).  t / noise default to the reference's fresh draws
        (dgail_network.py:90-91); pass them to make the loss a function of its inputs (what the HIP entry point
        ec_diffusion_reward does)."""
        rows = sa.size(0)
        if t is None:
            t = th.randint(0, 2, (rows,), device=sa.device)
        if noise is None:
            noise = th.randn_like(sa)
        x_t = (sa + t.to(sa.dtype).unsqueeze(-1) * noise).requires_grad_(True)
        eps = self.forward(x_t, t, c)
        mse = F.mse_loss(eps, noise, reduction="none").mean(dim=1)
        (grad,) = th.autograd.grad(eps.sum(), x_t, create_graph=True, retain_graph=True)
        return mse + grad.pow(2).reshape(rows, -1).sum(1) * self.R1_LAMBDA
