"""Networks of the reference's DGAIL agent (rl/policy_gradient_rl/dgail/dgail_network.py:6-108), kept as torch
modules so that training (autograd) and the reference's checkpoints work unchanged (`model.N.*` parameter
names).  The no-grad evaluations of the rollout / reward path do not run these modules: they hand the
parameters to the HIP library (marl_amd/utils/mlp_device.py).
"""
import torch as th
import torch.nn as nn
import torch.nn.functional as F

from marl_amd.utils import linear_device


def _chunked_tn(a, b, chunk):
    """a^T @ b for tall a [rows, m], b [rows, n], reduced in row chunks (a batched GEMM + a small sum)."""
    rows = a.shape[0]
    full = (rows // chunk) * chunk
    out = th.bmm(a[:full].reshape(-1, chunk, a.shape[1]).transpose(1, 2), b[:full].reshape(-1, chunk, b.shape[1])).sum(0)
    if full < rows:
        out = out + a[full:].t() @ b[full:]
    return out


class _TallMatmulFn(th.autograd.Function):
    """gy @ w for tall gy: the input-gradient product of a Linear.  Differentiated again (the R1 penalty of the
    diffusion discriminator differentiates through the first backward), its weight gradient is gy^T @ ggx with the
    row count as the contraction length: chunked like the first-order one."""

    @staticmethod
    def forward(ctx, gy, w):
        ctx.save_for_backward(gy, w)
        return gy @ w

    @staticmethod
    def backward(ctx, ggx):
        gy, w = ctx.saved_tensors
        g_gy = ggx @ w.t() if ctx.needs_input_grad[0] else None
        g_w = _chunked_tn(gy, ggx, _SplitKLinearFn.CHUNK) if ctx.needs_input_grad[1] else None
        return g_gy, g_w


class _SplitKLinearFn(th.autograd.Function):
    """y = x W^T + b whose weight gradient is reduced in row chunks (a batched GEMM + a small sum).  The stock
    backward hands [out, rows] x [rows, in] with rows ~ 10^5..10^6 to one GEMM without split-K: a handful of
    workgroups walk the whole row dimension (350 us for a 64K-row minibatch, 4 ms for 4 M rows on MI355X)."""

    CHUNK = 1024

    @staticmethod
    def forward(ctx, x, w, b, stats_out=None):
        """stats_out: a list; where the HIP forward runs, (partials, pivot) of y's column sums is appended to it
        (ec_linear_fwd_stats: a BatchNorm that follows needs no statistics pass of its own)"""
        ctx.save_for_backward(x, w)
        if linear_device.fwd_ok(x, w):  # (inside a Function's forward grad mode is off)
            if stats_out is not None and b is not None:
                y, partials, pivot = linear_device.fwd(x, w, b, want_stats=True)
                stats_out.append((partials, pivot))
                return y
            return linear_device.fwd(x, w, b)
        return F.linear(x, w, b)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            if th.is_grad_enabled():
                gx = _TallMatmulFn.apply(gy, w)
            elif linear_device.fwd_ok(gy, w.t()):
                gx = linear_device.fwd(gy, w.t())     # gy @ w = gy (w^T)^T: the same kernel on the transposed weight
            else:
                gx = gy @ w
        if (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]) and linear_device.wgrad_ok(gy, x):
            # one pass over gy and x with the row index as the MFMA's K dimension (ec_linear_wgrad)
            gw, gb = linear_device.wgrad(gy, x)
        elif ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            rows = gy.shape[0]
            c = _SplitKLinearFn.CHUNK
            full = (rows // c) * c
            if ctx.needs_input_grad[1]:
                gw = th.bmm(gy[:full].reshape(-1, c, gy.shape[1]).transpose(1, 2), x[:full].reshape(-1, c, x.shape[1])).sum(0)
                if full < rows:
                    gw = gw + gy[full:].t() @ x[full:]
            if ctx.needs_input_grad[2]:
                gb = gy[:full].reshape(-1, c, gy.shape[1]).sum(1).sum(0)
                if full < rows:
                    gb = gb + gy[full:].sum(0)
        return gx, gw, gb, None


class SplitKLinear(nn.Linear):
    """nn.Linear (same parameters, same state_dict keys) with the chunked weight-gradient reduction for tall inputs."""

    MIN_ROWS = 8192

    def forward(self, x, stats_out=None):
        if x.dim() == 2 and x.shape[0] >= self.MIN_ROWS:
            if th.is_grad_enabled():
                return _SplitKLinearFn.apply(x, self.weight, self.bias, stats_out)
            if linear_device.fwd_ok(x, self.weight):  # no-grad evaluations (GAIL's reward pass)
                return linear_device.fwd(x, self.weight, self.bias)
        return F.linear(x, self.weight, self.bias)


def _embed2(table, idx):
    """nn.Embedding(2, h) lookup as a select: same values, but the backward is two masked reductions instead of a
    sort + scatter of ~10^6 indices into two rows (150 ms per DGAIL iteration at 1 M rows)."""
    return th.where(idx.bool().unsqueeze(-1), table[1], table[0])


def _mlp3(n_in, n_hidden, n_out, wrap=lambda m: m, tail=()):
    """Linear-ReLU-Linear-ReLU-Linear(+tail); `wrap` decorates the two hidden Linears."""
    return nn.Sequential(wrap(SplitKLinear(n_in, n_hidden)), nn.ReLU(), wrap(SplitKLinear(n_hidden, n_hidden)), nn.ReLU(),
                         SplitKLinear(n_hidden, n_out), *tail)


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
        return self.mlp(th.cat([x, _embed2(self.time_embed.weight, t), _embed2(self.cond_embed.weight, c)], dim=-1))

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
