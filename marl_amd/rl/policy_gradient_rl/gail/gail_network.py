"""Networks of the reference's GAIL agent (rl/policy_gradient_rl/gail/gail_network.py:7-132); parameter names match
so that its actor.th / critic.th / discriminator.th checkpoints load unchanged."""
import torch as th
import torch.nn as nn
import torch.nn.functional as F

from marl_amd.rl.policy_gradient_rl.dgail.dgail_network import SplitKLinear
from marl_amd.utils import linear_device
from marl_amd.utils.dist_utils import all_reduce_sum_, dist_on

_ACT = {"relu": F.relu, "tanh": th.tanh, "leaky_relu": F.leaky_relu}


def _act(args, allow):
    name = getattr(args, "activate_fun", "relu")
    return _ACT[name] if name in allow else F.relu


class _ActorCritic(nn.Module):
    def __init__(self, args, n_out):
        super().__init__()
        self.args = args
        self.fc1 = nn.Linear(args.obs_shape + args.n_agents, args.hidden_dim)
        self.fc2 = nn.Linear(args.hidden_dim, args.hidden_dim)
        self.fc3 = nn.Linear(args.hidden_dim, n_out)
        self.activate_func = _act(args, ("relu", "tanh"))

    def forward(self, obs):
        return self.fc3(self.activate_func(self.fc2(self.activate_func(self.fc1(obs)))))


class Actor(_ActorCritic):
    """gail_network.py:7-30"""

    def __init__(self, args):
        super().__init__(args, args.n_actions)


class Critic(_ActorCritic):
    """gail_network.py:33-56"""

    def __init__(self, args):
        super().__init__(args, 1)


class _TallBatchNormFn(th.autograd.Function):
    """Training-mode BatchNorm1d over [rows, C] for inputs beyond 2^31 elements, out of column reductions and fused
    multiply-adds.  torch's own backward (batch_norm_backward_reduce_kernel with 64-bit indexing) takes 1.8 s per call on
    [32.8 M, 128] on MI355X -- 7 s of GAIL's update at BASELINE configs[4] (65.5 M rows).  Same arithmetic as
    F.batch_norm(training=True): biased variance for the normalisation, the exact backward through mean and variance."""

    CHUNK = 1 << 22  # rows per reduction chunk (bounds the elementwise temporaries)

    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        var, mean = th.var_mean(x, dim=0, unbiased=False)
        invstd = th.rsqrt(var + eps)
        scale = invstd * weight
        y = th.addcmul(bias - mean * scale, x, scale)
        ctx.save_for_backward(x, weight, mean, invstd)
        ctx.mark_non_differentiable(mean, var)
        return y, mean, var

    @staticmethod
    def backward(ctx, dy, _dmean, _dvar):
        x, weight, mean, invstd = ctx.saved_tensors
        n = x.shape[0]
        s1 = th.zeros_like(mean)
        s2 = th.zeros_like(mean)
        for lo in range(0, n, _TallBatchNormFn.CHUNK):
            d, xc = dy[lo:lo + _TallBatchNormFn.CHUNK], x[lo:lo + _TallBatchNormFn.CHUNK]
            s1 += d.sum(0)
            s2 += (d * xc).sum(0)
        s2 = (s2 - mean * s1) * invstd                 # sum dy * xhat
        a = weight * invstd
        b = -a * invstd * s2 / n
        c = -a * s1 / n - b * mean
        dx = th.addcmul(th.addcmul(c, dy, a), x, b)    # w * invstd * (dy - mean(dy) - xhat * mean(dy * xhat))
        return dx, s2, s1, None


class _SyncBatchNormFn(th.autograd.Function):
    """Training-mode BatchNorm1d over the rows of EVERY rank (data parallel: each rank holds a shard of the rollout and the
    reference normalises over the whole of it, gail_network.py:97-125).  Forward: the ranks' (n, sum x) and then their
    sums of squared deviations about the global mean are all-reduced.  Backward: as torch's SyncBatchNorm -- dx uses the
    all-reduced sums of dy and dy * x (so that, after the learner's gradient AVERAGE over ranks, everything upstream
    carries the gradient of the ranks' mean loss), the layer's own weight / bias gradients stay the local sums."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        stat = th.cat([th.full((1,), float(x.shape[0]), dtype=th.float64, device=x.device), x.double().sum(0)])
        all_reduce_sum_(stat)
        n = stat[0]
        mean = (stat[1:] / n).to(x.dtype)
        m2 = ((x - mean) ** 2).double().sum(0)
        all_reduce_sum_(m2)
        var = (m2 / n).to(x.dtype)
        invstd = th.rsqrt(var + eps)
        scale = invstd * weight
        y = th.addcmul(bias - mean * scale, x, scale)
        ctx.save_for_backward(x, weight, mean, invstd)
        ctx.n = float(n)
        ctx.mark_non_differentiable(mean, var)
        return y, mean, var

    @staticmethod
    def backward(ctx, dy, _dmean, _dvar):
        x, weight, mean, invstd = ctx.saved_tensors
        n = ctx.n
        s1_l = dy.sum(0)
        s2_l = (dy * x).sum(0)
        glob = th.stack([s1_l, s2_l]).double()
        all_reduce_sum_(glob)
        s1, s2 = glob[0].to(x.dtype), glob[1].to(x.dtype)
        s2 = (s2 - mean * s1) * invstd                 # sum over all ranks of dy * xhat
        a = weight * invstd
        b = -a * invstd * s2 / n
        c = -a * s1 / n - b * mean
        dx = th.addcmul(th.addcmul(c, dy, a), x, b)
        return dx, (s2_l - mean * s1_l) * invstd, s1_l, None


def _keep_mask(x, p):
    """bool [rows, C]: bernoulli(1 - p) per element, drawn in row chunks (no [rows, C] fp32 temporary)"""
    keep = th.empty(x.shape, dtype=th.bool, device=x.device)
    step = _TallBatchNormFn.CHUNK
    for lo in range(0, x.shape[0], step):
        th.ge(th.rand(x[lo:lo + step].shape, device=x.device), p, out=keep[lo:lo + step])
    return keep


class _TallBnReluDropFn(th.autograd.Function):
    """BatchNorm1d (training) -> ReLU -> Dropout(p) of one discriminator layer as ONE autograd node for tall inputs:
    keeps the pre-BN input and a one-byte keep mask instead of the four [rows, C] fp32 tensors the separate modules
    save (at 65.5 M rows x 128 that is the difference between fitting 288 GB with room and the allocator thrashing)."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, p):
        var, mean = th.var_mean(x, dim=0, unbiased=False)
        invstd = th.rsqrt(var + eps)
        scale = invstd * weight
        shift = bias - mean * scale
        keep = _keep_mask(x, p)                           # bernoulli(1 - p), as F.dropout draws it
        out = th.addcmul(shift, x, scale).clamp_(min=0).mul_(keep).mul_(1.0 / (1.0 - p))
        ctx.save_for_backward(x, keep, weight, mean, invstd, shift)
        ctx.p = p
        ctx.mark_non_differentiable(mean, var)
        return out, mean, var

    @staticmethod
    def backward(ctx, dout, _dmean, _dvar):
        x, keep, weight, mean, invstd, shift = ctx.saved_tensors
        n, C = x.shape
        scale = invstd * weight
        inv_keep = 1.0 / (1.0 - ctx.p)
        step = _TallBatchNormFn.CHUNK

        def dy_chunk(lo):  # d loss / d (BN output) of the rows [lo, lo + step): through dropout and ReLU
            xc = x[lo:lo + step]
            live = keep[lo:lo + step] & (th.addcmul(shift, xc, scale) > 0)
            return dout[lo:lo + step] * live * inv_keep, xc
        s1 = th.zeros_like(mean)
        s2 = th.zeros_like(mean)
        for lo in range(0, n, step):
            d, xc = dy_chunk(lo)
            s1 += d.sum(0)
            s2 += (d * xc).sum(0)
        s2 = (s2 - mean * s1) * invstd
        a = weight * invstd
        b = -a * invstd * s2 / n
        c = -a * s1 / n - b * mean
        dx = th.empty_like(x)
        for lo in range(0, n, step):
            d, xc = dy_chunk(lo)
            th.addcmul(th.addcmul(c, d, a), xc, b, out=dx[lo:lo + step])
        return dx, s2, s1, None, None


def _explicit_mask(x, p):
    """Hook of the HIP layer tail: None = the kernels draw the keep mask themselves (counter-based, from a seed); the
    reference-replay tests return the reference's recorded mask here."""
    return None


_no_explicit_mask = _explicit_mask   # (the fused no-grad pass steps aside when a test has put its own hook in)


def _draw_seed():
    """the dropout seed of one layer call: torch's CPU generator (follows th.manual_seed, no device sync)"""
    return int(th.empty((), dtype=th.int64).random_())


def _bn_batch_stats(x, weight, bias, eps, running, partials=None, pivot=None, sync=False, shape=None):
    """mean, var, invstd, scale, shift [C] of a training-mode BatchNorm over x [rows, C] (fp32, on the device), and the
    module's running statistics updated: one streaming pass (ec_bn_col_stats; torch.var_mean's column reduction runs at
    a third of the HBM rate) leaving per-block partial sums about the pivot row x[0], then ec_bn_finalize, which adds
    them up in float64 and derives everything per-column in one launch.
    running: (running_mean, running_var, num_batches_tracked, momentum or None) or None; updated through raw pointers
    (autograd does not see the buffers change).
    partials [n, 2, C] with pivot [C]: the statistics pass was already done by the kernel that wrote x (ec_linear_fwd_bn's
    sums of (x - pivot) and (x - pivot)^2); only ec_bn_finalize runs.
    sync (data parallel): the statistics are the whole job's -- every rank turns its partial sums into (n, mean, sum of
    squared deviations) in float64, the ranks' moments are merged (Chan's formula, two small all-reduces: the pivots
    differ per rank, so the partial sums themselves cannot be added), and the per-column tail runs as torch ops.
    x = None with shape = (rows, C): a tensor that was never written (its statistics came with `partials`)."""
    import ctypes as C

    from marl_amd import _abi
    rows, Cc = x.shape if x is not None else shape
    dev = x.device if x is not None else partials.device
    stream = C.c_void_p(th.cuda.current_stream(dev).cuda_stream)
    ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
    have = partials is not None
    if have:
        n_part = partials.shape[0]
    else:
        n_part = int(min(_BnReluDropHipFn.N_PARTIALS, max(1, rows // 64)))
        partials = th.empty(n_part, 2, Cc, dtype=th.float32, device=dev)
        pivot = x   # (its first row)
    cols = th.empty(5, Cc, dtype=th.float32, device=dev)
    mean, var, invstd, scale, shift = cols.unbind(0)
    rm, rv, nbt, momentum = running if running is not None else (None, None, None, 0.0)
    if nbt is not None:
        nbt.data += 1  # (before the launch: momentum=None weighs this batch 1 / count)
    if sync and dist_on():
        with th.cuda.device(dev):
            if not have:
                _abi.check(_abi.lib().ec_bn_col_stats(ptr(x), rows, Cc, n_part, ptr(partials), stream), "ec_bn_col_stats")
        with th.no_grad():
            s = partials.double().sum(0)                                  # [2, C] about this rank's pivot
            k = pivot.reshape(-1, Cc)[0].double() if pivot.dim() > 1 else pivot.double()
            m_r = k + s[0] / rows
            m2_r = s[1] - s[0] * s[0] / rows
            stat = th.cat([th.full((1,), float(rows), dtype=th.float64, device=dev), m_r * rows])
            all_reduce_sum_(stat)
            n = stat[0]
            g_mean = stat[1:] / n
            m2 = m2_r + rows * (m_r - g_mean) ** 2
            all_reduce_sum_(m2)
            g_var = (m2 / n).clamp_(min=0)
            mean.copy_(g_mean)
            var.copy_(g_var)
            invstd.copy_(th.rsqrt(var + eps))
            scale.copy_(invstd * weight)
            shift.copy_(bias - mean * scale)
            if rm is not None and rv is not None:
                mo = float(momentum) if momentum is not None else 1.0 / float(nbt)
                rm.data.mul_(1 - mo).add_(mean, alpha=mo)
                rv.data.mul_(1 - mo).add_((g_var * (n / (n - 1).clamp(min=1))).float(), alpha=mo)
        return mean, var, invstd, scale, shift
    with th.cuda.device(dev):
        if not have:
            _abi.check(_abi.lib().ec_bn_col_stats(ptr(x), rows, Cc, n_part, ptr(partials), stream), "ec_bn_col_stats")
        _abi.check(_abi.lib().ec_bn_finalize(ptr(partials), n_part, ptr(pivot), rows, Cc, ptr(weight), ptr(bias), C.c_float(eps),
                                             C.c_float(-1.0 if momentum is None else float(momentum)), ptr(rm), ptr(rv),
                                             ptr(nbt), ptr(mean), ptr(var), ptr(invstd), ptr(scale), ptr(shift), stream),
                   "ec_bn_finalize")
    return mean, var, invstd, scale, shift


class _BnReluDropHipFn(th.autograd.Function):
    """The same node on the HIP kernels of csrc/ec_bn.hip (fp32 on the device): forward 1 read + 1 write after the
    statistics pass, backward 4 reads + 1 write, only x (and an explicit mask, if one was given) kept."""

    N_PARTIALS = 1024

    @staticmethod
    def _args(x, scale, shift, mask, seed, p):
        import ctypes as C
        vp = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
        return vp, (vp(x), vp(scale), vp(shift), vp(mask), C.c_uint64(seed), C.c_float(p))

    @staticmethod
    def forward(ctx, x, weight, bias, eps, p, running=None, stats=None, sync=False):
        """running: (running_mean, running_var, num_batches_tracked, momentum or None) of the module, updated in the
        same launch that finalises the statistics (through raw pointers: autograd does not see the buffers change)"""
        import ctypes as C

        from marl_amd import _abi
        x = x.contiguous()
        rows, Cc = x.shape
        stream = C.c_void_p(th.cuda.current_stream(x.device).cuda_stream)
        # (stats: (partials, pivot) left by the kernel that wrote x, ec_linear_fwd_stats -- no statistics pass here then)
        mean, var, invstd, scale, shift = _bn_batch_stats(x, weight, bias, eps, running, *(stats or (None, None)), sync=sync)
        mask = _explicit_mask(x, p)
        if mask is not None:
            mask = mask.to(device=x.device, dtype=th.uint8).contiguous()
        seed = _draw_seed()
        out = th.empty_like(x)
        vp, head = _BnReluDropHipFn._args(x, scale, shift, mask, seed, p)
        with th.cuda.device(x.device):
            _abi.check(_abi.lib().ec_bn_relu_dropout_fwd(*head, rows, Cc, vp(out), stream), "ec_bn_relu_dropout_fwd")
        ctx.save_for_backward(x, weight, mean, invstd, scale, shift, mask)
        ctx.seed, ctx.p = seed, p
        ctx.sync = bool(sync) and dist_on()
        ctx.mark_non_differentiable(mean, var)
        return out, mean, var

    @staticmethod
    def backward(ctx, dout, _dmean, _dvar):
        import ctypes as C

        from marl_amd import _abi
        x, weight, mean, invstd, scale, shift, mask = ctx.saved_tensors
        dout = dout.contiguous()
        rows, Cc = x.shape
        n_part = int(min(_BnReluDropHipFn.N_PARTIALS, max(1, rows // 64)))
        partials = th.empty(n_part, 2, Cc, dtype=th.float32, device=x.device)
        stream = C.c_void_p(th.cuda.current_stream(x.device).cuda_stream)
        vp, head = _BnReluDropHipFn._args(x, scale, shift, mask, ctx.seed, ctx.p)
        with th.cuda.device(x.device):
            _abi.check(_abi.lib().ec_bn_relu_dropout_bwd_sums(vp(dout), *head, rows, Cc, n_part, vp(partials), stream),
                       "ec_bn_relu_dropout_bwd_sums")
            # a, b, c of dx = a dy + b x + c, d weight = sum dy xhat, d bias = sum dy: one launch (ec_bn_bwd_coeffs)
            cols = th.empty(5, Cc, dtype=th.float32, device=x.device)
            a, b, c, s2, s1 = cols.unbind(0)
            _abi.check(_abi.lib().ec_bn_bwd_coeffs(vp(partials), n_part, vp(mean), vp(invstd), vp(weight), rows, Cc, vp(a), vp(b),
                                                   vp(c), vp(s2), vp(s1), stream), "ec_bn_bwd_coeffs")
            if ctx.sync:
                # data parallel (see _SyncBatchNormFn): dx from the sums over every rank's rows; the layer's own weight /
                # bias gradients (s2, s1 above) stay this rank's
                glob = th.cat([partials.double().sum(0).reshape(-1),
                               th.full((1,), float(rows), dtype=th.float64, device=x.device)])
                all_reduce_sum_(glob)
                n = glob[-1]
                t1, t2 = glob[:Cc].float(), glob[Cc:2 * Cc].float()
                g2 = (t2 - mean * t1) * invstd
                a = weight * invstd
                b = -a * invstd * g2 / n
                c = (-a * t1 / n - b * mean).float().contiguous()
                a, b = a.float().contiguous(), b.float().contiguous()
            dx = th.empty_like(x)
            _abi.check(_abi.lib().ec_bn_relu_dropout_bwd_dx(vp(dout), *head, vp(a), vp(b), vp(c), rows, Cc, vp(dx), stream),
                       "ec_bn_relu_dropout_bwd_dx")
        return dx, s2, s1, None, None, None, None, None


class TallBatchNorm1d(nn.BatchNorm1d):
    """nn.BatchNorm1d (same parameters, buffers and state_dict keys); training-mode inputs of TALL_NUMEL elements or
    more take _TallBatchNormFn."""

    TALL_NUMEL = 1 << 30

    def forward(self, x, sync=False):
        if sync and dist_on() and self.training:   # statistics over every rank's rows (data parallel)
            y, mean, var = _SyncBatchNormFn.apply(x, self.weight, self.bias, self.eps)
            n = th.tensor([float(x.shape[0])], dtype=th.float64, device=x.device)
            self._track(int(all_reduce_sum_(n).item()), mean, var)
            return y
        if not self.tall(x):
            return super().forward(x)
        y, mean, var = _TallBatchNormFn.apply(x, self.weight, self.bias, self.eps)
        self._track(x.shape[0], mean, var)
        return y

    def hip_ok(self, x):
        return (x.dim() == 2 and x.is_cuda and x.dtype == th.float32 and x.shape[0] > 1 and x.shape[-1] == self.num_features
                and self.hip_ok_module())

    def hip_ok_module(self):
        """the module's side of hip_ok: training mode, affine, a width the kernels' thread mapping takes"""
        C = self.num_features
        return self.training and C % 4 == 0 and C <= 256 and 256 % (C // 4) == 0 and self.weight is not None

    def relu_dropout(self, x, p, stats=None, sync=False):
        """dropout(relu(self(x)), p) for a training-mode input, as one autograd node (HIP kernels on the device);
        stats: (partials, pivot) of x's column sums when the kernel that wrote x left them; sync: batch statistics over
        every rank's rows (data parallel)"""
        if self.hip_ok(x):
            running = None
            if self.track_running_stats:  # updated by the kernel that finalises the statistics
                running = (self.running_mean, self.running_var, self.num_batches_tracked, self.momentum)
            y, _, _ = _BnReluDropHipFn.apply(x, self.weight, self.bias, self.eps, float(p), running, stats, sync)
            return y
        y, mean, var = _TallBnReluDropFn.apply(x, self.weight, self.bias, self.eps, float(p))
        self._track(x.shape[0], mean, var)
        return y

    def tall(self, x):
        return self.training and x.dim() == 2 and x.numel() >= self.TALL_NUMEL

    def _track(self, n, mean, var):
        if self.track_running_stats:
            with th.no_grad():
                self.num_batches_tracked.data += 1
                m = self.momentum if self.momentum is not None else 1.0 / float(self.num_batches_tracked)
                # (through .data: torch's own batch_norm saves the running statistics for its backward, and an earlier
                # call of this module on a small batch -- GAIL's 64 expert rows -- may still be waiting for it)
                self.running_mean.data.mul_(1 - m).add_(mean, alpha=m)
                self.running_var.data.mul_(1 - m).add_(var * (n / max(n - 1, 1)), alpha=m)


class Discriminator(nn.Module):
    """gail_network.py:59-132: (obs | agent id | one-hot action) -> 2H -> H -> H/2 -> 1 logit, each hidden layer
    Linear -> BatchNorm (skipped for a single row) -> activation -> Dropout; Xavier-uniform weights, zero biases."""

    def __init__(self, args):
        super().__init__()
        self.args = args
        h = args.hidden_dim
        widths = (args.obs_shape + args.n_agents + args.n_actions, h * 2, h, h // 2)
        for i, p in enumerate((0.3, 0.3, 0.2), start=1):
            setattr(self, "fc%d" % i, SplitKLinear(widths[i - 1], widths[i]))   # tall-batch weight gradients
            setattr(self, "bn%d" % i, TallBatchNorm1d(widths[i]))
            setattr(self, "dropout%d" % i, nn.Dropout(p))
        # (SplitKLinear too: the stock weight gradient of a 1-output layer is a [1, rows] x [rows, 32] GEMM, 2.7 ms at 4 M rows)
        self.fc4 = SplitKLinear(widths[3], 1)
        self.activate_func = _act(args, ("relu", "tanh", "leaky_relu"))
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)
                nn.init.constant_(m.bias, 0)

    FUSE_FIRST_TWO = True   # the no-grad pass takes layers 1 and 2 in one kernel where ec_linear2_fwd_bn has the shape

    def forward(self, obs, actions):
        lead = obs.shape[:-1]
        x = self.forward_rows(self.rows(obs, actions))
        return x.view(*lead, 1) if len(lead) > 1 else x

    @staticmethod
    def rows(obs, actions):
        """the network's input rows [rows, obs + id + one-hot action] (built once by a caller that evaluates them twice)"""
        return th.cat([obs.reshape(-1, obs.shape[-1]), actions.reshape(-1, actions.shape[-1])], dim=-1)

    def _fused_tails_ok(self, x):
        """a pass that keeps nothing for a backward, with every layer's tail on the HIP kernels: the tails then fold
        into the next layer's forward (ec_linear_fwd_bn)"""
        if th.is_grad_enabled() or _explicit_mask is not _no_explicit_mask or self.activate_func is not F.relu:
            return False
        if not (x.is_cuda and x.dtype == th.float32 and x.dim() == 2 and x.shape[0] > 1):
            return False
        for i in (1, 2, 3):
            bn, drop, nxt = getattr(self, "bn%d" % i), getattr(self, "dropout%d" % i), getattr(self, "fc%d" % (i + 1))
            if not (bn.hip_ok_module() and bn.weight.is_cuda and drop.training and 0.0 <= drop.p < 1.0 and nxt.weight.shape[1] <= 128
                    and nxt.weight.shape[0] <= 128 and nxt.weight.dtype == th.float32):
                return False
        return True

    def _forward_rows_fused_tails(self, x, sync=False):
        """forward_rows for a no-grad pass in training mode (gail_agent.py:165-167, the reward pass): per layer the
        statistics pass over the Linear's output, then BatchNorm -> ReLU -> Dropout applied inside the NEXT Linear's
        kernel, with the draws (one seed per layer, in layer order) the two-step path would make"""
        partials = pivot = None
        first = 1
        if self.FUSE_FIRST_TWO and linear_device.fwd2_bn_ok(x, self.fc1.weight, self.fc2.weight) and self.fc2.bias is not None:
            # layers 1 and 2 in one kernel (ec_linear2_fwd_bn): bn1's statistics from the moments of the input rows
            # (ec_row_moments), then layer 1 in registers -> tail -> layer 2; the 128-wide tensor is never written
            bn, drop = self.bn1, self.dropout1
            running = (bn.running_mean, bn.running_var, bn.num_batches_tracked, bn.momentum) if bn.track_running_stats else None
            partials, pivot = linear_device.fwd_stats_from_moments(x, self.fc1.weight, self.fc1.bias)
            _, _, _, scale, shift = _bn_batch_stats(None, bn.weight, bn.bias, bn.eps, running, partials, pivot, sync=sync,
                                                    shape=(x.shape[0], self.fc1.weight.shape[0]))
            x, partials = linear_device.fwd2_bn(x, self.fc1.weight, self.fc1.bias, scale, shift, _draw_seed(), float(drop.p),
                                                self.fc2.weight, self.fc2.bias, want_stats=True)
            pivot = self.fc2.bias.detach()
            first = 2
        elif linear_device.fwd_ok(x, self.fc1.weight):   # the first layer's statistics from the kernel that computes it
            x, partials, pivot = linear_device.fwd(x, self.fc1.weight, self.fc1.bias, want_stats=True)
        else:
            x = self.fc1(x)
        for i in range(first, 4):
            bn, drop, nxt = getattr(self, "bn%d" % i), getattr(self, "dropout%d" % i), getattr(self, "fc%d" % (i + 1))
            running = None
            if bn.track_running_stats:
                running = (bn.running_mean, bn.running_var, bn.num_batches_tracked, bn.momentum)
            x = x.contiguous()
            _, _, _, scale, shift = _bn_batch_stats(x, bn.weight, bn.bias, bn.eps, running, partials, pivot, sync=sync)
            # (the kernel that writes layer i + 1's pre-activations also leaves their column sums for bn(i + 1), about
            # the layer's bias as the pivot: its inputs are normalised activations, so the outputs stay within a few
            # standard deviations of the bias and the one-pass variance does not cancel)
            want = i < 3 and nxt.bias is not None
            out = linear_device.fwd_bn(x, scale, shift, _draw_seed(), float(drop.p), nxt.weight, nxt.bias, want_stats=want)
            x, partials, pivot = (out[0], out[1], nxt.bias.detach()) if want else (out, None, None)
        return x

    def forward_rows(self, x, sync=False):
        """logits [rows, 1] of input rows as `rows` builds them.  sync: the rows are this rank's shard of a batch spread
        over the ranks of a data-parallel job, and every BatchNorm takes its statistics over the whole of it (the
        reference normalises over the whole rollout, gail_network.py:97-125); no effect without a process group."""
        sync = bool(sync) and dist_on()
        if self._fused_tails_ok(x):
            return self._forward_rows_fused_tails(x, sync)
        for i in (1, 2, 3):
            stats = []   # (the layer's kernel leaves the column sums of its output where it is the HIP one)
            x = getattr(self, "fc%d" % i)(x, stats)
            bn, drop = getattr(self, "bn%d" % i), getattr(self, "dropout%d" % i)
            if ((bn.hip_ok(x) or (bn.tall(x) and not sync)) and drop.training and 0.0 <= drop.p < 1.0
                    and self.activate_func is F.relu):
                # one autograd node for the layer's tail (memory at 10^7..10^8 rows)
                x = bn.relu_dropout(x, drop.p, stats[0] if stats else None, sync)
                continue
            if x.shape[0] > 1 or sync:
                x = bn(x, sync) if sync else bn(x)
            x = drop(self.activate_func(x))
        return self.fc4(x)
