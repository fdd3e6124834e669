"""Host side of ec_linear_wgrad / ec_linear_fwd (include/marl_amd.h): the weight / bias gradient and the forward (or
input gradient) of a Linear layer over a tall batch."""
import ctypes as C

import torch as th

from marl_amd import _abi

N_PARTIALS = 512  # blocks of four waves: two per CU
FUSE2_MAX_K = 32   # widest input for which ec_linear2_fwd_bn is the product's choice (fwd2_bn_ok)


def wgrad_ok(gy, x):
    return (gy.is_cuda and x.is_cuda and gy.dtype == th.float32 and x.dtype == th.float32 and gy.dim() == 2 and x.dim() == 2
            and gy.shape[0] == x.shape[0] and gy.shape[1] in (32, 64, 96, 128) and 1 <= x.shape[1] <= 128
            and not th.is_grad_enabled())


def wgrad(gy, x):
    """(gy^T @ x [c_out, k_in], gy.sum(0) [c_out]) in one pass over gy [rows, c_out] and x [rows, k_in]"""
    gy, x = gy.contiguous(), x.contiguous()
    rows, c_out = gy.shape
    k_in = x.shape[1]
    n_part = int(min(N_PARTIALS, max(1, (rows + 255) // 256)))
    partials = th.empty(n_part, c_out * k_in + c_out, dtype=th.float32, device=gy.device)
    with th.cuda.device(gy.device):
        _abi.check(_abi.lib().ec_linear_wgrad(C.c_void_p(gy.data_ptr()), C.c_void_p(x.data_ptr()), rows, c_out, k_in, n_part,
                                              C.c_void_p(partials.data_ptr()),
                                              C.c_void_p(th.cuda.current_stream(gy.device).cuda_stream)), "ec_linear_wgrad")
    s = partials.sum(0)
    return s[:c_out * k_in].view(c_out, k_in), s[c_out * k_in:]


def fwd_ok(x, w):
    """shapes for which ec_linear_fwd beats hipBLASLt (measured, tools/bench_linear.py): up to 64 inputs, and 96 or 128
    (whole 32-input tiles: the rows then come in through the staged load; 128 -> 64 0.87 against 0.95 ms at 4.1 M rows);
    other widths above 64 read row-per-lane and lose"""
    return (x.is_cuda and w.is_cuda and x.dtype == th.float32 and w.dtype == th.float32 and x.dim() == 2 and w.dim() == 2
            and x.shape[1] == w.shape[1] and (1 <= x.shape[1] <= 64 or (x.shape[1] in (96, 128) and x.data_ptr() % 16 == 0))
            and 1 <= w.shape[0] <= 128 and not th.is_grad_enabled())


def fwd(x, w, b=None, want_stats=False):
    """x @ w.t() + b for x [rows, k_in], w [c_out, k_in] (ec_linear_fwd).
    want_stats: returns (y, partials [n, 2, c_out], pivot [c_out]) -- per-block column sums of (y - pivot) and
    (y - pivot)^2 (ec_linear_fwd_stats), the input of ec_bn_finalize for a BatchNorm over y; the pivot is y's first row"""
    x, w = x.contiguous(), w.contiguous()
    rows, k_in = x.shape
    c_out = w.shape[0]
    y = th.empty(rows, c_out, dtype=th.float32, device=x.device)
    ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
    if want_stats:
        b_ = b.contiguous() if b is not None else None
        pivot = fwd(x[:1], w, b_).reshape(c_out)
        n_part = int(min(1024, max(1, (rows + 127) // 128)))
        partials = th.empty(n_part, 2, c_out, dtype=th.float32, device=x.device)
        with th.cuda.device(x.device):
            _abi.check(_abi.lib().ec_linear_fwd_stats(ptr(x), ptr(w), ptr(b_), rows, k_in, c_out, ptr(y), ptr(pivot), n_part,
                                                      ptr(partials), C.c_void_p(th.cuda.current_stream(x.device).cuda_stream)),
                       "ec_linear_fwd_stats")
        return y, partials, pivot
    with th.cuda.device(x.device):
        _abi.check(_abi.lib().ec_linear_fwd(ptr(x), ptr(w), ptr(b.contiguous() if b is not None else None), rows, k_in, c_out,
                                            ptr(y), C.c_void_p(th.cuda.current_stream(x.device).cuda_stream)), "ec_linear_fwd")
    return y


def fwd_stats_only(x, w, b):
    """(partials [n, 2, c_out], pivot [c_out]) of y = x @ w.t() + b WITHOUT y (ec_linear_fwd_stats with y = NULL): the same
    partials and pivot (y's first row) `fwd(..., want_stats=True)` leaves, for a layer whose output a later kernel
    recomputes (fwd2_bn)"""
    x, w = x.contiguous(), w.contiguous()
    rows, k_in = x.shape
    c_out = w.shape[0]
    ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
    b_ = b.contiguous() if b is not None else None
    pivot = fwd(x[:1], w, b_).reshape(c_out)
    n_part = int(min(1024, max(1, (rows + 127) // 128)))
    partials = th.empty(n_part, 2, c_out, dtype=th.float32, device=x.device)
    with th.cuda.device(x.device):
        _abi.check(_abi.lib().ec_linear_fwd_stats(ptr(x), ptr(w), ptr(b_), rows, k_in, c_out, ptr(None), ptr(pivot), n_part,
                                                  ptr(partials), C.c_void_p(th.cuda.current_stream(x.device).cuda_stream)),
                   "ec_linear_fwd_stats")
    return partials, pivot


def fwd_stats_from_moments(x, w, b):
    """(partials [1, 2, c_out], pivot [c_out]) for ec_bn_finalize over y = x @ w.t() + b, from the moments of x's rows
    (ec_row_moments) instead of a pass that computes y: with d = x[row] - x[0], sum(y - y0) = W sum(d) and
    sum((y - y0)^2)[o] = w_o^T (sum d d^T) w_o; the block sums are added and the two small products taken in float64"""
    x, w = x.contiguous(), w.contiguous()
    rows, k_in = x.shape
    c_out = w.shape[0]
    kp = 32 * ((k_in + 31) // 32)
    n_part = int(min(1024, max(1, (rows + 127) // 128)))
    part = th.empty(n_part, kp * kp + kp, dtype=th.float32, device=x.device)
    ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
    with th.cuda.device(x.device):
        _abi.check(_abi.lib().ec_row_moments(ptr(x), rows, k_in, n_part, ptr(part),
                                             C.c_void_p(th.cuda.current_stream(x.device).cuda_stream)), "ec_row_moments")
    tot = part.sum(0, dtype=th.float64)
    m2 = tot[:kp * kp].view(kp, kp)[:k_in, :k_in]
    wd = w.double()
    partials = th.stack([wd @ tot[kp * kp:kp * kp + k_in], ((wd @ m2) * wd).sum(1)]).float().reshape(1, 2, c_out)
    pivot = fwd(x[:1], w, b.contiguous() if b is not None else None).reshape(c_out)
    return partials, pivot


def fwd2_bn_ok(x, w1, w2):
    """shapes for which ec_linear2_fwd_bn wins: [rows, <= 32] -> 128 -> 64 (the GAIL discriminator's first two layers at
    hidden_dim 64 and 10 nodes).  The kernel takes up to 96 inputs, but beyond one 32-input tile it spills and loses to
    the layer-by-layer kernels (8.2 M rows x 90 inputs, the 40-node shape: 8.3 ms against 5.5, tools/prof_reward_pass.sh;
    4.1 M rows x 50 inputs, 20 nodes: reward pass 3.03 ms against 2.86 with FUSE2_MAX_K = 64, tools/bench_reward_pass.py)"""
    return (x.is_cuda and w1.is_cuda and w2.is_cuda and x.dtype == w1.dtype == w2.dtype == th.float32 and x.dim() == 2
            and w1.dim() == 2 and w2.dim() == 2 and x.shape[1] == w1.shape[1] and 1 <= x.shape[1] <= FUSE2_MAX_K
            and tuple(w1.shape[:1]) == (128,) and tuple(w2.shape) == (64, 128) and not th.is_grad_enabled())


def fwd2_bn(x, w1, b1, scale, shift, seed, p, w2, b2, want_stats=False):
    """dropout(relu((x @ w1.t() + b1) * scale + shift), p) @ w2.t() + b2 in one kernel (ec_linear2_fwd_bn): layer 1's output
    stays in registers.  Bit-equal to fwd_bn(fwd(x, w1, b1), scale, shift, seed, p, w2, b2); want_stats as in fwd_bn."""
    x, w1, w2 = x.contiguous(), w1.contiguous(), w2.contiguous()
    rows, k_in = x.shape
    c_mid, c_out = w1.shape[0], w2.shape[0]
    y = th.empty(rows, c_out, dtype=th.float32, device=x.device)
    b1 = b1.contiguous() if b1 is not None else None
    b2 = b2.contiguous() if b2 is not None else None
    partials, n_part = None, 0
    if want_stats:
        assert b2 is not None
        n_part = int(min(1024, max(1, (rows + 127) // 128)))
        partials = th.empty(n_part, 2, c_out, dtype=th.float32, device=x.device)
    ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
    with th.cuda.device(x.device):
        _abi.check(_abi.lib().ec_linear2_fwd_bn(ptr(x), ptr(w1), ptr(b1), ptr(scale.contiguous()), ptr(shift.contiguous()),
                                                C.c_uint64(seed), C.c_float(p), ptr(w2), ptr(b2), rows, k_in, c_mid, c_out, ptr(y),
                                                n_part, ptr(partials), C.c_void_p(th.cuda.current_stream(x.device).cuda_stream)),
                   "ec_linear2_fwd_bn")
    return (y, partials) if want_stats else y


def fwd_bn_ok(x, w):
    """shapes ec_linear_fwd_bn takes (any k_in <= 128: against a separate tail kernel plus a GEMM it wins at 128 too)"""
    return (x.is_cuda and w.is_cuda and x.dtype == th.float32 and w.dtype == th.float32 and x.dim() == 2 and w.dim() == 2
            and x.shape[1] == w.shape[1] and 1 <= x.shape[1] <= 128 and 1 <= w.shape[0] <= 128 and not th.is_grad_enabled())


def fwd_bn(x, scale, shift, seed, p, w, b=None, want_stats=False):
    """dropout(relu(x * scale + shift), p) @ w.t() + b with the keep flags of ec_bn_relu_dropout_fwd for `seed`
    (ec_linear_fwd_bn): nothing of the activated tensor is written.
    want_stats (needs b): returns (y, partials [n, 2, c_out]) -- per-block column sums of (y - b) and (y - b)^2, the
    input of ec_bn_finalize for a BatchNorm over y with b as the pivot row"""
    x, w = x.contiguous(), w.contiguous()
    rows, k_in = x.shape
    c_out = w.shape[0]
    y = th.empty(rows, c_out, dtype=th.float32, device=x.device)
    b = b.contiguous() if b is not None else None
    partials, n_part = None, 0
    if want_stats:
        assert b is not None
        n_part = int(min(1024, max(1, (rows + 127) // 128)))
        partials = th.empty(n_part, 2, c_out, dtype=th.float32, device=x.device)
    ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
    with th.cuda.device(x.device):
        _abi.check(_abi.lib().ec_linear_fwd_bn(ptr(x), ptr(scale.contiguous()), ptr(shift.contiguous()), C.c_uint64(seed),
                                               C.c_float(p), ptr(w), ptr(b), rows, k_in, c_out, ptr(y), n_part, ptr(partials),
                                               C.c_void_p(th.cuda.current_stream(x.device).cuda_stream)),
                   "ec_linear_fwd_bn")
    return (y, partials) if want_stats else y


def disc_rows(obs, actions, n_agents, n_actions):
    """(obs | one-hot agent id | one-hot action) rows [rows, O + N + A] of obs [..., N, O] and actions [..., N(, 1)] on the
    device (ec_disc_rows): what Discriminator.rows(cat([obs, eye]), one_hot(actions)) builds, in one write"""
    O = obs.shape[-1]
    obs = obs.reshape(-1, O).contiguous()
    actions = actions.reshape(-1).contiguous()
    assert obs.dtype == th.float32 and actions.dtype == th.int64 and obs.shape[0] == actions.shape[0]
    rows = obs.shape[0]
    out = th.empty(rows, O + n_agents + n_actions, dtype=th.float32, device=obs.device)
    with th.cuda.device(obs.device):
        _abi.check(_abi.lib().ec_disc_rows(C.c_void_p(obs.data_ptr()), C.c_void_p(actions.data_ptr()), rows, O, n_agents, n_actions,
                                           C.c_void_p(out.data_ptr()), C.c_void_p(th.cuda.current_stream(obs.device).cuda_stream)),
                   "ec_disc_rows")
    return out
