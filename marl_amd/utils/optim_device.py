"""Host side of ec_adam_clip_step (include/marl_amd.h): `clip_grad_norm_` + `Adam.step()` of up to four small networks
in one launch.  The optimiser objects stay the owners of their state (exp_avg, exp_avg_sq, step), so state_dicts and
checkpoints are torch's; the kernel updates those tensors in place."""
import ctypes as C

import torch as th

from marl_amd import _abi

MAX_NETS, MAX_TENSORS = 4, 16


def _eligible(opt):
    if type(opt) is not th.optim.Adam or len(opt.param_groups) != 1:
        return False
    g = opt.param_groups[0]
    if g.get("weight_decay", 0) != 0 or g.get("amsgrad", False) or g.get("maximize", False) or g.get("differentiable", False):
        return False
    ps = g["params"]
    if not 1 <= len(ps) <= MAX_TENSORS:
        return False
    for p in ps:
        if (not p.is_cuda or p.dtype != th.float32 or not p.is_contiguous() or p.grad is None or not p.grad.is_contiguous()
                or p.grad.dtype != th.float32 or p.grad.is_sparse):
            return False
    return True


def _state(opt, p):
    st = opt.state[p]
    if len(st) == 0:  # as Adam._init_group does
        capturable = opt.param_groups[0].get("capturable", False)
        st["step"] = (th.zeros((), dtype=th.float32, device=p.device) if capturable else th.tensor(0.0, dtype=th.float32))
        st["exp_avg"] = th.zeros_like(p, memory_format=th.preserve_format)
        st["exp_avg_sq"] = th.zeros_like(p, memory_format=th.preserve_format)
    return st


def clip_adam_step(optimizers, max_norm):
    """clip_grad_norm_(params of optimizers[i], max_norm) and optimizers[i].step() for every i, in one launch.
    max_norm: one number for all, or one per optimiser (None / <= 0: no clipping).  Returns False — having done nothing —
    when an optimiser is not a plain fp32 torch.optim.Adam on the device (the caller then takes the torch path)."""
    optimizers = list(optimizers)
    if not 1 <= len(optimizers) <= MAX_NETS or not all(_eligible(o) for o in optimizers):
        return False
    if not isinstance(max_norm, (list, tuple)):
        max_norm = [max_norm] * len(optimizers)
    dev = optimizers[0].param_groups[0]["params"][0].device
    n_t, P, G, M, V, S, numel, step_host, hyper, host_counted = [], [], [], [], [], [], [], [], [], []
    # every optimiser is checked before any state is touched: a refusal leaves all of them as they were
    for opt in optimizers:
        ps = opt.param_groups[0]["params"]
        on_dev = [("step" in opt.state[p] and opt.state[p]["step"].is_cuda) or
                  ("step" not in opt.state[p] and opt.param_groups[0].get("capturable", False)) for p in ps]
        if any(on_dev) != all(on_dev) or any(p.device != dev for p in ps):
            return False
    for opt, mn in zip(optimizers, max_norm):
        g = opt.param_groups[0]
        ps = g["params"]
        n_t.append(len(ps))
        sts = [_state(opt, p) for p in ps]
        if not sts[0]["step"].is_cuda:  # the host keeps the count (non-capturable Adam): advanced once the launch is made
            step_host.append(float(sts[0]["step"]) + 1.0)
            host_counted.append(sts)
        else:
            step_host.append(0.0)
        for p, st in zip(ps, sts):
            P.append(p.data_ptr())
            G.append(p.grad.data_ptr())
            M.append(st["exp_avg"].data_ptr())
            V.append(st["exp_avg_sq"].data_ptr())
            S.append(st["step"].data_ptr() if st["step"].is_cuda else 0)
            numel.append(p.numel())
        b1, b2 = g["betas"]
        hyper.append((float(g["lr"]), float(b1), float(b2), float(g["eps"]), float(mn) if mn else 0.0))
    n = len(P)
    vps = lambda xs: (C.c_void_p * n)(*xs)
    f32s = lambda xs: (C.c_double * len(xs))(*xs)
    with th.cuda.device(dev):
        _abi.check(_abi.lib().ec_adam_clip_step(
            len(optimizers), (C.c_int32 * len(n_t))(*n_t), vps(P), vps(G), vps(M), vps(V), vps(S), (C.c_int64 * n)(*numel),
            f32s(step_host), f32s([h[0] for h in hyper]), f32s([h[1] for h in hyper]), f32s([h[2] for h in hyper]),
            f32s([h[3] for h in hyper]), f32s([h[4] for h in hyper]),
            C.c_void_p(th.cuda.current_stream(dev).cuda_stream)), "ec_adam_clip_step")
    for sts in host_counted:  # (after the launch's return code: a failed call has advanced nothing)
        for st in sts:
            st["step"] += 1
    for opt in optimizers:
        opt._opt_called = True  # (what an lr_scheduler looks at to tell that step() came first)
    return True
