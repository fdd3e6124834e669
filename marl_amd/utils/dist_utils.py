"""Data-parallel helpers of the learners: every rank owns a shard of environments; parameters are
replicated; per optimiser step ONE flat all-reduce of the gradients plus a few scalar all-reduces for the whole-batch
statistics the reference computes with tensor.mean()/std()/sum().  torch.distributed: `nccl` (RCCL over xGMI) on the
device, `gloo` in the CPU tests.  With no process group everything is the identity."""
import torch as th
import torch.distributed as dist


def dist_on():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def world_size():
    return dist.get_world_size() if dist_on() else 1


def _host_staged():
    """gloo moves bytes through host memory anyway, and its device-tensor path is not dependable on every build: with
    a gloo group, device tensors are reduced through a host copy (rehearsals on one GPU, CPU tests).  nccl (RCCL)
    reduces device tensors in place, stream-ordered."""
    return dist.get_backend() == "gloo"


def all_reduce_(t, op=None):
    """in-place all-reduce of `t` over all ranks (identity with no process group)"""
    if not dist_on():
        return t
    op = dist.ReduceOp.SUM if op is None else op
    if t.is_cuda and _host_staged():
        h = t.detach().cpu()
        dist.all_reduce(h, op=op)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op)
    return t


def all_reduce_sum_(t):
    return all_reduce_(t)


def all_reduce_grads_(params, average=False):
    """Sum (or average) gradients over ranks with a single flat all-reduce."""
    if not dist_on():
        return
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return
    flat = th.cat([g.reshape(-1) for g in grads])
    all_reduce_(flat)
    if average:
        flat /= dist.get_world_size()
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n


def broadcast_(t, src=0):
    if not dist_on():
        return t
    if t.is_cuda and _host_staged():
        h = t.detach().cpu()
        dist.broadcast(h, src=src)
        t.copy_(h)
    else:
        dist.broadcast(t, src=src)
    return t


def broadcast_params_(modules, src=0):
    """identical initial parameters (and buffers) on every rank"""
    if not dist_on():
        return
    for m in modules:
        for t in list(m.parameters()) + list(m.buffers()):
            broadcast_(t.data, src)


def global_mean_std(x):
    """mean and UNBIASED std over all ranks' elements, as tensor.mean()/tensor.std() of the concatenation."""
    n = th.tensor([float(x.numel())], device=x.device, dtype=th.float64)
    s = x.double().sum().reshape(1)
    all_reduce_sum_(n)
    all_reduce_sum_(s)
    mean = s / n
    ss = ((x.double() - mean) ** 2).sum().reshape(1)
    all_reduce_sum_(ss)
    std = th.sqrt(ss / (n - 1))
    return mean.float(), std.float()


def global_mean(x):
    n = th.tensor([float(x.numel())], device=x.device, dtype=th.float64)
    s = x.double().sum().reshape(1)
    all_reduce_sum_(n)
    all_reduce_sum_(s)
    return (s / n).float()
