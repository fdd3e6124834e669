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


def all_reduce_sum_(t):
    if dist_on():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def all_reduce_grads_(params, average=False):
    """Sum (or average) gradients over ranks with a single flat all-reduce."""
    if not dist_on():
        return
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return
    flat = th.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    if average:
        flat /= dist.get_world_size()
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n


def broadcast_params_(modules, src=0):
    """identical initial parameters (and buffers) on every rank"""
    if not dist_on():
        return
    for m in modules:
        for t in list(m.parameters()) + list(m.buffers()):
            dist.broadcast(t.data, src=src)


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
