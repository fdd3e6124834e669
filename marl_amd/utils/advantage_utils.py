"""GAE on the HIP library.  `get_gae` keeps the reference's signature and return shape
(utils/advantage_utils.py:4-9); the T-step Python loop of torch ops is one kernel launch."""
import ctypes as C

import torch as th

from marl_amd import _abi


def gae_device(rewards, obs_values, next_obs_value, masks, gamma, lambda_):
    """All inputs [B,T,N,1] (or [B,T,N]) float32 on one ROCm device; returns the same shape."""
    shape = rewards.shape
    if rewards.dim() not in (3, 4):
        raise ValueError("expected [B,T,N,1] or [B,T,N]")
    B, T, N = int(shape[0]), int(shape[1]), int(shape[2])
    ts = []
    for t in (rewards, obs_values, next_obs_value, masks):
        if t.shape != shape:
            raise ValueError("shape mismatch")
        if not t.is_cuda:
            raise RuntimeError("gae_device needs device tensors: marl_amd has no CPU fallback")
        ts.append(t.detach().to(th.float32).contiguous())
    adv = th.empty(shape, dtype=th.float32, device=rewards.device)
    stream = C.c_void_p(th.cuda.current_stream(rewards.device).cuda_stream)
    with th.cuda.device(rewards.device):
        # gamma*lambda is formed in Python double precision and rounded once, as the reference's
        # `args.gamma * args.lambda_ * tensor` does
        _abi.check(_abi.lib().ec_gae(*[C.c_void_p(t.data_ptr()) for t in ts], C.c_void_p(adv.data_ptr()), B, T, N,
                                     C.c_float(gamma), C.c_float(gamma * lambda_), stream), "ec_gae")
    return adv


def get_gae(args, rewards, obs_values, next_obs_value, masks):
    return gae_device(rewards, obs_values, next_obs_value, masks, args.gamma, args.lambda_)
