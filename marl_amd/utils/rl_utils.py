"""Small torch helpers with the reference's names (utils/rl_utils.py:1-20)."""
from torch import nn


def soft_update(args, target_model, model):
    for tp, p in zip(target_model.parameters(), model.parameters()):
        tp.data.mul_(1 - args.tau).add_(p.data, alpha=args.tau)


def hard_update(target_model, model):
    target_model.load_state_dict(model.state_dict())


def orthogonal_init(layer, gain=1.0):
    nn.init.orthogonal_(layer.weight, gain=gain)
    nn.init.constant_(layer.bias, 0)


activate_funs = {"tanh": nn.Tanh(), "relu": nn.ReLU()}


class UnfusedShape(RuntimeError):
    """A learner on the GPU whose networks the fused HIP update kernels do not cover (they are built for hidden_dim = 64,
    ReLU, the env's 9-feature observation and at most 64 actions).  The torch-autograd formulation of the same update
    exists and is ~10x slower; it has to be asked for with --ec_torch_update=1 rather than taken silently."""

    def __init__(self, who, args):
        super().__init__(
            "%s: hidden_dim=%s, activate_fun=%s, obs_shape=%s, n_actions=%s is outside what the fused HIP update kernels "
            "cover (hidden_dim 64, relu, obs_shape 9, n_actions <= 64).  Pass --ec_torch_update=1 to run this learner's "
            "update on torch autograd instead (about 10x slower at 4096 envs)."
            % (who, getattr(args, "hidden_dim", None), getattr(args, "activate_fun", "relu"),
               getattr(args, "obs_shape", None), getattr(args, "n_actions", None)))
