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
