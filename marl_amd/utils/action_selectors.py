"""Action selection on device tensors; same functions as the reference's utils/action_selectors.py:6-20.
Sampling uses torch's generator, exactly as the reference does, so trajectory parity is defined by
action replay, not by identical samples."""
import torch as th
import torch.nn.functional as F
from torch.distributions import Categorical


def random_selector(avail_actions):
    return Categorical(avail_actions).sample().long()


def soft_selector(agent_outs):
    return Categorical(F.softmax(agent_outs, dim=-1)).sample().long()


def greedy_selector(agent_outs):
    return agent_outs.argmax(dim=-1)
