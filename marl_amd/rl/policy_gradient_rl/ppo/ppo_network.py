"""Actor / Critic MLPs of the reference's PPO (rl/policy_gradient_rl/ppo/ppo_network.py:6-47):
(obs_shape + n_agents) -> hidden -> hidden -> {n_actions, 1}.  Parameter names match the reference so that
its actor.th / critic.th checkpoints load unchanged."""
import torch.nn as nn

from marl_amd.utils.rl_utils import activate_funs, orthogonal_init


class _MLP(nn.Module):
    def __init__(self, args, n_out, out_gain):
        super().__init__()
        self.args = args
        self.fc1 = nn.Linear(args.obs_shape + args.n_agents, args.hidden_dim)
        self.fc2 = nn.Linear(args.hidden_dim, args.hidden_dim)
        self.fc3 = nn.Linear(args.hidden_dim, n_out)
        self.activate_fun = activate_funs[args.activate_fun]
        if args.use_orthogonal_init:
            orthogonal_init(self.fc1)
            orthogonal_init(self.fc2)
            orthogonal_init(self.fc3, gain=out_gain)

    def forward(self, inputs):
        x = self.activate_fun(self.fc1(inputs))
        x = self.activate_fun(self.fc2(x))
        return self.fc3(x)


class Actor(_MLP):
    def __init__(self, args):
        super().__init__(args, args.n_actions, 0.01)


class Critic(_MLP):
    def __init__(self, args):
        super().__init__(args, 1, 1.0)
