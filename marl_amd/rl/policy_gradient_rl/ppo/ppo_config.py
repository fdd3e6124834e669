"""PPO flags, same names/defaults as the reference's rl/policy_gradient_rl/ppo/ppo_config.py:1-30.
`lr_decay_gamma` and `grad_norm_clip` keep the reference's type=int declarations."""

_FLAGS = (
    ("name", str, "ppo"), ("algo_type", str, "rl"), ("runner", str, "episode"), ("n_threads", int, 1),
    ("hidden_dim", int, 64), ("gamma", float, 0.99), ("lambda_", float, 0.95), ("lr", float, 0.0003),
    ("lr_decay_step", int, 15000), ("lr_decay_gamma", int, 0.99), ("entropy_coef", float, 0.01),
    ("epochs", int, 4), ("eps_clip", float, 0.2), ("normalize_rewards", bool, True),
    ("normalize_advantages", bool, False), ("use_orthogonal_init", bool, False), ("activate_fun", str, "relu"),
    ("use_adm_epsilon", bool, False), ("adam_epsilon", float, 1e-5), ("use_grad_clip", bool, True),
    ("grad_norm_clip", int, 10), ("save_models", bool, True), ("save_buffers", bool, False),
    ("load_models", bool, False), ("checkpoint_path", str, ""),
)


def add_ppo_args(parser):
    for name, typ, default in _FLAGS:
        parser.add_argument("--" + name, type=typ, default=default)
    parser.add_argument("--use_entropy", default=True, action="store_true")
    return parser
