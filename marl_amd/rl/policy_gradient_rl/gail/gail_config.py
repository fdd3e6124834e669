"""GAIL flags, same names/defaults as the reference's rl/policy_gradient_rl/gail/gail_config.py:1-36."""

_FLAGS = (
    ("name", str, "gail"), ("algo_type", str, "rl"), ("runner", str, "gail_runner"),
    ("expert_guidance_steps", int, 40000), ("bc_guidance_steps", int, 2000), ("max_expert_buffer_size", int, 1000),
    ("n_threads", int, 1), ("hidden_dim", int, 64), ("gamma", float, 0.99), ("lambda_", float, 0.95),
    ("lr", float, 3e-4), ("epochs", int, 4), ("eps_clip", float, 0.1), ("bc_loss_weight", float, 2),
    ("bc_decay_rate", float, 0.998), ("min_bc_weight", float, 0.3), ("use_entropy", bool, True),
    ("entropy_coef", float, 0.005), ("use_grad_clip", bool, True), ("grad_norm_clip", float, 10.0),
    ("activate_fun", str, "relu"), ("normalize_rewards", bool, True), ("normalize_advantages", bool, True),
    ("expert", bool, False), ("load_models", bool, False), ("save_models", bool, True), ("save_buffers", bool, False),
)


def add_gail_args(parser):
    for name, typ, default in _FLAGS:
        parser.add_argument("--" + name, type=typ, default=default)
    return parser
