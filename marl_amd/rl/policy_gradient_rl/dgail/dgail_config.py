"""DGAIL flags, same names/defaults as the reference's rl/policy_gradient_rl/dgail/dgail_config.py:1-50."""

_FLAGS = (
    ("name", str, "dgail"), ("algo_type", str, "rl"), ("runner", str, "dgail_runner"),
    ("max_expert_buffer_size", int, 20000), ("n_threads", int, 1), ("hidden_dim", int, 64),
    ("batch_size_run", int, 100), ("ppo_batch_size", int, 64), ("gamma", float, 0.99), ("lambda_", float, 0.95),
    ("policy_lr", float, 2e-5), ("value_lr", float, 2e-5), ("ppo_epochs", int, 4), ("clip_param", float, 0.2),
    ("value_loss_coef", float, 0.2), ("entropy_coef", float, 8e-4), ("discriminator_lr", float, 1e-4),
    ("gail_discriminator_lr", float, 1e-4), ("grad_norm_clip", float, 1.0), ("load_models", bool, False),
    ("save_models", bool, True), ("switch_to_gail_at_step", int, -1), ("imit_weight_start", float, 1.5),
    ("imit_weight_end", float, 0.8), ("use_simplified_reward", bool, True), ("bc_loss_weight", float, 0.3),
    ("bc_decay_rate", float, 0.995), ("min_bc_weight", float, 0.05), ("batch_expert_transitions", int, 64),
)


def add_dgail_args(parser):
    for name, typ, default in _FLAGS:
        parser.add_argument("--" + name, type=typ, default=default)
    parser.add_argument("--activate_fun", type=str, default="relu", choices=["relu", "tanh", "leaky_relu"])
    return parser
