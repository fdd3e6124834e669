"""Driver-level flags, same names/defaults as the reference's default_config.py:4-23."""
import torch as th

_FLAGS = (
    ("t_max", int, 40000), ("t_start", int, 0), ("test_nepisode", int, 10), ("test_interval", int, 2000),
    ("load_models", bool, False), ("checkpoint_path", str, ""), ("buffers_path", str, ""),
    ("offline_models_path", str, ""), ("test_models", bool, False), ("train_models", bool, True),
    ("save_models", bool, False), ("save_buffers", bool, False), ("log_tag", str, "default_log_tag"),
    ("folder", str, "default_folder"), ("expert", bool, False),
)


def add_default_args(parser):
    parser.add_argument("--device", type=str, default="cuda" if th.cuda.is_available() else "cpu")
    for name, typ, default in _FLAGS:
        parser.add_argument("--" + name, type=typ, default=default)
    parser.add_argument("--offline", default=False)
    return parser
