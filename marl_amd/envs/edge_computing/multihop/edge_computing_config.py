"""Environment flags: same names, types and defaults as the reference's
envs/edge_computing/multihop/edge_computing_config.py:1-49, plus the `ec_*`
knobs of the batched HIP environment (all optional)."""

# (flag, type, default) — kept as a table so the two flag sets can be diffed at a glance.
# Quirks preserved on purpose: `deadline` and `beta` are declared int but default to floats
# (edge_computing_config.py:19,47); `cpu_core_list` is type=list (:22).
_REFERENCE_FLAGS = (
    ("env_name", str, "edge_computing"),
    ("episode_limit", int, 100),
    ("time_slot", float, 0.5),
    ("mini_time_slot", float, 0.05),
    ("mini_time_slot_num", int, 10),
    ("edge_node_num", int, 10),
    ("task_probability_min", float, 0),
    ("task_probability_max", float, 1.0),
    ("task_size_min", int, 1000),
    ("task_size_max", int, 2000),
    ("task_complexity_min", int, 800),
    ("task_complexity_max", int, 2400),
    ("deadline", int, 3.5),
    ("single_core_cpu_capacity", int, 3),
    ("cpu_core_list", list, [4, 8, 12, 16, 20, 24, 28, 32]),
    ("transmission_rate_min", int, 10),
    ("transmission_rate_max", int, 40),
    ("execution_failure_rate_min", float, 0),
    ("execution_failure_rate_max", float, 0.3),
    ("transmission_failure_rate_min", float, 0),
    ("transmission_failure_rate_max", float, 0.1),
    ("task_drop_penalty", float, -1),
    ("task_failure_penalty", float, -1),
    ("task_success_reward", float, 1),
    ("task_hop_penalty", float, -0.3),
    ("edge_node_seed", int, 500),
    ("task_seed", int, 100),
    ("failure_seed", int, 200),
    ("seed", int, 300),
    ("rl_queue_coeff", float, 1.0),
    ("link_seed", int, 601),
    ("k", int, 4),
    ("beta", int, 1.0),
)

_HIP_FLAGS = (
    ("ec_cap_buffer", int, 0),   # ring capacities (records per node); 0 = bound derived from the config
    ("ec_cap_recv", int, 0),
    ("ec_cap_wait", int, 0),
    ("ec_max_ticks", int, 0),
    ("ec_hop_hist_len", int, 32),
    ("ec_env_id_base", int, 0),  # first env id of this rank's shard (seeds are task_seed + env id + node)
    ("ec_graph_rollout", int, 0),  # 1: ParallelRunner replays one hipGraph per episode
    ("ec_torch_update", int, 0),   # 1: PPOAgent.train keeps the torch-autograd update instead of the fused kernels
    ("ec_kernel_rollout", int, 1),  # ParallelRunner + PPO: the whole episode (Actor + env) in one launch when the env supports it
    ("ec_check_buffer_flags", int, 0),  # 1: verify the buffer's whole-episode flags on every PPO update (first update always)
    ("ec_graph_update", int, 1),   # DGAIL: replay one hipGraph per minibatch step of the policy/value update (single process)
    # generate_graph2 (network_graph.py:267-307): a GraphML topology instead of the synthetic ring + chords; the
    # reference reads args.topology_file with getattr and selects the loader by editing edge_node.py:38-39
    ("topology_file", str, ""),
)


def add_edge_computing_env_args(parser):
    for name, typ, default in _REFERENCE_FLAGS + _HIP_FLAGS:
        parser.add_argument("--" + name, type=typ, default=default)
    return parser
