"""ctypes view of include/marl_amd.h and the loader for libmarl_amd.so.

The product path has no CPU fallback: if the HIP library is missing or fails to
load, `lib()` raises.  (The CPU oracle under oracle/ is test infrastructure and
is never imported from here.)
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MARL_AMD_LIB") or os.path.join(_HERE, "csrc", "libmarl_amd.so")

EC_OK = 0
EC_ST_BAD_ACTION = 1
EC_ST_BUFFER_OVERFLOW = 2
EC_ST_RECV_OVERFLOW = 4
EC_ST_WAIT_OVERFLOW = 8
EC_ST_TICK_OVERFLOW = 16
EC_ST_HOP_OVERFLOW = 32


class EcConfig(C.Structure):
    _fields_ = [
        ("n_nodes", C.c_int32),
        ("episode_limit", C.c_int32),
        ("mini_slots", C.c_int32),
        ("k", C.c_int32),
        ("mini_time_slot", C.c_double),
        ("deadline", C.c_double),
        ("task_size_min", C.c_double),
        ("task_size_max", C.c_double),
        ("task_complexity_min", C.c_double),
        ("task_complexity_max", C.c_double),
        ("task_drop_penalty", C.c_double),
        ("task_failure_penalty", C.c_double),
        ("task_success_reward", C.c_double),
        ("task_hop_penalty", C.c_double),
        ("execution_failure_rate_max", C.c_double),
        ("task_probability_max", C.c_double),
        ("core_norm", C.c_double),
        ("rl_queue_coeff", C.c_double),
        ("task_seed", C.c_int64),
        ("failure_seed", C.c_int64),
        ("hop_hist_len", C.c_int32),
        ("cap_buffer", C.c_int32),
        ("cap_recv", C.c_int32),
        ("cap_wait", C.c_int32),
        ("max_ticks", C.c_int32),
        ("reserved", C.c_int32),
    ]


class EcTopology(C.Structure):
    _fields_ = [
        ("cores", C.POINTER(C.c_int32)),
        ("task_p", C.POINTER(C.c_double)),
        ("efr", C.POINTER(C.c_double)),
        ("cpu_capacity", C.POINTER(C.c_double)),
        ("queue_len", C.POINTER(C.c_double)),
        ("rate", C.POINTER(C.c_double)),
        ("tfr", C.POINTER(C.c_double)),
        ("adj", C.POINTER(C.c_uint8)),
    ]


def config_from_args(args, hop_hist_len=32):
    """ec_config from the reference's argparse namespace
    (envs/edge_computing/multihop/edge_computing_config.py:1-49)."""
    c = EcConfig()
    c.n_nodes = int(args.edge_node_num)
    c.episode_limit = int(args.episode_limit)
    c.mini_slots = int(args.mini_time_slot_num)
    c.k = int(args.k)
    c.mini_time_slot = float(args.mini_time_slot)
    c.deadline = float(args.deadline)
    c.task_size_min = float(args.task_size_min)
    c.task_size_max = float(args.task_size_max)
    c.task_complexity_min = float(args.task_complexity_min)
    c.task_complexity_max = float(args.task_complexity_max)
    c.task_drop_penalty = float(args.task_drop_penalty)
    c.task_failure_penalty = float(args.task_failure_penalty)
    c.task_success_reward = float(args.task_success_reward)
    c.task_hop_penalty = float(args.task_hop_penalty)
    c.execution_failure_rate_max = float(args.execution_failure_rate_max)
    c.task_probability_max = float(args.task_probability_max)
    c.core_norm = float(args.cpu_core_list[-1])
    c.rl_queue_coeff = float(args.rl_queue_coeff)
    c.task_seed = int(args.task_seed)
    c.failure_seed = int(args.failure_seed)
    c.hop_hist_len = int(hop_hist_len)
    c.cap_buffer = int(getattr(args, "ec_cap_buffer", 0))
    c.cap_recv = int(getattr(args, "ec_cap_recv", 0))
    c.cap_wait = int(getattr(args, "ec_cap_wait", 0))
    c.max_ticks = int(getattr(args, "ec_max_ticks", 0))
    return c


def topology_from_tables(tables):
    """ec_topology over the numpy arrays of a StaticTables; returns (struct, keepalive)."""
    N = len(tables.cores)
    keep = dict(
        cores=np.ascontiguousarray(tables.cores, dtype=np.int32),
        task_p=np.ascontiguousarray(tables.p, dtype=np.float64),
        efr=np.ascontiguousarray(tables.efr, dtype=np.float64),
        cpu_capacity=np.ascontiguousarray(tables.cap, dtype=np.float64),
        queue_len=np.ascontiguousarray(tables.qlen_max, dtype=np.float64),
        rate=np.ascontiguousarray(tables.rate, dtype=np.float64).reshape(N, N),
        tfr=np.ascontiguousarray(tables.tfr, dtype=np.float64).reshape(N, N),
        adj=np.ascontiguousarray(tables.adj, dtype=np.uint8).reshape(N, N + 1),
    )
    t = EcTopology()
    t.cores = keep["cores"].ctypes.data_as(C.POINTER(C.c_int32))
    for name in ("task_p", "efr", "cpu_capacity", "queue_len", "rate", "tfr"):
        setattr(t, name, keep[name].ctypes.data_as(C.POINTER(C.c_double)))
    t.adj = keep["adj"].ctypes.data_as(C.POINTER(C.c_uint8))
    return t, keep


_lib = None


class HipLibraryMissing(RuntimeError):
    pass


def lib():
    """Load libmarl_amd.so (built in-tree by __graft_entry__.build()). Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryMissing(
            "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(marl_amd has no CPU fallback)" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, i64, i32, u64 = C.c_void_p, C.c_int64, C.c_int32, C.c_uint64
    L.ec_create.argtypes = [C.POINTER(EcConfig), C.POINTER(EcTopology), i64, i64, C.c_int, C.POINTER(vp)]
    L.ec_destroy.argtypes = [vp]
    L.ec_reset.argtypes = [vp, vp]
    L.ec_observe.argtypes = [vp, vp, vp, vp]
    L.ec_step.argtypes = [vp] + [vp] * 8 + [vp]
    L.ec_step_random_policy.argtypes = [vp] + [vp] * 8 + [u64, u64, vp, vp]
    L.ec_rollout_random_policy.argtypes = [vp, i32, i32] + [vp] * 8 + [u64, u64, vp, vp]
    L.ec_rollout_mlp_policy.argtypes = [vp, i32] + [vp] * 18
    L.ec_rollout_mlp_expert_policy.argtypes = [vp, i32] + [vp] * 21
    L.ec_random_policy.argtypes = [vp, i64, i32, u64, u64, vp, vp, vp]
    L.ec_expert_policy.argtypes = [vp, vp, vp, vp]
    L.ec_status.argtypes = [vp, C.POINTER(C.c_uint32), C.c_int, vp]
    L.ec_debug_dump.argtypes = [vp, i64, vp, i64, C.POINTER(i64)]
    L.ec_debug_rng_cursors.argtypes = [vp, vp]
    L.ec_info.argtypes = [vp, C.POINTER(i64), i32]
    L.ec_debug_live_tasks.argtypes = [vp, C.POINTER(i64)]
    L.ec_gae.argtypes = [vp, vp, vp, vp, vp, i64, i64, i64, C.c_float, C.c_float, vp]
    f32 = C.c_float
    L.ec_policy_forward.argtypes = [vp] * 9 + [i64, i32, i32, i32, vp, vp, vp, vp]
    L.ec_gail_reward.argtypes = [vp] * 9 + [i64, i32, i32, f32, f32, vp, vp, vp]
    L.ec_diffusion_reward.argtypes = [vp] * 14 + [i64, i32, i32, f32, vp, vp, vp, vp, vp]
    L.ec_diffusion_grad.argtypes = [vp, i32] + [vp] * 11 + [i64, i32, i32, f32, vp, vp, vp]
    L.ec_diffusion_grad_bce.argtypes = [vp, i32] + [vp] * 11 + [i32, i32, f32, i64, i32, i32, f32, vp, vp, vp]
    L.ec_diffusion_reward_from_loss.argtypes = [vp, i32] + [vp] * 12 + [f32, f32, i64, i32, i32, f32, vp, vp, vp]
    L.ec_ppo_actor_grad.argtypes = [vp] * 12 + [i64, i32, i32, i32, f32, f32, f32, vp, vp, vp, vp, vp]
    L.ec_ppo_critic_grad.argtypes = [vp] * 9 + [i64, i32, i32, f32, vp, vp, vp, vp]
    L.ec_ppo_actor_grad_ex.argtypes = [vp] * 12 + [i64, i32, i32, i32, f32, f32, f32, vp, vp, vp, vp, vp, i32, vp]
    L.ec_ppo_critic_grad_ex.argtypes = [vp] * 9 + [i64, i32, i32, f32, vp, vp, vp, vp, vp]
    L.ec_gail_disc_grad.argtypes = [vp] * 8 + [i64, i32, i32, i32, f32, vp, vp, vp, vp]
    L.ec_expert_mix.argtypes = [vp] * 5 + [i64, i32, i32, i64] + [vp] * 6
    L.ec_expert_append_episode.argtypes = [vp] * 5 + [i32, i64, i32, i32, i64] + [vp] * 6
    L.ec_adam_clip_step.argtypes = [i32] + [vp] * 13 + [vp]
    L.ec_expert_append_episode_workspace.argtypes = [i32, i64]
    L.ec_expert_append_episode_workspace.restype = i64
    u64 = C.c_uint64
    L.ec_bn_col_stats.argtypes = [vp, i64, i32, i32, vp, vp]
    L.ec_linear_wgrad.argtypes = [vp, vp, i64, i32, i32, i32, vp, vp]
    L.ec_linear_fwd.argtypes = [vp, vp, vp, i64, i32, i32, vp, vp]
    L.ec_linear_fwd_stats.argtypes = [vp, vp, vp, i64, i32, i32, vp, vp, i32, vp, vp]
    L.ec_disc_rows.argtypes = [vp, vp, i64, i32, i32, i32, vp, vp]
    L.ec_linear_fwd_bn.argtypes = [vp, vp, vp, u64, f32, vp, vp, i64, i32, i32, vp, i32, vp, vp]
    L.ec_row_moments.argtypes = [vp, i64, i32, i32, vp, vp]
    L.ec_linear2_fwd_bn.argtypes = [vp, vp, vp, vp, vp, u64, f32, vp, vp, i64, i32, i32, i32, vp, i32, vp, vp]
    L.ec_bn_finalize.argtypes = [vp, i32, vp, i64, i32, vp, vp, f32, f32] + [vp] * 9
    L.ec_bn_bwd_coeffs.argtypes = [vp, i32, vp, vp, vp, i64, i32] + [vp] * 6
    L.ec_bn_relu_dropout_fwd.argtypes = [vp, vp, vp, vp, u64, f32, i64, i32, vp, vp]
    L.ec_bn_relu_dropout_bwd_sums.argtypes = [vp, vp, vp, vp, vp, u64, f32, i64, i32, i32, vp, vp]
    L.ec_bn_relu_dropout_bwd_dx.argtypes = [vp, vp, vp, vp, vp, u64, f32, vp, vp, vp, i64, i32, vp, vp]
    L.ec_store_transition.argtypes = [vp] * 7 + [i64, i32, i32, i32, i64, i64] + [vp] * 8
    L.ec_store_episode.argtypes = [vp] * 9 + [i64, i32, i32, i32, i64] + [vp] * 10
    L.ec_store_episode_static.argtypes = [vp] * 9 + [i64, i32, i32, i32, i64] + [vp] * 8
    L.ec_debug_store_episode_static_scratch.argtypes = [i32, C.POINTER(vp), C.POINTER(i64), vp]
    L.ec_debug_store_episode_static_scratch.restype = C.c_int
    for fn in ("ec_create", "ec_destroy", "ec_reset", "ec_observe", "ec_step", "ec_random_policy", "ec_status",
               "ec_debug_dump", "ec_debug_rng_cursors", "ec_debug_live_tasks", "ec_info", "ec_gae",
               "ec_policy_forward", "ec_gail_reward", "ec_diffusion_reward", "ec_diffusion_grad", "ec_expert_policy",
               "ec_ppo_actor_grad", "ec_ppo_critic_grad", "ec_ppo_actor_grad_ex", "ec_ppo_critic_grad_ex", "ec_gail_disc_grad", "ec_store_transition", "ec_bn_relu_dropout_fwd", "ec_bn_relu_dropout_bwd_sums", "ec_bn_relu_dropout_bwd_dx", "ec_expert_mix", "ec_store_episode", "ec_store_episode_static",
               "ec_step_random_policy", "ec_rollout_random_policy", "ec_rollout_mlp_policy",
               "ec_rollout_mlp_expert_policy", "ec_expert_append_episode", "ec_adam_clip_step", "ec_bn_col_stats", "ec_diffusion_grad_bce", "ec_linear_wgrad", "ec_bn_finalize", "ec_bn_bwd_coeffs", "ec_linear_fwd", "ec_linear_fwd_bn", "ec_linear2_fwd_bn", "ec_row_moments", "ec_disc_rows", "ec_linear_fwd_stats",
               "ec_diffusion_reward_from_loss"):
        getattr(L, fn).restype = C.c_int
    L.ec_last_error.restype = C.c_char_p
    L.ec_version.restype = C.c_char_p
    _lib = L
    return L


class EcError(RuntimeError):
    pass


def check(rc, what):
    if rc != EC_OK:
        raise EcError("%s failed (%d): %s" % (what, rc, lib().ec_last_error().decode()))
