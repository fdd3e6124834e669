"""ctypes wrapper around oracle/libec_oracle.so — TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg;
never from marl_amd/.  See ec_oracle.c for what the oracle is and how it is pinned.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from marl_amd._abi import EcConfig, EcTopology, topology_from_tables

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libec_oracle.so")


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        vp, i64 = C.c_void_p, C.c_int64
        L.oc_create.argtypes = [C.POINTER(EcConfig), C.POINTER(EcTopology), i64, i64, C.c_int, C.POINTER(vp)]
        L.oc_destroy.argtypes = [vp]
        L.oc_reset.argtypes = [vp]
        L.oc_observe.argtypes = [vp, vp, vp]
        L.oc_step.argtypes = [vp, vp, vp, vp, vp, vp, vp]
        L.oc_debug_dump.argtypes = [vp, i64, vp, i64, C.POINTER(i64)]
        L.oc_debug_rng_cursors.argtypes = [vp, vp]
        L.oc_live_tasks.argtypes = [vp]
        L.oc_live_tasks.restype = i64
        L.oc_rng_doubles.argtypes = [C.c_uint32, i64, vp]
        L.oc_rng_doubles.restype = None
        L.oc_rng_task_draws.argtypes = [C.c_uint32, C.c_double, i64] + [C.c_double] * 4 + [vp]
        L.oc_rng_task_draws.restype = None
        L.oc_gae.argtypes = [vp] * 5 + [i64] * 3 + [C.c_float, C.c_float]
        L.oc_random_policy.argtypes = [vp, i64, C.c_int32, C.c_uint64, C.c_uint64, vp]
        L.oc_expert_policy.argtypes = [vp, vp, vp]
        L.oc_rollout_random.argtypes = [vp, i64, C.c_uint64, C.POINTER(C.c_double), C.POINTER(i64)]
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def rng_doubles(seed, n):
    out = np.empty(n, dtype=np.float64)
    lib().oc_rng_doubles(int(seed) & 0xFFFFFFFF, n, _p(out))
    return out


def rng_task_draws(seed, p, n, smin, smax, cmin, cmax):
    out = np.empty((n, 4), dtype=np.float64)
    lib().oc_rng_task_draws(int(seed) & 0xFFFFFFFF, p, n, smin, smax, cmin, cmax, _p(out))
    return out


def random_policy(avail, seed, counter):
    a = np.ascontiguousarray(avail, dtype=np.int32)
    rows, A = a.reshape(-1, a.shape[-1]).shape
    out = np.empty(rows, dtype=np.int64)
    lib().oc_random_policy(_p(a), rows, A, int(seed), int(counter), _p(out))
    return out.reshape(a.shape[:-1])


def gae(r, v, nv, m, gamma, lam):
    B, T, N = r.shape[:3]
    arrs = [np.ascontiguousarray(x, dtype=np.float32).reshape(B, T, N) for x in (r, v, nv, m)]
    adv = np.empty((B, T, N), dtype=np.float32)
    lib().oc_gae(*[_p(a) for a in arrs], _p(adv), B, T, N, np.float32(gamma), np.float32(gamma * lam))
    return adv


class OracleEnv:
    """Batch of reference-semantics environments on the CPU."""

    def __init__(self, cfg, tables, n_envs, env_id_base=0, n_threads=1):
        self.cfg = cfg
        self.N = cfg.n_nodes
        self.H = cfg.hop_hist_len if cfg.hop_hist_len > 0 else 32
        self.B = int(n_envs)
        topo, self._keep = topology_from_tables(tables)
        self._h = C.c_void_p()
        rc = lib().oc_create(C.byref(cfg), C.byref(topo), self.B, int(env_id_base), int(n_threads), C.byref(self._h))
        assert rc == 0

    def close(self):
        if self._h:
            lib().oc_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        lib().oc_reset(self._h)

    def observe(self):
        obs = np.empty((self.B, self.N, 9), dtype=np.float64)
        avail = np.empty((self.B, self.N, self.N + 1), dtype=np.int32)
        lib().oc_observe(self._h, _p(obs), _p(avail))
        return obs, avail

    def step(self, actions, want_hops=True):
        a = np.ascontiguousarray(actions, dtype=np.int64).reshape(self.B, self.N)
        reward = np.empty(self.B, dtype=np.float64)
        term = np.empty(self.B, dtype=np.uint8)
        counters = np.empty((self.B, 4), dtype=np.int32)
        ctime = np.empty(self.B, dtype=np.float64)
        hops = np.empty((self.B, self.H), dtype=np.int32) if want_hops else None
        bad = lib().oc_step(self._h, _p(a), _p(reward), _p(term), _p(counters), _p(ctime), _p(hops))
        return dict(reward=reward, terminated=term, counters=counters, completion_time=ctime, hop_hist=hops, bad=bad)

    def expert_policy(self, avail):
        """OptimalAgent.get_current_optimal_action for every env (optimal_agent.py:102-120)"""
        a = np.ascontiguousarray(avail, dtype=np.int32).reshape(self.B, self.N, self.N + 1)
        out = np.empty((self.B, self.N), dtype=np.int64)
        lib().oc_expert_policy(self._h, _p(a), _p(out))
        return out

    def step_fast(self, actions_ptr_arr):
        """bench leg: no output buffers beyond reward"""
        lib().oc_step(self._h, _p(actions_ptr_arr), None, None, None, None, None)

    def dump(self, env, max_rows=4096):
        rows = np.empty((max_rows, 11), dtype=np.float64)
        n = C.c_int64()
        lib().oc_debug_dump(self._h, int(env), _p(rows), max_rows, C.byref(n))
        assert n.value <= max_rows
        return rows[: n.value].copy()

    def rng_cursors(self):
        cur = np.empty((self.B, self.N, 2), dtype=np.uint64)
        lib().oc_debug_rng_cursors(self._h, _p(cur))
        return cur

    def rollout_random(self, n_steps, seed=0):
        """bench.py cpu_baseline leg: observe -> random-valid policy -> step, lock-step, reset per episode"""
        rs, fin = C.c_double(), C.c_int64()
        lib().oc_rollout_random(self._h, int(n_steps), int(seed), C.byref(rs), C.byref(fin))
        return rs.value, fin.value

    def live_tasks(self):
        return int(lib().oc_live_tasks(self._h))
