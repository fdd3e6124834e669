"""CPU-side checks of the C-ABI library: it loads and exports every symbol include/marl_amd.h declares
(no compute without a GPU), and the product refuses to run without it."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "marl_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ec_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_entry_points():
    syms = _declared_symbols()
    for s in ("ec_create", "ec_destroy", "ec_reset", "ec_observe", "ec_step", "ec_gae", "ec_last_error"):
        assert s in syms


def test_library_exports_every_declared_symbol():
    from marl_amd import _abi
    if not os.path.exists(_abi.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    L = C.CDLL(_abi.LIB_PATH)
    for s in _declared_symbols():
        assert hasattr(L, s), s
    L.ec_version.restype = C.c_char_p
    assert b"gfx950" in L.ec_version()


def test_struct_sizes_match_header():
    from marl_amd._abi import EcConfig, EcTopology
    # 4 i32 + 14 f64 + 2 i64 + 6 i32
    assert C.sizeof(EcConfig) == 16 + 14 * 8 + 16 + 24
    assert C.sizeof(EcTopology) == 8 * 8


def test_no_cpu_fallback():
    import torch as th
    from marl_amd.envs.edge_computing.multihop.edge_computing_env import BatchedEdgeComputingEnv
    from tests.util import make_args
    if th.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        BatchedEdgeComputingEnv(make_args(), n_envs=2)


def test_product_never_imports_oracle():
    bad = []
    for dp, _, fns in os.walk(os.path.join(ROOT, "marl_amd")):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, fn)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M) or "libec_oracle" in src:
                    bad.append(os.path.join(dp, fn))
    assert not bad, bad
