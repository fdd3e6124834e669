"""ec_gae (through the C ABI) vs the reference's get_gae golden vectors and the oracle."""
import ctypes as C

import numpy as np
import pytest

from tests.util import load_golden

pytestmark = pytest.mark.gpu


def _run(r, v, nv, m, gamma=0.99, lam=0.95):
    import torch as th
    from marl_amd.utils.advantage_utils import gae_device
    dev = th.device("cuda:0")
    ts = [th.as_tensor(np.ascontiguousarray(x, dtype=np.float32), device=dev) for x in (r, v, nv, m)]
    return gae_device(*ts, gamma, lam).cpu().numpy()


def test_gae_golden_bit_exact():
    g = load_golden("gae")
    for tag in ("a", "b"):
        adv = _run(g[tag + "_r"], g[tag + "_v"], g[tag + "_nv"], g[tag + "_m"])
        assert adv.shape == g[tag + "_adv"].shape
        assert np.array_equal(adv, g[tag + "_adv"]), tag


@pytest.mark.parametrize("shape", [(1, 1, 1), (5, 7, 3), (64, 100, 10), (4096, 100, 10), (33, 250, 40)])
def test_gae_vs_oracle(oracle_lib, shape):
    B, T, N = shape
    rng = np.random.RandomState(B + T)
    r, v, nv = (rng.standard_normal((B, T, N, 1)).astype(np.float32) for _ in range(3))
    m = (rng.random_sample((B, T, N, 1)) > 0.05).astype(np.float32)
    m[:, -1] = 0
    adv = _run(r, v, nv, m)
    want = oracle_lib.gae(r, v, nv, m, 0.99, 0.95)
    assert np.array_equal(adv[..., 0], want)


def test_gae_empty():
    adv = _run(*(np.zeros((0, 5, 3, 1), np.float32) for _ in range(4)))
    assert adv.shape == (0, 5, 3, 1)
