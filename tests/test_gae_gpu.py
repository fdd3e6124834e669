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


# (33, 250, 40), (3, 37, 64), (2, 3000, 1): the env's slab is cut into chunks of steps (aligned / 16 steps / 1024 steps);
# (5, 7, 3), (9, 33, 5): T * N not a multiple of 4, the 4-byte path; (3, 50, 100), (2, 19, 65): N > 64, one lane per chain;
# (6000, 20, 10): more envs than the grid holds waves
@pytest.mark.parametrize("shape", [(1, 1, 1), (5, 7, 3), (9, 33, 5), (64, 100, 10), (4096, 100, 10), (33, 250, 40),
                                   (3, 37, 64), (2, 3000, 1), (3, 50, 100), (2, 19, 65), (6000, 20, 10)])
def test_gae_vs_oracle(oracle_lib, shape):
    B, T, N = shape
    rng = np.random.RandomState(B + T)
    r, v, nv = (rng.standard_normal((B, T, N, 1)).astype(np.float32) for _ in range(3))
    m = (rng.random_sample((B, T, N, 1)) > 0.05).astype(np.float32)
    m[:, -1] = 0
    if N == 40:
        m *= rng.random_sample(m.shape).astype(np.float32)  # (any float is a legal mask: the two roundings of gl * adv * m)
    adv = _run(r, v, nv, m)
    want = oracle_lib.gae(r, v, nv, m, 0.99, 0.95)
    assert np.array_equal(adv[..., 0], want)


def test_gae_empty():
    adv = _run(*(np.zeros((0, 5, 3, 1), np.float32) for _ in range(4)))
    assert adv.shape == (0, 5, 3, 1)


def test_gae_unaligned_views(oracle_lib):
    """The 16-byte path needs aligned bases: a view that starts 4 bytes into its storage takes the 4-byte path, same bits."""
    import torch as th
    from marl_amd.utils.advantage_utils import gae_device
    B, T, N = 7, 100, 10
    rng = np.random.RandomState(3)
    host = [rng.standard_normal((B, T, N, 1)).astype(np.float32) for _ in range(3)]
    host.append((rng.random_sample((B, T, N, 1)) > 0.05).astype(np.float32))
    dev = th.device("cuda:0")
    views = []
    for k, x in enumerate(host):
        st = th.zeros(x.size + 1, device=dev)
        vw = st[1:] if k % 2 == 0 else st[:-1]
        vw.copy_(th.as_tensor(x.ravel(), device=dev))
        views.append(vw.view(B, T, N, 1))
    adv = gae_device(*views, 0.99, 0.95).cpu().numpy()
    assert np.array_equal(adv[..., 0], oracle_lib.gae(*host, 0.99, 0.95))
