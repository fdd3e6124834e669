"""ec_store_episode_static (the entry point that allocates nothing) against ec_store_episode, through the C ABI:
two calls in a row on one GPU, the second with a smaller input, every output compared bit for bit."""
import ctypes as C

import pytest
import torch as th

pytestmark = pytest.mark.gpu


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _episode(B, T, N, O, A, seed, dev):
    g = th.Generator(device="cpu").manual_seed(seed)
    h = dict(
        obs0=th.randn(B, N, O, generator=g),
        avail0=th.randint(0, 2, (B, N, A), generator=g, dtype=th.int32),
        obs=th.randn(T, B, N, O, generator=g),
        avail=th.randint(0, 2, (T, B, N, A), generator=g, dtype=th.int32),
        actions=th.randint(0, A, (T, B, N), generator=g, dtype=th.int64),
        reward=th.randn(T, B, generator=g) * 7,
        term=th.zeros(T, B, dtype=th.uint8),
        counters=th.randint(0, 40, (T, B, 4), generator=g, dtype=th.int32),
        ctime=th.rand(T, B, generator=g) * 30,
    )
    h["term"][-1] = 1
    return {k: v.to(dev).contiguous() for k, v in h.items()}


def _buffers(B, T, N, O, A, dev):
    # filled with a sentinel: every element must be overwritten by the call
    return [th.full((B, T, N, O), -7.0, device=dev), th.full((B, T, N, A), -7, dtype=th.int32, device=dev),
            th.full((B, T, N, 1), -7, dtype=th.int64, device=dev), th.full((B, T, N, 1), -7.0, device=dev),
            th.full((B, T, N, 1), -7.0, device=dev), th.full((B, T, N, O), -7.0, device=dev)]


def _run(static, h, B, T, N, O, A, totals, dev):
    from marl_amd import _abi
    L = _abi.lib()
    bufs = _buffers(B, T, N, O, A, dev)
    stream = C.c_void_p(th.cuda.current_stream(dev).cuda_stream)
    head = [_p(h["obs0"]), _p(h["avail0"]), _p(h["obs"]), _p(h["avail"]), _p(h["actions"]), _p(h["reward"]),
            _p(h["term"]), _p(h["counters"]), _p(h["ctime"]), B, N, O, A, T] + [_p(b) for b in bufs] + [_p(totals)]
    if static:
        _abi.check(L.ec_store_episode_static(*head, stream), "ec_store_episode_static")
    else:
        partial = th.zeros(T, 6, dtype=th.float64, device=dev)
        done = th.zeros(1, dtype=th.int32, device=dev)
        _abi.check(L.ec_store_episode(*head, _p(partial), _p(done), stream), "ec_store_episode")
    th.cuda.synchronize(dev)
    return bufs


def test_static_scratch_matches_workspace_variant_twice():
    dev = th.device("cuda:0")
    N, O = 10, 9
    A = N + 1
    tot_s = th.tensor([1.5, -2.25, 3, 4, 5, 6], dtype=th.float64, device=dev)
    tot_w = tot_s.clone()
    # second call smaller in both the env count and the episode length: stale partial rows of the first call lie
    # behind the rows the second one uses
    for call, (B, T) in enumerate([(96, 100), (17, 23)]):
        h = _episode(B, T, N, O, A, seed=10 + call, dev=dev)
        got = _run(True, h, B, T, N, O, A, tot_s, dev)
        want = _run(False, h, B, T, N, O, A, tot_w, dev)
        for i, (g, w) in enumerate(zip(got, want)):
            assert th.equal(g, w), (call, i)
            assert not (g == -7).all()
        assert th.equal(tot_s, tot_w), (call, tot_s, tot_w)
        if call == 0:  # and the arithmetic itself: the integer totals are the running sums over the history
            want_counts = th.tensor([3., 4, 5, 6], dtype=th.float64, device=dev) + h["counters"].sum((0, 1)).double()
            assert th.equal(tot_s[2:], want_counts)


def test_static_entry_point_allocates_no_device_memory_and_survives_a_poisoned_scratch():
    """The property the entry point exists for: a call leaves the device's free memory where it was (no hipMalloc behind
    it; the inputs and outputs are made beforehand and nothing goes through torch's allocator in between).  And the
    hipMemsetAsync that guards the arrival counter: a scratch filled with 0xFF -- counter 0xFFFFFFFF, partial sums NaN --
    gives the same buffers and totals as the workspace variant."""
    from marl_amd import _abi
    dev = th.device("cuda:0")
    L = _abi.lib()
    N, O, B, T = 10, 9, 64, 50
    A = N + 1
    h = _episode(B, T, N, O, A, seed=21, dev=dev)
    tot_s = th.zeros(6, dtype=th.float64, device=dev)
    tot_w = th.zeros(6, dtype=th.float64, device=dev)
    want = _run(False, h, B, T, N, O, A, tot_w, dev)
    _run(True, h, B, T, N, O, A, tot_s, dev)                        # warm call: the symbol's address is looked up here
    bufs = _buffers(B, T, N, O, A, dev)
    stream = C.c_void_p(th.cuda.current_stream(dev).cuda_stream)
    head = [_p(h["obs0"]), _p(h["avail0"]), _p(h["obs"]), _p(h["avail"]), _p(h["actions"]), _p(h["reward"]),
            _p(h["term"]), _p(h["counters"]), _p(h["ctime"]), B, N, O, A, T] + [_p(b) for b in bufs] + [_p(tot_s)]
    tot_s.zero_()
    ptr, nbytes = C.c_void_p(), C.c_int64()
    _abi.check(L.ec_debug_store_episode_static_scratch(0xFF, C.byref(ptr), C.byref(nbytes), stream), "scratch hook")
    assert ptr.value and nbytes.value == 256 + 4096 * 48
    th.cuda.synchronize(dev)
    free0 = th.cuda.mem_get_info(dev)[0]
    _abi.check(L.ec_store_episode_static(*head, stream), "ec_store_episode_static")
    th.cuda.synchronize(dev)
    free1 = th.cuda.mem_get_info(dev)[0]
    assert free1 == free0, (free0, free1)
    for i, (g, w) in enumerate(zip(bufs, want)):
        assert th.equal(g, w), i
    assert th.equal(tot_s, tot_w) and bool(th.isfinite(tot_s).all())


def test_static_scratch_statistics_only_and_limit():
    from marl_amd import _abi
    dev = th.device("cuda:0")
    B, T, N, O = 8, 12, 10, 9
    A = N + 1
    h = _episode(B, T, N, O, A, seed=3, dev=dev)
    L = _abi.lib()
    stream = C.c_void_p(th.cuda.current_stream(dev).cuda_stream)
    tot_s = th.zeros(6, dtype=th.float64, device=dev)
    tot_w = th.zeros(6, dtype=th.float64, device=dev)
    null = [_p(None)] * 6
    hist = [_p(None), _p(None), _p(None), _p(None), _p(None), _p(h["reward"]), _p(h["term"]), _p(h["counters"]),
            _p(h["ctime"])]
    _abi.check(L.ec_store_episode_static(*hist, B, N, O, A, T, *null, _p(tot_s), stream), "static")
    partial = th.zeros(T, 6, dtype=th.float64, device=dev)
    done = th.zeros(1, dtype=th.int32, device=dev)
    _abi.check(L.ec_store_episode(*hist, B, N, O, A, T, *null, _p(tot_w), _p(partial), _p(done), stream), "workspace")
    th.cuda.synchronize(dev)
    assert th.equal(tot_s, tot_w)
    assert tot_s[2].item() == float(h["counters"][..., 0].sum().item())
    # beyond the static array's size: a clear error, nothing launched
    rc = L.ec_store_episode_static(*hist, B, N, O, A, 4097, *null, _p(tot_s), stream)
    assert rc == -1 and b"exceeds" in L.ec_last_error()
