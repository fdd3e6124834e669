"""ec_adam_clip_step (clip_grad_norm_ + Adam.step() in one launch) against torch's own, through the C ABI."""
import copy

import pytest

pytestmark = pytest.mark.gpu


def _nets(th, seed):
    th.manual_seed(seed)
    mk = lambda i, o: th.nn.Sequential(th.nn.Linear(i, 64), th.nn.ReLU(), th.nn.Linear(64, 64), th.nn.ReLU(), th.nn.Linear(64, o))
    return mk(19, 11).cuda(), mk(9, 1).cuda()


@pytest.mark.parametrize("capturable", [True, False])
@pytest.mark.parametrize("max_norm", [10.0, 0.05, None])
def test_clip_adam_step_matches_torch(capturable, max_norm):
    """Five steps of two networks with fresh random gradients each: parameters, both moments, the step counters and the
    clipped gradients equal torch's clip_grad_norm_ + Adam.step() to fp32 rounding; max_norm 0.05 clips every step,
    10 never, None skips the clip."""
    import torch as th
    from marl_amd.utils import optim_device
    a = _nets(th, 3)
    b = copy.deepcopy(a)
    lrs = (3e-4, 1e-3)
    oa = [th.optim.Adam(n.parameters(), lr=lr, capturable=capturable) for n, lr in zip(a, lrs)]
    ob = [th.optim.Adam(n.parameters(), lr=lr, capturable=capturable) for n, lr in zip(b, lrs)]
    gen = th.Generator(device="cuda").manual_seed(5)
    for step in range(5):
        for na, nb in zip(a, b):
            for pa, pb in zip(na.parameters(), nb.parameters()):
                g = th.randn(pa.shape, device="cuda", generator=gen) * (0.01 if step % 2 else 1.0)
                pa.grad, pb.grad = g.clone(), g.clone()
        assert optim_device.clip_adam_step(oa, max_norm)
        for nb, o in zip(b, ob):
            if max_norm:
                th.nn.utils.clip_grad_norm_(nb.parameters(), max_norm)
            o.step()
        for na, nb, o1, o2 in zip(a, b, oa, ob):
            for pa, pb in zip(na.parameters(), nb.parameters()):
                assert th.allclose(pa, pb, rtol=1e-6, atol=1e-7), step
                assert th.allclose(pa.grad, pb.grad, rtol=1e-5, atol=1e-9), step
                s1, s2 = o1.state[pa], o2.state[pb]
                assert th.allclose(s1["exp_avg"], s2["exp_avg"], rtol=1e-5, atol=1e-7)   # (m = m + (g - m) * 0.1 cancels)
                assert th.allclose(s1["exp_avg_sq"], s2["exp_avg_sq"], rtol=1e-5, atol=1e-9)
                assert float(s1["step"]) == float(s2["step"]) == step + 1
                assert s1["step"].is_cuda == capturable
    # the state dicts are torch's: a fresh optimiser takes them
    fresh = th.optim.Adam(a[0].parameters(), lr=lrs[0], capturable=capturable)
    fresh.load_state_dict(oa[0].state_dict())


def test_clip_adam_step_declines_what_it_does_not_cover():
    import torch as th
    from marl_amd.utils import optim_device
    net, _ = _nets(th, 1)
    for p in net.parameters():
        p.grad = th.zeros_like(p)
    before = [p.clone() for p in net.parameters()]
    assert not optim_device.clip_adam_step([th.optim.Adam(net.parameters(), weight_decay=1e-2)], 1.0)
    assert not optim_device.clip_adam_step([th.optim.SGD(net.parameters(), lr=0.1)], 1.0)
    assert not optim_device.clip_adam_step([th.optim.AdamW(net.parameters())], 1.0)
    cpu = th.nn.Linear(3, 2)
    cpu.weight.grad, cpu.bias.grad = th.zeros(2, 3), th.zeros(2)
    assert not optim_device.clip_adam_step([th.optim.Adam(cpu.parameters())], 1.0)
    assert all(th.equal(x, p) for x, p in zip(before, net.parameters()))


@pytest.mark.parametrize("bad", [float("nan"), float("inf")])
def test_clip_adam_step_propagates_a_non_finite_norm_as_torch_does(bad):
    """A NaN / inf gradient entry: clip_grad_norm_'s coefficient is NaN (inf: 0) and torch's clamp lets it through, which
    poisons every gradient of the network -- how a diverged update gets noticed.  The kernel does the same."""
    import torch as th
    from marl_amd.utils import optim_device
    a = _nets(th, 9)
    b = copy.deepcopy(a)
    oa = [th.optim.Adam(n.parameters(), lr=1e-3) for n in a]
    ob = [th.optim.Adam(n.parameters(), lr=1e-3) for n in b]
    gen = th.Generator(device="cuda").manual_seed(1)
    for na, nb in zip(a, b):
        for pa, pb in zip(na.parameters(), nb.parameters()):
            g = th.randn(pa.shape, device="cuda", generator=gen)
            pa.grad, pb.grad = g.clone(), g.clone()
    for n in (a[0], b[0]):  # the first network only: the second must come out untouched by it
        next(n.parameters()).grad.view(-1)[3] = bad
    assert optim_device.clip_adam_step(oa, 1.0)
    for nb, o in zip(b, ob):
        th.nn.utils.clip_grad_norm_(nb.parameters(), 1.0)
        o.step()
    for na, nb in zip(a, b):
        for pa, pb in zip(na.parameters(), nb.parameters()):
            assert th.equal(th.isnan(pa), th.isnan(pb))
            assert th.equal(th.isnan(pa.grad), th.isnan(pb.grad))
            ok = ~th.isnan(pb)
            assert th.allclose(pa[ok], pb[ok], rtol=1e-6, atol=1e-7)
    assert bool(th.isnan(next(a[0].parameters())).any()) == bool(th.isnan(next(b[0].parameters())).any())
    assert not any(bool(th.isnan(p).any()) for p in a[1].parameters())


def test_clip_adam_step_refusal_leaves_earlier_optimisers_untouched():
    """mixed step placement in one call (one capturable, one not): refused before any host step count moves"""
    import torch as th
    from marl_amd.utils import optim_device
    a = _nets(th, 4)
    opts = [th.optim.Adam(a[0].parameters(), lr=1e-3), th.optim.Adam(a[1].parameters(), lr=1e-3, capturable=True)]
    for n in a:
        for p in n.parameters():
            p.grad = th.ones_like(p)
    for o in opts:  # materialise the states as two earlier steps would have
        assert optim_device.clip_adam_step([o], 1.0)
        assert optim_device.clip_adam_step([o], 1.0)
    before = [float(opts[0].state[p]["step"]) for p in a[0].parameters()]
    # make the SECOND optimiser the one that is refused (a tensor of its state on the wrong side)
    st = opts[1].state[next(a[1].parameters())]
    st["step"] = st["step"].cpu()
    assert not optim_device.clip_adam_step(opts, 1.0)
    assert [float(opts[0].state[p]["step"]) for p in a[0].parameters()] == before == [2.0] * len(before)


def test_bias_corrections_follow_torch_at_small_step_counts():
    """host step counts: 1 - beta^t in double, as torch's non-capturable Adam evaluates it (fp32 powf is ~1e-5 off at t = 1
    for beta2 = 0.999)"""
    import torch as th
    from marl_amd.utils import optim_device
    lin_a = th.nn.Linear(64, 64).cuda()
    lin_b = copy.deepcopy(lin_a)
    oa, ob = th.optim.Adam(lin_a.parameters(), lr=1e-2), th.optim.Adam(lin_b.parameters(), lr=1e-2)
    gen = th.Generator(device="cuda").manual_seed(2)
    for step in range(3):
        for pa, pb in zip(lin_a.parameters(), lin_b.parameters()):
            g = th.randn(pa.shape, device="cuda", generator=gen)
            pa.grad, pb.grad = g.clone(), g.clone()
        assert optim_device.clip_adam_step([oa], None)
        ob.step()
        for pa, pb in zip(lin_a.parameters(), lin_b.parameters()):
            # the step itself (~lr) to 2e-6 relative: a bias correction 1e-5 off would show
            assert float((pa - pb).abs().max()) <= 2e-6 * 1e-2 + 1e-9, step
