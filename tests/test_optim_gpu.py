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
