"""The HIP tail of a GAIL-discriminator layer (csrc/ec_bn.hip: BatchNorm1d in training mode -> ReLU -> Dropout as one
autograd node) against torch's own modules (gail_network.py:97-125 of the reference runs fc -> bn -> activation ->
dropout).  Tolerance: fp32 streaming arithmetic against a float64 reference, 1e-5 relative on outputs, 1e-4 of a
gradient tensor's largest entry."""
import pytest
import torch as th

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("rows,C,p", [(2, 32, 0.2), (640, 128, 0.3), (100003, 64, 0.3), (4097, 128, 0.0)])
def test_hip_layer_tail_matches_modules(rows, C, p, monkeypatch):
    import marl_amd.rl.policy_gradient_rl.gail.gail_network as gn
    dev = th.device("cuda")
    th.manual_seed(rows + C)
    x = th.randn(rows, C) * th.linspace(0.5, 2.0, C) + th.linspace(-1, 1, C)
    g = th.randn(rows, C)
    mask = th.rand(rows, C) >= p
    stock = th.nn.BatchNorm1d(C).double()
    tall = gn.TallBatchNorm1d(C)
    with th.no_grad():
        for m in (stock, tall):
            m.weight.copy_(th.linspace(0.5, 1.5, C))
            m.bias.copy_(th.linspace(-0.5, 0.5, C))
        # elements whose pre-activation is within fp32 rounding of the ReLU kink have no well-defined gradient to
        # compare (one in ~10^7 at these sizes): drop them through the mask
        y = th.nn.functional.batch_norm(x.double(), None, None, stock.weight, stock.bias, True, 0.0, stock.eps)
        mask &= y.abs() > 1e-4
    monkeypatch.setattr(gn, "_explicit_mask", lambda x_, p_: mask.to(x_.device))
    tall = tall.to(dev)
    xa = x.double().requires_grad_(True)
    out_a = th.relu(stock(xa)) * mask / (1 - p)
    (out_a * g.double()).sum().backward()
    xb = x.to(dev).requires_grad_(True)
    assert tall.hip_ok(xb)
    out_b = tall.relu_dropout(xb, p)
    (out_b * g.to(dev)).sum().backward()
    th.cuda.synchronize()
    assert th.allclose(out_b.detach().cpu().double(), out_a.detach(), rtol=1e-5, atol=1e-5)
    for name, a, b in (("dx", xa.grad, xb.grad), ("dweight", stock.weight.grad, tall.weight.grad),
                       ("dbias", stock.bias.grad, tall.bias.grad)):
        err = float((b.detach().cpu().double() - a).abs().max())
        # (two rows: the BatchNorm backward cancels to almost nothing, what is left is fp32 rounding of O(|dout| * scale) terms)
        assert err <= 1e-4 * float(a.abs().max()) + (1e-4 if rows < 64 else 1e-7), (name, err)
    assert th.allclose(tall.running_mean.cpu().double(), stock.running_mean, rtol=1e-5, atol=1e-6)
    assert th.allclose(tall.running_var.cpu().double(), stock.running_var, rtol=1e-5, atol=1e-6)


def test_hip_layer_tail_own_mask():
    """Without an explicit mask the kernels draw it from (seed, element index): about 1 - p of the elements are kept,
    the backward recomputes the same mask (its input gradient equals torch's for the mask read off the forward output),
    two seeds differ, and the same torch seed reproduces the same mask."""
    import marl_amd.rl.policy_gradient_rl.gail.gail_network as gn
    dev = th.device("cuda")
    rows, C, p = 200000, 128, 0.3
    bn = gn.TallBatchNorm1d(C).to(dev)
    ref = th.nn.BatchNorm1d(C).to(dev)
    with th.no_grad():
        bn.bias.fill_(5.0)    # pre-activations = xhat + 5: positive, so that out > 0 reads off the keep mask
        ref.bias.fill_(5.0)
    x = th.randn(rows, C, device=dev)
    g = th.randn(rows, C, device=dev)
    xb = x.clone().requires_grad_(True)
    th.manual_seed(11)
    out = bn.relu_dropout(xb, p)
    (out * g).sum().backward()
    kept = out > 0
    frac = float(kept.float().mean())
    assert abs(frac - (1 - p)) < 2e-3, frac
    xa = x.clone().requires_grad_(True)
    (th.relu(ref(xa)) * kept / (1 - p) * g).sum().backward()
    assert float((xa.grad - xb.grad).abs().max()) < 1e-4 * float(xa.grad.abs().max())
    th.manual_seed(11)
    again = bn.relu_dropout(x, p)
    th.manual_seed(12)
    other = bn.relu_dropout(x, p)
    assert th.equal(again > 0, kept) and not th.equal(other > 0, kept)
    cols, rws = kept.float().mean(0), kept.float().mean(1)          # no column or row pattern
    assert float(cols.min()) > 0.69 and float(cols.max()) < 0.71
    assert float(rws.min()) > 0.45 and float(rws.max()) < 0.92
    # neighbours in a row, neighbours in a column and the two seeds are independent draws: P(both kept) = (1 - p)^2
    k, o = kept.float(), (other > 0).float()
    for both in ((k[:, 1:] * k[:, :-1]).mean(), (k[1:] * k[:-1]).mean(), (k * o).mean()):
        assert abs(float(both) - (1 - p) ** 2) < 2e-3, float(both)


@pytest.mark.gpu
@pytest.mark.parametrize("rows,C,offset", [(2, 32, 0.0), (1000, 128, 0.0), (100003, 64, 50.0), (1 << 21, 32, -300.0), (65, 4, 1e4)])
def test_col_stats_matches_float64(rows, C, offset):
    """ec_bn_col_stats (the one-pass batch statistics of the layer tail): mean and biased variance per column against
    float64, also for columns whose mean is hundreds of standard deviations from zero (the sums are taken about the
    pivot row, so E[x^2] - mean^2 does not cancel)."""
    import ctypes as C_
    import torch as th
    from marl_amd import _abi
    gen = th.Generator().manual_seed(rows + C)
    x = (th.randn(rows, C, generator=gen) * th.linspace(0.1, 3.0, C) + offset * th.linspace(-1, 1, C)).cuda()
    n_part = min(1024, max(1, rows // 64))
    partials = th.full((n_part, 2, C), float("nan"), device="cuda")
    _abi.check(_abi.lib().ec_bn_col_stats(C_.c_void_p(x.data_ptr()), rows, C, n_part, C_.c_void_p(partials.data_ptr()),
                                          C_.c_void_p(th.cuda.current_stream().cuda_stream)), "ec_bn_col_stats")
    s = partials.double().sum(0) / rows
    mean = x[0].double() + s[0]
    var = s[1] - s[0] * s[0]
    want_var, want_mean = th.var_mean(x.double(), dim=0, unbiased=False)
    assert th.allclose(mean, want_mean, rtol=1e-6, atol=1e-6)
    assert th.allclose(var, want_var, rtol=2e-5, atol=1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("rows,c_out,k_in", [(1, 32, 1), (63, 64, 9), (1000, 128, 30), (100003, 64, 128), (4097, 32, 64),
                                             (70000, 96, 50), (300000, 128, 128)])
def test_linear_wgrad_matches_float64(rows, c_out, k_in):
    """ec_linear_wgrad (dW = dy^T x and db = sum dy over a tall batch, rows as the MFMA's K dimension) against float64:
    ragged row counts, every tile count, input widths that are not multiples of 32; and through SplitKLinear's backward
    against nn.Linear's."""
    import torch as th
    from marl_amd.rl.policy_gradient_rl.dgail.dgail_network import SplitKLinear
    from marl_amd.utils import linear_device
    gen = th.Generator().manual_seed(rows + c_out + k_in)
    gy = th.randn(rows, c_out, generator=gen).cuda()
    x = (th.randn(rows, k_in, generator=gen) + 0.5).cuda()
    with th.no_grad():
        assert linear_device.wgrad_ok(gy, x)
        gw, gb = linear_device.wgrad(gy, x)
    want_w, want_b = gy.double().t() @ x.double(), gy.double().sum(0)
    tol = 2e-6 * max(1.0, rows ** 0.5)
    assert float((gw.double() - want_w).abs().max()) <= tol * float(want_w.abs().max() + 1)
    assert float((gb.double() - want_b).abs().max()) <= tol * float(want_b.abs().max() + 1)
    if rows >= SplitKLinear.MIN_ROWS:
        a, b = SplitKLinear(k_in, c_out).cuda(), th.nn.Linear(k_in, c_out).cuda()
        b.load_state_dict(a.state_dict())
        xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
        (a(xa) * gy).sum().backward()
        (b(xb) * gy).sum().backward()
        assert th.allclose(xa.grad, xb.grad, rtol=1e-5, atol=1e-6)
        for pa, pb in zip(a.parameters(), b.parameters()):
            assert float((pa.grad - pb.grad).abs().max()) <= 1e-4 * float(pb.grad.abs().max() + 1)


@pytest.mark.gpu
@pytest.mark.parametrize("rows,k_in,c_out", [(1, 1, 1), (33, 30, 128), (100003, 64, 32), (5000, 9, 5), (70001, 50, 96),
                                             (8192, 32, 64)])
def test_linear_fwd_matches_float64(rows, k_in, c_out):
    """ec_linear_fwd (y = x W^T + b, rows as the MFMA's M dimension) against float64: ragged row counts, input widths that
    are not multiples of 4 or 32, output widths that are not multiples of 32, with and without bias; and SplitKLinear
    (forward, input gradient, weight gradient on the HIP kernels) against nn.Linear end to end."""
    import torch as th
    from marl_amd.rl.policy_gradient_rl.dgail.dgail_network import SplitKLinear
    from marl_amd.utils import linear_device
    gen = th.Generator().manual_seed(rows + k_in + c_out)
    x = th.randn(rows, k_in, generator=gen).cuda()
    w, b = (th.randn(c_out, k_in, generator=gen) * 0.3).cuda(), th.randn(c_out, generator=gen).cuda()
    with th.no_grad():
        assert linear_device.fwd_ok(x, w)
        for bias in (b, None):
            got = linear_device.fwd(x, w, bias)
            want = x.double() @ w.double().t() + (bias.double() if bias is not None else 0.0)
            assert float((got.double() - want).abs().max()) <= 2e-6 * float(want.abs().max() + 1)
        # ec_linear_fwd_stats: the same y, and partials that add up to the column sums about y's first row
        y_s, partials, pivot = linear_device.fwd(x, w, b, want_stats=True)
        assert th.equal(y_s, linear_device.fwd(x, w, b)) and th.equal(pivot, y_s[0])
        d = y_s.double() - pivot.double()
        got_s = partials.double().sum(0)
        for which, want_s in ((0, d.sum(0)), (1, (d * d).sum(0))):
            scale_ = (d.abs() if which == 0 else d * d).sum(0) + 1e-30
            assert float(((got_s[which] - want_s).abs() / scale_).max()) < 2e-6, which
    if rows >= SplitKLinear.MIN_ROWS and c_out in (32, 64, 96, 128):
        a, ref = SplitKLinear(k_in, c_out).cuda(), th.nn.Linear(k_in, c_out).cuda()
        ref.load_state_dict(a.state_dict())
        g = th.randn(rows, c_out, generator=gen).cuda()
        xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
        ya, yb = a(xa), ref(xb)
        assert th.allclose(ya, yb, rtol=1e-5, atol=1e-5)
        (ya * g).sum().backward()
        (yb * g).sum().backward()
        assert th.allclose(xa.grad, xb.grad, rtol=1e-5, atol=1e-5)
        for pa, pb in zip(a.parameters(), ref.parameters()):
            assert float((pa.grad - pb.grad).abs().max()) <= 1e-4 * float(pb.grad.abs().max() + 1)


@pytest.mark.parametrize("rows,k_in,c_out,p", [(70001, 128, 64, 0.3), (4097, 64, 32, 0.3), (100003, 32, 1, 0.2),
                                               (33, 128, 128, 0.0), (2, 36, 40, 0.5), (517, 30, 16, 0.25),
                                               (4099, 97, 5, 0.1)])
def test_linear_fwd_bn_equals_tail_then_linear(rows, k_in, c_out, p):
    """ec_linear_fwd_bn (the BatchNorm -> ReLU -> Dropout tail applied inside the next Linear's kernel) against the two
    kernels it replaces, ec_bn_relu_dropout_fwd then ec_linear_fwd, for the same seed: bit for bit (same arithmetic, same
    keep_draw, same MFMA order); and against float64 on the two-step's activated tensor."""
    import ctypes as C
    from marl_amd import _abi
    from marl_amd.utils import linear_device
    gen = th.Generator().manual_seed(rows + k_in + c_out)
    x = th.randn(rows, k_in, generator=gen).cuda()
    scale, shift = (th.rand(k_in, generator=gen) + 0.5).cuda(), (th.randn(k_in, generator=gen) * 0.5).cuda()
    w, b = (th.randn(c_out, k_in, generator=gen) * 0.3).cuda(), th.randn(c_out, generator=gen).cuda()
    seed = 0x1234567890abcdef + rows
    vp = lambda t: C.c_void_p(t.data_ptr())
    stream = C.c_void_p(th.cuda.current_stream().cuda_stream)
    with th.no_grad():
        assert linear_device.fwd_bn_ok(x, w)
        if k_in % 4 == 0 and 256 % (k_in // 4) == 0:
            act = th.empty_like(x)
            _abi.check(_abi.lib().ec_bn_relu_dropout_fwd(vp(x), vp(scale), vp(shift), C.c_void_p(0), C.c_uint64(seed), C.c_float(p),
                                                         rows, k_in, vp(act), stream), "ec_bn_relu_dropout_fwd")
            kept = float((act != 0).float().mean())
            want_kept = float(((x * scale + shift) > 0).float().mean()) * (1 - p)
            assert abs(kept - want_kept) < 0.02 + 2.0 / (rows * k_in) ** 0.5
        else:   # a width the tail kernel's thread mapping does not take: its formula in torch, keep flags off the fused output
            act = None
        for bias in (b, None):
            got = linear_device.fwd_bn(x, scale, shift, seed, p, w, bias)
            if act is not None:
                two_step = linear_device.fwd(act, w, bias)
                assert th.equal(got, two_step)
                want = act.double() @ w.double().t() + (bias.double() if bias is not None else 0.0)
                assert float((got.double() - want).abs().max()) <= 2e-6 * float(want.abs().max() + 1)
        # the optional statistics epilogue: y unchanged, partials add up to the column sums of (y - b) and (y - b)^2
        y_plain = linear_device.fwd_bn(x, scale, shift, seed, p, w, b)
        y_stats, partials = linear_device.fwd_bn(x, scale, shift, seed, p, w, b, want_stats=True)
        assert th.equal(y_plain, y_stats) and partials.shape[1:] == (2, c_out)
        d = y_stats.double() - b.double()
        got_s = partials.double().sum(0)
        for which, want_s in ((0, d.sum(0)), (1, (d * d).sum(0))):
            scale_ = (d.abs() if which == 0 else d * d).sum(0) + 1e-30
            assert float(((got_s[which] - want_s).abs() / scale_).max()) < 2e-6, which
        if act is None:   # identity weight block reads the activated values back out of the fused kernel
            eye = th.eye(k_in, device="cuda")
            a_out = th.cat([linear_device.fwd_bn(x, scale, shift, seed, p, eye[i:i + 32].contiguous()) for i in range(0, k_in, 32)], 1)
            y = x * scale + shift
            full = th.where(y > 0, y * (1.0 / (1.0 - p)), th.zeros_like(y))
            assert bool(((a_out == 0) | (a_out == full)).all())
            assert abs(float((a_out != 0).float().mean()) - float((y > 0).float().mean()) * (1 - p)) < 0.25
    # argument checks
    lib = _abi.lib()
    assert lib.ec_linear_fwd_bn(vp(x), vp(scale), vp(shift), C.c_uint64(1), C.c_float(1.0), vp(w), vp(b), rows, k_in, c_out, vp(x), 0, C.c_void_p(0), stream) != 0
    assert lib.ec_linear_fwd_bn(vp(x), vp(scale), vp(shift), C.c_uint64(1), C.c_float(0.1), vp(w), vp(b), rows, 129, c_out, vp(x), 0, C.c_void_p(0), stream) != 0
    assert lib.ec_linear_fwd_bn(vp(x), C.c_void_p(0), vp(shift), C.c_uint64(1), C.c_float(0.1), vp(w), vp(b), rows, k_in, c_out, vp(x), 0, C.c_void_p(0), stream) != 0


@pytest.mark.parametrize("rows,k_in,p", [(70001, 30, 0.3), (4097, 50, 0.3), (33, 90, 0.0), (2, 9, 0.5), (100003, 64, 0.25),
                                         (1, 30, 0.3), (262144 + 5, 96, 0.3), (4096000, 30, 0.3)])   # (the last: GAIL's full size)
def test_linear2_fwd_bn_equals_two_kernels(rows, k_in, p):
    """ec_linear2_fwd_bn (layers 1 and 2 of the GAIL discriminator in one kernel, layer 1's 128-wide output in registers)
    against the two kernels it replaces -- ec_linear_fwd, then ec_linear_fwd_bn -- for the same seed: y bit for bit (same
    products in the same order, the same tail arithmetic and keep_draw); its statistics epilogue against float64; and
    the statistics-only form of ec_linear_fwd_stats (y = NULL) leaves the partials of the form that writes y."""
    import ctypes as C
    from marl_amd import _abi
    from marl_amd.utils import linear_device
    gen = th.Generator().manual_seed(rows + k_in)
    x = th.randn(rows, k_in, generator=gen).cuda()
    w1, b1 = (th.randn(128, k_in, generator=gen) * 0.3).cuda(), th.randn(128, generator=gen).cuda()
    w2, b2 = (th.randn(64, 128, generator=gen) * 0.2).cuda(), th.randn(64, generator=gen).cuda()
    scale, shift = (th.rand(128, generator=gen) + 0.5).cuda(), (th.randn(128, generator=gen) * 0.5).cuda()
    seed = 0x0fedcba987654321 + rows
    with th.no_grad():
        assert linear_device.fwd2_bn_ok(x, w1, w2) == (k_in <= 32)   # (wider rows: the kernel is correct but not the product's choice)
        for bias1, bias2 in ((b1, b2), (None, None)):
            y1 = linear_device.fwd(x, w1, bias1)
            want = linear_device.fwd_bn(y1, scale, shift, seed, p, w2, bias2)
            got = linear_device.fwd2_bn(x, w1, bias1, scale, shift, seed, p, w2, bias2)
            assert th.equal(got, want)
        y_stats, partials = linear_device.fwd2_bn(x, w1, b1, scale, shift, seed, p, w2, b2, want_stats=True)
        assert th.equal(y_stats, got if bias2 is not None else linear_device.fwd2_bn(x, w1, b1, scale, shift, seed, p, w2, b2))
        assert partials.shape[1:] == (2, 64)
        d = y_stats.double() - b2.double()
        got_s = partials.double().sum(0)
        for which, want_s in ((0, d.sum(0)), (1, (d * d).sum(0))):
            scale_ = (d.abs() if which == 0 else d * d).sum(0) + 1e-30
            assert float(((got_s[which] - want_s).abs() / scale_).max()) < 2e-6, which
        # the statistics of layer 1 without writing it
        if k_in <= 64:
            _, part_y, piv_y = linear_device.fwd(x, w1, b1, want_stats=True)
            part_n, piv_n = linear_device.fwd_stats_only(x, w1, b1)
            assert th.equal(part_y, part_n) and th.equal(piv_y, piv_n)
    lib = _abi.lib()
    vp = lambda t: C.c_void_p(t.data_ptr())
    stream = C.c_void_p(th.cuda.current_stream().cuda_stream)
    y = th.empty(rows, 64, device="cuda")
    args = lambda k=k_in, cm=128, co=64, pp=0.1: (vp(x), vp(w1), vp(b1), vp(scale), vp(shift), C.c_uint64(1), C.c_float(pp), vp(w2),
                                                   vp(b2), rows, k, cm, co, vp(y), 0, C.c_void_p(0), stream)
    assert lib.ec_linear2_fwd_bn(*args()) == 0
    assert lib.ec_linear2_fwd_bn(*args(k=97)) != 0
    assert lib.ec_linear2_fwd_bn(*args(cm=64)) != 0
    assert lib.ec_linear2_fwd_bn(*args(co=32)) != 0
    assert lib.ec_linear2_fwd_bn(*args(pp=1.0)) != 0


@pytest.mark.parametrize("rows,k_in", [(1, 30), (33, 30), (100003, 30), (70001, 50), (5000, 9), (40000, 90), (8192, 32),
                                        (3000, 96), (262144 + 7, 64), (4096000, 30)])
def test_row_moments_give_layer_statistics(rows, k_in):
    """ec_row_moments (sums of d and d d^T over the rows, d = x[row] - x[0], the row index as the MFMA's K dimension)
    against float64, with columns far from zero mean and one-hot columns as the discriminator's rows have them; and the
    statistics of a Linear layer's output derived from them (linear_device.fwd_stats_from_moments) against the column
    sums of the layer's actual output in float64."""
    import ctypes as C
    from marl_amd import _abi
    from marl_amd.utils import linear_device
    gen = th.Generator().manual_seed(rows * 7 + k_in)
    x = th.randn(rows, k_in, generator=gen) * th.linspace(0.2, 3.0, k_in) + th.linspace(-5.0, 20.0, k_in)
    hot = th.randint(0, max(1, k_in // 3), (rows,), generator=gen)
    x[:, : max(1, k_in // 3)] = th.nn.functional.one_hot(hot, max(1, k_in // 3)).float()
    x = x.cuda()
    w, b = (th.randn(128, k_in, generator=gen) * 0.3).cuda(), th.randn(128, generator=gen).cuda()
    kp = 32 * ((k_in + 31) // 32)
    n_part = 64
    part = th.full((n_part, kp * kp + kp), float("nan"), device="cuda")
    vp = lambda t: C.c_void_p(t.data_ptr())
    stream = C.c_void_p(th.cuda.current_stream().cuda_stream)
    lib = _abi.lib()
    assert lib.ec_row_moments(vp(x), rows, k_in, n_part, vp(part), stream) == 0
    tot = part.double().sum(0)
    d = x.double() - x[0].double()
    m2, s1 = tot[:kp * kp].view(kp, kp), tot[kp * kp:]
    want_m2 = d.t() @ d
    assert float((m2[:k_in, :k_in] - want_m2).abs().max()) <= 2e-6 * float((d.abs().t() @ d.abs()).max() + 1e-30) + 1e-30
    assert float((s1[:k_in] - d.sum(0)).abs().max()) <= 2e-6 * float(d.abs().sum(0).max() + 1e-30) + 1e-30
    assert float(m2[k_in:].abs().sum() + m2[:, k_in:].abs().sum() + s1[k_in:].abs().sum()) == 0.0
    with th.no_grad():
        partials, pivot = linear_device.fwd_stats_from_moments(x, w, b)
        y = x.double() @ w.double().t() + b.double()
        assert partials.shape == (1, 2, 128) and th.equal(pivot, linear_device.fwd(x[:1], w, b).reshape(128))
        dy = y - pivot.double()
        if rows > 1:
            for which, want_s in ((0, dy.sum(0)), (1, (dy * dy).sum(0))):
                scale_ = (dy.abs() if which == 0 else dy * dy).sum(0) + 1e-30
                assert float(((partials[0, which].double() - want_s).abs() / scale_).max()) < 1e-5, which
    assert lib.ec_row_moments(vp(x), rows, 97, n_part, vp(part), stream) != 0
    assert lib.ec_row_moments(vp(x), 0, k_in, n_part, vp(part), stream) != 0
    assert lib.ec_row_moments(vp(x), rows, k_in, 0, vp(part), stream) != 0


@pytest.mark.parametrize("rows,n_agents,n_actions", [(50000, 10, 11), (20000, 40, 41)])
def test_discriminator_no_grad_pass_fused_tails(rows, n_agents, n_actions, monkeypatch):
    """Discriminator.forward_rows under no_grad in training mode (GAIL's reward pass, gail_agent.py:165-167) folds every
    layer's tail into the next Linear (ec_linear_fwd_bn).  Same torch seed => the same dropout draws as the layer-by-layer
    path: logits equal to fp32 rounding (the 128-wide GEMM of the layer-by-layer path is hipBLASLt's), running statistics
    and batch counters advanced alike."""
    import copy
    from types import SimpleNamespace
    import marl_amd.rl.policy_gradient_rl.gail.gail_network as gn
    args = SimpleNamespace(obs_shape=9, n_agents=n_agents, n_actions=n_actions, hidden_dim=64, activate_fun="relu")
    th.manual_seed(3)
    a = gn.Discriminator(args).cuda()
    with th.no_grad():
        for m in a.modules():
            if isinstance(m, th.nn.BatchNorm1d):
                m.weight.uniform_(0.5, 1.5)
                m.bias.uniform_(-0.3, 0.3)
    b = copy.deepcopy(a)
    x = th.randn(rows, 9 + n_agents + n_actions, device="cuda")
    with th.no_grad():
        assert a._fused_tails_ok(x)
        th.manual_seed(11)
        ya = a.forward_rows(x)
        monkeypatch.setattr(gn.Discriminator, "_fused_tails_ok", lambda self, x_: False)
        th.manual_seed(11)
        yb = b.forward_rows(x)
    assert ya.shape == yb.shape == (rows, 1)
    assert float((ya - yb).abs().max()) <= 1e-4 * float(yb.abs().max() + 1)
    for (na, ba), (_, bb) in zip(a.named_buffers(), b.named_buffers()):
        assert th.allclose(ba.float(), bb.float(), rtol=1e-5, atol=1e-6), na
    assert int(a.bn1.num_batches_tracked) == 1
    # layers 1 and 2 in one kernel (ec_linear2_fwd_bn, the default above) against one kernel per layer: the same draws,
    # bn1's statistics from the moments of the input rows, bn2's from a differently ordered sum
    monkeypatch.undo()
    c = gn.Discriminator(args).cuda()
    c.load_state_dict({k: v for k, v in a.state_dict().items()})
    with th.no_grad():
        for m in c.modules():
            if isinstance(m, th.nn.BatchNorm1d):
                m.reset_running_stats()
        d = copy.deepcopy(c)
        monkeypatch.setattr(gn.Discriminator, "FUSE_FIRST_TWO", False)
        th.manual_seed(11)
        yc = c.forward_rows(x)
        monkeypatch.setattr(gn.Discriminator, "FUSE_FIRST_TWO", True)
        th.manual_seed(11)
        yd = d.forward_rows(x)
    assert float((yc - yd).abs().max()) <= 2e-5 * float(yc.abs().max() + 1)
    for (na, ba), (_, bb) in zip(c.named_buffers(), d.named_buffers()):
        assert th.allclose(ba.float(), bb.float(), rtol=1e-5, atol=1e-6), na
    # with grad enabled, or with a test's mask hook in place, the layer-by-layer path runs
    monkeypatch.undo()
    assert not a._fused_tails_ok(x.requires_grad_(True)) or not th.is_grad_enabled()
    with th.enable_grad():
        assert not a._fused_tails_ok(x)
    monkeypatch.setattr(gn, "_explicit_mask", lambda x_, p_: None)
    with th.no_grad():
        assert not a._fused_tails_ok(x)


@pytest.mark.parametrize("B,T,N,A", [(3, 5, 10, 11), (7, 100, 40, 41), (1, 1, 2, 3), (300, 9, 10, 11)])
def test_disc_rows_equals_cat_of_one_hots(B, T, N, A):
    """ec_disc_rows against the torch construction of the discriminator's input (gail_agent.py:131-141: obs with the agent
    one-hot appended, concatenated with the one-hot action): exact."""
    import ctypes as C
    from marl_amd import _abi
    from marl_amd.utils import linear_device
    gen = th.Generator().manual_seed(B * T + N)
    obs = th.randn(B, T, N, 9, generator=gen).cuda()
    actions = th.randint(0, A, (B, T, N, 1), generator=gen).cuda()
    got = linear_device.disc_rows(obs, actions, N, A)
    eye = th.eye(N, device="cuda").expand(B, T, N, N)
    want = th.cat([obs, eye, th.nn.functional.one_hot(actions.squeeze(-1), num_classes=A).float()], dim=-1).reshape(-1, 9 + N + A)
    assert got.shape == want.shape and th.equal(got, want)
    # an out-of-range action sets nothing; rows wider than the staged form takes (obs_dim > 63) go the direct way
    bad = actions.clone()
    bad.view(-1)[::3] = A + 5
    bad.view(-1)[1::7] = -1
    got_bad = linear_device.disc_rows(obs, bad, N, A)
    keep = ((bad >= 0) & (bad < A)).reshape(-1, 1).float()
    assert th.equal(got_bad[:, :9 + N], want[:, :9 + N]) and th.equal(got_bad[:, 9 + N:], want[:, 9 + N:] * keep)
    wide = th.randn(B, T, N, 70, generator=gen).cuda()
    want_w = th.cat([wide, eye, th.nn.functional.one_hot(actions.squeeze(-1), num_classes=A).float()], dim=-1).reshape(-1, 70 + N + A)
    assert th.equal(linear_device.disc_rows(wide, actions, N, A), want_w)
    vp = lambda t: C.c_void_p(t.data_ptr())
    stream = C.c_void_p(th.cuda.current_stream().cuda_stream)
    assert _abi.lib().ec_disc_rows(vp(obs), vp(actions), B * T * N, 9, 0, A, vp(got), stream) != 0
    assert _abi.lib().ec_disc_rows(vp(obs), C.c_void_p(0), B * T * N, 9, N, A, vp(got), stream) != 0
    assert _abi.lib().ec_disc_rows(vp(obs), vp(actions), 0, 9, N, A, vp(got), stream) == 0
