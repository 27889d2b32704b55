"""GPU parity of the individual HIP kernels (through the C ABI) against plain
PyTorch fp32 CPU restatements of the same op chains.

fp32 instantiation: exact-fp32 MFMA (v_mfma_f32_16x16x4_f32), compared at
rtol 2e-5 / atol 2e-5 * scale (summation order differs from oneDNN).
bf16 instantiation: inputs/weights (and the prologue's output, as the kernel does) rounded to bf16 on BOTH
sides, fp32 accumulate; what is left is one bf16 rounding of the stored output (2^-9 relative) plus fp32
summation order, so the bound is atol 4e-3 * max|ref| + rtol 1e-2 (round 3 used 3e-2 / 3e-2, which a dropped
border pixel at K = 2304 could pass).  Tests whose operands cannot be pre-rounded on the reference side pass
`loose=` with the reason at the call site.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.bfloat16]


def _ops():
    from mcgen_amd import ops
    return ops


BF16_ATOL, BF16_RTOL = 4e-3, 1e-2


def _tol(dtype, ref, loose=None):
    scale = float(ref.abs().max()) + 1e-6
    if dtype == torch.float32:
        return (2e-5 * scale + 1e-6, 2e-5)
    a, r = loose if loose is not None else (BF16_ATOL, BF16_RTOL)
    return (a * scale, r)


def _assert_close(got, ref, dtype, what='', loose=None):
    atol, rtol = _tol(dtype, ref, loose)
    got = got.float().cpu()
    err = (got - ref).abs()
    bad = err > atol + rtol * ref.abs()
    assert not bad.any(), f'{what}: {int(bad.sum())}/{bad.numel()} mismatches, max err {float(err.max()):.3e} (atol {atol:.2e})'


def _rnd(gen, *shape):
    return torch.randn(*shape, generator=gen)


def _hot(x, amp=24.0):
    """Border probes: the same random tensor with a few entries made `amp` times larger than
    the rest, so that a kernel which drops a border pixel, the last channel, or a channel next to the compacted-pitch /
    32-channel chunk edges is off by a large fraction of max|ref| instead of 1/150 of it: all channels of the four corner
    pixels of the first and the last image, and channels {C-1, 31, 32, 95, 96, 159, 160} (those < C) of every pixel of
    the images in between.  Positive values, so a ReLU in the prologue keeps them."""
    x = x.clone()
    n, c, h, w = x.shape
    for img in {0, n - 1}:
        for r, q in ((0, 0), (0, w - 1), (h - 1, 0), (h - 1, w - 1)):
            x[img, :, r, q] = x[img, :, r, q].abs() * amp + amp
    mid = slice(1, n - 1) if n > 2 else slice(0, n)
    for ch in (c - 1, 31, 32, 95, 96, 159, 160):
        if 0 <= ch < c:
            x[mid, ch] = x[mid, ch].abs() * amp + amp
    return x


PROBES = [False, True]


def _nhwc(ops, x, dtype):
    return ops.to_nhwc(x.cuda(), dtype)


def _q(x, dtype):
    """round-trip through the compute dtype (what the kernel will actually read)"""
    return x.to(dtype).float()


def ref_prologue(x, scale, shift, relu, code, ups):
    if ups:
        x = x.repeat_interleave(2, 2).repeat_interleave(2, 3)
    if scale is not None:
        x = x * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    if relu:
        x = torch.relu(x)
    if code is not None:
        x = x * code.view(*code.shape, 1, 1)
    return x


CASES = [
    # N, H, W, Cin, Cout, ksize
    (3, 8, 8, 16, 24, 3),
    (2, 32, 32, 40, 130, 3),
    (5, 4, 4, 8, 3, 3),
    (6, 16, 16, 32, 64, 1),
    (2, 32, 32, 128, 128, 3),
    (9, 4, 4, 64, 48, 3),
    (5, 1, 1, 128, 64, 1),
]


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('case', CASES)
def test_conv_plain(case, dtype):
    ops = _ops()
    n, h, w, ci, co, ks = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x, wt, b = _rnd(g, n, ci, h, w), _rnd(g, co, ci, ks, ks) * 0.1, _rnd(g, co)
    ref = F.conv2d(_q(x, dtype), _q(wt, dtype), b, padding=ks // 2)
    wimg = ops.prep_weight(wt.cuda(), dtype)
    y, _ = ops.conv_fused([ops.Seg(_nhwc(ops, x, dtype), ksize=ks)], wimg, co, bias=b.cuda())
    _assert_close(ops.to_nchw(y, co), ref, dtype, 'conv')


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('ups', [False, True])
def test_conv_prologue_stats(dtype, ups):
    """BN-apply + ReLU + (upsample) + MC code prologue, bias, stats epilogue (G.conv_a, mcgan.py:15-19)."""
    ops = _ops()
    g = torch.Generator().manual_seed(7)
    n, hs, ci, co = 6, 8, 32, 40
    x = _rnd(g, n, ci, hs, hs)
    scale, shift = _rnd(g, ci) * 0.5 + 1, _rnd(g, ci) * 0.3
    code = (torch.rand(n, ci, generator=g) < 0.5).float()
    wt, b = _rnd(g, co, ci, 3, 3) * 0.1, _rnd(g, co)
    a = ref_prologue(_q(x, dtype), scale, shift, True, code, ups)
    ref = F.conv2d(_q(a, dtype) if dtype != torch.float32 else a, _q(wt, dtype), b, padding=1)
    seg = ops.Seg(_nhwc(ops, x, dtype), scale=scale.cuda(), shift=shift.cuda(), code=code.cuda(), ups=ups, relu=True)
    y, st = ops.conv_fused([seg], ops.prep_weight(wt.cuda(), dtype), co, bias=b.cuda(), stats_mode=1)
    _assert_close(ops.to_nchw(y, co), ref, dtype, 'conv_a')
    s = st.sum(0).cpu()                         # [2, Cy]
    yq = ops.to_nchw(y, co).cpu()
    # the kernel sums the fp32 accumulators, yq is the stored (rounded) output
    tol = dict(rtol=2e-3, atol=2e-2) if dtype == torch.float32 else dict(rtol=2e-2, atol=4.0)
    np.testing.assert_allclose(s[0, :co], yq.sum((0, 2, 3)), **tol)
    np.testing.assert_allclose(s[1, :co], (yq * yq).sum((0, 2, 3)), **tol)


@pytest.mark.parametrize('dtype', DTYPES)
def test_conv_two_segments(dtype):
    """G.conv_b: 3x3 over BN/ReLU/MC'd h  (+)  1x1 shortcut over Up(x)*mc_1  (mcgan.py:20-30,42)."""
    ops = _ops()
    g = torch.Generator().manual_seed(11)
    n, hs, c = 4, 8, 32
    h_in, x = _rnd(g, n, c, 2 * hs, 2 * hs), _rnd(g, n, c, hs, hs)
    scale, shift = _rnd(g, c) * 0.5 + 1, _rnd(g, c) * 0.3
    code1 = (torch.rand(n, c, generator=g) < 0.5).float()
    code2 = (torch.rand(n, c, generator=g) < 0.5).float()
    w2, ws, b = _rnd(g, c, c, 3, 3) * 0.1, _rnd(g, c, c, 1, 1) * 0.2, _rnd(g, c)
    a2 = ref_prologue(_q(h_in, dtype), scale, shift, True, code2, False)
    a1 = ref_prologue(_q(x, dtype), None, None, False, code1, True)
    if dtype != torch.float32:
        a2, a1 = _q(a2, dtype), _q(a1, dtype)
    ref = F.conv2d(a2, _q(w2, dtype), b, padding=1) + F.conv2d(a1, _q(ws, dtype))
    img = torch.cat([ops.prep_weight(w2.cuda(), dtype), ops.prep_weight(ws.cuda(), dtype)])
    segs = [ops.Seg(_nhwc(ops, h_in, dtype), scale=scale.cuda(), shift=shift.cuda(), code=code2.cuda(), relu=True),
            ops.Seg(_nhwc(ops, x, dtype), ksize=1, code=code1.cuda(), ups=True)]
    y, _ = ops.conv_fused(segs, img, c, bias=b.cuda())
    _assert_close(ops.to_nchw(y, c), ref, dtype, 'conv_b')


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('hw', [32, 16, 8, 4])
def test_conv_pool_residual(dtype, hw):
    """D.conv_b: ReLU->MC->conv3x3->AvgPool2 + residual (mcgan.py:105-115)."""
    ops = _ops()
    g = torch.Generator().manual_seed(13)
    n, c = 3, 16
    x, res = _rnd(g, n, c, hw, hw), _rnd(g, n, c, hw // 2, hw // 2)
    code = (torch.rand(n, c, generator=g) < 0.5).float()
    wt, b = _rnd(g, c, c, 3, 3) * 0.1, _rnd(g, c)
    a = ref_prologue(_q(x, dtype), None, None, True, code, False)
    ref = F.avg_pool2d(F.conv2d(a, _q(wt, dtype), None, padding=1), 2) + b.view(1, -1, 1, 1) + _q(res, dtype)
    y, _ = ops.conv_fused([ops.Seg(_nhwc(ops, x, dtype), code=code.cuda(), relu=True)], ops.prep_weight(wt.cuda(), dtype), c,
                          bias=b.cuda(), pool=True, alpha=0.25, res=_nhwc(ops, res, dtype))
    _assert_close(ops.to_nchw(y, c), ref, dtype, 'pool+res')


@pytest.mark.parametrize('dtype', DTYPES)
def test_conv_dgrad_gate_bnstats(dtype):
    """Input-gradient form: transposed weights, pooled (upsample-adjoint) output, MC code on the
    output channels, ReLU-after-BN gate and the two BatchNorm-backward partial sums."""
    ops = _ops()
    g = torch.Generator().manual_seed(17)
    n, hs, ci, co = 4, 8, 24, 32                 # forward conv: ci -> co at resolution 2*hs
    dy = _rnd(g, n, co, 2 * hs, 2 * hs)
    wt = _rnd(g, co, ci, 3, 3) * 0.1
    xlow = _rnd(g, n, ci, hs, hs)                # the BN input at low resolution
    mean, rstd = _rnd(g, ci) * 0.2, torch.rand(ci, generator=g) + 0.5
    gamma, beta = _rnd(g, ci) * 0.5 + 1, _rnd(g, ci) * 0.3
    gscale, gshift = gamma * rstd, beta - mean * gamma * rstd
    code = (torch.rand(n, ci, generator=g) < 0.5).float()
    dm = F.conv_transpose2d(_q(dy, dtype), _q(wt, dtype), padding=1)              # [n, ci, 2hs, 2hs]
    du = dm * code.view(n, ci, 1, 1)
    da = F.avg_pool2d(du, 2) * 4
    xq = _q(xlow, dtype)
    z = xq * gscale.view(1, -1, 1, 1) + gshift.view(1, -1, 1, 1)
    dz = da * (z > 0)
    xhat = (xq - mean.view(1, -1, 1, 1)) * rstd.view(1, -1, 1, 1)
    wimg_t = ops.prep_weight(wt.cuda(), dtype, transpose=True)
    y, st = ops.conv_fused([ops.Seg(_nhwc(ops, dy, dtype))], wimg_t, ci, pool=True, alpha=1.0,
                           ocode=code.cuda(), gate_x=_nhwc(ops, xlow, dtype), gscale=gscale.cuda(), gshift=gshift.cuda(),
                           gmean=mean.cuda(), grstd=rstd.cuda(), stats_mode=2)
    _assert_close(ops.to_nchw(y, ci), dz, dtype, 'dz')
    s = st.sum(0).cpu()
    tol = dict(rtol=2e-3, atol=2e-3) if dtype == torch.float32 else dict(rtol=3e-2, atol=0.5)
    np.testing.assert_allclose(s[0, :ci], dz.sum((0, 2, 3)), **tol)
    np.testing.assert_allclose(s[1, :ci], (dz * xhat).sum((0, 2, 3)), **tol)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('n', [4, 128])        # 128 images of 32x32 = 131072 pixels: the 256-pixel skinny tile of large maps
def test_conv_tanh_small_cout(dtype, n):
    """G head: BN->ReLU->MC->conv3x3(C->3)->tanh (mcgan.py:55-60)."""
    ops = _ops()
    g = torch.Generator().manual_seed(19)
    c = 32
    x = _rnd(g, n, c, 32, 32)
    scale, shift = _rnd(g, c) * 0.5 + 1, _rnd(g, c) * 0.3
    code = (torch.rand(n, c, generator=g) < 0.5).float()
    wt, b = _rnd(g, 3, c, 3, 3) * 0.1, _rnd(g, 3) * 0.1
    a = ref_prologue(_q(x, dtype), scale, shift, True, code, False)
    if dtype != torch.float32:
        a = _q(a, dtype)
    ref = torch.tanh(F.conv2d(a, _q(wt, dtype), b, padding=1))
    y, _ = ops.conv_fused([ops.Seg(_nhwc(ops, x, dtype), scale=scale.cuda(), shift=shift.cuda(), code=code.cuda(), relu=True)],
                          ops.prep_weight(wt.cuda(), dtype), 3, bias=b.cuda(), tanh=True)
    assert y.shape[-1] == 8
    assert float(y[..., 3:].float().abs().max()) == 0.0          # padded channels stay exactly zero
    _assert_close(ops.to_nchw(y, 3), ref, dtype, 'head')


WG_CASES = [
    # N, H, Cin, Cout, ksize, ups(x), dy_ups
    (3, 8, 16, 24, 3, False, False),
    (2, 32, 40, 72, 3, False, False),
    (4, 16, 32, 32, 3, True, False),
    (4, 16, 32, 48, 1, True, False),
    (4, 16, 16, 16, 3, False, True),
    (7, 4, 24, 16, 3, False, False),
    (2, 32, 8, 32, 3, False, False),
    (16, 1, 128, 64, 1, False, False),
    (8, 16, 160, 80, 1, False, False),      # 1x1 with 5 chunks: one chunk group of 4 + a ragged one (producer/consumer path)
    (8, 16, 256, 48, 1, True, False),
    (8, 16, 128, 128, 1, False, True),      # D shortcut: 1x1 chunk group with the pooled (dy_ups) gradient
    # many tiles per split (8th entry = splits): the producer/consumer kernels; in bf16 the LDS-DMA ring form
    (16, 32, 40, 72, 3, False, False, 2),
    (16, 32, 64, 64, 3, True, False, 2),    # x through the x2 upsample
    (16, 32, 32, 64, 3, False, True, 2),    # dy through the x2 upsample (pooled gradient)
    (16, 16, 40, 24, 3, True, True, 2),
    (16, 32, 8, 128, 1, False, True, 4),    # D's first shortcut
    (8, 16, 96, 40, 1, True, False, 2),
]


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('case', WG_CASES)
def test_wgrad(case, dtype):
    ops = _ops()
    n, h, ci, co, ks, ups, dy_ups = case[:7]
    splits = case[7] if len(case) > 7 else None
    g = torch.Generator().manual_seed(23 + h + ci)
    hs = h // 2 if ups else h
    x = _rnd(g, n, ci, hs, hs)
    scale, shift = _rnd(g, ci) * 0.5 + 1, _rnd(g, ci) * 0.3
    code = (torch.rand(n, ci, generator=g) < 0.5).float()
    dy = _rnd(g, n, co, h // 2 if dy_ups else h, h // 2 if dy_ups else h)
    a = ref_prologue(_q(x, dtype), scale, shift, True, code, ups)
    if dtype != torch.float32:
        a = _q(a, dtype)
    dyf = _q(dy, dtype)
    if dy_ups:
        dyf = dyf.repeat_interleave(2, 2).repeat_interleave(2, 3)
    wt = torch.zeros(co, ci, ks, ks, requires_grad=True)
    (F.conv2d(a, wt, padding=ks // 2) * dyf).sum().backward()
    ref = wt.grad * 0.25
    grad = torch.full((co, ci, ks, ks), 1.0, device='cuda')
    seg = ops.Seg(_nhwc(ops, x, dtype), ksize=ks, scale=scale.cuda(), shift=shift.cuda(), code=code.cuda(), ups=ups, relu=True)
    bg, bg2 = torch.full((co,), 2.0, device='cuda'), torch.full((co,), 3.0, device='cuda')
    ops.wgrad(seg, _nhwc(ops, dy, dtype), co, ci, grad, dy_ups=dy_ups, alpha=0.25, accumulate=True, bias_grad=bg, bias_grad2=bg2,
              splits=splits)
    _assert_close(grad - 1.0, ref, dtype, 'wgrad')
    _assert_close(bg - 2.0, 0.25 * dyf.sum((0, 2, 3)), dtype, 'fused bias grad')
    _assert_close(bg2 - 3.0, 0.25 * dyf.sum((0, 2, 3)), dtype, 'fused bias grad (second output)')


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('mode', ['d_pair', 'g'])
def test_fused_tail_equals_the_separate_launches(dtype, mode):
    """mcgen_dtail_hinge_fused (tail forward + d hinge / d logit + tail input gradient, mcgan.py:158-165 with
    train_gan.py:154 / :172) against the three launches it replaces -- bit for bit -- and against plain torch; the loss
    value that mcgen_dtail_pair_wgrad_loss adds equals mcgen_hinge_d's."""
    ops = _ops()
    g = torch.Generator().manual_seed(53)
    n2, c, hw = 48, 128, 8
    x = _rnd(g, n2, c, hw, hw)
    code = (torch.rand(n2, c, generator=g) < 0.5).float() * 1.25
    w, b, sigma = _rnd(g, c) * 0.2, _rnd(g, 1), torch.tensor([1.7])
    xt = _nhwc(ops, x, dtype)
    logit0, pooled0 = ops.dtail_fwd(xt, code.cuda(), w.cuda(), b.cuda(), sigma.cuda())
    if mode == 'd_pair':
        loss0, _, _, dl0 = ops.hinge_d(logit0[:n2 // 2], logit0[n2 // 2:], both=True)
    else:
        loss0, dl0 = ops.hinge_g(logit0)
    dx0 = ops.dtail_bwd(dl0.contiguous(), xt, code.cuda(), w.cuda(), sigma.cuda(), pooled0, None, None)
    logit, pooled, dl, dx = ops.dtail_hinge_fused(xt, code.cuda(), w.cuda(), b.cuda(), sigma.cuda(), mode)
    assert torch.equal(logit, logit0) and torch.equal(pooled, pooled0) and torch.equal(dl, dl0.contiguous()) and torch.equal(dx, dx0)
    # against torch: pooled = code * sum relu(x); logit = pooled . w / sigma + b
    xq = _q(x, dtype)
    pref = torch.relu(xq).sum((2, 3)) * code
    np.testing.assert_allclose(pooled.cpu().numpy(), pref.numpy(), rtol=2e-5, atol=2e-4)
    lref = pref @ (w / sigma) + b
    np.testing.assert_allclose(logit.cpu().numpy(), lref.numpy(), rtol=1e-4, atol=1e-4)
    assert float(dx.float().abs().max()) > 0
    if mode == 'd_pair':
        outs = [torch.empty(c, device='cuda'), torch.empty(1, device='cuda'), torch.empty(c, device='cuda'), torch.empty(1, device='cuda')]
        ref_outs = [torch.empty_like(t) for t in outs]
        ratio = torch.tensor([1.25]).cuda()
        loss = torch.full((1,), -1.0, device='cuda')
        ops.dtail_pair_wgrad(dl, pooled, ratio, *outs, logit=logit, loss=loss)
        ops.dtail_pair_wgrad(dl, pooled, ratio, *ref_outs)
        assert all(torch.equal(a, r) for a, r in zip(outs, ref_outs))
        assert torch.equal(loss[0], loss0)
        href = torch.relu(1 - lref[:n2 // 2]).mean() + torch.relu(1 + lref[n2 // 2:]).mean()
        assert abs(float(loss) - float(href)) < 1e-4


def test_dtail_pair_wgrad_and_hinge_both():
    """Paired discriminator pass: tail weight / bias gradients of both halves in one launch (the second half divided by
    sigma_1 / sigma_2), and the hinge gradients as one [2N] tensor."""
    ops = _ops()
    g = torch.Generator().manual_seed(41)
    n, c = 24, 40
    dlogit, pooled = _rnd(g, 2 * n), _rnd(g, 2 * n, c)
    ratio = torch.tensor([1.25])
    outs = [torch.full((c,), 7.0, device='cuda'), torch.full((1,), 7.0, device='cuda'),
            torch.full((c,), 7.0, device='cuda'), torch.full((1,), 7.0, device='cuda')]
    ops.dtail_pair_wgrad(dlogit.cuda(), pooled.cuda(), ratio.cuda(), *outs)
    gp = dlogit.view(-1, 1).double() * pooled.double()
    np.testing.assert_allclose(outs[0].cpu().numpy(), gp[:n].sum(0).numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(outs[2].cpu().numpy(), (gp[n:].sum(0) / 1.25).numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(outs[1].cpu().numpy(), dlogit[:n].double().sum().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(outs[3].cpu().numpy(), dlogit[n:].double().sum().numpy(), rtol=1e-5, atol=1e-5)
    real, fake = _rnd(g, n).cuda(), _rnd(g, n).cuda()
    loss, dr, df, both = ops.hinge_d(real, fake, both=True)
    assert both.shape == (2 * n,) and torch.equal(both[:n], dr) and torch.equal(both[n:], df)
    loss3, dr3, df3 = ops.hinge_d(real, fake)
    assert torch.equal(loss, loss3) and torch.equal(dr, dr3) and torch.equal(df, df3)


@pytest.mark.parametrize('dtype', DTYPES)
def test_linear_as_conv_rowperm(dtype):
    """Generator Linear(128 -> C*16) viewed [N, C, 4, 4] (mcgan.py:51,66-67) as a 1x1 conv whose
    output rows are permuted so the NHWC result is [N, 4, 4, C]; and its weight/bias gradients."""
    ops = _ops()
    g = torch.Generator().manual_seed(29)
    n, lat, c = 6, 128, 16
    z, wt, b = _rnd(g, n, lat), _rnd(g, c * 16, lat) * 0.1, _rnd(g, c * 16)
    ref = F.linear(_q(z, dtype), _q(wt, dtype), b).view(n, c, 4, 4)
    zt = z.cuda().to(dtype).view(n, 1, 1, lat).contiguous()
    wimg = ops.prep_weight(wt.cuda(), dtype, row_perm=16)
    bperm = b.view(c, 16).t().contiguous().view(-1).cuda()
    y, st = ops.conv_fused([ops.Seg(zt, ksize=1)], wimg, c * 16, bias=bperm, stats_mode=1)
    x0 = y.view(n, 4, 4, c)
    _assert_close(ops.to_nchw(x0, c), ref, dtype, 'linear')
    dy = _rnd(g, n, c, 4, 4)
    dyt = ops.to_nhwc(dy.cuda(), dtype).view(n, 1, 1, c * 16)
    gw = torch.zeros(c * 16, lat, device='cuda')
    gb2 = torch.zeros(c * 16, device='cuda')
    ops.wgrad(ops.Seg(zt, ksize=1), dyt, c * 16, lat, gw, row_perm=16, bias_grad=gb2)
    refw = _q(dy, dtype).reshape(n, -1).t() @ _q(z, dtype)
    _assert_close(gw, refw, dtype, 'linear wgrad')
    _assert_close(gb2, _q(dy, dtype).reshape(n, -1).sum(0), dtype, 'linear bias grad fused in wgrad')
    gb = torch.zeros(c * 16, device='cuda')
    ops.colsum(dyt, c * 16, gb, row_perm=16)
    _assert_close(gb, _q(dy, dtype).reshape(n, -1).sum(0), dtype, 'linear bias grad')


def test_mc_code_and_apply():
    import golden_util as gu
    ops = _ops()
    d = gu.load_npz('mc_unit.npz')
    cb = torch.from_numpy(d['codebook']).cuda()
    ind = F.one_hot(torch.from_numpy(d['label']), 10).float().cuda()
    code = ops.mc_code(ind, cb)
    assert torch.equal(code.cpu(), torch.from_numpy(d['codebook'])[torch.from_numpy(d['label'])])
    x = ops.to_nhwc(torch.from_numpy(d['x4']).cuda(), torch.float32)
    out = ops.to_nchw(ops.mc_apply(x, code), 32)
    assert np.array_equal(out.cpu().numpy(), d['out4'])                    # bit exact (one multiply)
    soft = ops.mc_code(torch.from_numpy(d['soft']).cuda(), cb)
    np.testing.assert_allclose(ops.to_nchw(ops.mc_apply(x, soft), 32).cpu().numpy(), d['out_soft'], rtol=1e-6, atol=1e-6)


def test_label_gather_equals_indicator_product():
    """CodeBatch.run_labels (mcgen_mc_gather_batch: codebook rows gathered by label, the label vector repeating `reps` times,
    the second half of the batch scaled per job) is bit for bit CodeBatch.run on the one-hot indicator (modules.py:73:
    indicator @ codebook) -- with 100 modes (COIL100) as with 10; run_any picks it when the indicator carries the hint."""
    ops = _ops()
    g = torch.Generator().manual_seed(77)

    class MC:
        pass
    for modes in (10, 100):
        mcs = []
        for c in (8, 64, 128, 512):
            m = MC(); m.codebook = (torch.rand(modes, c, generator=g) < 0.5).float().cuda(); mcs.append(m)
        n, reps = 24, 2
        label = torch.randint(0, modes, (n,), generator=g).cuda()
        ind = F.one_hot(label, modes).float().repeat(reps, 1)
        scale = (torch.rand(3, generator=g) + 0.5).cuda()
        cb = ops.CodeBatch(mcs, [0, -1, 2, 1])
        a = cb.run(ind, scale, n)
        b = cb.run_labels(label, reps, scale, n)
        hinted = ops.onehot_hint(ind.clone(), label, reps)
        c_ = cb.run_any(hinted, scale, n)
        for x, y, z in zip(a, b, c_):
            assert torch.equal(x, y) and torch.equal(x, z)
        assert float((a[0][n:] / a[0][:n].clamp_min(1e-9)).max()) > 1.0 or float(scale[0]) <= 1.0
        plain = cb.run_any(ind, scale, n)                   # no hint: the product path
        assert all(torch.equal(x, y) for x, y in zip(a, plain))
        # the one launch that builds the repeated indicator also hands out the labels as int32 (conv_fused's wsel)
        lab32 = torch.full((3 * n,), -1, dtype=torch.int32, device='cuda')
        rep3 = ops.onehot_rep(label, modes, 3, lab32=lab32)
        assert torch.equal(rep3, F.one_hot(label, modes).float().repeat(3, 1)) and torch.equal(lab32, label.int().repeat(3))


def test_bn_finalize_and_backward():
    ops = _ops()
    g = torch.Generator().manual_seed(31)
    n, c, h = 8, 16, 8
    x = (_rnd(g, n, c, h, h) * 2 + 0.5).requires_grad_(True)
    gamma, beta = (_rnd(g, c) * 0.2 + 1).requires_grad_(True), (_rnd(g, c) * 0.1).requires_grad_(True)
    rm, rv = torch.zeros(c), torch.ones(c)
    y = F.batch_norm(x, rm, rv, gamma, beta, True, 0.1, 1e-5)
    dzr = _rnd(g, n, c, h, h)
    y.backward(dzr)
    xt = ops.to_nhwc(x.detach().cuda(), torch.float32)
    xf = xt.view(-1, c)
    part = torch.stack([xf.sum(0), (xf * xf).sum(0)]).view(1, 2, c).contiguous()
    rmg, rvg = torch.zeros(c, device='cuda'), torch.ones(c, device='cuda')
    scale, shift, mean, rstd = ops.bn_finalize(part, n * h * h, gamma.detach().cuda(), beta.detach().cuda(), rmg, rvg)
    np.testing.assert_allclose(rmg.cpu(), rm, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(rvg.cpu(), rv, rtol=1e-5, atol=1e-6)
    yk = xt * scale + shift
    np.testing.assert_allclose(ops.to_nchw(yk, c).cpu(), y.detach(), rtol=1e-4, atol=1e-5)
    dz = ops.to_nhwc(dzr.cuda(), torch.float32)
    xhat = (xt - mean) * rstd
    bpart = torch.stack([dz.view(-1, c).sum(0), (dz * xhat).view(-1, c).sum(0)]).view(1, 2, c).contiguous()
    dg, db = torch.zeros(c, device='cuda'), torch.zeros(c, device='cuda')
    dx = ops.bn_backward(bpart, dz, xt, n * h * h, scale, mean, rstd, dg, db)
    np.testing.assert_allclose(ops.to_nchw(dx, c).cpu(), x.grad, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(dg.cpu(), gamma.grad, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(db.cpu(), beta.grad, rtol=1e-4, atol=1e-4)


def test_spectral_norm_kernels():
    """mcgen_sn_power_iter / mcgen_sn_grad_fix vs torch.nn.utils.spectral_norm on CPU."""
    ops = _ops()
    torch.manual_seed(5)
    convs = [torch.nn.Conv2d(3, 16, 3), torch.nn.Conv2d(16, 16, 3), torch.nn.Linear(16, 1), torch.nn.Conv2d(16, 24, 1)]
    sn = [torch.nn.utils.spectral_norm(m) for m in convs]
    flat_w, flat_uv, layers = [], [], []
    wo = uo = 0
    for m in sn:
        w = m.weight_orig.detach()
        rows, cols = w.shape[0], w[0].numel()
        layers.append((wo, uo, uo + rows, rows, cols))
        flat_w.append(w.reshape(-1)); flat_uv += [m.weight_u.detach().clone(), m.weight_v.detach().clone()]
        wo += w.numel(); uo += rows + cols
    W = torch.cat(flat_w).cuda(); UV = torch.cat(flat_uv).cuda()
    ld = ops.sn_layers_tensor(layers, 'cuda')
    sigma = torch.zeros(len(sn), device='cuda')
    for m in sn:
        m.train()
    # one training forward on CPU advances u, v and defines weight = W / sigma
    xs = [torch.randn(2, 3, 8, 8), torch.randn(2, 16, 8, 8), torch.randn(2, 16), torch.randn(2, 16, 4, 4)]
    outs = [m(x) for m, x in zip(sn, xs)]
    ops.sn_power_iter(W, UV, ld, len(sn), True, sigma, 24, 144)
    for i, m in enumerate(sn):
        wo_, uo_, vo_, r, c = layers[i]
        np.testing.assert_allclose(UV[uo_:uo_ + r].cpu(), m.weight_u.detach(), rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(UV[vo_:vo_ + c].cpu(), m.weight_v.detach(), rtol=1e-4, atol=1e-6)
        ref_sigma = (m.weight_orig.detach().reshape(r, c) / m.weight.detach().reshape(r, c)).flatten()
        ref_sigma = ref_sigma[torch.isfinite(ref_sigma)].median()
        np.testing.assert_allclose(float(sigma[i]), float(ref_sigma), rtol=1e-4)
    # gradient through W / sigma
    gs = [torch.randn_like(m.weight) for m in sn]
    for m, o, x, gw in zip(sn, outs, xs, gs):
        m.zero_grad()
        (m.weight * gw).sum().backward()
    G = torch.cat([g.reshape(-1) for g in gs]).cuda()
    ops.sn_grad_fix(G, G, W, UV, ld, len(sn), sigma)
    off = 0
    for m in sn:
        k = m.weight_orig.numel()
        np.testing.assert_allclose(G[off:off + k].cpu(), m.weight_orig.grad.reshape(-1), rtol=2e-4, atol=2e-5)
        off += k
    # the paired form (both halves of a discriminator update in one dot + one apply launch) == two single-half calls (to an
    # fp32 rounding step: the compiler contracts the sum differently)
    Ga, Gb = torch.randn_like(G), torch.randn_like(G)
    UV2, sigma2 = UV * 1.01, sigma * 0.97
    d1, d2 = torch.randn_like(G), None
    d2 = d1.clone()
    ops.sn_grad_fix(Ga, d1, W, UV, ld, len(sn), sigma, accumulate=True)
    ops.sn_grad_fix(Gb, d1, W, UV2, ld, len(sn), sigma2, accumulate=True)
    ops.sn_grad_fix_pair(Ga, Gb, d2, W, UV, UV2, ld, len(sn), sigma, sigma2, accumulate=True)
    np.testing.assert_allclose(d2.cpu(), d1.cpu(), rtol=2e-6, atol=2e-6)
    ops.sn_grad_fix(Ga, d1, W, UV, ld, len(sn), sigma)
    ops.sn_grad_fix(Gb, d1, W, UV2, ld, len(sn), sigma2, accumulate=True)
    ops.sn_grad_fix_pair(Ga, Gb, d2, W, UV, UV2, ld, len(sn), sigma, sigma2)
    np.testing.assert_allclose(d2.cpu(), d1.cpu(), rtol=2e-6, atol=2e-6)
    # the fused form: two rounds in one launch == two launches of the four-kernel form, snapshots included
    UVa, UVb = UV.clone(), UV.clone()
    sa1, sa2 = torch.zeros_like(sigma), torch.zeros_like(sigma)
    ops.sn_power_iter(W, UVa, ld, len(sn), True, sa1, 24, 144)
    snap1 = UVa.clone()
    ops.sn_power_iter(W, UVa, ld, len(sn), True, sa2, 24, 144)
    sf, snap = ops.sn_power_iter_fused(W, UVb, ld, len(sn), 2, True, 24, 144)
    np.testing.assert_allclose(sf[0].cpu(), sa1.cpu(), rtol=1e-5)
    np.testing.assert_allclose(sf[1].cpu(), sa2.cpu(), rtol=1e-5)
    np.testing.assert_allclose(UVb.cpu(), UVa.cpu(), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(snap[0].cpu(), snap1.cpu(), rtol=1e-4, atol=1e-6)
    assert torch.equal(snap[1], UVb)
    se, _ = ops.sn_power_iter_fused(W, UVb, ld, len(sn), 1, False, 24, 144, snapshot=False)
    np.testing.assert_allclose(se[0].cpu(), sf[1].cpu(), rtol=1e-5)
    # the rounds form (2 launches per round + 1; the next round's u formed from the previous round's t): same numbers
    for rounds in (1, 2, 3):
        UVc = UV.clone()
        UVr = UV.clone()
        refs = []
        for _ in range(rounds):
            sr = torch.zeros_like(sigma)
            ops.sn_power_iter(W, UVr, ld, len(sn), True, sr, 24, 144)
            refs.append((sr, UVr.clone()))
        sg, sp = ops.sn_power_iter_rounds(W, UVc, ld, len(sn), rounds, 24, 144)
        for r in range(rounds):
            np.testing.assert_allclose(sg[r].cpu(), refs[r][0].cpu(), rtol=1e-5)
            np.testing.assert_allclose(sp[r].cpu(), refs[r][1].cpu(), rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(UVc.cpu(), UVr.cpu(), rtol=1e-4, atol=1e-6)
        assert torch.equal(sp[rounds - 1], UVc)
        if rounds >= 2:                                   # the extra row: sigma of the last round but one over the last round's
            np.testing.assert_allclose(sg[rounds].cpu(), (sg[rounds - 2] / sg[rounds - 1]).cpu(), rtol=1e-6)
    # eval mode: no iteration, same sigma formula
    sig2 = torch.zeros_like(sigma)
    UV2 = UV.clone()
    ops.sn_power_iter(W, UV2, ld, len(sn), False, sig2, 24, 144)
    assert torch.equal(UV, UV2)
    np.testing.assert_allclose(sig2.cpu(), sigma.cpu(), rtol=1e-4)


def test_dtail_hinge_tanh_adam():
    ops = _ops()
    g = torch.Generator().manual_seed(37)
    n, c, h = 6, 16, 4
    x = _rnd(g, n, c, h, h).requires_grad_(True)
    code = (torch.rand(n, c, generator=g) < 0.5).float()
    w, b = _rnd(g, 1, c).requires_grad_(True), _rnd(g, 1).requires_grad_(True)
    sigma = torch.tensor([1.7])
    pooled_ref = (torch.relu(x) * code.view(n, c, 1, 1)).sum((2, 3))
    logit_ref = F.linear(pooled_ref, w / sigma, b).view(-1)
    fake_ref = (logit_ref * 0.5 - 1.2).detach()
    loss_ref = torch.relu(1 - logit_ref).mean() + torch.relu(1 + fake_ref).mean()
    loss_ref.backward()
    xt = ops.to_nhwc(x.detach().cuda(), torch.float32)
    sg = sigma.cuda()
    logit, pooled = ops.dtail_fwd(xt, code.cuda(), w.detach().view(-1).cuda(), b.detach().cuda(), sg)
    np.testing.assert_allclose(logit.cpu(), logit_ref.detach(), rtol=1e-5, atol=1e-5)
    loss, dreal, dfake = ops.hinge_d(logit, fake_ref.cuda())
    np.testing.assert_allclose(float(loss), float(loss_ref), rtol=1e-6)
    dw, db = torch.zeros(c, device='cuda'), torch.zeros(1, device='cuda')
    dx = ops.dtail_bwd(dreal, xt, code.cuda(), w.detach().view(-1).cuda(), sg, pooled, dw, db)
    np.testing.assert_allclose(ops.to_nchw(dx, c).cpu(), x.grad, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(dw.cpu() / 1.7, w.grad.view(-1), rtol=1e-5, atol=1e-6)   # dw is wrt W/sigma
    np.testing.assert_allclose(db.cpu(), b.grad, rtol=1e-5, atol=1e-6)
    lg, dfg = ops.hinge_g(logit)
    np.testing.assert_allclose(float(lg), float(-logit_ref.mean()), rtol=1e-6)
    assert torch.allclose(dfg.cpu(), torch.full((n,), -1.0 / n))
    # tanh backward
    t = torch.tanh(_rnd(g, 4, 8, 8, 8)); dy = _rnd(g, 4, 8, 8, 8)
    np.testing.assert_allclose(ops.tanh_bwd(dy.cuda(), t.cuda()).cpu(), dy * (1 - t * t), rtol=1e-6, atol=1e-6)
    # Adam: three steps against torch.optim.Adam
    p = _rnd(g, 1000).requires_grad_(True)
    opt = torch.optim.Adam([p], lr=2e-4, betas=(0.5, 0.999))
    pg, m, v = p.detach().clone().cuda(), torch.zeros(1000, device='cuda'), torch.zeros(1000, device='cuda')
    step = torch.zeros(2, dtype=torch.int64, device='cuda')          # {step counter, launch ticket}
    for _ in range(3):
        gr = _rnd(g, 1000)
        p.grad = gr.clone(); opt.step()
        ops.adam(pg, gr.cuda(), m, v, step, 2e-4, (0.5, 0.999))
    assert step.tolist() == [3, 0]                                    # the last workgroup of each launch bumps the counter
    np.testing.assert_allclose(pg.cpu(), p.detach(), rtol=1e-6, atol=1e-7)


# --------------------------------------------------------------------------------------------------------- #
# The tile / pipeline instantiations the headline bench dispatches (conv_fused.hip pick_tile): every case runs
# at a size that crosses the policy's thresholds, asserts which tile was picked and compares with F.conv2d on the
# CPU.  bf16 reference: operands rounded to bf16, fp32 math.
def _conv_logged(ops, *a, **k):
    ops.TILE_LOG = []
    try:
        out = ops.conv_fused(*a, **k)
        tiles = list(ops.TILE_LOG)
    finally:
        ops.TILE_LOG = None
    return out, tiles


BIG_A = [
    # N, H(out), C, tile expected in bf16, tile expected in fp32
    (128, 32, 256, (256, 256), (128, 128)),
    (64, 32, 256, (256, 256), (128, 128)),
    (128, 16, 256, (128, 256), (128, 128)),
    (128, 16, 128, (64, 128), (128, 128)),
]


@pytest.mark.parametrize('probe', PROBES)
@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('case', BIG_A)
def test_big_conv_a_upsample_bn_code_stats(case, dtype, probe):
    """G.conv_a at the bench's sizes: BN -> ReLU -> x2 upsample -> MC code -> conv3x3 + bias, next-BN sums."""
    ops = _ops()
    n, h, c, t16, t32 = case
    if dtype == torch.float32 and n * h * h > 65536:
        pytest.skip('fp32 parity instantiation covered at N=64 (same tile, half the CPU reference time)')
    if probe and dtype == torch.float32:
        pytest.skip('border probes target the bf16 forms the bench times')
    g = torch.Generator().manual_seed(101 + n + h + c)
    hs = h // 2
    x = _rnd(g, n, c, hs, hs)
    if probe:
        x = _hot(x)
    scale, shift = _rnd(g, c) * 0.5 + 1, _rnd(g, c) * 0.3
    code = (torch.rand(n, c, generator=g) < 0.5).float()
    wt, b = _rnd(g, c, c, 3, 3) * 0.03, _rnd(g, c)
    a = ref_prologue(_q(x, dtype), scale, shift, True, code, True)
    ref = F.conv2d(_q(a, dtype) if dtype != torch.float32 else a, _q(wt, dtype), b, padding=1)
    seg = ops.Seg(_nhwc(ops, x, dtype), scale=scale.cuda(), shift=shift.cuda(), code=code.cuda(), ups=True, relu=True)
    (y, st), tiles = _conv_logged(ops, [seg], ops.prep_weight(wt.cuda(), dtype), c, bias=b.cuda(), stats_mode=1)
    assert tiles == [t16 if dtype == torch.bfloat16 else t32], tiles
    yq = ops.to_nchw(y, c).cpu()
    _assert_close(yq, ref, dtype, 'big conv_a')
    s = st.double().sum(0).cpu()
    tol = dict(rtol=1e-3, atol=1e-3 * float(ref.abs().sum((0, 2, 3)).max())) if dtype == torch.float32 else \
        dict(rtol=2e-2, atol=4e-3 * float(ref.abs().sum((0, 2, 3)).max()))
    np.testing.assert_allclose(s[0, :c], ref.double().sum((0, 2, 3)), **tol)
    np.testing.assert_allclose(s[1, :c], (ref.double() ** 2).sum((0, 2, 3)), rtol=tol['rtol'] * 2,
                               atol=tol['atol'] * float(ref.abs().max()))


@pytest.mark.parametrize('probe', PROBES)
@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('n,h,c,t16', [(128, 32, 256, (256, 256)), (128, 16, 256, (128, 256))])
def test_big_conv_b_with_shortcut_segment(n, h, c, t16, dtype, probe):
    """G.conv_b at the bench's sizes: conv3x3(BN/ReLU/MC2(h)) + conv1x1(MC1(Up(x))) as one K-concatenated launch."""
    ops = _ops()
    if dtype == torch.float32:
        if probe:
            pytest.skip('border probes target the bf16 forms the bench times')
        n //= 2
    g = torch.Generator().manual_seed(211 + h)
    hs = h // 2
    h_in, x = _rnd(g, n, c, h, h), _rnd(g, n, c, hs, hs)
    if probe:
        h_in, x = _hot(h_in), _hot(x, 8.0)
    scale, shift = _rnd(g, c) * 0.5 + 1, _rnd(g, c) * 0.3
    code1 = (torch.rand(n, c, generator=g) < 0.5).float()
    code2 = (torch.rand(n, c, generator=g) < 0.5).float()
    w2, ws, b = _rnd(g, c, c, 3, 3) * 0.03, _rnd(g, c, c, 1, 1) * 0.08, _rnd(g, c)
    a2 = ref_prologue(_q(h_in, dtype), scale, shift, True, code2, False)
    a1 = ref_prologue(_q(x, dtype), None, None, False, code1, True)
    if dtype != torch.float32:
        a2, a1 = _q(a2, dtype), _q(a1, dtype)
    ref = F.conv2d(a2, _q(w2, dtype), b, padding=1) + F.conv2d(a1, _q(ws, dtype))
    img = torch.cat([ops.prep_weight(w2.cuda(), dtype), ops.prep_weight(ws.cuda(), dtype)])
    segs = [ops.Seg(_nhwc(ops, h_in, dtype), scale=scale.cuda(), shift=shift.cuda(), code=code2.cuda(), relu=True),
            ops.Seg(_nhwc(ops, x, dtype), ksize=1, code=code1.cuda(), ups=True)]
    (y, _), tiles = _conv_logged(ops, segs, img, c, bias=b.cuda())
    assert tiles == [t16 if dtype == torch.bfloat16 else ((128, 128) if n * h * h > 16384 else (64, 64))], tiles
    _assert_close(ops.to_nchw(y, c), ref, dtype, 'big conv_b + shortcut')


@pytest.mark.parametrize('probe', PROBES)
@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('pool', [False, True])
def test_big_dgrad_gate_bnstats(dtype, pool, probe):
    """Input-gradient launches of G's 32x32 layers at the bench's batch: transposed weights, MC code on the output
    channels, ReLU-after-BN gate, the two BatchNorm-backward sums; pool=True is conv_a's form (x2-upsample adjoint)."""
    ops = _ops()
    n, h, c = (128 if dtype == torch.bfloat16 else 64), 32, 256
    g = torch.Generator().manual_seed(307 + pool)
    ho = h // 2 if pool else h
    dy = _rnd(g, n, c, h, h)
    if probe:
        if dtype == torch.float32:
            pytest.skip('border probes target the bf16 forms the bench times')
        dy = _hot(dy)
    wt = _rnd(g, c, c, 3, 3) * 0.03
    xin = _rnd(g, n, c, ho, ho)
    mean, rstd = _rnd(g, c) * 0.2, torch.rand(c, generator=g) + 0.5
    gamma, beta = _rnd(g, c) * 0.5 + 1, _rnd(g, c) * 0.3
    gscale, gshift = gamma * rstd, beta - mean * gamma * rstd
    code = (torch.rand(n, c, generator=g) < 0.5).float()
    dm = F.conv_transpose2d(_q(dy, dtype), _q(wt, dtype), padding=1) * code.view(n, c, 1, 1)
    da = F.avg_pool2d(dm, 2) * 4 if pool else dm
    xq = _q(xin, dtype)
    dz = da * ((xq * gscale.view(1, -1, 1, 1) + gshift.view(1, -1, 1, 1)) > 0)
    xhat = (xq - mean.view(1, -1, 1, 1)) * rstd.view(1, -1, 1, 1)
    (y, st), tiles = _conv_logged(ops, [ops.Seg(_nhwc(ops, dy, dtype))], ops.prep_weight(wt.cuda(), dtype, transpose=True), c,
                                  pool=pool, alpha=1.0, ocode=code.cuda(), gate_x=_nhwc(ops, xin, dtype), gscale=gscale.cuda(),
                                  gshift=gshift.cuda(), gmean=mean.cuda(), grstd=rstd.cuda(), stats_mode=2)
    assert tiles == [(256, 256) if dtype == torch.bfloat16 else (128, 128)], tiles
    _assert_close(ops.to_nchw(y, c), dz, dtype, 'big dz')
    s = st.double().sum(0).cpu()
    mag = float(dz.abs().sum((0, 2, 3)).max())
    tol = dict(rtol=1e-3, atol=2e-4 * mag) if dtype == torch.float32 else dict(rtol=3e-2, atol=4e-3 * mag)
    np.testing.assert_allclose(s[0, :c], dz.double().sum((0, 2, 3)), **tol)
    np.testing.assert_allclose(s[1, :c], (dz * xhat).double().sum((0, 2, 3)), rtol=tol['rtol'], atol=tol['atol'] * 3)


@pytest.mark.parametrize('dtype', DTYPES)
def test_pool2_sum_and_shortcut_gradients_at_low_resolution(dtype):
    """mcgen_pool2_sum (adjoint of the nearest x2 upsample) and what the generator's backward uses it for: the weight
    gradient and the input gradient of the shortcut conv1x1(Up(x)) (mcgan.py:26-30,42) taken at x's resolution from
    the pooled dy equal the literal full-resolution forms."""
    ops = _ops()
    g = torch.Generator().manual_seed(611)
    n, h, c = (32, 32, 256) if dtype == torch.bfloat16 else (4, 16, 32)
    dy, xlo = _rnd(g, n, c, h, h), _rnd(g, n, c, h // 2, h // 2)
    code = (torch.rand(n, c, generator=g) < 0.5).float()
    ws = _rnd(g, c, c, 1, 1) * 0.05
    dyt, xt = _nhwc(ops, dy, dtype), _nhwc(ops, xlo, dtype)
    lo = ops.pool2_sum(dyt)
    ref_lo = F.avg_pool2d(_q(dy, dtype), 2) * 4
    _assert_close(ops.to_nchw(lo, c), _q(ref_lo, dtype) if dtype == torch.bfloat16 else ref_lo, dtype, 'pool2_sum')
    # weight gradient: sum_px dy (x) Up(x * code)  ==  sum_q pool(dy) (x) (x * code)
    g_full, g_low = torch.zeros(c, c, 1, 1, device='cuda'), torch.zeros(c, c, 1, 1, device='cuda')
    ops.wgrad(ops.Seg(xt, ksize=1, code=code.cuda(), ups=True), dyt, c, c, g_full)
    ops.wgrad(ops.Seg(xt, ksize=1, code=code.cuda()), lo, c, c, g_low)
    a = (_q(xlo, dtype) * code.view(n, c, 1, 1))
    a = _q(a, dtype) if dtype == torch.bfloat16 else a
    ref_w = torch.einsum('nohw,nihw->oi', _q(ref_lo, dtype) if dtype == torch.bfloat16 else ref_lo, a)
    tol = dict(rtol=3e-2, atol=0.5) if dtype == torch.bfloat16 else dict(rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(g_low.view(c, c).cpu(), ref_w, **tol)
    # (bf16: the pooled dy is rounded once more -- an error of ~ sqrt(pixels) * 2^-9 * |dy| |x| on sums of magnitude ~200)
    np.testing.assert_allclose(g_low.cpu(), g_full.cpu(), rtol=tol['rtol'], atol=4 * tol['atol'])
    # input gradient: pool(conv1x1^T(dy)) * code  ==  conv1x1^T(pool(dy)) * code
    wt = ops.prep_weight(ws.cuda(), dtype, transpose=True)
    d_full, _ = ops.conv_fused([ops.Seg(dyt, ksize=1)], wt, c, pool=True, alpha=1.0, ocode=code.cuda())
    d_low, _ = ops.conv_fused([ops.Seg(lo, ksize=1)], wt, c, ocode=code.cuda())
    ref_d = F.conv2d(_q(ref_lo, dtype) if dtype == torch.bfloat16 else ref_lo, _q(ws, dtype).permute(1, 0, 2, 3)) * code.view(n, c, 1, 1)
    _assert_close(ops.to_nchw(d_low, c), ref_d, dtype, 'shortcut dgrad from pooled dy')
    _assert_close(ops.to_nchw(d_full, c), ref_d, dtype, 'shortcut dgrad, literal form')


def test_pipelined_forms_with_a_negative_code_under_relu():
    """The software-pipelined bf16 convolution and the ring weight-gradient kernel fold the MultimodalController code into
    the affine in front of the ReLU, which equals code * relu(.) only for code >= 0.  MC codes are never negative
    (modules.py:58-76), but the C ABI takes any float: the convolution must FAIL LOUDLY (NaN for the tiles of the image that
    carries the entry, every other image exact), the weight gradient takes its literal form for such a tile and stays exact."""
    ops = _ops()
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(29)
    n, h, c = 64, 32, 256
    x = _rnd(g, n, c, h, h)
    wt, b = _rnd(g, c, c, 3, 3) * 0.03, _rnd(g, c)
    code = (torch.rand(n, c, generator=g) < 0.5).float()
    code[3, 17], code[5, 200] = -1.0, -0.5
    xt = _nhwc(ops, x, dtype)
    (y, _), tiles = _conv_logged(ops, [ops.Seg(xt, code=code.cuda(), relu=True)], ops.prep_weight(wt.cuda(), dtype), c, bias=b.cuda())
    assert tiles == [(256, 256)], tiles
    out = ops.to_nchw(y, c).cpu()
    a = _q(ref_prologue(_q(x, dtype), None, None, True, code, False), dtype)
    ref = F.conv2d(a, _q(wt, dtype), None, padding=1) + b.view(1, -1, 1, 1)
    keep = [i for i in range(n) if i not in (3, 5)]
    assert torch.isnan(out[3]).all() and torch.isnan(out[5]).all() and torch.isfinite(out[keep]).all()
    _assert_close(out[keep], ref[keep], dtype, 'pp form, images without a negative code')
    # without the ReLU the fold is exact for any sign
    y2, _ = ops.conv_fused([ops.Seg(xt, code=code.cuda())], ops.prep_weight(wt.cuda(), dtype), c, bias=b.cuda())
    a2 = _q(_q(x, dtype) * code.view(n, c, 1, 1), dtype)
    _assert_close(ops.to_nchw(y2, c), F.conv2d(a2, _q(wt, dtype), None, padding=1) + b.view(1, -1, 1, 1), dtype, 'pp form, no ReLU')
    # ring weight gradient on the same operands: exact
    dy = _rnd(g, n, c, h, h)
    gw = torch.zeros(c, c, 3, 3, device='cuda')
    ops.wgrad(ops.Seg(xt, code=code.cuda(), relu=True), _nhwc(ops, dy, dtype), c, c, gw)
    ref_w = torch.nn.grad.conv2d_weight(a, (c, c, 3, 3), _q(dy, dtype), padding=1)
    np.testing.assert_allclose(gw.cpu(), ref_w, rtol=3e-2, atol=0.6)


def test_big_conv_image_input_dma3_tile():
    """D's first convolution (3 image channels padded to 8, conv3x3 -> 128 at 32x32, N = 256): too few input channels for
    the pipelined form, so the dma3 form's 128 x 128 tile -- the instantiation every other 128-channel launch of the
    bench has left."""
    ops = _ops()
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(977)
    n, h, c = 256, 32, 128
    x = _rnd(g, n, 3, h, h)
    wt, b = _rnd(g, c, 3, 3, 3) * 0.2, _rnd(g, c)
    ref = F.conv2d(_q(x, dtype), _q(wt, dtype), None, padding=1) + b.view(1, -1, 1, 1)
    (y, _), tiles = _conv_logged(ops, [ops.Seg(_nhwc(ops, x, dtype))], ops.prep_weight(wt.cuda(), dtype), c, bias=b.cuda())
    assert tiles == [(128, 128)], tiles
    _assert_close(ops.to_nchw(y, c), ref, dtype, 'image-input conv')


@pytest.mark.parametrize('probe', PROBES)
@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('n,h,t16', [(128, 32, (256, 128)), (256, 32, (256, 128)), (256, 16, (256, 128)), (128, 16, (64, 128))])
def test_big_conv_pool_residual(n, h, t16, dtype, probe):
    """D's 128-channel layers at the bench's batch (128 images, 256 for the paired real + fake pass):
    ReLU -> MC -> conv3x3 -> AvgPool2 + residual.  bf16 from 65536 pixels up: the software-pipelined 256 x 128 tile."""
    ops = _ops()
    if dtype == torch.float32 and n > 128:
        pytest.skip('fp32: same tile as the N=128 case')
    if probe and dtype == torch.float32:
        pytest.skip('border probes target the bf16 forms the bench times')
    g = torch.Generator().manual_seed(401 + n + h)
    c = 128
    x, res = _rnd(g, n, c, h, h), _rnd(g, n, c, h // 2, h // 2)
    if probe:
        x = _hot(x)
    code = (torch.rand(n, c, generator=g) < 0.5).float() * (1.0 + 0.25 * (torch.arange(n) >= n // 2).float().view(n, 1))
    wt, b = _rnd(g, c, c, 3, 3) * 0.04, _rnd(g, c)
    a = ref_prologue(_q(x, dtype), None, None, True, code, False)
    if dtype != torch.float32:
        a = _q(a, dtype)
    ref = F.avg_pool2d(F.conv2d(a, _q(wt, dtype), None, padding=1), 2) + b.view(1, -1, 1, 1) + _q(res, dtype)
    (y, _), tiles = _conv_logged(ops, [ops.Seg(_nhwc(ops, x, dtype), code=code.cuda(), relu=True)], ops.prep_weight(wt.cuda(), dtype),
                                 c, bias=b.cuda(), pool=True, alpha=0.25, res=_nhwc(ops, res, dtype))
    want = t16 if dtype == torch.bfloat16 else ((128, 128) if n * h * h > 16384 else (64, 64))
    assert tiles == [want], tiles
    _assert_close(ops.to_nchw(y, c), ref, dtype, 'big pool+res')


@pytest.mark.parametrize('dtype', DTYPES)
def test_deep_1x1_grouped_form(dtype):
    """K-deep pure 1x1 launches (MCGlow's 512 -> 512 coupling convolution, MCPixelCNN's head): in bf16 the 'dma3g'
    form (three chunks per barrier round, conv_fused.hip) -- C = 512 >= 384."""
    ops = _ops()
    g = torch.Generator().manual_seed(503)
    n, h, c = 16, 16, 512
    x = _rnd(g, n, c, h, h)
    scale, shift = _rnd(g, c) * 0.5 + 1, _rnd(g, c) * 0.3
    code = (torch.rand(n, c, generator=g) < 0.5).float()
    wt, b = _rnd(g, c, c, 1, 1) * 0.05, _rnd(g, c)
    a = ref_prologue(_q(x, dtype), scale, shift, True, code, False)
    if dtype != torch.float32:
        a = _q(a, dtype)
    ref = F.conv2d(a, _q(wt, dtype), b)
    seg = ops.Seg(_nhwc(ops, x, dtype), ksize=1, scale=scale.cuda(), shift=shift.cuda(), code=code.cuda(), relu=True)
    (y, _), tiles = _conv_logged(ops, [seg], ops.prep_weight(wt.cuda(), dtype), c, bias=b.cuda())
    assert tiles == [(64, 64)], tiles
    _assert_close(ops.to_nchw(y, c), ref, dtype, 'deep 1x1')
    # a ragged chunk count (13 chunks: four full groups of three + one single)
    c2 = 416
    x2, w2 = _rnd(g, n, c2, h, h), _rnd(g, 96, c2, 1, 1) * 0.05
    ref2 = F.conv2d(_q(x2, dtype), _q(w2, dtype))
    y2, _ = ops.conv_fused([ops.Seg(_nhwc(ops, x2, dtype), ksize=1)], ops.prep_weight(w2.cuda(), dtype), 96)
    _assert_close(ops.to_nchw(y2, 96), ref2, dtype, 'deep 1x1 ragged')


@pytest.mark.parametrize('case', ['two_seg_stats', 'pooled', 'gated', 'ups_groups'])
def test_64_channel_layers_on_large_maps(case):
    """COIL100's 64-channel convolutions on 32x32 / 16x16 maps (G [512, 256, 128, 64], D [64, 128, 256, 512]) take the
    128-pixel x 64-channel four-wave tile from 65536 pixels up (pick_tile; the 64 x 64 tile below that): forward with
    BatchNorm partial sums and a fused 1x1 shortcut, the pooled first-block form, the ReLU-gated input gradient, the
    upsampled input with BatchNorm statistics groups -- each against F.conv2d."""
    ops = _ops()
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(977)
    n, h, c = 80, 32, 64
    code = (torch.rand(n, c, generator=g) < 0.5).float()
    b = _rnd(g, c)
    if case == 'two_seg_stats':
        x, xl = _hot(_rnd(g, n, c, h, h)), _rnd(g, n, 128, h // 2, h // 2)
        sc, sh = _rnd(g, c) * 0.5 + 1, _rnd(g, c) * 0.3
        code1 = (torch.rand(n, 128, generator=g) < 0.5).float()
        w3, w1 = _rnd(g, c, c, 3, 3) * 0.05, _rnd(g, c, 128, 1, 1) * 0.08
        a = _q(ref_prologue(_q(x, dtype), sc, sh, True, code, False), dtype)
        a1 = _q(_q(xl, dtype) * code1[:, :, None, None], dtype).repeat_interleave(2, 2).repeat_interleave(2, 3)
        ref = F.conv2d(a, _q(w3, dtype), b, padding=1) + F.conv2d(a1, _q(w1, dtype))
        segs = [ops.Seg(_nhwc(ops, x, dtype), scale=sc.cuda(), shift=sh.cuda(), code=code.cuda(), relu=True),
                ops.Seg(_nhwc(ops, xl, dtype), ksize=1, code=code1.cuda(), ups=True)]
        wimg = torch.cat([ops.prep_weight(w3.cuda(), dtype), ops.prep_weight(w1.cuda(), dtype)])
        (y, st), tiles = _conv_logged(ops, segs, wimg, c, bias=b.cuda(), stats_mode=1)
        s = st.double().sum(0).cpu()
        np.testing.assert_allclose(s[0, :c], ref.double().sum((0, 2, 3)), rtol=2e-2, atol=4e-3 * float(ref.abs().sum((0, 2, 3)).max()))
    elif case == 'pooled':
        x, img = _hot(_rnd(g, n, c, h, h)), _rnd(g, n, 3, h, h)
        w3, w1 = _rnd(g, c, c, 3, 3) * 0.05, _rnd(g, c, 3, 1, 1) * 0.3
        a = _q(torch.relu(_q(x, dtype)) * code[:, :, None, None], dtype)
        ref = F.avg_pool2d(F.conv2d(a, _q(w3, dtype), None, padding=1) + F.conv2d(_q(img, dtype), _q(w1, dtype)), 2) + b.view(1, -1, 1, 1)
        segs = [ops.Seg(_nhwc(ops, x, dtype), code=code.cuda(), relu=True), ops.Seg(_nhwc(ops, img, dtype), ksize=1)]
        wimg = torch.cat([ops.prep_weight(w3.cuda(), dtype), ops.prep_weight(w1.cuda(), dtype)])
        (y, _), tiles = _conv_logged(ops, segs, wimg, c, bias=b.cuda(), pool=True, alpha=0.25)
    elif case == 'gated':
        dy, xg = _rnd(g, n, c, h, h), _rnd(g, n, c, h, h)
        wt = _rnd(g, c, c, 3, 3) * 0.05
        ref = F.conv_transpose2d(_q(dy, dtype), _q(wt, dtype), padding=1) * code[:, :, None, None] * (_q(xg, dtype) > 0)
        (y, _), tiles = _conv_logged(ops, [ops.Seg(_nhwc(ops, dy, dtype))], ops.prep_weight(wt.cuda(), dtype, transpose=True), c,
                                     ocode=code.cuda(), gate_x=_nhwc(ops, xg, dtype))
    else:
        xl = _rnd(g, n, 128, h // 2, h // 2)
        sc, sh = _rnd(g, 5, 128) * 0.5 + 1, _rnd(g, 5, 128) * 0.3
        code1 = (torch.rand(n, 128, generator=g) < 0.5).float()
        w3 = _rnd(g, c, 128, 3, 3) * 0.05
        gi = torch.arange(n) // 16
        a = torch.relu(_q(xl, dtype) * sc[gi][:, :, None, None] + sh[gi][:, :, None, None]) * code1[:, :, None, None]
        ref = F.conv2d(_q(a, dtype).repeat_interleave(2, 2).repeat_interleave(2, 3), _q(w3, dtype), b, padding=1)
        seg = ops.Seg(_nhwc(ops, xl, dtype), scale=sc.cuda(), shift=sh.cuda(), code=code1.cuda(), ups=True, relu=True, group_n=16)
        (y, _), tiles = _conv_logged(ops, [seg], ops.prep_weight(w3.cuda(), dtype), c, bias=b.cuda(), stats_mode=1)
        y = y[0] if isinstance(y, tuple) else y
    assert tiles == [(128, 64)], tiles
    _assert_close(ops.to_nchw(y, c), ref, dtype, case)


@pytest.mark.parametrize('shape', [(64, 16, 8, 512, 3), (64, 8, 512, 16, 3), (128, 4, 8, 512, 3), (32, 16, 64, 48, 1)])
def test_same_shape_layers_share_one_launch(shape):
    """mcgen_wgrad_batch (ops.deferred_reduces queues the layers mcgen_wgrad_multi does not take and launches groups of identical
    shape together): MCGlow's skinny coupling-network gradients -- 8 -> 512 and 512 -> 16 at 16x16 / 8x8 / 4x4, 16 flows per
    level.  Five layers with their own operands in one pass: ONE batched launch, every gradient and bias gradient bit for bit
    what the layer gives alone, and within tolerance of the fp32 reference."""
    ops = _ops()
    dtype = torch.bfloat16
    n, h, ci, co, ks = shape
    g = torch.Generator().manual_seed(31 + h + ci)
    layers = []
    for _ in range(5):
        x, dy = _rnd(g, n, ci, h, h), _rnd(g, n, co, h, h) * 0.1
        code = (torch.rand(n, ci, generator=g) < 0.5).float()
        layers.append((x, dy, code))

    def run(batched):
        outs = []
        old = ops._BATCH
        ops._BATCH = batched
        ops.BATCH_LOG = []
        try:
            with ops.deferred_reduces():
                for x, dy, code in layers:
                    gw = torch.zeros(co, ci, ks, ks, device='cuda'); gb = torch.zeros(co, device='cuda')
                    ops.wgrad(ops.Seg(_nhwc(ops, x, dtype), ksize=ks, code=code.cuda(), relu=True), _nhwc(ops, dy, dtype), co, ci, gw, bias_grad=gb)
                    outs.append((gw, gb))
            torch.cuda.synchronize()
            return outs, list(ops.BATCH_LOG)
        finally:
            ops._BATCH = old
            ops.BATCH_LOG = None
    one, log1 = run(False)
    many, logn = run(True)
    assert log1 == [] and logn == [5], (log1, logn)
    for (gw1, gb1), (gwn, gbn), (x, dy, code) in zip(one, many, layers):
        assert torch.equal(gw1, gwn) and torch.equal(gb1, gbn)
        a = _q(torch.relu(_q(x, dtype)) * code[:, :, None, None], dtype)
        ref = torch.nn.grad.conv2d_weight(a, (co, ci, ks, ks), _q(dy, dtype), padding=ks // 2)
        _assert_close(gwn, ref, dtype, 'batched wgrad')
        np.testing.assert_allclose(gbn.cpu(), _q(dy, dtype).sum((0, 2, 3)), rtol=2e-2, atol=2e-2 * float(dy.abs().sum((0, 2, 3)).max()))


BIG_WG = [
    # N, H, Cin, Cout, ksize, ups(x), dy_ups: the weight-gradient launches of the bench (default split policy)
    (128, 32, 256, 256, 3, False, False),     # G block 2 conv_b
    (128, 32, 256, 256, 3, True, False),      # G block 2 conv_a (x through the upsample)
    (128, 32, 256, 256, 1, True, False),      # G block 2 shortcut
    (128, 32, 128, 128, 3, False, True),      # D block 0 conv2 (pooled gradient)
    (128, 16, 128, 128, 3, False, False),     # D block 1 conv1
    (128, 8, 128, 128, 3, False, False),      # D blocks 2, 3 (8x8 maps: producer/consumer form)
    (128, 32, 256, 3, 3, False, False),       # G head
]


@pytest.mark.parametrize('case', BIG_WG)
def test_big_wgrad_bf16(case):
    ops = _ops()
    dtype = torch.bfloat16
    n, h, ci, co, ks, ups, dy_ups = case
    g = torch.Generator().manual_seed(601 + h + ci + co + ks)
    hs = h // 2 if ups else h
    x = _rnd(g, n, ci, hs, hs)
    scale, shift = _rnd(g, ci) * 0.5 + 1, _rnd(g, ci) * 0.3
    code = (torch.rand(n, ci, generator=g) < 0.5).float()
    hd = h // 2 if dy_ups else h
    dy = _rnd(g, n, co, hd, hd) * 0.1
    a = _q(ref_prologue(_q(x, dtype), scale, shift, True, code, ups), dtype)
    dyf = _q(dy, dtype)
    if dy_ups:
        dyf = dyf.repeat_interleave(2, 2).repeat_interleave(2, 3)
    ref = torch.nn.grad.conv2d_weight(a, (co, ci, ks, ks), dyf, padding=ks // 2)
    grad = torch.zeros((co, ci, ks, ks), device='cuda')
    seg = ops.Seg(_nhwc(ops, x, dtype), ksize=ks, scale=scale.cuda(), shift=shift.cuda(), code=code.cuda(), ups=ups, relu=True)
    bg = torch.zeros((co,), device='cuda')
    ops.wgrad(seg, _nhwc(ops, dy, dtype), co, ci, grad, dy_ups=dy_ups, bias_grad=bg)
    _assert_close(grad, ref, dtype, 'big wgrad')
    _assert_close(bg, dyf.sum((0, 2, 3)), dtype, 'big wgrad bias')


@pytest.mark.parametrize('probe', PROBES)
@pytest.mark.parametrize('n,ks,dy_ups,halves', [(256, 3, False, True), (256, 1, True, True), (128, 3, False, False),
                                                (128, 1, True, False), (6, 3, False, True), (2, 1, False, False)])
def test_image_layer_wgrad_bf16(n, ks, dy_ups, halves, probe):
    """wgrad_c8.hip: the weight / bias gradients of FirstDisResBlock's image-side layers (mcgan.py:76-86: Conv3x3(3, 128) and the
    1x1 shortcut, whose gradient arrives 2x2-pooled) -- conv input = the raw image (pitch 8, no prologue), 128 outputs, 32x32 --
    per half of a paired batch, against torch.nn.grad.conv2d_weight on the CPU; run twice: bit-identical."""
    ops = _ops()
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(1801 + n + ks)
    x = _rnd(g, n, 3, 32, 32)
    hd = 16 if dy_ups else 32
    dy = _rnd(g, n, 128, hd, hd) * 0.1
    if probe:
        x, dy = _hot(x), _hot(dy, 8.0)
    xq, dyf = _q(x, dtype), _q(dy, dtype)
    if dy_ups:
        dyf = dyf.repeat_interleave(2, 2).repeat_interleave(2, 3)
    parts = (slice(0, n // 2), slice(n // 2, n)) if halves else (slice(None),)
    refs = [(torch.nn.grad.conv2d_weight(xq[sl], (128, 3, ks, ks), dyf[sl], padding=ks // 2), dyf[sl].sum((0, 2, 3))) for sl in parts]

    def run():
        gs = [torch.zeros((128, 3, ks, ks), device='cuda') for _ in parts]
        bs = [torch.zeros((128,), device='cuda') for _ in parts]
        ops.WGRAD_LOG = []
        try:
            ops.wgrad(ops.Seg(_nhwc(ops, x, dtype), ksize=ks), _nhwc(ops, dy, dtype), 128, 3, gs[0], dy_ups=dy_ups, bias_grad=bs[0],
                      second=(gs[1], bs[1], None) if halves else None)
            assert ops.WGRAD_LOG == ['c8'], ops.WGRAD_LOG
        finally:
            ops.WGRAD_LOG = None
        return gs, bs
    gs, bs = run()
    for hi, ((gr, br), gt, bt) in enumerate(zip(refs, gs, bs)):
        _assert_close(gt, gr, dtype, f'image-layer wgrad half {hi}')
        _assert_close(bt, br, dtype, f'image-layer bias grad half {hi}')
    gs2, bs2 = run()
    assert all(torch.equal(a, b) for a, b in zip(gs + bs, gs2 + bs2))


def test_big_wgrad_two_halves_bf16():
    """The paired discriminator pass: one launch over 2N images, one slab set (and one gradient) per half."""
    ops = _ops()
    dtype = torch.bfloat16
    n, h, c = 256, 16, 128
    g = torch.Generator().manual_seed(701)
    x = _rnd(g, n, c, h, h)
    code = (torch.rand(n, c, generator=g) < 0.5).float()
    dy = _rnd(g, n, c, h, h) * 0.1
    a = _q(ref_prologue(_q(x, dtype), None, None, True, code, False), dtype)
    dyf = _q(dy, dtype)
    refs = [torch.nn.grad.conv2d_weight(a[s], (c, c, 3, 3), dyf[s], padding=1) for s in (slice(0, n // 2), slice(n // 2, n))]
    g1, g2 = torch.zeros((c, c, 3, 3), device='cuda'), torch.zeros((c, c, 3, 3), device='cuda')
    b1, b2 = torch.zeros(c, device='cuda'), torch.zeros(c, device='cuda')
    seg = ops.Seg(_nhwc(ops, x, dtype), code=code.cuda(), relu=True)
    ops.wgrad(seg, _nhwc(ops, dy, dtype), c, c, g1, bias_grad=b1, second=(g2, b2, None))
    _assert_close(g1, refs[0], dtype, 'first half')
    _assert_close(g2, refs[1], dtype, 'second half')
    _assert_close(b1, dyf[:n // 2].sum((0, 2, 3)), dtype, 'first half bias')
    _assert_close(b2, dyf[n // 2:].sum((0, 2, 3)), dtype, 'second half bias')


@pytest.mark.parametrize('n,h,c,co,res,prol', [(128, 16, 512, 4, False, True), (128, 8, 512, 8, True, False), (128, 4, 512, 16, False, True),
                                               (16, 4, 256, 16, True, True), (8, 8, 160, 12, False, True)])
def test_skinny_split_k_conv_bf16(n, h, c, co, res, prol):
    """conv_skinny.hip: 3x3, Cout <= 16 over a deep K on 16x16 / 8x8 / 4x4 maps (MCGlow's ZeroConv2d forward, mcglow.py:119-130,
    and the coupling nets' input gradients): K split over the 16 waves of a workgroup, partial sums combined in LDS --
    with and without the prologue (ActNorm affine, ReLU, code) and the residual add, against F.conv2d on the CPU."""
    ops = _ops()
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(911 + h + co)
    x = _rnd(g, n, c, h, h)
    wt, b = _rnd(g, co, c, 3, 3) * 0.03, _rnd(g, co)
    scale, shift = (_rnd(g, c) * 0.5 + 1, _rnd(g, c) * 0.3) if prol else (None, None)
    code = (torch.rand(n, c, generator=g) < 0.5).float() if prol else None
    r = _rnd(g, n, co, h, h) if res else None
    a = _q(ref_prologue(_q(x, dtype), scale, shift, prol, code, False), dtype) if prol else _q(x, dtype)
    ref = F.conv2d(a, _q(wt, dtype), b, padding=1) * 1.0
    if res:
        ref = ref + _q(r, dtype)
    seg = ops.Seg(_nhwc(ops, x, dtype), scale=scale.cuda() if prol else None, shift=shift.cuda() if prol else None,
                  code=code.cuda() if prol else None, relu=prol)
    y, _ = ops.conv_fused([seg], ops.prep_weight(wt.cuda(), dtype), co, bias=b.cuda(), res=_nhwc(ops, r, dtype) if res else None)
    assert y.shape[-1] == ops.pad8(co)
    if ops.pad8(co) > co:
        assert float(y[..., co:].float().abs().max()) == 0.0          # padded channels stay exactly zero
    _assert_close(ops.to_nchw(y, co), ref, dtype, 'skinny conv')


@pytest.mark.parametrize('probe', PROBES)
@pytest.mark.parametrize('n,case', [(256, 'forward'), (128, 'forward'), (256, 'input gradient'), (128, 'input gradient'), (3, 'affine')])
def test_whole_image_conv_bf16(n, case, probe):
    """conv_smap.hip: 3x3, 128 -> 128 on 8x8 maps (the discriminator's 8x8 residual blocks, mcgan.py:95-138, forward and
    input-gradient direction): whole images per workgroup (two at N = 256), weight fragments straight from the image, K
    parts combined in LDS -- prologue (ReLU, code, affine) and epilogue (alpha, bias, bias2, output code, ReLU gate,
    residual) against F.conv2d on the CPU; the launch must take the new kernel (mcgen_conv_form == 2)."""
    ops = _ops()
    dtype = torch.bfloat16
    c = 128
    g = torch.Generator().manual_seed(1201 + n)
    x = _rnd(g, n, c, 8, 8)
    if probe:
        x = _hot(x)
    wt, b = _rnd(g, c, c, 3, 3) * 0.03, _rnd(g, c)
    kw, alpha = {}, 1.0
    if case == 'forward':
        code = (torch.rand(n, c, generator=g) < 0.5).float() * 1.25
        r = _rnd(g, n, c, 8, 8)
        a = _q(ref_prologue(_q(x, dtype), None, None, True, code, False), dtype)
        ref = F.conv2d(a, _q(wt, dtype), b, padding=1) + _q(r, dtype)
        seg = ops.Seg(_nhwc(ops, x, dtype), code=code.cuda(), relu=True)
        kw = dict(bias=b.cuda(), res=_nhwc(ops, r, dtype))
    elif case == 'input gradient':
        oc = (torch.rand(n, c, generator=g) < 0.5).float() * 0.8
        gx = _rnd(g, n, c, 8, 8)
        ref = F.conv2d(_q(x, dtype), _q(wt, dtype), None, padding=1) * oc[:, :, None, None] * (_q(gx, dtype) > 0).float()
        seg = ops.Seg(_nhwc(ops, x, dtype))
        kw = dict(ocode=oc.cuda(), gate_x=_nhwc(ops, gx, dtype))
    else:
        scale, shift = _rnd(g, c) * 0.5 + 1, _rnd(g, c) * 0.3
        b2 = _rnd(g, c)
        alpha = 0.5
        a = _q(ref_prologue(_q(x, dtype), scale, shift, False, None, False), dtype)
        ref = F.conv2d(a, _q(wt, dtype), None, padding=1) * alpha + (b + b2)[None, :, None, None]
        seg = ops.Seg(_nhwc(ops, x, dtype), scale=scale.cuda(), shift=shift.cuda())
        kw = dict(bias=b.cuda(), bias2=b2.cuda(), alpha=alpha)
    ops.KERNEL_LOG = []
    try:
        y, _ = ops.conv_fused([seg], ops.prep_weight(wt.cuda(), dtype), c, **kw)
        assert ops.KERNEL_LOG == [2], ops.KERNEL_LOG
    finally:
        ops.KERNEL_LOG = None
    _assert_close(ops.to_nchw(y, c), ref, dtype, f'whole-image conv: {case}')


# MCGatedPixelCNN's layers on its 8x8 code maps (mcpixelcnn.py:16-61, hidden 128): (segments [(Cin, ksize)], Cout, stats)
PIXEL_SHAPES = [([(128, 3)], 256, True),              # vertical stack
                ([(256, 1), (128, 3)], 256, True),     # vert_to_horiz (1x1) + horizontal stack, K-concatenated
                ([(256, 3)], 128, False),              # their input gradients
                ([(256, 1)], 256, False), ([(128, 1)], 128, True), ([(128, 1)], 128, False)]


@pytest.mark.parametrize('n', [128, 6])
@pytest.mark.parametrize('segs,co,stats', PIXEL_SHAPES)
def test_whole_image_conv_pixelcnn_shapes_bf16(segs, co, stats, n):
    """The same kernel on MCGatedPixelCNN's layer shapes: 128 / 256 channels, 1x1 and 3x3, a K-concatenated second segment,
    BatchNorm partial sums per image in the epilogue (one row of `stats` per image) -- against F.conv2d on the CPU."""
    ops = _ops()
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(1301 + n + co + 7 * len(segs))
    ref, sg, ws = 0, [], []
    for ci, ks in segs:
        x = _rnd(g, n, ci, 8, 8)
        wt = _rnd(g, co, ci, ks, ks) * (0.03 if ks == 3 else 0.08)
        code = (torch.rand(n, ci, generator=g) < 0.5).float()
        a = _q(ref_prologue(_q(x, dtype), None, None, True, code, False), dtype)
        ref = ref + F.conv2d(a, _q(wt, dtype), None, padding=ks // 2)
        sg.append(ops.Seg(_nhwc(ops, x, dtype), ksize=ks, code=code.cuda(), relu=True))
        ws.append(ops.prep_weight(wt.cuda(), dtype))
    b = _rnd(g, co)
    ref = ref + b[None, :, None, None]
    ops.KERNEL_LOG = []
    try:
        y, st = ops.conv_fused(sg, torch.cat(ws), co, bias=b.cuda(), stats_mode=1 if stats else 0)
        assert ops.KERNEL_LOG == [2], ops.KERNEL_LOG
    finally:
        ops.KERNEL_LOG = None
    _assert_close(ops.to_nchw(y, co), ref, dtype, 'whole-image conv, PixelCNN shape')
    if stats:
        assert st.shape == (n, 2, co)                  # one row per image
        yq = ops.to_nchw(y, co).cpu()
        np.testing.assert_allclose(st[:, 0].cpu(), yq.sum((2, 3)), rtol=2e-2, atol=0.5)
        np.testing.assert_allclose(st[:, 1].cpu(), (yq * yq).sum((2, 3)), rtol=2e-2, atol=0.5)


@pytest.mark.parametrize('probe', PROBES)
@pytest.mark.parametrize('n,c,gn', [(6, 256, 3), (5, 160, 0), (128, 256, 128)])
def test_image_head_conv_bf16(n, c, gn, probe):
    """conv_head.hip: the generator's image head (mcgan.py:55-60: BatchNorm -> ReLU -> MC -> Conv3x3(C, 3) -> Tanh on 32x32
    maps; C = 256, or a compacted 160) with double-buffered input windows; BatchNorm affine per statistics group of `gn`
    images -- against F.conv2d on the CPU."""
    ops = _ops()
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(1701 + n + c)
    groups = n // gn if gn else 1
    x = _rnd(g, n, c, 32, 32)
    if probe:
        x = _hot(x, 6.0)        # (tanh output: keep the pre-activation inside its sensitive range)
    scale, shift = _rnd(g, groups, c) * 0.5 + 1, _rnd(g, groups, c) * 0.3
    code = (torch.rand(n, c, generator=g) < 0.5).float()
    wt, b = _rnd(g, 3, c, 3, 3) * 0.05, _rnd(g, 3)
    idx = torch.arange(n) // gn if gn else torch.zeros(n, dtype=torch.long)
    a = _q(x, dtype) * scale[idx][:, :, None, None] + shift[idx][:, :, None, None]
    a = _q(torch.relu(a) * code[:, :, None, None], dtype)
    ref = torch.tanh(F.conv2d(a, _q(wt, dtype), b, padding=1))
    seg = ops.Seg(_nhwc(ops, x, dtype), scale=(scale if gn else scale[0]).cuda(), shift=(shift if gn else shift[0]).cuda(),
                  code=code.cuda(), relu=True, group_n=gn)
    ops.KERNEL_LOG = []
    try:
        y, _ = ops.conv_fused([seg], ops.prep_weight(wt.cuda(), dtype), 3, bias=b.cuda(), tanh=True)
        assert ops.KERNEL_LOG == [5], ops.KERNEL_LOG
    finally:
        ops.KERNEL_LOG = None
    assert y.shape[-1] == 8 and float(y[..., 3:].float().abs().max()) == 0.0
    _assert_close(ops.to_nchw(y, 3), ref, dtype, 'image head')


@pytest.mark.parametrize('probe', PROBES)
@pytest.mark.parametrize('n', [256, 128, 3])
def test_image_conv_bf16(n, probe):
    """conv_c8.hip: the discriminator's first convolution (FirstDisResBlock, mcgan.py:72-93: 3 -> 128 on the 32x32 image,
    channel pitch 8) with K = (tap, channel) and stores straight from the accumulators; per-sample code on the input (the
    paired pass's sigma ratio), bias -- against F.conv2d on the CPU."""
    ops = _ops()
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(1601 + n)
    x = _rnd(g, n, 3, 32, 32)
    if probe:
        x = _hot(x)
    wt, b = _rnd(g, 128, 3, 3, 3) * 0.2, _rnd(g, 128)
    code = torch.rand(n, 1, generator=g).expand(n, 8).contiguous() + 0.5        # one scalar per sample on every input channel
    a = _q(_q(x, dtype) * code[:, :3, None, None], dtype)
    ref = F.conv2d(a, _q(wt, dtype), b, padding=1)
    ops.KERNEL_LOG = []
    try:
        y, _ = ops.conv_fused([ops.Seg(_nhwc(ops, x, dtype), code=code.cuda())], ops.prep_weight(wt.cuda(), dtype), 128, bias=b.cuda())
        assert ops.KERNEL_LOG == [4], ops.KERNEL_LOG
    finally:
        ops.KERNEL_LOG = None
    _assert_close(ops.to_nchw(y, 128), ref, dtype, 'image convolution')


@pytest.mark.parametrize('n,gn', [(640, 128), (10, 2), (6, 3)])
@pytest.mark.parametrize('two_seg', [False, True])
def test_whole_image_conv_generator_first_block_bf16(n, gn, two_seg):
    """The same kernel on the generator's 8x8 block of the grouped pass (GenResBlock, mcgan.py:9-44): conv_a reads the 4x4
    input through the nearest x2 upsample with the BatchNorm affine of its statistics group (`gn` images per group), ReLU
    and code; conv_b ++ shortcut is a 3x3 segment on h plus a 1x1 segment on the upsampled block input; BatchNorm partial
    sums per image -- against F.conv2d on the CPU."""
    ops = _ops()
    dtype = torch.bfloat16
    c = 256
    g = torch.Generator().manual_seed(1501 + n + int(two_seg))
    groups = n // gn
    x = _rnd(g, n, c, 4, 4)
    code1 = (torch.rand(n, c, generator=g) < 0.5).float()
    b = _rnd(g, c)

    def grouped_affine(t, scale, shift):                  # row n // gn of [groups, C]
        idx = torch.arange(n) // gn
        return t * scale[idx][:, :, None, None] + shift[idx][:, :, None, None]
    if not two_seg:
        scale, shift = _rnd(g, groups, c) * 0.5 + 1, _rnd(g, groups, c) * 0.3
        wt = _rnd(g, c, c, 3, 3) * 0.02
        xu = _q(x, dtype).repeat_interleave(2, 2).repeat_interleave(2, 3)
        a = _q(torch.relu(grouped_affine(xu, scale, shift)) * code1[:, :, None, None], dtype)
        ref = F.conv2d(a, _q(wt, dtype), b, padding=1)
        segs = [ops.Seg(_nhwc(ops, x, dtype), scale=scale.cuda(), shift=shift.cuda(), code=code1.cuda(), ups=True, relu=True, group_n=gn)]
        wimg = ops.prep_weight(wt.cuda(), dtype)
    else:
        hmap = _rnd(g, n, c, 8, 8)
        scale, shift = _rnd(g, groups, c) * 0.5 + 1, _rnd(g, groups, c) * 0.3
        code2 = (torch.rand(n, c, generator=g) < 0.5).float()
        w2, ws = _rnd(g, c, c, 3, 3) * 0.02, _rnd(g, c, c, 1, 1) * 0.05
        a = _q(torch.relu(grouped_affine(_q(hmap, dtype), scale, shift)) * code2[:, :, None, None], dtype)
        xu = _q(_q(x, dtype).repeat_interleave(2, 2).repeat_interleave(2, 3) * code1[:, :, None, None], dtype)
        ref = F.conv2d(a, _q(w2, dtype), b, padding=1) + F.conv2d(xu, _q(ws, dtype), None)
        segs = [ops.Seg(_nhwc(ops, hmap, dtype), scale=scale.cuda(), shift=shift.cuda(), code=code2.cuda(), relu=True, group_n=gn),
                ops.Seg(_nhwc(ops, x, dtype), ksize=1, code=code1.cuda(), ups=True)]
        wimg = torch.cat([ops.prep_weight(w2.cuda(), dtype), ops.prep_weight(ws.cuda(), dtype)])
    ops.KERNEL_LOG = []
    try:
        y, st = ops.conv_fused(segs, wimg, c, bias=b.cuda(), stats_mode=1)
        assert ops.KERNEL_LOG == [2], ops.KERNEL_LOG
    finally:
        ops.KERNEL_LOG = None
    _assert_close(ops.to_nchw(y, c), ref, dtype, 'whole-image conv, generator first block')
    assert st.shape == (n, 2, c)
    yq = ops.to_nchw(y, c).cpu()
    np.testing.assert_allclose(st[:, 0].cpu(), yq.sum((2, 3)), rtol=2e-2, atol=0.5)
    np.testing.assert_allclose(st[:, 1].cpu(), (yq * yq).sum((2, 3)), rtol=2e-2, atol=0.5)


@pytest.mark.parametrize('n,h', [(8, 16), (128, 16), (16, 8), (32, 4)])
@pytest.mark.parametrize('case', ['forward', 'input gradient'])
def test_resident_tile_1x1_conv_bf16(n, h, case):
    """conv_px1.hip: the 512 -> 512 1x1 convolution of MCGlow's coupling networks (mcglow.py:133-160) on 16x16 / 8x8 / 4x4
    maps with the pixel tile resident in LDS.  Forward: ActNorm affine + ReLU + code prologue, bias.  Input gradient: output
    code, ReLU gate through the ActNorm affine of the gated tensor, residual, and the ActNorm-gradient partial sums
    (stats_mode 2: sum v and sum v * (x - mean) * rstd per tile) -- against F.conv2d on the CPU."""
    ops = _ops()
    dtype = torch.bfloat16
    c = 512
    g = torch.Generator().manual_seed(1401 + n + h)
    x = _rnd(g, n, c, h, h)
    wt = _rnd(g, c, c, 1, 1) * 0.04
    ops.KERNEL_LOG = []
    try:
        if case == 'forward':
            scale, shift = _rnd(g, c) * 0.5 + 1, _rnd(g, c) * 0.3
            code = (torch.rand(n, c, generator=g) < 0.5).float()
            b = _rnd(g, c)
            a = _q(ref_prologue(_q(x, dtype), scale, shift, True, code, False), dtype)
            ref = F.conv2d(a, _q(wt, dtype), b)
            seg = ops.Seg(_nhwc(ops, x, dtype), ksize=1, scale=scale.cuda(), shift=shift.cuda(), code=code.cuda(), relu=True)
            y, _ = ops.conv_fused([seg], ops.prep_weight(wt.cuda(), dtype), c, bias=b.cuda())
        else:
            oc = (torch.rand(n, c, generator=g) < 0.5).float()
            gx, r = _rnd(g, n, c, h, h), _rnd(g, n, c, h, h)
            gsc, gsh, gme = _rnd(g, c) * 0.5 + 1, _rnd(g, c) * 0.3, _rnd(g, c) * 0.2
            grs = torch.rand(c, generator=g) + 0.5
            gxq = _q(gx, dtype)
            gate = ((gxq * gsc[None, :, None, None] + gsh[None, :, None, None]) > 0).float()
            v = F.conv2d(_q(x, dtype), _q(wt, dtype), None) * oc[:, :, None, None] * gate
            ref = v + _q(r, dtype)
            y, st = ops.conv_fused([ops.Seg(_nhwc(ops, x, dtype), ksize=1)], ops.prep_weight(wt.cuda(), dtype), c, ocode=oc.cuda(),
                                   gate_x=_nhwc(ops, gx, dtype), gscale=gsc.cuda(), gshift=gsh.cuda(), gmean=gme.cuda(), grstd=grs.cuda(),
                                   res=_nhwc(ops, r, dtype), stats_mode=2)
            s = st.sum(0).cpu()
            xhat = (gxq - gme[None, :, None, None]) * grs[None, :, None, None]
            s1, s2 = v.sum((0, 2, 3)), (v * xhat).sum((0, 2, 3))
            tol = max(1.0, float(s2.abs().max())) * 1e-2
            np.testing.assert_allclose(s[0], s1, rtol=2e-2, atol=tol)
            np.testing.assert_allclose(s[1], s2, rtol=2e-2, atol=tol)
        assert ops.KERNEL_LOG == [3], ops.KERNEL_LOG
    finally:
        ops.KERNEL_LOG = None
    _assert_close(ops.to_nchw(y, c), ref, dtype, f'resident-tile 1x1: {case}')


MULTI_PASSES = {
    # one backward pass = the layers whose weight gradients share ONE mcgen_wgrad_multi launch:
    # (N, H, Cin, Cout, ups(x), dy_ups, affine, two halves[, ksize])
    'generator': [(128, 32, 256, 256, False, False, True, False),     # G block 2 conv_b
                  (128, 32, 256, 256, True, False, True, False),      # G block 2 conv_a (x through the upsample)
                  (128, 16, 256, 256, False, False, True, False),
                  (128, 16, 256, 256, True, False, True, False),
                  (128, 8, 256, 256, False, False, True, False),      # 8x8 maps: two images per 128-pixel step
                  (128, 8, 256, 256, True, False, True, False)],      # x at 4x4 through the upsample
    'discriminator': [(256, 32, 128, 128, False, True, False, True),  # D block 0 conv2: pooled gradient, one slab set per half
                      (256, 16, 128, 128, False, False, False, True),
                      (256, 16, 128, 128, False, True, False, True),
                      (256, 8, 128, 128, False, False, False, True),
                      (256, 8, 128, 128, False, False, False, True)],
    'small': [(4, 16, 64, 128, False, False, True, False),           # few steps per layer: more workgroups than steps must not happen
              (2, 8, 128, 128, True, False, False, False),
              (8, 32, 64, 128, False, True, True, True)],
    # 1x1 layers ride in the same launch: the generator's low-resolution shortcut, the discriminator's pooled shortcut,
    # MCGlow's 512 -> 512 coupling convolution on 16x16 / 8x8 / 4x4 maps (eight images per 128-pixel step)
    'with 1x1': [(128, 16, 256, 256, False, False, False, False, 1),
                 (256, 16, 128, 128, False, True, False, True, 1),
                 (128, 16, 512, 512, False, False, True, False, 1),
                 (128, 8, 512, 512, False, False, True, False, 1),
                 (128, 4, 512, 512, False, False, True, False, 1),
                 (128, 16, 256, 256, False, False, True, False)],
}


@pytest.mark.parametrize('probe', PROBES)
@pytest.mark.parametrize('name', list(MULTI_PASSES))
def test_wgrad_multi_pass_bf16(name, probe):
    """mcgen_wgrad_multi (wgrad_multi.hip): the 3x3 weight gradients of one backward pass queued inside a deferred_reduces
    context run as ONE launch with FLOP-proportional pixel splits; every layer against the CPU reference
    (torch.nn.grad.conv2d_weight on the prologue-applied, bf16-rounded operands), bias gradients and the per-half
    gradients of a paired pass included -- and bit-identical when the same pass is run twice (fixed-order reduction)."""
    ops = _ops()
    dtype = torch.bfloat16
    layers = MULTI_PASSES[name]
    g = torch.Generator().manual_seed(811 + len(layers))
    prob, refs = [], []
    for layer in layers:
        n, h, ci, co, ups, dy_ups, affine, halves = layer[:8]
        ks = layer[8] if len(layer) > 8 else 3
        hs = h // 2 if ups else h
        x = _rnd(g, n, ci, hs, hs)
        scale, shift = (_rnd(g, ci) * 0.5 + 1, _rnd(g, ci) * 0.3) if affine else (None, None)
        code = (torch.rand(n, ci, generator=g) < 0.5).float()
        hd = h // 2 if dy_ups else h
        dy = _rnd(g, n, co, hd, hd) * 0.1
        if probe:
            x, dy = _hot(x), _hot(dy, 8.0)
        a = _q(ref_prologue(_q(x, dtype), scale, shift, True, code, ups), dtype)
        dyf = _q(dy, dtype)
        if dy_ups:
            dyf = dyf.repeat_interleave(2, 2).repeat_interleave(2, 3)
        parts = (slice(0, n // 2), slice(n // 2, n)) if halves else (slice(None),)
        refs.append([(torch.nn.grad.conv2d_weight(a[sl], (co, ci, ks, ks), dyf[sl], padding=ks // 2), dyf[sl].sum((0, 2, 3))) for sl in parts])
        seg = ops.Seg(_nhwc(ops, x, dtype), ksize=ks, scale=scale.cuda() if affine else None, shift=shift.cuda() if affine else None,
                      code=code.cuda(), ups=ups, relu=True)
        prob.append((seg, _nhwc(ops, dy, dtype), co, ci, dy_ups, halves, ks))

    def run():
        outs = []
        ops._PROF = []                                   # record the launches of the pass
        try:
            with ops.deferred_reduces():
                for seg, dyt, co, ci, dy_ups, halves, ks in prob:
                    gs = [torch.zeros((co, ci, ks, ks), device='cuda') for _ in range(2 if halves else 1)]
                    bs = [torch.zeros((co,), device='cuda') for _ in range(2 if halves else 1)]
                    ops.wgrad(seg, dyt, co, ci, gs[0], dy_ups=dy_ups, bias_grad=bs[0], second=(gs[1], bs[1], None) if halves else None)
                    outs.append((gs, bs))
            names = [r[0] for r in ops._PROF]
        finally:
            ops._PROF = None
        return outs, names
    outs, names = run()
    assert names.count('wgrad_multi<bf16>') == 1 and not any(nm.startswith('wgrad<') for nm in names), names
    for li, ((gs, bs), ref) in enumerate(zip(outs, refs)):
        for hi, ((gr, br), gt, bt) in enumerate(zip(ref, gs, bs)):
            _assert_close(gt, gr, dtype, f'{name} layer {li} half {hi}')
            _assert_close(bt, br, dtype, f'{name} layer {li} half {hi} bias')
    outs2, _ = run()
    for (gs, bs), (gs2, bs2) in zip(outs, outs2):
        for a_, b_ in zip(gs + bs, gs2 + bs2):
            assert torch.equal(a_, b_)


def test_wgrad_multi_matches_the_per_layer_kernels():
    """Same operands through mcgen_wgrad_multi and through the per-layer kernels of wgrad.hip (the queue switched off):
    the two split the pixels differently, so they agree to fp32 summation order, not bitwise."""
    ops = _ops()
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(877)
    n, h, c = 64, 16, 128
    x, dy = _rnd(g, n, c, h, h), _rnd(g, n, c, h, h) * 0.1
    scale, shift = _rnd(g, c) * 0.5 + 1, _rnd(g, c) * 0.3
    code = (torch.rand(n, c, generator=g) < 0.5).float()
    seg = ops.Seg(_nhwc(ops, x, dtype), scale=scale.cuda(), shift=shift.cuda(), code=code.cuda(), relu=True)
    dyt = _nhwc(ops, dy, dtype)
    res = []
    for multi in (True, False):
        old = ops._MULTI
        ops._MULTI = multi
        try:
            gr, bg = torch.zeros((c, c, 3, 3), device='cuda'), torch.zeros((c,), device='cuda')
            with ops.deferred_reduces():
                ops.wgrad(seg, dyt, c, c, gr, bias_grad=bg, alpha=0.5)
            res.append((gr.cpu(), bg.cpu()))
        finally:
            ops._MULTI = old
    scale_g = float(res[1][0].abs().max())
    assert float((res[0][0] - res[1][0]).abs().max()) <= 2e-5 * scale_g + 1e-6
    assert float((res[0][1] - res[1][1]).abs().max()) <= 2e-5 * float(res[1][1].abs().max()) + 1e-6


@pytest.mark.parametrize('dtype', DTYPES)
def test_grouped_batchnorm_pass_equals_separate_passes(dtype):
    """mcgen_seg_t.group_n + mcgen_bn_finalize_groups: two training-mode BatchNorm batches pushed through
    conv -> BN -> ReLU -> MC -> conv as ONE pass (statistics per group, running statistics updated group by group) give
    bit-for-bit what two separate passes give -- the tiles, and so every partial sum, are the same."""
    ops = _ops()
    g = torch.Generator().manual_seed(811)
    n, gn, c, hw = 8, 4, 32, 8
    x = _rnd(g, n, c, hw, hw)
    w1, w2, b1 = _rnd(g, c, c, 3, 3) * 0.1, _rnd(g, c, c, 3, 3) * 0.1, _rnd(g, c)
    gamma, beta = _rnd(g, c) * 0.2 + 1, _rnd(g, c) * 0.1
    code = (torch.rand(n, c, generator=g) < 0.5).float().cuda()
    i1, i2 = ops.prep_weight(w1.cuda(), dtype), ops.prep_weight(w2.cuda(), dtype)
    xt = _nhwc(ops, x, dtype)

    def run(xs, cd, rm, rv, groups):
        nn_ = xs.shape[0]
        h, st = ops.conv_fused([ops.Seg(xs)], i1, c, bias=b1.cuda(), stats_mode=1)
        sc, sh, mean, rstd = ops.bn_finalize(st, (nn_ // groups) * hw * hw, gamma.cuda(), beta.cuda(), rm, rv, groups=groups)
        seg = ops.Seg(h, scale=sc, shift=sh, code=cd, relu=True, group_n=(nn_ // groups if groups > 1 else 0))
        y, _ = ops.conv_fused([seg], i2, c)
        return y, sc, sh, mean, rstd

    rm_a, rv_a = torch.zeros(c, device='cuda'), torch.ones(c, device='cuda')
    y_a, sc_a, sh_a, mean_a, rstd_a = run(xt, code, rm_a, rv_a, 2)
    assert sc_a.shape == (2, c)
    rm_b, rv_b = torch.zeros(c, device='cuda'), torch.ones(c, device='cuda')
    parts = [run(xt[i * gn:(i + 1) * gn].contiguous(), code[i * gn:(i + 1) * gn].contiguous(), rm_b, rv_b, 1) for i in range(2)]
    assert torch.equal(y_a, torch.cat([p[0] for p in parts]))
    for j, t in enumerate((sc_a, sh_a, mean_a, rstd_a), start=1):
        assert torch.equal(t, torch.stack([p[j] for p in parts])), j
    assert torch.equal(rm_a, rm_b) and torch.equal(rv_a, rv_b)
    # and against the definition: per-group batch statistics of the first convolution's output
    h_ref = F.conv2d(_q(x, dtype), _q(w1, dtype), b1, padding=1).view(2, gn, c, hw, hw)
    np.testing.assert_allclose(mean_a.cpu(), h_ref.mean((1, 3, 4)), rtol=2e-3 if dtype == torch.float32 else 3e-2, atol=2e-3 if dtype == torch.float32 else 3e-2)
    # a tile that would straddle two groups is refused before launch
    from mcgen_amd import _lib
    z = ops.to_nhwc(_rnd(g, 8, 16, 1, 1).cuda(), dtype)                      # 1x1 maps: 64 images per tile
    sc1 = torch.ones(2, 16, device='cuda')
    with pytest.raises(_lib.McgenError):
        ops.conv_fused([ops.Seg(z, ksize=1, scale=sc1, shift=sc1, group_n=4)], ops.prep_weight((_rnd(g, 32, 16, 1, 1)).cuda(), dtype), 32)


# --------------------------------------------------------------------------------------------------------- #
# Mode-compacted convolutions (conv_fused.hip "mc" form): the K loop visits only the channels whose
# MultimodalController code is non-zero (modules.py:58-76: controller_rate 0.5 zeroes half of them per sample).
def test_cmap_and_kmajor_image_layout():
    ops = _ops()
    g = torch.Generator().manual_seed(901)
    n, c = 7, 72
    code = (torch.rand(n, c, generator=g) < 0.5).float() * (1 + torch.rand(n, c, generator=g))
    code[2] = 0.0                                            # a sample with no active channel
    code[3] = 1.0                                            # and one with all of them
    cm = ops.mc_cmap(code.cuda()).cpu().numpy()
    stride = ops.cmap_stride(c)
    assert cm.shape == (n, stride) and stride % 8 == 0
    nd = (c + 31) // 32
    for i in range(n):
        act = np.nonzero(code[i].numpy())[0]
        cpos = np.full(c, -1); cpos[act] = np.arange(len(act))
        assert np.array_equal(cm[i, :c], cpos)
        cidx = cm[i, c:2 * c + 32]
        assert np.array_equal(cidx[:len(act)], act) and np.all(cidx[len(act):] == c)
        cpre = cm[i, 2 * c + 32:2 * c + 32 + 2 * (nd + 1)].copy().view(np.int32)
        assert np.array_equal(cpre, [int((act < 32 * d).sum()) for d in range(nd)] + [len(act)])
    w = _rnd(g, 20, 11, 3, 3)
    sig = torch.tensor([1.7])
    img = ops.prep_weight_k(w.cuda(), torch.float32, sigma=sig.cuda(), wscale=0.5).cpu().view(9, 17, 32)
    ref = torch.zeros(9, 17, 32)
    ref[:, :11, :20] = (w * 0.5 / 1.7).permute(2, 3, 1, 0).reshape(9, 11, 20)
    np.testing.assert_allclose(img.numpy(), ref.numpy(), rtol=1e-6, atol=1e-7)
    assert float(img[:, 16].abs().max()) == 0.0             # the zero row padded K slots point at


def _mc_segments(ops, xs, scales, shifts, codes, upss, relus, ksizes, dtype):
    segs_d, segs_c = [], []
    for x, sc, sh, cd, up, rl, ks in zip(xs, scales, shifts, codes, upss, relus, ksizes):
        xt = _nhwc(ops, x, dtype)
        kw = dict(ksize=ks, scale=None if sc is None else sc.cuda(), shift=None if sh is None else sh.cuda(), code=cd.cuda(), ups=up, relu=rl)
        segs_d.append(ops.Seg(xt, **kw))
        segs_c.append(ops.Seg(xt, cmap=ops.mc_cmap(cd.cuda()), **kw))
    return segs_d, segs_c


def _assert_bf16_twin(a, b, what):
    """Two bf16 results of the same fp32 sums taken in a different order: equal up to one bf16 ulp on a few elements."""
    a, b = a.float().cpu(), b.float().cpu()
    err = (a - b).abs()
    tol = 2.0 ** -7 * b.abs() + 2e-3 * float(b.abs().max())
    assert bool((err <= tol).all()), f'{what}: max err {float(err.max()):.3e}'
    assert float((err > 0).float().mean()) < 0.2, f'{what}: {float((err > 0).float().mean()):.3f} of the elements differ'


MC_CASES = [
    # N, H(out), Cin, Cout, ups, tile
    (128, 32, 256, 256, True, (256, 256)),      # G block 2 conv_a
    (128, 16, 256, 256, True, (128, 256)),      # G block 1 conv_a in the generator update (N = 128)
    (640, 16, 256, 256, True, (256, 256)),      # ... in the grouped pass of the discriminator updates (5 N images)
    (256, 32, 128, 128, False, (128, 128)),     # D's 128-channel layers, paired pass
    (32, 32, 200, 72, False, (128, 128)),       # ragged: 200 channels (6.25 chunks), 72 outputs (partial N tile)
]


@pytest.mark.parametrize('case', MC_CASES)
def test_mode_compacted_conv_matches_dense_and_reference(case):
    ops = _ops()
    dtype = torch.bfloat16
    n, h, ci, co, ups, tile = case
    g = torch.Generator().manual_seed(911 + n + h + ci)
    hs = h // 2 if ups else h
    x = _rnd(g, n, ci, hs, hs)
    scale, shift = _rnd(g, ci) * 0.5 + 1, _rnd(g, ci) * 0.3
    code = (torch.rand(n, ci, generator=g) < 0.5).float()
    code[0] = 0.0; code[1] = 1.0                            # no active channel / all active
    code[n // 2:] *= 1.25                                   # the paired pass scales the fake half's codes
    code[2, : ci // 2] = 0.0; code[2, ci // 2:] = 1.0       # a block pattern: whole dense chunks without a slot
    wt, b = _rnd(g, co, ci, 3, 3) * 0.03, _rnd(g, co)
    segs_d, segs_c = _mc_segments(ops, [x], [scale], [shift], [code], [ups], [True], [3], dtype)
    (yd, std), _ = _conv_logged(ops, segs_d, ops.prep_weight(wt.cuda(), dtype), co, bias=b.cuda(), stats_mode=1)
    (yc, stc), tiles = _conv_logged(ops, segs_c, ops.prep_weight_k(wt.cuda(), dtype), co, bias=b.cuda(), stats_mode=1, kmajor=True)
    assert tiles == [tile], tiles
    _assert_bf16_twin(yc, yd, 'compacted vs dense')
    np.testing.assert_allclose(stc.double().sum(0).cpu(), std.double().sum(0).cpu(), rtol=2e-4, atol=1e-2)
    if n <= 256:
        a = _q(ref_prologue(_q(x, dtype), scale, shift, True, code, ups), dtype)
        ref = F.conv2d(a, _q(wt, dtype), b, padding=1)
        _assert_close(ops.to_nchw(yc, co), ref, dtype, 'compacted vs fp32 reference')


@pytest.mark.parametrize('n,h,t', [(128, 32, (256, 256)), (128, 16, (128, 256))])
def test_mode_compacted_conv_b_two_segments(n, h, t):
    """G.conv_b: conv3x3(BN/ReLU/MC2(h)) + conv1x1(MC1(Up(x))): two segments, each with its own compaction map."""
    ops = _ops()
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(931 + h)
    c, hs = 256, h // 2
    h_in, x = _rnd(g, n, c, h, h), _rnd(g, n, c, hs, hs)
    scale, shift = _rnd(g, c) * 0.5 + 1, _rnd(g, c) * 0.3
    code1 = (torch.rand(n, c, generator=g) < 0.5).float()
    code2 = (torch.rand(n, c, generator=g) < 0.5).float()
    code1[3] = 0.0
    w2, ws, b = _rnd(g, c, c, 3, 3) * 0.03, _rnd(g, c, c, 1, 1) * 0.08, _rnd(g, c)
    segs_d, segs_c = _mc_segments(ops, [h_in, x], [scale, None], [shift, None], [code2, code1], [False, True], [True, False], [3, 1], dtype)
    img_d = torch.cat([ops.prep_weight(w2.cuda(), dtype), ops.prep_weight(ws.cuda(), dtype)])
    img_c = torch.cat([ops.prep_weight_k(w2.cuda(), dtype), ops.prep_weight_k(ws.cuda(), dtype)])
    yd, _ = ops.conv_fused(segs_d, img_d, c, bias=b.cuda())
    (yc, _), tiles = _conv_logged(ops, segs_c, img_c, c, bias=b.cuda(), kmajor=True)
    assert tiles == [t], tiles
    _assert_bf16_twin(yc, yd, 'compacted conv_b vs dense')
    a2 = _q(ref_prologue(_q(h_in, dtype), scale, shift, True, code2, False), dtype)
    a1 = _q(ref_prologue(_q(x, dtype), None, None, False, code1, True), dtype)
    ref = F.conv2d(a2, _q(w2, dtype), b, padding=1) + F.conv2d(a1, _q(ws, dtype))
    _assert_close(ops.to_nchw(yc, c), ref, dtype, 'compacted conv_b vs fp32 reference')


@pytest.mark.parametrize('ordered', [False, True])
@pytest.mark.parametrize('two_seg', [False, True])
def test_per_mode_weight_sets_match_the_masked_convolution(two_seg, ordered):
    """mcgen_conv_t.wsel / order + mcgen_prep_t.kmap: activations stored compacted per image (the channels the image's mode
    keeps, in order, zero-padded to the pitch), ONE dense weight image per mode whose input channels are that mode's active
    ones -- the software-pipelined form with a dense K loop over the pitch -- against F.conv2d on the masked dense tensors
    (modules.py:71-76: x * code in front of the convolution), with and without the mode-sorted walk, one 3x3 segment and
    3x3 ++ 1x1(Up(x)) (GenResBlock's conv_b ++ shortcut, mcgan.py:20-30)."""
    ops = _ops()
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(2101 + int(two_seg))
    n, h, c, co, modes = 64, 32, 256, 256, 5
    cb = (torch.rand(modes, c, generator=g) < 0.5).float()
    cb[0, c - 1] = 1.0; cb[1, 0] = 1.0
    cap = (int(cb.sum(1).max()) + 31) // 32 * 32
    label = torch.randint(0, modes, (n,), generator=g)
    cmaps = ops.mc_cmap(cb.cuda())                                   # records of the codebook rows = of the modes

    def compact(x):                                                  # [N, C, H, W] -> NHWC [N, H, W, cap]: active channels first
        out = torch.zeros(x.shape[0], x.shape[2], x.shape[3], cap)
        for i in range(x.shape[0]):
            idx = torch.nonzero(cb[label[i]]).flatten()
            out[i, :, :, :idx.numel()] = x[i, idx].permute(1, 2, 0)
        return out.to(dtype).cuda().contiguous()

    def mode_sets(w, ks):
        per = ops.weight_image_elems(w.shape[0], cap, ks)
        return per, [(w.cuda(), None, False, 1, -1, 1.0, False, cmaps[m, c:], cap) for m in range(modes)]
    x = _hot(_rnd(g, n, c, h, h))
    w3, b = _rnd(g, co, c, 3, 3) * 0.03, _rnd(g, co)
    code = cb[label]
    ref = F.conv2d(_q(_q(x, dtype) * code[:, :, None, None], dtype), _q(w3, dtype), b, padding=1)
    segs = [ops.Seg(compact(x))]
    per3, jobs3 = mode_sets(w3, 3)
    per = per3
    if two_seg:
        xl = _rnd(g, n, c, h // 2, h // 2)
        w1 = _rnd(g, co, c, 1, 1) * 0.08
        ref = ref + F.conv2d(_q(_q(xl, dtype) * code[:, :, None, None], dtype).repeat_interleave(2, 2).repeat_interleave(2, 3), _q(w1, dtype))
        segs.append(ops.Seg(compact(xl), ksize=1, ups=True))
        per1, jobs1 = mode_sets(w1, 1)
        per = per3 + per1
    sets = torch.empty(modes, per, dtype=dtype, device='cuda')
    jobs = []
    for m in range(modes):
        j = list(jobs3[m]); j[1] = sets[m, :per3]; jobs.append(tuple(j))
        if two_seg:
            j = list(jobs1[m]); j[1] = sets[m, per3:]; jobs.append(tuple(j))
    ops.PrepBatch(jobs, dtype).run()
    lab32 = label.to(torch.int32).cuda()
    if ordered:
        srt, oi = torch.sort(lab32, stable=True)
        wsel, order = srt.contiguous(), oi.to(torch.int32)
    else:
        wsel, order = lab32, None
    (y, st), tiles = _conv_logged(ops, segs, sets.view(-1), co, bias=b.cuda(), stats_mode=1, wsel=wsel, order=order)
    assert tiles == [(256, 256)], tiles
    got = ops.to_nchw(y, co)
    _assert_close(got, ref, dtype, 'per-mode weight sets')
    # the statistics rows keep their true tile index whatever the walk
    s = st.double().sum(0).cpu()
    np.testing.assert_allclose(s[0, :co], ref.double().sum((0, 2, 3)), rtol=2e-2, atol=4e-3 * float(ref.abs().sum((0, 2, 3)).max()))
    tile_sum = st[:, 0, :co].double().cpu().view(n, 4, co).sum(1)                # four 256-pixel tiles per 32x32 image
    np.testing.assert_allclose(tile_sum, got.double().sum((2, 3)).cpu(), rtol=2e-2, atol=0.5)
    # what the form refuses: fp32, small launches (no software-pipelined tile)
    with pytest.raises(Exception):
        ops.conv_fused([ops.Seg(compact(x)[:8])], sets.view(-1), co, bias=b.cuda(), wsel=lab32[:8].contiguous())


@pytest.mark.parametrize('two_seg', [False, True])
def test_permuted_weight_rows_store_the_compacted_output(two_seg):
    """mcgen_conv_t.yperm + mcgen_prep_t.rmap: the weight rows of a mode's image in the order the CONSUMER's mask keeps them
    -- the tile's columns are the compacted order, stored with plain 16-byte stores -- against the gather pass of `ycmap`
    on the same inputs: the active slots of every image bit for bit, the statistics rows (true channel order) bit for bit,
    and against F.conv2d on the masked tensors."""
    ops = _ops()
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(2203 + int(two_seg))
    n, h, c, co, modes = 64, 32, 256, 256, 4
    cb = (torch.rand(modes, c, generator=g) < 0.5).float()          # the mask in front of this convolution
    cb2 = (torch.rand(modes, co, generator=g) < 0.55).float()       # its consumer's mask
    cb2[0, co - 1] = 1.0; cb2[1, 0] = 1.0
    cap = (int(cb.sum(1).max()) + 31) // 32 * 32
    cap2 = (int(cb2.sum(1).max()) + 31) // 32 * 32
    label = torch.randint(0, modes, (n,), generator=g)
    cmaps = ops.mc_cmap(cb.cuda())
    perm2 = torch.argsort((cb2 == 0).to(torch.int8), dim=1, stable=True).to(torch.int16).cuda().contiguous()

    def compact(x):
        out = torch.zeros(x.shape[0], x.shape[2], x.shape[3], cap)
        for i in range(x.shape[0]):
            idx = torch.nonzero(cb[label[i]]).flatten()
            out[i, :, :, :idx.numel()] = x[i, idx].permute(1, 2, 0)
        return out.to(dtype).cuda().contiguous()
    x = _hot(_rnd(g, n, c, h, h))
    w3, b = _rnd(g, co, c, 3, 3) * 0.03, _rnd(g, co)
    code = cb[label]
    ref = F.conv2d(_q(_q(x, dtype) * code[:, :, None, None], dtype), _q(w3, dtype), b, padding=1)
    segs = [ops.Seg(compact(x))]
    per3 = ops.weight_image_elems(co, cap, 3)
    per = per3
    if two_seg:
        xl = _rnd(g, n, c, h // 2, h // 2)
        w1 = _rnd(g, co, c, 1, 1) * 0.08
        ref = ref + F.conv2d(_q(_q(xl, dtype) * code[:, :, None, None], dtype).repeat_interleave(2, 2).repeat_interleave(2, 3), _q(w1, dtype))
        segs.append(ops.Seg(compact(xl), ksize=1, ups=True))
        per += ops.weight_image_elems(co, cap, 1)

    def build(rows):
        sets = torch.empty(modes, per, dtype=dtype, device='cuda')
        jobs = []
        for m in range(modes):
            r = perm2[m] if rows else None
            jobs.append((w3.cuda(), sets[m, :per3], False, 1, -1, 1.0, False, cmaps[m, c:], cap, r))
            if two_seg:
                jobs.append((w1.cuda(), sets[m, per3:], False, 1, -1, 1.0, False, cmaps[m, c:], cap, r))
        ops.PrepBatch(jobs, dtype).run()
        return sets
    lab32 = label.to(torch.int32).cuda()
    ycm = ops.mc_cmap(cb2[label].cuda())
    ya, sta = ops.conv_fused(segs, build(False).view(-1), co, bias=b.cuda(), stats_mode=1, wsel=lab32, ycmap=ycm, cy=cap2)
    (yb, stb), tiles = _conv_logged(ops, segs, build(True).view(-1), co, bias=b.cuda(), stats_mode=1, wsel=lab32, yperm=perm2, cy=cap2)
    assert tiles == [(256, 256)] and tuple(yb.shape) == (n, h, h, cap2)
    assert torch.equal(sta, stb)
    for i in range(n):
        idx = torch.nonzero(cb2[label[i]]).flatten()
        k = idx.numel()
        assert torch.equal(ya[i, :, :, :k], yb[i, :, :, :k]), i
        _assert_close(yb[i, :, :, :k].float().permute(2, 0, 1).cpu(), ref[i, idx], dtype, f'image {i}')
        assert bool(torch.isfinite(yb[i].float()).all())
    # refused: without wsel, with order, a pitch beyond the channels
    with pytest.raises(Exception):
        ops.conv_fused(segs, build(True)[0].contiguous(), co, bias=b.cuda(), yperm=perm2[:1].contiguous(), cy=cap2)
    with pytest.raises(Exception):
        ops.conv_fused(segs, build(True).view(-1), co, bias=b.cuda(), wsel=lab32, order=torch.arange(n, dtype=torch.int32, device='cuda'), yperm=perm2, cy=cap2)


def test_mode_compacted_conv_rejects_what_it_cannot_run():
    from mcgen_amd import _lib
    ops = _ops()
    g = torch.Generator().manual_seed(941)
    x = _rnd(g, 4, 64, 8, 8)
    code = (torch.rand(4, 64, generator=g) < 0.5).float()
    wk = ops.prep_weight_k((_rnd(g, 64, 64, 3, 3) * 0.1).cuda(), torch.bfloat16)
    xt = _nhwc(ops, x, torch.bfloat16)
    with pytest.raises(_lib.McgenError):                    # 8x8 maps: no 128-pixel tile inside one image
        ops.conv_fused([ops.Seg(xt, code=code.cuda(), cmap=ops.mc_cmap(code.cuda()))], wk, 64, kmajor=True)
    with pytest.raises(_lib.McgenError):                    # a K-major launch without a map
        ops.conv_fused([ops.Seg(xt, code=code.cuda())], wk, 64, kmajor=True)


@pytest.mark.parametrize('n,h,t', [(128, 32, (256, 256)), (128, 16, (128, 256)), (640, 16, (256, 256))])
def test_compacted_activation_chain_matches_dense(n, h, t):
    """Forward-only chains keep activations compacted between launches: the producer stores, per image, only the
    channels the consumer's MultimodalController keeps (ycmap), the consumer ('gk' form) stages them contiguously and
    gathers the matching rows of its K-major weight image.  Against the dense chain conv -> BN -> ReLU -> MC -> conv
    (+ an upsampled 1x1 shortcut segment that stays dense): the stored channels are bit-identical, the statistics are
    the same sums (up to fp32 summation order), and the consumer's output is the dense launch's up to a bf16 rounding step."""
    ops = _ops()
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(951 + n + h)
    c = 256
    x0 = _rnd(g, n, c, h, h)
    xs = _rnd(g, n, c, h // 2, h // 2)                       # the block input of the shortcut segment
    w1, b1 = _rnd(g, c, c, 3, 3) * 0.03, _rnd(g, c)
    w2, ws, b2 = _rnd(g, c, c, 3, 3) * 0.03, _rnd(g, c, c, 1, 1) * 0.08, _rnd(g, c)
    scale, shift = (_rnd(g, c) * 0.5 + 1).cuda(), (_rnd(g, c) * 0.3).cuda()
    code2 = (torch.rand(n, c, generator=g) < 0.5).float()
    code2[0] = 0.0
    code2[1, :160] = 1.0; code2[1, 160:] = 0.0             # exactly fills the compacted pitch
    code1 = (torch.rand(n, c, generator=g) < 0.5).float().cuda()
    code2 = code2.cuda()
    ccap = 160
    assert int((code2 != 0).sum(1).max()) <= ccap
    cm = ops.mc_cmap(code2)
    x0t, xst = _nhwc(ops, x0, dtype), _nhwc(ops, xs, dtype)
    img1 = ops.prep_weight(w1.cuda(), dtype)
    h_d, st_d = ops.conv_fused([ops.Seg(x0t)], img1, c, bias=b1.cuda(), stats_mode=1)
    h_c, st_c = ops.conv_fused([ops.Seg(x0t)], img1, c, bias=b1.cuda(), stats_mode=1, ycmap=cm, cy=ccap)
    assert h_c.shape == (n, h, h, ccap)
    # (the same per-tile sums, accumulated in a different pixel order: the dense launch takes the merged epilogue passes)
    torch.testing.assert_close(st_c[..., :c], st_d[..., :c], rtol=2e-5, atol=2e-3)
    cidx = cm[:, c:c + ccap].long()                           # [n, ccap], value c beyond the active count
    gathered = torch.gather(torch.nn.functional.pad(h_d, (0, 1)), 3, cidx.view(n, 1, 1, ccap).expand(n, h, h, ccap))
    assert torch.equal(h_c, gathered)
    # consumer: BN affine + code folded into per-image rows; second segment dense (its code applied in the prologue)
    sc_rows, sh_rows = ops.mc_affine(code2, cm, ccap, scale, shift)
    ref_rows = torch.gather(torch.nn.functional.pad(scale.view(1, c) * code2, (0, 1)), 1, cidx)
    assert torch.equal(sc_rows, ref_rows)
    img_d = torch.cat([ops.prep_weight(w2.cuda(), dtype), ops.prep_weight(ws.cuda(), dtype)])
    img_k = torch.cat([ops.prep_weight_k(w2.cuda(), dtype), ops.prep_weight_k(ws.cuda(), dtype)])
    segs_d = [ops.Seg(h_d, scale=scale, shift=shift, code=code2, relu=True), ops.Seg(xst, ksize=1, code=code1, ups=True)]
    segs_c = [ops.Seg(h_c, scale=sc_rows, shift=sh_rows, relu=True, group_n=1, cmap=cm, cw=c), ops.Seg(xst, ksize=1, code=code1, ups=True)]
    y_d, sd = ops.conv_fused(segs_d, img_d, c, bias=b2.cuda(), stats_mode=1)
    (y_c, sc_), tiles = _conv_logged(ops, segs_c, img_k, c, bias=b2.cuda(), stats_mode=1, kmajor=2)
    assert tiles == [t], tiles
    _assert_bf16_twin(y_c, y_d, 'gathered-K consumer vs dense')
    np.testing.assert_allclose(sc_.double().sum(0).cpu(), sd.double().sum(0).cpu(), rtol=2e-4, atol=1e-2)
    # both segments compacted (the shortcut input stored compacted by the same consumer map, no BatchNorm: scale rows = code)
    cm1 = ops.mc_cmap(code1)
    cap1 = 160 if int((code1 != 0).sum(1).max()) <= 160 else 192
    s1, t1 = ops.mc_affine(code1, cm1, cap1)
    idx1 = cm1[:, c:c + cap1].long()
    valid = idx1 < c
    x_dense = xst                                          # the dense block input; its compacted twin by a torch gather
    xs_c = torch.gather(torch.nn.functional.pad(x_dense, (0, 1)), 3, idx1.view(n, 1, 1, cap1).expand(n, h // 2, h // 2, cap1)).contiguous()
    segs_d2 = [ops.Seg(h_d, scale=scale, shift=shift, code=code2, relu=True), ops.Seg(x_dense, ksize=1, code=code1, ups=True)]
    segs_c2 = [ops.Seg(h_c, scale=sc_rows, shift=sh_rows, relu=True, group_n=1, cmap=cm, cw=c),
               ops.Seg(xs_c, ksize=1, scale=s1, shift=t1, ups=True, group_n=1, cmap=cm1, cw=c)]
    y_d2, _ = ops.conv_fused(segs_d2, img_d, c, bias=b2.cuda())
    y_c2, _ = ops.conv_fused(segs_c2, img_k, c, bias=b2.cuda(), kmajor=2)
    _assert_bf16_twin(y_c2, y_d2, 'gathered-K consumer, both segments compacted')
    assert bool(valid.any())
