"""GPU parity of the MCGlow kernels and of the MCGlow likelihood forward / reverse / generate on the HIP path
against the reference-generated fixture (tests/golden/mcglow_small.npz) and the CPU oracle.
fp32 compute; tolerances: 2e-4 of the tensor's max for activations, 1e-4 absolute on bits/dim."""
import numpy as np
import pytest
import torch

import golden_util as gu

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.float().cpu(), torch.as_tensor(np.asarray(b)).float()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def _model(sd=None):
    from mcgen_amd import models
    from mcgen_amd.config import cfg
    cfg.update(model_name='mcglow', device='cuda', classes_size=12, controller_rate=0.5, data_shape=[1, 32, 32],
               compute_dtype='float32')
    cfg['glow'] = {'hidden_size': 32, 'K': 2, 'L': 3, 'affine': True, 'conv_lu': True}
    np.random.seed(0)
    m = models.mcglow()
    if sd is not None:
        m.load_state_dict(sd)
    return m.cuda()


def test_glow_kernels():
    from mcgen_amd import ops
    from oracle import mcglow_oracle as G
    g = torch.Generator().manual_seed(3)
    x = torch.randn(3, 6, 8, 8, generator=g)
    xs = ops.glow_squeeze(ops.to_nhwc(x.cuda(), torch.float32), 6)
    assert xs.shape == (3, 4, 4, 24)
    assert torch.equal(ops.to_nchw(xs, 24).cpu(), G.squeeze(x))
    assert torch.equal(ops.to_nchw(ops.glow_unsqueeze(xs, 24), 6).cpu(), x)
    # LU-parameterised weight and its inverse
    d = gu.load_npz('mcglow_small.npz')
    sd = gu.state_from_npz(d, 'sd_final/')
    p = 'blocks.2.flows.1.invconv.'
    w, winv = ops.invconv_weight(*(sd[p + k].cuda() for k in ('w_p', 'w_l', 'w_u', 'w_s', 's_sign')), inverse=True)
    ref = G.invconv_lu_weight(sd, p)
    assert _rel(w, ref) < 1e-5 and _rel(winv, torch.inverse(ref)) < 1e-4
    # ActNorm data init from channel statistics (unbiased std)
    a = torch.randn(5, 8, 6, 6, generator=g) * 3 + 1
    part = ops.channel_stats(ops.to_nhwc(a.cuda(), torch.float32))
    loc, scale = torch.zeros(1, 8, 1, 1, device='cuda'), torch.ones(1, 8, 1, 1, device='cuda')
    ops.actnorm_init(part, 5 * 36, loc, scale)
    assert _rel(loc.view(-1), -a.mean((0, 2, 3))) < 1e-5
    assert _rel(scale.view(-1), 1 / (a.std((0, 2, 3)) + 1e-6)) < 1e-5
    # affine coupling forward / reverse and its log-determinant
    c = 12
    xin, h = torch.randn(4, c, 4, 4, generator=g), torch.randn(4, c, 4, 4, generator=g)
    log_s, t = h.chunk(2, 1)
    s = torch.sigmoid(log_s + 2)
    ref_y = torch.cat([xin[:, :c // 2], (xin[:, c // 2:] + t) * s], 1)
    ld = torch.zeros(4, device='cuda')
    xt, ht = ops.to_nhwc(xin.cuda(), torch.float32), ops.to_nhwc(h.cuda(), torch.float32)
    y = ops.glow_coupling(xt, ht, c, ld)
    assert _rel(ops.to_nchw(y, c), ref_y) < 1e-6
    assert _rel(ld, torch.log(s).reshape(4, -1).sum(1)) < 1e-5
    back = ops.glow_coupling(y, ht, c, None, reverse=True)
    assert _rel(ops.to_nchw(back, c), xin) < 1e-5
    # Gaussian prior log-density
    z, prior = torch.randn(4, 6, 4, 4, generator=g), torch.randn(4, 12, 4, 4, generator=g) * 0.3
    lp = torch.zeros(4, device='cuda')
    ops.gaussian_logp(ops.to_nhwc(z.cuda(), torch.float32), 0, ops.to_nhwc(prior.cuda(), torch.float32), 6, lp)
    mean, lsd = prior.chunk(2, 1)
    assert _rel(lp, G.gaussian_log_p(z, mean, lsd).reshape(4, -1).sum(1)) < 1e-5


def test_mcglow_forward_init_and_reverse():
    d = gu.load_npz('mcglow_small.npz')
    img, lab = torch.from_numpy(d['img']).cuda(), torch.from_numpy(d['label']).cuda()
    # 1. data-dependent ActNorm initialisation on the first training forward (train_glow.py:60-67)
    m = _model(gu.state_from_npz(d))
    m.train(True)
    with torch.no_grad():
        m({'img': img, 'label': lab, 'noise': torch.from_numpy(d['noise/init/0']).cuda()})
    init = gu.state_from_npz(d, 'sd_init/')
    sd = m.state_dict()
    for k, v in init.items():
        if k.endswith(('loc', 'scale')):
            assert float((sd[k].cpu() - v).abs().max()) < 2e-3 * float(v.abs().max()) + 1e-5, k
        if k.endswith('initialized'):
            assert int(sd[k]) == 1
    # 2. training-mode likelihood on the initialised weights = first logged loss of the fixture
    m = _model(init)
    m.train(True)
    out = m({'img': img, 'label': lab, 'noise': torch.from_numpy(d['noise/0/0']).cuda()})
    assert abs(float(out['loss']) - float(d['losses'][0])) < 1e-4
    for i, z in enumerate(out['z']):
        assert _rel(z, d[f'z0/{i}']) < 5e-4, i
    # 3. eval forward, reverse(reconstruct) and generate on the reference's final weights
    m = _model(gu.state_from_npz(d, 'sd_final/'))
    m.train(False)
    out = m({'img': img, 'label': lab, 'noise': torch.from_numpy(d['noise/eval/0']).cuda()})
    assert abs(float(out['loss']) - float(d['loss_eval'])) < 1e-4
    rec = m.reverse({'z': out['z'], 'label': lab, 'reconstruct': True})['img']
    assert _rel(rec, d['reconstructed']) < 1e-3
    gz = [torch.from_numpy(d[f'gen_z/{i}']).cuda() for i in range(3)]
    assert _rel(m.generate(lab, gz), d['generated']) < 1e-3
    assert [tuple(s) for s in m.make_z_shapes()] == [(2, 16, 16), (4, 8, 8), (16, 4, 4)]
