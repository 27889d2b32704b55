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


def test_glow_backward_kernels():
    """coupling / prior backward kernels against torch autograd of the same formulas."""
    from mcgen_amd import ops
    g = torch.Generator().manual_seed(5)
    c, n = 12, 4
    v = torch.randn(n, c, 4, 4, generator=g, requires_grad=True)
    h = torch.randn(n, c, 4, 4, generator=g, requires_grad=True)
    dy = torch.randn(n, c, 4, 4, generator=g)
    g0 = -0.37
    log_s, t = h.chunk(2, 1)
    s = torch.sigmoid(log_s + 2)
    y = torch.cat([v[:, :c // 2], (v[:, c // 2:] + t) * s], 1)
    ((y * dy).sum() + g0 * torch.log(s).sum()).backward()
    f32 = torch.float32
    dv, dh = ops.glow_coupling_bwd(ops.to_nhwc(v.detach().cuda(), f32), ops.to_nhwc(h.detach().cuda(), f32),
                                   ops.to_nhwc(dy.cuda(), f32), c, g0)
    assert _rel(ops.to_nchw(dv, c), v.grad) < 1e-5 and _rel(ops.to_nchw(dh, c), h.grad) < 1e-5
    from oracle import mcglow_oracle as G
    z = torch.randn(n, 6, 4, 4, generator=g, requires_grad=True)
    prior = (torch.randn(n, 12, 4, 4, generator=g) * 0.3).requires_grad_(True)
    mean, lsd = prior.chunk(2, 1)
    (g0 * G.gaussian_log_p(z, mean, lsd).sum()).backward()
    zt = ops.to_nhwc(z.detach().cuda(), f32)
    dz = torch.zeros_like(zt)
    dp = ops.gaussian_logp_bwd(zt, 0, ops.to_nhwc(prior.detach().cuda(), f32), 6, dz, 0, g0, False)
    assert _rel(ops.to_nchw(dz, 6), z.grad) < 1e-5 and _rel(ops.to_nchw(dp, 12), prior.grad) < 1e-5
    # prod_colsum / clip_grad_norm
    a, b = torch.randn(700, 16, generator=g).cuda(), torch.randn(700, 16, generator=g).cuda()
    out = torch.zeros(10, device='cuda')
    ops.prod_colsum(a, b, 10, out, alpha=3.0)
    assert _rel(out, 3 * (a * b).sum(0)[:10].cpu()) < 1e-5
    gf = torch.randn(100003, generator=g).cuda()
    ref = gf.clone()
    nrm = ops.clip_grad_norm_(gf, 1.0)
    assert abs(float(nrm) - float(ref.norm())) < 1e-2
    assert _rel(gf, (ref / (ref.norm() + 1e-6)).cpu()) < 1e-5
    small = ref * 1e-4
    keep = small.clone()
    ops.clip_grad_norm_(small, 1.0)
    assert torch.equal(small, keep)


def _oracle_grads(sd, img, lab, noise):
    from oracle import mcglow_oracle as G
    sdg = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and not k.endswith(('w_p', 'u_mask', 'l_mask', 's_sign', 'l_eye', 'codebook'))
               else v.clone()) for k, v in sd.items()}
    out = G.forward(sdg, img, lab, 12, 2, 3, noise, train=True)
    out['loss'].backward()
    return float(out['loss'].detach()), {k: v.grad for k, v in sdg.items() if v.requires_grad and v.grad is not None}


def test_mcglow_gradients_vs_oracle():
    """d(bits/dim)/d(every parameter) from the HIP backward against autograd through the CPU oracle."""
    d = gu.load_npz('mcglow_small.npz')
    img, lab = torch.from_numpy(d['img']), torch.from_numpy(d['label'])
    noise = torch.from_numpy(d['noise/0/0'])
    init = gu.state_from_npz(d, 'sd_init/')
    # perturb the zero-initialised ZeroConv2d weights so every path carries gradient
    g = torch.Generator().manual_seed(11)
    for k in init:
        if '.conv.weight' in k or k.endswith('prior.scale') or k.endswith('8.module.scale'):
            init[k] = init[k] + 0.02 * torch.randn(init[k].shape, generator=g)
    loss_ref, gref = _oracle_grads(init, img, lab, noise)
    m = _model(init)
    m.train(True)
    out = m({'img': img.cuda(), 'label': lab.cuda(), 'noise': noise.cuda()})
    assert abs(float(out['loss']) - loss_ref) < 1e-4
    out['loss'].backward()
    named = dict(m.named_parameters())
    assert set(gref) == set(named), set(gref) ^ set(named)
    worst = 0.0
    for k, gr in gref.items():
        gg = named[k].grad
        assert gg is not None, k
        err = float((gg.cpu() - gr).abs().max())
        tol = 2e-4 * float(gr.abs().max()) + 1e-6
        assert err < tol, (k, err, tol)
        worst = max(worst, err / tol)
    print('worst err/tol', worst)


def test_mcglow_two_training_steps_vs_reference():
    """train_glow.py loop body x2 (clip_grad_norm_ 1, Adam 3e-4) from the fixture's initialised weights:
    logged losses and final weights.  Adam's first steps move every weight by ~lr * sign(g), so weights whose
    gradient is rounding noise may differ by up to 2 * lr per step; bound: 2 steps * 2 * 3e-4."""
    from mcgen_amd.trainer import GlowTrainer
    d = gu.load_npz('mcglow_small.npz')
    img, lab = torch.from_numpy(d['img']).cuda(), torch.from_numpy(d['label']).cuda()
    m = _model(gu.state_from_npz(d, 'sd_init/'))
    tr = GlowTrainer(m)
    losses = [float(tr.train_iteration(img, lab, torch.from_numpy(d[f'noise/{s}/0']).cuda())) for s in range(2)]
    assert abs(losses[0] - d['losses'][0]) < 1e-4 and abs(losses[1] - d['losses'][1]) < 5e-4, (losses, d['losses'])
    fin = gu.state_from_npz(d, 'sd_final/')
    sd = m.state_dict()
    far = 0
    for k, v in fin.items():
        if not v.dtype.is_floating_point:
            continue
        diff = (sd[k].cpu() - v).abs()
        assert float(diff.max()) < 1.3e-3, (k, float(diff.max()))
        far += int((diff > 1e-5 + 1e-3 * v.abs()).sum())
    total = sum(v.numel() for v in fin.values() if v.dtype.is_floating_point)
    assert far < 0.02 * total, (far, total)


def test_mcglow_graphed_trainer_tracks_eager():
    """HIP-graph replay of the train step, pinned: capture leaves the weights untouched, and replays with the
    fixture's dequantisation noise injected reproduce the reference's two logged losses and the eager trainer's."""
    from mcgen_amd.trainer import GlowTrainer
    d = gu.load_npz('mcglow_small.npz')
    img, lab = torch.from_numpy(d['img']).cuda(), torch.from_numpy(d['label']).cuda()
    init = gu.state_from_npz(d, 'sd_init/')
    ma, mb = _model(init), _model(init)
    ta, tb = GlowTrainer(ma), GlowTrainer(mb)
    tb.capture(img, lab, warmup=1)
    for k, v in mb.state_dict().items():
        assert torch.equal(v.cpu(), init[k]), k
    noise = [torch.from_numpy(d[f'noise/{s}/0']).cuda() for s in range(2)]
    la = [float(ta.train_iteration(img, lab, noise[s])) for s in range(2)]
    lb = [float(tb.train_iteration(img, lab, noise[s])) for s in range(2)]
    assert abs(lb[0] - d['losses'][0]) < 1e-4 and abs(lb[1] - d['losses'][1]) < 5e-4, (lb, d['losses'])
    assert max(abs(x - y) for x, y in zip(la, lb)) < 1e-5, (la, lb)
    # without injected noise the graph draws its own
    lc = [float(tb.train_iteration(img, lab)) for _ in range(2)]
    assert all(np.isfinite(lc))


def test_mcglow_bf16_tracks_fp32():
    """The throughput build computes in bf16 (fp32 accumulation, fp32 log-determinants): on the fixture the
    likelihood stays within 2e-2 bits/dim of the fp32 reference value and two training steps reduce the loss."""
    from mcgen_amd.trainer import GlowTrainer
    d = gu.load_npz('mcglow_small.npz')
    img, lab = torch.from_numpy(d['img']).cuda(), torch.from_numpy(d['label']).cuda()
    m = _model(gu.state_from_npz(d, 'sd_init/')).set_compute_dtype(torch.bfloat16)
    m.train(True)
    with torch.no_grad():
        out = m({'img': img, 'label': lab, 'noise': torch.from_numpy(d['noise/0/0']).cuda()})
    assert abs(float(out['loss']) - float(d['losses'][0])) < 2e-2
    tr = GlowTrainer(m)
    losses = [float(tr.train_iteration(img, lab, torch.from_numpy(d[f'noise/{s}/0']).cuda())) for s in range(2)]
    assert abs(losses[0] - d['losses'][0]) < 2e-2 and abs(losses[1] - d['losses'][1]) < 3e-2, (losses, d['losses'])


def _model_omniglot(sd, dtype=torch.float32):
    from mcgen_amd import models
    from mcgen_amd.config import cfg
    cfg.update(model_name='mcglow', device='cuda', classes_size=1623, controller_rate=0.5, data_shape=[1, 32, 32],
               compute_dtype='float32')
    cfg['glow'] = {'hidden_size': 512, 'K': 16, 'L': 3, 'affine': True, 'conv_lu': True}
    np.random.seed(0)
    m = models.mcglow()
    m.load_state_dict(sd)
    m = m.cuda()
    return m.set_compute_dtype(dtype) if dtype != torch.float32 else m


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_mcglow_omniglot_full_size(dtype):
    """BASELINE configs[3] as the reference runs it (utils.py:110-112,172-184; data.py:40): [1,32,32], 1623 modes,
    hidden 512, K=16, L=3 -- 15,844,992 parameters, label-gathered codes, the K-deep 512 -> 512 1x1 form -- against
    the reference-generated mcglow_full_digest.npz (procedural weights, B=4): ActNorm data init, the likelihood and
    latents of the first training forward, and the losses of two train_glow.py steps."""
    import ast
    from mcgen_amd.trainer import GlowTrainer
    d = gu.load_npz('mcglow_full_digest.npz')
    shapes = {str(k): ast.literal_eval(str(v)) for k, v in zip(d['shape_keys'], d['shape_vals'])}
    dtypes = {str(k): str(v) for k, v in zip(d['shape_keys'], d['dtype_vals'])}
    sd = gu.procedural_state_glow(shapes, dtypes, seed=777)
    m = _model_omniglot(sd, dtype)
    assert sum(p.numel() for p in m.parameters()) == 15844992
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == shapes
    img, lab = torch.from_numpy(d['img']).cuda(), torch.from_numpy(d['label']).cuda()
    f32 = dtype == torch.float32
    m.train(True)
    with torch.no_grad():
        m({'img': img, 'label': lab, 'noise': torch.from_numpy(d['noise/init/0']).cuda()})
    sdi = m.state_dict()
    for k in d:
        if k.startswith('init_digest/'):
            got, ref = gu.checksum(sdi[k[len('init_digest/'):]].float().cpu()), d[k]
            assert np.abs(got - ref).max() < (1e-3 if f32 else 3e-2) * ref[1], (k, got, ref)
        if k.endswith('initialized') and k in sdi:
            assert int(sdi[k]) == 1
    with torch.no_grad():
        out = m({'img': img, 'label': lab, 'noise': torch.from_numpy(d['noise/0/0']).cuda()})
    print('first training-mode loss', float(out['loss']), 'reference', float(d['losses'][0]))
    assert abs(float(out['loss']) - float(d['losses'][0])) < (1e-3 if f32 else 1.5e-1)
    for i, z in enumerate(out['z']):
        zs = z.float()[:, :, ::2, ::2]
        assert _rel(zs, d[f'z0_sample/{i}']) < (2e-3 if f32 else 1e-1), i
        got, ref = gu.checksum(z.float().cpu()), d[f'z0_digest/{i}']
        assert np.abs(got - ref).max() < (5e-4 if f32 else 3e-2) * ref[1], (i, got, ref)
    tr = GlowTrainer(m)
    losses = [float(tr.train_iteration(img, lab, torch.from_numpy(d[f'noise/{s}/0']).cuda())) for s in range(2)]
    print('train losses', losses, 'reference', d['losses'])
    assert abs(losses[0] - d['losses'][0]) < (1e-3 if f32 else 1.5e-1)
    assert abs(losses[1] - d['losses'][1]) < (5e-2 if f32 else 4e-1)
    if f32:
        fin = m.state_dict()
        for k in d:
            if k.startswith('final_digest/'):
                got, ref = gu.checksum(fin[k[len('final_digest/'):]].float().cpu()), d[k]
                assert np.abs(got - ref).max() < 2e-3 * ref[1], (k, got, ref)
