"""GPU parity of the MCVAE forward / backward / train step on the HIP path against the reference-generated fixture
(tests/golden/mcvae_small.npz) and autograd through the CPU oracle.  fp32 compute."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import golden_util as gu

pytestmark = pytest.mark.gpu
HIDDEN, LATENT = [8, 16, 32], 16


def _rel(a, b):
    a, b = a.float().cpu(), torch.as_tensor(np.asarray(b)).float()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def _model(sd):
    from mcgen_amd import models
    from mcgen_amd.config import cfg
    cfg.update(model_name='mcvae', data_name='CIFAR10', device='cuda', classes_size=10, controller_rate=0.5,
               data_shape=[3, 32, 32], compute_dtype='float32')
    cfg['vae'] = {'hidden_size': HIDDEN, 'latent_size': LATENT, 'num_res_block': 2, 'embedding_size': 32}
    m = models.mcvae()
    m.load_state_dict(sd)
    return m.cuda()


def test_strided_im2col_and_transposed_conv():
    """im2col(stride 2) + 1x1 == Conv2d(4,2,1); 1x1 + col2im(stride 2) == ConvTranspose2d(4,2,1)."""
    from mcgen_amd import ops
    from mcgen_amd.ops import Seg
    g = torch.Generator().manual_seed(9)
    f32 = torch.float32
    x = torch.randn(3, 8, 8, 8, generator=g)
    w = torch.randn(16, 8, 4, 4, generator=g) * 0.1
    b = torch.randn(16, generator=g)
    ref = F.conv2d(x, w, b, stride=2, padding=1)
    xt = ops.to_nhwc(x.cuda(), f32)
    col = ops.im2col(xt, 4, 4, 1, 1, stride=2)
    wm = w.permute(0, 2, 3, 1).reshape(16, 128, 1, 1).contiguous().cuda()
    y, _ = ops.conv_fused([Seg(col, ksize=1)], ops.prep_weight(wm, f32), 16, bias=b.cuda())
    assert _rel(ops.to_nchw(y, 16), ref) < 1e-5
    wt = torch.randn(8, 16, 4, 4, generator=g) * 0.1                       # ConvTranspose2d weight [ci, co, 4, 4]
    reft = F.conv_transpose2d(x, wt, b, stride=2, padding=1)
    wmt = wt.permute(2, 3, 1, 0).reshape(16 * 16, 8, 1, 1).contiguous().cuda()
    dcol, _ = ops.conv_fused([Seg(xt, ksize=1)], ops.prep_weight(wmt, f32), 256)
    out = ops.col2im(dcol, 16, 4, 4, 1, 1, stride=2, bias=b.cuda())
    assert _rel(ops.to_nchw(out, 16), reft) < 1e-5
    # prologue inside im2col: zero padding applies to the activated tensor
    sc, sh = torch.rand(8, generator=g) + 0.5, torch.randn(8, generator=g)
    code = (torch.rand(3, 8, generator=g) < 0.5).float()
    act = torch.relu(x * sc[None, :, None, None] + sh[None, :, None, None]) * code[:, :, None, None]
    col2 = ops.im2col(xt, 4, 4, 1, 1, stride=2, scale=sc.cuda(), shift=sh.cuda(), relu=True, code=code.cuda())
    y2, _ = ops.conv_fused([Seg(col2, ksize=1)], ops.prep_weight(wm, f32), 16, bias=b.cuda())
    assert _rel(ops.to_nchw(y2, 16), F.conv2d(act, w, b, stride=2, padding=1)) < 1e-5
    # BCE with logits
    a = torch.randn(2, 3, 4, 4, generator=g) * 4
    t = torch.rand(2, 3, 4, 4, generator=g)
    recon, s, da = ops.bce_logits(ops.to_nhwc(a.cuda(), f32), ops.to_nhwc(t.cuda(), f32), 3, 0.5, True)
    # float64 reference: torch's fp32 log(1 - sigmoid(a)) loses ~1e-2 per saturated element, the kernel uses softplus
    assert abs(float(s) - float(F.binary_cross_entropy(torch.sigmoid(a.double()), t.double(), reduction='sum'))) < 1e-3
    assert _rel(ops.to_nchw(da, 3), (torch.sigmoid(a) - t) * 0.5) < 1e-5


def test_mcvae_forward_vs_reference():
    d = gu.load_npz('mcvae_small.npz')
    img, lab = torch.from_numpy(d['img']).cuda(), torch.from_numpy(d['label']).cuda()
    m = _model(gu.state_from_npz(d))
    m.train(True)
    with torch.no_grad():
        out = m({'img': img, 'label': lab, 'eps': torch.from_numpy(d['noise/0/0']).cuda()})
    assert abs(float(out['loss']) - float(d['losses'][0])) < 1e-5
    assert _rel(out['mu'], d['mu0']) < 2e-4 and _rel(out['logvar'], d['logvar0']) < 2e-4
    assert _rel(out['img'], d['img0']) < 2e-4
    m = _model(gu.state_from_npz(d, 'sd_final/'))
    m.train(False)
    gen = m.generate(lab, torch.from_numpy(d['gen_z']).cuda())
    assert _rel(gen, d['generated_eval']) < 5e-4


def test_mcvae_gradients_vs_oracle():
    from oracle import mcvae_oracle as O
    d = gu.load_npz('mcvae_small.npz')
    img, lab, eps = torch.from_numpy(d['img']), torch.from_numpy(d['label']), torch.from_numpy(d['noise/0/0'])
    sd = gu.state_from_npz(d)
    skip = ('running_mean', 'running_var', 'num_batches_tracked', 'codebook')
    sdg = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and not k.endswith(skip) else v.clone()) for k, v in sd.items()}
    ref = O.forward(sdg, img, lab, 10, HIDDEN, 2, train=True, eps=eps)
    ref['loss'].backward()
    m = _model(gu.state_from_npz(d))
    m.train(True)
    out = m({'img': img.cuda(), 'label': lab.cuda(), 'eps': eps.cuda()})
    assert abs(float(out['loss'].detach()) - float(ref['loss'].detach())) < 1e-5
    out['loss'].backward()
    named = dict(m.named_parameters())
    checked = 0
    for k, v in sdg.items():
        if not v.requires_grad or v.grad is None:
            continue
        gg = named[k].grad
        assert gg is not None, k
        err = float((gg.cpu() - v.grad).abs().max())
        # conv / linear biases that feed straight into a BatchNorm have an exactly-zero gradient: absolute floor
        tol = 5e-4 * float(v.grad.abs().max()) + 2e-7
        assert err < tol, (k, err, tol)
        checked += 1
    assert checked == len(named)


def test_mcvae_train_steps_vs_reference():
    """train_vae.py loop body x3 (clip_grad_norm_ 1, Adam 3e-4) from the fixture's weights with its noise."""
    from mcgen_amd.trainer import VAETrainer
    d = gu.load_npz('mcvae_small.npz')
    img, lab = torch.from_numpy(d['img']).cuda(), torch.from_numpy(d['label']).cuda()
    m = _model(gu.state_from_npz(d))
    tr = VAETrainer(m)
    losses = [float(tr.train_iteration(img, lab, torch.from_numpy(d[f'noise/{s}/0']).cuda())) for s in range(3)]
    assert abs(losses[0] - d['losses'][0]) < 1e-5, (losses, d['losses'])
    assert max(abs(a - b) for a, b in zip(losses, d['losses'])) < 2e-3, (losses, d['losses'])
    fin = gu.state_from_npz(d, 'sd_final/')
    sd = m.state_dict()
    for k, v in fin.items():
        if v.dtype.is_floating_point and not k.endswith(('running_mean', 'running_var')):
            assert float((sd[k].cpu() - v).abs().max()) < 2e-3, k
    # graph replay (draws its own eps) keeps training
    m2 = _model(gu.state_from_npz(d))
    t2 = VAETrainer(m2)
    t2.capture(img, lab, warmup=1)
    l2 = [float(t2.train_iteration(img, lab)) for _ in range(3)]
    assert all(np.isfinite(l2)) and abs(l2[0] - losses[1]) < 2e-2


def test_replay_follows_a_learning_rate_change_without_recapture():
    """the fused Adam reads its learning rate from device memory (mcgen_adam's lr_dev), so a scheduler
    step between replays of a captured train step needs no re-capture: an eager trainer and a graph-replaying one, fed
    the same noise, with the rate halved after the first step and restored after the second, end in the same
    parameters; a parameter the kernels bake in (beta) still refuses to replay."""
    from mcgen_amd.trainer import VAETrainer
    d = gu.load_npz('mcvae_small.npz')
    img, lab = torch.from_numpy(d['img']).cuda(), torch.from_numpy(d['label']).cuda()
    eps = [torch.from_numpy(d[f'noise/{s}/0']).cuda() for s in range(3)]
    te, tg = VAETrainer(_model(gu.state_from_npz(d))), VAETrainer(_model(gu.state_from_npz(d)))
    tg.capture(img, lab, warmup=1)
    key = tg._hyper_key
    rates = [3e-4, 1.5e-4, 6e-4]
    for s in range(3):
        te.opt.set_lr(rates[s]); tg.opt.set_lr(rates[s])
        le, lg = te.train_iteration(img, lab, eps[s]), tg.train_iteration(img, lab, eps[s])
        assert abs(float(le) - float(lg)) < 1e-6 * (1 + abs(float(le))), (s, float(le), float(lg))
    assert tg._hyper_key == key and tg.opt.state_dict()['param_groups'][0]['lr'] == 6e-4
    for (k, a), (_, b) in zip(te.model.state_dict().items(), tg.model.state_dict().items()):
        assert float((a.float() - b.float()).abs().max()) <= 1e-6 * (1 + float(a.float().abs().max())), k
    # the rate really is applied: a third trainer that never changes it ends elsewhere
    t0 = VAETrainer(_model(gu.state_from_npz(d)))
    for s in range(3):
        t0.train_iteration(img, lab, eps[s])
    w0, wg = next(iter(t0.model.parameters())), next(iter(tg.model.parameters()))
    assert float((w0 - wg).abs().max()) > 1e-5
    tg.opt.betas = (0.8, 0.999)
    with pytest.raises(RuntimeError):
        tg.train_iteration(img, lab, eps[0])


def test_mcvae_config0_full_size():
    """BASELINE.json configs[0] (MCVAE CIFAR-10, hidden [64,128,256], latent 128, batch 32) on the HIP path:
    two optimizer steps against the reference's losses / digests (procedural weights, see golden_util)."""
    import ast
    from mcgen_amd import models
    from mcgen_amd.config import cfg
    from mcgen_amd.trainer import VAETrainer
    d = gu.load_npz('mcvae_full_digest.npz')
    shapes = {k: ast.literal_eval(v) for k, v in zip(d['shape_keys'].tolist(), d['shape_vals'].tolist())}
    sd = gu.procedural_state_generic(shapes, seed=4321)
    cfg.update(model_name='mcvae', data_name='CIFAR10', device='cuda', classes_size=10, controller_rate=0.5,
               data_shape=[3, 32, 32], compute_dtype='float32')
    cfg['vae'] = {'hidden_size': [64, 128, 256], 'latent_size': 128, 'num_res_block': 2, 'embedding_size': 32}
    m = models.mcvae()
    m.load_state_dict(sd)
    m = m.cuda()
    assert sum(p.numel() for p in m.parameters()) == 7628931
    img, lab = gu.synthetic_batch(32, 10, seed=1)
    img, lab = img.cuda(), lab.cuda()
    m.train(True)
    with torch.no_grad():
        first = m({'img': img, 'label': lab, 'eps': torch.from_numpy(d['noise/0/0']).cuda()})
    assert abs(float(first['loss']) - float(d['losses'][0])) < 1e-5
    np.testing.assert_allclose(gu.checksum(first['mu'].cpu()), d['mu0_digest'], rtol=1e-4, atol=1e-3)
    assert _rel(first['img'][:4, :, ::4, ::4], d['img0_sample']) < 5e-4
    m.load_state_dict(sd)                                        # the probe forward moved the BN running statistics
    tr = VAETrainer(m.cuda())
    losses = [float(tr.train_iteration(img, lab, torch.from_numpy(d[f'noise/{s}/0']).cuda())) for s in range(2)]
    assert abs(losses[0] - d['losses'][0]) < 1e-5 and abs(losses[1] - d['losses'][1]) < 2e-3, (losses, d['losses'])


def test_mcvae_bf16_tracks_fp32():
    """bf16 compute (the throughput build) on the fixture: loss within 1e-2 of the fp32 reference, reconstruction within
    3 % of its range, three training steps follow the reference losses."""
    from mcgen_amd.trainer import VAETrainer
    d = gu.load_npz('mcvae_small.npz')
    img, lab = torch.from_numpy(d['img']).cuda(), torch.from_numpy(d['label']).cuda()
    m = _model(gu.state_from_npz(d)).set_compute_dtype(torch.bfloat16)
    m.train(True)
    with torch.no_grad():
        out = m({'img': img, 'label': lab, 'eps': torch.from_numpy(d['noise/0/0']).cuda()})
    assert abs(float(out['loss']) - float(d['losses'][0])) < 1e-2
    assert _rel(out['img'], d['img0']) < 3e-2
    m = _model(gu.state_from_npz(d)).set_compute_dtype(torch.bfloat16)
    tr = VAETrainer(m)
    losses = [float(tr.train_iteration(img, lab, torch.from_numpy(d[f'noise/{s}/0']).cuda())) for s in range(3)]
    assert max(abs(a - b) for a, b in zip(losses, d['losses'])) < 2e-2, (losses, d['losses'])
