"""Pins the CPU oracles of MCVAE / MCPixelCNN / MCGlow to vectors produced by
running the reference models (tools/gen_golden.py).  Config 0 of BASELINE.json (MCVAE CIFAR-10, batch 32,
CPU) is covered at full size by `test_mcvae_config0_full_size`.

Train loop restated from train_vae.py:98-126 (identical in train_glow.py / train_pixelcnn.py):
zero_grad, forward, backward, clip_grad_norm_(1), Adam(lr 3e-4).step(); the noise the reference draws
inside its models is injected from the fixture."""
import ast

import numpy as np
import pytest
import torch

import golden_util as gu
from oracle import mcglow_oracle as G
from oracle import mcpixelcnn_oracle as P
from oracle import mcvae_oracle as V

torch.set_num_threads(8)
BUFFERS = ('running_mean', 'running_var', 'num_batches_tracked', 'codebook', 'initialized', 'w_p', 'u_mask',
           'l_mask', 's_sign', 'l_eye')


def _params(sd):
    return [k for k, v in sd.items() if v.is_floating_point() and not k.endswith(BUFFERS)]


def _train(sd, step_fn, steps):
    keys = _params(sd)
    for k in keys:
        sd[k].requires_grad_(True)
    opt = torch.optim.Adam([sd[k] for k in keys], lr=3e-4)
    losses, first = [], None
    for s in range(steps):
        opt.zero_grad()
        out = step_fn(s)
        out['loss'].backward()
        torch.nn.utils.clip_grad_norm_([sd[k] for k in keys], 1.0)
        opt.step()
        losses.append(float(out['loss'].detach()))
        first = first or out
    return losses, first


def _close(a, b, rtol=2e-4, atol=2e-5, what=''):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    np.testing.assert_allclose(a, np.asarray(b), rtol=rtol, atol=atol, err_msg=what)


def test_mcvae_small():
    d = gu.load_npz('mcvae_small.npz')
    sd = gu.state_from_npz(d)
    img, lab = torch.from_numpy(d['img']), torch.from_numpy(d['label'])
    hidden, n_res = [8, 16, 32], 2
    losses, first = _train(sd, lambda s: V.forward(sd, img, lab, 10, hidden, n_res, True,
                                                   torch.from_numpy(d[f'noise/{s}/0'])), 3)
    _close(first['mu'], d['mu0'], what='mu'); _close(first['logvar'], d['logvar0'], what='logvar')
    _close(first['img'], d['img0'], what='img')
    np.testing.assert_allclose(losses[0], d['losses'][0], rtol=0, atol=1e-5)
    np.testing.assert_allclose(losses, d['losses'], rtol=0, atol=2e-3)
    fin = gu.state_from_npz(d, 'sd_final/')
    for k in ('encoder.blocks.1.module.running_var', 'decoder.linear.2.module.running_mean'):
        _close(sd[k], fin[k], rtol=2e-3, atol=2e-4, what=k)
    with torch.no_grad():
        gen = V.generate(sd, lab, torch.from_numpy(d['gen_z']), 10, hidden, n_res)
    _close(gen, d['generated_eval'], rtol=1e-2, atol=5e-3, what='eval generate after training')


def test_mcvae_config0_full_size():
    """BASELINE.json configs[0]: MCVAE CIFAR-10 32x32 control=0.5 batch=32 on CPU."""
    d = gu.load_npz('mcvae_full_digest.npz')
    shapes = {k: ast.literal_eval(v) for k, v in zip(d['shape_keys'].tolist(), d['shape_vals'].tolist())}
    sd = gu.procedural_state_generic(shapes, seed=4321)
    n_params = sum(int(np.prod(s)) for k, s in shapes.items() if k in _params(sd))
    assert n_params == 7628931
    img, lab = gu.synthetic_batch(32, 10, seed=1)
    losses, first = _train(sd, lambda s: V.forward(sd, img, lab, 10, [64, 128, 256], 2, True,
                                                   torch.from_numpy(d[f'noise/{s}/0'])), 2)
    np.testing.assert_allclose(losses[0], d['losses'][0], rtol=0, atol=1e-5)
    np.testing.assert_allclose(losses[1], d['losses'][1], rtol=0, atol=2e-3)
    np.testing.assert_allclose(gu.checksum(first['mu']), d['mu0_digest'], rtol=1e-4, atol=1e-3)
    _close(first['img'][:4, :, ::4, ::4], d['img0_sample'], what='reconstruction sample')


def test_mcpixelcnn_small():
    d = gu.load_npz('mcpixelcnn_small.npz')
    sd = gu.state_from_npz(d)
    codes, lab = torch.from_numpy(d['codes']), torch.from_numpy(d['label'])
    losses, first = _train(sd, lambda s: P.forward(sd, codes, lab, 10, True), 3)
    _close(first['logits'], d['logits0'], what='logits')
    np.testing.assert_allclose(losses[0], d['losses'][0], rtol=0, atol=1e-5)
    np.testing.assert_allclose(losses, d['losses'], rtol=0, atol=2e-3)
    with torch.no_grad():
        _close(P.forward(sd, codes, lab, 10, False)['logits'], d['logits_eval'], rtol=1e-2, atol=5e-3, what='eval logits')
    # mask 'A' (make_causal, mcpixelcnn.py:43-45): the optimizer may move the masked taps, every forward
    # zeroes the last kernel row / column of layer 0 again, in place
    assert float(sd['layers.0.vert_stack.weight'].detach()[:, :, -1].abs().max()) == 0.0
    assert float(sd['layers.0.horiz_stack.weight'].detach()[:, :, :, -1].abs().max()) == 0.0


def test_mcpixelcnn_full_size_digest():
    """BASELINE configs[4] at its real size (15 layers, hidden 128, 512 codes: utils.py:139-143): the oracle on the
    procedural weights against the reference-generated mcpixelcnn_full_digest.npz (loss, logits digest + sample of the
    first training forward at the config's batch 128; the second loss after one clip + Adam step)."""
    import ast
    d = gu.load_npz('mcpixelcnn_full_digest.npz')
    shapes = {str(k): ast.literal_eval(str(v)) for k, v in zip(d['shape_keys'], d['shape_vals'])}
    sd = gu.procedural_state_generic(shapes, seed=4242)
    codes, lab = torch.from_numpy(d['codes']), torch.from_numpy(d['label'])
    losses, first = _train(sd, lambda s: P.forward(sd, codes, lab, 10, True), 2)
    np.testing.assert_allclose(losses[0], d['losses'][0], rtol=0, atol=2e-5)
    np.testing.assert_allclose(losses[1], d['losses'][1], rtol=0, atol=5e-3)
    np.testing.assert_allclose(gu.checksum(first['logits']), d['logits0_digest'], rtol=2e-4, atol=2e-2)
    _close(first['logits'][::16, ::16, ::2, ::2], d['logits0_sample'], rtol=1e-3, atol=1e-3, what='logits sample')


def test_mcglow_small():
    d = gu.load_npz('mcglow_small.npz')
    sd = gu.state_from_npz(d)
    img, lab = torch.from_numpy(d['img']), torch.from_numpy(d['label'])
    K, L = 2, 3
    with torch.no_grad():                                         # data-dependent ActNorm init (train_glow.py:60-67)
        G.forward(sd, img, lab, 12, K, L, torch.from_numpy(d['noise/init/0']), True)
    init = gu.state_from_npz(d, 'sd_init/')
    for k, v in init.items():
        if k.endswith(('loc', 'scale', 'initialized')):
            _close(sd[k].float(), v.float(), rtol=1e-3, atol=1e-4, what='init ' + k)
    losses, first = _train(sd, lambda s: G.forward(sd, img, lab, 12, K, L, torch.from_numpy(d[f'noise/{s}/0']), True), 2)
    np.testing.assert_allclose(losses[0], d['losses'][0], rtol=0, atol=1e-4)
    np.testing.assert_allclose(losses, d['losses'], rtol=0, atol=5e-3)
    for i, z in enumerate(first['z']):
        _close(z, d[f'z0/{i}'], rtol=1e-3, atol=1e-4, what=f'z[{i}]')
    # eval forward, exact reconstruction through reverse, sampling from fixed z (on the reference's final weights)
    sdf = gu.state_from_npz(d, 'sd_final/')
    with torch.no_grad():
        out = G.forward(sdf, img, lab, 12, K, L, torch.from_numpy(d['noise/eval/0']), False)
        np.testing.assert_allclose(float(out['loss']), float(d['loss_eval']), rtol=0, atol=1e-4)
        rec = G.reverse(sdf, out['z'], lab, 12, K, L, reconstruct=True)
        _close(rec, d['reconstructed'], rtol=1e-3, atol=1e-3, what='reverse(reconstruct)')
        gz = [torch.from_numpy(d[f'gen_z/{i}']) for i in range(L)]
        _close(G.reverse(sdf, gz, lab, 12, K, L, reconstruct=False), d['generated'], rtol=1e-3, atol=1e-3, what='generate')


def test_vqvae_encode_and_decode_code():
    """The frozen VQ-VAE in front of MCPixelCNN: encoder output, code map, quantised tensor,
    decode_code against the reference-generated fixture.  Codes must agree wherever the reference's own arg-min
    margin is above rounding noise."""
    from oracle import vqvae_oracle as Q
    d = gu.load_npz('vqvae_small.npz')
    sd = gu.state_from_npz(d)
    img = torch.from_numpy(d['img'])
    with torch.no_grad():
        x = Q.encoder(sd, img, 2, 2)
        _close(x, d['enc_out'], what='encoder output')
        q, mse, code, _ = Q.encode(sd, img, 2, 2)
        decisive = torch.from_numpy(d['dist_margin'] > 1e-4).view(code.shape)
        assert torch.equal(code[decisive], torch.from_numpy(d['code'])[decisive])
        assert float(decisive.float().mean()) > 0.9
        np.testing.assert_allclose(float(mse), float(d['vq_loss']), rtol=1e-4)
        _close(Q.decode_code(sd, torch.from_numpy(d['code']), 2, 2), d['decoded'], what='decode_code')
