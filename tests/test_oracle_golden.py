"""Pins the CPU oracle (oracle/mcgan_oracle.py) to vectors produced by running
the reference itself (tools/gen_golden.py -> tests/golden/*.npz).

Tolerances: the oracle spells BN / spectral norm / pooling out explicitly, so
its summation order differs from torch's fused kernels; fp32 agreement is
expected to ~1e-6 relative on activations, and 2e-5 absolute on step losses
after three full train iterations.
"""
import numpy as np
import pytest
import torch

import golden_util as gu
from oracle import mcgan_oracle as O

torch.set_num_threads(8)


def _close(a, b, rtol=2e-5, atol=2e-6, what=''):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol, err_msg=what)


def _final_state_close(sd, fin, iters, head_bias_key):
    """Conv/linear biases that feed straight into a BatchNorm have a gradient
    that is zero in exact arithmetic; Adam normalises the fp32 rounding noise
    into +-lr steps, so those entries are only comparable to iters * lr."""
    for k in fin:
        noisy = k.startswith('generator.') and k.endswith(
            ('linear.module.bias', 'conv.4.module.bias', 'conv.8.module.bias', 'shortcut.2.module.bias',
             'running_mean'))      # the BN running means absorb those biases
        if noisy:
            _close(sd[k], fin[k], rtol=0, atol=1.5 * iters * 2e-4, what='final(noise-grad) ' + k)
        else:
            _close(sd[k], fin[k], rtol=1e-3, atol=2e-5, what='final ' + k)


def test_mc_unit():
    d = gu.load_npz('mc_unit.npz')
    cb = torch.from_numpy(d['codebook'])
    ind = O.one_hot(torch.from_numpy(d['label']), 10)
    for tag in ('4', '2'):
        x = torch.from_numpy(d['x' + tag]).requires_grad_(True)
        out = O.mc_mask(x, ind, cb)
        out.backward(torch.from_numpy(d['g' + tag]))
        assert np.array_equal(out.detach().numpy(), d['out' + tag])          # bit-exact: one multiply
        assert np.array_equal(x.grad.numpy(), d['dx' + tag])
    _close(O.mc_mask(torch.from_numpy(d['x4']), torch.from_numpy(d['soft']), cb), d['out_soft'], what='soft indicator')
    assert np.all(d['ones_codebook'] == 1.0)
    assert set(np.unique(d['codebook'])) <= {0.0, 1.0}
    assert len({tuple(r) for r in d['codebook'].tolist()}) == 10


BLOCKS = {
    'gen': lambda sd, x, ind, tr: O.gen_res_block(sd, '', x, ind, tr),
    'dis_first': lambda sd, x, ind, tr: O.first_dis_block(sd, '', x, ind, tr),
    'dis_s2': lambda sd, x, ind, tr: (lambda hs: O._pool2(hs[0]) + O._pool2(hs[1]))(O.dis_res_block(sd, '', x, ind, tr)),
    'dis_s1': lambda sd, x, ind, tr: (lambda hs: hs[0] + hs[1])(O.dis_res_block(sd, '', x, ind, tr)),
}


@pytest.mark.parametrize('tag', list(BLOCKS))
def test_blocks(tag):
    d = gu.load_npz('mcgan_blocks.npz')
    sd = gu.state_from_npz(d, f'{tag}/sd0/')
    ind = O.one_hot(torch.from_numpy(d['label']), 10)
    pkeys = [k for k in sd if k.endswith(('weight', 'bias', 'weight_orig'))]
    for k in pkeys:
        sd[k].requires_grad_(True)
    x = torch.from_numpy(d[f'{tag}/x']).requires_grad_(True)
    out = BLOCKS[tag](sd, x, ind, True)
    out.backward(torch.from_numpy(d[f'{tag}/g']))
    _close(out, d[f'{tag}/out'], what='out')
    _close(x.grad, d[f'{tag}/dx'], rtol=1e-4, atol=1e-5, what='dx')
    for k in pkeys:
        _close(sd[k].grad, d[f'{tag}/grad/{k}'], rtol=1e-4, atol=2e-5, what=k)
    ref1 = gu.state_from_npz(d, f'{tag}/sd1/')
    for k in ref1:                                   # BN running stats, SN u/v after one forward
        _close(sd[k], ref1[k], what='sd1 ' + k)
    out2 = BLOCKS[tag](sd, x.detach(), ind, True)
    _close(out2, d[f'{tag}/out_second'], what='second forward')
    ref2 = gu.state_from_npz(d, f'{tag}/sd2/')
    for k in ref2:
        _close(sd[k], ref2[k], what='sd2 ' + k)
    _close(BLOCKS[tag](sd, x.detach(), ind, False), d[f'{tag}/out_eval'], what='eval forward')


def test_mcgan_small_train():
    d = gu.load_npz('mcgan_small.npz')
    m = O.OracleMCGAN(gu.state_from_npz(d), classes=10)
    img, lab = torch.from_numpy(d['img']), torch.from_numpy(d['label'])
    zs = [torch.from_numpy(z) for z in d['z']]
    with torch.no_grad():
        _close(m.generate(lab, zs[-1]), d['probe_generated'], what='probe G')
        _close(m.discriminate(img, lab), d['probe_d_real'], rtol=1e-4, atol=1e-5, what='probe D')
    after = gu.state_from_npz(d, 'sd_after_probe/')
    for k in after:
        _close(m.sd[k], after[k], what='after probe ' + k)
    m = O.OracleMCGAN(gu.state_from_npz(d), classes=10)
    losses = [m.train_iteration(img, lab, zs[6 * i:6 * i + 6]) for i in range(3)]
    np.testing.assert_allclose(np.array(losses), d['losses'], rtol=0, atol=2e-5)
    _final_state_close(m.sd, gu.state_from_npz(d, 'sd_final/'), 3, 'generator.blocks.6.module.bias')
    with torch.no_grad():
        # eval mode reads the running means, which carry the +-lr bias noise (see above)
        _close(m.generate(lab, zs[-1], train=False), d['final_generated_eval'], rtol=1e-3, atol=3e-3)
        _close(m.discriminate(img, lab, train=False), d['final_d_eval'], rtol=1e-3, atol=1e-4)


def test_mcgan_coil_schedule():
    d = gu.load_npz('mcgan_coil_small.npz')
    m = O.OracleMCGAN(gu.state_from_npz(d), classes=20, cifar_layout=False)
    img, lab = torch.from_numpy(d['img']), torch.from_numpy(d['label'])
    zs = [torch.from_numpy(z) for z in d['z']]
    losses = [m.train_iteration(img, lab, zs)]
    np.testing.assert_allclose(np.array(losses), d['losses'], rtol=0, atol=2e-5)
    _final_state_close(m.sd, gu.state_from_npz(d, 'sd_final/'), 1, 'generator.blocks.6.module.bias')


def test_mcgan_full_digest():
    """Full-size model (G [256]*4, D [128]*4), procedural weights, B=16."""
    d = gu.load_npz('mcgan_full_digest.npz')
    shapes = gu.mcgan_shapes([256] * 4, [128] * 4, 10)
    n_params = sum(int(np.prod(s)) for k, s in shapes.items()
                   if k.endswith(('.weight', '.bias', '.weight_orig')))
    assert n_params == 5330564                     # MCGAN param count
    sd = gu.procedural_state(shapes, seed=1234, num_mode=10)
    m = O.OracleMCGAN(sd, classes=10)
    img, lab = gu.synthetic_batch(16, 10, seed=1)
    zs = gu.latent_batches(12, 16, 128, seed=2)
    with torch.no_grad():
        gen0 = m.generate(lab, zs[0])
        _close(gen0[:, :, ::4, ::4], d['probe_generated'], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(gu.checksum(gen0), d['probe_generated_digest'], rtol=1e-4, atol=1e-3)
        _close(m.discriminate(img, lab), d['probe_d_real'], rtol=1e-4, atol=1e-4)
    m = O.OracleMCGAN(sd, classes=10)
    losses = [m.train_iteration(img, lab, zs[0:6]), m.train_iteration(img, lab, zs[6:12])]
    # iteration 1 is a pure function of the inputs; from iteration 2 on every
    # parameter has taken an Adam step of +-lr * g/|g|, so entries whose gradient
    # is below fp32 rounding noise moved in an implementation-defined direction.
    np.testing.assert_allclose(np.array(losses[0]), d['losses'][0], rtol=0, atol=5e-5)
    np.testing.assert_allclose(np.array(losses[1]), d['losses'][1], rtol=0, atol=2e-3)
    for k, v in d.items():
        if k.startswith('digest/'):
            np.testing.assert_allclose(gu.checksum(m.sd[k[7:]]), v, rtol=1e-3, atol=1e-3, err_msg=k)
