#!/usr/bin/env python3
"""Element-level comparison of the HIP discriminator-update gradient with autograd through the CPU oracle on a
full-size config (diagnostic, fp32): per tensor the max-norm error AND the error on the small-magnitude elements
(Adam turns every non-zero gradient into a +-lr step, so a wrong tiny gradient moves a weight a full step).
usage: tests/diag/diag_grads.py coil|cifar [batch]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
import golden_util as gu  # noqa: E402
from oracle import mcgan_oracle as O  # noqa: E402


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else 'coil'
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    from mcgen_amd import models
    from mcgen_amd.config import cfg, process_control
    from mcgen_amd.trainer import GANTrainer
    if which == 'coil':
        gh, dh, classes, name, cifar = [512, 256, 128, 64], [64, 128, 256, 512], 100, 'COIL100', False
        sd = gu.procedural_state(gu.mcgan_shapes(gh, dh, classes, cifar_layout=False), seed=4242, num_mode=classes)
        img, lab = gu.synthetic_batch(B, classes, seed=5); zs = gu.latent_batches(6, B, 128, seed=6)
    else:
        gh, dh, classes, name, cifar = [256] * 4, [128] * 4, 10, 'CIFAR10', True
        sd = gu.procedural_state(gu.mcgan_shapes(gh, dh, classes), seed=1234, num_mode=classes)
        img, lab = gu.synthetic_batch(B, classes, seed=1); zs = gu.latent_batches(6, B, 128, seed=2)
    cfg.update(data_name=name, model_name='mcgan', device='cuda'); cfg.pop('classes_size', None)
    process_control(); cfg['classes_size'] = classes
    cfg['gan']['generator_hidden_size'], cfg['gan']['discriminator_hidden_size'] = gh, dh
    m = models.mcgan(); m.load_state_dict(sd); m = m.cuda(); m.train(True)
    tr = GANTrainer(m, classes)
    imgc, labc = img.cuda(), lab.cuda()
    ind2 = F.one_hot(labc, classes).float().repeat(2, 1); ind = ind2[:B]
    fake, _ = tr.geng.forward(zs[0].cuda(), ind, True)
    loss = tr.d_compute(imgc, ind, fake, ind2)
    torch.cuda.synchronize()
    got = {k: tr.deng.flat_p.view_of(tr.grad_d, p).detach().cpu().clone() for k, p in m.discriminator.named_parameters()}
    st = {k: v.detach().clone() for k, v in sd.items()}
    for k in O.trainable_keys(st, 'discriminator.'):
        st[k].requires_grad_(True)
    indc = O.one_hot(lab, classes)
    fk = O.generator_forward(st, zs[0], indc, True).detach()
    ol = torch.relu(1.0 - O.discriminator_forward(st, img, indc, True, cifar_layout=cifar)).mean() \
        + torch.relu(1.0 + O.discriminator_forward(st, fk, indc, True, cifar_layout=cifar)).mean()
    ol.backward()
    print(f'loss hip {float(loss):.7f} oracle {float(ol):.7f}')
    for k, g in got.items():
        r = st['discriminator.' + k].grad
        err = (g - r).abs()
        scale = float(r.abs().max())
        small = r.abs() < 1e-3 * scale
        flips = ((g * r) < 0) & (r.abs() > 1e-8)
        zero_ref = (r == 0)
        print(f'{k:44s} max|ref| {scale:.2e} max err {float(err.max()):.2e} | small elems {int(small.sum()):7d} '
              f'max err there {float(err[small].max()) if small.any() else 0:.2e} | sign flips (|ref|>1e-8) {int(flips.sum()):6d} '
              f'worst |ref| among flips {float(r.abs()[flips].max()) if flips.any() else 0:.2e} | ref==0: {int(zero_ref.sum())} '
              f'hip!=0 there: {int((g[zero_ref] != 0).sum())}')


if __name__ == '__main__':
    main()
