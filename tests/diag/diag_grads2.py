#!/usr/bin/env python3
"""As tests/diag/diag_grads.py, but for the SECOND (and later) discriminator updates of an iteration: both sides first run
`k` whole D updates (their states then differ by Adam-level amounts only), then the gradients of update k are compared
element by element.  usage: tests/diag/diag_grads2.py coil|cifar [batch] [k]"""
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
    K = int(sys.argv[3]) if len(sys.argv) > 3 else 1
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
    orc = O.OracleMCGAN(sd, classes=classes, cifar_layout=cifar)
    imgc, labc = img.cuda(), lab.cuda()
    ind2 = F.one_hot(labc, classes).float().repeat(2, 1); ind = ind2[:B]
    for k in range(K + 1):
        fake, _ = tr.geng.forward(zs[k].cuda(), ind, True)
        if k < K:
            tr.d_update(imgc, ind, fake, ind2)
        else:
            loss = tr.d_compute(imgc, ind, fake, ind2)
        orc._zero()
        d_x = orc.discriminate(img, lab)
        fk = orc.generate(lab, zs[k])
        d_g = orc.discriminate(fk.detach(), lab)
        ol = torch.relu(1.0 - d_x).mean() + torch.relu(1.0 + d_g).mean()
        ol.backward()
        if k < K:
            orc.opt_d.step()
    torch.cuda.synchronize()
    print(f'update {K}: loss hip {float(loss):.7f} oracle {float(ol):.7f}')
    got = {k: tr.deng.flat_p.view_of(tr.grad_d, p).detach().cpu().clone() for k, p in m.discriminator.named_parameters()}
    sdg = m.state_dict()
    for k, g in got.items():
        r = orc.sd['discriminator.' + k].grad
        pd = (sdg['discriminator.' + k].cpu() - orc.sd['discriminator.' + k].detach()).abs()
        err = (g - r).abs()
        scale = float(r.abs().max())
        flips = ((g * r) < 0)
        big = flips & (r.abs() > 1e-6 * max(scale, 1e-30))
        print(f'{k:40s} max|ref| {scale:.2e} max err {float(err.max()):.2e} rel {float(err.max()) / max(scale, 1e-30):.1e} | param diff before the '
              f'update: max {float(pd.max()):.2e} | sign flips {int(flips.sum()):6d}, with |ref| > 1e-6 max: {int(big.sum()):5d} '
              f'(worst |ref| {float(r.abs()[flips].max()) if flips.any() else 0:.2e})')
        if k.endswith('bias') and float(err.max()) > 1e-3 * max(scale, 1e-30):
            top = torch.topk(err.view(-1), min(6, err.numel())).indices.tolist()
            print('      worst elements (index, ref grad, hip grad, param diff before):',
                  [(i, f'{float(r.view(-1)[i]):+.3e}', f'{float(g.view(-1)[i]):+.3e}',
                    f'{float((sdg["discriminator." + k].cpu() - orc.sd["discriminator." + k].detach()).view(-1)[i]):+.2e}') for i in top])


if __name__ == '__main__':
    main()
