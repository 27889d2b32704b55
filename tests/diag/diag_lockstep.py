#!/usr/bin/env python3
"""Lock-step diagnostic (fp32): before every discriminator update the HIP model is loaded with the ORACLE's current
state, so each update's HIP gradient is compared with autograd on identical weights / u / v; then both take their own
Adam step (FusedAdam state vs torch.optim.Adam state are compared too).  usage: tests/diag/diag_lockstep.py coil|cifar [batch]"""
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
    mode = sys.argv[3] if len(sys.argv) > 3 else 'pair'
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
    from mcgen_amd import trainer as T, gan_engine as GE
    if mode == 'nopair':
        T._PAIR_D = False
    if mode == 'nopairwgrad':
        GE._PAIR_WGRAD = False
    print('mode', mode)
    tr = GANTrainer(m, classes)
    orc = O.OracleMCGAN(sd, classes=classes, cifar_layout=cifar)
    imgc, labc = img.cuda(), lab.cuda()
    ind2 = F.one_hot(labc, classes).float().repeat(2, 1); ind = ind2[:B]
    dparams = dict(m.discriminator.named_parameters())
    for k in range(5):
        with torch.no_grad():
            m.load_state_dict({kk: v.detach().clone() for kk, v in orc.sd.items()})
        tr.deng._ensure_flat()
        fake, _ = tr.geng.forward(zs[k].cuda(), ind, True)
        loss = tr.d_compute(imgc, ind, fake, ind2)
        torch.cuda.synchronize()
        got = {n: tr.deng.flat_p.view_of(tr.grad_d, p).detach().cpu().clone() for n, p in dparams.items()}
        orc._zero()
        d_x = orc.discriminate(img, lab)
        fk = orc.generate(lab, zs[k])
        d_g = orc.discriminate(fk.detach(), lab)
        ol = torch.relu(1.0 - d_x).mean() + torch.relu(1.0 + d_g).mean()
        ol.backward()
        worst = []
        for n, g in got.items():
            r = orc.sd['discriminator.' + n].grad
            err = float((g - r).abs().max()); scale = float(r.abs().max()) + 1e-30
            flips = int((((g * r) < 0) & (r.abs() > 1e-7)).sum())
            worst.append((err / scale, err, scale, flips, n))
        worst.sort(reverse=True)
        print(f'update {k}: loss hip {float(loss):.7f} oracle {float(ol):.7f}; active hinge terms real {int((1 - d_x > 0).sum())} fake {int((1 + d_g > 0).sum())}')
        for rel, err, scale, flips, n in worst[:5]:
            print(f'    {n:44s} rel err {rel:.2e} (abs {err:.2e} of {scale:.2e}) sign flips with |ref|>1e-7: {flips}')
        # Adam: both sides step from identical weights; compare the step taken
        before = {n: p.detach().cpu().clone() for n, p in dparams.items()}
        tr.d_apply()
        orc.opt_d.step()
        torch.cuda.synchronize()
        wd = []
        for n, p in dparams.items():
            step_h = p.detach().cpu() - before[n]
            step_o = orc.sd['discriminator.' + n].detach() - before[n]
            wd.append((float((step_h - step_o).abs().max()), n))
        wd.sort(reverse=True)
        print('    largest Adam-step differences:', [(f'{a:.2e}', n) for a, n in wd[:3]])
        # FusedAdam state must follow torch's: load the oracle's optimizer state for the next round
        tr.opt_d.load_state_dict(orc.opt_d.state_dict())


if __name__ == '__main__':
    main()
