#!/usr/bin/env python3
"""Which ReLU gates does the HIP discriminator forward decide differently from the CPU oracle on identical weights?
(diagnostic, fp32).  The oracle takes `nupd` discriminator updates, its state is loaded into the HIP model, and every
pre-ReLU tensor of one forward over the real batch is compared element by element.
usage: tests/diag/diag_gates.py coil|cifar [batch] [nupd]"""
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
    nupd = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    from mcgen_amd import models, ops
    from mcgen_amd.config import cfg, process_control
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
    orc = O.OracleMCGAN(sd, classes=classes, cifar_layout=cifar)
    for k in range(nupd):
        orc._zero()
        d_x = orc.discriminate(img, lab)
        fk = orc.generate(lab, zs[k])
        d_g = orc.discriminate(fk.detach(), lab)
        (torch.relu(1.0 - d_x).mean() + torch.relu(1.0 + d_g).mean()).backward()
        orc.opt_d.step()
    state = {k: v.detach().clone() for k, v in orc.sd.items()}
    m = models.mcgan(); m.load_state_dict(state); m = m.cuda(); m.train(True)
    ind = F.one_hot(lab, classes).float()
    for tag, xin in (('real', img), ('fake', orc.generate(lab, zs[nupd]).detach())):
        m.load_state_dict(state)
        deng = m.discriminator._engine()
        with torch.no_grad():
            logit, ctx = deng.forward(xin.cuda(), ind.cuda(), True)
        hip = {'b0.c1': ctx['blocks'][0]['c1']}
        for j in range(1, len(ctx['blocks'])):
            hip[f'b{j}.x'] = ctx['blocks'][j]['x']; hip[f'b{j}.c1'] = ctx['blocks'][j]['c1']
        hip['tail.x'] = ctx['xt']
        hip = {k: ops.to_nchw(v, v.shape[-1]).cpu() for k, v in hip.items()}
        # oracle, block by block (mcgan.py:88-93, 101-138, 155-176), on a copy of the same state
        st = {k: v.clone() for k, v in state.items()}
        p = 'discriminator.'
        with torch.no_grad():
            ref = {}
            x = xin
            c1 = O._conv(st, p + 'blocks.0.conv.0.module', x, 1, True, True)
            ref['b0.c1'] = c1
            h = O._conv(st, p + 'blocks.0.conv.3.module', O.mc_mask(torch.relu(c1), ind, st[p + 'blocks.0.mc_1.codebook']), 1, True, True)
            s = O._conv(st, p + 'blocks.0.shortcut.0.module', x, 0, True, True)
            x = O._pool2(h) + O._pool2(s)
            nres = len(ctx['blocks']) - 1
            n_stride1 = 2 if cifar else 1
            for j in range(1, nres + 1):
                pj = p + f'blocks.{j}.'
                cb1, cb2 = st[pj + 'mc_1.codebook'], st[pj + 'mc_2.codebook']
                ref[f'b{j}.x'] = x
                s = O._conv(st, pj + 'shortcut.1.module', O.mc_mask(x, ind, cb1), 0, True, True) if (pj + 'shortcut.1.module.weight_orig') in st else x
                c1 = O._conv(st, pj + 'conv.2.module', O.mc_mask(torch.relu(x), ind, cb1), 1, True, True)
                ref[f'b{j}.c1'] = c1
                h = O._conv(st, pj + 'conv.5.module', O.mc_mask(torch.relu(c1), ind, cb2), 1, True, True)
                x = (O._pool2(h) + O._pool2(s)) if j <= nres - n_stride1 else (h + s)
            ref['tail.x'] = x
        print(f'--- {tag} batch after {nupd} oracle updates: logit max diff {float((logit.cpu().view(-1) - O.discriminator_forward({k: v.clone() for k, v in state.items()}, xin, ind, True, cifar_layout=cifar).view(-1)).abs().max()):.2e}')
        for k in ref:
            a, b = hip[k], ref[k]
            dis = (a > 0) != (b > 0)
            err = (a - b).abs()
            print(f'  {k:8s} shape {tuple(b.shape)} max|ref| {float(b.abs().max()):.3f} max err {float(err.max()):.2e} '
                  f'min |ref| {float(b.abs().min()):.2e}  gate disagreements {int(dis.sum())}')
            for idx in dis.nonzero()[:4]:
                i = tuple(int(t) for t in idx)
                print(f'      at {i}: hip {float(a[i]):+.3e} oracle {float(b[i]):+.3e}')


if __name__ == '__main__':
    main()
