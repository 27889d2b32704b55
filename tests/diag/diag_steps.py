#!/usr/bin/env python3
"""Step-by-step comparison of the HIP trainer with the CPU oracle on a full-size config (diagnostic, fp32):
per discriminator update the hinge loss and the mean logits, after the updates the largest parameter differences,
then the generator loss.  usage: tests/diag/diag_steps.py coil|cifar [batch]"""
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
    orc = O.OracleMCGAN(sd, classes=classes, cifar_layout=cifar)
    imgc, labc = img.cuda(), lab.cuda()
    ind2 = F.one_hot(labc, classes).float().repeat(2, 1); ind = ind2[:B]

    def cmp_state(tag):
        sdg = m.state_dict()
        worst = []
        for k, v in orc.sd.items():
            if v.dtype.is_floating_point and 'codebook' not in k:
                dlt = float((sdg[k].cpu() - v.detach()).abs().max())
                worst.append((dlt, k))
        worst.sort(reverse=True)
        print(f'  [{tag}] largest parameter / buffer differences:', [(f'{a:.2e}', k) for a, k in worst[:6]])

    for k in range(5):
        fake, _ = tr.geng.forward(zs[k].cuda(), ind, True)
        loss = tr.d_update(imgc, ind, fake, ind2)
        orc._zero()
        d_x = orc.discriminate(img, lab)
        fk = orc.generate(lab, zs[k])
        d_g = orc.discriminate(fk.detach(), lab)
        ol = torch.relu(1.0 - d_x).mean() + torch.relu(1.0 + d_g).mean()
        ol.backward(); orc.opt_d.step()
        print(f'D update {k}: hip {float(loss):.7f} oracle {float(ol):.7f} diff {float(loss) - float(ol):+.2e}; '
              f'fake batch max diff {float((fake.cpu() - fk.detach()).abs().max()):.2e}')
        cmp_state(f'after D update {k}')
    # the generator loss on the HIP weights, evaluated by the ORACLE: separates "the weights drifted" from "the G-step
    # forward differs"
    hip_sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    with torch.no_grad():
        indc = O.one_hot(lab, classes)
        fk_h = O.generator_forward({k: v.clone() for k, v in hip_sd.items()}, zs[5], indc, True)
        og_h = -O.discriminator_forward({k: v.clone() for k, v in hip_sd.items()}, fk_h, indc, True, cifar_layout=cifar).mean()
        fk_o = O.generator_forward({k: v.detach().clone() for k, v in orc.sd.items()}, zs[5], indc, True)
        og_o = -O.discriminator_forward({k: v.detach().clone() for k, v in orc.sd.items()}, fk_o, indc, True, cifar_layout=cifar).mean()
        # mixed: oracle G weights with HIP D weights and vice versa
        mix1 = {k: (hip_sd[k] if k.startswith('discriminator.') else orc.sd[k].detach()).clone() for k in hip_sd}
        og_m1 = -O.discriminator_forward(mix1, O.generator_forward(mix1, zs[5], indc, True), indc, True, cifar_layout=cifar).mean()
    print(f'oracle forward of the G loss: on HIP weights {float(og_h):.7f}, on oracle weights {float(og_o):.7f}, '
          f'HIP D + oracle G {float(og_m1):.7f}')
    for key in ('weight_u', 'weight_v'):
        worst = max((float((hip_sd[k] - orc.sd[k].detach()).abs().max()), k) for k in hip_sd if k.endswith(key))
        print(f'  largest {key} difference: {worst}')
    gl = tr.g_update(ind, zs[5].cuda())
    orc._zero()
    fk = orc.generate(lab, zs[5])
    og = -orc.discriminate(fk, lab).mean()
    og.backward(); orc.opt_g.step()
    print(f'G update: hip {float(gl):.7f} oracle {float(og):.7f} diff {float(gl) - float(og):+.2e}')
    cmp_state('after G update')


if __name__ == '__main__':
    main()
