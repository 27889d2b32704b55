#!/usr/bin/env python3
"""One parameter element through the first discriminator update, HIP vs oracle (diagnostic, fp32).
usage: tests/diag/diag_elem.py <state_dict key under discriminator.> <index> [<index> ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
import golden_util as gu  # noqa: E402
from oracle import mcgan_oracle as O  # noqa: E402


def main():
    key = sys.argv[1]; idx = [int(a) for a in sys.argv[2:]]
    from mcgen_amd import models
    from mcgen_amd.config import cfg, process_control
    from mcgen_amd.trainer import GANTrainer
    gh, dh, classes, name = [512, 256, 128, 64], [64, 128, 256, 512], 100, 'COIL100'
    sd = gu.procedural_state(gu.mcgan_shapes(gh, dh, classes, cifar_layout=False), seed=4242, num_mode=classes)
    img, lab = gu.synthetic_batch(8, classes, seed=5); zs = gu.latent_batches(6, 8, 128, seed=6)
    cfg.update(data_name=name, model_name='mcgan', device='cuda'); cfg.pop('classes_size', None)
    process_control(); cfg['classes_size'] = classes
    cfg['gan']['generator_hidden_size'], cfg['gan']['discriminator_hidden_size'] = gh, dh
    m = models.mcgan(); m.load_state_dict(sd); m = m.cuda(); m.train(True)
    tr = GANTrainer(m, classes)
    orc = O.OracleMCGAN(sd, classes=classes, cifar_layout=False)
    imgc, labc = img.cuda(), lab.cuda()
    ind2 = F.one_hot(labc, classes).float().repeat(2, 1); ind = ind2[:8]
    p_h = dict(m.discriminator.named_parameters())[key]
    p_o = orc.sd['discriminator.' + key]
    for k in range(2):
        before_h = p_h.detach().view(-1)[idx].cpu().clone(); before_o = p_o.detach().view(-1)[idx].clone()
        fake, _ = tr.geng.forward(zs[k].cuda(), ind, True)
        tr.d_update(imgc, ind, fake, ind2)
        g_h = tr.deng.flat_p.view_of(tr.grad_d, p_h).detach().view(-1)[idx].cpu()
        orc._zero()
        ol = torch.relu(1.0 - orc.discriminate(img, lab)).mean() + torch.relu(1.0 + orc.discriminate(orc.generate(lab, zs[k]).detach(), lab)).mean()
        ol.backward(); g_o = p_o.grad.view(-1)[idx].clone(); orc.opt_d.step()
        after_h = p_h.detach().view(-1)[idx].cpu(); after_o = p_o.detach().view(-1)[idx]
        for j, i in enumerate(idx):
            print(f'update {k} [{i}]: grad hip {float(g_h[j]):+.6e} oracle {float(g_o[j]):+.6e} | param before hip {float(before_h[j]):+.8e} oracle {float(before_o[j]):+.8e} '
                  f'| step hip {float(after_h[j] - before_h[j]):+.4e} oracle {float(after_o[j] - before_o[j]):+.4e}')


if __name__ == '__main__':
    main()
