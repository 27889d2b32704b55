#!/usr/bin/env python3
"""DESIGN.md section 2, COIL100 full-width item: is the generator-loss residual Adam amplifying rounding residues?
Several discriminator gradients in this case are pure cancellation residues (the true value is 0: the real and fake halves
contribute +-w/N times an equal count of live pixels), so they come out 0 or +-2^-26 depending on summation order, and
Adam(eps 1e-8) turns a +-1.5e-8 gradient into a +-1.2e-4 step (tests/diag/diag_elem.py).  This runs the SAME iteration on the
HIP path (fp32) and on the oracle for a sweep of Adam eps: the arithmetic of every kernel is unchanged, only the
optimiser's amplification of |g| ~ 1e-8 is switched off as eps grows.  Diagnostic only (needs a GPU)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402
import golden_util as gu  # noqa: E402
from oracle import mcgan_oracle as O  # noqa: E402


def main():
    from mcgen_amd import models
    from mcgen_amd.config import cfg, process_control
    from mcgen_amd.trainer import GANTrainer
    gh, dh, classes, name = [512, 256, 128, 64], [64, 128, 256, 512], 100, 'COIL100'
    sd = gu.procedural_state(gu.mcgan_shapes(gh, dh, classes, cifar_layout=False), seed=4242, num_mode=classes)
    img, lab = gu.synthetic_batch(8, classes, seed=5); zs = gu.latent_batches(6, 8, 128, seed=6)
    cfg.update(data_name=name, model_name='mcgan', device='cuda'); cfg.pop('classes_size', None)
    process_control(); cfg['classes_size'] = classes
    cfg['gan']['generator_hidden_size'], cfg['gan']['discriminator_hidden_size'] = gh, dh
    for eps in [float(a) for a in sys.argv[1:]] or [1e-8, 1e-6, 1e-4, 1e-2]:
        m = models.mcgan(); m.load_state_dict(sd); m = m.cuda(); m.train(True)
        tr = GANTrainer(m, classes)
        tr.opt_d.eps = tr.opt_g.eps = eps
        d_h, g_h = tr.train_iteration(img.cuda(), lab.cuda(), [z.cuda() for z in zs])
        orc = O.OracleMCGAN(sd, classes=classes, cifar_layout=False)
        for o in (orc.opt_d, orc.opt_g):
            for grp in o.param_groups:
                grp['eps'] = eps
        d_o, g_o = orc.train_iteration(img, lab, zs)
        print(f'adam eps {eps:.0e}: D hip {float(d_h):.7f} oracle {float(d_o):.7f} diff {float(d_h) - float(d_o):+.2e} | '
              f'G hip {float(g_h):.7f} oracle {float(g_o):.7f} diff {float(g_h) - float(g_o):+.2e}', flush=True)


if __name__ == '__main__':
    main()
