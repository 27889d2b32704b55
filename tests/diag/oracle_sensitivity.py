#!/usr/bin/env python3
"""CPU-only probe for DESIGN.md section 2's open item: how far does the COIL100 full-width G loss of the pinned CPU
oracle move when the discriminator gradients of its five D updates carry errors of the size the HIP path's fp32
gradients show against it (tests/diag/diag_grads.py: absolute errors up to ~1e-8 on elements of magnitude <= 1e-5, relative
~1e-6 elsewhere)?  Adam turns a gradient element of size comparable to that error into a step of up to lr with a
different sign.  usage: python tests/diag/oracle_sensitivity.py [abs_err] [rel_err] [seeds]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402
import golden_util as gu  # noqa: E402
from oracle import mcgan_oracle as O  # noqa: E402


def run(abs_err, rel_err, seed):
    gh, dh, classes = [512, 256, 128, 64], [64, 128, 256, 512], 100
    sd = gu.procedural_state(gu.mcgan_shapes(gh, dh, classes, cifar_layout=False), seed=4242, num_mode=classes)
    img, lab = gu.synthetic_batch(8, classes, seed=5)
    zs = gu.latent_batches(6, 8, 128, seed=6)
    orc = O.OracleMCGAN(sd, classes=classes, cifar_layout=False)
    g = torch.Generator().manual_seed(seed)
    step = orc.opt_d.step

    def noisy_step():
        if abs_err or rel_err:
            for k in orc.dkeys:
                p = orc.sd[k]
                if p.grad is not None:
                    n = torch.randn(p.grad.shape, generator=g)
                    p.grad.add_(n * (abs_err + rel_err * p.grad.abs()))
        return step()
    orc.opt_d.step = noisy_step
    return orc.train_iteration(img, lab, zs)


if __name__ == '__main__':
    abs_err = float(sys.argv[1]) if len(sys.argv) > 1 else 5e-9
    rel_err = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-6
    seeds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    base = run(0.0, 0.0, 0)
    print(f'exact gradients:      D {base[0]:.7f}  G {base[1]:.7f}')
    for s in range(seeds):
        r = run(abs_err, rel_err, 100 + s)
        print(f'abs {abs_err:g} rel {rel_err:g} seed {s}: D {r[0]:.7f} ({r[0] - base[0]:+.2e})  G {r[1]:.7f} ({r[1] - base[1]:+.2e})')
