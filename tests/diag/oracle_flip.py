#!/usr/bin/env python3
"""CPU-only: DESIGN.md section 2, COIL100 full-width item.  tests/diag/diag_elem.py shows that in the FIRST discriminator update
the gradient of `discriminator.blocks.3.conv.5.module.bias[146]` (and of the shortcut bias that shares it) is a pure
cancellation residue: +1.49e-8 (= 2^-26) in the oracle / reference, -1.49e-8 on the HIP path; element 58 is exactly 0 in
the oracle and +3.7e-9 on the HIP path.  Adam turns either into a step of about lr/2 with the residue's sign.  This script
runs the pinned oracle with exactly those two residues replaced by the HIP path's values and prints the iteration's
losses: if the generator loss lands on the HIP path's value, the difference between the two is this rounding residue
amplified by Adam and a ReLU boundary -- not a difference in the computation."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402
import golden_util as gu  # noqa: E402
from oracle import mcgan_oracle as O  # noqa: E402


def run(patch):
    gh, dh, classes = [512, 256, 128, 64], [64, 128, 256, 512], 100
    sd = gu.procedural_state(gu.mcgan_shapes(gh, dh, classes, cifar_layout=False), seed=4242, num_mode=classes)
    img, lab = gu.synthetic_batch(8, classes, seed=5)
    zs = gu.latent_batches(6, 8, 128, seed=6)
    orc = O.OracleMCGAN(sd, classes=classes, cifar_layout=False)
    step, calls = orc.opt_d.step, [0]

    def patched_step():
        if patch and calls[0] == 0:
            for key in ('discriminator.blocks.3.conv.5.module.bias', 'discriminator.blocks.3.shortcut.1.module.bias'):
                g = orc.sd[key].grad
                print(f'  {key}: oracle residues [146] {float(g[146]):+.3e} [58] {float(g[58]):+.3e} -> HIP values')
                g[146] = -1.490116e-08
                g[58] = +3.725290e-09
        calls[0] += 1
        return step()
    orc.opt_d.step = patched_step
    return orc.train_iteration(img, lab, zs)


if __name__ == '__main__':
    a = run(False)
    print(f'oracle:                          D {a[0]:.7f}  G {a[1]:.7f}   (reference fixture: 1.6796330 -0.1955019)')
    b = run(True)
    print(f'oracle with the two HIP residues: D {b[0]:.7f}  G {b[1]:.7f}   (HIP path, fp32:     1.6796678 -0.1806532)')
