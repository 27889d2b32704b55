#!/usr/bin/env python3
"""Find which parameter gradients of one D / G compute differ between 64x64-tile variants (MCGEN_CONV_SMALL)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import torch.nn.functional as F
import bench
from mcgen_amd.trainer import GANTrainer

dev = torch.device('cuda', 0)
res = {}
for mode in ('5', '11'):
    os.environ['MCGEN_CONV_SMALL'] = mode
    torch.manual_seed(0)
    model, sd = bench.build_model(torch.bfloat16, dev)
    g = torch.Generator(device=dev).manual_seed(1)
    img = torch.rand(128, 3, 32, 32, device=dev, generator=g) * 2 - 1
    lab = torch.randint(0, 10, (128,), device=dev, generator=g)
    z = torch.randn(128, 128, device=dev, generator=g)
    tr = GANTrainer(model, 10)
    model.train(True)
    ind = F.one_hot(lab, 10).float()
    ld = tr.d_compute(img, ind, z)
    gd = tr.grad_d.clone()
    lg = tr.g_compute(ind, z)
    gg = tr.grad_g.clone()
    torch.cuda.synchronize()
    res[mode] = (float(ld), float(lg), gd, gg, tr)
print('losses', res['5'][:2], res['11'][:2])
for name, idx, eng in (('D', 2, 'deng'), ('G', 3, 'geng')):
    a, b = res['5'][idx], res['11'][idx]
    tr = res['11'][4]
    e = getattr(tr, eng)
    fs = e.flat_p
    net = tr.model.discriminator if name == 'D' else tr.model.generator
    for pname, p in net.named_parameters():
        try:
            va, vb = fs.view_of(a, p), fs.view_of(b, p)
        except Exception:
            continue
        d = float((va - vb).abs().max()); s = float(va.abs().max()) + 1e-12
        flag = '  <<<<' if d / s > 2e-2 else ''
        print(f'{name} {pname:50s} rel diff {d / s:.3e}{flag}')
