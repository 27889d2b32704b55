#!/usr/bin/env python3
"""Layer-by-layer comparison of one generator forward between 64x64-tile variants (MCGEN_CONV_SMALL)."""
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
    model, sd = bench.build_model(torch.bfloat16, dev)
    g = torch.Generator(device=dev).manual_seed(1)
    lab = torch.randint(0, 10, (128,), device=dev, generator=g)
    z = torch.randn(128, 128, device=dev, generator=g)
    tr = GANTrainer(model, 10)
    model.train(True)
    out, ctx = tr.geng.forward(z, F.one_hot(lab, 10).float(), True)
    torch.cuda.synchronize()
    res[mode] = (out, ctx)


def walk(a, b, path=''):
    if torch.is_tensor(a):
        d = float((a.float() - b.float()).abs().max()); s = float(a.float().abs().max()) + 1e-12
        print(f'{path:40s} {tuple(a.shape)!s:28s} rel diff {d / s:.3e}')
    elif isinstance(a, dict):
        for k in a:
            walk(a[k], b[k], path + '.' + str(k))
    elif isinstance(a, (list, tuple)):
        for i, (x, y) in enumerate(zip(a, b)):
            walk(x, y, path + f'[{i}]')
    elif hasattr(a, '__dict__'):
        for k, v in vars(a).items():
            if torch.is_tensor(v):
                walk(v, getattr(b, k), path + '.' + k)


walk(res['5'][0], res['11'][0], 'out')
walk(res['5'][1], res['11'][1], 'ctx')
