#!/usr/bin/env python3
"""Times the 512 -> 512 1x1 convolution of MCGlow's coupling nets (mcglow.py:148-151) on the fused convolution, with and
without its prologue / statistics epilogue, on the three map sizes.  usage (GPU box): python tools/bench_glow1x1.py"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mcgen_amd import ops

dev = torch.device('cuda')
g = torch.Generator(device=dev).manual_seed(0)
for h in (16, 8, 4):
    n, c = 128, 512
    x = torch.randn(n, h, h, c, device=dev, generator=g).bfloat16()
    w = ops.prep_weight((torch.randn(c, c, 1, 1, device=dev, generator=g) * 0.05), torch.bfloat16)
    b = torch.randn(c, device=dev, generator=g)
    sc, sh = torch.rand(c, device=dev, generator=g) + 0.5, torch.randn(c, device=dev, generator=g) * 0.1
    code = (torch.rand(n, c, device=dev, generator=g) < 0.5).float()
    gx = torch.randn(n, h, h, c, device=dev, generator=g).bfloat16()
    ones = torch.ones(c, device=dev)
    variants = {
        'plain': lambda: ops.conv_fused([ops.Seg(x, ksize=1)], w, c, bias=b),
        'prologue': lambda: ops.conv_fused([ops.Seg(x, ksize=1, scale=sc, shift=sh, relu=True, code=code)], w, c, bias=b),
        'prologue+stats1': lambda: ops.conv_fused([ops.Seg(x, ksize=1, scale=sc, shift=sh, relu=True, code=code)], w, c, bias=b, stats_mode=1),
        'dgrad gate+stats2': lambda: ops.conv_fused([ops.Seg(x, ksize=1)], w, c, ocode=code, gate_x=gx, gscale=sc, gshift=sh, gmean=sh, grstd=ones, stats_mode=2),
    }
    for name, fn in variants.items():
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
        for s, e in ev:
            s.record(); fn(); e.record()
        torch.cuda.synchronize()
        ts = sorted(s.elapsed_time(e) * 1e3 for s, e in ev)
        fl = 2.0 * n * h * h * c * c
        print(f'{h:2d}x{h:<2d} {name:20s} median {ts[10]:7.1f} us   {fl / ts[10] / 1e6:6.0f} TFLOP/s')
