#!/usr/bin/env python3
"""Microbenchmark of conv_px1.hip (512 -> 512 1x1 with the pixel tile resident) on MCGlow's three levels, forward and
input-gradient form.  Run under rocprofv3 --kernel-trace --stats for kernel durations.  usage: python tools/bench_px1.py [reps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mcgen_amd import ops  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dt = torch.bfloat16
g = torch.Generator(device='cuda').manual_seed(1)
n, c = 128, 512
w = ops.prep_weight(torch.randn(c, c, 1, 1, device='cuda', generator=g) * 0.04, dt)
b = torch.randn(c, device='cuda', generator=g)
for h in (16, 8, 4):
    x = torch.randn(n, h, h, c, device='cuda', generator=g).to(dt)
    gx = torch.randn(n, h, h, c, device='cuda', generator=g).to(dt)
    sc, sh = torch.rand(c, device='cuda', generator=g) + 0.5, torch.randn(c, device='cuda', generator=g) * 0.1
    code = (torch.rand(n, c, device='cuda', generator=g) < 0.5).float()
    one = torch.ones(c, device='cuda')
    fwd = lambda: ops.conv_fused([ops.Seg(x, ksize=1, scale=sc, shift=sh, code=code, relu=True)], w, c, bias=b)
    bwd = lambda: ops.conv_fused([ops.Seg(x, ksize=1)], w, c, ocode=code, gate_x=gx, gscale=sc, gshift=sh, gmean=sh, grstd=one, stats_mode=2)
    for name, fn in (('fwd', fwd), ('bwd', bwd)):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        print(f'{h:2d}x{h:<2d} {name}  {us:7.2f} us/launch back to back  {2 * n * h * h * c * c / us / 1e6:7.1f} TFLOP/s', flush=True)
