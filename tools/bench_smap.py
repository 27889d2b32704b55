#!/usr/bin/env python3
"""Microbenchmark of conv_smap.hip (3x3, 128 -> 128, 8x8 maps, one image per workgroup) against the general 64x64 tile on
the same shape (made ineligible by asking for BatchNorm statistics).  Run under rocprofv3 --kernel-trace --stats for the
kernels' own durations; prints HIP-event averages.  usage (GPU box): python tools/bench_smap.py [reps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mcgen_amd import ops  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dt = torch.bfloat16
g = torch.Generator(device='cuda').manual_seed(1)


def timeit(name, fn, flops):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print(f'{name:58s} {us:7.2f} us/launch (back to back, incl. launch)  {flops / us / 1e6:7.1f} TFLOP/s', flush=True)


for n in (256, 128):
    x = torch.randn(n, 8, 8, 128, device='cuda', generator=g).to(dt)
    r = torch.randn(n, 8, 8, 128, device='cuda', generator=g).to(dt)
    code = (torch.rand(n, 128, device='cuda', generator=g) < 0.5).float()
    w = ops.prep_weight(torch.randn(128, 128, 3, 3, device='cuda', generator=g) * 0.03, dt)
    b = torch.randn(128, device='cuda', generator=g)
    seg = ops.Seg(x, code=code, relu=True)
    fl = 2 * n * 64 * 128 * 1152
    timeit(f'N={n} 128k3->128 fwd (relu, code, bias, res)', lambda: ops.conv_fused([seg], w, 128, bias=b, res=r), fl)
    timeit(f'N={n} 128k3->128 bwd (ocode, gate)', lambda: ops.conv_fused([ops.Seg(x)], w, 128, ocode=code, gate_x=r), fl)
    timeit(f'N={n} 128k3->128 general tile (pooled output)', lambda: ops.conv_fused([seg], w, 128, bias=b, pool=True, alpha=0.25), fl)
# MCGatedPixelCNN's layer shapes (N = 128)
n = 128
for segs, co, st in (([(128, 3)], 256, 1), ([(256, 1), (128, 3)], 256, 1), ([(256, 3)], 128, 0), ([(256, 1)], 256, 0), ([(128, 1)], 128, 1)):
    sg, ws, k = [], [], 0
    for ci, ks in segs:
        x = torch.randn(n, 8, 8, ci, device='cuda', generator=g).to(dt)
        code = (torch.rand(n, ci, device='cuda', generator=g) < 0.5).float()
        sg.append(ops.Seg(x, ksize=ks, code=code, relu=True))
        ws.append(ops.prep_weight(torch.randn(co, ci, ks, ks, device='cuda', generator=g) * 0.03, dt))
        k += ci * ks * ks
    w = torch.cat(ws)
    b = torch.randn(co, device='cuda', generator=g)
    name = '+'.join(f'{ci}k{ks}' for ci, ks in segs) + f'->{co}' + ('s1' if st else '')
    timeit(f'N={n} {name}', lambda: ops.conv_fused(sg, w, co, bias=b, stats_mode=st), 2 * n * 64 * co * k)
