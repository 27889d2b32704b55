#!/usr/bin/env python3
"""Latency probe for small fused-conv launches: per-launch GPU time inside a HIP graph (no host launch floor)
as a function of Cin (K chunks), Cout, spatial size and epilogue options."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mcgen_amd import ops
from mcgen_amd.ops import Seg

dt = torch.bfloat16
N = 128


def probe(h, cin, cout, ks=3, code=True, res=True, reps=40):
    x = torch.randn(N, h, h, cin, device='cuda').to(dt)
    w = ops.prep_weight(torch.randn(cout, cin, ks, ks, device='cuda') * 0.05, dt)
    cd = (torch.rand(N, cin, device='cuda') < 0.5).float() if code else None
    r = torch.randn(N, h, h, ops.pad8(cout), device='cuda').to(dt) if res else None
    b = torch.randn(cout, device='cuda')
    y = torch.empty(N, h, h, ops.pad8(cout), device='cuda', dtype=dt)
    f = lambda: ops.conv_fused([Seg(x, ksize=ks, code=cd, relu=code)], w, cout, bias=b, res=r, out=y)
    f(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            f()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (5 * reps)
    fl = 2.0 * N * h * h * cout * cin * ks * ks
    print(f'h={h:2d} cin={cin:4d} cout={cout:4d} k={ks} code={int(code)} res={int(res)}: {us:7.2f} us  {fl / us * 1e-6:7.1f} TFLOP/s', flush=True)


if os.environ.get('PROBE_1X1'):
    probe(8, 512, 512, ks=1, res=False); probe(4, 512, 512, ks=1, res=False); probe(8, 128, 128, ks=1); probe(8, 128, 512, ks=1); probe(8, 512, 128, ks=1)
    sys.exit(0)
if os.environ.get('PROBE_SKINNY'):
    probe(16, 512, 8); probe(8, 512, 16); probe(4, 512, 16); probe(16, 512, 16, code=True, res=False)
    sys.exit(0)
if os.environ.get('PROBE_SMALL'):
    probe(8, 128, 128); probe(8, 256, 256); probe(4, 256, 256)
    sys.exit(0)
for cin in (32, 64, 128, 256):
    probe(8, cin, 128)
probe(8, 128, 128, code=False, res=False)
probe(8, 128, 128, ks=1)
probe(8, 128, 64)
probe(8, 128, 256)
probe(4, 256, 256)
probe(16, 128, 128)
probe(16, 32, 128)
probe(32, 128, 128)
probe(32, 32, 128)
