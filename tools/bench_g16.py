#!/usr/bin/env python3
"""The grouped generator pass's 16x16 launches (640 images: 640 tiles of 256 pixels = 2.5 per CU) under tile overrides
(tuning build: MCGEN_CONV_CFG=bm,bn,pipe).  usage (GPU box): MCGEN_TUNING=1 python tools/bench_g16.py 256,256,20 128,256,20"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mcgen_amd import ops
from mcgen_amd.ops import Seg

dt, dev = torch.bfloat16, 'cuda'
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: torch.randn(*s, device=dev, generator=g)
n, c = 640, 256
x8 = rnd(n, 8, 8, c).to(dt); h16 = rnd(n, 16, 16, c).to(dt)
sc, sh = rnd(5, c), rnd(5, c); code = (torch.rand(n, c, device=dev, generator=g) < 0.5).float()
b = rnd(c)
w1 = ops.prep_weight(rnd(c, c, 3, 3) * 0.05, dt)
w2 = torch.cat([ops.prep_weight(rnd(c, c, 3, 3) * 0.05, dt), ops.prep_weight(rnd(c, c, 1, 1) * 0.05, dt)])
cases = {
    'N640 16x16 256k3->256 s1 (ups)': (lambda: ops.conv_fused([Seg(x8, scale=sc, shift=sh, code=code, ups=True, relu=True, group_n=128)], w1, c, bias=b, stats_mode=1), 2.0 * n * 256 * c * c * 9),
    'N640 16x16 256k3+256k1->256 s1': (lambda: ops.conv_fused([Seg(h16, scale=sc, shift=sh, code=code, relu=True, group_n=128), Seg(x8, ksize=1, code=code, ups=True)], w2, c, bias=b, stats_mode=1), 2.0 * n * 256 * c * c * 10),
}
for cfg in sys.argv[1:] or ['256,256,20']:
    os.environ['MCGEN_CONV_CFG'] = cfg
    for name, (fn, flops) in cases.items():
        try:
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(20):
                fn()
            e.record(); torch.cuda.synchronize()
            us = s.elapsed_time(e) * 50
            print(f'{cfg:12s} {name:34s} {us:8.1f} us  {flops / us / 1e6:7.0f} TFLOP/s')
        except Exception as ex:
            print(f'{cfg:12s} {name:34s} failed: {str(ex)[:90]}')
