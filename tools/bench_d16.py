#!/usr/bin/env python3
"""The discriminator's 16x16 layers (128 -> 128, 2N = 256 images: exactly one 256 x 128 tile per CU) under tile overrides
(tuning build: MCGEN_CONV_CFG=bm,bn,pipe).  usage (GPU box): MCGEN_TUNING=1 python tools/bench_d16.py 256,128,20 128,128,5 ..."""
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
n, c = 256, 128
x = rnd(n, 16, 16, c).to(dt); x2 = rnd(n, 16, 16, c).to(dt); dy = rnd(n, 16, 16, c).to(dt)
code = (torch.rand(n, c, device=dev, generator=g) < 0.5).float()
b = rnd(c)
w = ops.prep_weight(rnd(c, c, 3, 3) * 0.05, dt)
wt = ops.prep_weight(rnd(c, c, 3, 3) * 0.05, dt, transpose=True)
w2 = torch.cat([ops.prep_weight(rnd(c, c, 3, 3) * 0.05, dt), ops.prep_weight(rnd(c, c, 1, 1) * 0.05, dt)])
x32 = rnd(n, 32, 32, c).to(dt); dy32 = rnd(n, 32, 32, c).to(dt)
cases = {
    '16x16 fwd 128k3->128': (lambda: ops.conv_fused([Seg(x, code=code, relu=True)], w, c, bias=b), 2.0 * n * 256 * c * c * 9),
    '16x16 fwd 128k3+128k1->128 pool': (lambda: ops.conv_fused([Seg(x, code=code, relu=True), Seg(x2, ksize=1, code=code)], w2, c, bias=b, pool=True, alpha=0.25), 2.0 * n * 256 * c * c * 10),
    '16x16 dgrad 128k3->128 gate': (lambda: ops.conv_fused([Seg(dy)], wt, c, ocode=code, gate_x=x), 2.0 * n * 256 * c * c * 9),
    '32x32 dgrad 128k3->128 gate': (lambda: ops.conv_fused([Seg(dy32)], wt, c, ocode=code, gate_x=x32), 2.0 * n * 1024 * c * c * 9),
}
for cfg in sys.argv[1:] or ['256,128,20']:
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
