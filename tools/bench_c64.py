#!/usr/bin/env python3
"""COIL100's 64-channel convolutions on 32x32 maps under tile overrides (tuning build: MCGEN_CONV_CFG=bm,bn,pipe).
usage (GPU box, tuning library): MCGEN_TUNING=1 python tools/bench_c64.py 64,64,11 128,64,5 256,64,5 256,64,15"""
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


def cases():
    out = {}
    n = 640
    x16 = rnd(n, 16, 16, 128).to(dt); h32 = rnd(n, 32, 32, 64).to(dt)
    sc, sh = rnd(5, 128), rnd(5, 128); code = (torch.rand(n, 128, device=dev, generator=g) < 0.5).float()
    sc2, sh2 = rnd(5, 64), rnd(5, 64); code2 = (torch.rand(n, 64, device=dev, generator=g) < 0.5).float()
    w1 = ops.prep_weight(rnd(64, 128, 3, 3) * 0.05, dt)
    w2s = torch.cat([ops.prep_weight(rnd(64, 64, 3, 3) * 0.05, dt), ops.prep_weight(rnd(64, 128, 1, 1) * 0.05, dt)])
    b = rnd(64)
    out['G N640 128k3->64 s1'] = (lambda: ops.conv_fused([Seg(x16, scale=sc, shift=sh, code=code, ups=True, relu=True, group_n=128)], w1, 64, bias=b, stats_mode=1),
                                  2.0 * n * 1024 * 64 * 128 * 9)
    out['G N640 64k3+128k1->64 s1'] = (lambda: ops.conv_fused([Seg(h32, scale=sc2, shift=sh2, code=code2, relu=True, group_n=128), Seg(x16, ksize=1, code=code, ups=True)], w2s, 64, bias=b, stats_mode=1),
                                       2.0 * n * 1024 * 64 * (64 * 9 + 128))
    n = 256
    c1 = rnd(n, 32, 32, 64).to(dt); img = rnd(n, 32, 32, 8).to(dt)
    cd = (torch.rand(n, 64, device=dev, generator=g) < 0.5).float(); ones = torch.ones(n, 8, device=dev)
    wd = torch.cat([ops.prep_weight(rnd(64, 64, 3, 3) * 0.05, dt), ops.prep_weight(rnd(64, 8, 1, 1) * 0.05, dt)])
    out['D N256 64k3+8k1->64 pool'] = (lambda: ops.conv_fused([Seg(c1, code=cd, relu=True), Seg(img, ksize=1, code=ones)], wd, 64, bias=b, pool=True),
                                       2.0 * n * 1024 * 64 * (64 * 9 + 8))
    wt = ops.prep_weight(rnd(64, 64, 3, 3) * 0.05, dt, transpose=True)
    dy = rnd(n, 32, 32, 64).to(dt)
    out['D N256 64k3->64 gate'] = (lambda: ops.conv_fused([Seg(dy)], wt, 64, ocode=cd, gate_x=c1), 2.0 * n * 1024 * 64 * 64 * 9)
    x16d = rnd(n, 16, 16, 128).to(dt); cd16 = (torch.rand(n, 128, device=dev, generator=g) < 0.5).float()
    w16 = ops.prep_weight(rnd(64, 128, 3, 3) * 0.05, dt, transpose=False)
    out['D N256 16x16 128k3->64'] = (lambda: ops.conv_fused([Seg(x16d, code=cd16, relu=True)], w16, 64, bias=b), 2.0 * n * 256 * 64 * 128 * 9)
    if 'small' in os.environ.get('C64_SET', ''):
        out = {}
        x4 = rnd(n, 4, 4, 512).to(dt); x4b = rnd(n, 4, 4, 256).to(dt)
        c512 = (torch.rand(n, 512, device=dev, generator=g) < 0.5).float(); c256 = (torch.rand(n, 256, device=dev, generator=g) < 0.5).float()
        b512 = rnd(512)
        wa = torch.cat([ops.prep_weight(rnd(512, 512, 3, 3) * 0.03, dt), ops.prep_weight(rnd(512, 256, 1, 1) * 0.05, dt)])
        out['D N256 4x4 512k3+256k1->512'] = (lambda: ops.conv_fused([Seg(x4, code=c512, relu=True), Seg(x4b, ksize=1, code=c256)], wa, 512, bias=b512),
                                              2.0 * n * 16 * 512 * (512 * 9 + 256))
        wt = ops.prep_weight(rnd(512, 512, 3, 3) * 0.03, dt, transpose=True)
        out['D N256 4x4 512k3->512 gate'] = (lambda: ops.conv_fused([Seg(x4)], wt, 512, ocode=c512, gate_x=x4), 2.0 * n * 16 * 512 * 512 * 9)
        wb = ops.prep_weight(rnd(512, 256, 3, 3) * 0.03, dt)
        out['D N256 4x4 256k3->512'] = (lambda: ops.conv_fused([Seg(x4b, code=c256, relu=True)], wb, 512, bias=b512), 2.0 * n * 16 * 512 * 256 * 9)
        x8 = rnd(n, 8, 8, 256).to(dt); x8b = rnd(n, 8, 8, 128).to(dt); c128 = (torch.rand(n, 128, device=dev, generator=g) < 0.5).float()
        wc = torch.cat([ops.prep_weight(rnd(256, 256, 3, 3) * 0.03, dt), ops.prep_weight(rnd(256, 128, 1, 1) * 0.05, dt)])
        out['D N256 8x8 256k3+128k1->256 pool'] = (lambda: ops.conv_fused([Seg(x8, code=c256, relu=True), Seg(x8b, ksize=1, code=c128)], wc, 256, bias=rnd(256), pool=True, alpha=0.25),
                                                   2.0 * n * 64 * 256 * (256 * 9 + 128))
    return out


cs = cases()
for cfg in sys.argv[1:] or ['64,64,11']:
    os.environ['MCGEN_CONV_CFG'] = cfg
    for name, (fn, flops) in cs.items():
        try:
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10):
                fn()
            e.record(); torch.cuda.synchronize()
            us = s.elapsed_time(e) * 100
            print(f'{cfg:12s} {name:30s} {us:8.1f} us  {flops / us / 1e6:7.0f} TFLOP/s')
        except Exception as ex:
            print(f'{cfg:12s} {name:30s} failed: {str(ex)[:100]}')
