#!/usr/bin/env python3
"""Micro-benchmark of the fused conv / wgrad kernels on the headline model's layer shapes.
Tuning aid: loops over MCGEN_CONV_CFG tile overrides in one process.  Usage (GPU box):
    python tools/bench_conv.py [--cfgs 128,128,0 128,256,1 ...] [--wgrad]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from mcgen_amd import ops  # noqa: E402
from mcgen_amd.ops import Seg  # noqa: E402

N = 128
dt = torch.bfloat16
dev = 'cuda'


def rnd(*s):
    return torch.randn(*s, device=dev)


def act(n, h, c):
    return rnd(n, h, h, c).to(dt)


def layers():
    out = {}
    for name, h, c in (('G2', 32, 256), ('G1', 16, 256), ('G0', 8, 256)):
        x_lo, hmid = act(N, h // 2, c), act(N, h, c)
        sc, sh, code = rnd(c), rnd(c), (torch.rand(N, c, device=dev) < 0.5).float()
        w1 = ops.prep_weight(rnd(c, c, 3, 3) * 0.05, dt)
        w2s = torch.cat([ops.prep_weight(rnd(c, c, 3, 3) * 0.05, dt), ops.prep_weight(rnd(c, c, 1, 1) * 0.05, dt)])
        b = rnd(c)
        out[name + 'a'] = (lambda x_lo=x_lo, sc=sc, sh=sh, code=code, w1=w1, b=b, c=c:
                           ops.conv_fused([Seg(x_lo, scale=sc, shift=sh, code=code, ups=True, relu=True)], w1, c, bias=b, stats_mode=1),
                           2.0 * N * h * h * c * c * 9)
        out[name + 'b'] = (lambda hmid=hmid, x_lo=x_lo, sc=sc, sh=sh, code=code, w2s=w2s, b=b, c=c:
                           ops.conv_fused([Seg(hmid, scale=sc, shift=sh, code=code, relu=True), Seg(x_lo, ksize=1, code=code, ups=True)],
                                          w2s, c, bias=b, stats_mode=1),
                           2.0 * N * h * h * c * c * 10)
        wt = ops.prep_weight(rnd(c, c, 3, 3) * 0.05, dt, transpose=True)
        mean, rstd = rnd(c), rnd(c).abs() + 0.5
        out[name + 'dg'] = (lambda hmid=hmid, wt=wt, code=code, sc=sc, sh=sh, mean=mean, rstd=rstd, c=c:
                            ops.conv_fused([Seg(hmid)], wt, c, ocode=code, gate_x=hmid, gscale=sc, gshift=sh, gmean=mean, grstd=rstd, stats_mode=2),
                            2.0 * N * h * h * c * c * 9)
    for name, h, c, pool in (('D0', 32, 128, True), ('D1', 16, 128, False), ('D2', 8, 128, False)):
        x = act(N, h, c)
        code = (torch.rand(N, c, device=dev) < 0.5).float()
        w = ops.prep_weight(rnd(c, c, 3, 3) * 0.05, dt)
        b = rnd(c)
        res = act(N, h // 2 if pool else h, c)
        out[name] = (lambda x=x, code=code, w=w, b=b, c=c, pool=pool, res=res:
                     ops.conv_fused([Seg(x, code=code, relu=True)], w, c, bias=b, pool=pool, alpha=0.25 if pool else 1.0, res=res),
                     2.0 * N * h * h * c * c * 9)
    return out


def wgrad_layers():
    out = {}
    for name, h, c in (('G2', 32, 256), ('G1', 16, 256), ('D0', 32, 128), ('D1', 16, 128), ('D2', 8, 128)):
        x, dy = act(N, h, c), act(N, h, c)
        sc, sh, code = rnd(c), rnd(c), (torch.rand(N, c, device=dev) < 0.5).float()
        g = torch.zeros(c, c, 3, 3, device=dev)
        bg = torch.zeros(c, device=dev)
        out['wg' + name] = (lambda x=x, dy=dy, sc=sc, sh=sh, code=code, g=g, bg=bg, c=c:
                            ops.wgrad(Seg(x, scale=sc, shift=sh, code=code, relu=True), dy, c, c, g, bias_grad=bg),
                            2.0 * N * h * h * c * c * 9)
    for name, h, c in (('K1_16', 16, 512), ('K1_8', 8, 512)):
        x, dy = act(N, h, c), act(N, h, c)
        sc, sh, code = rnd(c), rnd(c), (torch.rand(N, c, device=dev) < 0.5).float()
        g = torch.zeros(c, c, 1, 1, device=dev)
        bg = torch.zeros(c, device=dev)
        out['wg' + name] = (lambda x=x, dy=dy, sc=sc, sh=sh, code=code, g=g, bg=bg, c=c:
                            ops.wgrad(Seg(x, ksize=1, scale=sc, shift=sh, code=code, relu=True), dy, c, c, g, bias_grad=bg),
                            2.0 * N * h * h * c * c)
    return out


def timeit(fn, reps=10):
    fn(); fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e-3


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--cfgs', nargs='*', default=['default'])
    ap.add_argument('--wgrad', action='store_true')
    ap.add_argument('--only', default=None)
    ap.add_argument('--n', type=int, default=128)
    ap.add_argument('--us', action='store_true', help='print microseconds instead of TFLOP/s')
    a = ap.parse_args()
    N = a.n
    globals()['N'] = a.n
    L = wgrad_layers() if a.wgrad else layers()
    names = [n for n in L if a.only is None or n in a.only.split(',')]
    print('cfg'.ljust(14) + ''.join(n.rjust(9) for n in names) + '   (TFLOP/s)')
    for cfg in a.cfgs:
        if cfg == 'default':
            os.environ.pop('MCGEN_CONV_CFG', None)
        else:
            os.environ['MCGEN_CONV_CFG'] = cfg
        row = cfg.ljust(14)
        for n in names:
            fn, flops = L[n]
            try:
                t = timeit(fn)
                row += f'{t * 1e6:9.1f}' if a.us else f'{flops / t / 1e12:9.0f}'
            except Exception as ex:
                row += '      err'
        print(row, flush=True)
