#!/usr/bin/env python3
"""Fixed cost vs per-chunk cost of the software-pipelined 256 x 256 convolution tile:
dense 3x3, N x 32 x 32 -> 256 channels with BN + ReLU + code prologue and stats epilogue, for Cin = 32 .. 256.
time(Cin) = launches-per-CU * (F + c * Cin/32); prints the fit.  GPU only.
usage: tools/pp_fc.py [N] [cout] [path of an alternative libmcgen_hip.so (tuning build)]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    from mcgen_amd import _lib
    if len(sys.argv) > 3:
        _lib.LIB_PATH = os.path.abspath(sys.argv[3])
    from mcgen_amd import ops
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    cout = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(3)
    rows = []
    for stats in (1, 0):
        for cin in range(64, 257, 64):
            x = torch.randn(n, 32, 32, cin, generator=g).to(dt).cuda()
            sc, sh = (torch.rand(cin, generator=g) + 0.5).cuda(), (torch.randn(cin, generator=g) * 0.2).cuda()
            code = (torch.rand(n, cin, generator=g) < 0.5).float().cuda()
            wimg = ops.prep_weight((torch.randn(cout, cin, 3, 3, generator=g) * 0.03).cuda(), dt)
            seg = ops.Seg(x, scale=sc, shift=sh, code=code, relu=True)
            y = torch.empty(n, 32, 32, cout, dtype=dt, device='cuda')
            ops.FORM_LOG = []
            for _ in range(5):
                ops.conv_fused([seg], wimg, cout, stats_mode=stats, out=y)
            form = ops.FORM_LOG[-1] if ops.FORM_LOG else None
            ops.FORM_LOG = None
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 40
            e0.record()
            for _ in range(reps):
                ops.conv_fused([seg], wimg, cout, stats_mode=stats, out=y)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / reps
            tf = 2.0 * n * 1024 * cout * cin * 9 / us / 1e6
            rows.append((stats, cin, us, tf))
            print(f'stats {stats} Cin {cin:3d}: {us:7.1f} us  {tf:7.1f} TF  form {form}', flush=True)
    tiles_per_cu = n * 1024 / 256 / 256
    for stats in (1, 0):
        r = [(c / 32, us) for s, c, us, _ in rows if s == stats]
        a = np.polyfit([q for q, _ in r], [u for _, u in r], 1)
        print(f'stats {stats}: per tile F = {a[1] / tiles_per_cu:.2f} us, per 32-channel chunk c = {a[0] / tiles_per_cu:.2f} us '
              f'(ideal at 2.5 PF: {2 * 256 * cout * 32 * 9 / (2.5e15 / 256) * 1e6:.2f} us); {tiles_per_cu:g} tiles per CU')


if __name__ == '__main__':
    main()
