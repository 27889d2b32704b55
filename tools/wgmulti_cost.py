#!/usr/bin/env python3
"""Calibrates the cost model of ops._launch_multi: one layer per launch (256 -> 256 channels, so 8 tiles x 32 splits = 256
workgroups), steps per workgroup 16 / 32 / 64 / 128 through the batch size, for 3x3 and 1x1 on 32x32 / 16x16 / 8x8 maps.
Prints microseconds per launch and the fitted fixed cost + cost per step.  usage (GPU box): python tools/wgmulti_cost.py"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mcgen_amd import ops

dev = torch.device('cuda')
g = torch.Generator(device=dev).manual_seed(0)
for ks in (3, 1):
    for side in (32, 16, 8):
        pts = []
        for spw in (16, 32, 64, 128):
            steps = spw * 32
            n = steps * 128 // (side * side)
            ci = co = 256
            x = torch.randn(n, side, side, ci, device=dev, generator=g).bfloat16()
            dy = (torch.randn(n, side, side, co, device=dev, generator=g) * 0.1).bfloat16()
            code = (torch.rand(n, ci, device=dev, generator=g) < 0.5).float()
            sc = torch.rand(ci, device=dev, generator=g) + 0.5
            sh = torch.randn(ci, device=dev, generator=g) * 0.3
            seg = ops.Seg(x, ksize=ks, scale=sc, shift=sh, code=code, relu=True)
            gw = torch.zeros(co, ci, ks, ks, device=dev)
            gb = torch.zeros(co, device=dev)

            def run():
                with ops.deferred_reduces():
                    ops.wgrad(seg, dy, co, ci, gw, bias_grad=gb)
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            ops._PROF = []
            for _ in range(10):
                run()
            torch.cuda.synchronize()
            rec, ops._PROF = ops._PROF, None
            ts = sorted(s.elapsed_time(e) * 1e3 for nm, _, s, e, *_ in rec if nm == 'wgrad_multi<bf16>')
            pts.append((spw, ts[len(ts) // 2]))
            del x, dy
        (a0, t0), (a1, t1) = pts[0], pts[-1]
        per = (t1 - t0) / (a1 - a0)
        print(f'k{ks} {side:2d}x{side:<2d}: ' + '  '.join(f'{a:3d} steps {t:7.1f} us' for a, t in pts) + f'   -> {per:5.2f} us/step, fixed {t0 - per * a0:5.1f} us')
