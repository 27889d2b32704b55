#!/usr/bin/env python3
"""Times mcgen_wgrad_multi on the two passes of the headline step (the 3x3 weight gradients of one discriminator update
over 2N = 256 images and of the generator update over N = 128), random bf16 operands, HIP events over `reps` launches.
usage (GPU box): python tools/bench_wgmulti.py [reps]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mcgen_amd import ops

PASSES = {
    'D (2N=256)': [(256, 32, 128, 128, False, True, False, True), (256, 16, 128, 128, False, False, False, True),
                   (256, 16, 128, 128, False, True, False, True)] + [(256, 8, 128, 128, False, False, False, True)] * 4,
    'G (N=128)': [(128, 32, 256, 256, False, False, True, False), (128, 32, 256, 256, True, False, True, False),
                  (128, 16, 256, 256, False, False, True, False), (128, 16, 256, 256, True, False, True, False),
                  (128, 8, 256, 256, False, False, True, False), (128, 8, 256, 256, True, False, True, False)],
    'G 32x32 only': [(128, 32, 256, 256, False, False, True, False), (128, 32, 256, 256, True, False, True, False)],
}


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    nobias = 'nobias' in sys.argv[2:]
    noaff = 'noaffine' in sys.argv[2:]
    dev = torch.device('cuda')
    g = torch.Generator(device=dev).manual_seed(0)
    for name, layers in PASSES.items():
        prob, flops = [], 0.0
        for (n, h, ci, co, ups, dy_ups, affine, halves) in layers:
            hs, hd = (h // 2 if ups else h), (h // 2 if dy_ups else h)
            x = torch.randn(n, hs, hs, ci, device=dev, generator=g).bfloat16()
            dy = (torch.randn(n, hd, hd, co, device=dev, generator=g) * 0.1).bfloat16()
            code = (torch.rand(n, ci, device=dev, generator=g) < 0.5).float()
            sc = torch.rand(ci, device=dev, generator=g) + 0.5 if (affine and not noaff) else None
            sh = torch.randn(ci, device=dev, generator=g) * 0.3 if (affine and not noaff) else None
            seg = ops.Seg(x, scale=sc, shift=sh, code=None if noaff else code, ups=ups, relu=not noaff)
            gs = [torch.zeros(co, ci, 3, 3, device=dev) for _ in range(2 if halves else 1)]
            bs = [torch.zeros(co, device=dev) for _ in range(2 if halves else 1)]
            prob.append((seg, dy, co, ci, dy_ups, halves, gs, bs))
            flops += 2.0 * n * h * h * co * ci * 9

        def run():
            with ops.deferred_reduces():
                for seg, dy, co, ci, dy_ups, halves, gs, bs in prob:
                    ops.wgrad(seg, dy, co, ci, gs[0], dy_ups=dy_ups, bias_grad=None if nobias else bs[0],
                              second=(gs[1], None if nobias else bs[1], None) if halves else None)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        ops._PROF = []
        for _ in range(reps):
            run()
        torch.cuda.synchronize()
        rec, ops._PROF = ops._PROF, None
        for key in ('wgrad_multi<bf16>', 'wgrad_reduce'):
            ts = sorted(s.elapsed_time(e) * 1e3 for nm, _, s, e, *_ in rec if nm == key)
            if ts:
                med = ts[len(ts) // 2]
                print(f'{name:14s} {key:22s} median {med:8.1f} us  min {ts[0]:8.1f}' + (f'   {flops / med / 1e6:7.0f} TFLOP/s' if 'multi' in key else ''))


if __name__ == '__main__':
    main()
