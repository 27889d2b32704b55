#!/usr/bin/env python3
"""Same-box timing of the mode-compacted convolution against the dense dma3 form (bf16), interleaved rounds.
usage: tools/bench_mc.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from mcgen_amd import ops  # noqa: E402


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


def case(name, n, h, ci, co, ups, two_seg=False, pool_res=False):
    dt = torch.bfloat16
    g = torch.Generator(device='cuda').manual_seed(1)
    hs = h // 2 if ups else h
    x = torch.randn(n, hs, hs, ci, device='cuda', generator=g).to(dt)
    scale, shift = torch.rand(ci, device='cuda', generator=g) + 0.5, torch.randn(ci, device='cuda', generator=g) * 0.3
    code = (torch.rand(n, ci, device='cuda', generator=g) < 0.5).float()
    w = torch.randn(co, ci, 3, 3, device='cuda', generator=g) * 0.03
    b = torch.randn(co, device='cuda', generator=g)
    kw = dict(scale=scale, shift=shift, code=code, ups=ups, relu=True)
    segs_d, segs_c = [ops.Seg(x, **kw)], [ops.Seg(x, cmap=ops.mc_cmap(code), **kw)]
    img_d, img_c = ops.prep_weight(w, dt), ops.prep_weight_k(w, dt)
    flops = 2.0 * n * h * h * co * ci * 9
    if two_seg:
        xs = torch.randn(n, h // 2, h // 2, ci, device='cuda', generator=g).to(dt)
        code1 = (torch.rand(n, ci, device='cuda', generator=g) < 0.5).float()
        ws = torch.randn(co, ci, 1, 1, device='cuda', generator=g) * 0.05
        segs_d.append(ops.Seg(xs, ksize=1, code=code1, ups=True)); segs_c.append(ops.Seg(xs, ksize=1, code=code1, ups=True, cmap=ops.mc_cmap(code1)))
        img_d = torch.cat([img_d, ops.prep_weight(ws, dt)]); img_c = torch.cat([img_c, ops.prep_weight_k(ws, dt)])
        flops += 2.0 * n * h * h * co * ci
    yd = torch.empty((n, h, h, co), dtype=dt, device='cuda'); yc = torch.empty_like(yd)
    fd = lambda: ops.conv_fused(segs_d, img_d, co, bias=b, stats_mode=1, out=yd)                 # noqa: E731
    fc = lambda: ops.conv_fused(segs_c, img_c, co, bias=b, stats_mode=1, out=yc, kmajor=True)    # noqa: E731
    res = []
    for _ in range(3):
        res.append((timeit(fd), timeit(fc)))
    td, tc = min(r[0] for r in res), min(r[1] for r in res)
    print(f'{name:34s} dense {td:8.1f} us ({flops / td / 1e6:7.1f} TF)   compacted {tc:8.1f} us ({flops / tc / 1e6:7.1f} dense-equivalent TF)   x{td / tc:.2f}',
          flush=True)


def case_gk(name, n, h, c, ups, two_seg=False, ccap=160):
    """Consumer reading compacted activations (gathered-K form) against the dense launch; and the producer's compacted store."""
    dt = torch.bfloat16
    g = torch.Generator(device='cuda').manual_seed(2)
    hs = h // 2 if ups else h
    code = (torch.rand(n, c, device='cuda', generator=g) < 0.5).float()
    cm = ops.mc_cmap(code)
    x = torch.randn(n, hs, hs, c, device='cuda', generator=g).to(dt)
    idx = cm[:, c:c + ccap].long()
    xc = torch.gather(torch.nn.functional.pad(x, (0, 1)), 3, idx.view(n, 1, 1, ccap).expand(n, hs, hs, ccap)).contiguous()
    scale, shift = torch.rand(c, device='cuda', generator=g) + 0.5, torch.randn(c, device='cuda', generator=g) * 0.3
    sr, tr = ops.mc_affine(code, cm, ccap, scale, shift)
    w = torch.randn(c, c, 3, 3, device='cuda', generator=g) * 0.03
    b = torch.randn(c, device='cuda', generator=g)
    segs_d = [ops.Seg(x, scale=scale, shift=shift, code=code, ups=ups, relu=True)]
    segs_c = [ops.Seg(xc, scale=sr, shift=tr, ups=ups, relu=True, group_n=1, cmap=cm, cw=c)]
    img_d, img_k = ops.prep_weight(w, dt), ops.prep_weight_k(w, dt)
    flops = 2.0 * n * h * h * c * c * 9
    if two_seg:
        code1 = (torch.rand(n, c, device='cuda', generator=g) < 0.5).float()
        cm1 = ops.mc_cmap(code1)
        xs = torch.randn(n, h // 2, h // 2, c, device='cuda', generator=g).to(dt)
        idx1 = cm1[:, c:c + ccap].long()
        xsc = torch.gather(torch.nn.functional.pad(xs, (0, 1)), 3, idx1.view(n, 1, 1, ccap).expand(n, h // 2, h // 2, ccap)).contiguous()
        s1, t1 = ops.mc_affine(code1, cm1, ccap)
        ws = torch.randn(c, c, 1, 1, device='cuda', generator=g) * 0.05
        segs_d.append(ops.Seg(xs, ksize=1, code=code1, ups=True))
        segs_c.append(ops.Seg(xsc, ksize=1, scale=s1, shift=t1, ups=True, group_n=1, cmap=cm1, cw=c))
        img_d = torch.cat([img_d, ops.prep_weight(ws, dt)]); img_k = torch.cat([img_k, ops.prep_weight_k(ws, dt)])
        flops += 2.0 * n * h * h * c * c
    code_next = (torch.rand(n, c, device='cuda', generator=g) < 0.5).float()
    cmn = ops.mc_cmap(code_next)
    yd = torch.empty((n, h, h, c), dtype=dt, device='cuda'); yc = torch.empty((n, h, h, ccap), dtype=dt, device='cuda')
    fd = lambda: ops.conv_fused(segs_d, img_d, c, bias=b, stats_mode=1, out=yd)                                         # noqa: E731
    fc = lambda: ops.conv_fused(segs_c, img_k, c, bias=b, stats_mode=1, out=yd, kmajor=2)                               # noqa: E731
    fcc = lambda: ops.conv_fused(segs_c, img_k, c, bias=b, stats_mode=1, out=yc, kmajor=2, ycmap=cmn, cy=ccap)          # noqa: E731
    res = []
    for _ in range(3):
        res.append((timeit(fd), timeit(fc), timeit(fcc)))
    td, tc, tcc = (min(r[i] for r in res) for i in range(3))
    print(f'{name:34s} dense {td:8.1f} us ({flops / td / 1e6:7.1f} TF)   compacted in {tc:8.1f} us (x{td / tc:.2f})   in + out {tcc:8.1f} us (x{td / tcc:.2f}, '
          f'{flops / tcc / 1e6:7.1f} dense-equivalent TF)', flush=True)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'gk':
        case_gk('G conv_a 32x32 N=640', 640, 32, 256, True)
        case_gk('G conv_b+sc 32x32 N=640', 640, 32, 256, False, two_seg=True)
        case_gk('G conv_b+sc 16x16 N=640', 640, 16, 256, False, two_seg=True)
        case_gk('G conv_a 32x32 N=128', 128, 32, 256, True)
        case_gk('G conv_b+sc 32x32 N=128', 128, 32, 256, False, two_seg=True)
        sys.exit(0)
    case('G conv_a 32x32 256->256 N=128', 128, 32, 256, 256, True)
    case('G conv_b+sc 32x32 256->256 N=128', 128, 32, 256, 256, False, two_seg=True)
    case('G conv_a 32x32 256->256 N=640', 640, 32, 256, 256, True)
    case('G conv_a 16x16 256->256 N=640', 640, 16, 256, 256, True)
    case('G conv_a 16x16 256->256 N=128', 128, 16, 256, 256, True)
    case('D conv 32x32 128->128 N=256', 256, 32, 128, 128, False)
    case('D conv 16x16 128->128 N=256', 256, 16, 128, 128, False)
