#!/usr/bin/env python3
"""Event-timed single convolution launch (bf16, BN + ReLU + code prologue), optionally from an alternative build of the
library.  GPU only.  usage: tools/conv_time.py N H Cin Cout [ksize=3] [tanh=0] [lib.so]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    a = sys.argv[1:]
    n, h, cin, cout = (int(v) for v in a[:4])
    ks = int(a[4]) if len(a) > 4 else 3
    tanh = bool(int(a[5])) if len(a) > 5 else False
    from mcgen_amd import _lib
    if len(a) > 6:
        _lib.LIB_PATH = os.path.abspath(a[6])
    from mcgen_amd import ops
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(3)
    x = torch.randn(n, h, h, cin, generator=g).to(dt).cuda()
    sc, sh = (torch.rand(cin, generator=g) + 0.5).cuda(), (torch.randn(cin, generator=g) * 0.2).cuda()
    code = (torch.rand(n, cin, generator=g) < 0.5).float().cuda()
    wimg = ops.prep_weight((torch.randn(cout, cin, ks, ks, generator=g) * 0.03).cuda(), dt)
    seg = ops.Seg(x, ksize=ks, scale=sc, shift=sh, code=code, relu=True)
    bias = torch.randn(cout, generator=g).cuda()
    for _ in range(5):
        y, _ = ops.conv_fused([seg], wimg, cout, bias=bias, tanh=tanh)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 30
    e0.record()
    for _ in range(reps):
        ops.conv_fused([seg], wimg, cout, bias=bias, tanh=tanh, out=y)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    gb = (x.numel() + y.numel()) * 2 / 1e9
    print(f'N {n} {h}x{h} {cin}k{ks}->{cout}{" tanh" if tanh else ""}: {us:8.1f} us  {gb / us * 1e6:7.0f} GB/s  '
          f'{2.0 * n * h * h * cin * cout * ks * ks / us / 1e6:7.1f} TF  checksum {float(y.float().sum()):.6e}  [{os.path.basename(_lib.LIB_PATH)}]', flush=True)


if __name__ == '__main__':
    main()
