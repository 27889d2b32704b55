#!/usr/bin/env python3
"""Compare the 64x64-tile main-loop variants (MCGEN_CONV_SMALL) on the headline model's small-map layers: outputs must
agree bit for bit between variants (same accumulation order per output element)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import torch
import bench_conv as B
from mcgen_amd import ops
from mcgen_amd.ops import Seg

if os.environ.get('CMP_N'):
    B.N = int(os.environ['CMP_N'])
L = B.layers()
names = [n for n in L if n.startswith(('G0', 'D2', 'D1'))] if not os.environ.get('CMP_N') else list(L)
dt = torch.bfloat16
z = torch.randn(128, 1, 1, 128, device='cuda').to(dt)
wl = ops.prep_weight(torch.randn(4096, 128, device='cuda') * 0.1, dt, row_perm=16)
L['lin'] = (lambda: ops.conv_fused([Seg(z, ksize=1)], wl, 4096, bias=torch.ones(4096, device='cuda'), stats_mode=1), 0)
names.append('lin')
x4 = torch.randn(128, 4, 4, 256, device='cuda').to(dt)
w4 = ops.prep_weight(torch.randn(256, 256, 3, 3, device='cuda') * 0.05, dt, transpose=True)
cd = (torch.rand(128, 256, device='cuda') < 0.5).float()
L['dg_pool'] = (lambda: ops.conv_fused([Seg(torch.randn(128, 8, 8, 256, device='cuda', generator=torch.Generator(device='cuda').manual_seed(1)).to(dt))], w4, 256,
                                        pool=True, alpha=1.0, ocode=cd, gate_x=x4, gscale=torch.ones(256, device='cuda'), gshift=torch.zeros(256, device='cuda'),
                                        gmean=torch.zeros(256, device='cuda'), grstd=torch.ones(256, device='cuda'), stats_mode=2), 0)
names.append('dg_pool')
gen = torch.Generator(device='cuda').manual_seed(5)
dy8 = torch.randn(128, 8, 8, 256, device='cuda', generator=gen).to(dt)
ws = ops.prep_weight(torch.randn(256, 256, 1, 1, device='cuda', generator=gen) * 0.05, dt, transpose=True)
L['sc_pool'] = (lambda: ops.conv_fused([Seg(dy8, ksize=1)], ws, 256, pool=True, alpha=1.0, ocode=cd), 0)
c1 = torch.randn(128, 8, 8, 128, device='cuda', generator=gen).to(dt)
dy128 = torch.randn(128, 8, 8, 128, device='cuda', generator=gen).to(dt)
w128 = ops.prep_weight(torch.randn(128, 128, 3, 3, device='cuda', generator=gen) * 0.05, dt, transpose=True)
cd128 = (torch.rand(128, 128, device='cuda', generator=gen) < 0.5).float()
L['d_gate'] = (lambda: ops.conv_fused([Seg(dy128)], w128, 128, ocode=cd128, gate_x=c1), 0)
L['d_gate_res'] = (lambda: ops.conv_fused([Seg(dy128)], w128, 128, ocode=cd128, gate_x=c1, res=dy128), 0)
z4 = torch.randn(128, 4, 4, 256, device='cuda', generator=gen).to(dt)
L['g4_gate_stats'] = (lambda: ops.conv_fused([Seg(dy8)], w4, 256, pool=True, alpha=1.0, ocode=cd, gate_x=z4, gscale=torch.rand(256, device='cuda', generator=torch.Generator(device='cuda').manual_seed(3)) + 0.5,
                                              gshift=torch.randn(256, device='cuda', generator=torch.Generator(device='cuda').manual_seed(4)), gmean=torch.zeros(256, device='cuda'), grstd=torch.ones(256, device='cuda'), stats_mode=2), 0)
lin_dy = torch.randn(128, 1, 1, 4096, device='cuda', generator=gen).to(dt)
names += ['sc_pool', 'd_gate', 'd_gate_res', 'g4_gate_stats']
for n in names:
    outs = {}
    for mode in ('5', '11', '10'):
        os.environ['MCGEN_CONV_SMALL'] = mode
        y, st = L[n][0]()
        torch.cuda.synchronize()
        outs[mode] = (y.float().clone(), None if st is None else st.clone())
    for mode in ('11', '10'):
        dy = float((outs[mode][0] - outs['5'][0]).abs().max())
        ds = 0.0 if outs['5'][1] is None else max(float((outs[mode][1][:, j] - outs['5'][1][:, j]).abs().max() / (outs['5'][1][:, j].abs().max() + 1e-9)) for j in (0, 1))
        print(f'{n:8s} mode {mode}: max |dy| = {dy:.3e}   stats rel diff = {ds:.3e}   nan={bool(torch.isnan(outs[mode][0]).any())}')
