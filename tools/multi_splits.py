#!/usr/bin/env python3
"""Prints how ops._launch_multi split the layers of the headline step's weight-gradient launches (one eager iteration).
usage (GPU box): [MCGEN_TUNING=1 MCGEN_WG_*=..] python tools/multi_splits.py"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import bench
from mcgen_amd import ops
from mcgen_amd.trainer import GANTrainer

dev = torch.device('cuda')
model, _ = bench.build_model(torch.bfloat16, dev)
tr = GANTrainer(model, 10)
g = torch.Generator(device=dev).manual_seed(1)
img = torch.rand(128, 3, 32, 32, device=dev, generator=g) * 2 - 1
label = torch.randint(0, 10, (128,), device=dev, generator=g)
tr.train_iteration(img, label)
ops.MULTI_LOG = []
tr.train_iteration(img, label)
torch.cuda.synchronize()
seen = []
for launch in ops.MULTI_LOG:
    if launch in seen:
        continue
    seen.append(launch)
    print('launch:', sum(b * s for _, _, _, b, s in launch), 'workgroups')
    for side, ks, steps, blocks, splits in launch:
        print(f'   {side:3d}x{side:<3d} k{ks}  steps {steps:5d}  tiles {blocks:3d}  splits {splits:4d}  -> {steps / splits:7.1f} steps per workgroup')
