#!/usr/bin/env python3
"""Which torch-level ops (not our HIP library) one eager MCGAN iteration launches: torch.profiler, grouped by op."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mcgen_amd.trainer import GANTrainer

dev = torch.device('cuda')
model, sd = bench.build_model(torch.bfloat16, dev, 'CIFAR10')
tr = GANTrainer(model, 10)
img = torch.rand(128, 3, 32, 32, device=dev) * 2 - 1
lab = torch.randint(0, 10, (128,), device=dev)
for _ in range(2):
    tr.train_iteration(img, lab)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
    tr.train_iteration(img, lab)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by='count', row_limit=120, max_name_column_width=60))
