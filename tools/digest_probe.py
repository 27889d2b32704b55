#!/usr/bin/env python3
"""Print the two-iteration bf16 losses of the full-size digest run (tests/test_mcgan_gpu.py::test_full_size_digest)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import golden_util as gu
import test_mcgan_gpu as T
from mcgen_amd.trainer import GANTrainer
d = gu.load_npz('mcgan_full_digest.npz')
sd = gu.procedural_state(gu.mcgan_shapes([256] * 4, [128] * 4, 10), seed=1234, num_mode=10)
dt = torch.float32 if os.environ.get('PROBE_F32') else torch.bfloat16
m = T._build([256] * 4, [128] * 4, 10, 'CIFAR10', sd, dt)
img, lab = gu.synthetic_batch(16, 10, seed=1)
zs = [z.cuda() for z in gu.latent_batches(12, 16, 128, seed=2)]
m.train(True)
tr = GANTrainer(m, 10)
l0 = tr.train_iteration(img.cuda(), lab.cuda(), zs[0:6])
l1 = tr.train_iteration(img.cuda(), lab.cuda(), zs[6:12])
print('it1 %.5f %.5f  it2 %.5f %.5f   ref it2 %.5f %.5f' % (float(l0[0]), float(l0[1]), float(l1[0]), float(l1[1]), d['losses'][1][0], d['losses'][1][1]))
