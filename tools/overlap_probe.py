#!/usr/bin/env python3
"""Does the generator's forward pass overlap the discriminator's update when both run on their own stream?
(eager launches; compares sequential and two-stream wall time of: fake_next = G(z)  ||  D update on fake_cur)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
import bench
from mcgen_amd import ops
from mcgen_amd.trainer import GANTrainer

dev = torch.device('cuda')
model, sd = bench.build_model(torch.bfloat16, dev, 'CIFAR10')
tr = GANTrainer(model, 10)
n = 128
img = torch.rand(n, 3, 32, 32, device=dev) * 2 - 1
lab = torch.randint(0, 10, (n,), device=dev)
ind = F.one_hot(lab, 10).float()
ind2 = ind.repeat(2, 1)
z = torch.randn(n, tr.latent, device=dev)
tr.model.train(True)

def d_pass(fake):
    logits, ctx = tr.deng.forward_pair(img, fake, ind, ind2)
    lg = logits.view(-1)
    loss, _, _, dboth = ops.hinge_d(lg[:n], lg[n:], both=True)
    tr.deng.backward(ctx, dboth, tr.grad_d, False, False)

def seq(k):
    fake, _ = tr.geng.forward(z, ind, True)
    for _ in range(k):
        nxt, _ = tr.geng.forward(z, ind, True)
        d_pass(fake)
        fake = nxt

side = torch.cuda.Stream()
def par(k):
    fake, _ = tr.geng.forward(z, ind, True)
    for _ in range(k):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            nxt, _ = tr.geng.forward(z, ind, True)
        d_pass(fake)
        torch.cuda.current_stream().wait_stream(side)
        fake = nxt

for name, f in (('sequential', seq), ('two streams', par), ('sequential', seq), ('two streams', par)):
    f(3); torch.cuda.synchronize()
    t0 = time.perf_counter(); f(20); torch.cuda.synchronize()
    print(f'{name:12s} {1e3 * (time.perf_counter() - t0) / 20:.3f} ms per (G forward + D update)')
