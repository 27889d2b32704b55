#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/ by RUNNING THE
REFERENCE (its ``models``/``modules`` packages, imported on CPU).

Runs only in the build container, where /root/reference exists; the reference
never travels to the GPU box -- only the .npz vectors do.  The reference's
``utils.py``/``data.py`` need torchvision and do not import here, so the cfg
keys that ``process_control`` would derive (utils.py:104-192) are set by hand
and the train loop body (train_gan.py:139-176) is driven around the imported
model with explicitly injected latents.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden.py [--only NAME]
"""
import argparse
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get('MC_REFERENCE', '/root/reference/src')
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(REPO, 'tests'))
os.chdir(REF)                     # config.py opens config.yml relatively (config.py:4)
sys.path.insert(0, REF)

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from config import cfg  # noqa: E402

import golden_util as gu  # noqa: E402

torch.set_num_threads(8)
OUT = gu.GOLDEN_DIR


def set_gan_cfg(g_hidden, d_hidden, classes, data_name='CIFAR10'):
    cfg['model_name'] = 'mcgan'
    cfg['data_name'] = data_name
    cfg['device'] = 'cpu'
    cfg['classes_size'] = classes
    cfg['controller_rate'] = 0.5
    cfg['data_shape'] = [3, 32, 32]
    cfg['gan'] = {'latent_size': 128, 'generator_hidden_size': list(g_hidden),
                  'discriminator_hidden_size': list(d_hidden), 'embedding_size': 32}


def np_state(sd, prefix='sd/'):
    return {prefix + k: v.detach().cpu().numpy().copy() for k, v in sd.items()}


def save(name, **arrays):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrays)
    print(f'wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)')


# --------------------------------------------------------------------------- #
def fx_mc_unit():
    """MultimodalController forward/backward, 4-D and 2-D inputs (modules.py:71-76)."""
    from modules import MultimodalController
    torch.manual_seed(0)
    mc = MultimodalController(32, 10, 0.5)
    x4 = torch.randn(4, 32, 4, 4, requires_grad=True)
    x2 = torch.randn(4, 32, requires_grad=True)
    lab = torch.tensor([3, 0, 9, 3])
    ind = F.one_hot(lab, 10).float()
    g4, g2 = torch.randn(4, 32, 4, 4), torch.randn(4, 32)
    o4 = mc([x4, ind])[0]; o4.backward(g4)
    o2 = mc([x2, ind])[0]; o2.backward(g2)
    soft = torch.softmax(torch.randn(4, 10), -1)         # non one-hot indicator
    os_ = mc([x4.detach(), soft])[0]
    ones = MultimodalController(32, 10, 1)                # rate 1 -> identity
    save('mc_unit.npz', codebook=mc.codebook.numpy(), x4=x4.detach().numpy(), x2=x2.detach().numpy(),
         label=lab.numpy(), g4=g4.numpy(), g2=g2.numpy(), out4=o4.detach().numpy(), out2=o2.detach().numpy(),
         dx4=x4.grad.numpy(), dx2=x2.grad.numpy(), soft=soft.numpy(), out_soft=os_.numpy(),
         ones_codebook=ones.codebook.numpy())


def fx_blocks():
    """Single residual blocks of the reference: outputs, input/param grads,
    BN running stats and SN u/v after one and two training forwards."""
    import importlib
    import models  # noqa: F401
    M = sys.modules.get('models.mcgan') or importlib.import_module('models.mcgan')
    from models.utils import make_SpectralNormalization
    set_gan_cfg([16] * 4, [16] * 4, 10)
    torch.manual_seed(3)
    lab = torch.tensor([1, 7, 7, 0, 4, 9])
    ind = F.one_hot(lab, 10).float()
    arrays = {'label': lab.numpy()}

    def run(tag, blk, x):
        for p in blk.parameters():
            torch.nn.init.normal_(p, 0.0, 0.2) if p.dim() > 1 else torch.nn.init.uniform_(p, 0.5, 1.5)
        blk.train(True)
        arrays.update(np_state(blk.state_dict(), f'{tag}/sd0/'))
        x = x.clone().requires_grad_(True)
        out = blk([x, ind])[0]
        g = torch.randn_like(out)
        out.backward(g)
        arrays[f'{tag}/x'] = x.detach().numpy(); arrays[f'{tag}/g'] = g.numpy()
        arrays[f'{tag}/out'] = out.detach().numpy(); arrays[f'{tag}/dx'] = x.grad.numpy()
        for n, p in blk.named_parameters():
            arrays[f'{tag}/grad/{n}'] = p.grad.numpy().copy()
        arrays.update(np_state(blk.state_dict(), f'{tag}/sd1/'))
        out2 = blk([x.detach(), ind])[0]                   # second forward: SN/BN state advances again
        arrays[f'{tag}/out_second'] = out2.detach().numpy()
        arrays.update(np_state(blk.state_dict(), f'{tag}/sd2/'))
        blk.train(False)
        arrays[f'{tag}/out_eval'] = blk([x.detach(), ind])[0].detach().numpy()

    run('gen', M.GenResBlock(16, 16, 10, 0.5, 2), torch.randn(6, 16, 4, 4))
    run('dis_first', M.FirstDisResBlock(3, 16, 10, 0.5).apply(make_SpectralNormalization), torch.randn(6, 3, 8, 8))
    run('dis_s2', M.DisResBlock(16, 16, 10, 0.5, 2).apply(make_SpectralNormalization), torch.randn(6, 16, 8, 8))
    run('dis_s1', M.DisResBlock(16, 16, 10, 0.5, 1).apply(make_SpectralNormalization), torch.randn(6, 16, 4, 4))
    save('mcgan_blocks.npz', **arrays)


def ref_train_iteration(model, opt, img, label, zs, d_iters=5, g_iters=1):
    """train_gan.py:139-176 around the imported model, latents injected."""
    zi = iter(zs)
    for _ in range(d_iters):
        opt['discriminator'].zero_grad(); opt['generator'].zero_grad()
        d_x = model.discriminate(img, label)
        generated = model.generate(label, next(zi))
        d_gz = model.discriminate(generated.detach(), label)
        d_loss = F.relu(1.0 - d_x).mean() + F.relu(1.0 + d_gz).mean()
        d_loss.backward()
        opt['discriminator'].step()
    for _ in range(g_iters):
        opt['discriminator'].zero_grad(); opt['generator'].zero_grad()
        generated = model.generate(label, next(zi))
        g_loss = -model.discriminate(generated, label).mean()
        g_loss.backward()
        opt['generator'].step()
    return d_loss.item(), g_loss.item()


def make_opt(model):
    return {'generator': torch.optim.Adam(model.generator.parameters(), lr=2e-4, betas=(0.5, 0.999)),
            'discriminator': torch.optim.Adam(model.discriminator.parameters(), lr=2e-4, betas=(0.5, 0.999))}


def fx_mcgan_small():
    """Reduced-width MCGAN, reference init (seed 0), 3 train iterations, B=8."""
    import models
    set_gan_cfg([32] * 4, [16] * 4, 10)
    torch.manual_seed(0)
    model = models.mcgan()
    model.train(True)
    arrays = np_state(model.state_dict(), 'sd/')
    B, iters = 8, 3
    img, lab = gu.synthetic_batch(B, 10, seed=1)
    zs = gu.latent_batches(6 * iters + 1, B, 128, seed=2)
    arrays['img'] = img.numpy(); arrays['label'] = lab.numpy()
    arrays['z'] = torch.stack(zs).numpy()
    # single forward/backward probes before any update
    probe = model.generate(lab, zs[-1])
    d_probe = model.discriminate(img, lab)
    arrays['probe_generated'] = probe.detach().numpy()
    arrays['probe_d_real'] = d_probe.detach().numpy()
    arrays.update(np_state(model.state_dict(), 'sd_after_probe/'))
    model.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in arrays.items() if k.startswith('sd/')})
    opt = make_opt(model)
    losses = []
    for it in range(iters):
        losses.append(ref_train_iteration(model, opt, img, lab, zs[6 * it:6 * it + 6]))
    arrays['losses'] = np.array(losses, dtype=np.float64)
    arrays.update(np_state(model.state_dict(), 'sd_final/'))
    model.train(False)
    with torch.no_grad():
        arrays['final_generated_eval'] = model.generate(lab, zs[-1]).numpy()
        arrays['final_d_eval'] = model.discriminate(img, lab).numpy()
    save('mcgan_small.npz', **arrays)


def fx_mcgan_full():
    """Full-size MCGAN (utils.py:156-162) with procedural weights: only inputs
    that cannot be regenerated, losses and digests are stored."""
    import models
    g_hidden, d_hidden = [256] * 4, [128] * 4
    set_gan_cfg(g_hidden, d_hidden, 10)
    torch.manual_seed(0)
    model = models.mcgan()
    shapes = gu.mcgan_shapes(g_hidden, d_hidden, 10)
    ref_shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    assert shapes == ref_shapes, set(shapes.items()) ^ set(ref_shapes.items())
    sd = gu.procedural_state(shapes, seed=1234, num_mode=10)
    model.load_state_dict(sd)
    model.train(True)
    B = 16
    img, lab = gu.synthetic_batch(B, 10, seed=1)
    zs = gu.latent_batches(6 * 2, B, 128, seed=2)
    arrays = {}
    gen0 = model.generate(lab, zs[0])
    d0 = model.discriminate(img, lab)
    arrays['probe_generated'] = gen0.detach().numpy()[:, :, ::4, ::4].copy()   # strided sample
    arrays['probe_generated_digest'] = gu.checksum(gen0)
    arrays['probe_d_real'] = d0.detach().numpy()
    model.load_state_dict(sd)
    opt = make_opt(model)
    losses = [ref_train_iteration(model, opt, img, lab, zs[0:6]),
              ref_train_iteration(model, opt, img, lab, zs[6:12])]
    arrays['losses'] = np.array(losses, dtype=np.float64)
    fin = model.state_dict()
    for k in ['generator.blocks.2.conv.8.module.weight', 'generator.linear.module.bias',
              'discriminator.blocks.1.conv.2.module.weight_orig', 'discriminator.blocks.4.conv.5.module.weight_u'
              if 'discriminator.blocks.4.conv.5.module.weight_u' in fin else 'discriminator.blocks.3.conv.5.module.weight_u',
              'generator.blocks.3.module.running_var', 'discriminator.blocks.7.module.weight_orig']:
        arrays['digest/' + k] = gu.checksum(fin[k])
    save('mcgan_full_digest.npz', **arrays)


def fx_mcgan_coil():
    """Non-CIFAR block schedule (models/mcgan.py:166-175): G [64,32,16,8],
    D [8,16,32,64], 20 modes -- exercises channel-changing blocks and the
    stride-1 block with a 1x1 shortcut."""
    import models
    set_gan_cfg([64, 32, 16, 8], [8, 16, 32, 64], 20, data_name='COIL100')
    torch.manual_seed(5)
    model = models.mcgan()
    model.train(True)
    arrays = np_state(model.state_dict(), 'sd/')
    B = 4
    img, lab = gu.synthetic_batch(B, 20, seed=11)
    zs = gu.latent_batches(6, B, 128, seed=12)
    arrays['img'] = img.numpy(); arrays['label'] = lab.numpy(); arrays['z'] = torch.stack(zs).numpy()
    opt = make_opt(model)
    arrays['losses'] = np.array([ref_train_iteration(model, opt, img, lab, zs)], dtype=np.float64)
    arrays.update(np_state(model.state_dict(), 'sd_final/'))
    save('mcgan_coil_small.npz', **arrays)


FIXTURES = {'mc_unit': fx_mc_unit, 'blocks': fx_blocks, 'mcgan_small': fx_mcgan_small,
            'mcgan_full': fx_mcgan_full, 'mcgan_coil': fx_mcgan_coil}

def fx_dp_emulation():
    """Two-replica data parallel as the reference's nn.DataParallel computes it (train_gan.py:96-98):
    the same weights see two batch shards (per-shard BatchNorm statistics), gradients are averaged.
    Stored: the averaged gradients of one D loss and one G loss on the reduced-width model."""
    import models
    set_gan_cfg([32] * 4, [16] * 4, 10)
    torch.manual_seed(0)
    model = models.mcgan()
    model.train(True)
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    B = 8
    img, lab = gu.synthetic_batch(B, 10, seed=21)
    z = gu.latent_batches(2, B, 128, seed=22)
    arrays = np_state(sd0, 'sd/')
    arrays['img'] = img.numpy(); arrays['label'] = lab.numpy(); arrays['z'] = torch.stack(z).numpy()
    shards = [slice(0, B // 2), slice(B // 2, B)]
    for step in ('d', 'g'):
        acc = None
        for sh in shards:
            model.load_state_dict(sd0)
            model.zero_grad()
            if step == 'd':
                d_x = model.discriminate(img[sh], lab[sh])
                fake = model.generate(lab[sh], z[0][sh])
                loss = F.relu(1.0 - d_x).mean() + F.relu(1.0 + model.discriminate(fake.detach(), lab[sh])).mean()
            else:
                loss = -model.discriminate(model.generate(lab[sh], z[1][sh]), lab[sh]).mean()
            loss.backward()
            net = model.discriminator if step == 'd' else model.generator
            g = {n: p.grad.clone() for n, p in net.named_parameters()}
            acc = g if acc is None else {n: acc[n] + g[n] for n in g}
        for n, v in acc.items():
            arrays[f'grad_{step}/{n}'] = (v / len(shards)).numpy()
    save('mcgan_dp2.npz', **arrays)


FIXTURES['dp'] = fx_dp_emulation

class _PatchedNoise:
    """Make torch.randn_like / torch.rand_like return (and record) tensors from a seeded CPU generator,
    so the noise the reference draws INSIDE its models becomes part of the fixture."""

    def __init__(self, seed):
        self.g = torch.Generator().manual_seed(seed)
        self.drawn = []

    def __enter__(self):
        self._randn_like, self._rand_like = torch.randn_like, torch.rand_like
        torch.randn_like = lambda t, **k: self._draw(torch.randn(t.shape, generator=self.g))
        torch.rand_like = lambda t, **k: self._draw(torch.rand(t.shape, generator=self.g))
        return self

    def _draw(self, t):
        self.drawn.append(t.clone())
        return t

    def __exit__(self, *a):
        torch.randn_like, torch.rand_like = self._randn_like, self._rand_like


def _single_opt_steps(model, inputs, noise_seed, steps, arrays, clip=1.0, lr=3e-4):
    """train_vae.py:98-126 / train_glow.py / train_pixelcnn.py loop body: zero_grad, forward, backward,
    clip_grad_norm_(1), Adam(lr 3e-4).step()."""
    opt = torch.optim.Adam(model.parameters(), lr=lr)
    losses = []
    for s in range(steps):
        with _PatchedNoise(noise_seed + s) as pn:
            opt.zero_grad()
            out = model({k: (v.clone() if torch.is_tensor(v) else v) for k, v in inputs.items()})
            out['loss'].backward()
            torch.nn.utils.clip_grad_norm_(model.parameters(), clip)
            opt.step()
        losses.append(out['loss'].item())
        for j, t in enumerate(pn.drawn):
            arrays[f'noise/{s}/{j}'] = t.numpy()
        if s == 0:
            first = out
    arrays['losses'] = np.array(losses, dtype=np.float64)
    return first


def fx_mcvae_small():
    """MCVAE (config 0 family), reduced width [8, 16, 32], latent 16, B=8, 3 optimizer steps."""
    import models
    cfg['model_name'] = 'mcvae'; cfg['data_name'] = 'CIFAR10'; cfg['device'] = 'cpu'; cfg['classes_size'] = 10
    cfg['controller_rate'] = 0.5; cfg['data_shape'] = [3, 32, 32]
    cfg['vae'] = {'hidden_size': [8, 16, 32], 'latent_size': 16, 'num_res_block': 2, 'embedding_size': 32}
    torch.manual_seed(0)
    model = models.mcvae(); model.train(True)
    arrays = np_state(model.state_dict(), 'sd/')
    img, lab = gu.synthetic_batch(8, 10, seed=31)
    arrays['img'] = img.numpy(); arrays['label'] = lab.numpy()
    first = _single_opt_steps(model, {'img': img, 'label': lab}, 100, 3, arrays)
    arrays['mu0'] = first['mu'].detach().numpy(); arrays['logvar0'] = first['logvar'].detach().numpy()
    arrays['img0'] = first['img'].detach().numpy()
    arrays.update(np_state(model.state_dict(), 'sd_final/'))
    model.train(False)
    z = torch.randn(8, 16, generator=torch.Generator().manual_seed(5))
    arrays['gen_z'] = z.numpy()
    with torch.no_grad():
        arrays['generated_eval'] = model.generate(lab, z).numpy()
    save('mcvae_small.npz', **arrays)


def fx_mcvae_full():
    """Config 0 of BASELINE.json: MCVAE CIFAR-10 (hidden [64,128,256], latent 128), batch 32, on CPU:
    losses of 2 optimizer steps + output digests; weights = the reference's own seed-0 init, stored."""
    import models
    cfg['model_name'] = 'mcvae'; cfg['data_name'] = 'CIFAR10'; cfg['device'] = 'cpu'; cfg['classes_size'] = 10
    cfg['controller_rate'] = 0.5; cfg['data_shape'] = [3, 32, 32]
    cfg['vae'] = {'hidden_size': [64, 128, 256], 'latent_size': 128, 'num_res_block': 2, 'embedding_size': 32}
    torch.manual_seed(0)
    model = models.mcvae(); model.train(True)
    assert sum(p.numel() for p in model.parameters()) == 7628931
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = gu.procedural_state_generic(shapes, seed=4321)
    model.load_state_dict(sd)
    arrays = {'shape_keys': np.array(sorted(shapes)), 'shape_vals': np.array([str(shapes[k]) for k in sorted(shapes)])}
    img, lab = gu.synthetic_batch(32, 10, seed=1)
    first = _single_opt_steps(model, {'img': img, 'label': lab}, 200, 2, arrays)
    arrays['mu0_digest'] = gu.checksum(first['mu']); arrays['img0_digest'] = gu.checksum(first['img'])
    arrays['img0_sample'] = first['img'].detach().numpy()[:4, :, ::4, ::4].copy()
    save('mcvae_full_digest.npz', **arrays)


def fx_mcpixelcnn_small():
    """MCGatedPixelCNN, hidden 16, 4 layers, 32 codes, 8x8 code maps, B=6, 3 optimizer steps."""
    import models
    cfg['model_name'] = 'mcpixelcnn'; cfg['device'] = 'cpu'; cfg['classes_size'] = 10; cfg['controller_rate'] = 0.5
    cfg['pixelcnn'] = {'num_layer': 4, 'hidden_size': 16, 'num_embedding': 32}
    torch.manual_seed(0)
    model = models.mcpixelcnn(); model.train(True)
    arrays = np_state(model.state_dict(), 'sd/')
    g = torch.Generator().manual_seed(41)
    codes = torch.randint(0, 32, (6, 8, 8), generator=g)
    lab = torch.randint(0, 10, (6,), generator=g)
    arrays['codes'] = codes.numpy(); arrays['label'] = lab.numpy()
    first = _single_opt_steps(model, {'img': codes, 'label': lab}, 300, 3, arrays)
    arrays['logits0'] = first['logits'].detach().numpy()
    arrays.update(np_state(model.state_dict(), 'sd_final/'))
    model.train(False)
    with torch.no_grad():
        arrays['logits_eval'] = model({'img': codes, 'label': lab})['logits'].numpy()
    save('mcpixelcnn_small.npz', **arrays)


def fx_mcglow_small():
    """MCGlow [1,32,32], K=2, L=3, hidden 32, 12 modes, B=4: ActNorm data init (first forward), 2 optimizer
    steps, reverse(reconstruct) and sampling from fixed z."""
    import models
    cfg['model_name'] = 'mcglow'; cfg['device'] = 'cpu'; cfg['classes_size'] = 12; cfg['controller_rate'] = 0.5
    cfg['data_shape'] = [1, 32, 32]
    cfg['glow'] = {'hidden_size': 32, 'K': 2, 'L': 3, 'affine': True, 'conv_lu': True}
    torch.manual_seed(0); np.random.seed(0)
    model = models.mcglow(); model.train(True)
    arrays = np_state(model.state_dict(), 'sd/')
    img, lab = gu.synthetic_batch(4, 12, seed=51, shape=(1, 32, 32))
    arrays['img'] = img.numpy(); arrays['label'] = lab.numpy()
    with _PatchedNoise(400) as pn, torch.no_grad():                       # train_glow.py:60-67 data-dependent init
        model({'img': img.clone(), 'label': lab})
    arrays['noise/init/0'] = pn.drawn[0].numpy()
    arrays.update(np_state(model.state_dict(), 'sd_init/'))
    first = _single_opt_steps(model, {'img': img, 'label': lab}, 500, 2, arrays)
    for i, z in enumerate(first['z']):
        arrays[f'z0/{i}'] = z.detach().numpy()
    arrays.update(np_state(model.state_dict(), 'sd_final/'))
    model.train(False)
    with torch.no_grad(), _PatchedNoise(600) as pn:
        out = model({'img': img.clone(), 'label': lab})
        arrays['noise/eval/0'] = pn.drawn[0].numpy()
        arrays['loss_eval'] = np.array(out['loss'].item())
        rec = model.reverse({'z': out['z'], 'label': lab, 'reconstruct': True})['img']
        arrays['reconstructed'] = rec.numpy()
        gz = [torch.randn(4, *s, generator=torch.Generator().manual_seed(7 + i)) * 0.7 for i, s in enumerate(model.make_z_shapes())]
        for i, z in enumerate(gz):
            arrays[f'gen_z/{i}'] = z.numpy()
        arrays['generated'] = model.generate(lab, gz).numpy()
    save('mcglow_small.npz', **arrays)


def fx_vqvae_small():
    """VQ-VAE (the frozen auto-encoder in front of MCPixelCNN, train_pixelcnn.py:111-113), reduced: hidden [16, 16],
    64 codes of size 8, B=4.  Two training-mode forwards move the BN running statistics and the EMA codebook off
    their initial values; then the eval-mode encode (encoder output, code map, quantised tensor) and decode_code."""
    import models
    cfg['model_name'] = 'vqvae'; cfg['device'] = 'cpu'; cfg['data_shape'] = [3, 32, 32]
    cfg['vqvae'] = {'hidden_size': [16, 16], 'num_res_block': 2, 'embedding_size': 8, 'num_embedding': 64, 'vq_commit': 0.25}
    torch.manual_seed(0)
    model = models.vqvae(); model.train(True)
    img, _ = gu.synthetic_batch(4, 10, seed=61)
    with torch.no_grad():
        for _ in range(2):
            model({'img': img.clone()})
    model.train(False)
    with torch.no_grad():
        # a random-init encoder maps everything next to one code; spread the codebook over the encoder's outputs
        # (the codebook is state, i.e. an input of this fixture) so that the arg-min is exercised
        flat0 = model.encoder(img).transpose(1, -1).contiguous().view(-1, 8)
        gsel = torch.Generator().manual_seed(62)
        model.quantizer.embedding.copy_((flat0[::4] + 0.02 * torch.randn(64, 8, generator=gsel)).t())
    arrays = np_state(model.state_dict(), 'sd/')
    arrays['img'] = img.numpy()
    with torch.no_grad():
        x = model.encoder(img)
        encoded, vq_loss, code = model.encode(img)
        arrays['enc_out'] = x.numpy(); arrays['encoded'] = encoded.numpy(); arrays['code'] = code.numpy()
        arrays['vq_loss'] = np.array(vq_loss.item())
        arrays['decoded'] = model.decode_code(code).numpy()
        flat = x.transpose(1, -1).contiguous().view(-1, 8)
        emb = model.quantizer.embedding
        dist = flat.pow(2).sum(1, keepdim=True) - 2 * flat @ emb + emb.pow(2).sum(0, keepdim=True)
        top2 = dist.topk(2, dim=1, largest=False).values
        arrays['dist_margin'] = (top2[:, 1] - top2[:, 0]).numpy()          # how decisive each argmin is
    save('vqvae_small.npz', **arrays)


FIXTURES.update(vqvae=fx_vqvae_small)
FIXTURES.update(mcvae_small=fx_mcvae_small, mcvae_full=fx_mcvae_full, mcpixelcnn=fx_mcpixelcnn_small, mcglow=fx_mcglow_small)


def _hook_checksums(model, arrays, tag):
    """Forward hooks on every top-level block of G and D: order-sensitive checksums of the block outputs
    (NCHW, as the reference holds them) of the NEXT forward of each network."""
    handles = []
    for net_name in ('generator', 'discriminator'):
        net = getattr(model, net_name)
        for i, blk in enumerate(net.blocks.children()):
            def hook(mod, inp, out, key=f'{tag}/{net_name}.blocks.{i}'):
                t = out[0] if isinstance(out, (list, tuple)) else out
                if torch.is_tensor(t) and t.dim() == 4 and key not in arrays:
                    arrays[key] = gu.checksum(t)
            handles.append(blk.register_forward_hook(hook))
    return handles


def fx_mcgan_full_b128():
    """BASELINE configs[1] at its stated batch: full-size MCGAN (utils.py:156-162), procedural weights, B=128,
    ONE train iteration (train_gan.py:139-176): losses, a strided sample + digest of the probe batch, per-block
    activation digests of one G and one D forward, digests of a few final tensors."""
    import models
    g_hidden, d_hidden = [256] * 4, [128] * 4
    set_gan_cfg(g_hidden, d_hidden, 10)
    torch.manual_seed(0)
    model = models.mcgan()
    sd = gu.procedural_state(gu.mcgan_shapes(g_hidden, d_hidden, 10), seed=1234, num_mode=10)
    model.load_state_dict(sd)
    model.train(True)
    B = 128
    img, lab = gu.synthetic_batch(B, 10, seed=1)
    zs = gu.latent_batches(6, B, 128, seed=2)
    arrays = {}
    hs = _hook_checksums(model, arrays, 'act')
    with torch.no_grad():
        gen0 = model.generate(lab, zs[0])
        d0 = model.discriminate(img, lab)
    for h in hs:
        h.remove()
    arrays['probe_generated'] = gen0.numpy()[:, :, ::4, ::4].copy()
    arrays['probe_generated_digest'] = gu.checksum(gen0)
    arrays['probe_d_real'] = d0.numpy()
    model.load_state_dict(sd)
    opt = make_opt(model)
    arrays['losses'] = np.array([ref_train_iteration(model, opt, img, lab, zs)], dtype=np.float64)
    fin = model.state_dict()
    for k in ['generator.blocks.2.conv.8.module.weight', 'generator.blocks.1.conv.4.module.weight', 'generator.linear.module.bias',
              'discriminator.blocks.0.conv.3.module.weight_orig', 'discriminator.blocks.1.conv.2.module.weight_orig',
              'discriminator.blocks.3.conv.5.module.weight_u', 'generator.blocks.3.module.running_var',
              'generator.blocks.2.conv.5.module.running_mean']:
        arrays['digest/' + k] = gu.checksum(fin[k])
    save('mcgan_full_digest_b128.npz', **arrays)


def fx_mcgan_coil_full():
    """BASELINE configs[2] as the reference runs it (utils.py:116-118,163-165; data.py:51): COIL100 at 32x32,
    G [512,256,128,64], D [64,128,256,512], 100 modes, non-CIFAR block schedule; procedural weights, B=8,
    one train iteration + probes."""
    import models
    g_hidden, d_hidden = [512, 256, 128, 64], [64, 128, 256, 512]
    set_gan_cfg(g_hidden, d_hidden, 100, data_name='COIL100')
    torch.manual_seed(0)
    model = models.mcgan()
    shapes = gu.mcgan_shapes(g_hidden, d_hidden, 100, cifar_layout=False)
    ref_shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    assert shapes == ref_shapes, set(shapes.items()) ^ set(ref_shapes.items())
    sd = gu.procedural_state(shapes, seed=4242, num_mode=100)
    model.load_state_dict(sd)
    model.train(True)
    B = 8
    img, lab = gu.synthetic_batch(B, 100, seed=5)
    zs = gu.latent_batches(6, B, 128, seed=6)
    arrays = {}
    hs = _hook_checksums(model, arrays, 'act')
    with torch.no_grad():
        gen0 = model.generate(lab, zs[0])
        d0 = model.discriminate(img, lab)
    for h in hs:
        h.remove()
    arrays['probe_generated'] = gen0.numpy().copy()
    arrays['probe_d_real'] = d0.numpy()
    model.load_state_dict(sd)
    opt = make_opt(model)
    arrays['losses'] = np.array([ref_train_iteration(model, opt, img, lab, zs)], dtype=np.float64)
    fin = model.state_dict()
    for k in ['generator.blocks.0.conv.4.module.weight', 'generator.blocks.2.shortcut.2.module.weight',
              'discriminator.blocks.3.conv.5.module.weight_orig', 'discriminator.blocks.2.shortcut.1.module.weight_u']:
        arrays['digest/' + k] = gu.checksum(fin[k])
    save('mcgan_coil_full_digest.npz', **arrays)


def fx_mcglow_full():
    """BASELINE configs[3] as the reference runs it (utils.py:110-112,172-184; data.py:40): MCGlow on Omniglot
    [1,32,32], 1623 modes, hidden 512, K=16, L=3 (15,844,992 parameters); procedural weights (golden_util.
    procedural_state_glow), B=4: ActNorm data init forward, then two train_glow.py steps; losses + z digests."""
    import models
    cfg['model_name'] = 'mcglow'; cfg['device'] = 'cpu'; cfg['classes_size'] = 1623; cfg['controller_rate'] = 0.5
    cfg['data_shape'] = [1, 32, 32]
    cfg['glow'] = {'hidden_size': 512, 'K': 16, 'L': 3, 'affine': True, 'conv_lu': True}
    torch.manual_seed(0); np.random.seed(0)
    model = models.mcglow(); model.train(True)
    assert sum(p.numel() for p in model.parameters()) == 15844992
    ref_sd = model.state_dict()
    keys = sorted(ref_sd)
    shapes = {k: tuple(ref_sd[k].shape) for k in keys}
    dtypes = {k: str(ref_sd[k].dtype).replace('torch.', '') for k in keys}
    sd = gu.procedural_state_glow(shapes, dtypes, seed=777)
    model.load_state_dict(sd)
    arrays = {'shape_keys': np.array(keys), 'shape_vals': np.array([str(shapes[k]) for k in keys]),
              'dtype_vals': np.array([dtypes[k] for k in keys])}
    img, lab = gu.synthetic_batch(4, 1623, seed=71, shape=(1, 32, 32))
    arrays['img'] = img.numpy(); arrays['label'] = lab.numpy()
    with _PatchedNoise(800) as pn, torch.no_grad():                       # train_glow.py:60-67 data-dependent init
        model({'img': img.clone(), 'label': lab})
    arrays['noise/init/0'] = pn.drawn[0].numpy()
    init = model.state_dict()
    for k in ['blocks.0.flows.0.actnorm.loc', 'blocks.0.flows.0.coupling.net.1.module.scale',
              'blocks.1.flows.7.coupling.net.5.module.loc', 'blocks.2.flows.15.actnorm.scale']:
        arrays['init_digest/' + k] = gu.checksum(init[k])
    first = _single_opt_steps(model, {'img': img, 'label': lab}, 900, 2, arrays)
    for i, z in enumerate(first['z']):
        arrays[f'z0_digest/{i}'] = gu.checksum(z)
        arrays[f'z0_sample/{i}'] = z.detach().numpy()[:, :, ::2, ::2].copy()
    fin = model.state_dict()
    for k in ['blocks.0.flows.3.coupling.net.4.module.weight', 'blocks.2.flows.9.invconv.w_l', 'blocks.1.prior.conv.weight']:
        arrays['final_digest/' + k] = gu.checksum(fin[k])
    save('mcglow_full_digest.npz', **arrays)


def fx_mcpixelcnn_full():
    """BASELINE configs[4] as the reference runs it (utils.py:139-143): MCGatedPixelCNN with 15 layers, hidden 128,
    512 codes, 10 modes (6,367,616 parameters) on 8x8 code maps at the config's batch 128; procedural weights
    (golden_util.procedural_state_generic).  Loss + logits digest of the first training forward, the digest of EVERY
    parameter's gradient of that step (before clip_grad_norm_), two train_pixelcnn.py steps (losses), digests of a few
    final tensors -- a key that is absent from the state dict is an error, not a silent skip."""
    import models
    cfg['model_name'] = 'mcpixelcnn'; cfg['device'] = 'cpu'; cfg['classes_size'] = 10; cfg['controller_rate'] = 0.5
    cfg['pixelcnn'] = {'num_layer': 15, 'hidden_size': 128, 'num_embedding': 512}
    torch.manual_seed(0)
    model = models.mcpixelcnn(); model.train(True)
    assert sum(p.numel() for p in model.parameters()) == 6367616
    ref_sd = model.state_dict()
    keys = sorted(ref_sd)
    shapes = {k: tuple(ref_sd[k].shape) for k in keys}
    model.load_state_dict(gu.procedural_state_generic(shapes, seed=4242))
    arrays = {'shape_keys': np.array(keys), 'shape_vals': np.array([str(shapes[k]) for k in keys])}
    g = torch.Generator().manual_seed(43)
    B = 128
    codes = torch.randint(0, 512, (B, 8, 8), generator=g)
    lab = torch.randint(0, 10, (B,), generator=g)
    arrays['codes'] = codes.numpy(); arrays['label'] = lab.numpy()
    opt = torch.optim.Adam(model.parameters(), lr=3e-4)                           # train_pixelcnn.py:33,108-121
    losses = []
    for step in range(2):
        opt.zero_grad()
        out = model({'img': codes.clone(), 'label': lab})
        out['loss'].backward()
        if step == 0:
            arrays['logits0_digest'] = gu.checksum(out['logits'].detach())
            arrays['logits0_sample'] = out['logits'].detach().numpy()[::16, ::16, ::2, ::2].copy()
            names, dead = [], []
            for k, p in model.named_parameters():
                if p.grad is None:                       # (the last layer's vertical stream feeds nothing: mcpixelcnn.py:56-61)
                    dead.append(k)
                    continue
                names.append(k)
                arrays['grad0_digest/' + k] = gu.checksum(p.grad)
                arrays['grad0_absmax/' + k] = np.array(float(p.grad.abs().max()))
            arrays['grad_keys'] = np.array(names)
            arrays['nograd_keys'] = np.array(dead)
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1)
        opt.step()
        losses.append(out['loss'].item())
    arrays['losses'] = np.array(losses, dtype=np.float64)
    fin = model.state_dict()
    final = ['layers.0.vert_stack.weight', 'layers.7.horiz_resid.0.module.weight', 'layers.14.gate_h.bn.running_var', 'output_conv.4.module.weight']
    for k in final:
        if k not in fin:
            raise KeyError(f'{k} is not a key of the reference state dict (have e.g. {[x for x in fin if x.startswith(k.split(".")[0])][:8]})')
        arrays['final_digest/' + k] = gu.checksum(fin[k].float())
    arrays['final_keys'] = np.array(final)
    save('mcpixelcnn_full_digest.npz', **arrays)


def fx_classifier():
    """The feature network of IS / FID on COIL100 / Omniglot (models/classifier.py:14-52; metrics.py:49-62,89-113):
    hidden [8, 16, 32, 64] (utils.py:185), evaluation mode, on both data shapes ([3,32,32] and Omniglot's [1,32,32]); weights from the reference's own
    initialisation with non-trivial BatchNorm running statistics.  Outputs: features and logits of a batch, and the
    Inception Score / FID the reference's formulas (metrics.py:75-82,139-161) give on them."""
    import models
    from scipy import linalg
    arrays = {}
    for tag, shape, classes in (('coil100', [3, 32, 32], 100), ('gray', [1, 32, 32], 40)):     # (Omniglot's shape; 40 of its 1623 classes keep the file small)
        cfg['model_name'] = 'classifier'; cfg['device'] = 'cpu'; cfg['classes_size'] = classes; cfg['data_shape'] = shape
        cfg['classifier'] = {'hidden_size': [8, 16, 32, 64]}
        torch.manual_seed(5)
        model = models.classifier()
        g = torch.Generator().manual_seed(61)
        with torch.no_grad():
            for m in model.modules():
                if isinstance(m, torch.nn.BatchNorm2d):
                    m.running_mean.copy_(0.1 * torch.randn(m.running_mean.shape, generator=g))
                    m.running_var.copy_(0.5 + torch.rand(m.running_var.shape, generator=g))
        model.train(False)
        arrays.update(np_state(model.state_dict(), f'{tag}/sd/'))
        img = torch.rand(20, *shape, generator=g) * 2 - 1
        real = torch.rand(24, *shape, generator=g) * 2 - 1
        arrays[f'{tag}/img'] = img.numpy(); arrays[f'{tag}/real'] = real.numpy()
        with torch.no_grad():
            feat = model.feature({'img': img})
            out = model({'img': img, 'label': torch.zeros(20, dtype=torch.long)})
            rfeat = model.feature({'img': real})
        arrays[f'{tag}/feature'] = feat.numpy(); arrays[f'{tag}/logits'] = out['label'].numpy()
        pred = F.softmax(out['label'], dim=-1)                                    # metrics.py:61-62,75-81
        py = pred.mean(0)
        arrays[f'{tag}/inception_score'] = np.array(F.kl_div(py.log().view(1, -1).expand_as(pred), pred, reduction='batchmean').exp().item())
        a, b = rfeat.numpy().astype(np.float64), feat.numpy().astype(np.float64)  # metrics.py:139-161
        mu1, mu2 = a.mean(0), b.mean(0)
        s1, s2 = np.cov(a, rowvar=False), np.cov(b, rowvar=False)
        covmean, _ = linalg.sqrtm(s1.dot(s2), disp=False)
        if not np.isfinite(covmean).all():
            off = np.eye(s1.shape[0]) * 1e-6
            covmean = linalg.sqrtm((s1 + off).dot(s2 + off))
        covmean = covmean.real if np.iscomplexobj(covmean) else covmean
        d = mu1 - mu2
        arrays[f'{tag}/fid'] = np.array(d.dot(d) + np.trace(s1) + np.trace(s2) - 2 * np.trace(covmean))
    save('classifier_small.npz', **arrays)


FIXTURES.update(mcgan_full_b128=fx_mcgan_full_b128, mcgan_coil_full=fx_mcgan_coil_full, mcglow_full=fx_mcglow_full,
                mcpixelcnn_full=fx_mcpixelcnn_full, classifier=fx_classifier)

if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', default=None)
    a = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    for name, fn in FIXTURES.items():
        if a.only in (None, name):
            fn()
