"""Helpers shared by tools/gen_golden.py and the parity tests.

Procedural weights: the full-size fixtures do not store weights.  They are
regenerated from a numpy PCG64 stream (not the torch RNG, whose streams differ
between CPU and GPU builds) in sorted-key order, so the generator script and
the tests build bit-identical state dicts from a seed alone.
"""
from __future__ import annotations

import os
from typing import Dict

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def golden_path(name: str) -> str:
    return os.path.join(GOLDEN_DIR, name)


def load_npz(name: str) -> Dict[str, np.ndarray]:
    with np.load(golden_path(name), allow_pickle=False) as f:
        return {k: f[k] for k in f.files}


def state_from_npz(d: Dict[str, np.ndarray], prefix: str = 'sd/') -> Dict[str, torch.Tensor]:
    """Rebuild a state dict; aliased codebook keys (shared mc_1 instance) are
    re-aliased so in-place edits of one key stay visible through the other."""
    out: Dict[str, torch.Tensor] = {}
    for k in sorted(d):
        if k.startswith(prefix):
            out[k[len(prefix):]] = torch.from_numpy(np.array(d[k]))
    return out


def procedural_state(shapes: Dict[str, tuple], seed: int, num_mode: int) -> Dict[str, torch.Tensor]:
    """Deterministic stand-in weights for a reference-format state dict.

    conv/linear weights ~ U(-a, a) with a = sqrt(6 / (fan_in + fan_out))
    (the xavier bound the reference uses for mcgan, models/utils.py:11-13),
    BN weight ~ N(1, 0.02), biases ~ U(-0.05, 0.05) so that bias paths are
    exercised, running stats at their torch defaults, u/v unit vectors,
    codebooks = distinct Bernoulli(0.5) rows.
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    sd: Dict[str, torch.Tensor] = {}
    codebooks: Dict[tuple, torch.Tensor] = {}
    for k in sorted(shapes):
        shp = tuple(shapes[k])
        leaf = k.rsplit('.', 1)[-1]
        if leaf == 'codebook':
            # aliased keys of one shared MultimodalController must agree:
            # blocks.i.mc_1 == blocks.i.conv.{3|1|2} == blocks.i.shortcut.{0|1}
            owner = _codebook_owner(k)
            if owner not in codebooks:
                while True:
                    cb = (rng.random(shp) < 0.5).astype(np.float32)
                    if len({tuple(r) for r in cb.tolist()}) == shp[0]:
                        break
                codebooks[owner] = torch.from_numpy(cb)
            sd[k] = codebooks[owner]
        elif leaf == 'running_mean':
            sd[k] = torch.zeros(shp)
        elif leaf == 'running_var':
            sd[k] = torch.ones(shp)
        elif leaf == 'num_batches_tracked':
            sd[k] = torch.zeros(shp, dtype=torch.int64)
        elif leaf in ('weight_u', 'weight_v'):
            v = rng.standard_normal(shp).astype(np.float32)
            sd[k] = torch.from_numpy(v / max(np.linalg.norm(v), 1e-12))
        elif leaf == 'bias':
            sd[k] = torch.from_numpy(rng.uniform(-0.05, 0.05, shp).astype(np.float32))
        elif len(shp) == 1:                      # BN weight
            sd[k] = torch.from_numpy((1.0 + 0.02 * rng.standard_normal(shp)).astype(np.float32))
        else:                                    # conv / linear weight(_orig)
            rf = int(np.prod(shp[2:])) if len(shp) > 2 else 1
            a = float(np.sqrt(6.0 / (shp[1] * rf + shp[0] * rf)))
            sd[k] = torch.from_numpy(rng.uniform(-a, a, shp).astype(np.float32))
    return sd


def procedural_state_generic(shapes: Dict[str, tuple], seed: int) -> Dict[str, torch.Tensor]:
    """Stand-in weights for any reference-format state dict (used where aliasing does not matter):
    matrices/filters ~ U(+-sqrt(3/fan_in)), BN weights ~ N(1, 0.02), biases small, codebooks distinct
    Bernoulli(0.5) rows shared per (modes, width) position by key order."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sd: Dict[str, torch.Tensor] = {}
    for k in sorted(shapes):
        shp = tuple(shapes[k])
        leaf = k.rsplit('.', 1)[-1]
        if leaf == 'codebook':
            while True:
                cb = (rng.random(shp) < 0.5).astype(np.float32)
                if len({tuple(r) for r in cb.tolist()}) == shp[0]:
                    break
            sd[k] = torch.from_numpy(cb)
        elif leaf == 'running_mean':
            sd[k] = torch.zeros(shp)
        elif leaf == 'running_var':
            sd[k] = torch.ones(shp)
        elif leaf == 'num_batches_tracked':
            sd[k] = torch.zeros(shp, dtype=torch.int64)
        elif leaf == 'bias':
            sd[k] = torch.from_numpy(rng.uniform(-0.05, 0.05, shp).astype(np.float32))
        elif len(shp) == 1:
            sd[k] = torch.from_numpy((1.0 + 0.02 * rng.standard_normal(shp)).astype(np.float32))
        else:
            fan_in = int(np.prod(shp[1:]))
            a = float(np.sqrt(3.0 / fan_in))
            sd[k] = torch.from_numpy(rng.uniform(-a, a, shp).astype(np.float32))
    return sd



def procedural_state_glow(shapes: Dict[str, tuple], dtypes: Dict[str, str], seed: int) -> Dict[str, torch.Tensor]:
    """Stand-in weights for a reference-format MCGlow state dict (models/mcglow.py:24-130), from key -> shape /
    dtype tables alone: coupling convolutions ~ N(0, 0.05) (mcglow.py:148-151), ZeroConv2d weights / scales small
    but non-zero (so every path carries signal), LU factors of the invertible 1x1 convolutions built structurally
    (w_p a permutation matrix, strictly triangular w_l / w_u, masks and identity as the reference registers them),
    ActNorm at (loc 0, scale 1, initialized 0) so that the first training forward runs the data-dependent init,
    codebooks Bernoulli(0.5)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sd: Dict[str, torch.Tensor] = {}
    for k in sorted(shapes):
        shp = tuple(shapes[k])
        leaf = k.rsplit('.', 1)[-1]
        zero_conv = k.endswith(('8.module.scale', 'prior.scale')) or '.module.conv.' in k or '.prior.conv.' in k
        if leaf == 'codebook':
            v = (rng.random(shp) < 0.5).astype(np.float32)
        elif leaf == 'initialized':
            v = np.zeros(shp, dtype=np.uint8)
        elif leaf == 'w_p':
            v = np.eye(shp[0], dtype=np.float32)[rng.permutation(shp[0])]
        elif leaf in ('w_l', 'w_u'):
            m = np.tril(np.ones(shp, dtype=np.float32), -1) if leaf == 'w_l' else np.triu(np.ones(shp, dtype=np.float32), 1)
            v = (0.2 * rng.standard_normal(shp)).astype(np.float32) * m
        elif leaf == 'w_s':
            v = (0.1 * rng.standard_normal(shp)).astype(np.float32)
        elif leaf == 's_sign':
            v = np.where(rng.random(shp) < 0.5, -1.0, 1.0).astype(np.float32)
        elif leaf == 'u_mask':
            v = np.triu(np.ones(shp, dtype=np.float32), 1)
        elif leaf == 'l_mask':
            v = np.tril(np.ones(shp, dtype=np.float32), -1)
        elif leaf == 'l_eye':
            v = np.eye(shp[0], dtype=np.float32)
        elif leaf == 'loc':
            v = np.zeros(shp, dtype=np.float32)
        elif leaf == 'scale':
            v = (0.05 * rng.standard_normal(shp)).astype(np.float32) if zero_conv else np.ones(shp, dtype=np.float32)
        elif leaf == 'bias':
            v = rng.uniform(-0.02, 0.02, shp).astype(np.float32)
        elif leaf == 'weight':
            v = ((0.01 if zero_conv else 0.05) * rng.standard_normal(shp)).astype(np.float32)
        else:
            raise KeyError(f'procedural_state_glow: unexpected key {k}')
        t = torch.from_numpy(np.ascontiguousarray(v))
        want = getattr(torch, dtypes[k])
        sd[k] = t if t.dtype == want else t.to(want)
    return sd


def _codebook_owner(key: str) -> str:
    """Map an aliased codebook key to its owning module's key (mc_1 / mc_2)."""
    parts = key.split('.')
    # <net>.blocks.<i>.<branch>.<j>.codebook  -> which mc does <branch>.<j> alias?
    if len(parts) >= 6 and parts[-3] in ('conv', 'shortcut'):
        net, i, branch, j = parts[0], parts[2], parts[-3], int(parts[-2])
        base = '.'.join(parts[:-3])
        if net == 'generator':
            which = 'mc_2' if (branch == 'conv' and j == 7) else 'mc_1'
        else:
            if i == '0':
                which = 'mc_1'
            else:
                which = 'mc_2' if (branch == 'conv' and j == 4) else 'mc_1'
        return base + '.' + which + '.codebook'
    return key


def mcgan_shapes(g_hidden, d_hidden, num_mode: int, latent: int = 128, in_ch: int = 3,
                 cifar_layout: bool = True) -> Dict[str, tuple]:
    """State-dict key -> shape table of the reference MCGAN
    (models/mcgan.py:47-69,150-191), written out from the module structure."""
    s: Dict[str, tuple] = {}

    def bn(p, c):
        s[p + '.weight'] = (c,); s[p + '.bias'] = (c,)
        s[p + '.running_mean'] = (c,); s[p + '.running_var'] = (c,)
        s[p + '.num_batches_tracked'] = ()

    def conv(p, co, ci, k, sn):
        s[p + '.bias'] = (co,)
        if sn:
            s[p + '.weight_orig'] = (co, ci, k, k)
            s[p + '.weight_u'] = (co,); s[p + '.weight_v'] = (ci * k * k,)
        else:
            s[p + '.weight'] = (co, ci, k, k)

    g = 'generator.'
    s[g + 'linear.module.weight'] = (g_hidden[0] * 16, latent)
    s[g + 'linear.module.bias'] = (g_hidden[0] * 16,)
    nb = len(g_hidden) - 1
    for i in range(nb):
        p = g + f'blocks.{i}.'
        ci, co = g_hidden[i], g_hidden[i + 1]
        s[p + 'mc_1.codebook'] = (num_mode, ci); s[p + 'mc_2.codebook'] = (num_mode, co)
        bn(p + 'conv.0.module', ci)
        s[p + 'conv.3.codebook'] = (num_mode, ci)
        conv(p + 'conv.4.module', co, ci, 3, False)
        bn(p + 'conv.5.module', co)
        s[p + 'conv.7.codebook'] = (num_mode, co)
        conv(p + 'conv.8.module', co, co, 3, False)
        s[p + 'shortcut.1.codebook'] = (num_mode, ci)
        conv(p + 'shortcut.2.module', co, ci, 1, False)
    bn(g + f'blocks.{nb}.module', g_hidden[-1])
    s[g + f'blocks.{nb + 2}.codebook'] = (num_mode, g_hidden[-1])
    conv(g + f'blocks.{nb + 3}.module', in_ch, g_hidden[-1], 3, False)

    d = 'discriminator.'
    p = d + 'blocks.0.'
    s[p + 'mc_1.codebook'] = (num_mode, d_hidden[0])
    conv(p + 'conv.0.module', d_hidden[0], in_ch, 3, True)
    s[p + 'conv.2.codebook'] = (num_mode, d_hidden[0])
    conv(p + 'conv.3.module', d_hidden[0], d_hidden[0], 3, True)
    conv(p + 'shortcut.0.module', d_hidden[0], in_ch, 1, True)
    if cifar_layout:
        plan = [(d_hidden[i], d_hidden[i + 1], 2) for i in range(len(d_hidden) - 3)]
        plan += [(d_hidden[-3], d_hidden[-2], 1), (d_hidden[-2], d_hidden[-1], 1)]
    else:
        plan = [(d_hidden[i], d_hidden[i + 1], 2) for i in range(len(d_hidden) - 2)]
        plan += [(d_hidden[-2], d_hidden[-1], 1)]
    for j, (ci, co, stride) in enumerate(plan, start=1):
        p = d + f'blocks.{j}.'
        s[p + 'mc_1.codebook'] = (num_mode, ci); s[p + 'mc_2.codebook'] = (num_mode, co)
        s[p + 'conv.1.codebook'] = (num_mode, ci)
        conv(p + 'conv.2.module', co, ci, 3, True)
        s[p + 'conv.4.codebook'] = (num_mode, co)
        conv(p + 'conv.5.module', co, co, 3, True)
        if stride > 1 or ci != co:
            s[p + 'shortcut.0.codebook'] = (num_mode, ci)
            conv(p + 'shortcut.1.module', co, ci, 1, True)
    t = len(plan) + 1
    s[d + f'blocks.{t + 1}.codebook'] = (num_mode, d_hidden[-1])
    k = d + f'blocks.{t + 3}.module'
    s[k + '.bias'] = (1,)
    s[k + '.weight_orig'] = (1, d_hidden[-1]); s[k + '.weight_u'] = (1,); s[k + '.weight_v'] = (d_hidden[-1],)
    return s


def synthetic_batch(batch: int, classes: int, seed: int = 1, shape=(3, 32, 32)):
    """Inputs of SURVEY 8(c): images U(-1,1), labels U{0..classes-1} from a
    seeded torch CPU generator (CPU generator streams are stable)."""
    g = torch.Generator().manual_seed(seed)
    img = torch.rand(batch, *shape, generator=g) * 2 - 1
    lab = torch.randint(0, classes, (batch,), generator=g)
    return img, lab


def latent_batches(n: int, batch: int, latent: int, seed: int = 2):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(batch, latent, generator=g) for _ in range(n)]


def checksum(t: torch.Tensor) -> np.ndarray:
    """Order-sensitive, tolerance-friendly digest: (sum, sum|x|, sum x*ramp)."""
    x = t.detach().double().flatten()
    ramp = torch.linspace(-1, 1, x.numel(), dtype=torch.float64)
    return np.array([x.sum().item(), x.abs().sum().item(), (x * ramp).sum().item()])
