"""CPU restatement (plain fp32 torch ops) of the reference MCVAE (models/mcvae.py).

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.  Pure functions over the reference's
``state_dict`` keys; the reparameterisation noise is an explicit input (the reference draws it inside
the model with ``torch.randn_like``, mcvae.py:58-61)."""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import torch
import torch.nn.functional as F

from .mcgan_oracle import batch_norm, mc_mask, one_hot

Tensor = torch.Tensor
State = Dict[str, Tensor]


def _conv(sd, key, x, stride, pad):
    return F.conv2d(x, sd[key + '.weight'], sd[key + '.bias'], stride=stride, padding=pad)


def _convt(sd, key, x):
    """nn.ConvTranspose2d(ci, co, 4, 2, 1) (mcvae.py:89,95)."""
    return F.conv_transpose2d(x, sd[key + '.weight'], sd[key + '.bias'], stride=2, padding=1)


def res_block(sd: State, p: str, x: Tensor, ind: Tensor, train: bool) -> Tensor:
    """ResBlock.forward (mcvae.py:17-35): conv-BN-ReLU-MC-conv-BN-MC, + input, ReLU."""
    h = _conv(sd, p + 'conv.0.module', x, 1, 1)
    h = torch.relu(batch_norm(sd, p + 'conv.1.module', h, train))
    h = mc_mask(h, ind, sd[p + 'conv.3.codebook'])
    h = _conv(sd, p + 'conv.4.module', h, 1, 1)
    h = batch_norm(sd, p + 'conv.5.module', h, train)
    h = mc_mask(h, ind, sd[p + 'conv.6.codebook'])
    return torch.relu(h + x)


def encode(sd: State, x: Tensor, ind: Tensor, train: bool, n_stage: int, n_res: int, eps: Optional[Tensor]):
    """Encoder.forward (mcvae.py:63-68): n_stage x (Conv4x4 s2 -> BN -> ReLU -> MC), ResBlocks, mu/logvar."""
    p = 'encoder.blocks.'
    for i in range(n_stage):
        x = _conv(sd, p + f'{4 * i}.module', x, 2, 1)
        x = torch.relu(batch_norm(sd, p + f'{4 * i + 1}.module', x, train))
        x = mc_mask(x, ind, sd[p + f'{4 * i + 3}.codebook'])
    for r in range(n_res):
        x = res_block(sd, p + f'{4 * n_stage + r}.', x, ind, train)
    flat = x.reshape(x.shape[0], -1)
    mu = F.linear(flat, sd['encoder.mu.weight'], sd['encoder.mu.bias'])
    logvar = F.linear(flat, sd['encoder.logvar.weight'], sd['encoder.logvar.bias'])
    z = mu + eps * torch.exp(0.5 * logvar) if train else mu           # reparameterize, mcvae.py:58-61,67
    return z, mu, logvar


def decode(sd: State, z: Tensor, ind: Tensor, train: bool, n_stage: int, n_res: int, encoded_shape: Sequence[int]):
    """Decoder.forward (mcvae.py:97-101)."""
    x = mc_mask(z, ind, sd['decoder.linear.0.codebook'])
    x = F.linear(x, sd['decoder.linear.1.module.weight'], sd['decoder.linear.1.module.bias'])
    x = torch.relu(batch_norm(sd, 'decoder.linear.2.module', x, train))            # BatchNorm1d
    x = x.reshape(x.shape[0], *encoded_shape)
    p = 'decoder.blocks.'
    x = mc_mask(x, ind, sd[p + '0.codebook'])
    for r in range(n_res):
        x = res_block(sd, p + f'{1 + r}.', x, ind, train)
    k = 1 + n_res
    for _ in range(n_stage - 1):
        x = _convt(sd, p + f'{k}.module', x)
        x = torch.relu(batch_norm(sd, p + f'{k + 1}.module', x, train))
        x = mc_mask(x, ind, sd[p + f'{k + 3}.codebook'])
        k += 4
    return torch.sigmoid(_convt(sd, p + f'{k}.module', x))


def vae_loss(img01: Tensor, recon: Tensor, mu: Tensor, logvar: Tensor) -> Tensor:
    """loss() (mcvae.py:10-14): (BCE_sum + KLD) / numel."""
    bce = F.binary_cross_entropy(recon, img01, reduction='sum')
    kld = 0.5 * torch.sum(mu.pow(2) + logvar.exp() - 1 - logvar)
    return (bce + kld) / img01.numel()


def forward(sd: State, img: Tensor, label: Tensor, classes: int, hidden: Sequence[int], n_res: int,
            train: bool = True, eps: Optional[Tensor] = None):
    """MCVAE.forward (mcvae.py:133-144): images come in (-1, 1), are mapped to (0, 1) for the BCE,
    and the reconstruction goes back out in (-1, 1)."""
    ind = one_hot(label, classes)
    x = (img + 1) / 2
    n_stage = len(hidden)
    enc_shape = (hidden[-1], img.shape[2] // 2 ** n_stage, img.shape[3] // 2 ** n_stage)
    z, mu, logvar = encode(sd, x, ind, train, n_stage, n_res, eps)
    recon = decode(sd, z, ind, train, n_stage, n_res, enc_shape)
    return {'loss': vae_loss(x, recon, mu, logvar), 'mu': mu, 'logvar': logvar, 'img': recon * 2 - 1}


def generate(sd: State, label: Tensor, z: Tensor, classes: int, hidden: Sequence[int], n_res: int, side: int = 32):
    """MCVAE.generate (mcvae.py:124-131), eval-mode decode."""
    n_stage = len(hidden)
    enc_shape = (hidden[-1], side // 2 ** n_stage, side // 2 ** n_stage)
    return decode(sd, z, one_hot(label, classes), False, n_stage, n_res, enc_shape) * 2 - 1
