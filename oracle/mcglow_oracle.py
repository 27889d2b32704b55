"""CPU restatement (plain fp32 torch ops) of the reference MCGlow (models/mcglow.py), affine coupling +
LU-parameterised invertible 1x1 convolution (the configuration process_control selects, utils.py:183-184).

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.  The dequantisation noise (mcglow.py:299,
``torch.rand_like``) is an explicit input."""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

from .mcgan_oracle import mc_mask, one_hot

Tensor = torch.Tensor
State = Dict[str, Tensor]


def actnorm(sd: State, p: str, x: Tensor, train: bool):
    """ActNorm.forward (mcglow.py:41-51) incl. the data-dependent initialisation on the first
    training-mode call (mcglow.py:32-39: loc = -mean, scale = 1 / (unbiased std + 1e-6))."""
    if int(sd[p + 'initialized']) == 0 and train:
        with torch.no_grad():
            mean = x.mean(dim=(0, 2, 3), keepdim=True)
            std = x.std(dim=(0, 2, 3), keepdim=True)
            sd[p + 'loc'].copy_(-mean)
            sd[p + 'scale'].copy_(1 / (std + 1e-6))
            sd[p + 'initialized'].fill_(1)
    scale, loc = sd[p + 'scale'], sd[p + 'loc']
    logdet = x.shape[2] * x.shape[3] * torch.sum(torch.log(torch.abs(scale)))
    return scale * (x + loc), logdet


def invconv_lu_weight(sd: State, p: str) -> Tensor:
    """InvConv2dLU.calc_weight (mcglow.py:105-111): P (L o mask + I) (U o mask + diag(sign * exp(w_s)))."""
    lower = sd[p + 'w_l'] * sd[p + 'l_mask'] + sd[p + 'l_eye']
    upper = sd[p + 'w_u'] * sd[p + 'u_mask'] + torch.diag(sd[p + 's_sign'] * torch.exp(sd[p + 'w_s']))
    return sd[p + 'w_p'] @ lower @ upper


def zero_conv(sd: State, p: str, x: Tensor) -> Tensor:
    """ZeroConv2d.forward (mcglow.py:127-130)."""
    out = F.conv2d(x, sd[p + 'conv.weight'], sd[p + 'conv.bias'], padding=1)
    return out * torch.exp(sd[p + 'scale'] * 3)


def coupling_net(sd: State, p: str, x: Tensor, ind: Tensor, train: bool) -> Tensor:
    """AffineCoupling.net (mcglow.py:137-147): Conv3 -> ActNorm -> ReLU -> MC -> Conv1 -> ActNorm -> ReLU -> MC -> ZeroConv3."""
    h = F.conv2d(x, sd[p + '0.module.weight'], sd[p + '0.module.bias'], padding=1)
    h = mc_mask(torch.relu(actnorm(sd, p + '1.module.', h, train)[0]), ind, sd[p + '3.codebook'])
    h = F.conv2d(h, sd[p + '4.module.weight'], sd[p + '4.module.bias'])
    h = mc_mask(torch.relu(actnorm(sd, p + '5.module.', h, train)[0]), ind, sd[p + '7.codebook'])
    return zero_conv(sd, p + '8.module.', h)


def flow_forward(sd: State, p: str, x: Tensor, ind: Tensor, train: bool):
    """Flow.forward (mcglow.py:188-195)."""
    out, logdet = actnorm(sd, p + 'actnorm.', x, train)
    w = invconv_lu_weight(sd, p + 'invconv.')
    out = F.conv2d(out, w[:, :, None, None])
    logdet = logdet + x.shape[2] * x.shape[3] * torch.sum(sd[p + 'invconv.w_s'])
    in_a, in_b = out.chunk(2, 1)
    log_s, t = coupling_net(sd, p + 'coupling.net.', in_a, ind, train).chunk(2, 1)
    s = torch.sigmoid(log_s + 2)
    out = torch.cat([in_a, (in_b + t) * s], 1)
    return out, logdet + torch.log(s).reshape(x.shape[0], -1).sum(1)


def flow_reverse(sd: State, p: str, y: Tensor, ind: Tensor):
    """Flow.reverse (mcglow.py:197-201)."""
    out_a, out_b = y.chunk(2, 1)
    log_s, t = coupling_net(sd, p + 'coupling.net.', out_a, ind, False).chunk(2, 1)
    x = torch.cat([out_a, out_b / torch.sigmoid(log_s + 2) - t], 1)
    w = invconv_lu_weight(sd, p + 'invconv.')
    x = F.conv2d(x, torch.inverse(w)[:, :, None, None])
    return x / sd[p + 'actnorm.scale'] - sd[p + 'actnorm.loc']


def gaussian_log_p(x, mean, log_sd):
    return -0.5 * math.log(2 * math.pi) - log_sd - 0.5 * (x - mean) ** 2 / torch.exp(2 * log_sd)


def squeeze(x: Tensor) -> Tensor:
    b, c, h, w = x.shape
    return x.reshape(b, c, h // 2, 2, w // 2, 2).permute(0, 1, 3, 5, 2, 4).reshape(b, c * 4, h // 2, w // 2)


def unsqueeze(x: Tensor) -> Tensor:
    b, c, h, w = x.shape
    return x.reshape(b, c // 4, 2, 2, h, w).permute(0, 1, 4, 2, 5, 3).reshape(b, c // 4, h * 2, w * 2)


def block_forward(sd: State, p: str, x: Tensor, ind: Tensor, train: bool, K: int, split: bool):
    """Block.forward (mcglow.py:219-240)."""
    out = squeeze(x)
    logdet = 0
    for k in range(K):
        out, det = flow_forward(sd, p + f'flows.{k}.', out, ind, train)
        logdet = logdet + det
    if split:
        out, z_new = out.chunk(2, 1)
        mean, log_sd = zero_conv(sd, p + 'prior.', out).chunk(2, 1)
        log_p = gaussian_log_p(z_new, mean, log_sd).reshape(x.shape[0], -1).sum(1)
    else:
        mean, log_sd = zero_conv(sd, p + 'prior.', torch.zeros_like(out)).chunk(2, 1)
        log_p = gaussian_log_p(out, mean, log_sd).reshape(x.shape[0], -1).sum(1)
        z_new = out
    return out, logdet, log_p, z_new


def forward(sd: State, img: Tensor, label: Tensor, classes: int, K: int, L: int, noise: Tensor, train: bool = True):
    """MCGlow.forward + loss_fn (mcglow.py:283-312): bits per dimension, NaN per-sample losses zeroed in training."""
    ind = one_hot(label, classes)
    x = img * 0.5 + noise / 256
    n_pixel = img[0].numel()
    zs: List[Tensor] = []
    log_p_sum, logdet = 0, 0
    for i in range(L):
        x, det, log_p, z_new = block_forward(sd, f'blocks.{i}.', x, ind, train, K, split=(i < L - 1))
        zs.append(z_new)
        logdet = logdet + det
        log_p_sum = log_p_sum + log_p
    loss = -(-math.log(256.) * n_pixel + logdet + log_p_sum) / (math.log(2.) * n_pixel)
    loss = torch.where(torch.isnan(loss), torch.zeros_like(loss), loss) if train else loss[~torch.isnan(loss)]
    return {'loss': loss.mean(), 'z': zs}


def reverse(sd: State, zs: List[Tensor], label: Tensor, classes: int, K: int, L: int, reconstruct: bool):
    """MCGlow.reverse / Block.reverse (mcglow.py:242-265, 314-325)."""
    ind = one_hot(label, classes)
    x: Optional[Tensor] = None
    for i in reversed(range(L)):
        p = f'blocks.{i}.'
        split = i < L - 1
        eps = zs[i]
        if reconstruct:
            inp = torch.cat([x, eps], 1) if split else eps
        elif split:
            mean, log_sd = zero_conv(sd, p + 'prior.', x).chunk(2, 1)
            inp = torch.cat([x, mean + torch.exp(log_sd) * eps], 1)
        else:
            mean, log_sd = zero_conv(sd, p + 'prior.', torch.zeros_like(eps)).chunk(2, 1)
            inp = mean + torch.exp(log_sd) * eps
        for k in reversed(range(K)):
            inp = flow_reverse(sd, p + f'flows.{k}.', inp, ind)
        x = unsqueeze(inp)
    return torch.clamp(x, -.5, .5) * 2
