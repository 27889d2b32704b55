"""CPU restatement (plain fp32 torch ops) of the reference VQ-VAE's inference path (models/vqvae.py,
modules/modules.py:6-46): the frozen encoder + vector quantiser that turns images into the code maps MCPixelCNN
trains on (train_pixelcnn.py:111-113), and ``decode_code``.

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``."""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

from .mcgan_oracle import batch_norm

Tensor = torch.Tensor
State = Dict[str, Tensor]


def _bn(sd: State, p: str, x: Tensor) -> Tensor:
    return batch_norm(sd, p, x, False)                     # the auto-encoder is frozen: eval-mode statistics


def _res(sd: State, p: str, x: Tensor) -> Tensor:
    """ResBlock.forward (vqvae.py:21-24)."""
    h = F.conv2d(x, sd[p + 'conv.0.weight'], sd[p + 'conv.0.bias'], padding=1)
    h = torch.relu(_bn(sd, p + 'conv.1', h))
    h = F.conv2d(h, sd[p + 'conv.3.weight'], sd[p + 'conv.3.bias'], padding=1)
    return torch.relu(_bn(sd, p + 'conv.4', h) + x)


def encoder(sd: State, img: Tensor, n_stage: int, n_res: int) -> Tensor:
    """Encoder.forward (vqvae.py:27-47)."""
    p = 'encoder.blocks.'
    x = img
    for i in range(n_stage):
        x = F.conv2d(x, sd[p + f'{3 * i}.weight'], sd[p + f'{3 * i}.bias'], stride=2, padding=1)
        x = torch.relu(_bn(sd, p + f'{3 * i + 1}', x))
    k = 3 * n_stage
    for r in range(n_res):
        x = _res(sd, p + f'{k + r}.', x)
    k += n_res
    return F.conv2d(x, sd[p + f'{k}.weight'], sd[p + f'{k}.bias'], padding=1)


def quantize(sd: State, x: Tensor):
    """VectorQuantization.forward in eval mode (modules.py:18-42): note transpose(1, -1) -- the code map comes out
    as [N, W, H].  Returns (quantised NCHW, mse, code, squared distances [N*W*H, K])."""
    emb = sd['quantizer.embedding']
    inp = x.transpose(1, -1).contiguous()
    flat = inp.view(-1, emb.shape[0])
    dist = flat.pow(2).sum(1, keepdim=True) - 2 * flat @ emb + emb.pow(2).sum(0, keepdim=True)
    ind = dist.min(1)[1].view(*inp.shape[:-1])
    q = F.embedding(ind, emb.t())
    return q.transpose(1, -1).contiguous(), F.mse_loss(q, inp), ind, dist


def encode(sd: State, img: Tensor, n_stage: int, n_res: int):
    """VQVAE.encode (vqvae.py:92-95)."""
    return quantize(sd, encoder(sd, img, n_stage, n_res))


def decode_code(sd: State, code: Tensor, n_stage: int, n_res: int) -> Tensor:
    """VQVAE.decode_code + Decoder.forward (vqvae.py:50-75, 101-104)."""
    x = F.embedding(code, sd['quantizer.embedding'].t()).transpose(1, -1).contiguous()
    p = 'decoder.blocks.'
    x = F.conv2d(x, sd[p + '0.weight'], sd[p + '0.bias'], padding=1)
    x = torch.relu(_bn(sd, p + '1', x))
    k = 3
    for r in range(n_res):
        x = _res(sd, p + f'{k + r}.', x)
    k += n_res
    for _ in range(n_stage - 1):
        x = F.conv_transpose2d(x, sd[p + f'{k}.weight'], sd[p + f'{k}.bias'], stride=2, padding=1)
        x = torch.relu(_bn(sd, p + f'{k + 1}', x))
        k += 3
    return torch.tanh(F.conv_transpose2d(x, sd[p + f'{k}.weight'], sd[p + f'{k}.bias'], stride=2, padding=1))
