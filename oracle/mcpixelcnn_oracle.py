"""CPU restatement (plain fp32 torch ops) of the reference MCGatedPixelCNN (models/mcpixelcnn.py).

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``."""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

from .mcgan_oracle import batch_norm, mc_mask, one_hot

Tensor = torch.Tensor
State = Dict[str, Tensor]


def gated_activation(sd: State, p: str, x: Tensor, ind: Tensor, train: bool) -> Tensor:
    """MCGatedActivation.forward (mcpixelcnn.py:16-20): BN+ReLU on the first half, sigmoid gate from the
    second half, then the MultimodalController mask."""
    a, b = x.chunk(2, dim=1)
    a = torch.relu(batch_norm(sd, p + 'bn', a, train))
    return mc_mask(a * torch.sigmoid(b), ind, sd[p + 'mc.codebook'])


def gated_masked_layer(sd: State, p: str, x_v: Tensor, x_h: Tensor, ind: Tensor, train: bool,
                       kernel: int, mask_a: bool, residual: bool):
    """MCGatedMaskedConv2d.forward (mcpixelcnn.py:47-61).  Mask 'A' zeroes the last kernel row of the
    vertical stack and the last kernel column of the horizontal stack IN PLACE on every forward
    (make_causal, mcpixelcnn.py:43-45) -- the state dict is edited the same way here."""
    wv, wh = sd[p + 'vert_stack.weight'], sd[p + 'horiz_stack.weight']
    if mask_a:
        with torch.no_grad():
            wv[:, :, -1].zero_()
            wh[:, :, :, -1].zero_()
    k2 = kernel // 2
    h_vert = F.conv2d(x_v, wv, sd[p + 'vert_stack.bias'], padding=(k2, k2))[:, :, :x_v.size(-1), :]
    out_v = gated_activation(sd, p + 'gate_v.', h_vert, ind, train)
    h_horiz = F.conv2d(x_h, wh, sd[p + 'horiz_stack.bias'], padding=(0, k2))[:, :, :, :x_h.size(-2)]
    v2h = F.conv2d(h_vert, sd[p + 'vert_to_horiz.weight'], sd[p + 'vert_to_horiz.bias'])
    out_h = gated_activation(sd, p + 'gate_h.', v2h + h_horiz, ind, train)
    r = F.conv2d(out_h, sd[p + 'horiz_resid.0.module.weight'], sd[p + 'horiz_resid.0.module.bias'])
    r = mc_mask(batch_norm(sd, p + 'horiz_resid.1.module', r, train), ind, sd[p + 'horiz_resid.2.codebook'])
    return out_v, (r + x_h if residual else r)


def forward(sd: State, codes: Tensor, label: Tensor, classes: int, train: bool = True):
    """MCGatedPixelCNN.forward (mcpixelcnn.py:89-101): embed the code map, 1 mask-A 7x7 layer + mask-B 3x3
    layers, 1x1 head, cross-entropy against the input codes."""
    ind = one_hot(label, classes)
    x = F.embedding(codes.reshape(-1), sd['embedding.weight']).reshape(*codes.shape, -1).permute(0, 3, 1, 2).contiguous()
    n_layer = 1 + max(int(k.split('.')[1]) for k in sd if k.startswith('layers.'))
    x_v = x_h = x
    for i in range(n_layer):
        x_v, x_h = gated_masked_layer(sd, f'layers.{i}.', x_v, x_h, ind, train, kernel=7 if i == 0 else 3,
                                      mask_a=(i == 0), residual=(i != 0))
    h = F.conv2d(x_h, sd['output_conv.0.module.weight'], sd['output_conv.0.module.bias'])
    h = torch.relu(batch_norm(sd, 'output_conv.1.module', h, train))
    h = mc_mask(h, ind, sd['output_conv.3.codebook'])
    logits = F.conv2d(h, sd['output_conv.4.module.weight'], sd['output_conv.4.module.bias'])
    return {'logits': logits, 'loss': F.cross_entropy(logits, codes)}
