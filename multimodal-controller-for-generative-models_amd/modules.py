"""MultimodalController / Wrapper with the reference's module surface
(reference: src/modules/modules.py:49-85), computed by HIP kernels.

``MultimodalController`` keeps the reference contract that the codebook is a
LIVE buffer: ``models.utils.create`` / ``transit`` re-register it (possibly with
a different number of modes), and every forward -- standalone or from inside a
fused block -- reads ``self.codebook`` at call time.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops


def sample_codebook(num_mode: int, input_size: int, rate: float) -> torch.Tensor:
    """Distinct Bernoulli(rate) rows, all-ones when rate == 1 (modules.py:58-69).

    The reference keeps the first ``num_mode`` rows of a Python ``set`` (hash
    order), so its row order is not reproducible from a seed; parity runs load
    the reference's ``codebook`` buffers instead.  Here rows
    keep their sampling order.
    """
    if rate == 1:
        return torch.ones(num_mode, input_size)
    rows, seen = [], set()
    while len(rows) < num_mode:
        for r in torch.bernoulli(torch.full((num_mode, input_size), float(rate))).tolist():
            t = tuple(r)
            if t not in seen and len(rows) < num_mode:
                seen.add(t)
                rows.append(r)
    return torch.tensor(rows, dtype=torch.float)


class _MaskFn(torch.autograd.Function):
    """x * code with the code treated as a constant (code.detach(), modules.py:75)."""

    @staticmethod
    def forward(ctx, x, code):
        ctx.save_for_backward(code)
        return ops.mc_apply(x.contiguous(), code, channels_last=False)

    @staticmethod
    def backward(ctx, g):
        (code,) = ctx.saved_tensors
        return ops.mc_apply(g.contiguous(), code, channels_last=False), None


class MultimodalController(nn.Module):
    def __init__(self, input_size, num_mode, controller_rate=0.5):
        super().__init__()
        self.input_size = input_size
        self.num_mode = num_mode
        self.controller_rate = controller_rate
        self.register_buffer('codebook', self.make_codebook())

    def make_codebook(self):
        return sample_codebook(self.num_mode, self.input_size, self.controller_rate)

    def code(self, indicator: torch.Tensor) -> torch.Tensor:
        """[N, C] = indicator @ codebook, from the buffer as it is right now."""
        return ops.mc_code(indicator, self.codebook)

    def code_of_labels(self, label: torch.Tensor) -> torch.Tensor:
        """The same [N, C] code for one-hot indicators, as a row gather: one_hot(label) @ codebook == codebook[label]
        exactly.  Used by the fused model paths, which see the labels before they are one-hot encoded."""
        return self.codebook.index_select(0, label)

    def forward(self, input):
        # list protocol of the reference: [x, indicator, ...] -> [x * code, indicator, ...]
        x, indicator = input[0], input[1]
        if x.dtype != torch.float32:
            raise TypeError('MultimodalController expects float32 activations at the module boundary')
        return [_MaskFn.apply(x, self.code(indicator)), *input[1:]]

    def extra_repr(self):
        return f'{self.input_size}, num_mode={self.num_mode}, rate={self.controller_rate}'


class Wrapper(nn.Module):
    """Lifts a tensor module onto the [x, indicator, ...] list protocol (modules.py:79-85)."""

    def __init__(self, module):
        super().__init__()
        self.module = module

    def forward(self, input):
        return [self.module(input[0]), *input[1:]]


class VectorQuantization(nn.Module):
    """modules.py:6-46 -- the EMA codebook of the VQ-VAE.  Buffers as in the reference (``embedding`` [D, K],
    ``cluster_size`` [K], ``embedding_mean`` [D, K]).  The HIP path implements the inference side (nearest code +
    gather), which is what runs in front of MCPixelCNN (train_pixelcnn.py:111-113); the EMA update of a VQ-VAE
    training run is not built."""

    def __init__(self, embedding_size, num_embedding, decay=0.99, eps=1e-5):
        super().__init__()
        self.embedding_size, self.num_embedding, self.decay, self.eps = embedding_size, num_embedding, decay, eps
        embedding = torch.randn(embedding_size, num_embedding)
        self.register_buffer('embedding', embedding)
        self.register_buffer('cluster_size', torch.zeros(num_embedding))
        self.register_buffer('embedding_mean', embedding.clone())

    def embedding_code(self, embedding_ind: torch.Tensor) -> torch.Tensor:
        return torch.nn.functional.embedding(embedding_ind, self.embedding.transpose(0, 1))

    def nearest(self, feat_nhwc: torch.Tensor) -> torch.Tensor:
        """[N, H, W, D] fp32 features -> int64 code indices [N, H, W]: argmin_k |f - e_k|^2 = argmin_k (|e_k|^2 - 2 f.e_k),
        one fused 1x1 convolution over the codebook + the arg-min kernel."""
        if self.training:
            raise NotImplementedError('VectorQuantization: only the inference path (frozen codebook) is built')
        e = self.embedding                                                   # [D, K]
        w = (-2.0 * e.t()).contiguous().reshape(self.num_embedding, self.embedding_size, 1, 1)
        dist, _ = ops.conv_fused([ops.Seg(feat_nhwc, ksize=1)], ops.prep_weight(w, feat_nhwc.dtype), self.num_embedding,
                                 bias=e.pow(2).sum(0))
        return ops.argmin_channels(dist, self.num_embedding)
