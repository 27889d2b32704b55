"""VQ-VAE with the reference's module surface (src/models/vqvae.py), inference side on the HIP kernels: ``encode``
(images -> quantised features, commitment MSE, code map) is the step in front of every MCPixelCNN iteration
(train_pixelcnn.py:111-113) and ``decode_code`` turns sampled code maps back into images (generate.py).
The auto-encoder is frozen there (``ae.train(False)``): BatchNorm uses running statistics and the codebook is fixed.
Training the VQ-VAE itself (EMA codebook update, straight-through gradient) is not built."""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ..config import cfg
from ..modules import VectorQuantization
from ..ops import Seg, pad8
from .utils import init_param


class ResBlock(nn.Module):
    """vqvae.py:9-24."""

    def __init__(self, hidden_size):
        super().__init__()
        self.activation = nn.ReLU(inplace=True)
        self.conv = nn.Sequential(nn.Conv2d(hidden_size, hidden_size, 3, 1, 1), nn.BatchNorm2d(hidden_size), nn.ReLU(inplace=True),
                                  nn.Conv2d(hidden_size, hidden_size, 3, 1, 1), nn.BatchNorm2d(hidden_size))


class Encoder(nn.Module):
    """vqvae.py:27-47."""

    def __init__(self, data_shape, hidden_size, num_res_block, embedding_size):
        super().__init__()
        blocks, cin = [], data_shape[0]
        for h in hidden_size:
            blocks.extend([nn.Conv2d(cin, h, 4, 2, 1), nn.BatchNorm2d(h), nn.ReLU(inplace=True)])
            cin = h
        for _ in range(num_res_block):
            blocks.append(ResBlock(hidden_size[-1]))
        blocks.append(nn.Conv2d(hidden_size[-1], embedding_size, 3, 1, 1))
        self.blocks = nn.Sequential(*blocks)


class Decoder(nn.Module):
    """vqvae.py:50-75."""

    def __init__(self, data_shape, hidden_size, num_res_block, embedding_size):
        super().__init__()
        blocks = [nn.Conv2d(embedding_size, hidden_size[-1], 3, 1, 1), nn.BatchNorm2d(hidden_size[-1]), nn.ReLU(inplace=True)]
        for _ in range(num_res_block):
            blocks.append(ResBlock(hidden_size[-1]))
        for i in range(len(hidden_size) - 1, 0, -1):
            blocks.extend([nn.ConvTranspose2d(hidden_size[i], hidden_size[i - 1], 4, 2, 1), nn.BatchNorm2d(hidden_size[i - 1]),
                           nn.ReLU(inplace=True)])
        blocks.extend([nn.ConvTranspose2d(hidden_size[0], data_shape[0], 4, 2, 1), nn.Tanh()])
        self.blocks = nn.Sequential(*blocks)


def _affine(bn):
    return ops.bn_eval_affine(bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var, bn.eps)


class VQVAE(nn.Module):
    """vqvae.py:78-114."""

    def __init__(self, data_shape=(3, 32, 32), hidden_size=(128, 128), num_res_block=2, embedding_size=64,
                 num_embedding=512, vq_commit=0.25):
        super().__init__()
        self.data_shape, self.hidden_size, self.num_res_block = data_shape, hidden_size, num_res_block
        self.embedding_size, self.vq_commit = embedding_size, vq_commit
        if any(h % 8 for h in hidden_size) or embedding_size % 8:
            raise ValueError('Not valid hidden/embedding size: the fused path needs multiples of 8')
        self.encoder = Encoder(data_shape, hidden_size, num_res_block, embedding_size)
        self.quantizer = VectorQuantization(embedding_size, num_embedding)
        self.decoder = Decoder(data_shape, hidden_size, num_res_block, embedding_size)

    def set_compute_dtype(self, dtype):
        self.__dict__['_cdt'] = dtype
        return self

    def _dt(self):
        return self.__dict__.get('_cdt') or {'float32': torch.float32, 'bfloat16': torch.bfloat16}[cfg.get('compute_dtype', 'float32')]

    def _frozen(self):
        if self.training:
            raise NotImplementedError('VQVAE: only the frozen (eval-mode) encode / decode_code paths are built')

    # ---- fused building blocks (eval-mode BatchNorm folded into prologues) -------------------------------------
    def _res(self, blk, x):
        """relu(BN(conv(relu(BN(conv(x))))) + x)  (vqvae.py:21-24)."""
        dt = x.dtype
        c0, b1, c3, b4 = blk.conv[0], blk.conv[1], blk.conv[3], blk.conv[4]
        h1, _ = ops.conv_fused([Seg(x)], ops.prep_weight(c0.weight.detach(), dt), c0.out_channels, bias=c0.bias.detach())
        s1, t1 = _affine(b1)
        h2, _ = ops.conv_fused([Seg(h1, scale=s1, shift=t1, relu=True)], ops.prep_weight(c3.weight.detach(), dt), c3.out_channels,
                               bias=c3.bias.detach())
        s4, t4 = _affine(b4)
        return ops.affine_code_res(h2, s4, t4, None, x, post_relu=True)

    def _encoder(self, img):
        dt = self._dt()
        blocks = self.encoder.blocks
        x = ops.to_nhwc(img.contiguous(), dt)
        ns = len(self.hidden_size)
        prol = {}
        for i in range(ns):
            conv, bn = blocks[3 * i], blocks[3 * i + 1]
            cp = x.shape[-1]
            col = ops.im2col(x, 4, 4, 1, 1, stride=2, **prol)                    # previous stage's BN + ReLU applied here
            w = conv.weight.detach().permute(0, 2, 3, 1)
            wm = F.pad(w, (0, cp - w.shape[-1])).reshape(w.shape[0], 16 * cp, 1, 1).contiguous()
            x, _ = ops.conv_fused([Seg(col, ksize=1)], ops.prep_weight(wm, dt), conv.out_channels, bias=conv.bias.detach())
            s, t = _affine(bn)
            prol = dict(scale=s, shift=t, relu=True)
        x = ops.affine_code_res(x, prol['scale'], prol['shift'], None, None, pre_relu=True)   # materialise for the residual
        k = 3 * ns
        for r in range(self.num_res_block):
            x = self._res(blocks[k + r], x)
        last = blocks[k + self.num_res_block]
        out, _ = ops.conv_fused([Seg(x)], ops.prep_weight(last.weight.detach(), dt), last.out_channels, bias=last.bias.detach())
        return out

    def encode(self, input):
        """-> (quantised features NCHW, commitment MSE, code map [N, W, H]) as the reference returns them
        (modules.py:18-42: the quantiser works on ``input.transpose(1, -1)``, so the code map is spatially transposed)."""
        self._frozen()
        with torch.no_grad():
            feat = self._encoder(input).float()                                  # [N, H, W, D]; nearest-code search in fp32
            d = self.embedding_size
            idx = self.quantizer.nearest(feat[..., :d].contiguous() if feat.shape[-1] != d else feat)
            q = self.quantizer.embedding_code(idx)                               # [N, H, W, D]
            diff = F.mse_loss(q, feat[..., :d])
            return q.permute(0, 3, 1, 2).contiguous(), diff, idx.transpose(1, 2).contiguous()

    def decode_code(self, code):
        """code map [N, W, H] (as `encode` returns it) -> images in (-1, 1)  (vqvae.py:101-104)."""
        self._frozen()
        dt = self._dt()
        with torch.no_grad():
            x = self.quantizer.embedding_code(code.transpose(1, 2)).to(dt).contiguous()       # NHWC [N, H, W, D]
            blocks = self.decoder.blocks
            c0, b1 = blocks[0], blocks[1]
            h, _ = ops.conv_fused([Seg(x)], ops.prep_weight(c0.weight.detach(), dt), c0.out_channels, bias=c0.bias.detach())
            s, t = _affine(b1)
            x = ops.affine_code_res(h, s, t, None, None, pre_relu=True)
            k = 3
            for r in range(self.num_res_block):
                x = self._res(blocks[k + r], x)
            k += self.num_res_block
            prol = None
            for stage in range(len(self.hidden_size)):
                convt = blocks[k]
                co, cop = convt.out_channels, pad8(convt.out_channels)
                w = convt.weight.detach().permute(2, 3, 1, 0)
                wm = F.pad(w, (0, 0, 0, cop - co)).reshape(16 * cop, w.shape[-1], 1, 1).contiguous()
                seg = Seg(x, ksize=1) if prol is None else Seg(x, ksize=1, scale=prol[0], shift=prol[1], relu=True)
                dcol, _ = ops.conv_fused([seg], ops.prep_weight(wm, dt), 16 * cop)
                x = ops.col2im(dcol, cop, 4, 4, 1, 1, stride=2, bias=convt.bias.detach())
                if stage < len(self.hidden_size) - 1:
                    prol = _affine(blocks[k + 1])                                # BN + ReLU ride in the next 1x1's prologue
                    k += 3
            return torch.tanh(ops.to_nchw(x, self.data_shape[0]))

    def forward(self, input):
        raise NotImplementedError('VQVAE.forward (auto-encoder training) is not on the MultimodalController hot path')


def vqvae():
    v = cfg['vqvae']
    model = VQVAE(data_shape=cfg['data_shape'], hidden_size=v['hidden_size'], num_res_block=v['num_res_block'],
                  embedding_size=v['embedding_size'], num_embedding=v['num_embedding'], vq_commit=v['vq_commit'])
    model.apply(init_param)
    return model
