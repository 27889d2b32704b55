"""MCVAE with the reference's module surface (src/models/mcvae.py): strided-conv encoder / transposed-conv decoder
with a MultimodalController after every activation and on the latent.  The module tree carries the reference's
parameter / buffer names (``state_dict`` compatible); the arithmetic runs in ``vae_engine.py`` on HIP kernels."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from ..config import cfg
from ..modules import MultimodalController, Wrapper
from ..vae_engine import VAEEngine
from .utils import init_param


class _VAEFn(torch.autograd.Function):
    """One autograd node for the whole model: only the loss carries gradient (train_vae.py:107-109)."""

    @staticmethod
    def forward(ctx, engine, img, label, eps, holder, *params):
        tape = []
        out = engine.forward(img, label, True, eps, tape, want_grad=True)
        holder.update(out)
        ctx.engine, ctx.tape, ctx.params, ctx.label = engine, tape, params, label
        return out['loss']

    @staticmethod
    def backward(ctx, gloss):
        eng = ctx.engine
        sink = {}
        eng._gsink = sink
        try:
            eng.backward(ctx.tape, ctx.label)
        finally:
            eng._gsink = None
        ctx.tape = None
        return (None, None, None, None, None) + tuple(sink[id(p)] * gloss if id(p) in sink else None for p in ctx.params)


def _mc(width, modes, rate):
    return MultimodalController(width, modes, rate)


def _wrapped(*layers):
    """Plain torch layers inside the reference's list-protocol ``Wrapper`` (parameter containers only here)."""
    return [Wrapper(layer) for layer in layers]


def _stage(conv, width, modes, rate):
    """conv -> BatchNorm2d -> ReLU -> MC: the four entries one encoder / decoder stage adds to ``blocks``."""
    return _wrapped(conv, nn.BatchNorm2d(width), nn.ReLU(inplace=True)) + [_mc(width, modes, rate)]


def _encoded_shape(data_shape, widths):
    shrink = 2 ** len(widths)
    return (widths[-1], data_shape[1] // shrink, data_shape[2] // shrink)


class ResBlock(nn.Module):
    """mcvae.py:17-35 -- ``conv`` holds conv, BN, ReLU, MC, conv, BN, MC (indices 0..6); the skip + ReLU live in
    the fused tail kernel."""

    def __init__(self, hidden_size, num_mode, controller_rate):
        super().__init__()
        w = hidden_size
        first = _stage(nn.Conv2d(w, w, 3, 1, 1), w, num_mode, controller_rate)
        second = _wrapped(nn.Conv2d(w, w, 3, 1, 1), nn.BatchNorm2d(w)) + [_mc(w, num_mode, controller_rate)]
        self.conv = nn.Sequential(*(first + second))
        self.activation = Wrapper(nn.ReLU(inplace=True))


class Encoder(nn.Module):
    """mcvae.py:38-68 -- ``blocks``: len(hidden) strided stages, then the residual blocks; ``mu`` / ``logvar`` heads."""

    def __init__(self, data_shape, hidden_size, latent_size, num_res_block, num_mode, controller_rate):
        super().__init__()
        layers, width_in = [], data_shape[0]
        for width in hidden_size:
            layers += _stage(nn.Conv2d(width_in, width, 4, 2, 1), width, num_mode, controller_rate)
            width_in = width
        layers += [ResBlock(width_in, num_mode, controller_rate) for _ in range(num_res_block)]
        self.blocks = nn.Sequential(*layers)
        self.encoded_shape = _encoded_shape(data_shape, hidden_size)
        features = int(np.prod(self.encoded_shape))
        self.mu, self.logvar = nn.Linear(features, latent_size), nn.Linear(features, latent_size)


class Decoder(nn.Module):
    """mcvae.py:71-101 -- ``linear``: MC, Linear, BatchNorm1d, ReLU; ``blocks``: MC, residual blocks, transposed-conv
    stages up to the image, Sigmoid."""

    def __init__(self, data_shape, hidden_size, latent_size, num_res_block, num_mode, controller_rate):
        super().__init__()
        self.encoded_shape = _encoded_shape(data_shape, hidden_size)
        features = int(np.prod(self.encoded_shape))
        self.linear = nn.Sequential(_mc(latent_size, num_mode, controller_rate),
                                    *_wrapped(nn.Linear(latent_size, features), nn.BatchNorm1d(features), nn.ReLU(inplace=True)))
        top = hidden_size[-1]
        layers = [_mc(top, num_mode, controller_rate)] + [ResBlock(top, num_mode, controller_rate) for _ in range(num_res_block)]
        for wide, narrow in zip(reversed(hidden_size[1:]), reversed(hidden_size[:-1])):
            layers += _stage(nn.ConvTranspose2d(wide, narrow, 4, 2, 1), narrow, num_mode, controller_rate)
        layers += _wrapped(nn.ConvTranspose2d(hidden_size[0], data_shape[0], 4, 2, 1), nn.Sigmoid())
        self.blocks = nn.Sequential(*layers)


class MCVAE(nn.Module):
    """mcvae.py:104-144."""

    def __init__(self, data_shape=(3, 32, 32), hidden_size=(64, 128, 256), latent_size=128, num_res_block=2,
                 num_mode=None, controller_rate=0.5):
        super().__init__()
        self.data_shape, self.hidden_size, self.latent_size = data_shape, hidden_size, latent_size
        self.num_res_block, self.num_mode, self.controller_rate = num_res_block, num_mode, controller_rate
        self.encoder = Encoder(data_shape, hidden_size, latent_size, num_res_block, num_mode, controller_rate)
        self.decoder = Decoder(data_shape, hidden_size, latent_size, num_res_block, num_mode, controller_rate)

    def _engine(self):
        eng = self.__dict__.get('_eng')
        dt = {'float32': torch.float32, 'bfloat16': torch.bfloat16}[cfg.get('compute_dtype', 'float32')]
        dt = self.__dict__.get('_cdt') or dt
        if eng is None or eng.dtype != dt:
            eng = VAEEngine(self, dt)
            self.__dict__['_eng'] = eng
        return eng

    def set_compute_dtype(self, dtype):
        self.__dict__['_cdt'] = dtype
        return self

    @staticmethod
    def _label(label):
        if label.dtype != torch.int64 or label.dim() != 1:
            raise ValueError('Not valid label: expected an int64 vector of class indices')
        return label

    def generate(self, C, z=None):
        """Eval-style decode of a latent (mcvae.py:124-131) -> images in (-1, 1)."""
        if z is None:
            z = torch.randn([C.size(0), self.latent_size], device=cfg['device'])
        from .. import ops
        with torch.no_grad():
            logits = self._engine().decode(z, self._label(C), self.training, None)
        return torch.sigmoid(ops.to_nchw(logits, self.data_shape[0])) * 2 - 1

    def forward(self, input):
        """{'img' in (-1,1), 'label'[, 'eps']} -> {'loss', 'mu', 'logvar', 'img'} (mcvae.py:133-144); `eps` injects
        the reparameterisation noise (parity runs), otherwise it is drawn here."""
        eng = self._engine()
        label = self._label(input['label'])
        eps = input.get('eps')
        if torch.is_grad_enabled() and self.training:
            if eps is None:
                eps = torch.randn(input['img'].shape[0], self.latent_size, device=input['img'].device)
            holder = {}
            params = [p for p in self.parameters() if p.requires_grad]
            loss = _VAEFn.apply(eng, input['img'], label, eps, holder, *params)
            return {'loss': loss, 'mu': holder['mu'], 'logvar': holder['logvar'], 'img': holder['img']}
        return eng.forward(input['img'], label, self.training, eps)


def mcvae():
    v = cfg['vae']
    model = MCVAE(data_shape=cfg['data_shape'], hidden_size=v['hidden_size'], latent_size=v['latent_size'],
                  num_res_block=v['num_res_block'], num_mode=cfg['classes_size'], controller_rate=cfg['controller_rate'])
    model.apply(init_param)
    return model
