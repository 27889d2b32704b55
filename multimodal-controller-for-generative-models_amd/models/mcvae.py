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


class ResBlock(nn.Module):
    """mcvae.py:17-35."""

    def __init__(self, hidden_size, num_mode, controller_rate):
        super().__init__()
        self.conv = nn.Sequential(
            Wrapper(nn.Conv2d(hidden_size, hidden_size, 3, 1, 1)), Wrapper(nn.BatchNorm2d(hidden_size)),
            Wrapper(nn.ReLU(inplace=True)), MultimodalController(hidden_size, num_mode, controller_rate),
            Wrapper(nn.Conv2d(hidden_size, hidden_size, 3, 1, 1)), Wrapper(nn.BatchNorm2d(hidden_size)),
            MultimodalController(hidden_size, num_mode, controller_rate))
        self.activation = Wrapper(nn.ReLU(inplace=True))


class Encoder(nn.Module):
    """mcvae.py:38-68."""

    def __init__(self, data_shape, hidden_size, latent_size, num_res_block, num_mode, controller_rate):
        super().__init__()
        blocks = []
        cin = data_shape[0]
        for h in hidden_size:
            blocks.extend([Wrapper(nn.Conv2d(cin, h, 4, 2, 1)), Wrapper(nn.BatchNorm2d(h)), Wrapper(nn.ReLU(inplace=True)),
                           MultimodalController(h, num_mode, controller_rate)])
            cin = h
        for _ in range(num_res_block):
            blocks.append(ResBlock(hidden_size[-1], num_mode, controller_rate))
        self.blocks = nn.Sequential(*blocks)
        self.encoded_shape = (hidden_size[-1], data_shape[1] // (2 ** len(hidden_size)), data_shape[2] // (2 ** len(hidden_size)))
        self.mu = nn.Linear(int(np.prod(self.encoded_shape)), latent_size)
        self.logvar = nn.Linear(int(np.prod(self.encoded_shape)), latent_size)


class Decoder(nn.Module):
    """mcvae.py:71-101."""

    def __init__(self, data_shape, hidden_size, latent_size, num_res_block, num_mode, controller_rate):
        super().__init__()
        self.encoded_shape = (hidden_size[-1], data_shape[1] // (2 ** len(hidden_size)), data_shape[2] // (2 ** len(hidden_size)))
        feat = int(np.prod(self.encoded_shape))
        self.linear = nn.Sequential(MultimodalController(latent_size, num_mode, controller_rate), Wrapper(nn.Linear(latent_size, feat)),
                                    Wrapper(nn.BatchNorm1d(feat)), Wrapper(nn.ReLU(inplace=True)))
        blocks = [MultimodalController(hidden_size[-1], num_mode, controller_rate)]
        for _ in range(num_res_block):
            blocks.append(ResBlock(hidden_size[-1], num_mode, controller_rate))
        for i in range(len(hidden_size) - 1, 0, -1):
            blocks.extend([Wrapper(nn.ConvTranspose2d(hidden_size[i], hidden_size[i - 1], 4, 2, 1)),
                           Wrapper(nn.BatchNorm2d(hidden_size[i - 1])), Wrapper(nn.ReLU(inplace=True)),
                           MultimodalController(hidden_size[i - 1], num_mode, controller_rate)])
        blocks.extend([Wrapper(nn.ConvTranspose2d(hidden_size[0], data_shape[0], 4, 2, 1)), Wrapper(nn.Sigmoid())])
        self.blocks = nn.Sequential(*blocks)


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
