"""MCGlow with the reference's module surface (src/models/mcglow.py): ActNorm, LU-parameterised invertible
1x1 convolution, affine coupling with MultimodalController-masked coupling networks, multi-scale blocks.

The module tree carries the reference's parameter / buffer names (``state_dict`` compatible); the arithmetic
runs in ``glow_engine.py`` on HIP kernels: the likelihood forward (incl. the data-dependent ActNorm
initialisation), its backward (``output['loss'].backward()`` works through one ``autograd.Function`` that
replays the engine's tape), ``reverse`` and ``generate``.
"""
from __future__ import annotations

import numpy as np
import scipy.linalg as la
import torch
import torch.nn as nn

from ..config import cfg
from ..glow_engine import GlowEngine
from ..modules import MultimodalController, Wrapper
from .utils import init_param


class _GlowFn(torch.autograd.Function):
    """loss = engine.forward(...); backward = engine.backward over the saved tape (one node for the whole model)."""

    @staticmethod
    def forward(ctx, engine, img, label, noise, train, holder, *params):
        tape = []
        loss, zs = engine.forward(img, None, noise, train, tape, label=label)
        holder['z'] = zs
        ctx.engine, ctx.tape, ctx.params = engine, tape, params
        ctx.n, ctx.n_pixel = img.shape[0], float(img[0].numel())
        return loss

    @staticmethod
    def backward(ctx, gloss):
        eng = ctx.engine
        sink = {}
        eng._gsink = sink
        try:
            eng.backward(ctx.tape, ctx.n, ctx.n_pixel)
        finally:
            eng._gsink = None
        ctx.tape = None
        grads = tuple(sink[id(p)] * gloss if id(p) in sink else None for p in ctx.params)
        return (None, None, None, None, None, None) + grads


class ActNorm(nn.Module):
    """mcglow.py:24-55: per-channel loc/scale, data-initialised on the first training forward."""

    def __init__(self, input_size, logdet=True):
        super().__init__()
        self.loc = nn.Parameter(torch.zeros(1, input_size, 1, 1))
        self.scale = nn.Parameter(torch.ones(1, input_size, 1, 1))
        self.register_buffer('initialized', torch.tensor(0, dtype=torch.uint8))
        self.logdet = logdet


class InvConv2dLU(nn.Module):
    """mcglow.py:76-116: W = P (L o mask + I) (U o mask + diag(sign * exp(w_s))), initialised from the LU
    factors of a random orthogonal matrix (NumPy's global RNG, as in the reference)."""

    def __init__(self, input_size):
        super().__init__()
        q, _ = la.qr(np.random.randn(input_size, input_size))
        w_p, w_l, w_u = la.lu(q.astype(np.float32))
        w_s = np.diag(w_u).copy()
        u_mask = np.triu(np.ones_like(w_u), 1)
        self.register_buffer('w_p', torch.from_numpy(np.ascontiguousarray(w_p)))
        self.register_buffer('u_mask', torch.from_numpy(u_mask))
        self.register_buffer('l_mask', torch.from_numpy(np.ascontiguousarray(u_mask.T)))
        self.register_buffer('s_sign', torch.sign(torch.from_numpy(w_s)))
        self.register_buffer('l_eye', torch.eye(input_size))
        self.w_l = nn.Parameter(torch.from_numpy(np.ascontiguousarray(w_l)))
        self.w_s = nn.Parameter(torch.log(torch.abs(torch.from_numpy(w_s))))
        self.w_u = nn.Parameter(torch.from_numpy(np.triu(w_u, 1).copy()))


class ZeroConv2d(nn.Module):
    """mcglow.py:119-130: zero-initialised 3x3 conv whose output is multiplied by exp(3 * scale)."""

    def __init__(self, input_size, output_size):
        super().__init__()
        self.conv = nn.Conv2d(input_size, output_size, 3, 1, 1)
        self.conv.weight.data.zero_()
        self.conv.bias.data.zero_()
        self.scale = nn.Parameter(torch.zeros(1, output_size, 1, 1))


class AffineCoupling(nn.Module):
    """mcglow.py:133-175 (affine form)."""

    def __init__(self, input_size, hidden_size=512, affine=True, num_mode=None, controller_rate=None):
        super().__init__()
        if not affine:
            raise ValueError('Not valid coupling: only the affine form (cfg glow.affine = True) is built')
        self.affine = affine
        conv_in, conv_mid = nn.Conv2d(input_size // 2, hidden_size, 3, padding=1), nn.Conv2d(hidden_size, hidden_size, 1)
        for conv in (conv_in, conv_mid):
            conv.weight.data.normal_(0, 0.05)
            conv.bias.data.zero_()
        self.net = nn.Sequential(
            Wrapper(conv_in), Wrapper(ActNorm(hidden_size, logdet=False)), Wrapper(nn.ReLU(inplace=True)),
            MultimodalController(hidden_size, num_mode, controller_rate),
            Wrapper(conv_mid), Wrapper(ActNorm(hidden_size, logdet=False)), Wrapper(nn.ReLU(inplace=True)),
            MultimodalController(hidden_size, num_mode, controller_rate),
            Wrapper(ZeroConv2d(hidden_size, input_size)))


class Flow(nn.Module):
    """mcglow.py:178-201."""

    def __init__(self, input_size, hidden_size, affine=True, conv_lu=True, num_mode=None, controller_rate=None):
        super().__init__()
        if not conv_lu:
            raise ValueError('Not valid invertible conv: only the LU form (cfg glow.conv_lu = True) is built')
        self.actnorm = ActNorm(input_size)
        self.invconv = InvConv2dLU(input_size)
        self.coupling = AffineCoupling(input_size, hidden_size, affine, num_mode, controller_rate)


class Block(nn.Module):
    """mcglow.py:204-265: squeeze, K flows, split prior (or the unconditional prior of the last block)."""

    def __init__(self, input_size, hidden_size, K, split=True, affine=True, conv_lu=True, num_mode=None, controller_rate=None):
        super().__init__()
        self.flows = nn.ModuleList(Flow(input_size * 4, hidden_size, affine, conv_lu, num_mode, controller_rate) for _ in range(K))
        self.split = split
        self.prior = ZeroConv2d(input_size * 2, input_size * 4) if split else ZeroConv2d(input_size * 4, input_size * 8)


class MCGlow(nn.Module):
    """mcglow.py:268-350."""

    def __init__(self, data_shape, hidden_size, K, L, affine=True, conv_lu=True, num_mode=None, controller_rate=0.5):
        super().__init__()
        self.data_shape, self.K, self.L = data_shape, K, L
        self.num_mode, self.controller_rate = num_mode, controller_rate
        self.blocks = nn.ModuleList()
        c = data_shape[0]
        for _ in range(L - 1):
            self.blocks.append(Block(c, hidden_size, K, True, affine, conv_lu, num_mode, controller_rate))
            c *= 2
        self.blocks.append(Block(c, hidden_size, K, False, affine, conv_lu, num_mode, controller_rate))

    # ---- fused path ------------------------------------------------------------------------------------------
    def _engine(self):
        eng = self.__dict__.get('_eng')
        dt = {'float32': torch.float32, 'bfloat16': torch.bfloat16}[cfg.get('compute_dtype', 'float32')]
        dt = self.__dict__.get('_cdt') or dt
        if eng is None or eng.dtype != dt:
            eng = GlowEngine(self, dt)
            self.__dict__['_eng'] = eng
        return eng

    def set_compute_dtype(self, dtype):
        self.__dict__['_cdt'] = dtype
        return self

    def forward(self, input):
        """Negative log-likelihood in bits/dim (mcglow.py:283-312).  The dequantisation noise U(0,1)/256 is
        drawn here unless `input['noise']` supplies it (parity runs)."""
        label = self._check_label(input['label'])        # one_hot(label) @ codebook == codebook[label] (modules.py:73)
        noise = input['noise'] if 'noise' in input else torch.rand_like(input['img'])
        if torch.is_grad_enabled() and self.training:
            holder = {}
            params = [p for p in self.parameters() if p.requires_grad]
            loss = _GlowFn.apply(self._engine(), input['img'], label, noise, True, holder, *params)
            return {'loss': loss, 'z': holder['z']}
        loss, z = self._engine().forward(input['img'], None, noise, self.training, label=label)
        return {'loss': loss, 'z': z}

    @staticmethod
    def _check_label(label):
        if label.dtype != torch.int64 or label.dim() != 1:
            raise ValueError('Not valid label: expected an int64 vector of class indices')
        return label

    def reverse(self, input):
        return {'img': self._engine().reverse(input['z'], None, bool(input['reconstruct']), label=self._check_label(input['label']))}

    def make_z_shapes(self):
        c, h, w = self.data_shape
        shapes = []
        for _ in range(self.L - 1):
            h, w, c = h // 2, w // 2, c * 2
            shapes.append((c, h, w))
        shapes.append((c * 4, h // 2, w // 2))
        return shapes

    def generate(self, C, x=None, temperature=1):
        if x is None:
            x = [torch.randn([C.size(0), *s], device=cfg['device']) * temperature for s in self.make_z_shapes()]
        return self.reverse({'z': x, 'reconstruct': False, 'label': C})['img']


def mcglow():
    g = cfg['glow']
    model = MCGlow(cfg['data_shape'], g['hidden_size'], g['K'], g['L'], g['affine'], g['conv_lu'], cfg['classes_size'],
                   cfg['controller_rate'])
    model.apply(init_param)
    return model
