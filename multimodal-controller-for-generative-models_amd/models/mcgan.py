"""MCGAN with the reference's module surface (src/models/mcgan.py), computed by the fused
HIP engines in ``gan_engine.py``.

The module TREE (names, indices, parameter/buffer keys) is the reference's, so a reference
``state_dict`` loads unchanged and ``models.utils.create/transit`` find the
``MultimodalController`` children by class name.  The per-layer children are containers:
``Generator.forward`` / ``Discriminator.forward`` run the whole network as fused kernels.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..config import cfg
from ..gan_engine import DiscriminatorEngine, GeneratorEngine
from ..modules import MultimodalController, Wrapper
from .utils import init_param, make_SpectralNormalization


def _chain(*mods):
    """nn.Sequential over the list protocol: bare tensor modules get a Wrapper."""
    return nn.Sequential(*[m if isinstance(m, (MultimodalController, Wrapper)) else Wrapper(m) for m in mods])


def _compute_dtype():
    name = cfg.get('compute_dtype', 'float32')
    return {'float32': torch.float32, 'bfloat16': torch.bfloat16}[name]


class _FusedNet(nn.Module):
    """Shared plumbing: lazily built engine, compute dtype switch, autograd bridge."""
    _engine_cls = None

    def _engine(self):
        eng = self.__dict__.get('_eng')
        if eng is None or eng.dtype != self.compute_dtype:
            eng = self._engine_cls(self, self.compute_dtype)
            self.__dict__['_eng'] = eng
        return eng

    @property
    def compute_dtype(self):
        return self.__dict__.get('_cdt') or _compute_dtype()

    def set_compute_dtype(self, dtype):
        self.__dict__['_cdt'] = dtype
        return self


# --------------------------------------------------------------------------------------------- #
class GenResBlock(nn.Module):
    """mcgan.py:9-44 (stride-2 form).  Children are parameter containers for the fused engine."""

    def __init__(self, input_size, output_size, num_mode, controller_rate, stride):
        super().__init__()
        if stride != 2:
            raise ValueError('Not valid stride')          # Generator only builds stride-2 blocks (mcgan.py:54)
        self.mc_1 = MultimodalController(input_size, num_mode, controller_rate)
        self.mc_2 = MultimodalController(output_size, num_mode, controller_rate)
        self.conv = _chain(nn.BatchNorm2d(input_size), nn.ReLU(), nn.Upsample(scale_factor=stride, mode='nearest'),
                           self.mc_1, nn.Conv2d(input_size, output_size, 3, 1, 1),
                           nn.BatchNorm2d(output_size), nn.ReLU(), self.mc_2,
                           nn.Conv2d(output_size, output_size, 3, 1, 1))
        self.shortcut = _chain(nn.Upsample(scale_factor=stride, mode='nearest'), self.mc_1,
                               nn.Conv2d(input_size, output_size, 1, 1, 0))


class _GenFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eng, z, indicator, train, *params):
        img, saved = eng.forward(z, indicator, train)
        ctx.eng, ctx.saved = eng, saved
        ctx.z_needs = z.requires_grad
        return img

    @staticmethod
    def backward(ctx, dimg):
        if ctx.needs_input_grad[1]:
            raise NotImplementedError('gradient w.r.t. the latent z is not produced by the fused generator')
        eng = ctx.eng
        gflat = torch.empty_like(eng.flat_p.flat)
        eng.backward(ctx.saved, dimg.contiguous(), gflat, accumulate=False)
        return (None, None, None, None, *eng.flat_p.views(gflat))


class Generator(_FusedNet):
    _engine_cls = GeneratorEngine

    def __init__(self, data_shape, latent_size, hidden_size, num_mode, controller_rate):
        super().__init__()
        self.latent_size = latent_size
        self.linear = Wrapper(nn.Linear(latent_size, hidden_size[0] * 4 * 4))
        stages = [GenResBlock(a, b, num_mode, controller_rate, stride=2) for a, b in zip(hidden_size[:-1], hidden_size[1:])]
        head = _chain(nn.BatchNorm2d(hidden_size[-1]), nn.ReLU(),
                      MultimodalController(hidden_size[-1], num_mode, controller_rate),
                      nn.Conv2d(hidden_size[-1], data_shape[0], 3, 1, 1), nn.Tanh())
        self.blocks = nn.Sequential(*stages, *head)

    def forward(self, input, indicator):
        eng = self._engine()
        eng.flat_p.ensure()
        return _GenFn.apply(eng, input, indicator, self.training, *eng.flat_p.tensors)


# --------------------------------------------------------------------------------------------- #
class FirstDisResBlock(nn.Module):
    """mcgan.py:72-93."""

    def __init__(self, input_size, output_size, num_mode, controller_rate):
        super().__init__()
        self.mc_1 = MultimodalController(output_size, num_mode, controller_rate)
        self.conv = _chain(nn.Conv2d(input_size, output_size, 3, 1, 1), nn.ReLU(), self.mc_1,
                           nn.Conv2d(output_size, output_size, 3, 1, 1), nn.AvgPool2d(2))
        self.shortcut = _chain(nn.Conv2d(input_size, output_size, 1, 1, 0), nn.AvgPool2d(2))


class DisResBlock(nn.Module):
    """mcgan.py:96-138: stride 2 pools both branches; stride 1 keeps an identity shortcut unless
    the channel count changes."""

    def __init__(self, input_size, output_size, num_mode, controller_rate, stride):
        super().__init__()
        self.mc_1 = MultimodalController(input_size, num_mode, controller_rate)
        self.mc_2 = MultimodalController(output_size, num_mode, controller_rate)
        main = [nn.ReLU(), self.mc_1, nn.Conv2d(input_size, output_size, 3, 1, 1),
                nn.ReLU(), self.mc_2, nn.Conv2d(output_size, output_size, 3, 1, 1)]
        side = []
        if stride > 1 or input_size != output_size:
            side = [self.mc_1, nn.Conv2d(input_size, output_size, 1, 1, 0)]
        if stride > 1:
            main.append(nn.AvgPool2d(2))
            side.append(nn.AvgPool2d(2))
        self.conv = _chain(*main)
        self.shortcut = _chain(*side)


class GlobalSumPooling(nn.Module):
    def forward(self, input):          # mcgan.py:141-147; container only, the tail kernel does the sum
        return input.sum(dim=[-2, -1]).view(input.size(0), -1)


class _DisFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eng, x, indicator, train, *params):
        logit, saved = eng.forward(x, indicator, train)
        ctx.eng, ctx.saved = eng, saved
        return logit

    @staticmethod
    def backward(ctx, dlogit):
        eng = ctx.eng
        want_w = any(ctx.needs_input_grad[4:])
        gflat = torch.empty_like(eng.flat_p.flat) if want_w else None
        dimg = eng.backward(ctx.saved, dlogit.contiguous().view(-1), gflat, accumulate=False,
                            need_input_grad=ctx.needs_input_grad[1])
        grads = eng.flat_p.views(gflat) if want_w else [None] * len(eng.flat_p.tensors)
        return (None, dimg, None, None, *grads)


class Discriminator(_FusedNet):
    _engine_cls = DiscriminatorEngine

    def __init__(self, data_shape, hidden_size, num_mode, controller_rate):
        super().__init__()
        self.data_shape = data_shape
        h = hidden_size
        stages = [FirstDisResBlock(data_shape[0], h[0], num_mode, controller_rate)]
        # mcgan.py:155-175: CIFAR keeps two stride-1 blocks at 8x8, the other datasets one
        n_down = len(h) - 3 if cfg['data_name'] in ['CIFAR10', 'CIFAR100'] else len(h) - 2
        for i in range(len(h) - 1):
            stages.append(DisResBlock(h[i], h[i + 1], num_mode, controller_rate, stride=2 if i < n_down else 1))
        tail = _chain(nn.ReLU(), MultimodalController(h[-1], num_mode, controller_rate), GlobalSumPooling(),
                      nn.Linear(h[-1], 1))
        self.blocks = nn.Sequential(*stages, *tail)

    def forward(self, input, indicator):
        eng = self._engine()
        eng._ensure_flat()
        return _DisFn.apply(eng, input, indicator, self.training, *eng.flat_p.tensors)


# --------------------------------------------------------------------------------------------- #
class MCGAN(nn.Module):
    """mcgan.py:184-209."""

    def __init__(self, data_shape, latent_size, generator_hidden_size, discriminator_hidden_size, num_mode,
                 controller_rate):
        super().__init__()
        self.latent_size = latent_size
        self.generator = Generator(data_shape, latent_size, generator_hidden_size, num_mode, controller_rate)
        self.discriminator = Discriminator(data_shape, discriminator_hidden_size, num_mode, controller_rate)
        self.discriminator.apply(make_SpectralNormalization)

    def set_compute_dtype(self, dtype):
        self.generator.set_compute_dtype(dtype)
        self.discriminator.set_compute_dtype(dtype)
        return self

    def generate(self, C, x=None):
        if x is None:
            x = torch.randn([C.size(0), self.latent_size], device=cfg['device'])
        return self.generator(x, F.one_hot(C, cfg['classes_size']).float())

    def discriminate(self, x, C):
        return self.discriminator(x, F.one_hot(C, cfg['classes_size']).float())

    def forward(self, input):
        x = torch.randn(input['img'].size(0), self.latent_size, device=cfg['device'])
        return self.discriminate(self.generate(input['label'], x), input['label'])


def mcgan():
    """Zero-argument factory reading cfg, as models/mcgan.py:212-221."""
    model = MCGAN(cfg['data_shape'], cfg['gan']['latent_size'], cfg['gan']['generator_hidden_size'],
                  cfg['gan']['discriminator_hidden_size'], cfg['classes_size'], cfg['controller_rate'])
    model.apply(init_param)
    return model
