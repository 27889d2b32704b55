"""CPU restatement (plain fp32 torch ops) of the reference MCGAN hot path.

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.

The model is not rebuilt as an ``nn.Module`` tree: it is a set of pure
functions over a flat ``{state_dict key: tensor}`` mapping that uses the
reference's own key names, so a reference ``state_dict`` is consumed as is.
Every function cites the reference lines it restates (paths relative to
``/root/reference/src``).

Arithmetic is written out explicitly (batch-norm from its definition,
spectral norm as the two mat-vecs) instead of delegating to the torch module
the reference uses, so that the oracle is an independent statement of the
algorithm; ``tests/test_oracle_golden.py`` pins it to the reference's output.
"""
from __future__ import annotations

import re
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
State = Dict[str, Tensor]

BN_EPS = 1e-5          # torch.nn.BatchNorm2d default used by models/mcgan.py:15,20,55
BN_MOMENTUM = 0.1
SN_EPS = 1e-12         # torch.nn.utils.spectral_norm default (models/utils.py:19)


# --------------------------------------------------------------------------- #
# primitive ops
# --------------------------------------------------------------------------- #
def mc_mask(x: Tensor, indicator: Tensor, codebook: Tensor) -> Tensor:
    """MultimodalController.forward (modules/modules.py:71-76).

    ``code = indicator @ codebook`` broadcast over the trailing dims; the code
    is a constant for autograd (``code.detach()``).
    """
    code = indicator.matmul(codebook).detach()
    return x * code.reshape(*code.shape, *([1] * (x.dim() - 2)))


def batch_norm(sd: State, key: str, x: Tensor, train: bool) -> Tensor:
    """nn.BatchNorm2d as used at models/mcgan.py:15,20,55 (affine, eps 1e-5,
    momentum 0.1, biased batch variance for normalisation, unbiased variance
    into the running estimate, ``num_batches_tracked`` incremented)."""
    w, b = sd[key + '.weight'], sd[key + '.bias']
    dims = [0] + list(range(2, x.dim()))
    shape = [1, -1] + [1] * (x.dim() - 2)
    if train:
        m = x.numel() // x.shape[1]
        mean = x.mean(dims)
        var = (x - mean.reshape(shape)).pow(2).mean(dims)
        with torch.no_grad():
            sd[key + '.running_mean'].mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * mean)
            sd[key + '.running_var'].mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * var * m / max(m - 1, 1))
            sd[key + '.num_batches_tracked'].add_(1)
    else:
        mean, var = sd[key + '.running_mean'], sd[key + '.running_var']
    xhat = (x - mean.reshape(shape)) * torch.rsqrt(var.reshape(shape) + BN_EPS)
    return xhat * w.reshape(shape) + b.reshape(shape)


def sn_weight(sd: State, key: str, train: bool) -> Tensor:
    """torch.nn.utils.spectral_norm (legacy hook) applied by
    make_SpectralNormalization (models/utils.py:17-21): one power iteration per
    training-mode forward on W.reshape(out, -1), u/v updated in place without
    grad, sigma = u . (W v) with grad through W only, weight = W_orig / sigma.
    Eval mode skips the iteration but still divides by sigma(u, v)."""
    w = sd[key + '.weight_orig']
    u, v = sd[key + '.weight_u'], sd[key + '.weight_v']
    wm = w.reshape(w.shape[0], -1)
    if train:
        with torch.no_grad():
            nv = wm.t().mv(u)
            nv = nv / nv.norm().clamp_min(SN_EPS)
            nu = wm.mv(nv)
            nu = nu / nu.norm().clamp_min(SN_EPS)
            v.copy_(nv)
            u.copy_(nu)
    uu, vv = u.clone(), v.clone()
    sigma = uu.dot(wm.mv(vv))
    return w / sigma


def _conv(sd: State, key: str, x: Tensor, pad: int, sn: bool, train: bool) -> Tensor:
    w = sn_weight(sd, key, train) if sn else sd[key + '.weight']
    return F.conv2d(x, w, sd[key + '.bias'], stride=1, padding=pad)


def _up2(x: Tensor) -> Tensor:
    """nn.Upsample(scale_factor=2, mode='nearest') (models/mcgan.py:17,27)."""
    return x.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)


def _pool2(x: Tensor) -> Tensor:
    """nn.AvgPool2d(2) (models/mcgan.py:82,85,110,114)."""
    n, c, h, w = x.shape
    return x.reshape(n, c, h // 2, 2, w // 2, 2).mean(dim=(3, 5))


# --------------------------------------------------------------------------- #
# generator  (models/mcgan.py:9-69)
# --------------------------------------------------------------------------- #
def gen_res_block(sd: State, p: str, x: Tensor, ind: Tensor, train: bool) -> Tensor:
    """GenResBlock.forward (models/mcgan.py:38-44) with stride 2 (the only
    configuration Generator builds, models/mcgan.py:54)."""
    cb1, cb2 = sd[p + 'mc_1.codebook'], sd[p + 'mc_2.codebook']
    h = torch.relu(batch_norm(sd, p + 'conv.0.module', x, train))
    h = mc_mask(_up2(h), ind, cb1)
    h = _conv(sd, p + 'conv.4.module', h, 1, False, train)
    h = torch.relu(batch_norm(sd, p + 'conv.5.module', h, train))
    h = _conv(sd, p + 'conv.8.module', mc_mask(h, ind, cb2), 1, False, train)
    s = _conv(sd, p + 'shortcut.2.module', mc_mask(_up2(x), ind, cb1), 0, False, train)
    return h + s


def _block_ids(sd: State, prefix: str) -> List[int]:
    ids = set()
    for k in sd:
        m = re.match(re.escape(prefix) + r'blocks\.(\d+)\.', k)
        if m:
            ids.add(int(m.group(1)))
    return sorted(ids)


def generator_forward(sd: State, z: Tensor, ind: Tensor, train: bool = True,
                      prefix: str = 'generator.') -> Tensor:
    """Generator.forward (models/mcgan.py:64-69)."""
    x = F.linear(z, sd[prefix + 'linear.module.weight'], sd[prefix + 'linear.module.bias'])
    x = x.reshape(x.shape[0], -1, 4, 4)
    nres = sum(1 for i in _block_ids(sd, prefix) if (prefix + f'blocks.{i}.mc_1.codebook') in sd)
    for i in range(nres):
        x = gen_res_block(sd, prefix + f'blocks.{i}.', x, ind, train)
    x = torch.relu(batch_norm(sd, prefix + f'blocks.{nres}.module', x, train))
    x = mc_mask(x, ind, sd[prefix + f'blocks.{nres + 2}.codebook'])
    x = _conv(sd, prefix + f'blocks.{nres + 3}.module', x, 1, False, train)
    return torch.tanh(x)


# --------------------------------------------------------------------------- #
# discriminator  (models/mcgan.py:72-181)
# --------------------------------------------------------------------------- #
def first_dis_block(sd: State, p: str, x: Tensor, ind: Tensor, train: bool) -> Tensor:
    """FirstDisResBlock.forward (models/mcgan.py:88-93)."""
    h = torch.relu(_conv(sd, p + 'conv.0.module', x, 1, True, train))
    h = _conv(sd, p + 'conv.3.module', mc_mask(h, ind, sd[p + 'mc_1.codebook']), 1, True, train)
    s = _conv(sd, p + 'shortcut.0.module', x, 0, True, train)
    return _pool2(h) + _pool2(s)


def dis_res_block(sd: State, p: str, x: Tensor, ind: Tensor, train: bool) -> Tuple[Tensor, Tensor]:
    """Both branches of DisResBlock before the optional AvgPool2d and the add
    (models/mcgan.py:101-138).  A stride-2 block and a stride-1 block with a
    channel change carry the same keys (masked 1x1 shortcut); whether the two
    branches are pooled is decided by the caller from the block's position.
    The identity shortcut of a stride-1 block returns the raw block input."""
    cb1, cb2 = sd[p + 'mc_1.codebook'], sd[p + 'mc_2.codebook']
    # the shortcut is evaluated before the main branch (mcgan.py:134-135): the
    # spectral-norm power iterations of the three convs happen in that order,
    # which does not matter numerically because the layers are independent.
    if (p + 'shortcut.1.module.weight_orig') in sd:
        s = _conv(sd, p + 'shortcut.1.module', mc_mask(x, ind, cb1), 0, True, train)
    else:
        s = x
    h = _conv(sd, p + 'conv.2.module', mc_mask(torch.relu(x), ind, cb1), 1, True, train)
    h = _conv(sd, p + 'conv.5.module', mc_mask(torch.relu(h), ind, cb2), 1, True, train)
    return h, s


def discriminator_forward(sd: State, x: Tensor, ind: Tensor, train: bool = True,
                          prefix: str = 'discriminator.', cifar_layout: bool = True) -> Tensor:
    """Discriminator.forward (models/mcgan.py:178-181).  ``cifar_layout``
    selects the block schedule of models/mcgan.py:155-165 (len-3 stride-2
    blocks then two stride-1) versus :166-175 (len-2 stride-2, one stride-1)."""
    ids = [i for i in _block_ids(sd, prefix) if (prefix + f'blocks.{i}.mc_1.codebook') in sd]
    nres = len(ids) - 1
    n_stride1 = 2 if cifar_layout else 1
    x = first_dis_block(sd, prefix + 'blocks.0.', x, ind, train)
    for j in range(1, nres + 1):
        h, s = dis_res_block(sd, prefix + f'blocks.{j}.', x, ind, train)
        if j <= nres - n_stride1:
            x = _pool2(h) + _pool2(s)
        else:
            x = h + s
    t = nres + 1
    x = mc_mask(torch.relu(x), ind, sd[prefix + f'blocks.{t + 1}.codebook'])
    x = x.sum(dim=(-2, -1)).reshape(x.shape[0], -1)            # GlobalSumPooling, mcgan.py:145-147
    key = prefix + f'blocks.{t + 3}.module'
    return F.linear(x, sn_weight(sd, key, train), sd[key + '.bias'])


# --------------------------------------------------------------------------- #
# model-level API + train step
# --------------------------------------------------------------------------- #
def one_hot(label: Tensor, classes: int) -> Tensor:
    return F.one_hot(label, classes).float()          # models/mcgan.py:196,201


def trainable_keys(sd: State, prefix: str) -> List[str]:
    skip = ('running_mean', 'running_var', 'num_batches_tracked', 'codebook', 'weight_u', 'weight_v')
    return [k for k in sd if k.startswith(prefix) and not k.endswith(skip)]


class OracleMCGAN:
    """Holds a reference-format state dict and runs the reference train step.

    ``train_iteration`` restates the loop body of train_gan.py:139-176:
    5 x {zero_grad; D(real); G(z) in train mode; D(G(z).detach()); hinge;
    backward; Adam(D)} then 1 x {zero_grad; G(z); D(G(z)); -mean; backward;
    Adam(G)} with Adam(lr 2e-4, betas (0.5, 0.999), eps 1e-8, wd 0)
    (train_gan.py:43-47,223-236).
    """

    def __init__(self, state: State, classes: int, cifar_layout: bool = True,
                 lr: float = 2e-4, betas: Tuple[float, float] = (0.5, 0.999),
                 d_iters: int = 5, g_iters: int = 1):
        self.sd: State = {}
        seen = {}
        for k, v in state.items():
            t = v.detach().clone()
            # aliased codebook keys (shared mc_1 instance) stay aliased
            ident = (v.data_ptr(), tuple(v.shape)) if v.numel() else None
            if ident is not None and ident in seen and k.endswith('codebook'):
                t = self.sd[seen[ident]]
            elif ident is not None:
                seen[ident] = k
            self.sd[k] = t
        self.classes = classes
        self.cifar_layout = cifar_layout
        self.gkeys = trainable_keys(self.sd, 'generator.')
        self.dkeys = trainable_keys(self.sd, 'discriminator.')
        for k in self.gkeys + self.dkeys:
            self.sd[k].requires_grad_(True)
        self.opt_g = torch.optim.Adam([self.sd[k] for k in self.gkeys], lr=lr, betas=betas)
        self.opt_d = torch.optim.Adam([self.sd[k] for k in self.dkeys], lr=lr, betas=betas)
        self.d_iters, self.g_iters = d_iters, g_iters

    def generate(self, label: Tensor, z: Tensor, train: bool = True) -> Tensor:
        return generator_forward(self.sd, z, one_hot(label, self.classes), train)

    def discriminate(self, x: Tensor, label: Tensor, train: bool = True) -> Tensor:
        return discriminator_forward(self.sd, x, one_hot(label, self.classes), train,
                                     cifar_layout=self.cifar_layout)

    def _zero(self):
        self.opt_d.zero_grad()
        self.opt_g.zero_grad()

    def train_iteration(self, img: Tensor, label: Tensor, zs: List[Tensor]):
        """``zs`` holds d_iters + g_iters latent batches, consumed in the
        reference's draw order (z1 per D update, then z2)."""
        zi = iter(zs)
        for _ in range(self.d_iters):
            self._zero()
            d_x = self.discriminate(img, label)
            fake = self.generate(label, next(zi))
            d_gz = self.discriminate(fake.detach(), label)
            d_loss = torch.relu(1.0 - d_x).mean() + torch.relu(1.0 + d_gz).mean()
            d_loss.backward()
            self.opt_d.step()
        for _ in range(self.g_iters):
            self._zero()
            fake = self.generate(label, next(zi))
            g_loss = -self.discriminate(fake, label).mean()
            g_loss.backward()
            self.opt_g.step()
        return float(d_loss.detach()), float(g_loss.detach())
