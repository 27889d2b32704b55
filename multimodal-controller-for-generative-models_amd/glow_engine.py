"""MCGlow on the HIP kernels: likelihood forward (with the data-dependent ActNorm initialisation) and the
reverse / sampling pass.  Reference chains: Flow.forward / reverse (mcglow.py:188-201), Block.forward /
reverse (mcglow.py:219-265), MCGlow.forward / loss_fn / reverse (mcglow.py:283-325).

Per flow:  ActNorm is folded into the prologue of the invertible 1x1 convolution (fused conv, 1x1);
the coupling network is three fused convolutions (3x3, 1x1, 3x3) whose prologues carry the inner ActNorms,
ReLU and the MultimodalController codes; the affine transform + per-sample log-determinant is one kernel.
The coupling network reads only the first C/2 channels: its first weight image is zero over the others,
so no split/concat copy is needed inside a flow.
"""
from __future__ import annotations

import math
from typing import List

import torch
import torch.nn.functional as F

from . import ops
from .ops import Seg, pad8

Tensor = torch.Tensor


class GlowEngine:
    def __init__(self, model, dtype: torch.dtype = torch.float32):
        self.m = model
        self.dtype = dtype

    # ---- helpers ---------------------------------------------------------------------------------------------
    def _actnorm(self, an, x_stats_fn, count, train: bool, cp: int):
        """Prologue vectors (a, b) of an ActNorm; runs the data-dependent init on the first training forward."""
        if train and int(an.initialized) == 0:
            ops.actnorm_init(x_stats_fn(), count, an.loc.data, an.scale.data)
            an.initialized.fill_(1)
        return ops.actnorm_affine(an.loc.data, an.scale.data, cp)

    def _zero_conv_image(self, zc, cin_pad: int):
        """ZeroConv2d (mcglow.py:119-130) as weight image + bias with exp(3*scale) folded into the rows."""
        rs = torch.exp(zc.scale.detach().reshape(-1) * 3)
        w = zc.conv.weight.detach()
        if w.shape[1] < cin_pad:
            w = F.pad(w, (0, 0, 0, 0, 0, cin_pad - w.shape[1]))
        return ops.prep_weight_rows(w, self.dtype, rs), zc.conv.bias.detach() * rs

    def _coupling_net(self, cp_net, x: Tensor, c: int, codes, train: bool):
        """AffineCoupling.net on the first c/2 channels of x -> [log_s | t] (c channels)."""
        net = cp_net
        dt = self.dtype
        n, h, w, cp = x.shape
        count = n * h * w
        conv0, an1, mc1, conv1, an5, mc2, zc = (net[0].module, net[1].module, net[3], net[4].module, net[5].module,
                                                 net[7], net[8].module)
        w0 = F.pad(conv0.weight.detach(), (0, 0, 0, 0, 0, cp - conv0.weight.shape[1]))      # zero over channels >= c/2
        need1 = train and int(an1.initialized) == 0
        h1, st1 = ops.conv_fused([Seg(x)], ops.prep_weight(w0, dt), conv0.out_channels, bias=conv0.bias,
                                 stats_mode=1 if need1 else 0)
        hid = conv0.out_channels
        a1, b1 = self._actnorm(an1, lambda: st1, count, train, hid)
        need5 = train and int(an5.initialized) == 0
        h2, st2 = ops.conv_fused([Seg(h1, ksize=1, scale=a1, shift=b1, relu=True, code=codes[0])],
                                 ops.prep_weight(conv1.weight.detach(), dt), hid, bias=conv1.bias,
                                 stats_mode=1 if need5 else 0)
        a5, b5 = self._actnorm(an5, lambda: st2, count, train, hid)
        wz, bz = self._zero_conv_image(zc, hid)
        hz, _ = ops.conv_fused([Seg(h2, scale=a5, shift=b5, relu=True, code=codes[1])], wz, c, bias=bz, cy=cp)
        return hz

    def _flow_forward(self, flow, x: Tensor, c: int, indicator: Tensor, logdet: Tensor, train: bool) -> Tensor:
        dt = self.dtype
        n, h, w, cp = x.shape
        a, b = self._actnorm(flow.actnorm, lambda: ops.channel_stats(x), n * h * w, train, cp)
        ic = flow.invconv
        wmat, _ = ops.invconv_weight(ic.w_p, ic.w_l.data, ic.w_u.data, ic.w_s.data, ic.s_sign)
        wpad = F.pad(wmat, (0, cp - c)).reshape(c, cp, 1, 1)
        out, _ = ops.conv_fused([Seg(x, ksize=1, scale=a, shift=b)], ops.prep_weight(wpad, dt), c, cy=cp)
        # parameter-only log-determinants: H*W * (sum log|scale| + sum w_s)   (mcglow.py:46-47,101)
        logdet += (h * w) * (torch.log(torch.abs(flow.actnorm.scale.detach())).sum() + ic.w_s.detach().sum())
        net = flow.coupling.net
        codes = (net[3].code(indicator), net[7].code(indicator))
        hz = self._coupling_net(net, out, c, codes, train)
        return ops.glow_coupling(out, hz, c, logdet, reverse=False, accumulate=True)

    def _flow_reverse(self, flow, y: Tensor, c: int, indicator: Tensor) -> Tensor:
        dt = self.dtype
        cp = y.shape[-1]
        net = flow.coupling.net
        codes = (net[3].code(indicator), net[7].code(indicator))
        hz = self._coupling_net(net, y, c, codes, False)
        x = ops.glow_coupling(y, hz, c, None, reverse=True)
        ic = flow.invconv
        _, winv = ops.invconv_weight(ic.w_p, ic.w_l.data, ic.w_u.data, ic.w_s.data, ic.s_sign, inverse=True)
        # ActNorm.reverse folded in: x_prev = (W^-1 x) / scale - loc
        an = flow.actnorm
        wpad = F.pad(winv, (0, cp - c)).reshape(c, cp, 1, 1)
        wimg = ops.prep_weight_rows(wpad, dt, 1.0 / an.scale.detach().reshape(-1))
        out, _ = ops.conv_fused([Seg(x, ksize=1)], wimg, c, bias=-an.loc.detach().reshape(-1), cy=cp)
        return out

    # ---- forward: bits per dimension -----------------------------------------------------------------------------
    def forward(self, img: Tensor, indicator: Tensor, noise: Tensor, train: bool):
        m, dt = self.m, self.dtype
        n = img.shape[0]
        x0 = img * 0.5 + noise / 256                                   # mcglow.py:298-299
        c = m.data_shape[0]
        x = ops.to_nhwc(x0.contiguous(), dt)
        logdet = torch.zeros(n, dtype=torch.float32, device=img.device)
        logp = torch.zeros(n, dtype=torch.float32, device=img.device)
        zs: List[Tensor] = []
        for blk in m.blocks:
            x = ops.glow_squeeze(x, c)
            c *= 4
            for flow in blk.flows:
                x = self._flow_forward(flow, x, c, indicator, logdet, train)
            nb, h, w, cp = x.shape
            if blk.split:
                half = c // 2
                keep = torch.zeros((nb, h, w, pad8(half)), dtype=dt, device=x.device)
                ops.copy_channels(x, 0, keep, 0, half)
                wz, bz = self._zero_conv_image(blk.prior, keep.shape[-1])
                prior, _ = ops.conv_fused([Seg(keep)], wz, c, bias=bz)
                ops.gaussian_logp(x, half, prior, half, logp)
                znew = torch.zeros((nb, h, w, pad8(half)), dtype=dt, device=x.device)
                ops.copy_channels(x, half, znew, 0, half)
                zs.append(ops.to_nchw(znew, half))
                x, c = keep, half
            else:
                wz, bz = self._zero_conv_image(blk.prior, cp)
                prior, _ = ops.conv_fused([Seg(torch.zeros_like(x))], wz, 2 * c, bias=bz)
                ops.gaussian_logp(x, 0, prior, c, logp)
                zs.append(ops.to_nchw(x, c))
        n_pixel = float(img[0].numel())
        loss = -(-math.log(256.) * n_pixel + logdet + logp) / (math.log(2.) * n_pixel)       # loss_fn, mcglow.py:283-293
        loss = torch.where(torch.isnan(loss), torch.zeros_like(loss), loss) if train else loss[~torch.isnan(loss)]
        return loss.mean(), zs

    # ---- reverse: reconstruction / sampling ------------------------------------------------------------------------
    def reverse(self, zs: List[Tensor], indicator: Tensor, reconstruct: bool) -> Tensor:
        m, dt = self.m, self.dtype
        L = len(m.blocks)
        x = None
        c_in = [m.data_shape[0] * 2 ** i for i in range(L)]               # channels entering block i
        for i in reversed(range(L)):
            blk = m.blocks[i]
            c = c_in[i] * 4
            eps = ops.to_nhwc(zs[i].contiguous(), dt)
            nb, h, w, _ = eps.shape
            inp = torch.zeros((nb, h, w, pad8(c)), dtype=dt, device=eps.device)
            if blk.split:
                half = c // 2
                ops.copy_channels(x, 0, inp, 0, half)
                if reconstruct:
                    ops.copy_channels(eps, 0, inp, half, half)
                else:
                    wz, bz = self._zero_conv_image(blk.prior, x.shape[-1])
                    prior, _ = ops.conv_fused([Seg(x)], wz, c, bias=bz)
                    ops.gaussian_sample(eps, prior, inp, half, half)
            else:
                if reconstruct:
                    ops.copy_channels(eps, 0, inp, 0, c)
                else:
                    wz, bz = self._zero_conv_image(blk.prior, inp.shape[-1])
                    prior, _ = ops.conv_fused([Seg(torch.zeros_like(inp))], wz, 2 * c, bias=bz)
                    ops.gaussian_sample(eps, prior, inp, 0, c)
            for flow in reversed(list(blk.flows)):
                inp = self._flow_reverse(flow, inp, c, indicator)
            x = ops.glow_unsqueeze(inp, c)
        return torch.clamp(ops.to_nchw(x, m.data_shape[0]), -.5, .5) * 2
