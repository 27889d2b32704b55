"""MCVAE on the HIP kernels: forward (reconstruction, mu, logvar, loss) and the hand-derived backward.
Reference chain: MCVAE.forward (mcvae.py:133-144) -> Encoder.forward (:63-68) / Decoder.forward (:97-101) ->
ResBlock.forward (:31-35), loss (:10-14).

* Conv2d(.., 4, 2, 1)          = strided NHWC im2col (16 taps) + the fused 1x1 convolution (bias + BN partial sums).
* ConvTranspose2d(.., 4, 2, 1) = fused 1x1 convolution producing the 16 tap planes + col2im (the adjoint gather,
                                 bias added there); BN partial sums from the channel-statistics kernel.
* ResBlock                     = two fused 3x3 convolutions (the second carries BN -> ReLU -> MC as its prologue)
                                 + one tail kernel  relu(BN(h2) * code + x).
* BN -> ReLU -> MC stage tails = one elementwise kernel each; their backward is the two-pass BN-backward kernel pair.
* Linear layers (mu | logvar in one launch; the decoder's latent -> 4x4 map) are 1x1 convolutions whose weight
  rows / columns are permuted between the reference's (c, h, w) flattening and NHWC.
The [N, F] BatchNorm1d backward, the reparameterisation and the KL term are a few tensor ops on [N, <=4096] arrays.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn.functional as F

from . import ops
from .ops import Seg, pad8

Tensor = torch.Tensor


def _t1x1(w: Tensor) -> Tensor:
    return w.reshape(w.shape[0], -1).t().contiguous().reshape(-1, w.shape[0], 1, 1)


def _t3x3(w: Tensor) -> Tensor:
    return w.flip(2, 3).transpose(0, 1).contiguous()


class VAEEngine:
    def __init__(self, model, dtype: torch.dtype = torch.float32):
        self.m = model
        self.dtype = dtype
        self._gsink = None
        if any(h % 8 for h in model.hidden_size) or model.latent_size % 8:
            raise ValueError('Not valid hidden/latent size: the fused path needs multiples of 8')
        self._perm = None

    # ---- helpers -----------------------------------------------------------------------------------------------
    def _grad(self, p: Tensor) -> Tensor:
        if self._gsink is not None:
            g = self._gsink.get(id(p))
            if g is None:
                g = self._gsink[id(p)] = torch.zeros_like(p)
            return g
        if p.grad is None:
            p.grad = torch.zeros_like(p)
        return p.grad

    @staticmethod
    def _bn(bn, stats: Optional[Tensor], count: int, train: bool):
        if train:
            sc, sh, mean, rstd = ops.bn_finalize(stats, count, bn.weight.detach(), bn.bias.detach(), bn.running_mean,
                                                 bn.running_var, bn.momentum, bn.eps)
            bn.num_batches_tracked += 1
            return sc, sh, mean, rstd
        sc, sh = ops.bn_eval_affine(bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var, bn.eps)
        return sc, sh, None, None

    def _flat_perm(self, device) -> Tensor:
        """perm[p * C + c] = c * hw + p : NHWC feature index -> the reference's (c, h, w) flattening."""
        if self._perm is None or self._perm.device != device:
            c, h, w = self.m.encoder.encoded_shape
            idx = torch.arange(c * h * w, device=device).reshape(c, h * w)
            self._perm = idx.t().reshape(-1).contiguous()
        return self._perm

    # ---- building blocks -----------------------------------------------------------------------------------------
    def _down_fwd(self, conv, bn, mc, x: Tensor, label, train: bool, tape):
        dt = self.dtype
        cp = x.shape[-1]
        col = ops.im2col(x, 4, 4, 1, 1, stride=2)
        w = conv.weight.detach().permute(0, 2, 3, 1)                            # [co, 4, 4, ci]
        wm = F.pad(w, (0, cp - w.shape[-1])).reshape(w.shape[0], 16 * cp, 1, 1).contiguous()
        h, st = ops.conv_fused([Seg(col, ksize=1)], ops.prep_weight(wm, dt), conv.out_channels, bias=conv.bias.detach(),
                               stats_mode=1 if train else 0)
        n, ho, wo, _ = h.shape
        b = self._bn(bn, st, n * ho * wo, train)
        code = mc.code_of_labels(label)
        a = ops.affine_code_res(h, b[0], b[1], code, None, pre_relu=True)
        if tape is not None:
            tape.append(dict(kind='down', col=col, wm=wm, h=h, bn=b, code=code, cin_p=cp))
        return a

    def _down_bwd(self, conv, bn, r, g: Tensor, need_dx: bool):
        dt = self.dtype
        sc, sh, mean, rstd = r['bn']
        d_h = ops.code_bn_bwd(g, r['code'], r['h'], sc, mean, rstd, self._grad(bn.weight), self._grad(bn.bias), shift=sh, pre_relu=True)
        co, cp = conv.out_channels, r['cin_p']
        gw = torch.empty((co, 16 * cp), dtype=torch.float32, device=g.device)
        ops.wgrad(Seg(r['col'], ksize=1), d_h, co, 16 * cp, gw, bias_grad=self._grad(conv.bias))
        self._grad(conv.weight).copy_(gw.view(co, 4, 4, cp)[..., :conv.in_channels].permute(0, 3, 1, 2))
        if not need_dx:
            return None
        wt = F.pad(_t1x1(r['wm']), (0, 0, 0, 0, 0, d_h.shape[-1] - co)).contiguous()
        dcol, _ = ops.conv_fused([Seg(d_h, ksize=1)], ops.prep_weight(wt, dt), 16 * cp)
        return ops.col2im(dcol, cp, 4, 4, 1, 1, stride=2)

    def _res_fwd(self, blk, x: Tensor, label, train: bool, tape):
        dt = self.dtype
        conv0, bn1m, mc3, conv4, bn5m, mc6 = (blk.conv[0].module, blk.conv[1].module, blk.conv[3], blk.conv[4].module,
                                              blk.conv[5].module, blk.conv[6])
        c = conv0.out_channels
        n, h, w, _ = x.shape
        sm = 1 if train else 0
        h1, st1 = ops.conv_fused([Seg(x)], ops.prep_weight(conv0.weight.detach(), dt), c, bias=conv0.bias.detach(), stats_mode=sm)
        b1 = self._bn(bn1m, st1, n * h * w, train)
        code3 = mc3.code_of_labels(label)
        h2, st2 = ops.conv_fused([Seg(h1, scale=b1[0], shift=b1[1], relu=True, code=code3)],
                                 ops.prep_weight(conv4.weight.detach(), dt), c, bias=conv4.bias.detach(), stats_mode=sm)
        b2 = self._bn(bn5m, st2, n * h * w, train)
        code6 = mc6.code_of_labels(label)
        y = ops.affine_code_res(h2, b2[0], b2[1], code6, x, post_relu=True)
        if tape is not None:
            tape.append(dict(kind='res', x=x, h1=h1, b1=b1, code3=code3, h2=h2, b2=b2, code6=code6, y=y))
        return y

    def _res_bwd(self, blk, r, g: Tensor):
        dt = self.dtype
        conv0, bn1m, conv4, bn5m = blk.conv[0].module, blk.conv[1].module, blk.conv[4].module, blk.conv[5].module
        c = conv0.out_channels
        sc2, sh2, mean2, rstd2 = r['b2']
        d_h2, g_res = ops.code_bn_bwd(g, r['code6'], r['h2'], sc2, mean2, rstd2, self._grad(bn5m.weight), self._grad(bn5m.bias),
                                      y_post=r['y'], want_gated=True)
        sc1, sh1, mean1, rstd1 = r['b1']
        h1, x = r['h1'], r['x']
        ops.wgrad(Seg(h1, scale=sc1, shift=sh1, relu=True, code=r['code3']), d_h2, c, c, self._grad(conv4.weight),
                  bias_grad=self._grad(conv4.bias))
        dz1, st = ops.conv_fused([Seg(d_h2)], ops.prep_weight(_t3x3(conv4.weight.detach()), dt), c, ocode=r['code3'], gate_x=h1,
                                 gscale=sc1, gshift=sh1, gmean=mean1, grstd=rstd1, stats_mode=2)
        n, h, w, _ = h1.shape
        d_h1 = ops.bn_backward(st, dz1, h1, n * h * w, sc1, mean1, rstd1, self._grad(bn1m.weight), self._grad(bn1m.bias))
        ops.wgrad(Seg(x), d_h1, c, c, self._grad(conv0.weight), bias_grad=self._grad(conv0.bias))
        dx, _ = ops.conv_fused([Seg(d_h1)], ops.prep_weight(_t3x3(conv0.weight.detach()), dt), c, res=g_res)
        return dx

    def _up_fwd(self, convt, x: Tensor, tape):
        """ConvTranspose2d(ci, co, 4, 2, 1) -> pre-activation output [N, 2h, 2w, pad8(co)]."""
        dt = self.dtype
        co = convt.out_channels
        cop = pad8(co)
        w = convt.weight.detach().permute(2, 3, 1, 0)                           # [4, 4, co, ci]
        wm = F.pad(w, (0, 0, 0, cop - co)).reshape(16 * cop, w.shape[-1], 1, 1).contiguous()
        dcol, _ = ops.conv_fused([Seg(x, ksize=1)], ops.prep_weight(wm, dt), 16 * cop)
        out = ops.col2im(dcol, cop, 4, 4, 1, 1, stride=2, bias=convt.bias.detach())
        if tape is not None:
            tape.append(dict(kind='up', x=x, wm=wm, out=out))
        return out

    def _up_bwd(self, convt, r, d_out: Tensor, need_dx: bool = True):
        dt = self.dtype
        co, ci = convt.out_channels, convt.in_channels
        cop = d_out.shape[-1]
        ops.colsum(d_out, co, self._grad(convt.bias))
        ddcol = ops.im2col(d_out, 4, 4, 1, 1, stride=2)
        x = r['x']
        cip = x.shape[-1]
        gw = torch.empty((16 * cop, cip), dtype=torch.float32, device=d_out.device)
        ops.wgrad(Seg(x, ksize=1), ddcol, 16 * cop, cip, gw)
        self._grad(convt.weight).copy_(gw.view(4, 4, cop, cip)[:, :, :co, :ci].permute(3, 2, 0, 1))
        if not need_dx:
            return None
        dx, _ = ops.conv_fused([Seg(ddcol, ksize=1)], ops.prep_weight(_t1x1(r['wm']), dt), ci)
        return dx

    # ---- forward -------------------------------------------------------------------------------------------------
    def encode(self, img01: Tensor, label: Tensor, train: bool, eps: Optional[Tensor], tape):
        m, dt = self.m, self.dtype
        enc = m.encoder
        ns, nr = len(m.hidden_size), m.num_res_block
        x = ops.to_nhwc(img01.contiguous(), dt)
        blocks = enc.blocks
        for i in range(ns):
            x = self._down_fwd(blocks[4 * i].module, blocks[4 * i + 1].module, blocks[4 * i + 3], x, label, train, tape)
        for r in range(nr):
            x = self._res_fwd(blocks[4 * ns + r], x, label, train, tape)
        n = x.shape[0]
        perm = self._flat_perm(x.device)
        flat = x.reshape(n, 1, 1, -1)
        wcat = torch.cat([enc.mu.weight.detach(), enc.logvar.weight.detach()])[:, perm].contiguous()
        bcat = torch.cat([enc.mu.bias.detach(), enc.logvar.bias.detach()])
        L = m.latent_size
        ml, _ = ops.conv_fused([Seg(flat, ksize=1)], ops.prep_weight(wcat, dt), 2 * L, bias=bcat)
        ml = ml.reshape(n, -1).float()
        mu, logvar = ml[:, :L].contiguous(), ml[:, L:2 * L].contiguous()
        z = mu + eps * torch.exp(0.5 * logvar) if train else mu
        if tape is not None:
            tape.append(dict(kind='latent', flat=flat, wcat=wcat, mu=mu, logvar=logvar, eps=eps, xshape=x.shape))
        return z, mu, logvar

    def decode(self, z: Tensor, label: Tensor, train: bool, tape):
        m, dt = self.m, self.dtype
        dec = m.decoder
        ns, nr = len(m.hidden_size), m.num_res_block
        c, h, w = dec.encoded_shape
        hw = h * w
        n = z.shape[0]
        perm = self._flat_perm(z.device)
        mc_z, lin, bn1 = dec.linear[0], dec.linear[1].module, dec.linear[2].module
        code_z = mc_z.code_of_labels(label)
        zt = z.to(dt).reshape(n, 1, 1, -1).contiguous()
        wl = lin.weight.detach()[perm].contiguous()
        a_lin, st = ops.conv_fused([Seg(zt, ksize=1, code=code_z)], ops.prep_weight(wl, dt), wl.shape[0],
                                   bias=lin.bias.detach()[perm].contiguous(), stats_mode=1 if train else 0)
        # BatchNorm1d over the N samples, parameters gathered into the NHWC feature order and scattered back
        if train:
            rm, rv = bn1.running_mean[perm].contiguous(), bn1.running_var[perm].contiguous()
            sc, sh, mean, rstd = ops.bn_finalize(st, n, bn1.weight.detach()[perm].contiguous(), bn1.bias.detach()[perm].contiguous(),
                                                 rm, rv, bn1.momentum, bn1.eps)
            bn1.running_mean.index_copy_(0, perm, rm)
            bn1.running_var.index_copy_(0, perm, rv)
            bn1.num_batches_tracked += 1
        else:
            sc, sh = ops.bn_eval_affine(bn1.weight.detach()[perm].contiguous(), bn1.bias.detach()[perm].contiguous(),
                                        bn1.running_mean[perm].contiguous(), bn1.running_var[perm].contiguous(), bn1.eps)
            mean = rstd = None
        blocks = dec.blocks
        code0 = blocks[0].code_of_labels(label)
        code_t = code0.repeat(1, hw)                                            # [N, hw*C]: index p*C + c
        x = ops.affine_code_res(a_lin, sc, sh, code_t, None, pre_relu=True).reshape(n, h, w, c)
        if tape is not None:
            tape.append(dict(kind='declin', zt=zt, code_z=code_z, wl=wl, a_lin=a_lin, bn=(sc, sh, mean, rstd), code_t=code_t))
        for r in range(nr):
            x = self._res_fwd(blocks[1 + r], x, label, train, tape)
        k = 1 + nr
        for _ in range(ns - 1):
            out = self._up_fwd(blocks[k].module, x, tape)
            nb, ho, wo, _ = out.shape
            bn = blocks[k + 1].module
            b = self._bn(bn, ops.channel_stats(out) if train else None, nb * ho * wo, train)
            code = blocks[k + 3].code_of_labels(label)
            code = F.pad(code, (0, out.shape[-1] - code.shape[1])) if code.shape[1] != out.shape[-1] else code
            x = ops.affine_code_res(out, self._padv(b[0], out.shape[-1]), self._padv(b[1], out.shape[-1]), code, None, pre_relu=True)
            if tape is not None:
                tape.append(dict(kind='uptail', out=out, bn=b, code=code))
            k += 4
        return self._up_fwd(blocks[k].module, x, tape)                           # logits of the final Sigmoid

    @staticmethod
    def _padv(v: Tensor, n: int) -> Tensor:
        return v if v.numel() == n else F.pad(v, (0, n - v.numel()))

    def forward(self, img: Tensor, label: Tensor, train: bool, eps: Optional[Tensor] = None, tape=None, want_grad: bool = False):
        """-> dict(loss, mu, logvar, img) with img back in (-1, 1) as NCHW fp32 (mcvae.py:133-144)."""
        m = self.m
        img01 = (img + 1) / 2
        if train and eps is None:
            eps = torch.randn(img.shape[0], m.latent_size, device=img.device)
        z, mu, logvar = self.encode(img01, label, train, eps, tape)
        logits = self.decode(z, label, train, tape)
        numel = float(img.numel())
        c = m.data_shape[0]
        target = ops.to_nhwc(img01.contiguous(), torch.float32, logits.shape[-1])
        recon, bce, dlogits = ops.bce_logits(logits, target, c, 1.0 / numel, want_grad)
        kld = 0.5 * torch.sum(mu.pow(2) + logvar.exp() - 1 - logvar)
        loss = (bce + kld) / numel
        if tape is not None:
            tape.append(dict(kind='loss', dlogits=dlogits, numel=numel))
        return {'loss': loss, 'mu': mu, 'logvar': logvar, 'img': ops.to_nchw(recon, c) * 2 - 1}

    # ---- backward ------------------------------------------------------------------------------------------------
    def backward(self, tape, label: Tensor):
        m, dt = self.m, self.dtype
        enc, dec = m.encoder, m.decoder
        ns, nr = len(m.hidden_size), m.num_res_block
        recs = list(tape)
        loss_rec = recs.pop()
        numel = loss_rec['numel']
        dblocks = dec.blocks
        # decoder, last to first
        k_last = 1 + nr + 4 * (ns - 1)
        g = self._up_bwd(dblocks[k_last].module, recs.pop(), loss_rec['dlogits'])
        k = k_last
        for _ in range(ns - 1):
            k -= 4
            tail = recs.pop()
            bn = dblocks[k + 1].module
            sc, sh, mean, rstd = tail['bn']
            cp = tail['out'].shape[-1]
            co = bn.weight.numel()
            dgam = torch.zeros(cp, dtype=torch.float32, device=g.device)
            dbet = torch.zeros(cp, dtype=torch.float32, device=g.device)
            d_out = ops.code_bn_bwd(g, tail['code'], tail['out'], self._padv(sc, cp), self._padv(mean, cp), self._padv(rstd, cp),
                                    dgam, dbet, shift=self._padv(sh, cp), pre_relu=True)
            self._grad(bn.weight).copy_(dgam[:co]); self._grad(bn.bias).copy_(dbet[:co])
            g = self._up_bwd(dblocks[k].module, recs.pop(), d_out)
        for r in reversed(range(nr)):
            g = self._res_bwd(dblocks[1 + r], recs.pop(), g)
        # decoder linear: relu(BN1d(lin)) * code  -- [N, F] tensor ops
        dl = recs.pop()
        perm = self._flat_perm(g.device)
        lin, bn1 = dec.linear[1].module, dec.linear[2].module
        sc, sh, mean, rstd = dl['bn']
        n = g.shape[0]
        a = dl['a_lin'].reshape(n, -1).float()
        gz = g.reshape(n, -1).float() * dl['code_t'] * ((a * sc + sh) > 0)
        dbeta = gz.sum(0)
        xhat = (a - mean) * rstd
        dgamma = (gz * xhat).sum(0)
        d_lin = (sc * (gz - (dbeta + xhat * dgamma) / n)).to(dt).reshape(n, 1, 1, -1).contiguous()
        self._grad(bn1.weight).index_copy_(0, perm, dgamma)
        self._grad(bn1.bias).index_copy_(0, perm, dbeta)
        F_, L = dl['wl'].shape
        gw = torch.empty((F_, L), dtype=torch.float32, device=g.device)
        gb = torch.empty(F_, dtype=torch.float32, device=g.device)
        ops.wgrad(Seg(dl['zt'], ksize=1, code=dl['code_z']), d_lin, F_, L, gw, bias_grad=gb)
        self._grad(lin.weight).index_copy_(0, perm, gw)
        self._grad(lin.bias).index_copy_(0, perm, gb)
        dz, _ = ops.conv_fused([Seg(d_lin, ksize=1)], ops.prep_weight(_t1x1(dl['wl']), dt), L, ocode=dl['code_z'])
        dz = dz.reshape(n, -1).float()[:, :L]
        # latent: z = mu + eps * exp(logvar / 2); KL term
        lat = recs.pop()
        mu, logvar, eps = lat['mu'], lat['logvar'], lat['eps']
        dmu = dz + mu / numel
        dlv = dz * eps * 0.5 * torch.exp(0.5 * logvar) + 0.5 * (logvar.exp() - 1) / numel
        dml = torch.cat([dmu, dlv], 1).to(dt).reshape(n, 1, 1, -1).contiguous()
        wcat, flat = lat['wcat'], lat['flat']
        gwc = torch.empty(wcat.shape, dtype=torch.float32, device=g.device)
        gbc = torch.empty(2 * L, dtype=torch.float32, device=g.device)
        ops.wgrad(Seg(flat, ksize=1), dml, 2 * L, wcat.shape[1], gwc, bias_grad=gbc)
        gfull = torch.empty_like(gwc)
        gfull.index_copy_(1, perm, gwc)
        self._grad(enc.mu.weight).copy_(gfull[:L]); self._grad(enc.logvar.weight).copy_(gfull[L:])
        self._grad(enc.mu.bias).copy_(gbc[:L]); self._grad(enc.logvar.bias).copy_(gbc[L:])
        wt = F.pad(_t1x1(wcat), (0, 0, 0, 0, 0, dml.shape[-1] - 2 * L)).contiguous()
        g, _ = ops.conv_fused([Seg(dml, ksize=1)], ops.prep_weight(wt, dt), wcat.shape[1])
        g = g.reshape(lat['xshape'])
        # encoder
        eblocks = enc.blocks
        for r in reversed(range(nr)):
            g = self._res_bwd(eblocks[4 * ns + r], recs.pop(), g)
        for i in reversed(range(ns)):
            g = self._down_bwd(eblocks[4 * i].module, eblocks[4 * i + 1].module, recs.pop(), g, need_dx=(i > 0))
        assert not recs
