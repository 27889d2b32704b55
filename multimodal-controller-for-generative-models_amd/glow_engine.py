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

from . import ops
from .ops import Seg, pad8

Tensor = torch.Tensor


class GlowEngine:
    def __init__(self, model, dtype: torch.dtype = torch.float32):
        self.m = model
        self.dtype = dtype
        self._gsink = None          # id(param) -> gradient tensor while an autograd backward is collecting
        self.assume_initialized = False   # set by a graph capture: skip the host read of ActNorm.initialized
        self._const = {}
        self._pass = None                          # per-pass batches (ActNorm vectors, codes): _begin_pass
        self._dfr = None                           # deferred parameter-gradient reductions of a backward pass

    def _full(self, value: float, n: int, device) -> Tensor:
        key = (value, n, str(device))
        t = self._const.get(key)
        if t is None:
            t = self._const[key] = torch.full((n,), value, dtype=torch.float32, device=device)
        return t

    # ---- weight images of a whole pass in a few batched launches (ops.prep_weight_ex_many) -------------------------
    def _block_dims(self):
        """[(c, cp)] per block: logical / padded channel count seen by that block's flows."""
        c, out = self.m.data_shape[0], []
        for blk in self.m.blocks:
            c *= 4
            out.append((c, pad8(c)))
            if blk.split:
                c //= 2
        return out

    def _prepare(self, backward: bool, wmats=None):
        """Fill self._I with every weight image of the pass: {(id(module), tag): image}; self._rs with exp(3 * scale) of
        every ZeroConv2d.  Forward also computes the LU weights (self._wmat) the backward reuses."""
        m, dt = self.m, self.dtype
        zcs = [f.coupling.net[8].module for b in m.blocks for f in b.flows] + [b.prior for b in m.blocks]
        if not backward:
            rs = torch._foreach_exp(torch._foreach_mul([z.scale.detach().reshape(-1) for z in zcs], 3.0))
            self._rs = {id(z): r for z, r in zip(zcs, rs)}
            # ... and their biases times it (mcglow.py:127-130: (conv(x) + b) * exp(3 scale)): one launch, not one per flow
            self._bz = {id(z): b for z, b in zip(zcs, torch._foreach_mul([z.conv.bias.detach() for z in zcs], rs))}
            self._wmat = {}
        jobs, keys = [], []
        if not backward:
            ics = [f.invconv for b in m.blocks for f in b.flows]
            for ic, wmat in zip(ics, ops.invconv_weight_batch(ics)):      # every flow's LU weight, two launches
                self._wmat[id(ic)] = wmat

        def add(key, w, **kw):
            keys.append(key); jobs.append((w.detach(), dict(kw, transpose=backward)))

        for blk, (c, cp) in zip(m.blocks, self._block_dims()):
            for flow in blk.flows:
                net = flow.coupling.net
                conv0, an1, conv1, an5, zc = net[0].module, net[1].module, net[4].module, net[5].module, net[8].module
                hid, ic, rs = conv0.out_channels, flow.invconv, self._rs[id(zc)]
                if not backward:
                    wmat = self._wmat[id(ic)]
                    add((id(conv0), 'f'), conv0.weight, k_img=cp)
                    add((id(conv1), 'f'), conv1.weight)
                    add((id(zc), 'f'), zc.conv.weight, row_scale=rs, k_img=hid)
                    add((id(ic), 'f'), wmat, ksize=1, k_img=cp)
                else:
                    add((id(zc), 'b'), zc.conv.weight, row_scale=rs, k_img=cp)
                    add((id(conv1), 'b'), conv1.weight, row_scale=an5.scale.detach().reshape(-1))
                    add((id(conv0), 'b'), conv0.weight, row_scale=an1.scale.detach().reshape(-1), rows_img=c)
                    add((id(ic), 'b'), self._wmat[id(ic)], ksize=1, col_scale=flow.actnorm.scale.detach().reshape(-1), k_img=cp)
            pz, prs = blk.prior, self._rs[id(blk.prior)]
            if not backward:
                add((id(pz), 'f'), pz.conv.weight, row_scale=prs, k_img=(pad8(c // 2) if blk.split else cp))
            elif blk.split:
                add((id(pz), 'b'), pz.conv.weight, row_scale=prs, k_img=cp)
        self._I = dict(zip(keys, ops.prep_weight_ex_many(jobs, dt)))

    def _img(self, key, build):
        t = self._I.get(key) if getattr(self, '_I', None) else None
        return t if t is not None else build()

    def all_initialized(self) -> bool:
        flags = [mod.initialized for mod in self.m.modules() if hasattr(mod, 'initialized')]
        return bool(torch.stack([f.reshape(()) for f in flags]).min().item() != 0)

    # ---- helpers ---------------------------------------------------------------------------------------------
    def _actnorm(self, an, x_stats_fn, count, train: bool, cp: int):
        """Prologue vectors (a, b) of an ActNorm; runs the data-dependent init on the first training forward."""
        cached = self._pass.get('an') if self._pass else None
        if cached is not None:
            return cached[id(an)]
        if train and not self.assume_initialized and int(an.initialized) == 0:
            ops.actnorm_init(x_stats_fn(), count, an.loc.data, an.scale.data)
            an.initialized.fill_(1)
        return ops.actnorm_affine(an.loc.data, an.scale.data, cp, with_negloc=True)

    def _zero_conv_image(self, zc, cin_pad: int):
        """ZeroConv2d (mcglow.py:119-130) as weight image + bias with exp(3*scale) folded into the rows."""
        rs = getattr(self, '_rs', {}).get(id(zc)) if getattr(self, '_I', None) else None
        if rs is None:
            rs = torch.exp(zc.scale.detach().reshape(-1) * 3)
        img = self._img((id(zc), 'f'), lambda: ops.prep_weight_ex(zc.conv.weight.detach(), self.dtype, row_scale=rs, k_img=cin_pad))
        bz = getattr(self, '_bz', {}).get(id(zc)) if getattr(self, '_I', None) else None
        return img, (bz if bz is not None else zc.conv.bias.detach() * rs)

    def _coupling_net(self, cp_net, x: Tensor, c: int, codes, train: bool, saved=None):
        """AffineCoupling.net on the first c/2 channels of x -> [log_s | t] (c channels)."""
        net = cp_net
        dt = self.dtype
        n, h, w, cp = x.shape
        count = n * h * w
        conv0, an1, mc1, conv1, an5, mc2, zc = (net[0].module, net[1].module, net[3], net[4].module, net[5].module,
                                                 net[7], net[8].module)
        need1 = train and not self.assume_initialized and int(an1.initialized) == 0
        # the image is zero over the input channels >= c/2: the coupling net reads only the first half of x
        w0img = self._img((id(conv0), 'f'), lambda: ops.prep_weight_ex(conv0.weight.detach(), dt, k_img=cp))
        h1, st1 = ops.conv_fused([Seg(x)], w0img, conv0.out_channels, bias=conv0.bias,
                                 stats_mode=1 if need1 else 0)
        hid = conv0.out_channels
        a1, b1, nl1 = self._actnorm(an1, lambda: st1, count, train, hid)
        need5 = train and not self.assume_initialized and int(an5.initialized) == 0
        h2, st2 = ops.conv_fused([Seg(h1, ksize=1, scale=a1, shift=b1, relu=True, code=codes[0])],
                                 self._img((id(conv1), 'f'), lambda: ops.prep_weight(conv1.weight.detach(), dt)), hid, bias=conv1.bias,
                                 stats_mode=1 if need5 else 0)
        a5, b5, nl5 = self._actnorm(an5, lambda: st2, count, train, hid)
        wz, bz = self._zero_conv_image(zc, hid)
        hz, _ = ops.conv_fused([Seg(h2, scale=a5, shift=b5, relu=True, code=codes[1])], wz, c, bias=bz, cy=cp)
        if saved is not None:
            saved.update(h1=h1, h2=h2, a1=a1, b1=b1, a5=a5, b5=b5, nl1=nl1, nl5=nl5)
        return hz

    def _flow_forward(self, flow, x: Tensor, c: int, indicator: Tensor, logdet: Tensor, train: bool, tape=None, label=None) -> Tensor:
        dt = self.dtype
        n, h, w, cp = x.shape
        a, b, nl = self._actnorm(flow.actnorm, lambda: ops.channel_stats(x), n * h * w, train, cp)
        ic = flow.invconv
        wmat = self._wmat.get(id(ic)) if getattr(self, '_I', None) else None
        if wmat is None:
            wmat, _ = ops.invconv_weight(ic.w_p, ic.w_l.data, ic.w_u.data, ic.w_s.data, ic.s_sign)
        out, _ = ops.conv_fused([Seg(x, ksize=1, scale=a, shift=b)],
                                self._img((id(ic), 'f'), lambda: ops.prep_weight_ex(wmat, dt, 1, k_img=cp)), c, cy=cp)
        # parameter-only log-determinants: H*W * (sum log|scale| + sum w_s)   (mcglow.py:46-47,101)
        if not (self._pass and self._pass.get('logdet')):
            ops.glow_param_logdet(flow.actnorm.scale.detach(), ic.w_s.detach(), h * w, logdet)
        net = flow.coupling.net
        codes = self._codes(net, indicator, label)
        rec = None if tape is None else dict(x=x, a=a, b=b, nl=nl, out=out, codes=codes, wmat=wmat)
        hz = self._coupling_net(net, out, c, codes, train, rec)
        if tape is not None:
            rec['hz'] = hz
            tape.append(rec)
        return ops.glow_coupling(out, hz, c, logdet, reverse=False, accumulate=True)

    def _codes(self, net, indicator, label):
        cached = self._pass.get('codes') if self._pass else None
        if cached is not None:
            return cached[id(net[3])], cached[id(net[7])]
        if label is not None:
            return net[3].code_of_labels(label), net[7].code_of_labels(label)
        return net[3].code(indicator), net[7].code(indicator)

    def _begin_pass(self, train: bool, label, n: int, logdet):
        """Per-pass batches of what depends on the parameters (and labels) alone: every ActNorm's prologue vectors, every
        MultimodalController's codes, all parameter-only log-determinants -- one launch per kind instead of one per module
        (~390 of a step's launches).  Needs initialised ActNorms (the data-dependent init runs module by module)."""
        self._pass = {}
        if not (self.assume_initialized or not train):
            return
        m = self.m
        ans, cps, lds = [], [], []
        for blk, (c, cp) in zip(m.blocks, self._block_dims()):
            for flow in blk.flows:
                net = flow.coupling.net
                hid = net[0].module.out_channels
                ans += [flow.actnorm, net[1].module, net[5].module]; cps += [cp, hid, hid]
        self._pass['an'] = {id(an): v for an, v in zip(ans, ops.actnorm_affine_batch(ans, cps))}
        if label is not None:
            mcs = [mc for blk in m.blocks for f in blk.flows for mc in (f.coupling.net[3], f.coupling.net[7])]
            if getattr(self, '_code_batch', None) is None or [id(x) for x in self._code_batch.mcs] != [id(x) for x in mcs]:
                self._code_batch = ops.CodeBatch(mcs)
            self._pass['codes'] = {id(mc): cd for mc, cd in zip(mcs, self._code_batch.run_labels(label))}
        if logdet is not None:
            hw = int(m.data_shape[1]) * int(m.data_shape[2])
            items = []
            for blk in m.blocks:
                hw //= 4
                items += [(f.actnorm.scale.detach(), f.invconv.w_s.detach(), hw) for f in blk.flows]
            ops.glow_param_logdet_batch(items, logdet)
            self._pass['logdet'] = True

    def _flow_reverse(self, flow, y: Tensor, c: int, indicator: Tensor, label=None) -> Tensor:
        dt = self.dtype
        cp = y.shape[-1]
        net = flow.coupling.net
        codes = self._codes(net, indicator, label)
        hz = self._coupling_net(net, y, c, codes, False)
        x = ops.glow_coupling(y, hz, c, None, reverse=True)
        ic = flow.invconv
        _, winv = ops.invconv_weight(ic.w_p, ic.w_l.data, ic.w_u.data, ic.w_s.data, ic.s_sign, inverse=True)
        # ActNorm.reverse folded in: x_prev = (W^-1 x) / scale - loc
        an = flow.actnorm
        wimg = ops.prep_weight_ex(winv, dt, 1, row_scale=1.0 / an.scale.detach().reshape(-1), k_img=cp)
        out, _ = ops.conv_fused([Seg(x, ksize=1)], wimg, c, bias=-an.loc.detach().reshape(-1), cy=cp)
        return out

    # ---- forward: bits per dimension -----------------------------------------------------------------------------
    def forward(self, img: Tensor, indicator: Tensor, noise: Tensor, train: bool, tape=None, label=None):
        m, dt = self.m, self.dtype
        n = img.shape[0]
        self._prepare(False)
        x0 = img * 0.5 + noise / 256                                   # mcglow.py:298-299
        c = m.data_shape[0]
        x = ops.to_nhwc(x0.contiguous(), dt)
        logdet = torch.zeros(n, dtype=torch.float32, device=img.device)
        logp = torch.zeros(n, dtype=torch.float32, device=img.device)
        self._begin_pass(train, label, n, logdet)
        zs: List[Tensor] = []
        for blk in m.blocks:
            x = ops.glow_squeeze(x, c)
            c *= 4
            brec = None if tape is None else dict(flows=[], c=c)
            for flow in blk.flows:
                x = self._flow_forward(flow, x, c, indicator, logdet, train, None if tape is None else brec['flows'], label)
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
                if tape is not None:
                    brec.update(x=x, keep=keep, prior=prior)
                x, c = keep, half
            else:
                wz, bz = self._zero_conv_image(blk.prior, cp)
                prior, _ = ops.conv_fused([Seg(torch.zeros_like(x))], wz, 2 * c, bias=bz)
                ops.gaussian_logp(x, 0, prior, c, logp)
                zs.append(ops.to_nchw(x, c))
                if tape is not None:
                    brec.update(x=x, prior=prior)
            if tape is not None:
                tape.append(brec)
        n_pixel = float(img[0].numel())
        loss = -(-math.log(256.) * n_pixel + logdet + logp) / (math.log(2.) * n_pixel)       # loss_fn, mcglow.py:283-293
        loss = torch.where(torch.isnan(loss), torch.zeros_like(loss), loss) if train else loss[~torch.isnan(loss)]
        self._pass = None                           # (the batches belong to this pass's parameters)
        return loss.mean(), zs

    # ---- backward: gradients of the mean bits/dim w.r.t. every parameter -------------------------------------------
    # Autograd of the same reference lines, written out.  g0 = d loss / d (logdet_n | logp_n) = -1 / (N log2 n_pixel)
    # (train-mode loss: NaN samples are replaced by 0 in the reference; a batch that produces NaNs is not handled here).
    def _grad(self, p: Tensor) -> Tensor:
        if self._gsink is not None:
            g = self._gsink.get(id(p))
            if g is None:
                g = self._gsink[id(p)] = torch.zeros_like(p)
            return g
        if p.grad is None:
            p.grad = torch.zeros_like(p)
        return p.grad

    def _zero_conv_bwd(self, zc, seg_in, out: Tensor, dout: Tensor, cout: int, cin: int, need_dx: bool, res=None, **dgrad_kw):
        """ZeroConv2d backward: out = (conv(A) + b) * exp(3 scale).  Returns (dA, stats) when need_dx."""
        dt = self.dtype
        rs = getattr(self, '_rs', {}).get(id(zc)) if getattr(self, '_I', None) else None
        if rs is None:
            rs = torch.exp(zc.scale.detach().reshape(-1) * 3)
        (self._dfr or ops).prod_colsum(out, dout, cout, self._grad(zc.scale).view(-1), alpha=3.0)
        if seg_in is not None:
            # weight / bias gradients land in place: the split-K reduce scales row co by exp(3 * scale[co])
            ops.wgrad(seg_in, dout, cout, cin, self._grad(zc.conv.weight), bias_grad=self._grad(zc.conv.bias), row_scale=rs)
        else:
            gb = torch.empty(cout, dtype=torch.float32, device=dout.device)
            self._grad(zc.conv.weight).zero_()
            ops.colsum(dout, cout, gb)
            self._grad(zc.conv.bias).copy_(gb * rs)
        if not need_dx:
            return None, None
        wt = self._img((id(zc), 'b'), lambda: ops.prep_weight_ex(zc.conv.weight.detach(), dt, transpose=True, row_scale=rs, k_img=dout.shape[-1]))
        return ops.conv_fused([Seg(dout)], wt, cin, res=res, **dgrad_kw)

    def _flow_backward(self, flow, r, dy: Tensor, c: int, g0: float) -> Tensor:
        dt = self.dtype
        out, hz, x, codes = r['out'], r['hz'], r['x'], r['codes']
        n, h, w, cp = out.shape
        ld_coef = g0 * n * h * w
        net = flow.coupling.net
        conv0, an1, conv1, an5, zc = net[0].module, net[1].module, net[4].module, net[5].module, net[8].module
        hid = conv0.out_channels
        dev = out.device
        ones = self._full(1.0, hid, dev)
        dv, dhz = ops.glow_coupling_bwd(out, hz, dy, c, g0)
        # ZeroConv2d <- MC <- ReLU <- ActNorm(5)
        v5, st5 = self._zero_conv_bwd(zc, Seg(r['h2'], scale=r['a5'], shift=r['b5'], relu=True, code=codes[1]), hz, dhz, c, hid,
                                      True, ocode=codes[1], gate_x=r['h2'], gscale=r['a5'], gshift=r['b5'],
                                      gmean=r['nl5'], grstd=ones, stats_mode=2)
        (self._dfr or ops).actnorm_bwd(st5, an5.scale.detach(), 0.0, False, self._grad(an5.loc), self._grad(an5.scale))
        s5 = an5.scale.detach().reshape(-1)
        # 1x1 conv <- MC <- ReLU <- ActNorm(1)
        ops.wgrad(Seg(r['h1'], ksize=1, scale=r['a1'], shift=r['b1'], relu=True, code=codes[0]), v5, hid, hid,
                  self._grad(conv1.weight), bias_grad=self._grad(conv1.bias), row_scale=s5)
        w1t = self._img((id(conv1), 'b'), lambda: ops.prep_weight_ex(conv1.weight.detach(), dt, transpose=True, row_scale=s5))
        v1, st1 = ops.conv_fused([Seg(v5, ksize=1)], w1t, hid, ocode=codes[0], gate_x=r['h1'],
                                 gscale=r['a1'], gshift=r['b1'], gmean=r['nl1'], grstd=ones, stats_mode=2)
        (self._dfr or ops).actnorm_bwd(st1, an1.scale.detach(), 0.0, False, self._grad(an1.loc), self._grad(an1.scale))
        s1 = an1.scale.detach().reshape(-1)
        # 3x3 conv on the first c/2 channels of v; its input gradient joins the direct coupling gradient dv
        ops.wgrad(Seg(out), v1, hid, c // 2, self._grad(conv0.weight), bias_grad=self._grad(conv0.bias), row_scale=s1)
        w0t = self._img((id(conv0), 'b'), lambda: ops.prep_weight_ex(conv0.weight.detach(), dt, transpose=True, row_scale=s1, rows_img=c))   # rows >= c/2: zero
        dvt, _ = ops.conv_fused([Seg(v1)], w0t, c, res=dv, cy=cp)
        # invertible 1x1 conv <- ActNorm
        an, ic = flow.actnorm, flow.invconv
        gW = torch.empty((c, cp), dtype=torch.float32, device=dev)
        ops.wgrad(Seg(x, ksize=1, scale=r['a'], shift=r['b']), dvt, c, cp, gW)
        gl, gu, gs = self._grad(ic.w_l), self._grad(ic.w_u), self._grad(ic.w_s)
        # the LU-parameter gradients need the reduced dW: after the pass's batched split-K reduction
        if self._dfr is not None:
            self._dfr.invconv_bwd(ic, gW, ld_coef, gl, gu, gs)
        else:
            self._post.append(lambda: ops.invconv_bwd(ic.w_p, ic.w_l.data, ic.w_u.data, ic.w_s.data, ic.s_sign, gW, ld_coef, gl, gu, gs))
        s = an.scale.detach().reshape(-1)
        wmat = r['wmat']
        wt = self._img((id(ic), 'b'), lambda: ops.prep_weight_ex(wmat, dt, 1, transpose=True, col_scale=s, k_img=cp))   # [ci, co] = W[co, ci] * s[ci]
        dx, st = ops.conv_fused([Seg(dvt, ksize=1)], wt, c, cy=cp, gate_x=x,
                                gscale=self._full(0.0, c, dev), gshift=self._full(1.0, c, dev),
                                gmean=r['nl'], grstd=self._full(1.0, c, dev),
                                stats_mode=2)
        (self._dfr or ops).actnorm_bwd(st, an.scale.detach(), ld_coef, True, self._grad(an.loc), self._grad(an.scale))
        return dx

    def backward(self, tape, n: int, n_pixel: float):
        """Fill `.grad` of every parameter with d(mean bits/dim)/d(parameter) from the tape of one forward."""
        m = self.m
        g0 = -1.0 / (n * math.log(2.) * n_pixel)
        self._post = []
        self._prepare(True)
        self._dfr = ops.GlowDeferred()             # parameter-gradient reductions nothing in the pass reads: batched at its end
        try:
            with ops.deferred_reduces():           # every split-K reduction of the pass: a handful of batched launches
                dkeep = self._backward_body(tape, g0)
            for f in self._post:
                f()
            self._dfr.run()
        finally:
            self._dfr = None
            self._post = []
        return dkeep

    def _backward_body(self, tape, g0: float):
        m = self.m
        dkeep = None
        for blk, rec in reversed(list(zip(m.blocks, tape))):
            x, c = rec['x'], rec['c']
            dy = torch.zeros_like(x)
            if blk.split:
                half = c // 2
                dprior = ops.gaussian_logp_bwd(x, half, rec['prior'], half, dy, half, g0, False)
                dk, _ = self._zero_conv_bwd(blk.prior, Seg(rec['keep']), rec['prior'], dprior, c, half, True, res=dkeep)
                ops.copy_channels(dk, 0, dy, 0, half)
            else:
                dprior = ops.gaussian_logp_bwd(x, 0, rec['prior'], c, dy, 0, g0, False)
                self._zero_conv_bwd(blk.prior, None, rec['prior'], dprior, 2 * c, c, False)
            for flow, frec in reversed(list(zip(blk.flows, rec['flows']))):
                dy = self._flow_backward(flow, frec, dy, c, g0)
            dkeep = ops.glow_unsqueeze(dy, c)
        return dkeep

    # ---- reverse: reconstruction / sampling ------------------------------------------------------------------------
    def reverse(self, zs: List[Tensor], indicator: Tensor, reconstruct: bool, label=None) -> Tensor:
        m, dt = self.m, self.dtype
        self._I = None                              # images are built per use on this path
        self._pass = None
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
                inp = self._flow_reverse(flow, inp, c, indicator, label)
            x = ops.glow_unsqueeze(inp, c)
        return torch.clamp(ops.to_nchw(x, m.data_shape[0]), -.5, .5) * 2
