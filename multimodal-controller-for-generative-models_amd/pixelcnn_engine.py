"""MCGatedPixelCNN on the HIP kernels: forward (logits + cross-entropy) and the hand-derived backward.
Reference chain: MCGatedPixelCNN.forward (mcpixelcnn.py:89-101) -> MCGatedMaskedConv2d.forward (:47-61) ->
MCGatedActivation.forward (:16-20).

Per layer (C = hidden):
  h_vert = vert_stack(x_v)                     fused 3x3 conv (the (k/2+1) x k kernel + crop == a 3x3 kernel whose last
                                               row is zero); the 7x7 mask-A layer: im2col (4x7 taps) + fused 1x1 conv
  s      = vert_to_horiz(h_vert) + horiz_stack(x_h)    ONE K-concatenated launch (1x1 segment + 3x3 segment, or
                                               1x1 + im2col'ed 1x4 taps for layer 0); both epilogues emit BN partial sums
  out_v  = gate_v(h_vert), out_h = gate_h(s)   gated activation kernel (BN affine + ReLU, sigmoid gate, MC code)
  x_h'   = MC(BN(conv1x1(out_h))) (+ x_h)      fused 1x1 conv (+ BN sums) and one elementwise pass
Head: conv1x1 -> BN -> ReLU -> MC folded into the prologue of the final conv1x1; cross-entropy kernel.
Backward mirrors it with the same fused kernel on transposed images, the wgrad kernel, and two-pass BN backward.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn.functional as F

from . import ops
from ._tuning import flag as _flag
from .ops import Seg, pad8

_PAIR_GATES = _flag('MCGEN_PAIR_GATES', '1') != '0'      # a layer's two gates: BatchNorm statistics / activations as one launch each

Tensor = torch.Tensor


def _t1x1(w: Tensor) -> Tensor:
    """[Cout, Cin(,1,1)] -> transposed 1x1 master weight [Cin, Cout, 1, 1]."""
    return w.reshape(w.shape[0], -1).t().contiguous().reshape(-1, w.shape[0], 1, 1)


class PixelCNNEngine:
    def __init__(self, model, dtype: torch.dtype = torch.float32):
        self.m = model
        self.dtype = dtype
        self._gsink = None
        if model.hidden_size % 8 != 0:
            raise ValueError('Not valid hidden size: the fused path needs a multiple of 8')

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

    def _bn(self, bn, stats: Optional[Tensor], count: int, train: bool):
        if train:
            sc, sh, mean, rstd = ops.bn_finalize(stats, count, bn.weight.detach(), bn.bias.detach(), bn.running_mean,
                                                 bn.running_var, bn.momentum, bn.eps)
            self._nbt.append(bn.num_batches_tracked)       # bumped together at the end of the forward (46 launches -> 1)
            return sc, sh, mean, rstd
        sc, sh = ops.bn_eval_affine(bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var, bn.eps)
        return sc, sh, None, None

    def _stack_weights(self, L):
        """Forward-orientation master weights of the two stacks as the fused kernel wants them.
        3x3 layers: [2C, C, 3, 3] with zero taps; layer 0: im2col matrices [2C, T*C, 1, 1] (tap-major)."""
        wv, wh = L.vert_stack.weight.detach(), L.horiz_stack.weight.detach()
        if L.kernel == 3:
            return F.pad(wv, (0, 0, 0, 1)), F.pad(wh, (0, 1, 1, 1))
        return (wv.permute(0, 2, 3, 1).reshape(wv.shape[0], -1, 1, 1).contiguous(),
                wh.permute(0, 2, 3, 1).reshape(wh.shape[0], -1, 1, 1).contiguous())

    # ---- weight images of a whole pass in a few launches ------------------------------------------------------------
    def _images(self, backward: bool):
        """{(layer index | 'head', name): image}.  Forward: the stacks embedded into 3x3 images, the 1x1 links, the head;
        backward: their transposed (+ flipped) counterparts.  The 7x7 layer's im2col matrices are built separately."""
        m, dt = self.m, self.dtype
        jobs, keys = [], []

        def add(key, w, **kw):
            keys.append(key); jobs.append((w.detach(), dict(kw, transpose=backward)))

        cats = {}
        for i, L in enumerate(m.layers):
            if L.kernel == 3:
                add((i, 'v'), L.vert_stack.weight, ksize=3)
                if not backward:
                    # vert_to_horiz ++ horiz_stack feed ONE K-concatenated launch: both images into one buffer
                    c2, c = L.vert_to_horiz.weight.shape[0], L.horiz_stack.weight.shape[1]
                    n1, n2 = ops.weight_image_elems(c2, c2, 1, False), ops.weight_image_elems(c2, c, 3, False)
                    buf = torch.empty(n1 + n2, dtype=dt, device=L.vert_to_horiz.weight.device)
                    cats[(i, 'v2h+h')] = buf
                    add((i, 'v2h'), L.vert_to_horiz.weight, out=buf[:n1])
                    add((i, 'h'), L.horiz_stack.weight, ksize=3, kh0=1, out=buf[n1:])
                    add((i, 'r'), L.horiz_resid[0].module.weight)
                    continue
                add((i, 'h'), L.horiz_stack.weight, ksize=3, kh0=1)
            add((i, 'v2h'), L.vert_to_horiz.weight)
            add((i, 'r'), L.horiz_resid[0].module.weight)
        oc = m.output_conv
        add(('head', 0), oc[0].module.weight)
        add(('head', 4), oc[4].module.weight, **({'k_img': pad8(oc[4].module.out_channels)} if backward else {}))
        images = dict(zip(keys, ops.prep_weight_ex_many(jobs, dt)))
        images.update(cats)
        return images

    def _code(self, mc, label):
        cached = getattr(self, '_codes', None)
        return cached[id(mc)] if cached is not None and id(mc) in cached else mc.code_of_labels(label)

    # ---- forward ------------------------------------------------------------------------------------------------
    def _layer_forward(self, L, x_v: Tensor, x_h: Tensor, label: Tensor, train: bool, tape, I, li):
        dt = self.dtype
        n, h, w, c = x_v.shape
        count = n * h * w
        sm = 1 if train else 0
        if L.mask_type == 'A':
            L.make_causal()                                   # zeroes the parameters in place, as the reference does
        k2 = L.kernel // 2
        if L.kernel == 3:
            # the (2 x 3) and (1 x 2) stacks sit at taps (0, 0) and (1, 0) of 3x3 images (zero elsewhere)
            wv = wh = None
            in_v, in_h = Seg(x_v), Seg(x_h)
            img_v, img_h = I[(li, 'v')], I[(li, 'h')]
        else:
            wv, wh = self._stack_weights(L)
            in_v = Seg(ops.im2col(x_v, k2 + 1, L.kernel, k2, k2), ksize=1)
            in_h = Seg(ops.im2col(x_h, 1, k2 + 1, 0, k2), ksize=1)
            img_v, img_h = ops.prep_weight(wv, dt), ops.prep_weight(wh, dt)
        h_vert, st_v = ops.conv_fused([in_v], img_v, 2 * c, bias=L.vert_stack.bias.detach(), stats_mode=sm)
        wimg = I.get((li, 'v2h+h'))
        if wimg is None:
            wimg = torch.cat([I[(li, 'v2h')], img_h])
        s, st_s = ops.conv_fused([Seg(h_vert, ksize=1), in_h], wimg, 2 * c,
                                 bias=L.vert_to_horiz.bias.detach(), bias2=L.horiz_stack.bias.detach(), stats_mode=sm)
        code_v, code_h = self._code(L.gate_v.mc, label), self._code(L.gate_h.mc, label)
        if train and _PAIR_GATES:
            # the two gates of a layer wait for the same two convolutions and for nothing else: their BatchNorm statistics in
            # one launch, their activations in another (4 launches -> 2 per layer)
            bv, bh = L.gate_v.bn, L.gate_h.bn
            bn_v, bn_h = ops.bn_finalize_batch([
                (st_v, count, bv.weight.detach(), bv.bias.detach(), bv.running_mean, bv.running_var, bv.momentum, bv.eps),
                (st_s, count, bh.weight.detach(), bh.bias.detach(), bh.running_mean, bh.running_var, bh.momentum, bh.eps)])
            self._nbt += [bv.num_batches_tracked, bh.num_batches_tracked]
            out_v, out_h = ops.gated_fwd_batch([(h_vert, bn_v[0], bn_v[1], code_v), (s, bn_h[0], bn_h[1], code_h)])
        else:
            bn_v = self._bn(L.gate_v.bn, st_v, count, train)
            bn_h = self._bn(L.gate_h.bn, st_s, count, train)
            out_v = ops.gated_fwd(h_vert, bn_v[0], bn_v[1], code_v)
            out_h = ops.gated_fwd(s, bn_h[0], bn_h[1], code_h)
        conv_r, bn_rm, mc_r = L.horiz_resid[0].module, L.horiz_resid[1].module, L.horiz_resid[2]
        r, st_r = ops.conv_fused([Seg(out_h, ksize=1)], I[(li, 'r')], c,
                                 bias=conv_r.bias.detach(), stats_mode=sm)
        bn_r = self._bn(bn_rm, st_r, count, train)
        code_r = self._code(mc_r, label)
        x_h_new = ops.affine_code_res(r, bn_r[0], bn_r[1], code_r, x_h if L.residual else None)
        if tape is not None:
            tape.append(dict(in_v=in_v, in_h=in_h, h_vert=h_vert, s=s, out_h=out_h, r=r, bn_v=bn_v, bn_h=bn_h, bn_r=bn_r,
                             code_v=code_v, code_h=code_h, code_r=code_r, wv=wv, wh=wh))
        return out_v, x_h_new

    def forward(self, codes: Tensor, label: Tensor, train: bool, tape=None, want_grad: bool = False):
        """-> (loss, logits NHWC, dlogits or None).  `codes` int64 [N, H, W]."""
        m, dt = self.m, self.dtype
        n, h, w = codes.shape
        x = F.embedding(codes, m.embedding.weight.detach()).to(dt).contiguous()           # [N, H, W, C] is already NHWC
        x_v = x_h = x
        layers = [] if tape is not None else None
        self._nbt = []
        I = self._images(False)
        # every MultimodalController's code rows of this batch in one launch (a row gather per module: one_hot(label) @ codebook)
        mcs = [mc for L in m.layers for mc in (L.gate_v.mc, L.gate_h.mc, L.horiz_resid[2])] + [m.output_conv[3]]
        if getattr(self, '_code_batch', None) is None or [id(x_) for x_ in self._code_batch.mcs] != [id(x_) for x_ in mcs]:
            self._code_batch = ops.CodeBatch(mcs)
        self._codes = {id(mc): cd for mc, cd in zip(mcs, self._code_batch.run_labels(label))}
        for li, L in enumerate(m.layers):
            x_v, x_h = self._layer_forward(L, x_v, x_h, label, train, layers, I, li)
        oc = m.output_conv
        conv0, bn0, mc0, conv4 = oc[0].module, oc[1].module, oc[3], oc[4].module
        count = n * h * w
        h0, st0 = ops.conv_fused([Seg(x_h, ksize=1)], I[('head', 0)], conv0.out_channels,
                                 bias=conv0.bias.detach(), stats_mode=1 if train else 0)
        bn = self._bn(bn0, st0, count, train)
        code0 = self._code(mc0, label)
        logits, _ = ops.conv_fused([Seg(h0, ksize=1, scale=bn[0], shift=bn[1], relu=True, code=code0)],
                                   I[('head', 4)], conv4.out_channels, bias=conv4.bias.detach())
        self._codes = None
        if self._nbt:
            torch._foreach_add_(self._nbt, 1)
        self._nbt = []
        rows, dlogits = ops.cross_entropy(logits, codes.reshape(-1), conv4.out_channels, want_grad)
        if tape is not None:
            tape.update(layers=layers, codes=codes, x_h=x_h, h0=h0, bn0=bn, code0=code0, dlogits=dlogits)
        return rows.mean(), logits, dlogits

    # ---- backward -----------------------------------------------------------------------------------------------
    def _conv1x1_bwd(self, conv, seg_in: Seg, dy: Tensor, need_dx: bool = True, wt=None, **dgrad_kw):
        """Weight/bias gradients of a 1x1 convolution and (optionally) its input gradient."""
        dt = self.dtype
        cout = conv.out_channels
        cin_p = seg_in.x.shape[-1]
        ops.wgrad(seg_in, dy, cout, conv.in_channels, self._grad(conv.weight), bias_grad=self._grad(conv.bias))
        if not need_dx:
            return None, None
        if wt is None:
            wt = ops.prep_weight_ex(conv.weight.detach(), dt, transpose=True, k_img=dy.shape[-1])
        return ops.conv_fused([Seg(dy, ksize=1)], wt, conv.in_channels, **dgrad_kw)

    def _layer_backward(self, L, r, g_v: Optional[Tensor], g_h: Tensor, need_dx: bool, I, li):
        """g_v / g_h: gradients w.r.t. this layer's (out_v, x_h').  Returns gradients w.r.t. (x_v, x_h)."""
        dt = self.dtype
        c = L.hidden_size
        conv_r, bn_rm = L.horiz_resid[0].module, L.horiz_resid[1].module
        sc_r, _, mean_r, rstd_r = r['bn_r']
        d_r = ops.code_bn_bwd(g_h, r['code_r'], r['r'], sc_r, mean_r, rstd_r, self._grad(bn_rm.weight), self._grad(bn_rm.bias))
        d_out_h, _ = self._conv1x1_bwd(conv_r, Seg(r['out_h'], ksize=1), d_r, wt=I[(li, 'r')])
        sc, sh, mean, rstd = r['bn_h']
        ds = ops.gated_bwd(r['s'], sc, sh, mean, rstd, r['code_h'], d_out_h, self._grad(L.gate_h.bn.weight), self._grad(L.gate_h.bn.bias))
        # s = vert_to_horiz(h_vert) + horiz_stack(x_h): both biases see sum(ds)
        c2 = 2 * c
        ops.wgrad(Seg(r['h_vert'], ksize=1), ds, c2, c2, self._grad(L.vert_to_horiz.weight), bias_grad=self._grad(L.vert_to_horiz.bias),
                  bias_grad2=self._grad(L.horiz_stack.bias))
        k2 = L.kernel // 2
        in_h = r['in_h']
        cin_h = in_h.x.shape[-1]
        gh = self._grad(L.horiz_stack.weight)
        if L.kernel == 3:
            # the (1 x 2) stack sits at taps (1, 0), (1, 1) of the 3x3 image (mcpixelcnn.py:32-35): the reduce writes that window
            ops.wgrad(in_h, ds, c2, cin_h, gh, taps=(3, 2))
        else:
            gwh = torch.empty((c2, cin_h, in_h.ksize, in_h.ksize), dtype=torch.float32, device=ds.device)
            ops.wgrad(in_h, ds, c2, cin_h, gwh)
            self._post.append(lambda: gh.copy_(gwh.reshape(c2, 1, k2 + 1, c).permute(0, 3, 1, 2)))
        # gate_v and the vertical stack
        d_hv = None
        if g_v is not None:
            sc, sh, mean, rstd = r['bn_v']
            d_hv = ops.gated_bwd(r['h_vert'], sc, sh, mean, rstd, r['code_v'], g_v, self._grad(L.gate_v.bn.weight),
                                 self._grad(L.gate_v.bn.bias))
        d_hv, _ = ops.conv_fused([Seg(ds, ksize=1)], I[(li, 'v2h')], c2, res=d_hv)
        in_v = r['in_v']
        cin_v = in_v.x.shape[-1]
        gv = self._grad(L.vert_stack.weight)
        if L.kernel == 3:
            # the (2 x 3) stack = rows 0, 1 of the 3x3 image (mcpixelcnn.py:29-31): taps 0 .. 5
            ops.wgrad(in_v, d_hv, c2, cin_v, gv, bias_grad=self._grad(L.vert_stack.bias), taps=(0, 6))
        else:
            gwv = torch.empty((c2, cin_v, in_v.ksize, in_v.ksize), dtype=torch.float32, device=ds.device)
            ops.wgrad(in_v, d_hv, c2, cin_v, gwv, bias_grad=self._grad(L.vert_stack.bias))
            self._post.append(lambda: gv.copy_(gwv.reshape(c2, k2 + 1, L.kernel, c).permute(0, 3, 1, 2)))
        if not need_dx:
            return None, None
        res_h = g_h if L.residual else None
        if L.kernel == 3:
            d_xh, _ = ops.conv_fused([Seg(ds)], I[(li, 'h')], c, res=res_h)
            d_xv, _ = ops.conv_fused([Seg(d_hv)], I[(li, 'v')], c)
        else:
            dcol_h, _ = ops.conv_fused([Seg(ds, ksize=1)], ops.prep_weight(_t1x1(r['wh']), dt), r['wh'].shape[1])
            d_xh = ops.col2im(dcol_h, c, 1, k2 + 1, 0, k2)
            if res_h is not None:
                d_xh = d_xh + res_h
            dcol_v, _ = ops.conv_fused([Seg(d_hv, ksize=1)], ops.prep_weight(_t1x1(r['wv']), dt), r['wv'].shape[1])
            d_xv = ops.col2im(dcol_v, c, k2 + 1, L.kernel, k2, k2)
        return d_xv, d_xh

    def backward(self, tape):
        """Fill the gradient of the mean cross-entropy for every parameter from the tape of one forward."""
        self._post = []
        with ops.deferred_reduces():               # every split-K reduction of the pass: a few batched launches
            self._backward_body(tape)
        for f in self._post:                       # slices of the 3x3-embedded stack gradients -> the (2x3)/(1x2) parameters
            f()
        self._post = []

    def _backward_body(self, tape):
        m, dt = self.m, self.dtype
        oc = m.output_conv
        conv0, bn0, conv4 = oc[0].module, oc[1].module, oc[4].module
        sc, sh, mean, rstd = tape['bn0']
        h0, code0 = tape['h0'], tape['code0']
        # logits = conv4(code * relu(BN(h0))): input gradient through the gate with BN-backward sums in the epilogue
        I = self._images(True)
        dz, st = self._conv1x1_bwd(conv4, Seg(h0, ksize=1, scale=sc, shift=sh, relu=True, code=code0), tape['dlogits'], wt=I[('head', 4)],
                                   ocode=code0, gate_x=h0, gscale=sc, gshift=sh, gmean=mean, grstd=rstd, stats_mode=2)
        n, h, w, _ = h0.shape
        d_h0 = ops.bn_backward(st, dz, h0, n * h * w, sc, mean, rstd, self._grad(bn0.weight), self._grad(bn0.bias))
        g_h, _ = self._conv1x1_bwd(conv0, Seg(tape['x_h'], ksize=1), d_h0, wt=I[('head', 0)])
        g_v = None                                              # the last layer's out_v feeds nothing
        layers = tape['layers']
        d_emb = None
        for i in reversed(range(len(m.layers))):
            g_v, g_h = self._layer_backward(m.layers[i], layers[i], g_v, g_h, True, I, i)
        d_x = g_v + g_h                                          # layer 0: x_v and x_h are the same embedding output
        ge = self._grad(m.embedding.weight)
        ge.zero_()
        ge.index_add_(0, tape['codes'].reshape(-1), d_x.reshape(-1, d_x.shape[-1]).float()[:, :ge.shape[1]])
