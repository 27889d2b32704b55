"""Fused forward/backward schedules of the MCGAN generator and discriminator.

The nn.Module trees in ``models/mcgan.py`` keep the reference's structure (for
``state_dict`` / ``apply`` / ``create`` / ``transit``); their forwards hand the
whole network to the engines below, which launch one fused convolution per conv
layer (prologue = BN-apply + ReLU + nearest-upsample + MultimodalController code,
epilogue = bias / avg-pool / residual / next-BN statistics) plus the small
finalize kernels, and the hand-derived backward of the same chain
("Backward math for the fused kernels").

Reference op chains restated here: GenResBlock / Generator (mcgan.py:9-69),
FirstDisResBlock / DisResBlock / Discriminator (mcgan.py:72-181), spectral norm
(models/utils.py:17-21 -> torch.nn.utils.spectral_norm).

Parameters of a network live in ONE flat fp32 buffer (``FlatState``); gradients
are produced into a flat buffer of the same layout, so Adam and the data-parallel
all-reduce are one launch / one message per network.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.nn as nn

from . import ops
from ._tuning import flag as _flag
from .ops import Seg

Tensor = torch.Tensor


def _bump(t: Tensor):
    """Tell autograd's version counter that a kernel wrote `t` behind torch's back."""
    torch.autograd.graph.increment_version(t)


class FlatState:
    """Re-points a list of tensors (parameters or buffers) at slices of one flat fp32
    buffer, so that whole-network kernels (spectral norm, Adam, gradient all-reduce)
    address them as base + offset.  Re-flattens when the tensors were moved
    (``module.to(device)``) or replaced."""

    def __init__(self, tensors: List[Tensor]):
        self.tensors = tensors
        self.flat: Optional[Tensor] = None
        self.offsets: List[int] = []
        self._index = {id(t): i for i, t in enumerate(tensors)}

    def ensure(self) -> Tensor:
        ts = self.tensors
        ok = self.flat is not None
        if ok:
            base = self.flat.data_ptr()
            for t, off in zip(ts, self.offsets):
                if t.data_ptr() != base + 4 * off:
                    ok = False
                    break
        if not ok:
            dev = ts[0].device
            total, offs = 0, []
            for t in ts:
                offs.append(total)
                total += (t.numel() + 3) // 4 * 4            # keep every slice 16-byte aligned
            flat = torch.zeros(total, dtype=torch.float32, device=dev)
            for t, off in zip(ts, offs):
                flat[off:off + t.numel()].copy_(t.detach().reshape(-1))
                t.data = flat[off:off + t.numel()].view(t.shape)
            self.flat, self.offsets = flat, offs
        return self.flat

    def offset_of(self, t: Tensor) -> int:
        return self.offsets[self._index[id(t)]]

    def view_of(self, flat: Tensor, t: Tensor) -> Tensor:
        """The slice of another flat buffer (same layout) that corresponds to tensor t."""
        off = self.offset_of(t)
        return flat[off:off + t.numel()].view(t.shape)

    def views(self, flat: Tensor) -> List[Tensor]:
        return [self.view_of(flat, t) for t in self.tensors]


class _BN:
    """Per-forward BatchNorm state: affine used by the consumer's prologue + what backward needs."""
    __slots__ = ('scale', 'shift', 'mean', 'rstd', 'count')


_PAIR_WGRAD = _flag('MCGEN_PAIR_WGRAD', '1') != '0'
# one-launch multi-round power iteration (mcgen_sn_power_iter_fused): opt-in -- one workgroup per layer streams W through a
# single CU and measured 0.08 ms / iteration SLOWER than the four row-sliced kernels that fill the chip (tools/ab_bench.sh)
_SN_FUSED = _flag('MCGEN_SN_FUSED', '0') == '1'
# training-mode power iterations as 2 launches per round + 1 (ops.sn_power_iter_rounds) instead of 4 per round
_SN_ROUNDS = _flag('MCGEN_SN_ROUNDS', '1') != '0'
_BUCKETS = _flag('MCGEN_BUCKETS', '1') != '0'        # two gradient buckets per network (callers ask for them only when world > 1)
# mode-compacted forward convolutions (bf16, maps >= 16x16, conv_a launches): opt-in -- measured x1.10 on those launches
# (tools/bench_mc.py), about 0.5 % of the step after the map / K-major image launches are paid: see DESIGN.md section 4.6
_MC = _flag('MCGEN_MC', '0') == '1'
# compacted activations between the launches of the FORWARD-ONLY grouped generator pass (producer-side compacted store +
# gathered-K consumer, conv_fused.hip "gk"): tools/bench_mc.py gk measured x1.24-1.31 on the consumer launches
_GK = _flag('MCGEN_GK', '1') != '0'
_LOWRES_SC_BWD = _flag('MCGEN_LOWRES_SC_BWD', '1') != '0'
# per-mode DENSE weight images for the launches that read compacted activations (instead of the gathered-K form's per-sample
# row gather): needs one-hot indicators (the mode of an image is its label) and few modes (10 x 0.8 MB per layer at CIFAR-10)
_PM = _flag('MCGEN_PM', '1') != '0'
_PREP_CODES = _flag('MCGEN_PREP_CODES', '1') != '0'   # a discriminator pass's codes ride in its weight-image launch
_PM_HEAD = _flag('MCGEN_PM_HEAD', '1') != '0'         # the last block stores its output compacted for the image head
_PM_MAX_MODES = 16
class Nhwc:
    """An image batch in the engines' own layout ([N, H, W, C padded to 8] of the compute dtype, padding channels zero) with
    its true channel count: what the trainer hands from one engine to the other, instead of converting to the module
    boundary's NCHW fp32 and straight back (the values are the same bits either way)."""
    __slots__ = ('t', 'c')

    def __init__(self, t: Tensor, c: int):
        self.t, self.c = t, c

    @property
    def shape(self):                                        # the NCHW shape a caller would see
        n, h, w, _ = self.t.shape
        return torch.Size((n, self.c, h, w))


_SN_SNAP = _flag('MCGEN_SN_SNAP', '1') != '0'    # the power iteration's kernels write the forward's u/v copy (0: clone afterwards)
# FirstDisResBlock: the 1x1 shortcut as a second K segment of conv2's launch (0: its own launch + a residual read)
_D0_FUSE = _flag('MCGEN_D0_FUSE', '1') != '0'
# grouped passes: BatchNorm statistics groups finalized in parallel, running statistics of the whole pass in one launch
_BN_PAR = _flag('MCGEN_BN_PAR', '1') != '0'
_pending_counters: Dict[int, List[Tensor]] = {}
_pending_running: List = []                     # (running_mean, running_var, mean [groups, C], unb [groups, C], momentum)


def _flush_counters():
    for k, ts in _pending_counters.items():
        if ts:
            torch._foreach_add_(ts, k)
    _pending_counters.clear()
    if _pending_running:
        ops.bn_running_batch(_pending_running)
        _pending_running.clear()


def _bn_forward(bn: nn.BatchNorm2d, stats: Optional[Tensor], count: int, train: bool, fold: int = 1, groups: int = 1) -> _BN:
    """`groups` > 1: the batch is `groups` independent BatchNorm batches (`count` elements per channel each): the affine
    comes back as [groups, C], the running statistics and num_batches_tracked advance `groups` times."""
    s = _BN()
    s.count = count
    if train:
        mom = 0.1 if bn.momentum is None else bn.momentum
        if groups > 1 and _BN_PAR and bn.running_mean is not None:      # (track_running_stats=False: the serial path takes NULL)
            # the groups in parallel; the running statistics of all layers of the pass in one launch at its end (_flush_counters)
            s.scale, s.shift, s.mean, s.rstd, unb = ops.bn_finalize_par(stats, count, bn.weight, bn.bias, bn.eps, fold=fold, groups=groups)
            _pending_running.append((bn.running_mean, bn.running_var, s.mean, unb, mom))
        else:
            s.scale, s.shift, s.mean, s.rstd = ops.bn_finalize(stats, count, bn.weight, bn.bias, bn.running_mean,
                                                               bn.running_var, mom, bn.eps, fold=fold, groups=groups)
        if bn.running_mean is not None:
            _bump(bn.running_mean); _bump(bn.running_var)
            _pending_counters.setdefault(groups, []).append(bn.num_batches_tracked)   # bumped together at the end of the forward
    else:
        s.scale, s.shift = ops.bn_eval_affine(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps)
        s.mean, s.rstd = None, None
    return s


# ============================================================================================= #
#  Generator
# ============================================================================================= #
class GeneratorEngine:
    def __init__(self, gen, dtype: torch.dtype = torch.float32):
        self.gen = gen
        self.dtype = dtype
        self.flat_p = FlatState(list(gen.parameters()))
        self._img_key = None
        self.img: Dict[str, Tensor] = {}
        self._prep_fwd = self._prep_bwd = None
        lin, res, head_bn, head_mc, head_conv = self._layers()
        self._codes = ops.CodeBatch([m for b in res for m in (b.mc_1, b.mc_2)] + [head_mc])

    def _layers(self):
        g = self.gen
        blocks = list(g.blocks.children())
        res = [b for b in blocks if hasattr(b, 'mc_1')]
        n = len(res)
        head_bn, head_mc, head_conv = blocks[n].module, blocks[n + 2], blocks[n + 3].module
        return g.linear.module, res, head_bn, head_mc, head_conv

    # ---- weight images -------------------------------------------------------------------------
    def refresh_images(self, force: bool = False):
        """(Re)build the kernel weight images when a parameter changed (torch version counters;
        kernels that write parameters bump them, see fused_adam)."""
        lin, res, head_bn, head_mc, head_conv = self._layers()
        ws = [lin.weight, lin.bias, head_conv.weight]
        for b in res:
            ws += [b.conv[4].module.weight, b.conv[8].module.weight, b.shortcut[2].module.weight,
                   b.conv[8].module.bias, b.shortcut[2].module.bias]
        cbs = ([m.codebook for b in res for m in (b.mc_1, b.mc_2)] + [head_mc.codebook]) if self._pm_enabled() else []
        pm_key = tuple((c.data_ptr(), tuple(c.shape), c._version) for c in cbs)
        key = (self.dtype, tuple((w.data_ptr(), w._version) for w in ws), pm_key)
        if not force and key == self._img_key:
            return
        if getattr(self, '_pm_key', None) != pm_key:            # a codebook changed (create / transit): the per-mode jobs are stale
            self._prep_fwd = None
            self._pm_key = pm_key
        dt = self.dtype
        dev = lin.weight.device
        c0 = lin.out_features // 16

        def buf(name, numel, dtype):
            # persistent buffers: a captured HIP graph keeps reading the same addresses
            t = self.img.get(name)
            if t is None or t.numel() != numel or t.dtype != dtype or t.device != dev:
                t = torch.empty(numel, dtype=dtype, device=dev)
                self.img[name] = t
            return t

        if self._prep_fwd is None or self._prep_fwd.dtype != dt or not self._prep_fwd.valid():
            jobs = [(lin.weight, buf('lin', ops.weight_image_elems(lin.out_features, lin.in_features, 1), dt), False, 16, -1, 1.0)]
            for i, b in enumerate(res):
                w1, w2, wsc = b.conv[4].module.weight, b.conv[8].module.weight, b.shortcut[2].module.weight
                jobs.append((w1, buf(f'b{i}.w1', ops.weight_image_elems(w1.shape[0], w1.shape[1], 3), dt), False, 1, -1, 1.0))
                n2 = ops.weight_image_elems(w2.shape[0], w2.shape[1], 3)
                ns = ops.weight_image_elems(wsc.shape[0], wsc.shape[1], 1)
                cat = buf(f'b{i}.w2s', n2 + ns, dt)
                jobs.append((w2, cat[:n2], False, 1, -1, 1.0))
                jobs.append((wsc, cat[n2:], False, 1, -1, 1.0))
            jobs.append((head_conv.weight, buf('head', ops.weight_image_elems(head_conv.out_channels, head_conv.in_channels, 3), dt),
                         False, 1, -1, 1.0))
            if self._gk_enabled():
                # K-major images of the launches that read compacted activations in the grouped forward-only pass:
                # conv_b ++ shortcut of every block from 16x16 up, conv_a of every block from 32x32 up
                for i, b in enumerate(res):
                    if not self._gk_block(i):
                        continue
                    w1, w2, wsc = b.conv[4].module.weight, b.conv[8].module.weight, b.shortcut[2].module.weight
                    n2 = ops.weight_image_k_elems(w2.shape[0], w2.shape[1], 3)
                    ns = ops.weight_image_k_elems(wsc.shape[0], wsc.shape[1], 1)
                    catk = buf(f'b{i}.w2sk', n2 + ns, dt)
                    jobs.append((w2, catk[:n2], False, 1, -1, 1.0, True))
                    jobs.append((wsc, catk[n2:], False, 1, -1, 1.0, True))
                    if i > 0 and self._gk_block(i - 1):
                        jobs.append((w1, buf(f'b{i}.w1g', ops.weight_image_k_elems(w1.shape[0], w1.shape[1], 3), dt), False, 1, -1, 1.0, True))
            if self._pm_enabled():
                # per-mode dense images of the same launches: for every mode m, the chunked image whose input channels are the
                # mode's active ones in order (PrepBatch kmap = the cidx part of the mode's compaction record), padded with
                # zero columns to the compacted pitch -- conv_b ++ shortcut as two jobs into one slot, conv_a where x arrives
                # compacted.  img['b{i}.w2sm'] / ['b{i}.w1m'] hold the M sets back to back.
                self._pm_maps = {}
                for i, b in enumerate(res):
                    if not self._gk_block(i):
                        continue
                    w1, w2, wsc = b.conv[4].module.weight, b.conv[8].module.weight, b.shortcut[2].module.weight
                    cap_h, cm2 = self._cap(b.mc_2), self._mode_maps(b.mc_2)
                    x_compact = i > 0 and self._gk_block(i - 1)
                    cap_x, cm1 = (self._cap(b.mc_1), self._mode_maps(b.mc_1)) if x_compact else (None, None)
                    if cap_h is None or (x_compact and cap_x is None):
                        continue
                    modes = b.mc_2.codebook.shape[0]
                    n2 = ops.weight_image_elems(w2.shape[0], cap_h, 3)
                    ns = ops.weight_image_elems(wsc.shape[0], cap_x if x_compact else wsc.shape[1], 1)
                    slot = buf(f'b{i}.w2sm', modes * (n2 + ns), dt).view(modes, n2 + ns)
                    # output channels in the order their consumer keeps them (mcgen_conv_t.yperm): h by mc_2, y by the next mask
                    ymc = self._y_mc(i, True)
                    py = self._mode_perms(ymc) if ymc is not None else None
                    ph = self._mode_perms(b.mc_2)
                    for m in range(modes):
                        ry = None if py is None else py[m]
                        jobs.append((w2, slot[m, :n2], False, 1, -1, 1.0, False, cm2[m, w2.shape[1]:], cap_h, ry))
                        if x_compact:
                            jobs.append((wsc, slot[m, n2:], False, 1, -1, 1.0, False, cm1[m, wsc.shape[1]:], cap_x, ry))
                        else:
                            jobs.append((wsc, slot[m, n2:], False, 1, -1, 1.0, False, None, 0, ry))
                    if x_compact:
                        n1 = ops.weight_image_elems(w1.shape[0], cap_x, 3)
                        slot1 = buf(f'b{i}.w1m', modes * n1, dt).view(modes, n1)
                        for m in range(modes):
                            jobs.append((w1, slot1[m], False, 1, -1, 1.0, False, cm1[m, w1.shape[1]:], cap_x, ph[m]))
                # the image head behind a compacting last block (conv_head.hip reads 5 chunks instead of 8)
                if res and self._y_mc(len(res) - 1, True) is head_mc and f'b{len(res) - 1}.w2sm' in self.img:
                    cap_o, cmo = self._cap(head_mc), self._mode_maps(head_mc)
                    modes = head_mc.codebook.shape[0]
                    nh = ops.weight_image_elems(head_conv.out_channels, cap_o, 3)
                    sloth = buf('headm', modes * nh, dt).view(modes, nh)
                    for m in range(modes):
                        jobs.append((head_conv.weight, sloth[m], False, 1, -1, 1.0, False, cmo[m, head_conv.in_channels:], cap_o))
            if self._mc_enabled():
                # K-major images of the blocks whose maps are large enough for a tile to lie inside one image (the
                # mode-compacted kernel gathers the active channels' rows from them): conv_a, and conv_b ++ shortcut
                for i, b in enumerate(res):
                    if not self._mc_block(i):
                        continue
                    w1 = b.conv[4].module.weight
                    jobs.append((w1, buf(f'b{i}.w1k', ops.weight_image_k_elems(w1.shape[0], w1.shape[1], 3), dt), False, 1, -1, 1.0, True))
            self._prep_fwd = ops.PrepBatch([(j[0].detach(),) + tuple(j[1:]) for j in jobs], dt)
        self._prep_fwd.run()
        buf('lin_bias', lin.out_features, torch.float32).view(16, c0).copy_(lin.bias.detach().view(c0, 16).t())
        self._img_key = key

    def _prep_backward_images(self):
        """Transposed/flipped images for the input gradients, one launch (called at the start of backward)."""
        lin, res, head_bn, head_mc, head_conv = self._layers()
        dt = self.dtype
        dev = lin.weight.device
        if self._prep_bwd is None or self._prep_bwd.dtype != dt or not self._prep_bwd.valid():
            jobs = []
            self.img_t = {}
            def tbuf(name, w):
                ks = w.shape[2]
                t = torch.empty(ops.weight_image_elems(w.shape[0], w.shape[1], ks, True), dtype=dt, device=dev)
                self.img_t[name] = t
                jobs.append((w.detach(), t, True, 1, -1, 1.0))
            tbuf('head', head_conv.weight)
            for i, b in enumerate(res):
                tbuf(f'b{i}.w1', b.conv[4].module.weight)
                tbuf(f'b{i}.w2', b.conv[8].module.weight)
                tbuf(f'b{i}.ws', b.shortcut[2].module.weight)
            self._prep_bwd = ops.PrepBatch(jobs, dt)
        self._prep_bwd.run()

    # ---- mode-compacted convolutions ---------------------------------------------------------------
    def _mc_enabled(self) -> bool:
        return _MC and self.dtype == torch.bfloat16

    def _mc_block(self, i: int) -> bool:
        """Block i's convolutions run at side 8 * 2^i: from 16x16 up a 128- or 256-pixel tile lies inside one image, and
        the channel counts must suit the kernel (multiples of 8, at least 64 outputs)."""
        lin, res, head_bn, head_mc, head_conv = self._layers()
        c1 = res[i].conv[4].module
        return self._mc_enabled() and (8 << i) >= 16 and c1.out_channels >= 64 and c1.in_channels % 8 == 0 and c1.out_channels % 8 == 0

    # ---- compacted activations in the forward-only grouped pass -------------------------------------
    def _gk_enabled(self) -> bool:
        return _GK and self.dtype == torch.bfloat16

    def _gk_block(self, i: int) -> bool:
        """Block i (convolutions at side 8 * 2^i) can keep its inner activation h compacted: from 16x16 up a tile lies
        inside one image, and one 256-channel tile holds every output channel."""
        lin, res, head_bn, head_mc, head_conv = self._layers()
        c1 = res[i].conv[4].module
        return (self._gk_enabled() and (8 << i) >= 16 and 64 <= c1.out_channels <= 256 and c1.out_channels % 32 == 0
                and c1.in_channels % 8 == 0)

    def _pm_enabled(self) -> bool:
        lin, res, head_bn, head_mc, head_conv = self._layers()
        return _PM and self._gk_enabled() and res[0].mc_1.codebook.shape[0] <= _PM_MAX_MODES

    def _y_mc(self, i: int, pm: bool):
        """The MultimodalController that masks block i's OUTPUT in front of its next convolutions, when those read it compacted
        (the next block's mc_1, or -- per-mode weight sets only -- the image head's), else None."""
        lin, res, head_bn, head_mc, head_conv = self._layers()
        if not self._gk_block(i) or self._cap(res[i].mc_2) is None:
            return None
        if i + 1 < len(res):
            nxt = res[i + 1]
            ok = self._gk_block(i + 1) and self._cap(nxt.mc_2) is not None and self._cap(nxt.mc_1) is not None
            return nxt.mc_1 if ok else None
        # what conv_head.hip takes: 32x32 maps, <= 8 output channels, whole 32-channel chunks
        ok = pm and _PM_HEAD and self._pm_enabled() and (8 << i) == 32 and head_conv.out_channels <= 8 \
            and head_conv.in_channels % 32 == 0 and self._cap(head_mc) is not None and self._cap(head_mc) >= 64
        return head_mc if ok else None

    def _pm_need(self):
        """The MultimodalControllers whose compaction records the grouped pass reads (per-mode weight sets in use)."""
        lin, res, head_bn, head_mc, head_conv = self._layers()
        need = []
        for i, b in enumerate(res):
            if not self._gk_block(i) or self._cap(b.mc_2) is None:
                continue
            need.append(b.mc_2)
            ymc = self._y_mc(i, 'headm' in self.img)
            if ymc is not None:
                need.append(ymc)
        return need

    def _stacked_maps(self, need):
        """(records of all modes of all `need` controllers back to back [K * modes, stride], row offsets [K, 1] int32), or None
        when their record strides differ.  Cached per codebook versions."""
        tabs = [self._mode_maps(mc) for mc in need]
        if not tabs or any(t.shape != tabs[0].shape for t in tabs):
            return None
        key = tuple((id(mc), mc.codebook.data_ptr(), mc.codebook._version) for mc in need)
        if getattr(self, '_stack_key', None) != key:
            if tabs[0].is_cuda and torch.cuda.is_current_stream_capturing():
                raise McgenError('compacted generator pass: a codebook changed since the last eager pass; call '
                                 'GeneratorEngine.warm_caps() before capturing')
            modes = tabs[0].shape[0]
            self._stack = (torch.cat(tabs, 0).contiguous(),
                           (torch.arange(len(tabs), dtype=torch.int32, device=tabs[0].device) * modes).view(-1, 1))
            self._stack_key = key
        return self._stack

    def _gather_cmaps(self, need, wsel):
        """-> {id(mc): int16 [N, stride]} the per-image compaction records of `need` (ops.mc_cmap of the one-hot samples' codes),
        gathered from the per-mode records by label in ONE launch (+ one index add)."""
        st = self._stacked_maps(need)
        if st is None:
            return {}
        table, offs = st
        idx = (wsel.view(1, -1) + offs).view(-1)
        rows = table.index_select(0, idx).view(len(need), wsel.numel(), table.shape[1])
        return {id(mc): rows[k] for k, mc in enumerate(need)}

    def _mode_perms(self, mc):
        """int16 [modes, C]: per codebook row, its active channels in order, then the others (mcgen_conv_t.yperm /
        mcgen_prep_t.rmap).  Cached per codebook version."""
        cb = mc.codebook
        key = (cb.data_ptr(), tuple(cb.shape), cb._version)
        cache = self.__dict__.setdefault('_perm_cache', {})
        if cache.get(id(mc), (None,))[0] != key:
            perm = torch.argsort((cb == 0).to(torch.int8), dim=1, stable=True).to(torch.int16).contiguous()
            cache[id(mc)] = (key, perm)
        return cache[id(mc)][1]

    def _mode_maps(self, mc):
        """Compaction records of the codebook's ROWS (ops.mc_cmap: int16 [modes, stride]; the cidx part starts at column C):
        the map of a one-hot sample is its mode's.  Cached per codebook version."""
        cb = mc.codebook
        key = (cb.data_ptr(), tuple(cb.shape), cb._version)
        cache = self.__dict__.setdefault('_mm_cache', {})
        if cache.get(id(mc), (None,))[0] != key:
            cache[id(mc)] = (key, ops.mc_cmap(cb.detach().contiguous()))
        return cache[id(mc)][1]

    def _cap(self, mc):
        """Compacted channel pitch for activations masked by `mc`: the largest number of active channels any mode keeps,
        rounded up to 32 (None when compaction would not pay, or a code is negative).  Read from the codebook on the
        host once per codebook version (one-hot indicators: a sample's code is one codebook row, modules.py:73)."""
        cb = mc.codebook
        key = (cb.data_ptr(), tuple(cb.shape), cb._version)
        cache = self.__dict__.setdefault('_cap_cache', {})
        if cache.get(id(mc), (None,))[0] != key:
            if cb.is_cuda and torch.cuda.is_current_stream_capturing():
                raise McgenError('compacted generator pass: a codebook changed since the last eager pass; call '
                                 'GeneratorEngine.warm_caps() before capturing (the compacted pitch is read on the host)')
            cnt = int((cb != 0).sum(1).max())
            cap = (cnt + 31) // 32 * 32
            ok = cap <= cb.shape[1] * 3 // 4 and float(cb.min()) >= 0.0
            cache[id(mc)] = (key, cap if ok else None)
        return cache[id(mc)][1]

    def warm_caps(self):
        """Read the compacted pitches of the current codebooks (a host synchronisation): graph capture calls this after it
        has restored the model state, so that the captured pass finds them cached."""
        if self._gk_enabled():
            lin, res, head_bn, head_mc, head_conv = self._layers()
            for b in res:
                self._cap(b.mc_1); self._cap(b.mc_2)
                if self._pm_enabled():
                    self._mode_maps(b.mc_1); self._mode_maps(b.mc_2)
            self._cap(head_mc)
            if self._pm_enabled():
                self._mode_maps(head_mc); self._mode_perms(head_mc)
                for b in res:
                    self._mode_perms(b.mc_1); self._mode_perms(b.mc_2)
                self._stacked_maps(self._pm_need())

    # ---- forward ---------------------------------------------------------------------------------
    def groups_supported(self, n_total: int, groups: int) -> bool:
        """Can `groups` training-mode forwards of n_total / groups images each run as ONE pass?  Every launch's tile
        must lie inside one statistics group: the tiles of the Linear layer and of the 4x4 maps hold several images."""
        if groups <= 1:
            return True
        if n_total % groups:
            return False
        gn = n_total // groups
        lin, res, head_bn, head_mc, head_conv = self._layers()
        shapes = [(1, lin.out_features)]                                      # (map side, output channels) per launch
        side = 4
        for b in res:
            side *= 2
            shapes += [(side, b.conv[4].module.out_channels)] * 2
        shapes.append((side, head_conv.out_channels))
        return all(gn % ops.tile_images(n_total, sd, sd, co, self.dtype) == 0 for sd, co in shapes)

    def forward(self, z: Tensor, indicator: Tensor, train: bool, groups: int = 1, nhwc: bool = False, one_hot: bool = False,
                pair_out: Optional[Tensor] = None):
        """`groups` > 1 (training mode, forward only): z / indicator hold `groups` batches back to back, each normalised
        with its OWN BatchNorm batch statistics -- `groups` successive generator forwards on unchanged weights
        (the five discriminator updates of train_gan.py:139-158) as one pass over groups * N images.
        `nhwc`: return the images as `Nhwc` (for the discriminator engine) instead of NCHW fp32.
        `one_hot`: the caller guarantees `indicator` rows are one-hot (the trainer builds them with F.one_hot); only then
        may a grouped pass keep its activations compacted -- the compacted pitch is the largest active-channel count of a
        single codebook row (`_cap`), which a soft or multi-hot indicator could exceed.
        `pair_out` ([groups * 2 * (N / groups), H, W, 8] in the compute dtype: GANTrainer.pair_buffers): the images go
        into the second halves of its `groups` paired [real (+) generated] batches -- written there by the image head itself
        (mcgen_conv_t.y_group) when the launch qualifies, by one strided copy otherwise."""
        # (a forward that raised part-way -- an McgenError, an out-of-memory, a failed capture -- leaves its queued running-
        # statistic updates behind: they belong to that pass, not to this one)
        _pending_counters.clear(); _pending_running.clear()
        self.flat_p.ensure()
        lin, res, head_bn, head_mc, head_conv = self._layers()
        dt = self.dtype
        n = z.shape[0]
        if groups > 1 and not (train and self.groups_supported(n, groups)):
            raise RuntimeError(f'generator forward: {groups} statistics groups over {n} images is not supported')
        gn = n // groups if groups > 1 else 0          # images per statistics group (0: one batch)
        ng = n // groups                               # BatchNorm batch size
        self.refresh_images()
        st_mode = 1 if train else 0
        ctx = {'train': train, 'n': n, 'groups': groups}
        zt = ops.to_nhwc(z.detach().reshape(n, -1, 1, 1), dt)                 # [N,1,1,L]
        c0 = lin.out_features // 16
        x0, st = ops.conv_fused([Seg(zt, ksize=1)], self.img['lin'], 16 * c0, bias=self.img['lin_bias'],
                                stats_mode=st_mode)
        x = x0.view(n, 4, 4, c0)
        fold = 16                              # Linear output column p*C0+c belongs to channel c
        ctx['zt'] = zt
        blocks_ctx = []
        codes = self._codes.run_any(indicator)
        # Forward-only grouped pass: activations between the launches stay COMPACTED -- the producer stores, per image, only
        # the channels the consumer's MultimodalController keeps (ycmap), the consumer gathers the matching weight rows.
        gk = groups > 1 and one_hot and self._gk_enabled()
        # per-mode dense weight sets for the launches that read compacted activations: the mode of an image is its label
        hint = getattr(indicator, '_mcgen_onehot', None)
        pm = gk and self._pm_enabled() and hint is not None and hint[0].shape[0] * hint[1] == n
        # (walking the images mode by mode -- mcgen_conv_t.order -- measured neutral: 7.249 vs 7.256 ms/step; not used)
        wsel = None
        if pm:
            wsel = hint[2] if len(hint) > 2 and hint[2] is not None else hint[0].to(torch.int32).repeat(hint[1])
        x_cm = None                                # compaction map / pitch of the block input x when it arrives compacted
        # one-hot samples: an image's compaction record is its mode's -- one row gather for every map of the pass
        cmaps = self._gather_cmaps(self._pm_need(), wsel) if pm else {}
        caps_h = [self._cap(b.mc_2) if (gk and self._gk_block(i)) else None for i, b in enumerate(res)]
        for i, b in enumerate(res):
            s = x.shape[1]
            code1, code2 = codes[2 * i], codes[2 * i + 1]
            bn1 = _bn_forward(b.conv[0].module, st, ng * s * s, train, fold, groups)
            fold = 1
            co = b.conv[4].module.out_channels
            ci = b.conv[4].module.in_channels
            cap_h = caps_h[i]                                                            # h of this block, masked by mc_2
            nxt = res[i + 1] if i + 1 < len(res) else None
            # the block's output x stays compacted only if the NEXT block reads it compacted in both of its launches
            # the block's output x stays compacted only if its consumers read it compacted (the next block's two launches; the
            # image head through its own per-mode images)
            ymc = self._y_mc(i, pm and 'headm' in self.img) if cap_h is not None else None
            cap_y = self._cap(ymc) if ymc is not None else None
            cm_h = (cmaps[id(b.mc_2)] if id(b.mc_2) in cmaps else ops.mc_cmap(code2)) if cap_h else None
            cm_y = (cmaps[id(ymc)] if id(ymc) in cmaps else ops.mc_cmap(codes[2 * (i + 1)])) if cap_y else None
            # ---- conv_a: BN -> ReLU -> Up -> MC1 -> conv3x3 (mcgan.py:15-19)
            if x_cm is not None:
                sa, ta = ops.mc_affine(code1, x_cm[0], x_cm[1], bn1.scale, bn1.shift, group_n=gn)
                if pm and f'b{i}.w1m' in self.img:
                    # dense K loop over the compacted pitch on the image's mode's own weight image
                    seg_a = Seg(x, scale=sa, shift=ta, ups=True, relu=True, group_n=1)
                    h, st_h = ops.conv_fused([seg_a], self.img[f'b{i}.w1m'], co, bias=b.conv[4].module.bias, stats_mode=st_mode,
                                             cy=cap_h, wsel=wsel, yperm=self._mode_perms(b.mc_2))
                else:
                    seg_a = Seg(x, scale=sa, shift=ta, ups=True, relu=True, group_n=1, cmap=x_cm[0], cw=ci)
                    h, st_h = ops.conv_fused([seg_a], self.img[f'b{i}.w1g'], co, bias=b.conv[4].module.bias, stats_mode=st_mode,
                                             kmajor=2, ycmap=cm_h, cy=cap_h)
            else:
                # large maps: the K loop visits only each sample's active channels (mode-compacted kernel, K-major images)
                mc = self._mc_block(i) and f'b{i}.w1k' in self.img
                cm1 = ops.mc_cmap(code1) if mc else None
                seg_a = Seg(x, scale=bn1.scale, shift=bn1.shift, code=code1, ups=True, relu=True, group_n=gn, cmap=cm1)
                h, st_h = ops.conv_fused([seg_a], self.img[f'b{i}.w1k' if mc else f'b{i}.w1'], co, bias=b.conv[4].module.bias,
                                         stats_mode=st_mode, kmajor=int(mc), ycmap=cm_h, cy=cap_h)
            bn2 = _bn_forward(b.conv[5].module, st_h, ng * 4 * s * s, train, 1, groups)
            # ---- conv_b ++ shortcut: conv3x3(MC2(ReLU(BN(h)))) + conv1x1(MC1(Up(x))) (mcgan.py:20-30,42)
            if cm_h is not None:
                sb, tb = ops.mc_affine(code2, cm_h, cap_h, bn2.scale, bn2.shift, group_n=gn)
                use_pm = pm and f'b{i}.w2sm' in self.img
                seg_b = Seg(h, scale=sb, shift=tb, relu=True, group_n=1) if use_pm else \
                    Seg(h, scale=sb, shift=tb, relu=True, group_n=1, cmap=cm_h, cw=co)
                if x_cm is not None:
                    ss, ts = ops.mc_affine(code1, x_cm[0], x_cm[1])
                    seg_s = Seg(x, ksize=1, scale=ss, shift=ts, ups=True, group_n=1) if use_pm else \
                        Seg(x, ksize=1, scale=ss, shift=ts, ups=True, group_n=1, cmap=x_cm[0], cw=ci)
                else:
                    seg_s = Seg(x, ksize=1, code=code1, ups=True)
                if use_pm:
                    y, st = ops.conv_fused([seg_b, seg_s], self.img[f'b{i}.w2sm'].view(-1), co, bias=b.conv[8].module.bias,
                                           bias2=b.shortcut[2].module.bias, stats_mode=st_mode, cy=cap_y, wsel=wsel,
                                           yperm=self._mode_perms(ymc) if ymc is not None else None)
                else:
                    # (the gather pass of a compacted output takes one bias vector)
                    y, st = ops.conv_fused([seg_b, seg_s], self.img[f'b{i}.w2sk'], co,
                                           bias=b.conv[8].module.bias.detach() + b.shortcut[2].module.bias.detach(),
                                           stats_mode=st_mode, kmajor=2, ycmap=cm_y, cy=cap_y)
            else:
                # (conv_b ++ 1x1 shortcut stays dense in the mode-compacted form: one-tap K steps cost more than they save)
                seg_b = Seg(h, scale=bn2.scale, shift=bn2.shift, code=code2, relu=True, group_n=gn)
                seg_s = Seg(x, ksize=1, code=code1, ups=True)
                # (the shortcut's own bias rides along as bias2: the sum is formed in fp32 in front of the accumulator)
                y, st = ops.conv_fused([seg_b, seg_s], self.img[f'b{i}.w2s'], co, bias=b.conv[8].module.bias,
                                       bias2=b.shortcut[2].module.bias, stats_mode=st_mode)
            blocks_ctx.append(dict(x=x, h=h, code1=code1, code2=code2, bn1=bn1, bn2=bn2))
            x = y
            x_cm = (cm_y, cap_y) if cm_y is not None else None
        s = x.shape[1]
        bnh = _bn_forward(head_bn, st, ng * s * s, train, fold, groups)
        codeh = codes[-1]
        hw, hsel = self.img['head'], None
        if x_cm is not None:
            sh_, th_ = ops.mc_affine(codeh, x_cm[0], x_cm[1], bnh.scale, bnh.shift, group_n=gn)
            seg_h = Seg(x, scale=sh_, shift=th_, relu=True, group_n=1)
            hw, hsel = self.img['headm'], wsel
        else:
            seg_h = Seg(x, scale=bnh.scale, shift=bnh.shift, code=codeh, relu=True, group_n=gn)
        cimg = head_conv.out_channels
        if pair_out is not None:
            if tuple(pair_out.shape) != (2 * n, x.shape[1], x.shape[2], ops.pad8(cimg)) or pair_out.dtype != dt:
                raise McgenError(f'pair_out must be {(2 * n, x.shape[1], x.shape[2], ops.pad8(cimg))} {dt}')
            # what mcgen_conv_head_ok asks for (conv_head.hip): the paired layout is the image head's
            direct = (dt == torch.bfloat16 and x.shape[1] == 32 and x.shape[2] == 32 and cimg <= 8 and x.shape[-1] % 32 == 0
                      and x.shape[-1] >= 64 and n * 32 * 32 * x.shape[-1] < (1 << 31))
            if direct:
                ops.conv_fused([seg_h], hw, cimg, bias=head_conv.bias, tanh=True, out=pair_out, y_group=ng, wsel=hsel)
                out = None
            else:
                out, _ = ops.conv_fused([seg_h], hw, cimg, bias=head_conv.bias, tanh=True, wsel=hsel)
                pv = pair_out.view(groups, 2 * ng, *pair_out.shape[1:])
                pv[:, ng:].copy_(out.view(groups, ng, *out.shape[1:]))
        else:
            out, _ = ops.conv_fused([seg_h], hw, cimg, bias=head_conv.bias, tanh=True, wsel=hsel)
        ctx.update(blocks=blocks_ctx, y=x, bnh=bnh, codeh=codeh, out=out)
        _flush_counters()
        if pair_out is not None:
            return None, ctx
        return (Nhwc(out, head_conv.out_channels) if nhwc else ops.to_nchw(out, head_conv.out_channels)), ctx

    # ---- backward ----------------------------------------------------------------------------------
    def backward(self, ctx, dimg: Tensor, gflat: Tensor, accumulate: bool = False):
        """dimg: [N, C, H, W] fp32.  Writes (or adds) every generator parameter's gradient into
        `gflat`, a flat fp32 buffer laid out like ``flat_p``."""
        for _ in self.backward_iter(ctx, dimg, gflat, accumulate, split=False):
            pass

    def bucket_cut(self) -> int:
        """First residual block of the LATE gradient bucket (see DiscriminatorEngine.bucket_cut): the backward pass
        finishes the head and blocks cut .. last first."""
        lin, res, head_bn, head_mc, head_conv = self._layers()
        return max(1, len(res) // 2) if len(res) > 1 else 0

    def backward_iter(self, ctx, dimg: Tensor, gflat: Tensor, accumulate: bool = False, split: bool = True):
        """As `backward`, as a generator (`split` False: one bucket, handed out at the end): yields (lo, hi) each time gflat[lo:hi] is final -- the head and the late
        blocks mid-pass, the early blocks and the Linear layer at the end."""
        if not ctx['train']:
            raise RuntimeError('generator backward needs a training-mode forward (batch statistics)')
        if ctx.get('groups', 1) != 1:
            raise RuntimeError('a grouped generator pass is forward-only (its images feed discriminator updates detached)')
        lin, res, head_bn, head_mc, head_conv = self._layers()
        dt = self.dtype
        n = ctx['n']
        acc = accumulate
        out = ctx['out']
        self._prep_backward_images()
        dout = dimg.t if isinstance(dimg, Nhwc) else ops.to_nhwc(dimg.contiguous(), dt, out.shape[-1])
        dtn = ops.tanh_bwd(dout, out)
        y, bnh, codeh = ctx['y'], ctx['bnh'], ctx['codeh']
        c_img, c = head_conv.out_channels, head_conv.in_channels
        cut = self.bucket_cut()
        off_cut = self.flat_p.offset_of(next(p for p in res[cut].parameters())) if res else 0
        G = lambda p: self.flat_p.view_of(gflat, p)                           # noqa: E731
        red = ops.deferred_reduces()           # the split-K reductions of one bucket: one launch
        red.__enter__()
        try:
            # head conv: bias / weight grads, then the input gradient through MC, ReLU and BN
            seg_h = Seg(y, scale=bnh.scale, shift=bnh.shift, code=codeh, relu=True)
            ops.wgrad(seg_h, dtn, c_img, c, G(head_conv.weight), accumulate=acc, bias_grad=G(head_conv.bias))
            wt = self.img_t['head']
            dz, part = ops.conv_fused([Seg(dtn)], wt, c, ocode=codeh, gate_x=y, gscale=bnh.scale, gshift=bnh.shift,
                                      gmean=bnh.mean, grstd=bnh.rstd, stats_mode=2)
            dy = ops.bn_backward(part, dz, y, bnh.count, bnh.scale, bnh.mean, bnh.rstd,
                                 G(head_bn.weight), G(head_bn.bias), accumulate=acc)
            for i in reversed(range(len(res))):
                b, bc = res[i], ctx['blocks'][i]
                x, h, code1, code2, bn1, bn2 = bc['x'], bc['h'], bc['code1'], bc['code2'], bc['bn1'], bc['bn2']
                conv1, conv2, convs = b.conv[4].module, b.conv[8].module, b.shortcut[2].module
                bnm1, bnm2 = b.conv[0].module, b.conv[5].module
                ci, co = conv1.in_channels, conv1.out_channels
                # second conv and the 1x1 shortcut both see dy
                seg_b = Seg(h, scale=bn2.scale, shift=bn2.shift, code=code2, relu=True)
                seg_s = Seg(x, ksize=1, code=code1, ups=True)
                ops.wgrad(seg_b, dy, co, co, G(conv2.weight), accumulate=acc, bias_grad=G(conv2.bias), bias_grad2=G(convs.bias))
                # the shortcut conv1x1(Up(x)) commutes with the upsample: in bf16 its weight gradient and input gradient are
                # taken at x's resolution from the 2x2-pooled dy (a quarter of the FLOPs; fp32 keeps the literal form)
                lowres = _LOWRES_SC_BWD and dt == torch.bfloat16 and dy.shape[1] >= 16
                dy_lo = ops.pool2_sum(dy) if lowres else None
                if lowres:
                    ops.wgrad(Seg(x, ksize=1, code=code1), dy_lo, co, ci, G(convs.weight), accumulate=acc)
                else:
                    ops.wgrad(seg_s, dy, co, ci, G(convs.weight), accumulate=acc)
                w2t = self.img_t[f'b{i}.w2']
                dz2, part2 = ops.conv_fused([Seg(dy)], w2t, co, ocode=code2, gate_x=h, gscale=bn2.scale, gshift=bn2.shift,
                                            gmean=bn2.mean, grstd=bn2.rstd, stats_mode=2)
                dh = ops.bn_backward(part2, dz2, h, bn2.count, bn2.scale, bn2.mean, bn2.rstd,
                                     G(bnm2.weight), G(bnm2.bias), accumulate=acc)
                # first conv: gradient goes through MC, the nearest-upsample adjoint (2x2 sum), ReLU, BN
                seg_a = Seg(x, scale=bn1.scale, shift=bn1.shift, code=code1, ups=True, relu=True)
                ops.wgrad(seg_a, dh, co, ci, G(conv1.weight), accumulate=acc, bias_grad=G(conv1.bias))
                wst = self.img_t[f'b{i}.ws']
                if lowres:
                    dx_sc, _ = ops.conv_fused([Seg(dy_lo, ksize=1)], wst, ci, ocode=code1)
                else:
                    dx_sc, _ = ops.conv_fused([Seg(dy, ksize=1)], wst, ci, pool=True, alpha=1.0, ocode=code1)
                w1t = self.img_t[f'b{i}.w1']
                dz1, part1 = ops.conv_fused([Seg(dh)], w1t, ci, pool=True, alpha=1.0, ocode=code1, gate_x=x,
                                            gscale=bn1.scale, gshift=bn1.shift, gmean=bn1.mean, grstd=bn1.rstd, stats_mode=2)
                dy = ops.bn_backward(part1, dz1, x, bn1.count, bn1.scale, bn1.mean, bn1.rstd,
                                     G(bnm1.weight), G(bnm1.bias), add=dx_sc, accumulate=acc)
                if i == cut and cut > 0 and _BUCKETS and split:
                    red.__exit__(None, None, None)           # head + blocks cut .. last are final
                    if _BUCKETS and split:
                        yield (off_cut, gflat.numel(), False)
                    red = ops.deferred_reduces()
                    red.__enter__()
            # linear layer: dy is [N,4,4,C0] == [N,1,1,16*C0] in the permuted row order
            c0 = lin.out_features // 16
            dflat = dy.view(n, 1, 1, 16 * c0)
            ops.wgrad(Seg(ctx['zt'], ksize=1), dflat, 16 * c0, lin.in_features, G(lin.weight), row_perm=16, accumulate=acc,
                      bias_grad=G(lin.bias))
        except BaseException as e:
            red.__exit__(type(e), e, None)
            raise
        red.__exit__(None, None, None)
        yield (0, off_cut if (cut > 0 and _BUCKETS and split) else gflat.numel(), True)


# ============================================================================================= #
#  Discriminator
# ============================================================================================= #
class _SNConv:
    """One spectrally normalised Conv2d/Linear: module + index into the per-forward sigma vector."""

    def __init__(self, module: nn.Module, idx: int):
        self.m = module
        self.idx = idx
        w = module.weight_orig
        self.cout, self.cin = w.shape[0], w.shape[1]
        self.ks = w.shape[2] if w.dim() == 4 else 1


class DiscriminatorEngine:
    def __init__(self, dis, dtype: torch.dtype = torch.float32):
        self.dis = dis
        self.dtype = dtype
        blocks = list(dis.blocks.children())
        self.res = [b for b in blocks if hasattr(b, 'mc_1')]
        t = len(self.res)
        self.tail_mc, self.tail_lin = blocks[t + 1], blocks[t + 3].module
        self.sn: List[_SNConv] = []
        for m in dis.modules():
            if hasattr(m, 'weight_orig'):
                self.sn.append(_SNConv(m, len(self.sn)))
        self.sn_of = {s.m: s for s in self.sn}
        params = list(dis.parameters())
        sn_w = {id(s.m.weight_orig) for s in self.sn}
        self.plain = [p for p in params if id(p) not in sn_w]
        self.flat_p = FlatState(params)
        uv = []
        for s in self.sn:
            uv += [s.m.weight_u, s.m.weight_v]
        self.flat_uv = FlatState(uv)
        self._layers_dev = None          # SN layers first, then plain parameters (rows == 0)
        self._layers_key = None
        mcs = [self.res[0].mc_1] + [m for b in self.res[1:] for m in (b.mc_1, b.mc_2)] + [self.tail_mc]
        self._codes = ops.CodeBatch(mcs)
        self._prep_fwd = self._prep_bwd = None
        self.img: Dict[str, Tensor] = {}

    def _ensure_flat(self):
        fp, fuv = self.flat_p.ensure(), self.flat_uv.ensure()
        key = (fp.data_ptr(), fuv.data_ptr())
        if key != self._layers_key:
            rows = []
            for s in self.sn:
                w = s.m.weight_orig
                rows.append((self.flat_p.offset_of(w), self.flat_uv.offset_of(s.m.weight_u),
                             self.flat_uv.offset_of(s.m.weight_v), w.shape[0], w[0].numel()))
            for p in self.plain:
                rows.append((self.flat_p.offset_of(p), 0, 0, 0, p.numel()))
            self._layers_dev = ops.sn_layers_tensor(rows, fp.device)
            self._layers_key = key
        return fp, fuv

    def _build_preps(self):
        """Job lists of the forward and backward weight images (persistent buffers, sigma by index)."""
        dt = self.dtype
        dev = self.flat_p.flat.device
        self.img = {}

        def img(name, s: _SNConv, transpose, parts=None):
            n = ops.weight_image_elems(s.cout, s.cin, s.ks, transpose)
            t = torch.empty(n, dtype=dt, device=dev)
            self.img[name] = t
            return t

        def cat(name, a: _SNConv, b: _SNConv, transpose):
            na = ops.weight_image_elems(a.cout, a.cin, a.ks, transpose)
            nb = ops.weight_image_elems(b.cout, b.cin, b.ks, transpose)
            t = torch.empty(na + nb, dtype=dt, device=dev)
            self.img[name] = t
            return t[:na], t[na:]

        fwd, bwd = [], []
        W = lambda s: s.m.weight_orig.detach()                                # noqa: E731
        b0 = self.res[0]
        c1m, c2m, scm = (self.sn_of[b0.conv[0].module], self.sn_of[b0.conv[3].module], self.sn_of[b0.shortcut[0].module])
        fwd.append((W(c1m), img('0.c1', c1m, False), False, 1, c1m.idx, 1.0))
        if _D0_FUSE:
            ta, tb = cat('0.c2s', c2m, scm, False)                           # conv2 + the 1x1 shortcut: one launch, two K segments
            fwd += [(W(c2m), ta, False, 1, c2m.idx, 1.0), (W(scm), tb, False, 1, scm.idx, 1.0)]
        else:
            fwd += [(W(scm), img('0.sc', scm, False), False, 1, scm.idx, 1.0),
                    (W(c2m), img('0.c2', c2m, False), False, 1, c2m.idx, 1.0)]
        bwd.append((W(c2m), img('0.c2t', c2m, True), True, 1, c2m.idx, 0.25))
        ta, tb = cat('0.dimg', c1m, scm, True)
        bwd += [(W(c1m), ta, True, 1, c1m.idx, 1.0), (W(scm), tb, True, 1, scm.idx, 0.25)]
        for i, b in enumerate(self.res[1:], start=1):
            c1m, c2m = self.sn_of[b.conv[2].module], self.sn_of[b.conv[5].module]
            has_sc, pooled = len(b.shortcut) > 0, len(b.conv) == 7
            a = 0.25 if pooled else 1.0
            fwd.append((W(c1m), img(f'{i}.c1', c1m, False), False, 1, c1m.idx, 1.0))
            if has_sc:
                scm = self.sn_of[b.shortcut[1].module]
                ta, tb = cat(f'{i}.c2s', c2m, scm, False)
                fwd += [(W(c2m), ta, False, 1, c2m.idx, 1.0), (W(scm), tb, False, 1, scm.idx, 1.0)]
                bwd.append((W(scm), img(f'{i}.sct', scm, True), True, 1, scm.idx, a))
            else:
                fwd.append((W(c2m), img(f'{i}.c2', c2m, False), False, 1, c2m.idx, 1.0))
            bwd += [(W(c2m), img(f'{i}.c2t', c2m, True), True, 1, c2m.idx, a),
                    (W(c1m), img(f'{i}.c1t', c1m, True), True, 1, c1m.idx, 1.0)]
        self._prep_fwd, self._prep_bwd = ops.PrepBatch(fwd, dt), ops.PrepBatch(bwd, dt)
        # training passes build both sets in ONE launch at the start of the forward (same sigma): the backward pass that
        # follows finds its transposed images ready (`_bwd_sigma` remembers whose they are)
        self._prep_all = ops.PrepBatch(fwd + bwd, dt)
        self._bwd_sigma = None

    def _ensure_preps(self):
        if self._prep_fwd is None or self._prep_fwd.dtype != self.dtype or not self._prep_fwd.valid():
            self._build_preps()

    # ---- forward ---------------------------------------------------------------------------------
    def _power_iters(self, rounds: int, train: bool):
        """`rounds` successive spectral-norm power iterations of every layer (one launch): [(sigma, u/v snapshot)] per
        round -- torch's hook clones u, v for the backward pass, the snapshot is that clone."""
        fp, fuv = self._ensure_flat()
        nsn = len(self.sn)
        if train and _SN_ROUNDS and not _SN_FUSED and rounds >= int(_flag('MCGEN_SN_ROUNDS_MIN', '1')):
            sigma, snap = ops.sn_power_iter_rounds(fp, fuv, self._layers_dev, nsn, rounds,
                                                   max(s.cout for s in self.sn), max(s.cin * s.ks * s.ks for s in self.sn))
            for s in self.sn:
                _bump(s.m.weight_u); _bump(s.m.weight_v)
            self._sn_ratio = sigma[rounds] if rounds >= 2 else None       # sigma[rounds - 2] / sigma[rounds - 1], from the same launch
            return [(sigma[r], snap[r]) for r in range(rounds)]
        self._sn_ratio = None
        if not _SN_FUSED:                                   # the four-kernel form, one round per call
            out = []
            for _ in range(rounds):
                sg = torch.empty(nsn, dtype=torch.float32, device=fp.device)
                # (training mode: the kernels that produce the new u, v also write this forward's copy of them)
                snap = torch.empty_like(fuv) if (train and _SN_SNAP) else None
                ops.sn_power_iter(fp, fuv, self._layers_dev, nsn, train, sg,
                                  max(s.cout for s in self.sn), max(s.cin * s.ks * s.ks for s in self.sn), snap=snap)
                out.append((sg, snap if snap is not None else fuv.clone()))
            if train:
                for s in self.sn:
                    _bump(s.m.weight_u); _bump(s.m.weight_v)
            return out
        sigma, snap = ops.sn_power_iter_fused(fp, fuv, self._layers_dev, nsn, rounds, train,
                                              max(s.cout for s in self.sn), max(s.cin * s.ks * s.ks for s in self.sn))
        if train:
            for s in self.sn:
                _bump(s.m.weight_u); _bump(s.m.weight_v)
        return [(sigma[r], snap[r]) for r in range(rounds)]

    def _power_iter(self, train: bool):
        return self._power_iters(1, train)[0]

    def forward(self, x_nchw: Tensor, indicator: Tensor, train: bool, tail_loss: Optional[str] = None):
        """`tail_loss` 'g' (the generator update, train_gan.py:172): the tail's launch also forms d(hinge_g)/d(logit) and the
        tail's input gradient (ops.dtail_hinge_fused); ctx['tail'] = (dlogit, d tail input) is what `backward_iter` then
        starts from -- pass ctx['tail'][0] as its dlogit."""
        sigma, uv = self._power_iter(train)
        ctx = {'n': x_nchw.shape[0], 'sigma': sigma, 'uv': uv, 'blocks': [], 'pair': None, 'train': train}
        lab = self._codes.labels_of(indicator) if _PREP_CODES else None
        if lab is not None:
            # the codes ride in the weight-image launch (both wait for the power iteration only)
            ctx['codes_job'] = (self._codes, lab[0], lab[1], None, 0, lambda outs: ctx.__setitem__('codes', outs))
        else:
            ctx['codes'] = self._codes.run_any(indicator)
        return self._forward_body(x_nchw, ctx, lambda mc_i, sn_idx: ctx['codes'][mc_i] if mc_i is not None else None, tail_loss)

    def pair_codes(self, ind2: Tensor):
        """The UNSCALED MultimodalController codes of a paired pass ([2N, C] per MC: what the weight gradients multiply their
        conv inputs with).  They depend on the labels alone, so the d_iters updates of one iteration can share them."""
        return self._codes.run_any(ind2)

    def forward_pair(self, real_nchw: Tensor, fake_nchw: Tensor, indicator: Tensor, ind2: Optional[Tensor] = None,
                     x2: Optional[Nhwc] = None, codes=None, tail_loss: Optional[str] = None):
        """D(real) and D(fake) of one discriminator update (train_gan.py:144-150) as ONE pass over the 2N batch.
        The two forwards of the reference differ only in the spectral-norm state (each runs its own power iteration,
        which depends on the weights alone): conv(x; W / sigma_2) = (sigma_1 / sigma_2) * conv(x; W / sigma_1), so the
        fake half runs on the first pass's weight images with sigma_1 / sigma_2 folded into its per-sample
        MultimodalController codes (the prologue multiply sits after the ReLU, i.e. directly on the conv input).
        `x2` (optional, then real / fake are ignored): the 2N batch real (+) fake already in the engine's layout.
        `codes` (optional): `pair_codes(ind2)`, computed once for several updates on the same labels.
        `tail_loss` 'd_pair': the tail's launch also forms d(hinge_d)/d(logit) of the 2N batch and the tail's input gradient
        (ctx['tail']); `backward_iter` writes the loss value into ctx['loss'] (train_gan.py:154)."""
        n = x2.shape[0] // 2 if x2 is not None else real_nchw.shape[0]
        (sigma1, uv1), (sigma2, uv2) = self._power_iters(2, True)
        ratio = self._sn_ratio if getattr(self, '_sn_ratio', None) is not None else sigma1 / sigma2
        if ind2 is None:
            ind2 = torch.cat([indicator, indicator])
        uses = self._code_uses()
        if getattr(self, '_codes_pair', None) is None:
            # one job per convolution input: the MC's codebook (an all-ones table for the image inputs, which have no
            # MC) and the SN layer whose sigma ratio scales the fake half
            class _Ones:                                        # stands in for an MC on the raw image (8 padded channels)
                pass
            ones = _Ones()
            ones.codebook = torch.ones(indicator.shape[1], 8, device=indicator.device)
            self._ones_mc = ones
            self._codes_pair = ops.CodeBatch([self._codes.mcs[u[0]] if u[0] is not None else ones for u in uses],
                                             [u[1] for u in uses])
        scaled = {}
        if codes is None:
            codes = self._codes.run_any(ind2)                      # unscaled [2N, C] codes: the weight gradients' conv inputs
        x = x2 if x2 is not None else torch.cat([real_nchw.detach(), fake_nchw.detach()])
        ctx = {'n': 2 * n, 'sigma': sigma1, 'uv': uv1, 'blocks': [], 'codes': codes,
               'pair': {'n': n, 'sigma2': sigma2, 'uv2': uv2, 'ratio': ratio}}
        lab = self._codes_pair.labels_of(ind2) if _PREP_CODES else None
        if lab is not None:
            # all scaled codes of the pass ride in the weight-image launch (both wait for the power iteration only)
            ctx['codes_job'] = (self._codes_pair, lab[0], lab[1], ratio, n, lambda outs: scaled.update(zip(uses, outs)))
        else:
            scaled.update(zip(uses, self._codes_pair.run_any(ind2, ratio, n)))     # all scaled codes of the pass: one launch
        return self._forward_body(x, ctx, lambda mc_i, sn_idx: scaled[(mc_i, sn_idx)], tail_loss)

    def _code_uses(self):
        """(MC index or None for the image, SN layer index) of every convolution input of the network."""
        if getattr(self, '_uses', None) is None:
            b0 = self.res[0]
            c1m, c2m, scm = (self.sn_of[b0.conv[0].module], self.sn_of[b0.conv[3].module], self.sn_of[b0.shortcut[0].module])
            uses = [(None, c1m.idx), (None, scm.idx), (0, c2m.idx)]
            for i, b in enumerate(self.res[1:], start=1):
                c1m, c2m = self.sn_of[b.conv[2].module], self.sn_of[b.conv[5].module]
                uses += [(2 * i - 1, c1m.idx), (2 * i, c2m.idx)]
                if len(b.shortcut) > 0:
                    uses.append((2 * i - 1, self.sn_of[b.shortcut[1].module].idx))
            uses.append((len(self._codes.mcs) - 1, self.sn_of[self.tail_lin].idx))
            self._uses = uses
            self._use_idx = {}
        return self._uses

    def _forward_body(self, x_nchw: Tensor, ctx, code_of, tail_loss: Optional[str] = None):
        dt = self.dtype
        n = x_nchw.shape[0]
        sigma = ctx['sigma']
        self._ensure_preps()
        prep = self._prep_all if ctx.get('train', True) else self._prep_fwd
        job = ctx.pop('codes_job', None)
        if job is not None:
            cb, label, reps, scale, n_half, sink = job
            sink(ops.prep_and_codes(prep, sigma, cb, label, reps, scale, n_half))
        else:
            prep.run(sigma)                       # every W / sigma image of this pass, forward and transposed, in one launch
        if ctx.get('train', True):
            self._bwd_sigma = sigma
        I = self.img
        img = x_nchw.t if isinstance(x_nchw, Nhwc) else ops.to_nhwc(x_nchw.detach().contiguous(), dt)
        if img.dtype != dt:
            raise RuntimeError(f'Nhwc input is {img.dtype}, the engine computes in {dt}')
        ctx['img'], ctx['nhwc'] = img, isinstance(x_nchw, Nhwc)
        # --- FirstDisResBlock (mcgan.py:72-93)
        b0 = self.res[0]
        c1m, c2m, scm = (self.sn_of[b0.conv[0].module], self.sn_of[b0.conv[3].module], self.sn_of[b0.shortcut[0].module])
        co = c1m.cout
        c1, _ = ops.conv_fused([Seg(img, code=code_of(None, c1m.idx))], I['0.c1'], co, bias=c1m.m.bias)
        # conv2 and the shortcut's 1x1 (mcgan.py:83-86: conv then AvgPool, like conv2) share the pooled epilogue: the shortcut is
        # a second K segment of the same launch, as in the later blocks (its own launch + the residual read: 21 us of 80)
        code = code_of(0, c2m.idx)
        if _D0_FUSE:
            y, _ = ops.conv_fused([Seg(c1, code=code, relu=True), Seg(img, ksize=1, code=code_of(None, scm.idx))], I['0.c2s'], co,
                                  bias=c2m.m.bias, bias2=scm.m.bias, pool=True, alpha=0.25)
        else:
            sc, _ = ops.conv_fused([Seg(img, ksize=1, code=code_of(None, scm.idx))], I['0.sc'], co, bias=scm.m.bias, pool=True, alpha=0.25)
            y, _ = ops.conv_fused([Seg(c1, code=code, relu=True)], I['0.c2'], co, bias=c2m.m.bias, pool=True, alpha=0.25, res=sc)
        ctx['blocks'].append({'c1': c1, 'code': code})
        x = y
        # --- DisResBlocks (mcgan.py:96-138)
        for i, b in enumerate(self.res[1:], start=1):
            c1m, c2m = self.sn_of[b.conv[2].module], self.sn_of[b.conv[5].module]
            has_sc = len(b.shortcut) > 0
            pooled = len(b.conv) == 7
            code1, code2 = code_of(2 * i - 1, c1m.idx), code_of(2 * i, c2m.idx)
            code1s = None
            c1, _ = ops.conv_fused([Seg(x, code=code1, relu=True)], I[f'{i}.c1'], c1m.cout, bias=c1m.m.bias)
            if has_sc:
                scm = self.sn_of[b.shortcut[1].module]
                code1s = code_of(2 * i - 1, scm.idx)
                y, _ = ops.conv_fused([Seg(c1, code=code2, relu=True), Seg(x, ksize=1, code=code1s)], I[f'{i}.c2s'], c2m.cout,
                                      bias=c2m.m.bias, bias2=scm.m.bias, pool=pooled, alpha=0.25 if pooled else 1.0)
            else:
                y, _ = ops.conv_fused([Seg(c1, code=code2, relu=True)], I[f'{i}.c2'], c2m.cout,
                                      bias=c2m.m.bias, res=x)
            ctx['blocks'].append({'x': x, 'c1': c1, 'code1': code1, 'code1s': code1s, 'code2': code2, 'pooled': pooled, 'has_sc': has_sc})
            x = y
        # --- tail: ReLU -> MC -> global sum pool -> SN linear (mcgan.py:158-165)
        tl = self.sn_of[self.tail_lin]
        codet = code_of(len(self._codes.mcs) - 1, tl.idx)
        if tail_loss is not None and ops.dtail_hinge_ok(x):
            if (tail_loss == 'd_pair') != (ctx['pair'] is not None):
                raise McgenError("tail_loss: 'd_pair' goes with forward_pair, 'g' with forward")
            logit, pooled_feat, dlogit, dxt = ops.dtail_hinge_fused(x, codet, self.tail_lin.weight_orig.detach().view(-1), self.tail_lin.bias,
                                                                    sigma[tl.idx:tl.idx + 1], tail_loss)
            ctx['tail'] = (dlogit, dxt)
            if tail_loss == 'd_pair':
                ctx['loss'] = torch.empty(1, dtype=torch.float32, device=x.device)
                ctx['logit'] = logit
        else:
            logit, pooled_feat = ops.dtail_fwd(x, codet, self.tail_lin.weight_orig.detach().view(-1), self.tail_lin.bias,
                                               sigma[tl.idx:tl.idx + 1])
        ctx.update(xt=x, codet=codet, pooled=pooled_feat)
        return logit.view(n, 1), ctx

    # ---- backward ----------------------------------------------------------------------------------
    def backward(self, ctx, dlogit: Tensor, gflat: Optional[Tensor], accumulate: bool, need_input_grad: bool):
        """Runs `backward_iter` to its end; returns d(input image) as NCHW fp32, or None."""
        it = self.backward_iter(ctx, dlogit, gflat, accumulate, need_input_grad, split=False)
        while True:
            try:
                next(it)
            except StopIteration as stop:
                return stop.value

    def bucket_cut(self) -> int:
        """First residual block of the LATE gradient bucket: the backward pass finishes blocks cut .. last and the tail
        first, so gflat[offset(block cut) :] is final while blocks cut-1 .. 0 are still running -- a data-parallel run
        starts that bucket's all-reduce there (GANTrainer), under the rest of the backward.
        (0: a discriminator with a single residual block has nothing to split -- one bucket, as the generator's rule.)"""
        return max(1, len(self.res) // 2) if len(self.res) > 1 else 0

    def _bucket_tables(self):
        """Spectral-norm fix-up tables of the two buckets: SN layers with index >= the cut block's first layer + the plain
        parameters stored behind the cut offset, and the rest."""
        fp, _ = self._ensure_flat()
        key = (self._layers_key, self.bucket_cut())
        if getattr(self, '_bk_key', None) != key:
            cut_block = self.res[self.bucket_cut()]
            first = next(p for p in cut_block.parameters())
            off_cut = self.flat_p.offset_of(first)
            sn_hi = [sn for sn in self.sn if self.flat_p.offset_of(sn.m.weight_orig) >= off_cut]
            i_cut = min(sn.idx for sn in sn_hi)
            assert [sn.idx for sn in sn_hi] == list(range(i_cut, len(self.sn))), 'SN layers of the late bucket are not contiguous'

            def table(sns, plains):
                rows = [(self.flat_p.offset_of(sn.m.weight_orig), self.flat_uv.offset_of(sn.m.weight_u),
                         self.flat_uv.offset_of(sn.m.weight_v), sn.m.weight_orig.shape[0], sn.m.weight_orig[0].numel()) for sn in sns]
                rows += [(self.flat_p.offset_of(p), 0, 0, 0, p.numel()) for p in plains]
                return (ops.sn_layers_tensor(rows, fp.device) if rows else None), len(rows)
            hi = table(sn_hi, [p for p in self.plain if self.flat_p.offset_of(p) >= off_cut])
            lo = table([sn for sn in self.sn if sn.idx < i_cut], [p for p in self.plain if self.flat_p.offset_of(p) < off_cut])
            # (a single-rank update has nothing to overlap with: one table for the fused fix + Adam launch)
            every = table(sorted(self.sn, key=lambda sn: sn.idx), list(self.plain))
            self._bk = {'off_cut': off_cut, 'i_cut': i_cut, 'hi': hi, 'lo': lo, 'all': every}
            self._bk_key = key
        return self._bk

    def backward_iter(self, ctx, dlogit: Tensor, gflat: Optional[Tensor], accumulate: bool, need_input_grad: bool,
                      split: bool = True, defer_fix: bool = False):
        """dlogit [N] fp32.  (`split` False: one gradient bucket, handed out at the end.)  Parameter gradients (w.r.t. weight_orig and the biases) are written or
        added into `gflat` (flat, laid out like ``flat_p``; None skips them, e.g. in the generator
        step).  A generator: yields (lo, hi) each time gflat[lo:hi] is final -- the late bucket (tail + blocks
        bucket_cut() ..) mid-pass, the early bucket at the end -- and returns d(input image) as NCHW fp32, or None.
        For a `forward_pair` context dlogit is [2N] (real half, fake half): input gradients run once over the 2N
        batch (sigma_1 / sigma_2 rides in the output codes), weight gradients are taken per half, because each half
        owns its spectral-norm state (u, v, sigma) in the fix-up from d/d(W/sigma) to d/d(weight_orig)."""
        dt = self.dtype
        fp, _ = self._ensure_flat()
        sigma, uv = ctx['sigma'], ctx['uv']
        pair = ctx['pair']
        want_w = gflat is not None
        # unscaled codes of every MC (weight gradients see the true conv input)
        codes_full = ctx['codes']

        def U(i, sl):                                  # code rows matching the activation rows x[sl]
            return codes_full[i][sl]
        # raw gradients (w.r.t. the NORMALISED weights, and the biases) land here first: one buffer per pass
        if pair is None:
            passes = [(slice(None), torch.empty_like(fp) if want_w else None)]
        else:
            passes = [(slice(0, pair['n']), torch.empty_like(fp)), (slice(pair['n'], 2 * pair['n']), torch.empty_like(fp))]

        def wgrad_all(seg_fn, dy, cout, cin, param, bias=None, bias2=None, **kw):
            def dests(gtmp):
                T = lambda p: self.flat_p.view_of(gtmp, p)                         # noqa: E731
                return T(param), (T(bias) if bias is not None else None), (T(bias2) if bias2 is not None else None)
            hq = dy.shape[1] * dy.shape[2] * (4 if kw.get('dy_ups') else 1)          # pixels per image at the conv resolution
            if pair is not None and _PAIR_WGRAD and (pair['n'] * hq) % 128 == 0:
                # one launch over the 2N batch, one slab set per half; the codes of the two halves are equal
                seg = seg_fn(slice(None))
                (g1, b1, b12), second = dests(passes[0][1]), dests(passes[1][1])
                ops.wgrad(seg, dy, cout, cin, g1, bias_grad=b1, bias_grad2=b12, second=second, **kw)
                return
            for sl, gtmp in passes:
                g1, b1, b12 = dests(gtmp)
                ops.wgrad(seg_fn(sl), dy[sl], cout, cin, g1, bias_grad=b1, bias_grad2=b12, **kw)

        def fix(tab, sg_off):
            """d/d(W/sigma) -> d/d(weight_orig) with the u, v, sigma THIS forward used (biases are moved as is), for the
            layers of one bucket table."""
            table, nl = tab
            if nl == 0:
                return
            if pair is not None:
                ops.sn_grad_fix_pair(passes[0][1], passes[1][1], gflat, fp, uv, pair['uv2'], table, nl, sigma[sg_off:],
                                     pair['sigma2'][sg_off:], accumulate=accumulate)
            else:
                ops.sn_grad_fix(passes[0][1], gflat, fp, uv, table, nl, sigma[sg_off:], accumulate=accumulate)

        bk = self._bucket_tables() if want_w else None
        cut = self.bucket_cut()
        split = split and cut > 0                      # (a single residual block: one bucket)
        self._ensure_preps()
        if self._bwd_sigma is not sigma:          # (another forward rebuilt the images since this pass's own)
            self._prep_bwd.run(sigma)             # transposed W / sigma images of THIS pass's sigma
            self._bwd_sigma = sigma
        I = self.img
        tl = self.sn_of[self.tail_lin]
        sg_t = sigma[tl.idx:tl.idx + 1]
        wl = self.tail_lin.weight_orig
        red = ops.deferred_reduces()           # the split-K reductions of one bucket: one launch, before its SN fix-up
        red.__enter__()
        try:
            # a forward that ran the fused tail already holds the tail's input gradient for ITS dlogit (ctx['tail'])
            tail = ctx.get('tail')
            pre_dy = tail[1] if (tail is not None and dlogit is tail[0]) else None
            if pair is None:
                gt = passes[0][1]
                if pre_dy is not None and not want_w:
                    dy = pre_dy
                else:
                    dy = ops.dtail_bwd(dlogit, ctx['xt'], ctx['codet'], wl.detach().view(-1), sg_t, ctx['pooled'],
                                       self.flat_p.view_of(gt, wl).view(-1) if want_w else None,
                                       self.flat_p.view_of(gt, self.tail_lin.bias) if want_w else None)
            else:
                dy = pre_dy if pre_dy is not None else \
                    ops.dtail_bwd(dlogit, ctx['xt'], ctx['codet'], wl.detach().view(-1), sg_t, ctx['pooled'], None, None)
                # tail weight / bias gradients per half; the fake half's pooled features carry sigma_1 / sigma_2
                (_, g_a), (_, g_b) = passes
                with_loss = pre_dy is not None and 'loss' in ctx
                ops.dtail_pair_wgrad(dlogit, ctx['pooled'], pair['ratio'][tl.idx:tl.idx + 1],
                                     self.flat_p.view_of(g_a, wl).view(-1), self.flat_p.view_of(g_a, self.tail_lin.bias).view(-1),
                                     self.flat_p.view_of(g_b, wl).view(-1), self.flat_p.view_of(g_b, self.tail_lin.bias).view(-1),
                                     logit=ctx['logit'] if with_loss else None, loss=ctx['loss'] if with_loss else None)
            for bi in reversed(range(1, len(self.res))):
                b, bc = self.res[bi], ctx['blocks'][bi]
                x, c1, code1, code2, pooled, has_sc = bc['x'], bc['c1'], bc['code1'], bc['code2'], bc['pooled'], bc['has_sc']
                code1s = bc['code1s']
                i1, i2 = 2 * bi - 1, 2 * bi                        # MC indices of the block's unscaled codes
                c1m, c2m = self.sn_of[b.conv[2].module], self.sn_of[b.conv[5].module]
                scm = self.sn_of[b.shortcut[1].module] if has_sc else None
                a = 0.25 if pooled else 1.0
                if want_w:
                    wgrad_all(lambda sl: Seg(c1[sl], code=U(i2, sl), relu=True), dy, c2m.cout, c2m.cin, c2m.m.weight_orig,
                              c2m.m.bias, scm.m.bias if has_sc else None, dy_ups=pooled, alpha=a)
                    if has_sc:
                        wgrad_all(lambda sl: Seg(x[sl], ksize=1, code=U(i1, sl)), dy, scm.cout, scm.cin, scm.m.weight_orig,
                                  dy_ups=pooled, alpha=a)
                dc1, _ = ops.conv_fused([Seg(dy, ups=pooled)], I[f'{bi}.c2t'], c2m.cin, ocode=code2, gate_x=c1)
                if want_w:
                    wgrad_all(lambda sl: Seg(x[sl], code=U(i1, sl), relu=True), dc1, c1m.cout, c1m.cin, c1m.m.weight_orig, c1m.m.bias)
                if has_sc:
                    res, _ = ops.conv_fused([Seg(dy, ksize=1, ups=pooled)], I[f'{bi}.sct'], scm.cin, ocode=code1s)
                else:
                    res = dy
                dy, _ = ops.conv_fused([Seg(dc1)], I[f'{bi}.c1t'], c1m.cin, ocode=code1, gate_x=x, res=res)
                if want_w and bi == cut and _BUCKETS and split:
                    # blocks cut .. last and the tail are done: reduce their slabs, fix them up, hand the bucket out
                    red.__exit__(None, None, None)
                    fix(bk['hi'], bk['i_cut'])
                    if _BUCKETS and split:
                        yield (bk['off_cut'], fp.numel(), False)
                    red = ops.deferred_reduces()
                    red.__enter__()
            # FirstDisResBlock
            b0, bc = self.res[0], ctx['blocks'][0]
            c1m, c2m, scm = (self.sn_of[b0.conv[0].module], self.sn_of[b0.conv[3].module], self.sn_of[b0.shortcut[0].module])
            c1, code, img = bc['c1'], bc['code'], ctx['img']
            if want_w:
                wgrad_all(lambda sl: Seg(c1[sl], code=U(0, sl), relu=True), dy, c2m.cout, c2m.cin, c2m.m.weight_orig,
                          c2m.m.bias, scm.m.bias, dy_ups=True, alpha=0.25)
                wgrad_all(lambda sl: Seg(img[sl], ksize=1), dy, scm.cout, scm.cin, scm.m.weight_orig, dy_ups=True, alpha=0.25)
            dc1, _ = ops.conv_fused([Seg(dy, ups=True)], I['0.c2t'], c2m.cin, ocode=code, gate_x=c1)
            if want_w:
                wgrad_all(lambda sl: Seg(img[sl]), dc1, c1m.cout, c1m.cin, c1m.m.weight_orig, c1m.m.bias)
            dimg = None
            if need_input_grad:
                if pair is not None:
                    raise RuntimeError('input gradients of a paired pass are not needed by the D update and not built')
                dimg_t, _ = ops.conv_fused([Seg(dc1), Seg(dy, ksize=1, ups=True)], I['0.dimg'], c1m.cin, cy=img.shape[-1])
                dimg = Nhwc(dimg_t, c1m.cin) if ctx.get('nhwc') else ops.to_nchw(dimg_t, c1m.cin)
        except BaseException as e:
            red.__exit__(type(e), e, None)
            raise
        red.__exit__(None, None, None)
        if want_w:
            if defer_fix and pair is not None and not (_BUCKETS and split) and not accumulate:
                # the caller finishes the update itself (FusedAdam.step_fused_sn_pair: fix-up + Adam in one launch per table):
                # hand it the raw per-half gradients and the spectral-norm state of the two forwards; gflat stays unwritten
                self.pending_fix = {'g0': passes[0][1], 'g1': passes[1][1], 'uv0': uv, 'uv1': pair['uv2'], 'sigma0': sigma,
                                    'sigma1': pair['sigma2'], 'tables': [(bk['all'], 0)]}
            else:
                if not (_BUCKETS and split):
                    fix(bk['hi'], bk['i_cut'])      # single bucket: the late layers were not fixed up mid-pass
                fix(bk['lo'], 0)
            yield ((0, bk['off_cut'], True) if (_BUCKETS and split) else (0, fp.numel(), True))
        return dimg
