"""Tensor-level wrappers over the C ABI (include/mcgen_hip.h).

Activations are torch tensors of shape [N, H, W, Cp] (channels last, Cp a
multiple of 8), dtype float32 or bfloat16, on a ROCm device.  All launches go
to torch's current stream, so they compose with torch.cuda.graphs and streams.
Nothing here has a CPU path: CPU tensors raise.
"""
from __future__ import annotations

import ctypes as C
import os as _os
from dataclasses import dataclass
from typing import Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import check
from ._tuning import flag as _flag

Tensor = torch.Tensor


def _dt(dtype: torch.dtype) -> int:
    if dtype == torch.float32:
        return _lib.F32
    if dtype == torch.bfloat16:
        return _lib.BF16
    raise _lib.McgenError(f'unsupported compute dtype {dtype}')


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[Tensor]) -> Optional[int]:
    if t is None:
        return None
    if not t.is_cuda:
        raise _lib.McgenError('mcgen_amd has no CPU path: tensor is not on a ROCm device')
    if not t.is_contiguous():
        raise _lib.McgenError('mcgen_amd kernels need contiguous tensors')
    return t.data_ptr()


def _f32(t: Optional[Tensor]) -> Optional[int]:
    if t is not None and t.dtype != torch.float32:
        raise _lib.McgenError(f'expected float32, got {t.dtype}')
    return _p(t)


# ---- per-launch profiling (bench.py roofline): HIP events on the launch stream -------------------------
_PROF = None
_PROF_SHAPES = _flag('MCGEN_PROF_SHAPES', '') != ''      # per-shape kernel names in the profile (tools/shape_table.py)
TILE_LOG = None        # tests set this to a list: every conv_fused launch appends the (BM, BN) tile the policy picked
FORM_LOG = None        # tests set this to a list: every conv_fused launch appends its weight layout (0 dense, 1 mc, 2 gk)
KERNEL_LOG = None      # tests set this to a list: every conv_fused launch appends mcgen_conv_form (0 tiled, 1 skinny, 2 whole images per workgroup, 3 resident-tile 1x1, 4 image convolution, 5 image head)


def _timed(name_fn, flops: float, launch, nbytes_fn=None, extra_fn=None):
    """Run `launch()`; when profiling is on, bracket it with events on the current stream.  `nbytes_fn()` = the
    launch's ALGORITHMIC HBM bytes (operands read once + results written once); `extra_fn()` = bytes the implementation
    moves on top of that (the split-K slabs of a weight-gradient launch), reported apart."""
    if _PROF is None:
        return launch()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    r = launch()
    e.record()
    _PROF.append((name_fn(), flops, s, e, float(nbytes_fn()) if nbytes_fn is not None else 0.0,
                  float(extra_fn()) if extra_fn is not None else 0.0))
    return r


HBM_PEAK_BYTES_PER_S = 8.0e12      # MI355X_MICROARCH.md: HBM3E spec peak


def _nbytes(*ts) -> int:
    return sum(t.numel() * t.element_size() for t in ts if t is not None)


def profile_step(fn, peak_tflops: float, iters: int = 5):
    """Run `fn` once untimed, then `iters` times with every conv / wgrad / slab-reduce launch bracketed by HIP events on
    the launch stream; returns the `roofline` object for the kernel instantiation with the largest total time:
    algorithmic FLOPs of its launches / sum of their measured durations, averaged over all `iters` passes, with the
    per-pass fractions' min / median beside it (`fn(i)` is called with the pass index if it takes an argument)."""
    global _PROF
    import inspect
    takes_i = len(inspect.signature(fn).parameters) >= 1
    call = (lambda i: fn(i)) if takes_i else (lambda i: fn())
    call(-1)
    torch.cuda.synchronize()
    recs = []
    for i in range(max(1, iters)):
        _PROF = []
        try:
            call(i)
            torch.cuda.synchronize()
            recs.append(_PROF)
        finally:
            _PROF = None
    # An event pair with nothing between its records already reads ~4.8 us on this stack (the second event's own
    # completion); measured live and taken off every bracket, otherwise short launches look 10-15 % slower than rocprofv3's
    # kernel durations (checked: 55.2 -> 50.4 us against 48.2 us for the 128x128 tile, 163.3 -> 158.5 against 159.3).
    pairs = []
    for _ in range(64):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); e.record(); pairs.append((s, e))
    torch.cuda.synchronize()
    over_ms = sorted(s.elapsed_time(e) for s, e in pairs)[len(pairs) // 2]

    def aggregate(rec):
        agg = {}
        for name, flops, s, e, nbytes, extra in rec:
            a = agg.setdefault(name, [0, 0.0, 0.0, 0.0, 0.0, 0.0])
            raw = s.elapsed_time(e)
            a[0] += 1; a[1] += max(raw - over_ms, 0.1 * raw) * 1e-3; a[2] += flops; a[3] += raw * 1e-3; a[4] += nbytes; a[5] += extra
        return agg
    per_pass = [aggregate(r) for r in recs]
    agg = {}
    for pa in per_pass:
        for k, v in pa.items():
            a = agg.setdefault(k, [0, 0.0, 0.0, 0.0, 0.0, 0.0])
            for j in range(6):
                a[j] += v[j]
    if not agg:
        return None
    np_ = len(per_pass)
    top = max(agg, key=lambda k: agg[k][1])
    cnt, secs, flops, raw_secs, nbytes, extra = agg[top]
    ach = flops / secs / 1e12
    # which roof bounds the kernel: its algorithmic intensity against the machine balance (MFMA peak / HBM peak)
    intensity = flops / nbytes if nbytes > 0 else float('inf')
    balance = peak_tflops * 1e12 / HBM_PEAK_BYTES_PER_S
    hbm = intensity < balance
    # the same figure per pass: how reproducible one eager iteration's reading is
    fr = sorted(((pa[top][4] / pa[top][1] / HBM_PEAK_BYTES_PER_S) if hbm else (pa[top][2] / pa[top][1] / 1e12 / peak_tflops))
                for pa in per_pass if top in pa and pa[top][1] > 0)
    out = {'bound': 'hbm' if hbm else 'mfma', 'kernel': top, 'achieved': ach, 'peak': peak_tflops,
           'unit': 'TFLOP/s', 'frac': ach / peak_tflops, 'traffic': None, 'launches_per_step': cnt / np_,
           'passes': np_, 'frac_min': fr[0], 'frac_median': fr[len(fr) // 2], 'frac_max': fr[-1],
           'avg_launch_us': secs / cnt * 1e6, 'avg_launch_us_uncorrected': raw_secs / cnt * 1e6,
           'event_pair_overhead_us': over_ms * 1e3, 'flops_per_launch': flops / cnt,
           'algorithmic_bytes_per_launch': nbytes / cnt, 'split_k_bytes_per_launch': extra / cnt,
           'intensity_flop_per_byte': intensity, 'machine_balance_flop_per_byte': balance,
           'by_kernel': {k: {'launches': v[0] / np_, 'total_ms': v[1] * 1e3 / np_, 'tflops': v[2] / v[1] / 1e12,
                             'gbytes_per_s': v[4] / v[1] / 1e9, **({'split_k_gbytes_per_s': v[5] / v[1] / 1e9} if v[5] else {})}
                         for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])}}
    if hbm:                                        # an HBM-bound dominant kernel is priced in bytes
        out.update(achieved=nbytes / secs / 1e9, peak=HBM_PEAK_BYTES_PER_S / 1e9, unit='GB/s',
                   frac=nbytes / secs / HBM_PEAK_BYTES_PER_S)
    return out


def pad8(c: int) -> int:
    return (c + 7) // 8 * 8


def pad16(c: int) -> int:
    return (c + 15) // 16 * 16


@dataclass
class Seg:
    """One K-segment of a fused convolution (mcgen_seg_t)."""
    x: Tensor                       # [N, H>>ups, W>>ups, C]
    ksize: int = 3
    scale: Optional[Tensor] = None  # BN affine
    shift: Optional[Tensor] = None
    code: Optional[Tensor] = None   # [N, C] MC code
    ups: bool = False
    relu: bool = False
    group_n: int = 0                # > 0: scale / shift are [N // group_n, C], one BatchNorm batch per group_n images
    cmap: Optional[Tensor] = None   # int16 [N, cmap_stride(C)] compaction map of `code` (mc_cmap): mode-compacted K loop
    cw: int = 0                     # kmajor=2: channels of the segment's WEIGHTS when x holds compacted channels (else 0)

    def fill(self, s: _lib.Seg):
        s.x = _p(self.x)
        s.scale, s.shift, s.code = _f32(self.scale), _f32(self.shift), _f32(self.code)
        s.C = self.x.shape[-1]
        s.ups, s.relu, s.ksize = int(self.ups), int(self.relu), self.ksize
        s.group_n = int(self.group_n)
        s.cmap, s.cmap_stride, s.Cw = None, 0, int(self.cw)
        if self.cmap is not None:
            cm = self.cw or s.C                                 # the map is over the weights' (true) channels
            if self.cmap.dtype != torch.int16 or tuple(self.cmap.shape) != (self.x.shape[0], cmap_stride(cm)):
                raise _lib.McgenError(f'cmap must be int16 {(self.x.shape[0], cmap_stride(cm))}, got {self.cmap.dtype} {tuple(self.cmap.shape)}')
            s.cmap, s.cmap_stride = _p(self.cmap), self.cmap.shape[1]
        if self.group_n and self.scale is not None:
            g = self.x.shape[0] // self.group_n
            if self.x.shape[0] % self.group_n or tuple(self.scale.shape) != (g, s.C) or tuple(self.shift.shape) != (g, s.C):
                raise _lib.McgenError(f'grouped BatchNorm affine must be {(g, s.C)}, got {tuple(self.scale.shape)}')
        if self.code is not None and tuple(self.code.shape) != (self.x.shape[0], self.x.shape[-1]):
            raise _lib.McgenError(f'code shape {tuple(self.code.shape)} does not match x {tuple(self.x.shape)}')


# --------------------------------------------------------------------------- #
def to_nhwc(x: Tensor, dtype: torch.dtype, cp: Optional[int] = None, out: Optional[Tensor] = None) -> Tensor:
    """NCHW fp32 -> NHWC `dtype`, channels zero-padded to a multiple of 8 (`out`: a contiguous [N, H, W, cp] buffer to fill)."""
    n, c, h, w = x.shape
    cp = cp or pad8(c)
    y = out if out is not None else torch.empty((n, h, w, cp), dtype=dtype, device=x.device)
    if tuple(y.shape) != (n, h, w, cp) or y.dtype != dtype or not y.is_contiguous():
        raise _lib.McgenError(f'to_nhwc: out must be a contiguous {(n, h, w, cp)} {dtype} tensor')
    check(_lib.load().mcgen_nchw_to_nhwc(_f32(x.contiguous()), _p(y), _dt(dtype), n, c, h, w, cp, _stream()), 'nchw_to_nhwc')
    return y


def to_nchw(x: Tensor, c: int) -> Tensor:
    n, h, w, cp = x.shape
    y = torch.empty((n, c, h, w), dtype=torch.float32, device=x.device)
    check(_lib.load().mcgen_nhwc_to_nchw(_p(x), _f32(y), _dt(x.dtype), n, c, h, w, cp, _stream()), 'nhwc_to_nchw')
    return y


def pool2_sum(x: Tensor) -> Tensor:
    """2x2 sums of an NHWC tensor (adjoint of the nearest x2 upsample)."""
    n, h, w, c = x.shape
    y = torch.empty((n, h // 2, w // 2, c), dtype=x.dtype, device=x.device)
    check(_lib.load().mcgen_pool2_sum(_p(x), _p(y), _dt(x.dtype), n, h // 2, w // 2, c, _stream()), 'pool2_sum')
    return y


def mc_code(indicator: Tensor, codebook: Tensor, cp: Optional[int] = None) -> Tensor:
    """code = indicator @ codebook  (modules.py:73), zero-padded to `cp` columns."""
    n, m = indicator.shape
    c = codebook.shape[1]
    code = torch.empty((n, c), dtype=torch.float32, device=indicator.device)
    check(_lib.load().mcgen_mc_code(_f32(indicator.contiguous()), _f32(codebook.contiguous()), _f32(code), n, m, c, _stream()), 'mc_code')
    if cp is not None and cp != c:
        code = torch.nn.functional.pad(code, (0, cp - c))
    return code


def mc_apply(x: Tensor, code: Tensor, channels_last: bool = True) -> Tensor:
    """x * code; x is [N, ..., C] (channels_last) or [N, C, ...] (the reference's layout)."""
    n = x.shape[0]
    c = x.shape[-1] if channels_last else x.shape[1]
    hw = x.numel() // (n * c)
    y = torch.empty_like(x)
    check(_lib.load().mcgen_mc_apply(_p(x), _f32(code), _p(y), _dt(x.dtype), n, hw, c, int(channels_last), _stream()), 'mc_apply')
    return y


def tile_images(n: int, h: int, w: int, cout: int, dtype: torch.dtype) -> int:
    """Images per output tile of a fused convolution over an [n, h, w] map with `cout` output channels, as the
    launcher's tile policy picks it (1 when a tile lies inside one image)."""
    p = _lib.Conv()
    p.nseg, p.N, p.H, p.W, p.Cout, p.Cout_w = 1, n, h, w, cout, pad16(cout)
    bm, bn = C.c_int(), C.c_int()
    check(_lib.load().mcgen_conv_tile(C.byref(p), _dt(dtype), C.byref(bm), C.byref(bn)), 'conv_tile')
    return max(1, bm.value // (h * w))


def cmap_stride(c: int) -> int:
    return int(_lib.load().mcgen_cmap_stride(c))


def onehot_rep(label: Tensor, classes: int, reps: int, out: Optional[Tensor] = None, lab32: Optional[Tensor] = None) -> Tensor:
    """[reps * N, classes] fp32: F.one_hot(label, classes).float() (mcgan.py:196,201) `reps` times back to back, one launch.
    `lab32` (optional, int32 [reps * N]) receives the labels themselves, repeated (the per-image weight-set index of conv_fused)."""
    n = label.shape[0]
    if label.dtype != torch.int64 or not label.is_cuda:
        raise _lib.McgenError('onehot_rep: int64 labels on the GPU')
    if out is None:
        out = torch.empty((reps * n, classes), dtype=torch.float32, device=label.device)
    assert tuple(out.shape) == (reps * n, classes) and out.dtype == torch.float32 and out.is_contiguous()
    if lab32 is not None:
        assert lab32.dtype == torch.int32 and lab32.numel() == reps * n and lab32.is_contiguous()
    check(_lib.load().mcgen_onehot_rep(_p(label.contiguous()), _f32(out), _p(lab32), n, classes, reps, _stream()), 'onehot_rep')
    return out


def onehot_hint(indicator: Tensor, label: Tensor, reps: int = 1, lab32: Optional[Tensor] = None) -> Tensor:
    """Marks `indicator` ([reps * N, modes]) as F.one_hot(label) repeated `reps` times: CodeBatch.run_any then gathers codebook
    rows by label instead of multiplying through every mode (the tensor is returned; the mark does not survive slicing)."""
    indicator._mcgen_onehot = (label, int(reps), lab32)
    return indicator


def mc_cmap(code: Tensor) -> Tensor:
    """Per-sample compaction map of a code tensor [N, C] (mcgen_mc_cmap): int16 [N, cmap_stride(C)]."""
    n, c = code.shape
    out = torch.empty((n, cmap_stride(c)), dtype=torch.int16, device=code.device)
    check(_lib.load().mcgen_mc_cmap(_f32(code), n, c, _p(out), _stream()), 'mc_cmap')
    return out


def mc_affine(code: Tensor, cmap: Tensor, ccap: int, scale: Optional[Tensor] = None, shift: Optional[Tensor] = None,
              group_n: int = 0):
    """Per-image prologue rows [N, ccap] of a launch that reads compacted activations (mcgen_mc_affine): the
    BatchNorm affine (scale / shift: [C] or [N // group_n, C]) gathered through the map and multiplied by the code."""
    n, c = code.shape
    out = torch.empty((2, n, ccap), dtype=torch.float32, device=code.device)
    check(_lib.load().mcgen_mc_affine(_f32(scale), _f32(shift), group_n, _f32(code), _p(cmap), n, c, ccap,
                                      out[0].data_ptr(), out[1].data_ptr(), _stream()), 'mc_affine')
    return out[0], out[1]


def weight_image_k_elems(cout: int, cin: int, ksize: int) -> int:
    return int(_lib.load().mcgen_weight_image_k_elems(cout, cin, ksize))


def prep_weight_k(w: Tensor, dtype: torch.dtype, sigma: Optional[Tensor] = None, wscale: float = 1.0,
                  out: Optional[Tensor] = None) -> Tensor:
    """Master weights [Cout, Cin, k, k] -> K-major image [tap][round_up(Cin, 8) + 1][pad16(Cout)] (last row zero)."""
    cout, cin = w.shape[0], w.shape[1]
    ks = w.shape[2] if w.dim() == 4 else 1
    n = weight_image_k_elems(cout, cin, ks)
    if out is None:
        out = torch.empty(n, dtype=dtype, device=w.device)
    assert out.numel() == n and out.dtype == dtype
    check(_lib.load().mcgen_prep_weight_k(_f32(w.contiguous()), _p(out), _dt(dtype), cout, cin, ks, _f32(sigma), float(wscale), _stream()),
          'prep_weight_k')
    return out


def weight_image_elems(cout: int, cin: int, ksize: int, transpose: bool = False) -> int:
    return int(_lib.load().mcgen_weight_image_elems(cout, cin, ksize, int(transpose)))


def prep_weight(w: Tensor, dtype: torch.dtype, transpose: bool = False, row_perm: int = 1,
                sigma: Optional[Tensor] = None, wscale: float = 1.0, out: Optional[Tensor] = None) -> Tensor:
    """Master weights [Cout, Cin, k, k] (or [Cout, Cin]) -> kernel weight image."""
    cout, cin = w.shape[0], w.shape[1]
    ks = w.shape[2] if w.dim() == 4 else 1
    n = weight_image_elems(cout, cin, ks, transpose)
    if out is None:
        out = torch.empty(n, dtype=dtype, device=w.device)
    assert out.numel() == n and out.dtype == dtype
    check(_lib.load().mcgen_prep_weight(_f32(w), _p(out), _dt(dtype), cout, cin, ks, int(transpose), row_perm,
                                        _f32(sigma), float(wscale), _stream()), 'prep_weight')
    return out


def conv_fused(segs: Sequence[Seg], wimg: Tensor, cout: int, *, bias: Optional[Tensor] = None, bias2: Optional[Tensor] = None,
               pool: bool = False, alpha: float = 1.0, res: Optional[Tensor] = None,
               ocode: Optional[Tensor] = None, gate_x: Optional[Tensor] = None,
               gscale: Optional[Tensor] = None, gshift: Optional[Tensor] = None,
               gmean: Optional[Tensor] = None, grstd: Optional[Tensor] = None,
               tanh: bool = False, stats_mode: int = 0, cy: Optional[int] = None,
               out: Optional[Tensor] = None, kmajor: int = 0, ycmap: Optional[Tensor] = None,
               y_group: int = 0, wsel: Optional[Tensor] = None, order: Optional[Tensor] = None,
               yperm: Optional[Tensor] = None) -> Tuple[Tensor, Optional[Tensor]]:
    """Launch mcgen_conv_fused; returns (y, per-tile stats partials or None).  `kmajor`: `wimg` is the K-major image
    (prep_weight_k, segments concatenated); 1: every segment carries a compaction map and is compacted while staged;
    2: segments hold ALREADY compacted channels (Seg.cw, cmap) or are dense.  `ycmap` (+ `cy` = compacted pitch): the
    output keeps, per image, only the channels of that map, compacted (forward-only passes).  `y_group` (image head only):
    `out` holds 2 N images and output image n lands in slot (n // y_group) * 2 * y_group + y_group + n % y_group -- the
    second halves of N / y_group paired [real (+) generated] batches (mcgen_conv_t.y_group).
    `wsel` (int32 [N]) + `order` (int32 [N], a permutation; optional): per-mode dense weight sets -- `wimg` holds S images of
    the segments' (compacted) shapes back to back, the image walked at position i is order[i] and multiplies weight set
    wsel[i] (mcgen_conv_t.wsel: bf16, software-pipelined form only).  `yperm` (int16 [S, stride], with wsel and `cy`): the sets'
    weight ROWS were permuted at prep time (PrepBatch rmap = yperm[s]): the output comes out compacted (pitch cy) without a
    gather pass; bias and statistics stay in true channel order (mcgen_conv_t.yperm)."""
    s0 = segs[0]
    n = s0.x.shape[0]
    h = s0.x.shape[1] * (2 if s0.ups else 1)
    w = s0.x.shape[2] * (2 if s0.ups else 1)
    dtype = s0.x.dtype
    cy = cy or pad8(cout)
    ho, wo = (h // 2, w // 2) if pool else (h, w)
    p = _lib.Conv()
    p.nseg = len(segs)
    for i, s in enumerate(segs):
        if s.x.dtype != dtype:
            raise _lib.McgenError('all segments must share the compute dtype')
        hs = s.x.shape[1] * (2 if s.ups else 1)
        if hs != h or s.x.shape[0] != n:
            raise _lib.McgenError('segments disagree on the convolution resolution')
        s.fill(p.seg[i])
    if wimg.dtype != dtype:
        raise _lib.McgenError(f'weight image dtype {wimg.dtype} != activation dtype {dtype}')
    if kmajor:
        need = sum(s.ksize * s.ksize * ((s.cw or s.x.shape[-1]) + 1) for s in segs) * pad16(cout)
    else:
        need = sum(((s.x.shape[-1] + 31) // 32) * s.ksize * s.ksize for s in segs) * pad16(cout) * 32
    if wsel is not None:
        if wsel.dtype != torch.int32 or wsel.numel() != n or (order is not None and (order.dtype != torch.int32 or order.numel() != n)):
            raise _lib.McgenError('wsel / order must be int32 [N]')
        if wimg.numel() % need:
            raise _lib.McgenError(f'weight sets: {wimg.numel()} elements are not a multiple of one image ({need})')
    elif order is not None:
        raise _lib.McgenError('order comes with wsel')
    elif wimg.numel() != need:
        raise _lib.McgenError(f'weight image has {wimg.numel()} elements, the segments need {need}')
    y = out if out is not None else torch.empty((n, ho, wo, cy), dtype=dtype, device=s0.x.device)
    if y_group:
        if out is None or tuple(y.shape) != (2 * n, ho, wo, cy) or y.dtype != dtype or not y.is_contiguous():
            raise _lib.McgenError(f'y_group: `out` must be a contiguous {(2 * n, ho, wo, cy)} {dtype} buffer')
    else:
        assert tuple(y.shape) == (n, ho, wo, cy) and y.dtype == dtype
    p.w, p.bias, p.y = _p(wimg), _f32(bias), _p(y)
    p.bias2 = _f32(bias2)
    p.N, p.H, p.W = n, h, w
    p.Cout, p.Cout_w, p.Cy = cout, pad16(cout), cy
    p.pool, p.alpha = int(pool), float(alpha)
    for name, t in (('res', res), ('gate_x', gate_x)):
        if t is not None and (tuple(t.shape) != (n, ho, wo, cy) or t.dtype != dtype):
            raise _lib.McgenError(f'{name} must be {(n, ho, wo, cy)} {dtype}, got {tuple(t.shape)} {t.dtype}')
    if ocode is not None and tuple(ocode.shape) != (n, cout):
        raise _lib.McgenError(f'ocode must be {(n, cout)}')
    p.res, p.ocode, p.gate_x = _p(res), _f32(ocode), _p(gate_x)
    p.gscale, p.gshift, p.gmean, p.grstd = _f32(gscale), _f32(gshift), _f32(gmean), _f32(grstd)
    p.tanh_out, p.stats_mode = int(tanh), stats_mode
    p.w_layout = int(kmajor)
    p.y_group = int(y_group)
    p.wsel, p.wsel_stride, p.order = _p(wsel), (need if wsel is not None else 0), _p(order)
    p.yperm, p.yperm_stride = None, 0
    if yperm is not None:
        if wsel is None or yperm.dtype != torch.int16 or yperm.dim() != 2 or yperm.shape[0] * need != wimg.numel() or not yperm.is_contiguous():
            raise _lib.McgenError('yperm must be a contiguous int16 [sets, stride] table that comes with wsel')
        p.yperm, p.yperm_stride = _p(yperm), yperm.shape[1]
    p.ycmap, p.ycmap_stride = None, 0
    if ycmap is not None:
        if ycmap.dtype != torch.int16 or tuple(ycmap.shape) != (n, cmap_stride(pad8(cout))):
            raise _lib.McgenError(f'ycmap must be int16 {(n, cmap_stride(pad8(cout)))}')
        p.ycmap, p.ycmap_stride = _p(ycmap), ycmap.shape[1]
    stats = None
    lib = _lib.load()
    if stats_mode:
        tiles = lib.mcgen_conv_m_tiles(C.byref(p), _dt(dtype))
        stats = torch.empty((tiles, 2, pad16(cout) if (ycmap is not None or yperm is not None) else cy), dtype=torch.float32, device=y.device)
        p.stats = _p(stats)
    kflops = 2.0 * n * h * w * cout * sum(s.ksize * s.ksize * s.x.shape[-1] for s in segs)

    def _name():
        bm, bn = C.c_int(), C.c_int()
        lib.mcgen_conv_tile(C.byref(p), _dt(dtype), C.byref(bm), C.byref(bn))
        form = lib.mcgen_conv_form(C.byref(p), _dt(dtype))
        base = f'conv_fused<{"bf16" if dtype == torch.bfloat16 else "f32"},{bm.value},{bn.value}{(",mc", ",gk")[kmajor - 1] if kmajor else ""}>'
        if form:
            base = ('conv_skinny<bf16>', 'conv_smap<bf16>', 'conv_px1<bf16>', 'conv_c8<bf16>', 'conv_head<bf16>')[form - 1]
        if _PROF_SHAPES:
            base += f' N{n} {h}x{w} ' + '+'.join(f'{s.x.shape[-1]}k{s.ksize}' for s in segs) + f'->{cout}' + \
                ('g' if gate_x is not None else '') + ('p' if pool else '') + (f's{stats_mode}' if stats_mode else '')
        return base
    if TILE_LOG is not None:
        bm, bn = C.c_int(), C.c_int()
        check(lib.mcgen_conv_tile(C.byref(p), _dt(dtype), C.byref(bm), C.byref(bn)), 'conv_tile')
        TILE_LOG.append((bm.value, bn.value))
    if FORM_LOG is not None:
        FORM_LOG.append(kmajor)
    if KERNEL_LOG is not None:
        KERNEL_LOG.append(int(lib.mcgen_conv_form(C.byref(p), _dt(dtype))))
    _timed(_name, kflops, lambda: check(lib.mcgen_conv_fused(C.byref(p), _dt(dtype), _stream()), 'conv_fused'),
           lambda: _nbytes(wimg, y, res, gate_x, *[s.x for s in segs]))
    return y, stats


def wgrad(seg: Seg, dy: Tensor, cout: int, cin: int, grad: Tensor, *, dy_ups: bool = False,
          alpha: float = 1.0, accumulate: bool = False, row_perm: int = 1, splits: Optional[int] = None,
          bias_grad: Optional[Tensor] = None, bias_grad2: Optional[Tensor] = None, second=None,
          row_scale: Optional[Tensor] = None, taps: Optional[Tuple[int, int]] = None):
    """grad[Cout, Cin, k, k] (+)= alpha * dW of one fused-conv segment; optionally also the bias
    gradient bias_grad[Cout] (+)= alpha * sum_pixels dy (and a copy into bias_grad2).
    `cin` may be smaller than the (padded) channel count of seg.x: only the first cin input channels are written.
    `row_scale[Cout]` (optional) multiplies row co of the weight gradient and entry co of the bias gradients.
    `second` = (grad_b, bias_grad_b, bias_grad2_b): the batch is two halves (paired discriminator pass) and the
    second half's gradient goes to these tensors instead -- one launch, one slab set per half.
    `taps` = (tap0, n): grad is [Cout, Cin, n] and receives taps tap0 .. tap0 + n - 1 of the k x k gradient only (a sub-kernel
    embedded in the 3x3 image: MCGatedMaskedConv2d's (2 x 3) / (1 x 2) stacks) -- no scratch tensor + slice copy."""
    n = seg.x.shape[0]
    h = seg.x.shape[1] * (2 if seg.ups else 1)
    w = seg.x.shape[2] * (2 if seg.ups else 1)
    dtype = seg.x.dtype
    p = _lib.Wgrad()
    seg.fill(p.seg)
    p.dy = _p(dy)
    p.N, p.H, p.W = n, h, w
    p.Cout, p.Cout_w, p.Cdy = cout, pad16(cout), dy.shape[-1]
    p.dy_ups = int(dy_ups)
    exp = (n, h // 2, w // 2) if dy_ups else (n, h, w)
    if tuple(dy.shape[:3]) != exp or dy.dtype != dtype:
        raise _lib.McgenError(f'dy must be {exp}+[C] {dtype}, got {tuple(dy.shape)} {dy.dtype}')
    m_tiles = (n * h * w + 127) // 128
    nchunk = (seg.x.shape[-1] + 31) // 32
    explicit_splits = splits
    # the image-side layers (conv input = the 8-channel image tensor) have their own kernel and compact slabs (wgrad_c8.hip)
    c8 = bool(_lib.load().mcgen_wgrad_c8_ok(C.byref(p), _dt(dtype)))
    if WGRAD_LOG is not None:
        WGRAD_LOG.append('c8' if c8 else 'general')
    if splits is None:
        # 1x1 gradients with >= 4 chunks run as chunk groups of 4 (wgrad.hip): a quarter of the workgroups per split
        # (wgrad.hip: the ring form -- bf16, tiles inside one image -- comes first; chunk groups are for the small maps)
        ring = dtype == torch.bfloat16 and h * w >= 128 and w <= 64 and (n * h * w) % 128 == 0
        grouped = seg.ksize == 1 and nchunk >= _WG_GROUP_MIN and dtype == torch.bfloat16 and not ring
        blocks = ((pad16(cout) + 63) // 64) * ((nchunk + 3) // 4 if grouped else nchunk)
        # enough workgroups to fill the chip matters more than the slab traffic (measured: 8x8 layers lose
        # 20 % with 16 instead of 64 splits)
        target = _WG_TARGET if m_tiles >= _WG_BIG_TILES else _WG_TARGET_SMALL
        splits = max(1, min(m_tiles, (target + blocks - 1) // blocks, _WG_MAX_SPLITS))
        if c8:
            # the image-layer kernel streams dy with every output channel in one workgroup (256-pixel steps): one per CU
            splits = max(1, min(m_tiles // 2, _cu_count(dy.device)))
        if second is not None:
            splits = max(2, splits + (splits & 1))
    if second is not None and ((n * h * w) % 256 != 0 or splits % 2):
        raise _lib.McgenError('wgrad: a two-half launch needs whole 128-pixel tiles per half and even splits')
    p.splits = splits
    p.halves = int(second is not None)
    lib = _lib.load()
    tap0, ntap_out = taps if taps is not None else (0, 0)
    if taps is not None and (second is not None or not (0 <= tap0 and 0 < ntap_out and tap0 + ntap_out <= seg.ksize ** 2)):
        raise _lib.McgenError('wgrad: bad tap window (and tap windows do not combine with two-half launches)')
    if grad.numel() != cout * cin * (ntap_out or seg.ksize * seg.ksize):
        raise _lib.McgenError(f'grad has {grad.numel()} elements, expected {cout * cin * (ntap_out or seg.ksize ** 2)}')
    if (_deferred is not None and _MULTI and explicit_splits is None and dtype == torch.bfloat16
            and lib.mcgen_wgrad_multi_ok(C.byref(p), _dt(dtype))):
        # The 3x3 layers of a backward pass share ONE launch (mcgen_wgrad_multi): queued here, launched when the pass's
        # deferred_reduces context closes -- its pixel splits are sized by the layer's share of the pass's FLOPs.
        if not grad.is_contiguous():
            raise _lib.McgenError('deferred wgrad reduce needs a contiguous gradient tensor')
        _deferred.append(_PendingMulti(p, seg, dy, cout, cin, grad, bias_grad, bias_grad2, second, alpha, accumulate, row_perm,
                                       row_scale, m_tiles, (pad16(cout) // 128) * (seg.x.shape[-1] // 64), (tap0, ntap_out)))
        return
    if (_deferred is not None and _BATCH and dtype == torch.bfloat16 and not c8 and second is None and explicit_splits is None
            and not _SIDE and splits * _lib.WGRAD_MULTI_MAX <= 65535):
        # Layers the pass-wide kernel does not take (skinny channel counts, 4x4 maps ...) wait for the end of the pass too: those of
        # IDENTICAL shape share one launch of their kernel (mcgen_wgrad_batch: MCGlow has 16 flows per level)
        if not grad.is_contiguous():
            raise _lib.McgenError('deferred wgrad reduce needs a contiguous gradient tensor')
        q = _PendingMulti(p, seg, dy, cout, cin, grad, bias_grad, bias_grad2, None, alpha, accumulate, row_perm, row_scale, m_tiles, 0,
                          (tap0, ntap_out))
        q.splits, q.batch = splits, True
        _deferred.append(q)
        return
    elems = int(lib.mcgen_wgrad_c8_slab_elems(C.byref(p)) if c8 else lib.mcgen_wgrad_slab_elems(C.byref(p)))
    # Inside a deferred_reduces() pass the split-K kernel goes to a side stream: it depends only on tensors that
    # already exist, and nothing reads its slabs before the pass's batched reduce, so it overlaps the
    # input-gradient chain that continues on the main stream.
    side = _side_stream(dy.device) if (_deferred is not None and _SIDE) else None
    if side is not None:
        side.wait_stream(torch.cuda.current_stream())
        _side_keep.extend((seg.x, dy, seg.scale, seg.shift, seg.code))      # keep alive until the join
    with (torch.cuda.stream(side) if side is not None else _nullctx()):
        slabs = torch.empty((splits, elems), dtype=torch.float32, device=dy.device)
        p.slabs = _p(slabs)
        bias_slabs = None
        if bias_grad is not None:
            bias_slabs = torch.empty((splits * 4, pad16(cout)), dtype=torch.float32, device=dy.device)
        p.bias_slabs = _p(bias_slabs)
        _timed(lambda: f'wgrad<{"bf16" if dtype == torch.bfloat16 else "f32"},{seg.ksize}>' + (
            f' N{n} {h}x{w} {seg.x.shape[-1]}->{cout} s{splits}' if _PROF_SHAPES else ''),
               2.0 * n * h * w * cout * seg.x.shape[-1] * seg.ksize ** 2,
               lambda: check(lib.mcgen_wgrad(C.byref(p), _dt(dtype), _stream()), 'wgrad'),
               # algorithmic bytes: x + dy read once, dW (+ db) written once; the split-K slabs are the implementation's
               lambda: _nbytes(seg.x, dy) + 4 * (cout * cin * seg.ksize ** 2 + (cout if bias_grad is not None else 0)) * (2 if second is not None else 1),
               lambda: _nbytes(slabs, bias_slabs))
    if second is None:
        parts = [(slabs, bias_slabs, grad, bias_grad, bias_grad2, splits)]
    else:
        hs = splits // 2
        parts = [(slabs[:hs], bias_slabs[:hs * 4] if bias_slabs is not None else None, grad, bias_grad, bias_grad2, hs),
                 (slabs[hs:], bias_slabs[hs * 4:] if bias_slabs is not None else None, second[0], second[1], second[2], hs)]
    for sl, bs, gr, bg, bg2, ns in parts:
        if _deferred is not None:
            if not gr.is_contiguous():
                raise _lib.McgenError('deferred wgrad reduce needs a contiguous gradient tensor')
            _deferred.append((sl, gr, bs, bg, bg2, ns, cout, cin, seg.ksize, pad16(cout), row_perm, int(accumulate), float(alpha),
                              row_scale, seg.x.shape[-1], int(c8), tap0, ntap_out))
        else:
            _timed(lambda: 'wgrad_reduce', 0.0,
                   lambda: check(lib.mcgen_wgrad_reduce(_p(sl), ns, _f32(gr), cout, cin, seg.ksize, pad16(cout), row_perm,
                                                        float(alpha), int(accumulate), _p(bs), _f32(bg), _f32(bg2), _f32(row_scale),
                                                        seg.x.shape[-1], int(c8), tap0, ntap_out, _stream()), 'wgrad_reduce'),
                   lambda: _nbytes(gr), lambda: _nbytes(sl, bs))


MULTI_LOG = None         # tools: receives (map side, ksize, 128-pixel steps, tiles, splits) per layer of every wgrad_multi launch
WGRAD_LOG = None         # tests: a list that receives which weight-gradient kernel family a call with default splits takes
_deferred = None
_MULTI = _flag('MCGEN_WGRAD_MULTI', '1') != '0'        # eligible 3x3 weight gradients of a pass as one mcgen_wgrad_multi launch
_BATCH = _flag('MCGEN_WGRAD_BATCH', '1') != '0'        # the other layers of a pass: same-shape groups as one mcgen_wgrad_batch launch
BATCH_LOG = None                                       # tools / tests: receives the layer count of every mcgen_wgrad_batch launch
_CU_COUNT = {}


def _cu_count(device) -> int:
    n = _CU_COUNT.get(device)
    if n is None:
        n = _CU_COUNT[device] = torch.cuda.get_device_properties(device).multi_processor_count
    return n


class _PendingMulti:
    """One queued layer of a mcgen_wgrad_multi launch (see ops.wgrad)."""
    __slots__ = ('p', 'seg', 'dy', 'cout', 'cin', 'grad', 'bias_grad', 'bias_grad2', 'second', 'alpha', 'accumulate', 'row_perm',
                 'row_scale', 'm_tiles', 'blocks', 'splits', 'slabs', 'bias_slabs', 'taps', 'batch')

    def __init__(self, p, seg, dy, cout, cin, grad, bias_grad, bias_grad2, second, alpha, accumulate, row_perm, row_scale, m_tiles, blocks,
                 taps=(0, 0)):
        self.taps = taps
        self.batch = False            # True: a layer of a same-shape batch (mcgen_wgrad_batch), splits already chosen
        self.p, self.seg, self.dy, self.cout, self.cin = p, seg, dy, cout, cin
        self.grad, self.bias_grad, self.bias_grad2, self.second = grad, bias_grad, bias_grad2, second
        self.alpha, self.accumulate, self.row_perm, self.row_scale = alpha, accumulate, row_perm, row_scale
        self.m_tiles, self.blocks = m_tiles, blocks

    def jobs(self):
        """The slab-reduce job(s) of the layer, in deferred_reduces' tuple form."""
        cw, ks, cs = pad16(self.cout), self.seg.ksize, self.seg.x.shape[-1]
        tail = (self.cout, self.cin, ks, cw, self.row_perm, int(self.accumulate), float(self.alpha), self.row_scale, cs, 0) + tuple(self.taps)
        if self.second is None:
            return [(self.slabs, self.grad, self.bias_slabs, self.bias_grad, self.bias_grad2, self.splits) + tail]
        hs = self.splits // 2
        b0 = self.bias_slabs[:hs * 4] if self.bias_slabs is not None else None
        b1 = self.bias_slabs[hs * 4:] if self.bias_slabs is not None else None
        return [(self.slabs[:hs], self.grad, b0, self.bias_grad, self.bias_grad2, hs) + tail,
                (self.slabs[hs:], self.second[0], b1, self.second[1], self.second[2], hs) + tail]


_WG_W1 = float(_flag('MCGEN_WG_W1', '0.55'))        # cost of a 1x1 layer's 128-pixel step relative to a 3x3 layer's (the same staging, a ninth of the MFMAs)
_WG_W16 = float(_flag('MCGEN_WG_W16', '1.0'))      # ... of a step on 16x16 maps, on 8x8 maps (more halo per step)
_WG_W8 = float(_flag('MCGEN_WG_W8', '1.0'))
_WG_FIX = float(_flag('MCGEN_WG_FIX', '5'))        # fixed cost of a workgroup (setup, the accumulator flush) in steps


def _launch_multi(pend):
    """Size the queued layers' pixel splits by their share of the pass's work (128-pixel steps x workgroup tiles), one
    workgroup per CU over the whole launch, allocate the slabs and launch mcgen_wgrad_multi (<= MCGEN_WGRAD_MULTI_MAX layers each)."""
    lib = _lib.load()
    dev = pend[0].dy.device
    for base in range(0, len(pend), _lib.WGRAD_MULTI_MAX):
        grp = pend[base:base + _lib.WGRAD_MULTI_MAX]
        budget = _cu_count(dev)
        # Cost model of one workgroup of layer q with `sp` pixel splits: fix(q) + (m_tiles / sp) * w(q) in units of a 3x3
        # layer's 128-pixel step; the launch lasts as long as its slowest workgroup, so the splits are the smallest that
        # bring every layer under a common time T, T as small as the CU budget allows.  Calibrated with
        # tools/wgmulti_cost.py (one layer per launch, 16 .. 128 steps per workgroup): 3x3 4.0 us per step + 20 us fixed on
        # every map size, 1x1 2.15 us per step + 6 us: w(1x1) = 0.55, fix = 5 steps (3x3) / 1.5 (1x1).  (The first version
        # split in proportion to w * steps with w(1x1) = 0.35 and no fixed part: the generator's three shortcut layers got one
        # workgroup per tile and ran 256 steps each while the 3x3 layers' workgroups were done after 86: 518 us for a pass
        # that now takes 388; the discriminator's launches 143 -> 136 us.)
        def wq(q):
            side = q.p.H
            w = 1.0 if q.seg.ksize == 3 else _WG_W1
            return w * (_WG_W8 if side <= 8 else (_WG_W16 if side <= 16 else 1.0))

        def need(q, t):
            unit = 2 if q.second is not None else 1
            cap = max(unit, q.m_tiles // unit * unit)
            fix = _WG_FIX if q.seg.ksize == 3 else 0.3 * _WG_FIX
            if t <= fix:
                return cap
            sp = int(-(-q.m_tiles * wq(q) // (t - fix)))
            sp = -(-sp // unit) * unit
            return max(unit, min(cap, sp))
        lo, hi = _WG_FIX, _WG_FIX + max(q.m_tiles * wq(q) for q in grp) + 1.0
        for _ in range(40):
            mid = 0.5 * (lo + hi)
            if sum(need(q, mid) * q.blocks for q in grp) <= budget:
                hi = mid
            else:
                lo = mid
        for q in grp:
            q.splits = need(q, hi)
        # hand the workgroups left over to the layers with the most work per workgroup
        used = sum(q.splits * q.blocks for q in grp)
        while True:
            best = None
            for q in grp:
                unit = 2 if q.second is not None else 1
                if q.splits + unit <= q.m_tiles // unit * unit and used + unit * q.blocks <= budget:
                    load = q.m_tiles / q.splits * wq(q)
                    if best is None or load > best[0]:
                        best = (load, q, unit)
            if best is None:
                break
            best[1].splits += best[2]
            used += best[2] * best[1].blocks
        if MULTI_LOG is not None:
            MULTI_LOG.append([(q.p.H, q.seg.ksize, q.m_tiles, q.blocks, q.splits) for q in grp])
        arr = (_lib.Wgrad * len(grp))()
        flops = nbytes = extra = 0.0
        for a, q in zip(arr, grp):
            elems = int(lib.mcgen_wgrad_slab_elems(C.byref(q.p)))
            q.slabs = torch.empty((q.splits, elems), dtype=torch.float32, device=dev)
            q.bias_slabs = (torch.empty((q.splits * 4, pad16(q.cout)), dtype=torch.float32, device=dev)
                            if q.bias_grad is not None else None)
            q.p.splits, q.p.slabs, q.p.bias_slabs = q.splits, _p(q.slabs), _p(q.bias_slabs)
            C.memmove(C.byref(a), C.byref(q.p), C.sizeof(_lib.Wgrad))
            n, hh, ww = q.p.N, q.p.H, q.p.W
            flops += 2.0 * n * hh * ww * q.cout * q.seg.x.shape[-1] * q.seg.ksize ** 2
            nbytes += _nbytes(q.seg.x, q.dy) + 4 * (q.cout * q.cin * q.seg.ksize ** 2 + (q.cout if q.bias_grad is not None else 0)) * (2 if q.second is not None else 1)
            extra += _nbytes(q.slabs, q.bias_slabs)
        _timed(lambda: 'wgrad_multi<bf16>', flops,
               lambda: check(lib.mcgen_wgrad_multi(arr, len(grp), _lib.BF16, _stream()), 'wgrad_multi'),
               lambda: nbytes, lambda: extra)


def _launch_batches(pend):
    """Queued layers outside mcgen_wgrad_multi: groups of identical shape as one mcgen_wgrad_batch launch each (<= MCGEN_WGRAD_MULTI_MAX
    layers), the rest one by one; slabs allocated here, the reduce jobs follow with the pass's batched reduce."""
    lib = _lib.load()
    groups = {}
    for q in pend:
        p = q.p
        key = (p.N, p.H, p.W, p.Cout, p.Cout_w, p.Cdy, p.dy_ups, p.splits, p.seg.C, p.seg.ksize, p.seg.ups, q.bias_grad is not None)
        groups.setdefault(key, []).append(q)
    for grp_all in groups.values():
        for base in range(0, len(grp_all), _lib.WGRAD_MULTI_MAX):
            grp = grp_all[base:base + _lib.WGRAD_MULTI_MAX]
            dev = grp[0].dy.device
            arr = (_lib.Wgrad * len(grp))()
            flops = nbytes = extra = 0.0
            for a, q in zip(arr, grp):
                elems = int(lib.mcgen_wgrad_slab_elems(C.byref(q.p)))
                q.slabs = torch.empty((q.splits, elems), dtype=torch.float32, device=dev)
                q.bias_slabs = (torch.empty((q.splits * 4, pad16(q.cout)), dtype=torch.float32, device=dev)
                                if q.bias_grad is not None else None)
                q.p.slabs, q.p.bias_slabs = _p(q.slabs), _p(q.bias_slabs)
                C.memmove(C.byref(a), C.byref(q.p), C.sizeof(_lib.Wgrad))
                flops += 2.0 * q.p.N * q.p.H * q.p.W * q.cout * q.seg.x.shape[-1] * q.seg.ksize ** 2
                nbytes += _nbytes(q.seg.x, q.dy) + 4 * (q.cout * q.cin * q.seg.ksize ** 2 + (q.cout if q.bias_grad is not None else 0))
                extra += _nbytes(q.slabs, q.bias_slabs)
            ks = grp[0].seg.ksize
            if BATCH_LOG is not None:
                BATCH_LOG.append(len(grp))
            if len(grp) == 1:
                _timed(lambda: f'wgrad<bf16,{ks}>', flops,
                       lambda: check(lib.mcgen_wgrad(C.byref(grp[0].p), _lib.BF16, _stream()), 'wgrad'), lambda: nbytes, lambda: extra)
            else:
                _timed(lambda: f'wgrad<bf16,{ks}>', flops,
                       lambda: check(lib.mcgen_wgrad_batch(arr, len(grp), _lib.BF16, _stream()), 'wgrad_batch'), lambda: nbytes, lambda: extra)


_SIDE = _flag('MCGEN_SIDE_STREAM', '0') == '1'     # measured slower on MI355X (16.7 vs 15.7 ms/step): opt-in only
_side_streams = {}
_side_keep = []


def _side_stream(device):
    s = _side_streams.get(device)
    if s is None:
        s = _side_streams[device] = torch.cuda.Stream(device=device)
    return s


class _nullctx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


_WG_GROUP_MIN = 1 << 30 if _flag('MCGEN_WGRAD_GROUP', '1') == '0' else 4      # chunks from which a 1x1 gradient runs as chunk groups
_WG_MAX_SPLITS = int(_flag('MCGEN_WGRAD_MAX_SPLITS', '128'))   # (64 left the thin image layers -- 2 blocks -- on half the chip)
_WG_TARGET = int(_flag('MCGEN_WGRAD_TARGET', '256'))   # workgroups a weight-gradient launch aims for
_WG_TARGET_SMALL = int(_flag('MCGEN_WGRAD_TARGET_SMALL', '256'))
_WG_BIG_TILES = int(_flag('MCGEN_WGRAD_BIG_TILES', '256'))


class deferred_reduces:
    """Within this context every ops.wgrad launches only its split-K kernel; the slab reductions of the whole pass
    run as ONE table-driven launch on exit (a backward pass has ~10 of them, each too small to fill the chip)."""

    def __enter__(self):
        global _deferred
        self._outer = _deferred
        _deferred = []
        return self

    def __exit__(self, et, ev, tb):
        global _deferred
        jobs, _deferred = _deferred, self._outer
        pend = [j for j in jobs if isinstance(j, _PendingMulti) and not j.batch]
        same = [j for j in jobs if isinstance(j, _PendingMulti) and j.batch]
        if et is None and same:
            _launch_batches(same)
        if et is None and (pend or same):
            if pend:
                _launch_multi(pend)
            jobs = [t for j in jobs for t in (j.jobs() if isinstance(j, _PendingMulti) else [j])]
        elif pend or same:
            jobs = [j for j in jobs if not isinstance(j, _PendingMulti)]
        if _SIDE and _side_keep:
            for s in _side_streams.values():                 # join: the slabs are complete before they are reduced
                torch.cuda.current_stream().wait_stream(s)
            if self._outer is None:
                _side_keep.clear()
        if et is None and jobs:
            arr = (_lib.WReduce * len(jobs))()
            for a, (slabs, grad, bs, bg, bg2, splits, cout, cin, ks, cout_w, row_perm, acc, alpha, rscale, cin_slab, tapcols, tap0, ntap_out) in zip(arr, jobs):
                a.slabs, a.grad, a.bias_slabs = _p(slabs), _f32(grad), _p(bs)
                a.bias_grad, a.bias_grad2 = _f32(bg) if bs is not None else None, _f32(bg2) if bs is not None else None
                a.splits, a.Cout, a.Cin, a.ksize, a.Cout_w = splits, cout, cin, ks, cout_w
                a.row_perm, a.accumulate, a.alpha = row_perm, acc, alpha
                a.row_scale, a.cin_slab, a.tapcols, a.tap0, a.ntap_out = _f32(rscale), cin_slab, tapcols, tap0, ntap_out
            _timed(lambda: 'wgrad_reduce', 0.0,
                   lambda: check(_lib.load().mcgen_wgrad_reduce_batch(arr, len(jobs), _stream()), 'wgrad_reduce_batch'),
                   lambda: sum(_nbytes(j[1]) for j in jobs), lambda: sum(_nbytes(j[0], j[2]) for j in jobs))
        return False


def bn_finalize(partials: Tensor, count: int, gamma: Tensor, beta: Tensor,
                running_mean: Optional[Tensor], running_var: Optional[Tensor],
                momentum: float = 0.1, eps: float = 1e-5, fold: int = 1, groups: int = 1):
    """-> (scale, shift, mean, rstd), each [C] ([groups, C] for groups > 1: `count` is per group and the running stats
    take the groups' momentum updates in order); running stats updated in place."""
    tiles, _, pitch = partials.shape
    c = gamma.numel()
    out = torch.empty((4, groups, c), dtype=torch.float32, device=partials.device)
    check(_lib.load().mcgen_bn_finalize_groups(_f32(partials), tiles, pitch, fold, c, float(count), groups, _f32(gamma), _f32(beta),
                                               _f32(running_mean), _f32(running_var), momentum, eps,
                                               out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), out[3].data_ptr(),
                                               _stream()), 'bn_finalize')
    if groups == 1:
        out = out.view(4, c)
    return out[0], out[1], out[2], out[3]


def bn_finalize_batch(items):
    """items = [(partials, count, gamma, beta, running_mean, running_var, momentum, eps)]: several independent BatchNorm layers
    finalized in ONE launch (mcgen_bn_finalize_batch, <= 4) -> [(scale, shift, mean, rstd)] as bn_finalize gives per layer."""
    arr = (_lib.BnFin * len(items))()
    outs = []
    for d, (partials, count, gamma, beta, rm, rv, momentum, eps) in zip(arr, items):
        tiles, _, pitch = partials.shape
        c = gamma.numel()
        out = torch.empty((4, c), dtype=torch.float32, device=partials.device)
        d.partials, d.tiles, d.pitch, d.fold, d.C, d.count = _f32(partials), tiles, pitch, 1, c, float(count)
        d.gamma, d.beta, d.running_mean, d.running_var = _f32(gamma), _f32(beta), _f32(rm), _f32(rv)
        d.momentum, d.eps = float(momentum), float(eps)
        d.scale, d.shift, d.mean, d.rstd = out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), out[3].data_ptr()
        outs.append((out[0], out[1], out[2], out[3]))
    check(_lib.load().mcgen_bn_finalize_batch(arr, len(items), _stream()), 'bn_finalize_batch')
    return outs


def bn_finalize_par(partials: Tensor, count: int, gamma: Tensor, beta: Tensor, eps: float = 1e-5, fold: int = 1, groups: int = 1):
    """bn_finalize with the statistics groups in parallel and the running statistics left alone
    -> (scale, shift, mean, rstd, unbiased var), each [groups, C]; pair with bn_running_batch at the end of the pass."""
    tiles, _, pitch = partials.shape
    c = gamma.numel()
    out = torch.empty((5, groups, c), dtype=torch.float32, device=partials.device)
    check(_lib.load().mcgen_bn_finalize_par(_f32(partials), tiles, pitch, fold, c, float(count), groups, _f32(gamma), _f32(beta), eps,
                                            out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), out[3].data_ptr(), out[4].data_ptr(),
                                            _stream()), 'bn_finalize_par')
    return out[0], out[1], out[2], out[3], out[4]


def bn_running_batch(items):
    """items = [(running_mean, running_var, mean [groups, C], unb [groups, C], momentum)]: every layer's running statistics
    take its groups' updates in order, one launch (mcgen_bn_running_batch)."""
    arr = (_lib.BnRun * len(items))()
    for d, (rm, rv, mean, unb, mom) in zip(arr, items):
        d.running_mean, d.running_var, d.mean, d.unb = _f32(rm), _f32(rv), _f32(mean), _f32(unb)
        d.groups, d.C, d.momentum = mean.shape[0], rm.numel(), float(mom)
    check(_lib.load().mcgen_bn_running_batch(arr, len(items), _stream()), 'bn_running_batch')


def bn_eval_affine(gamma, beta, running_mean, running_var, eps: float = 1e-5):
    c = gamma.numel()
    out = torch.empty((2, c), dtype=torch.float32, device=gamma.device)
    check(_lib.load().mcgen_bn_eval_affine(_f32(gamma), _f32(beta), _f32(running_mean), _f32(running_var), eps, c,
                                           out[0].data_ptr(), out[1].data_ptr(), _stream()), 'bn_eval_affine')
    return out[0], out[1]


def bn_backward(partials: Tensor, dz: Tensor, x: Tensor, count: int, scale, mean, rstd,
                dgamma: Optional[Tensor], dbeta: Optional[Tensor], add: Optional[Tensor] = None,
                accumulate: bool = False) -> Tensor:
    """BN backward from the conv epilogue's (sum dz, sum dz*xhat) partials."""
    tiles, _, pitch = partials.shape
    c = scale.numel()
    sums = torch.empty((2, c), dtype=torch.float32, device=dz.device)
    lib = _lib.load()
    check(lib.mcgen_bn_bwd_finalize(_f32(partials), tiles, pitch, c, _f32(dgamma), _f32(dbeta), _f32(sums),
                                    int(accumulate), _stream()), 'bn_bwd_finalize')
    dx = torch.empty_like(dz)
    assert dz.shape[-1] == c and x.shape == dz.shape
    check(lib.mcgen_bn_bwd_apply(_p(dz), _p(x), _p(add), _p(dx), _dt(dz.dtype), dz.numel() // c, c, _f32(sums),
                                 float(count), _f32(scale), _f32(mean), _f32(rstd), _stream()), 'bn_bwd_apply')
    return dx


_colsum_ws = {}


def colsum(x: Tensor, c: int, out: Tensor, alpha: float = 1.0, accumulate: bool = False, row_perm: int = 1):
    """out[c] (+)= alpha * sum over all leading dims of x[..., :c]."""
    pitch = x.shape[-1]
    rows = x.numel() // pitch
    ws = torch.empty(256 * c, dtype=torch.float32, device=x.device)
    check(_lib.load().mcgen_colsum(_p(x), _dt(x.dtype), rows, c, pitch, _f32(out), row_perm, float(alpha),
                                   int(accumulate), _f32(ws), _stream()), 'colsum')


def tanh_bwd(dy: Tensor, y: Tensor) -> Tensor:
    dx = torch.empty_like(dy)
    check(_lib.load().mcgen_tanh_bwd(_p(dy), _p(y), _p(dx), _dt(dy.dtype), dy.numel(), _stream()), 'tanh_bwd')
    return dx


def dtail_fwd(x: Tensor, code: Optional[Tensor], w: Tensor, b: Tensor, sigma: Tensor):
    n, c = x.shape[0], x.shape[-1]
    hw = x.numel() // (n * c)
    pooled = torch.empty((n, c), dtype=torch.float32, device=x.device)
    logit = torch.empty((n,), dtype=torch.float32, device=x.device)
    check(_lib.load().mcgen_dtail_fwd(_p(x), _dt(x.dtype), _f32(code), _f32(w), _f32(b), _f32(sigma), _f32(pooled),
                                      _f32(logit), n, hw, c, _stream()), 'dtail_fwd')
    return logit, pooled


def dtail_bwd(dlogit: Tensor, x: Tensor, code, w, sigma, pooled, dw: Optional[Tensor], db: Optional[Tensor],
              accumulate: bool = False) -> Tensor:
    n, c = x.shape[0], x.shape[-1]
    hw = x.numel() // (n * c)
    dx = torch.empty_like(x)
    check(_lib.load().mcgen_dtail_bwd(_f32(dlogit), _p(x), _dt(x.dtype), _f32(code), _f32(w), _f32(sigma), _f32(pooled),
                                      _p(dx), _f32(dw), _f32(db), n, hw, c, int(accumulate), _stream()), 'dtail_bwd')
    return dx


def dtail_hinge_ok(x: Tensor) -> bool:
    c = x.shape[-1]
    return c % 8 == 0 and c <= 2048


def dtail_hinge_fused(x: Tensor, code: Optional[Tensor], w: Tensor, b: Tensor, sigma: Tensor, mode: str):
    """Tail forward + d(hinge loss)/d(logit) + tail input gradient in one launch (mcgen_dtail_hinge_fused).
    mode 'd_pair': x is a paired batch (real half, generated half), hinge_d; 'g': hinge_g.
    -> (logit [N], pooled [N, C], dlogit [N], dx like x)"""
    n, c = x.shape[0], x.shape[-1]
    hw = x.numel() // (n * c)
    pooled = torch.empty((n, c), dtype=torch.float32, device=x.device)
    both = torch.empty((2, n), dtype=torch.float32, device=x.device)
    dx = torch.empty_like(x)
    check(_lib.load().mcgen_dtail_hinge_fused(_p(x), _dt(x.dtype), _f32(code), _f32(w), _f32(b), _f32(sigma), _f32(pooled),
                                              both[0].data_ptr(), both[1].data_ptr(), _p(dx), n, hw, c,
                                              {'d_pair': 0, 'g': 1}[mode], _stream()), 'dtail_hinge_fused')
    return both[0], pooled, both[1], dx


def dtail_pair_wgrad(dlogit: Tensor, pooled: Tensor, ratio: Tensor, dw1: Tensor, db1: Tensor, dw2: Tensor, db2: Tensor,
                     logit: Optional[Tensor] = None, loss: Optional[Tensor] = None):
    """Tail weight / bias gradients of the two halves of a paired discriminator pass (dw2 divided by ratio[0]); with
    `logit` [2N] and `loss` [1] the launch also writes the discriminator's hinge loss (train_gan.py:154)."""
    n2, c = pooled.shape
    for t in (dw1, db1, dw2, db2):
        if not t.is_contiguous() or t.dtype != torch.float32:
            raise _lib.McgenError('dtail_pair_wgrad: outputs must be contiguous fp32')
    if loss is not None:
        check(_lib.load().mcgen_dtail_pair_wgrad_loss(_f32(dlogit), _f32(pooled), _f32(ratio), _f32(logit.contiguous()), n2 // 2, c,
                                                      _f32(dw1), _f32(db1), _f32(dw2), _f32(db2), _f32(loss), _stream()),
              'dtail_pair_wgrad_loss')
        return
    check(_lib.load().mcgen_dtail_pair_wgrad(_f32(dlogit), _f32(pooled), _f32(ratio), n2 // 2, c,
                                             _f32(dw1), _f32(db1), _f32(dw2), _f32(db2), _stream()), 'dtail_pair_wgrad')


def hinge_d(real: Tensor, fake: Tensor, both: bool = False):
    """(loss, d loss / d real, d loss / d fake) of the discriminator hinge loss; `both`: also the two gradients as one
    contiguous [2N] tensor (they are views of it)."""
    n = real.numel()
    out = torch.empty(1 + 2 * n, dtype=torch.float32, device=real.device)
    check(_lib.load().mcgen_hinge_d(_f32(real), _f32(fake), n, out.data_ptr(), out[1:1 + n].data_ptr(),
                                    out[1 + n:].data_ptr(), _stream()), 'hinge_d')
    if both:
        return out[0], out[1:1 + n], out[1 + n:], out[1:]
    return out[0], out[1:1 + n], out[1 + n:]


def hinge_g(fake: Tensor):
    n = fake.numel()
    out = torch.empty(1 + n, dtype=torch.float32, device=fake.device)
    check(_lib.load().mcgen_hinge_g(_f32(fake), n, out.data_ptr(), out[1:].data_ptr(), _stream()), 'hinge_g')
    return out[0], out[1:]


def sn_layers_tensor(layers, device) -> Tensor:
    """Pack [(w_off, u_off, v_off, rows, cols)] into a device byte tensor of mcgen_sn_layer_t."""
    arr = (_lib.SnLayer * len(layers))()
    for i, (wo, uo, vo, r, c) in enumerate(layers):
        arr[i].w_off, arr[i].u_off, arr[i].v_off, arr[i].rows, arr[i].cols = wo, uo, vo, r, c
    raw = bytes(arr)
    return torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)


def sn_power_iter(w_base: Tensor, uv_base: Tensor, layers_dev: Tensor, nlayers: int, do_iter: bool, sigma: Tensor,
                  max_rows: int, max_cols: int, snap: Optional[Tensor] = None):
    """`snap` (training mode): a buffer shaped like uv_base that receives the new u, v as well (the forward's own copy)."""
    ws = torch.empty(nlayers * (32 * max_cols + max_rows), dtype=torch.float32, device=w_base.device)
    if snap is not None:
        if not do_iter or snap.numel() != uv_base.numel() or snap.dtype != torch.float32:
            raise _lib.McgenError('sn_power_iter: snap needs a training-mode iteration and a float32 buffer like uv_base')
        check(_lib.load().mcgen_sn_power_iter_snap(_f32(w_base), _f32(uv_base), _p(layers_dev), nlayers, _f32(sigma), _f32(ws),
                                                   max_rows, max_cols, _f32(snap), _stream()), 'sn_power_iter_snap')
        return
    check(_lib.load().mcgen_sn_power_iter(_f32(w_base), _f32(uv_base), _p(layers_dev), nlayers, int(do_iter),
                                          _f32(sigma), _f32(ws), max_rows, max_cols, _stream()), 'sn_power_iter')


def sn_power_iter_rounds(w_base: Tensor, uv_base: Tensor, layers_dev: Tensor, nlayers: int, rounds: int,
                         max_rows: int, max_cols: int, snapshot: bool = True):
    """`rounds` training-mode power iterations in 2 * rounds + 1 launches (mcgen_sn_power_iter_rounds)
    -> (sigma [rounds + 1, nlayers], u/v snapshots [rounds, uv] or None); for rounds >= 2 the extra last row of `sigma` is
    sigma[rounds - 2] / sigma[rounds - 1] (the ratio a paired pass scales its second half by)."""
    sigma = torch.empty((rounds + 1, nlayers), dtype=torch.float32, device=w_base.device)
    snap = torch.empty((rounds, uv_base.numel()), dtype=torch.float32, device=w_base.device) if snapshot else None
    ws = torch.empty(nlayers * (32 * max_cols + max_rows), dtype=torch.float32, device=w_base.device)
    check(_lib.load().mcgen_sn_power_iter_rounds(_f32(w_base), _f32(uv_base), _p(layers_dev), nlayers, rounds, _f32(sigma), _f32(ws),
                                                 max_rows, max_cols, _f32(snap), uv_base.numel(), sigma[rounds].data_ptr(), _stream()),
          'sn_power_iter_rounds')
    return sigma, snap


def sn_power_iter_fused(w_base: Tensor, uv_base: Tensor, layers_dev: Tensor, nlayers: int, rounds: int, do_iter: bool,
                        max_rows: int, max_cols: int, snapshot: bool = True):
    """`rounds` power iterations of all layers in one launch -> (sigma [rounds, nlayers], u/v snapshots [rounds, uv] or None)."""
    sigma = torch.empty((rounds, nlayers), dtype=torch.float32, device=w_base.device)
    snap = torch.empty((rounds, uv_base.numel()), dtype=torch.float32, device=w_base.device) if snapshot else None
    check(_lib.load().mcgen_sn_power_iter_fused(_f32(w_base), _f32(uv_base), _p(layers_dev), nlayers, rounds, int(do_iter),
                                                _f32(sigma), _f32(snap), uv_base.numel(), max_rows, max_cols, _stream()),
          'sn_power_iter_fused')
    return sigma, snap


def sn_grad_fix(g_src: Tensor, g_dst: Tensor, w_base: Tensor, uv_base: Tensor, layers_dev: Tensor, nlayers: int,
                sigma: Tensor, accumulate: bool = False):
    ws = torch.empty(32 * nlayers, dtype=torch.float32, device=g_src.device)
    check(_lib.load().mcgen_sn_grad_fix(_f32(g_src), _f32(g_dst), _f32(w_base), _f32(uv_base), _p(layers_dev), nlayers,
                                        _f32(sigma), int(accumulate), _f32(ws), _stream()), 'sn_grad_fix')


def sn_grad_fix_pair(g_src0: Tensor, g_src1: Tensor, g_dst: Tensor, w_base: Tensor, uv0: Tensor, uv1: Tensor,
                     layers_dev: Tensor, nlayers: int, sigma0: Tensor, sigma1: Tensor, accumulate: bool = False):
    """Both halves of a paired discriminator pass in one dot + one apply launch (mcgen_sn_grad_fix_pair)."""
    ws = torch.empty(2 * 32 * nlayers, dtype=torch.float32, device=g_dst.device)
    check(_lib.load().mcgen_sn_grad_fix_pair(_f32(g_src0), _f32(g_src1), _f32(g_dst), _f32(w_base), _f32(uv0), _f32(uv1),
                                             _p(layers_dev), nlayers, _f32(sigma0), _f32(sigma1), int(accumulate), _f32(ws),
                                             _stream()), 'sn_grad_fix_pair')


def sn_fix_pair_adam(g_src0: Tensor, g_src1: Tensor, p: Tensor, m: Tensor, v: Tensor, uv0: Tensor, uv1: Tensor,
                     layers_dev: Tensor, nlayers: int, sigma0: Tensor, sigma1: Tensor, step: Tensor, lr, betas,
                     eps: float, weight_decay: float, advance_step: bool):
    """mcgen_sn_fix_pair_adam: the spectral-norm gradient fix of both halves of a paired discriminator pass fused with
    Adam's update of the same layers (`step`: the int64[2] {counter, ticket} buffer of ops.adam; `lr` as in ops.adam)."""
    assert step.dtype == torch.int64 and step.numel() == 2
    ws = torch.empty(2 * 32 * nlayers, dtype=torch.float32, device=p.device)
    lr_f, lr_dev = _lr_args(lr)
    check(_lib.load().mcgen_sn_fix_pair_adam(_f32(g_src0), _f32(g_src1), _f32(p), _f32(m), _f32(v), _f32(uv0), _f32(uv1),
                                             _p(layers_dev), nlayers, _f32(sigma0), _f32(sigma1), _f32(ws), lr_f, lr_dev, betas[0], betas[1],
                                             eps, weight_decay, _p(step), int(advance_step), _stream()), 'sn_fix_pair_adam')


def _lr_args(lr):
    """(float, device pointer or None): a one-element fp32 device tensor is handed over as lr_dev (read when the kernel
    runs, so captured launches follow a scheduler), a Python number as the immediate."""
    if isinstance(lr, Tensor):
        assert lr.dtype == torch.float32 and lr.numel() == 1 and lr.is_cuda
        return 0.0, _f32(lr)
    return float(lr), None


def adam(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: Tensor, lr, betas=(0.9, 0.999),
         eps: float = 1e-8, weight_decay: float = 0.0):
    """`step`: int64[2] on the device -- [0] the step counter (incremented by the launch), [1] a ticket word (0 between calls).
    `lr`: a Python number, or a one-element fp32 device tensor read at execution time (mcgen_adam's lr_dev)."""
    assert step.dtype == torch.int64 and step.numel() == 2
    lr_f, lr_dev = _lr_args(lr)
    check(_lib.load().mcgen_adam(_f32(p), _f32(g), _f32(m), _f32(v), p.numel(), lr_f, lr_dev, betas[0], betas[1], eps,
                                 weight_decay, _p(step), _stream()), 'adam')


# ---- batched small launches (one launch per network pass instead of one per layer) ----------------------
def _struct_table(arr, device) -> Tensor:
    return torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device)


class PrepBatch:
    """A fixed list of weight-image jobs (master weight -> persistent image buffer) run as ONE launch.
    jobs: (weight tensor, image tensor, transpose, row_perm, sigma_idx or -1, wscale)."""

    def __init__(self, jobs, dtype: torch.dtype):
        self.dtype = dtype
        self.jobs = jobs
        arr = (_lib.Prep * len(jobs))()
        for d, job in zip(arr, jobs):
            w, img, transpose, row_perm, sidx, wscale = job[:6]
            kmajor = bool(job[6]) if len(job) > 6 else False          # K-major image (mode-compacted launches)
            kmap, kcount = (job[7], int(job[8])) if len(job) > 8 and job[7] is not None else (None, 0)
            rmap = job[9] if len(job) > 9 else None                   # image row r <- weight row rmap[r] (mcgen_conv_t.yperm)
            cout, cin = w.shape[0], w.shape[1]
            ks = w.shape[2] if w.dim() == 4 else 1
            if kmap is not None:                                      # a mode's compacted image: input channel k <- kmap[k], k < kcount
                assert not kmajor and not transpose and row_perm == 1 and kmap.dtype == torch.int16 and kmap.numel() >= kcount
                want = weight_image_elems(cout, kcount, ks, False)
            else:
                want = weight_image_k_elems(cout, cin, ks) if kmajor else weight_image_elems(cout, cin, ks, transpose)
            assert img.numel() == want and img.dtype == dtype and not (kmajor and (transpose or row_perm != 1))
            d.w, d.image = _f32(w), _p(img)
            d.Cout, d.Cin, d.ksize, d.transpose, d.row_perm, d.sigma_idx = cout, cin, ks, int(transpose), row_perm, sidx
            d.wscale, d.layout = float(wscale), int(kmajor)
            d.kmap, d.kcount = _p(kmap), kcount
            if rmap is not None:
                assert not kmajor and not transpose and row_perm == 1 and rmap.dtype == torch.int16 and rmap.numel() >= cout and rmap.is_contiguous()
            d.rmap = _p(rmap)
        self.key = tuple((w.data_ptr(), img.data_ptr()) for w, img, *_ in jobs)
        self.table = _struct_table(arr, jobs[0][0].device)
        self.n = len(jobs)

    def valid(self) -> bool:
        return self.key == tuple((w.data_ptr(), img.data_ptr()) for w, img, *_ in self.jobs)

    def run(self, sigma: Optional[Tensor] = None):
        check(_lib.load().mcgen_prep_weight_batch(_p(self.table), self.n, _f32(sigma), _dt(self.dtype), _stream()),
              'prep_weight_batch')


def prep_and_codes(prep: 'PrepBatch', sigma: Optional[Tensor], codes: 'CodeBatch', label: Tensor, reps: int = 1,
                   scale: Optional[Tensor] = None, n_half: int = 0):
    """PrepBatch.run(sigma) and CodeBatch.run_labels(label, reps, scale, n_half) as ONE launch (mcgen_prep_weight_batch_codes):
    a discriminator pass's weight images and codes both wait for the power iteration only.  -> the code tensors."""
    codes._ensure()
    n = label.shape[0] * reps
    codes._tables_for(n, label.device)
    if label.dtype != torch.int64 or any(d.C % 4 for d in codes._arr):
        raise _lib.McgenError('prep_and_codes: int64 labels and channel counts that are multiples of 4')
    buf = torch.empty(codes._total, dtype=torch.float32, device=label.device)
    check(_lib.load().mcgen_prep_weight_batch_codes(_p(prep.table), prep.n, _f32(sigma), _dt(prep.dtype), _p(label.contiguous()),
                                                    label.shape[0], _p(codes._table), len(codes.mcs), _f32(buf), n, _f32(scale), n_half,
                                                    _stream()), 'prep_weight_batch_codes')
    out, off = [], 0
    for d in codes._arr:
        out.append(buf[off:off + n * d.C].view(n, d.C))
        off += n * d.C
    return out


class CodeBatch:
    """Codes of a list of MultimodalController modules in ONE launch; rebuilt when a codebook buffer was
    re-registered (models.utils.create / transit) or moved."""

    def __init__(self, mcs, scale_idx=None):
        """`mcs`: modules with a `codebook` buffer; `scale_idx[i]` (optional): index into the `scale` vector given to
        run() whose entry multiplies the rows n >= n_half of job i (-1: none)."""
        self.mcs = list(mcs)
        self.scale_idx = list(scale_idx) if scale_idx is not None else [-1] * len(self.mcs)
        self._key = None

    def _ensure(self):
        key = tuple((m.codebook.data_ptr(), tuple(m.codebook.shape)) for m in self.mcs)
        if key != self._key:
            arr = (_lib.Code * len(self.mcs))()
            off = 0
            self.offsets = []
            for d, m in zip(arr, self.mcs):
                cb = m.codebook
                if cb.dtype != torch.float32 or not cb.is_contiguous():
                    raise _lib.McgenError('codebook must be a contiguous float32 buffer')
                d.codebook, d.out_off, d.M, d.C = _p(cb), off, cb.shape[0], cb.shape[1]
                d.scale_idx = self.scale_idx[len(self.offsets)]
                self.offsets.append((off, cb.shape[1]))
                off += 0                                   # per-forward: offsets scale with N, filled in run()
            self._arr = arr
            self._key = key
            self._n_cached = None

    def run_labels(self, label: Tensor, reps: int = 1, scale: Optional[Tensor] = None, n_half: int = 0):
        """The same for one-hot indicators given as int64 labels: code_i = codebook_i[label] (a row gather per module, all in
        one launch: mcgen_mc_gather_batch).  `reps`: the batch is the label vector `reps` times back to back; `scale` /
        `n_half` as in run()."""
        self._ensure()
        n = label.shape[0] * reps
        self._tables_for(n, label.device)
        if label.dtype != torch.int64 or any(d.C % 4 for d in self._arr):
            raise _lib.McgenError('run_labels: int64 labels and channel counts that are multiples of 4')
        buf = torch.empty(self._total, dtype=torch.float32, device=label.device)
        check(_lib.load().mcgen_mc_gather_batch(_p(label.contiguous()), label.shape[0], _p(self._table), len(self.mcs), _f32(buf), n,
                                                _f32(scale), n_half, _stream()), 'mc_gather_batch')
        out, off = [], 0
        for d in self._arr:
            out.append(buf[off:off + n * d.C].view(n, d.C))
            off += n * d.C
        return out

    def labels_of(self, indicator: Tensor):
        """(label, reps) when `indicator` carries its labels (`onehot_hint`) and run_labels applies, else None."""
        hint = getattr(indicator, '_mcgen_onehot', None)
        if hint is not None and hint[0].shape[0] * hint[1] == indicator.shape[0] and hint[0].is_cuda and hint[0].dtype == torch.int64 \
                and all(d.C % 4 == 0 for d in (self._ensure() or self._arr)):
            return hint[0], hint[1]
        return None

    def run_any(self, indicator: Tensor, scale: Optional[Tensor] = None, n_half: int = 0):
        """run(), or run_labels() when the indicator carries its labels (`onehot_hint`): the caller built it with
        F.one_hot / ops.onehot_rep, so a row gather gives indicator @ codebook exactly."""
        hint = getattr(indicator, '_mcgen_onehot', None)
        if hint is not None and hint[0].shape[0] * hint[1] == indicator.shape[0] and hint[0].is_cuda \
                and all(d.C % 4 == 0 for d in (self._ensure() or self._arr)):
            return self.run_labels(hint[0], hint[1], scale, n_half)
        return self.run(indicator, scale, n_half)

    def _tables_for(self, n: int, device):
        if self._n_cached != n:
            # one descriptor table per batch size (the paired discriminator pass alternates 2N and N)
            if not isinstance(getattr(self, '_tables', None), dict) or self._tables.get('key') != self._key:
                self._tables = {'key': self._key}
            if n not in self._tables:
                off = 0
                for d in self._arr:
                    d.out_off = off
                    off += n * d.C
                self._tables[n] = (off, _struct_table(self._arr, device))
            self._total, self._table = self._tables[n]
            self._n_cached = n

    def run(self, indicator: Tensor, scale: Optional[Tensor] = None, n_half: int = 0):
        """-> list of [N, C] code tensors (views of one buffer), in module order."""
        self._ensure()
        n = indicator.shape[0]
        self._tables_for(n, indicator.device)
        if indicator.shape[1] != self._arr[0].M:
            raise _lib.McgenError(f'indicator has {indicator.shape[1]} modes, codebook has {self._arr[0].M}')
        buf = torch.empty(self._total, dtype=torch.float32, device=indicator.device)
        check(_lib.load().mcgen_mc_code_batch(_f32(indicator.contiguous()), _p(self._table), len(self.mcs), _f32(buf), n,
                                              _f32(scale), n_half, _stream()), 'mc_code_batch')
        out, off = [], 0
        for d in self._arr:
            out.append(buf[off:off + n * d.C].view(n, d.C))
            off += n * d.C
        return out


# ---- MCGlow kernels ---------------------------------------------------------------------------------------------
def glow_squeeze(x: Tensor, c: int) -> Tensor:
    """[N,H,W,Cp] (c logical channels) -> [N,H/2,W/2,pad8(4c)]  (mcglow.py:221-223)."""
    n, h, w, cp = x.shape
    y = torch.empty((n, h // 2, w // 2, pad8(4 * c)), dtype=x.dtype, device=x.device)
    check(_lib.load().mcgen_glow_squeeze(_p(x), _p(y), _dt(x.dtype), n, h, w, c, cp, y.shape[-1], 0, _stream()), 'glow_squeeze')
    return y


def glow_unsqueeze(x: Tensor, c4: int) -> Tensor:
    """[N,H,W,Cp] with c4 = 4c logical channels -> [N,2H,2W,pad8(c)]  (mcglow.py:262-265)."""
    n, h, w, cp = x.shape
    c = c4 // 4
    y = torch.empty((n, 2 * h, 2 * w, pad8(c)), dtype=x.dtype, device=x.device)
    check(_lib.load().mcgen_glow_squeeze(_p(x), _p(y), _dt(x.dtype), n, 2 * h, 2 * w, c, y.shape[-1], cp, 1, _stream()), 'glow_unsqueeze')
    return y


def channel_stats(x: Tensor, blocks: int = 64) -> Tensor:
    cp = x.shape[-1]
    pixels = x.numel() // cp
    blocks = max(1, min(blocks, pixels))
    part = torch.empty((blocks, 2, cp), dtype=torch.float32, device=x.device)
    check(_lib.load().mcgen_channel_stats(_p(x), _dt(x.dtype), pixels, cp, _f32(part), blocks, _stream()), 'channel_stats')
    return part


def actnorm_init(partials: Tensor, count: int, loc: Tensor, scale: Tensor):
    """ActNorm.initialize (mcglow.py:32-39) into the parameter tensors loc / scale ([1,C,1,1])."""
    tiles, _, pitch = partials.shape
    check(_lib.load().mcgen_actnorm_init(_f32(partials), tiles, pitch, loc.numel(), float(count), _f32(loc), _f32(scale),
                                         _stream()), 'actnorm_init')


def actnorm_affine(loc: Tensor, scale: Tensor, cp: int, with_negloc: bool = False):
    """Prologue vectors of the op that consumes an ActNorm: y = x * a + b with a = scale, b = scale * loc (zero-padded to
    cp); with_negloc also returns -loc (the gate mean of the backward pass)."""
    c = loc.numel()
    ab = torch.empty((3 if with_negloc else 2, cp), dtype=torch.float32, device=loc.device)
    check(_lib.load().mcgen_actnorm_affine(_f32(loc), _f32(scale), c, cp, ab[0].data_ptr(), ab[1].data_ptr(),
                                           ab[2].data_ptr() if with_negloc else None, _stream()), 'actnorm_affine')
    return (ab[0], ab[1], ab[2]) if with_negloc else (ab[0], ab[1])


def glow_param_logdet(scale: Tensor, w_s: Tensor, hw: int, logdet: Tensor):
    check(_lib.load().mcgen_glow_param_logdet(_f32(scale), scale.numel(), _f32(w_s), w_s.numel(), float(hw), _f32(logdet),
                                              logdet.numel(), _stream()), 'glow_param_logdet')


def invconv_weight(w_p, w_l, w_u, w_s, s_sign, inverse: bool = False):
    c = w_s.numel()
    w = torch.empty((c, c), dtype=torch.float32, device=w_s.device)
    winv = torch.empty_like(w) if inverse else None
    check(_lib.load().mcgen_invconv_weight(_f32(w_p), _f32(w_l), _f32(w_u), _f32(w_s), _f32(s_sign), c, _f32(w), _f32(winv),
                                           _stream()), 'invconv_weight')
    return w, winv


# ---- batched MCGlow per-module ops (one launch per kind for all modules of a pass) --------------------------------------
def actnorm_affine_batch(ans, cps):
    """[(a, b, negloc)] for ActNorm modules `ans` (loc / scale parameters) padded to `cps[i]` channels: ONE buffer, one launch
    per MCGEN_GLOW_BATCH_MAX modules (mcgen_actnorm_affine_batch)."""
    dev = ans[0].loc.device
    buf = torch.empty(3 * sum(cps), dtype=torch.float32, device=dev)
    arr = (_lib.AnAffine * len(ans))()
    out, off = [], 0
    for d, an, cp in zip(arr, ans, cps):
        a, b, nl = buf[off:off + cp], buf[off + cp:off + 2 * cp], buf[off + 2 * cp:off + 3 * cp]
        off += 3 * cp
        d.loc, d.scale, d.a, d.b, d.negloc, d.C, d.Cp = _f32(an.loc.data), _f32(an.scale.data), _f32(a), _f32(b), _f32(nl), an.loc.numel(), cp
        out.append((a, b, nl))
    check(_lib.load().mcgen_actnorm_affine_batch(arr, len(ans), _stream()), 'actnorm_affine_batch')
    return out


def glow_param_logdet_batch(items, logdet: Tensor):
    """logdet[n] += sum over `items` = [(actnorm scale, w_s, H * W)] of HW * (sum log|scale| + sum w_s), one launch."""
    arr = (_lib.Pld * len(items))()
    for d, (scale, w_s, hw) in zip(arr, items):
        d.scale, d.w_s, d.C, d.Cw, d.hw = _f32(scale), _f32(w_s), scale.numel(), w_s.numel(), float(hw)
    check(_lib.load().mcgen_glow_param_logdet_batch(arr, len(items), _f32(logdet), logdet.numel(), _stream()), 'glow_param_logdet_batch')


def invconv_weight_batch(ics):
    """[W] of InvConv2dLU modules `ics` (mcglow.py:105-111), one launch per MCGEN_GLOW_BATCH_MAX modules."""
    arr = (_lib.Icw * len(ics))()
    out = []
    for d, ic in zip(arr, ics):
        c = ic.w_s.numel()
        w = torch.empty((c, c), dtype=torch.float32, device=ic.w_s.device)
        d.w_p, d.w_l, d.w_u, d.w_s, d.s_sign, d.weight, d.weight_inv, d.C = (_f32(ic.w_p), _f32(ic.w_l.data), _f32(ic.w_u.data),
                                                                            _f32(ic.w_s.data), _f32(ic.s_sign), _f32(w), None, c)
        out.append(w)
    check(_lib.load().mcgen_invconv_weight_batch(arr, len(ics), _stream()), 'invconv_weight_batch')
    return out


class GlowDeferred:
    """Parameter-gradient reductions of an MCGlow backward pass that nothing later in the pass reads (ActNorm loc / scale from
    the dgrad epilogues' partial sums, ZeroConv2d scale, the LU parameters): queued while the pass runs, launched batched at
    its end -- they were ~250 launches of 3-6 us."""

    def __init__(self):
        self.an, self.pcs, self.icb, self.keep = [], [], [], []

    def actnorm_bwd(self, partials, scale, ld_coef, input_side, dloc, dscale, accumulate=False):
        self.an.append((partials, scale, float(ld_coef), int(input_side), dloc, dscale, int(accumulate)))

    def prod_colsum(self, a, b, c, out, alpha=1.0, accumulate=False):
        self.pcs.append((a, b, c, out, float(alpha), int(accumulate)))

    def invconv_bwd(self, ic, dW, ld_coef, dw_l, dw_u, dw_s, accumulate=False):
        self.icb.append((ic, dW, float(ld_coef), dw_l, dw_u, dw_s, int(accumulate)))

    def run(self):
        lib = _lib.load()
        if self.an:
            arr = (_lib.AnBwd * len(self.an))()
            for d, (p, sc, ld, side, dl, ds, acc) in zip(arr, self.an):
                tiles, _, pitch = p.shape
                d.partials, d.scale, d.dloc, d.dscale = _f32(p), _f32(sc), _f32(dl), _f32(ds)
                d.tiles, d.pitch, d.C, d.input_side, d.accumulate, d.ld_coef = tiles, pitch, sc.numel(), side, acc, ld
            check(lib.mcgen_actnorm_bwd_batch(arr, len(self.an), _stream()), 'actnorm_bwd_batch')
        if self.pcs:
            dt = self.pcs[0][0].dtype
            arr = (_lib.Pcs * len(self.pcs))()
            cmax = max(j[2] for j in self.pcs)
            for d, (a, b, c, out, alpha, acc) in zip(arr, self.pcs):
                assert a.dtype == dt and b.dtype == dt
                d.a, d.b, d.out, d.pixels = _p(a), _p(b), _f32(out), a.numel() // a.shape[-1]
                d.pitch_a, d.pitch_b, d.C, d.accumulate, d.alpha = a.shape[-1], b.shape[-1], c, acc, alpha
            ws = torch.empty(len(self.pcs) * 256 * cmax, dtype=torch.float32, device=self.pcs[0][0].device)
            check(lib.mcgen_prod_colsum_batch(arr, len(self.pcs), _dt(dt), _f32(ws), _stream()), 'prod_colsum_batch')
        if self.icb:
            arr = (_lib.Icb * len(self.icb))()
            for d, (ic, dW, ld, dl, du, dsg, acc) in zip(arr, self.icb):
                d.w_p, d.w_l, d.w_u, d.w_s, d.s_sign = _f32(ic.w_p), _f32(ic.w_l.data), _f32(ic.w_u.data), _f32(ic.w_s.data), _f32(ic.s_sign)
                d.dW, d.dw_l, d.dw_u, d.dw_s = _f32(dW), _f32(dl), _f32(du), _f32(dsg)
                d.C, d.ldw, d.accumulate, d.ld_coef = ic.w_s.numel(), dW.shape[-1], acc, ld
            check(lib.mcgen_invconv_bwd_batch(arr, len(self.icb), _stream()), 'invconv_bwd_batch')
        self.an, self.pcs, self.icb = [], [], []


def prep_weight_rows(w: Tensor, dtype: torch.dtype, row_scale: Tensor) -> Tensor:
    cout, cin = w.shape[0], w.shape[1]
    ks = w.shape[2] if w.dim() == 4 else 1
    out = torch.empty(weight_image_elems(cout, cin, ks, False), dtype=dtype, device=w.device)
    check(_lib.load().mcgen_prep_weight_rows(_f32(w.contiguous()), _p(out), _dt(dtype), cout, cin, ks, _f32(row_scale), _stream()),
          'prep_weight_rows')
    return out


def glow_coupling(x: Tensor, h: Tensor, c: int, logdet: Optional[Tensor], reverse: bool = False, accumulate: bool = True) -> Tensor:
    n, cp = x.shape[0], x.shape[-1]
    hw = x.numel() // (n * cp)
    assert h.shape == x.shape and h.dtype == x.dtype
    y = torch.empty_like(x)
    check(_lib.load().mcgen_glow_coupling(_p(x), _p(h), _p(y), _dt(x.dtype), _f32(logdet), n, hw, c, cp, int(reverse),
                                          int(accumulate), _stream()), 'glow_coupling')
    return y


def gaussian_logp(z: Tensor, c0: int, prior: Tensor, cz: int, logp: Tensor, accumulate: bool = True):
    n = z.shape[0]
    hw = z.numel() // (n * z.shape[-1])
    check(_lib.load().mcgen_gaussian_logp(_p(z), z.shape[-1], c0, _p(prior), prior.shape[-1], _dt(z.dtype), n, hw, cz,
                                          _f32(logp), int(accumulate), _stream()), 'gaussian_logp')


def gaussian_sample(eps: Tensor, prior: Tensor, out: Tensor, c0: int, cz: int):
    pixels = eps.numel() // eps.shape[-1]
    check(_lib.load().mcgen_gaussian_sample(_p(eps), eps.shape[-1], _p(prior), prior.shape[-1], _p(out), out.shape[-1], c0,
                                            _dt(eps.dtype), pixels, cz, _stream()), 'gaussian_sample')


def copy_channels(src: Tensor, s0: int, dst: Tensor, c0: int, cn: int):
    pixels = src.numel() // src.shape[-1]
    check(_lib.load().mcgen_copy_channels(_p(src), src.shape[-1], s0, _p(dst), dst.shape[-1], c0, _dt(src.dtype), pixels, cn,
                                          _stream()), 'copy_channels')


def glow_coupling_bwd(v: Tensor, h: Tensor, dy: Tensor, c: int, g: float):
    dv, dh = torch.empty_like(v), torch.empty_like(v)
    pixels = v.numel() // v.shape[-1]
    check(_lib.load().mcgen_glow_coupling_bwd(_p(v), _p(h), _p(dy), _p(dv), _p(dh), _dt(v.dtype), float(g), pixels, c,
                                              v.shape[-1], _stream()), 'glow_coupling_bwd')
    return dv, dh


def gaussian_logp_bwd(z: Tensor, c0: int, prior: Tensor, cz: int, dz: Tensor, d0: int, g: float, accumulate_dz: bool):
    dprior = torch.zeros_like(prior)
    pixels = z.numel() // z.shape[-1]
    check(_lib.load().mcgen_gaussian_logp_bwd(_p(z), z.shape[-1], c0, _p(prior), prior.shape[-1], _p(dz), dz.shape[-1], d0,
                                              _p(dprior), _dt(z.dtype), float(g), pixels, cz, int(accumulate_dz), _stream()),
          'gaussian_logp_bwd')
    return dprior


def prod_colsum(a: Tensor, b: Tensor, c: int, out: Tensor, alpha: float = 1.0, accumulate: bool = False):
    pixels = a.numel() // a.shape[-1]
    ws = torch.empty(256 * c, dtype=torch.float32, device=a.device)
    check(_lib.load().mcgen_prod_colsum(_p(a), a.shape[-1], _p(b), b.shape[-1], _dt(a.dtype), pixels, c, _f32(out), float(alpha),
                                        int(accumulate), _f32(ws), _stream()), 'prod_colsum')


def actnorm_bwd(partials: Tensor, scale: Tensor, ld_coef: float, input_side: bool, dloc: Tensor, dscale: Tensor,
                accumulate: bool = False):
    tiles, _, pitch = partials.shape
    check(_lib.load().mcgen_actnorm_bwd(_f32(partials), tiles, pitch, scale.numel(), _f32(scale), float(ld_coef),
                                        int(input_side), _f32(dloc), _f32(dscale), int(accumulate), _stream()), 'actnorm_bwd')


def invconv_bwd(w_p, w_l, w_u, w_s, s_sign, dW: Tensor, ld_coef: float, dw_l, dw_u, dw_s, accumulate: bool = False):
    c = w_s.numel()
    check(_lib.load().mcgen_invconv_bwd(_f32(w_p), _f32(w_l), _f32(w_u), _f32(w_s), _f32(s_sign), _f32(dW), c, dW.shape[-1],
                                        float(ld_coef), _f32(dw_l), _f32(dw_u), _f32(dw_s), int(accumulate), _stream()),
          'invconv_bwd')


def clip_grad_norm_(gflat: Tensor, max_norm: float) -> Tensor:
    """torch.nn.utils.clip_grad_norm_ over one flat gradient buffer; returns the total norm (device scalar)."""
    ws = torch.empty(256 + 1, dtype=torch.float32, device=gflat.device)
    check(_lib.load().mcgen_clip_grad_norm(_f32(gflat), gflat.numel(), float(max_norm), ws[256:].data_ptr(), _f32(ws),
                                           _stream()), 'clip_grad_norm')
    return ws[256]


# --------------------------------------------------------------------------- #
# MCPixelCNN (models/mcpixelcnn.py)
def im2col(x: Tensor, kh: int, kw: int, oh: int, ow: int, stride: int = 1, scale: Optional[Tensor] = None,
           shift: Optional[Tensor] = None, relu: bool = False, code: Optional[Tensor] = None) -> Tensor:
    """[N,H,W,Cp] -> [N,H/stride,W/stride,kh*kw*Cp] of the (optionally activated) input; see mcgen_im2col."""
    n, h, w, cp = x.shape
    col = torch.empty((n, h // stride, w // stride, kh * kw * cp), dtype=x.dtype, device=x.device)
    if code is not None and tuple(code.shape) != (n, cp):
        raise _lib.McgenError(f'im2col: code must be {(n, cp)}')
    check(_lib.load().mcgen_im2col(_p(x), _p(col), _dt(x.dtype), n, h, w, cp, kh, kw, oh, ow, stride, _f32(scale), _f32(shift),
                                   int(relu), _f32(code), _stream()), 'im2col')
    return col


def col2im(dcol: Tensor, cp: int, kh: int, kw: int, oh: int, ow: int, stride: int = 1, bias: Optional[Tensor] = None,
           out: Optional[Tensor] = None) -> Tensor:
    """Adjoint of im2col: [N,Ho,Wo,kh*kw*cp] -> [N,Ho*stride,Wo*stride,cp] (+ bias); accumulates into `out` if given."""
    n, ho, wo, _ = dcol.shape
    acc = out is not None
    if out is None:
        out = torch.empty((n, ho * stride, wo * stride, cp), dtype=dcol.dtype, device=dcol.device)
    check(_lib.load().mcgen_col2im(_p(dcol), _p(out), _dt(dcol.dtype), n, ho * stride, wo * stride, cp, kh, kw, oh, ow, stride,
                                   _f32(bias), bias.numel() if bias is not None else 0, int(acc), _stream()), 'col2im')
    return out


def gated_fwd(s: Tensor, scale: Tensor, shift: Tensor, code: Tensor) -> Tensor:
    n, h, w, c2 = s.shape
    c = c2 // 2
    out = torch.empty((n, h, w, c), dtype=s.dtype, device=s.device)
    check(_lib.load().mcgen_gated_fwd(_p(s), _f32(scale), _f32(shift), _f32(code), _p(out), _dt(s.dtype), n, h * w, c, _stream()),
          'gated_fwd')
    return out


def gated_fwd_batch(items):
    """items = [(s, scale, shift, code)]: independent gated activations in ONE launch (mcgen_gated_fwd_batch, <= 4) -> [out]."""
    arr = (_lib.Gated * len(items))()
    outs = []
    for d, (s, scale, shift, code) in zip(arr, items):
        n, h, w, c2 = s.shape
        out = torch.empty((n, h, w, c2 // 2), dtype=s.dtype, device=s.device)
        d.s, d.scale, d.shift, d.code, d.out = _p(s), _f32(scale), _f32(shift), _f32(code), _p(out)
        d.N, d.HW, d.C = n, h * w, c2 // 2
        outs.append(out)
    check(_lib.load().mcgen_gated_fwd_batch(arr, len(items), _dt(items[0][0].dtype), _stream()), 'gated_fwd_batch')
    return outs


def _bwd_sums(partials: Tensor, c: int, dgamma: Optional[Tensor], dbeta: Optional[Tensor]) -> Tensor:
    tiles, _, pitch = partials.shape
    sums = torch.empty((2, c), dtype=torch.float32, device=partials.device)
    check(_lib.load().mcgen_bn_bwd_finalize(_f32(partials), tiles, pitch, c, _f32(dgamma), _f32(dbeta), _f32(sums), 0, _stream()),
          'bn_bwd_finalize')
    return sums


def gated_bwd(s: Tensor, scale, shift, mean, rstd, code: Tensor, g: Tensor, dgamma: Tensor, dbeta: Tensor) -> Tensor:
    """Backward of gated_fwd through the batch statistics: returns ds [.., 2C]; fills dgamma / dbeta."""
    n, h, w, c2 = s.shape
    c = c2 // 2
    pixels = n * h * w
    blocks = max(1, min(256, pixels // 16))
    ds = torch.empty_like(s)
    part = torch.empty((blocks, 2, c), dtype=torch.float32, device=s.device)
    lib = _lib.load()
    check(lib.mcgen_gated_bwd_stats(_p(s), _f32(scale), _f32(shift), _f32(mean), _f32(rstd), _f32(code), _p(g), _p(ds), _f32(part),
                                    blocks, _dt(s.dtype), n, h * w, c, _stream()), 'gated_bwd_stats')
    sums = _bwd_sums(part, c, dgamma, dbeta)
    check(lib.mcgen_gated_bwd_apply(_p(ds), _p(s), _f32(sums), _f32(scale), _f32(mean), _f32(rstd), float(pixels), _dt(s.dtype),
                                    pixels, c, _stream()), 'gated_bwd_apply')
    return ds


def affine_code_res(x: Tensor, scale, shift, code: Optional[Tensor], res: Optional[Tensor], pre_relu: bool = False,
                    post_relu: bool = False) -> Tensor:
    """y = post_relu?( pre_relu?(x*scale + shift) * code + res ); x is [N, ..., C], code [N, C]."""
    n, c = x.shape[0], x.shape[-1]
    hw = x.numel() // (n * c)
    y = torch.empty_like(x)
    check(_lib.load().mcgen_affine_code_res(_p(x), _f32(scale), _f32(shift), _f32(code), _p(res), _p(y), _dt(x.dtype), n, hw, c,
                                            int(pre_relu), int(post_relu), _stream()), 'affine_code_res')
    return y


def affine_relu_maxpool2(x: Tensor, scale: Tensor, shift: Tensor) -> Tensor:
    """MaxPool2d(2)(relu(x * scale + shift)) on [N, H, W, C] (mcgen_affine_relu_maxpool2)."""
    n, h, w, c = x.shape
    if h % 2 or w % 2 or scale.numel() != c or shift.numel() != c:
        raise _lib.McgenError(f'affine_relu_maxpool2: even H, W and [C] affine vectors, got {tuple(x.shape)} / {scale.numel()}')
    y = torch.empty((n, h // 2, w // 2, c), dtype=x.dtype, device=x.device)
    check(_lib.load().mcgen_affine_relu_maxpool2(_p(x), _f32(scale), _f32(shift), _p(y), _dt(x.dtype), n, h // 2, w // 2, c, _stream()),
          'affine_relu_maxpool2')
    return y


def code_bn_bwd(g: Tensor, code: Optional[Tensor], x: Tensor, scale, mean, rstd, dgamma: Tensor, dbeta: Tensor,
                shift=None, pre_relu: bool = False, y_post: Optional[Tensor] = None, want_gated: bool = False):
    """Backward of y = post_relu?( pre_relu?(BN(x)) * code + res ) w.r.t. x (training-mode BatchNorm); fills
    dgamma / dbeta.  Returns dx, or (dx, g * [y_post > 0]) when want_gated (the residual's gradient)."""
    n, c = x.shape[0], x.shape[-1]
    hw = x.numel() // (n * c)
    pixels = n * hw
    blocks = max(1, min(256, pixels // 16))
    dz = torch.empty_like(x)
    gated = torch.empty_like(x) if (want_gated and y_post is not None) else None
    part = torch.empty((blocks, 2, c), dtype=torch.float32, device=x.device)
    lib = _lib.load()
    check(lib.mcgen_code_bn_stats(_p(g), _f32(code), _p(x), _f32(mean), _f32(rstd), _p(dz), _f32(part), blocks, _dt(x.dtype),
                                  n, hw, c, _f32(scale), _f32(shift), int(pre_relu), _p(y_post), _p(gated), _stream()),
          'code_bn_stats')
    sums = _bwd_sums(part, c, dgamma, dbeta)
    dx = torch.empty_like(x)
    check(lib.mcgen_bn_bwd_apply(_p(dz), _p(x), None, _p(dx), _dt(x.dtype), pixels, c, _f32(sums), float(pixels), _f32(scale),
                                 _f32(mean), _f32(rstd), _stream()), 'bn_bwd_apply')
    if want_gated:
        return dx, (gated if gated is not None else g)
    return dx


def bce_logits(logits: Tensor, target: Tensor, c: int, gscale: float, want_grad: bool):
    """-> (recon = sigmoid(logits), sum of BCE(recon, target) as a device scalar, dlogits or None)."""
    pixels = logits.numel() // logits.shape[-1]
    if target.shape != logits.shape or target.dtype != torch.float32:
        raise _lib.McgenError('bce_logits: target must be fp32 in the logits\' NHWC shape')
    blocks = max(1, min(1024, (logits.numel() + 255) // 256))
    part = torch.empty(blocks, dtype=torch.float32, device=logits.device)
    recon = torch.empty_like(logits)
    dl = torch.empty_like(logits) if want_grad else None
    check(_lib.load().mcgen_bce_logits(_p(logits), _f32(target), _p(recon), _p(dl), _f32(part), blocks, float(gscale),
                                       _dt(logits.dtype), pixels, c, logits.shape[-1], _stream()), 'bce_logits')
    return recon, part.double().sum().float(), dl


def cross_entropy(logits: Tensor, target: Tensor, c: int, want_grad: bool):
    """-> (loss_rows [pixels] fp32, dlogits or None); dlogits is the gradient of the MEAN loss."""
    pixels = logits.numel() // logits.shape[-1]
    rows = torch.empty(pixels, dtype=torch.float32, device=logits.device)
    dl = torch.empty_like(logits) if want_grad else None
    if target.dtype != torch.int64 or target.numel() != pixels:
        raise _lib.McgenError('cross_entropy: target must be int64 with one entry per pixel')
    check(_lib.load().mcgen_cross_entropy(_p(logits), target.contiguous().data_ptr(), _f32(rows), _p(dl), 1.0 / pixels,
                                          _dt(logits.dtype), pixels, c, logits.shape[-1], _stream()), 'cross_entropy')
    return rows, dl


def argmin_channels(x: Tensor, c: int) -> Tensor:
    """idx[...] = argmin over the first c channels of x[..., Cp] (int64)."""
    pixels = x.numel() // x.shape[-1]
    idx = torch.empty(x.shape[:-1], dtype=torch.int64, device=x.device)
    check(_lib.load().mcgen_argmin_channels(_p(x), idx.data_ptr(), _dt(x.dtype), pixels, c, x.shape[-1], _stream()), 'argmin_channels')
    return idx


def prep_weight_ex(w: Tensor, dtype: torch.dtype, ksize: Optional[int] = None, *, kh0: int = 0, kw0: int = 0,
                   transpose: bool = False, row_scale: Optional[Tensor] = None, col_scale: Optional[Tensor] = None,
                   rows_img: Optional[int] = None, k_img: Optional[int] = None, wscale: float = 1.0) -> Tensor:
    """Weight image straight from a (possibly strided / narrower) master tensor [Cout, Cin(, KH, KW)]: the source taps
    sit at (kh0, kw0) of the ksize x ksize image, row_scale[co] / col_scale[ci] scale the SOURCE, `transpose` builds the
    input-gradient image, rows_img / k_img widen the image with zeros (see mcgen_prep_weight_ex)."""
    if w.dtype != torch.float32 or not w.is_cuda:
        raise _lib.McgenError('prep_weight_ex: float32 device tensor expected')
    cout, cin = w.shape[0], w.shape[1]
    kh, kw = (w.shape[2], w.shape[3]) if w.dim() == 4 else (1, 1)
    st = w.stride()
    s_kh, s_kw = (st[2], st[3]) if w.dim() == 4 else (0, 0)
    ks = ksize if ksize is not None else kh
    rows = rows_img if rows_img is not None else (cin if transpose else cout)
    kk = k_img if k_img is not None else (cout if transpose else cin)
    n = weight_image_elems(kk if transpose else rows, rows if transpose else kk, ks, transpose)
    out = torch.empty(n, dtype=dtype, device=w.device)
    check(_lib.load().mcgen_prep_weight_ex(w.data_ptr(), st[0], st[1], s_kh, s_kw, cout, cin, kh, kw, kh0, kw0, ks, int(transpose),
                                           rows, kk, _f32(row_scale), _f32(col_scale), float(wscale), _p(out), _dt(dtype), _stream()),
          'prep_weight_ex')
    return out


def prep_weight_ex_many(jobs, dtype: torch.dtype):
    """Several prep_weight_ex images in ceil(n / 16) launches.  `jobs`: list of (w, kwargs) with the keyword arguments of
    prep_weight_ex (ksize, kh0, kw0, transpose, row_scale, col_scale, rows_img, k_img, wscale).  Returns the images."""
    if not jobs:
        return []
    arr = (_lib.PrepEx * len(jobs))()
    outs, keep = [], []
    for a, (w, kw) in zip(arr, jobs):
        if w.dtype != torch.float32 or not w.is_cuda:
            raise _lib.McgenError('prep_weight_ex_many: float32 device tensors expected')
        cout, cin = w.shape[0], w.shape[1]
        kh, kwid = (w.shape[2], w.shape[3]) if w.dim() == 4 else (1, 1)
        st = w.stride()
        transpose = bool(kw.get('transpose', False))
        ks = kw.get('ksize') or kh
        rows = kw.get('rows_img') or (cin if transpose else cout)
        kk = kw.get('k_img') or (cout if transpose else cin)
        out = kw.get('out')                                 # (a caller-owned slice: K-concatenated images without a torch.cat)
        if out is None:
            out = torch.empty(weight_image_elems(rows, kk, ks, False), dtype=dtype, device=w.device)
        elif out.numel() != weight_image_elems(rows, kk, ks, False) or out.dtype != dtype or not out.is_contiguous():
            raise _lib.McgenError('prep_weight_ex_many: `out` must be a contiguous image-sized tensor of the compute dtype')
        a.w, a.s_co, a.s_ci = w.data_ptr(), st[0], st[1]
        a.s_kh, a.s_kw = (st[2], st[3]) if w.dim() == 4 else (0, 0)
        a.Cout, a.Cin, a.KH, a.KW = cout, cin, kh, kwid
        a.kh0, a.kw0, a.ksize, a.transpose = kw.get('kh0', 0), kw.get('kw0', 0), ks, int(transpose)
        a.rows_img, a.k_img = rows, kk
        rs, cs = kw.get('row_scale'), kw.get('col_scale')
        a.row_scale, a.col_scale, a.image, a.wscale = _f32(rs), _f32(cs), _p(out), float(kw.get('wscale', 1.0))
        outs.append(out)
        keep += [w, rs, cs]
    check(_lib.load().mcgen_prep_weight_ex_batch(arr, len(jobs), _dt(dtype), _stream()), 'prep_weight_ex_batch')
    return outs
