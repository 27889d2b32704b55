"""ctypes binding of libmcgen_hip.so (the C ABI in include/mcgen_hip.h)."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'csrc', 'libmcgen_hip.so')

F32, BF16 = 0, 1


class Seg(C.Structure):
    _fields_ = [('x', C.c_void_p), ('scale', C.c_void_p), ('shift', C.c_void_p), ('code', C.c_void_p), ('cmap', C.c_void_p),
                ('C', C.c_int32), ('ups', C.c_int32), ('relu', C.c_int32), ('ksize', C.c_int32),
                ('group_n', C.c_int32), ('cmap_stride', C.c_int32), ('Cw', C.c_int32), ('reserved_', C.c_int32)]


class Conv(C.Structure):
    _fields_ = [('seg', Seg * 2), ('nseg', C.c_int32),
                ('w', C.c_void_p), ('bias', C.c_void_p), ('y', C.c_void_p),
                ('N', C.c_int32), ('H', C.c_int32), ('W', C.c_int32),
                ('Cout', C.c_int32), ('Cout_w', C.c_int32), ('Cy', C.c_int32),
                ('pool', C.c_int32), ('alpha', C.c_float),
                ('res', C.c_void_p), ('ocode', C.c_void_p), ('gate_x', C.c_void_p),
                ('gscale', C.c_void_p), ('gshift', C.c_void_p), ('gmean', C.c_void_p), ('grstd', C.c_void_p),
                ('tanh_out', C.c_int32), ('stats', C.c_void_p), ('stats_mode', C.c_int32), ('w_layout', C.c_int32),
                ('ycmap', C.c_void_p), ('ycmap_stride', C.c_int32), ('y_group', C.c_int32), ('bias2', C.c_void_p),
                ('wsel', C.c_void_p), ('wsel_stride', C.c_int64), ('order', C.c_void_p),
                ('yperm', C.c_void_p), ('yperm_stride', C.c_int32), ('reserved_', C.c_int32)]


class Wgrad(C.Structure):
    _fields_ = [('seg', Seg), ('dy', C.c_void_p),
                ('N', C.c_int32), ('H', C.c_int32), ('W', C.c_int32),
                ('Cout', C.c_int32), ('Cout_w', C.c_int32), ('Cdy', C.c_int32),
                ('dy_ups', C.c_int32), ('slabs', C.c_void_p), ('splits', C.c_int32), ('halves', C.c_int32),
                ('bias_slabs', C.c_void_p)]


WGRAD_MULTI_MAX = 16       # MCGEN_WGRAD_MULTI_MAX (include/mcgen_hip.h)


class WReduce(C.Structure):
    _fields_ = [('slabs', C.c_void_p), ('grad', C.c_void_p), ('bias_slabs', C.c_void_p), ('bias_grad', C.c_void_p),
                ('bias_grad2', C.c_void_p),
                ('splits', C.c_int32), ('Cout', C.c_int32), ('Cin', C.c_int32), ('ksize', C.c_int32), ('Cout_w', C.c_int32),
                ('row_perm', C.c_int32), ('accumulate', C.c_int32), ('alpha', C.c_float),
                ('row_scale', C.c_void_p), ('cin_slab', C.c_int32), ('tapcols', C.c_int32), ('tap0', C.c_int32), ('ntap_out', C.c_int32)]


class PrepEx(C.Structure):
    _fields_ = [('w', C.c_void_p), ('s_co', C.c_int64), ('s_ci', C.c_int64), ('s_kh', C.c_int64), ('s_kw', C.c_int64),
                ('Cout', C.c_int32), ('Cin', C.c_int32), ('KH', C.c_int32), ('KW', C.c_int32), ('kh0', C.c_int32), ('kw0', C.c_int32),
                ('ksize', C.c_int32), ('transpose', C.c_int32), ('rows_img', C.c_int32), ('k_img', C.c_int32),
                ('row_scale', C.c_void_p), ('col_scale', C.c_void_p), ('image', C.c_void_p), ('wscale', C.c_float), ('_pad', C.c_int32)]


class Prep(C.Structure):
    _fields_ = [('w', C.c_void_p), ('image', C.c_void_p),
                ('Cout', C.c_int32), ('Cin', C.c_int32), ('ksize', C.c_int32), ('transpose', C.c_int32),
                ('row_perm', C.c_int32), ('sigma_idx', C.c_int32), ('wscale', C.c_float), ('layout', C.c_int32),
                ('kmap', C.c_void_p), ('kcount', C.c_int32), ('_pad', C.c_int32), ('rmap', C.c_void_p)]


class BnFin(C.Structure):                                  # mcgen_bn_fin_t
    _fields_ = [('partials', C.c_void_p), ('tiles', C.c_int32), ('pitch', C.c_int32), ('fold', C.c_int32), ('C', C.c_int32),
                ('count', C.c_double), ('gamma', C.c_void_p), ('beta', C.c_void_p), ('running_mean', C.c_void_p),
                ('running_var', C.c_void_p), ('momentum', C.c_float), ('eps', C.c_float),
                ('scale', C.c_void_p), ('shift', C.c_void_p), ('mean', C.c_void_p), ('rstd', C.c_void_p)]


class Gated(C.Structure):                                  # mcgen_gated_t
    _fields_ = [('s', C.c_void_p), ('scale', C.c_void_p), ('shift', C.c_void_p), ('code', C.c_void_p), ('out', C.c_void_p),
                ('N', C.c_int32), ('HW', C.c_int32), ('C', C.c_int32), ('_pad', C.c_int32)]


class Code(C.Structure):
    _fields_ = [('codebook', C.c_void_p), ('out_off', C.c_int64), ('M', C.c_int32), ('C', C.c_int32),
                ('scale_idx', C.c_int32), ('_pad', C.c_int32)]


class AnAffine(C.Structure):                               # mcgen_an_affine_t
    _fields_ = [('loc', C.c_void_p), ('scale', C.c_void_p), ('a', C.c_void_p), ('b', C.c_void_p), ('negloc', C.c_void_p),
                ('C', C.c_int32), ('Cp', C.c_int32)]


class Pld(C.Structure):                                    # mcgen_pld_t
    _fields_ = [('scale', C.c_void_p), ('w_s', C.c_void_p), ('C', C.c_int32), ('Cw', C.c_int32), ('hw', C.c_float), ('_pad', C.c_int32)]


class Icw(C.Structure):                                    # mcgen_icw_t
    _fields_ = [('w_p', C.c_void_p), ('w_l', C.c_void_p), ('w_u', C.c_void_p), ('w_s', C.c_void_p), ('s_sign', C.c_void_p),
                ('weight', C.c_void_p), ('weight_inv', C.c_void_p), ('C', C.c_int32), ('_pad', C.c_int32)]


class Icb(C.Structure):                                    # mcgen_icb_t
    _fields_ = [('w_p', C.c_void_p), ('w_l', C.c_void_p), ('w_u', C.c_void_p), ('w_s', C.c_void_p), ('s_sign', C.c_void_p),
                ('dW', C.c_void_p), ('dw_l', C.c_void_p), ('dw_u', C.c_void_p), ('dw_s', C.c_void_p),
                ('C', C.c_int32), ('ldw', C.c_int32), ('accumulate', C.c_int32), ('ld_coef', C.c_float)]


class AnBwd(C.Structure):                                  # mcgen_an_bwd_t
    _fields_ = [('partials', C.c_void_p), ('scale', C.c_void_p), ('dloc', C.c_void_p), ('dscale', C.c_void_p),
                ('tiles', C.c_int32), ('pitch', C.c_int32), ('C', C.c_int32), ('input_side', C.c_int32), ('accumulate', C.c_int32),
                ('ld_coef', C.c_float)]


class Pcs(C.Structure):                                    # mcgen_pcs_t
    _fields_ = [('a', C.c_void_p), ('b', C.c_void_p), ('out', C.c_void_p), ('pixels', C.c_int64),
                ('pitch_a', C.c_int32), ('pitch_b', C.c_int32), ('C', C.c_int32), ('accumulate', C.c_int32),
                ('alpha', C.c_float), ('_pad', C.c_int32)]


class BnRun(C.Structure):                                  # mcgen_bn_run_t
    _fields_ = [('running_mean', C.c_void_p), ('running_var', C.c_void_p), ('mean', C.c_void_p), ('unb', C.c_void_p),
                ('groups', C.c_int32), ('C', C.c_int32), ('momentum', C.c_float), ('_pad', C.c_int32)]


class SnLayer(C.Structure):
    _fields_ = [('w_off', C.c_int64), ('u_off', C.c_int64), ('v_off', C.c_int64),
                ('rows', C.c_int32), ('cols', C.c_int32)]


# every symbol include/mcgen_hip.h declares: name -> (restype, argtypes)
_vp, _i, _f, _d, _i64 = C.c_void_p, C.c_int, C.c_float, C.c_double, C.c_int64
SYMBOLS = {
    'mcgen_last_error': (C.c_char_p, []),
    'mcgen_abi_version': (_i, []),
    'mcgen_conv_m_tiles': (_i, [C.POINTER(Conv), _i]),
    'mcgen_conv_tile': (_i, [C.POINTER(Conv), _i, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    'mcgen_conv_form': (_i, [C.POINTER(Conv), _i]),
    'mcgen_conv_fused': (_i, [C.POINTER(Conv), _i, _vp]),
    'mcgen_wgrad_slab_elems': (_i64, [C.POINTER(Wgrad)]),
    'mcgen_wgrad': (_i, [C.POINTER(Wgrad), _i, _vp]),
    'mcgen_wgrad_multi_ok': (_i, [C.POINTER(Wgrad), _i]),
    'mcgen_wgrad_c8_ok': (_i, [_vp, _i]),
    'mcgen_wgrad_c8_slab_elems': (_i64, [_vp]),
    'mcgen_wgrad_multi': (_i, [C.POINTER(Wgrad), _i, _i, _vp]),
    'mcgen_wgrad_batch': (_i, [C.POINTER(Wgrad), _i, _i, _vp]),
    'mcgen_bn_finalize_batch': (_i, [C.POINTER(BnFin), _i, _vp]),
    'mcgen_gated_fwd_batch': (_i, [C.POINTER(Gated), _i, _i, _vp]),
    'mcgen_wgrad_reduce': (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _f, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'mcgen_weight_image_elems': (_i64, [_i, _i, _i, _i]),
    'mcgen_prep_weight': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _f, _vp]),
    'mcgen_prep_weight_rows': (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    'mcgen_glow_squeeze': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'mcgen_channel_stats': (_i, [_vp, _i, _i64, _i, _vp, _i, _vp]),
    'mcgen_actnorm_init': (_i, [_vp, _i, _i, _i, _d, _vp, _vp, _vp]),
    'mcgen_actnorm_affine': (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp]),
    'mcgen_glow_param_logdet': (_i, [_vp, _i, _vp, _i, _f, _vp, _i, _vp]),
    'mcgen_invconv_weight': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp]),
    'mcgen_glow_coupling': (_i, [_vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'mcgen_gaussian_logp': (_i, [_vp, _i, _i, _vp, _i, _i, _i, _i, _i, _vp, _i, _vp]),
    'mcgen_gaussian_sample': (_i, [_vp, _i, _vp, _i, _vp, _i, _i, _i, _i64, _i, _vp]),
    'mcgen_copy_channels': (_i, [_vp, _i, _i, _vp, _i, _i, _i, _i64, _i, _vp]),
    'mcgen_glow_coupling_bwd': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _f, _i64, _i, _i, _vp]),
    'mcgen_gaussian_logp_bwd': (_i, [_vp, _i, _i, _vp, _i, _vp, _i, _i, _vp, _i, _f, _i64, _i, _i, _vp]),
    'mcgen_actnorm_affine_batch': (_i, [_vp, _i, _vp]),
    'mcgen_glow_param_logdet_batch': (_i, [_vp, _i, _vp, _i, _vp]),
    'mcgen_invconv_weight_batch': (_i, [_vp, _i, _vp]),
    'mcgen_invconv_bwd_batch': (_i, [_vp, _i, _vp]),
    'mcgen_actnorm_bwd_batch': (_i, [_vp, _i, _vp]),
    'mcgen_prod_colsum_batch': (_i, [_vp, _i, _i, _vp, _vp]),
    'mcgen_prod_colsum': (_i, [_vp, _i, _vp, _i, _i, _i64, _i, _vp, _f, _i, _vp, _vp]),
    'mcgen_actnorm_bwd': (_i, [_vp, _i, _i, _i, _vp, _f, _i, _vp, _vp, _i, _vp]),
    'mcgen_invconv_bwd': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp, _vp, _vp, _i, _vp]),
    'mcgen_clip_grad_norm': (_i, [_vp, _i64, _f, _vp, _vp, _vp]),
    'mcgen_im2col': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp]),
    'mcgen_col2im': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _i, _vp]),
    'mcgen_argmin_channels': (_i, [_vp, _vp, _i, _i64, _i, _i, _vp]),
    'mcgen_bce_logits': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _f, _i, _i64, _i, _i, _vp]),
    'mcgen_gated_fwd': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'mcgen_gated_bwd_stats': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'mcgen_gated_bwd_apply': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_double, _i, _i64, _i, _vp]),
    'mcgen_affine_code_res': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'mcgen_affine_relu_maxpool2': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'mcgen_code_bn_stats': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp]),
    'mcgen_cross_entropy': (_i, [_vp, _vp, _vp, _vp, _f, _i, _i64, _i, _i, _vp]),
    'mcgen_wgrad_reduce_batch': (_i, [_vp, _i, _vp]),
    'mcgen_prep_weight_ex': (_i, [_vp, _i64, _i64, _i64, _i64, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _f, _vp, _i, _vp]),
    'mcgen_prep_weight_ex_batch': (_i, [_vp, _i, _i, _vp]),
    'mcgen_weight_image_k_elems': (_i64, [_i, _i, _i]),
    'mcgen_prep_weight_k': (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _f, _vp]),
    'mcgen_mc_affine': (_i, [_vp, _vp, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    'mcgen_cmap_stride': (C.c_int32, [_i]),
    'mcgen_mc_cmap': (_i, [_vp, _i, _i, _vp, _vp]),
    'mcgen_prep_weight_batch': (_i, [_vp, _i, _vp, _i, _vp]),
    'mcgen_mc_code_batch': (_i, [_vp, _vp, _i, _vp, _i, _vp, _i, _vp]),
    'mcgen_nchw_to_nhwc': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'mcgen_nhwc_to_nchw': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'mcgen_pool2_sum': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'mcgen_mc_gather_batch': (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp]),
    'mcgen_prep_weight_batch_codes': (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp]),
    'mcgen_mc_code': (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    'mcgen_mc_apply': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'mcgen_bn_finalize': (_i, [_vp, _i, _i, _i, _i, _d, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp]),
    'mcgen_bn_finalize_groups': (_i, [_vp, _i, _i, _i, _i, _d, _i, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp]),
    'mcgen_bn_finalize_par': (_i, [_vp, _i, _i, _i, _i, _d, _i, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
    'mcgen_bn_running_batch': (_i, [_vp, _i, _vp]),
    'mcgen_bn_eval_affine': (_i, [_vp, _vp, _vp, _vp, _f, _i, _vp, _vp, _vp]),
    'mcgen_bn_bwd_finalize': (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _i, _vp]),
    'mcgen_bn_bwd_apply': (_i, [_vp, _vp, _vp, _vp, _i, _i64, _i, _vp, _d, _vp, _vp, _vp, _vp]),
    'mcgen_colsum': (_i, [_vp, _i, _i64, _i, _i, _vp, _i, _f, _i, _vp, _vp]),
    'mcgen_sn_power_iter': (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _i, _i, _vp]),
    'mcgen_sn_power_iter_snap': (_i, [_vp, _vp, _vp, _i, _vp, _vp, _i, _i, _vp, _vp]),
    'mcgen_sn_power_iter_rounds': (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _i, _i, _vp, _i64, _vp, _vp]),
    'mcgen_sn_power_iter_fused': (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _i64, _i, _i, _vp]),
    'mcgen_sn_grad_fix': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp, _vp]),
    'mcgen_sn_grad_fix_pair': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp]),
    'mcgen_dtail_fwd': (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    'mcgen_dtail_bwd': (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'mcgen_dtail_pair_wgrad': (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    'mcgen_dtail_pair_wgrad_loss': (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    'mcgen_dtail_hinge_fused': (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'mcgen_hinge_d': (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp]),
    'mcgen_hinge_g': (_i, [_vp, _i, _vp, _vp, _vp]),
    'mcgen_tanh_bwd': (_i, [_vp, _vp, _vp, _i, _i64, _vp]),
    'mcgen_onehot_rep': (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    'mcgen_adam': (_i, [_vp, _vp, _vp, _vp, _i64, _f, _vp, _f, _f, _f, _f, _vp, _vp]),
    'mcgen_sn_fix_pair_adam': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _f, _vp, _f, _f, _f, _f, _vp, _i, _vp]),
}

_lib = None


class McgenError(RuntimeError):
    pass


def load():
    """Load the HIP library; raises (never falls back) if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise McgenError(f'{LIB_PATH} is missing: run __graft_entry__.build() '
                             f'(or csrc/build.sh); there is no CPU fallback')
        # PyTorch-ROCm ships its own libamdhip64; the process must have ONE HIP runtime, and it has to be the one torch's
        # streams and allocations live in: import torch first, so that this library's libamdhip64 dependency resolves to
        # the copy already loaded (loaded the other way round, every launch fails with "no ROCm-capable device").
        import torch  # noqa: F401
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)          # AttributeError if the .so lacks a declared symbol
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def check(rc: int, what: str = ''):
    if rc != 0:
        raise McgenError(f'{what}: {load().mcgen_last_error().decode()}')
