"""ctypes binding of libctdet.so (include/ctdet.h).

The HIP library is the product: if it is missing or fails to load, importing the ops
FAILS LOUDLY -- there is no PyTorch / CPU fallback for the device path.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('CTDET_LIB') or os.path.join(os.path.dirname(_HERE), 'lib', 'libctdet.so')

CT_OK = 0


class CtdetError(RuntimeError):
    pass


class OutSegment(C.Structure):
    _fields_ = [('ptr', C.c_void_p), ('co_begin', C.c_int), ('co_end', C.c_int),
                ('pix_stride', C.c_int), ('img_stride', C.c_longlong), ('base', C.c_longlong)]


class ConvDesc(C.Structure):
    _fields_ = [
        ('in_', C.c_void_p),
        ('batch', C.c_int), ('cin', C.c_int), ('h', C.c_int), ('w', C.c_int),
        ('in_ctot', C.c_int), ('in_coff', C.c_int),
        ('wpacked', C.c_void_p), ('scale', C.c_void_p), ('shift', C.c_void_p),
        ('cout', C.c_int), ('m_pad', C.c_int), ('k_pad', C.c_int),
        ('kh', C.c_int), ('kw', C.c_int), ('stride', C.c_int),
        ('pad_h', C.c_int), ('pad_w', C.c_int), ('dil', C.c_int),
        ('oh', C.c_int), ('ow', C.c_int),
        ('out', C.c_void_p), ('out_ctot', C.c_int), ('out_coff', C.c_int),
        ('res', C.c_void_p), ('res_ctot', C.c_int), ('res_coff', C.c_int),
        ('res_scale', C.c_float), ('relu', C.c_int), ('lo', C.c_void_p),
        ('nseg', C.c_int), ('seg', OutSegment * 3),
        ('config', C.c_int),
        ('transposed', C.c_int),
        ('ksplit', C.c_int), ('ksplit_ws', C.c_void_p), ('ksplit_ws_floats', C.c_longlong),
        ('in_absmax', C.c_void_p), ('out_absmax', C.c_void_p),
    ]


ABSMAX_LINE_BYTES = 128         # CT_ABSMAX_LINE_BYTES: one line per image in ct_conv_desc.in_absmax / out_absmax


class ProfileRecord(C.Structure):
    _fields_ = [('name', C.c_char_p), ('ms', C.c_float)]


class CtxParams(C.Structure):
    _fields_ = [('theta_w', C.c_void_p), ('theta_b', C.c_void_p), ('phi_w', C.c_void_p),
                ('phi_b', C.c_void_p), ('g_w', C.c_void_p), ('g_b', C.c_void_p),
                ('wz', C.c_void_p), ('obj_w', C.c_void_p), ('fc_w', C.c_void_p), ('fc_b', C.c_void_p),
                ('scale', C.c_float), ('d', C.c_int), ('t', C.c_int)]


class CtxGrads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ('theta_w', 'theta_b', 'phi_w', 'phi_b', 'g_w', 'g_b', 'wz', 'obj_w',
                                          'fc_w', 'fc_b')]


_P = C.c_void_p
_I = C.c_int
_F = C.c_float
_Z = C.c_size_t
_L = C.c_long
_LL = C.c_longlong

# name -> (restype, argtypes); every symbol include/ctdet.h declares
SIGNATURES = {
    'ct_abi_version': (_I, []),
    'ct_last_error_string': (C.c_char_p, []),
    'ct_device_info': (_I, [_I, C.POINTER(_I), C.POINTER(_I), C.c_char_p, _I]),
    'ct_pack_record_begin': (_I, []),
    'ct_pack_record_bytes': (_Z, []),
    'ct_pack_record_end': (_I, [_P, _Z, C.POINTER(_I), C.POINTER(_I), _P]),
    'ct_pack_run': (_I, [_P, _I, _I, _P]),
    'ct_profile_enable': (_I, [_I]),
    'ct_profile_collect': (_I, [C.POINTER(ProfileRecord), _I, C.POINTER(_I)]),
    'ct_nms_sorted_host': (_I, [_P, _P, _P, _I, _I, _F, _I]),
    'ct_nms_sorted_host_mode': (_I, [_P, _P, _P, _I, _I, _F, _I, _I]),
    'ct_nms_batched_workspace_bytes': (_Z, [_I, _I]),
    'ct_nms_batched_dev': (_I, [_P, _P, _I, _I, _F, _I, _P, _P, _P, _Z, _P]),
    'ct_cpu_nms': (_I, [_P, _I, _F, _I, _P, _P]),
    'ct_cpu_soft_nms': (_I, [_P, _I, _F, _F, _F, C.c_uint, _P]),
    'ct_decode': (_I, [_P, _P, _I, _I, _F, _F, _P, _I, _P, _P]),
    'ct_encode': (_I, [_P, _P, _I, _F, _F, _P, _P]),
    'ct_detect_fused': (_I, [_P, _P, _P, _P, _I, _I, _I, _F, _F, _I, _P, _I, _P, _P, _P]),
    'ct_softmax_lastdim': (_I, [_P, _P, _L, _I, _P]),
    'ct_jaccard': (_I, [_P, _I, _P, _I, _I, _P, _P]),
    'ct_match_workspace_bytes': (_Z, [_I, _I, _I]),
    'ct_match_batched': (_I, [_P, _P, _I, _I, _P, _I, _F, _F, _F, _P, _P, _P, _P, _P, _Z, _P]),
    'ct_postprocess_workspace_bytes': (_Z, [_I, _I, _I]),
    'ct_postprocess_batched': (_I, [_P, _P, _I, _I, _I, _F, _F, _I, _I, _I, _P, _P, _P, _P, _P, _Z, _P]),
    'ct_conv_kpad': (_I, [_I, _I, _I]),
    'ct_conv_mpad': (_I, [_I]),
    'ct_conv_num_configs': (_I, []),
    'ct_conv_config_name': (C.c_char_p, [_I]),
    'ct_conv_pack_weights': (_I, [_P, _P, _I, _I, _I, _I, _P, _I, _I, _P]),
    'ct_conv_pack_weights_dgrad': (_I, [_P, _P, _I, _I, _I, _I, _P, _I, _I, _P]),
    'ct_conv2d_wgrad': (_I, [C.POINTER(ConvDesc), _P, _I, _I, _P, _P]),
    'ct_conv_bf16_cin_pad': (_I, [_I]),
    'ct_conv_bf16_cout_pad': (_I, [_I]),
    'ct_conv_bf16_packed_elems': (C.c_size_t, [_I, _I, _I, _I]),
    'ct_conv_pack_weights_bf16': (_I, [_P, _P, _I, _I, _I, _I, _P, _P]),
    'ct_nchw_f32_to_nhwc_bf16': (_I, [_P, _I, _I, _I, _I, _P, _P]),
    'ct_nhwc_bf16_to_nchw_f32': (_I, [_P, _I, _I, _I, _I, _I, _P, _P]),
    'ct_maxpool2d_nhwc_bf16': (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    'ct_conv2d_bf16_fwd': (_I, [C.POINTER(ConvDesc), _P]),
    'ct_conv_wgrad_wino_supported': (_I, [C.POINTER(ConvDesc)]),
    'ct_conv_wgrad_wino_workspace_bytes': (C.c_size_t, [C.POINTER(ConvDesc)]),
    'ct_conv2d_wgrad_wino': (_I, [C.POINTER(ConvDesc), _P, _I, _I, _P, _P, _P]),
    'ct_conv_wgrad_wino4_supported': (_I, [C.POINTER(ConvDesc)]),
    'ct_conv_wgrad_wino4_workspace_bytes': (C.c_size_t, [C.POINTER(ConvDesc)]),
    'ct_conv2d_wgrad_wino4': (_I, [C.POINTER(ConvDesc), _P, _I, _I, _P, _P, _P]),
    'ct_conv_wgrad_wino4s_supported': (_I, [C.POINTER(ConvDesc)]),
    'ct_conv_wgrad_wino4s_workspace_bytes': (_Z, [C.POINTER(ConvDesc)]),
    'ct_conv2d_wgrad_wino4s': (_I, [C.POINTER(ConvDesc), _P, _I, _I, _P, _P, _Z, _P]),
    'ct_bn_train_stats': (_I, [_P, _I, _I, _I, _I, _I, _P, _P, _F, _P, _P, _P, _P]),
    'ct_bn_train_apply': (_I, [_P, _P, _P, _P, _P, _F, _I, _P, _P, _I, _I, _F, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    'ct_bn_train_backward': (_I, [_P, _I, _I, _P, _I, _I, _P, _P, _P, _P, _F, _I, _P, _F, _P, _I, _I, _I,
                                  _P, _P, _P, _I, _I, _I, _I, _I, _P, _P]),
    'ct_bn_eval_backward': (_I, [_P, _I, _I, _P, _I, _I, _P, _P, _P, _P, _F, _I, _P, _F, _P, _I, _I, _I,
                                 _P, _P, _P, _I, _I, _I, _I, _I, _P, _P]),
    'ct_bias_act_backward': (_I, [_P, _I, _I, _P, _I, _I, _I, _I, _I, _I, _P, _I, _I, _P, _P]),
    'ct_maxpool2x2_bias_relu_bwd': (_I, [_P, _I, _I, _P, _I, _I, _I, _I, _I, _I, _P, _I, _I, _P, _P, _P]),
    'ct_bias_act_backward_amax': (_I, [_P, _I, _I, _P, _I, _I, _I, _I, _I, _I, _P, _I, _I, _P, _P, _P]),
    'ct_maxpool2d_bwd': (_I, [_P, _P, _P, _L, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    'ct_head_grad_gather': (_I, [C.POINTER(OutSegment), _I, _I, _I, _I, _P, _P]),
    'ct_conv_fold_epilogue': (_I, [_P, _P, _P, _P, _F, _P, _I, _I, _P, _P, _P]),
    'ct_conv_wino_supported': (_I, [C.POINTER(ConvDesc)]),
    'ct_conv_wino_packed_floats': (_Z, [_I, _I]),
    'ct_conv_pack_weights_wino': (_I, [_P, _P, _I, _I, _P, _P]),
    'ct_conv_pack_weights_wino_dgrad': (_I, [_P, _P, _I, _I, _P, _P]),
    'ct_conv2d_wino_fwd': (_I, [C.POINTER(ConvDesc), _P, _P]),
    'ct_conv2d_wino_pool_fwd': (_I, [C.POINTER(ConvDesc), _P, _P, _I, _I, _I, _I, _I, _P]),
    'ct_scratch_prezeroed': (_I, [_I]),
    'ct_conv_wino_x3_supported': (_I, [C.POINTER(ConvDesc)]),
    'ct_conv_wino_x3_packed_bytes': (_Z, [_I, _I]),
    'ct_conv_pack_weights_wino_x3': (_I, [_P, _P, _I, _I, _P, _P]),
    'ct_conv_pack_weights_wino_x3_dgrad': (_I, [_P, _P, _I, _I, _P, _P]),
    'ct_conv2d_wino_x3_fwd': (_I, [C.POINTER(ConvDesc), _P, _I, _P]),
    'ct_conv2d_wino_x3_pool_fwd': (_I, [C.POINTER(ConvDesc), _P, _I, _P, _I, _I, _I, _I, _I, _P]),
    'ct_conv_wino4s_supported': (_I, [C.POINTER(ConvDesc)]),
    'ct_conv_wino4s_packed_bytes': (_Z, [_I, _I]),
    'ct_conv_wino4s_workspace_bytes': (_Z, [C.POINTER(ConvDesc)]),
    'ct_conv_pack_weights_wino4s': (_I, [_P, _P, _I, _I, _P, _P]),
    'ct_conv_pack_weights_wino4s_dgrad': (_I, [_P, _P, _I, _I, _P, _P]),
    'ct_conv_wino4s_h2_packed_bytes': (_Z, [_I, _I]),
    'ct_conv_wino_h2_pack_item_bytes': (_Z, []),
    'ct_conv_wino_h2_pack_item': (_I, [_P, _P, _I, _I, _I, _I, _P, _P]),
    'ct_conv_wino_h2_pack_run': (_I, [_P, _I, _P]),
    'ct_conv_pack_weights_wino4s_h2': (_I, [_P, _P, _I, _I, _P, _P]),
    'ct_conv_pack_weights_wino4s_h2_dgrad': (_I, [_P, _P, _I, _I, _P, _P]),
    'ct_conv2d_wino4s_fwd': (_I, [C.POINTER(ConvDesc), _P, _P, _Z, _I, _P]),
    'ct_conv2d_wino4s_pool_fwd': (_I, [C.POINTER(ConvDesc), _P, _P, _Z, _I, _P, _I, _I, _I, _I, _I, _P]),
    'ct_conv_wino4f_supported': (_I, [C.POINTER(ConvDesc)]),
    'ct_conv_wino4f_packed_bytes': (_Z, [_I, _I]),
    'ct_conv_pack_weights_wino4f': (_I, [_P, _P, _I, _I, _P, _P]),
    'ct_conv_pack_weights_wino4f_dgrad': (_I, [_P, _P, _I, _I, _P, _P]),
    'ct_conv_wino4f_h2_packed_bytes': (_Z, [_I, _I]),
    'ct_conv_pack_weights_wino4f_h2': (_I, [_P, _P, _I, _I, _P, _P]),
    'ct_conv_pack_weights_wino4f_h2_dgrad': (_I, [_P, _P, _I, _I, _P, _P]),
    'ct_conv2d_wino4f_pool_fwd_v': (_I, [C.POINTER(ConvDesc), _P, _I, _P, _I, _I, _I, _I, _I, _P]),
    'ct_conv2d_wino4f_fwd': (_I, [C.POINTER(ConvDesc), _P, _P]),
    'ct_conv2d_wino4f_pool_fwd': (_I, [C.POINTER(ConvDesc), _P, _P, _I, _I, _I, _I, _I, _P]),
    'ct_conv_wino4_supported': (_I, [C.POINTER(ConvDesc)]),
    'ct_conv_wino4_packed_floats': (_Z, [_I, _I]),
    'ct_conv_pack_weights_wino4': (_I, [_P, _P, _I, _I, _P, _P]),
    'ct_conv_pack_weights_wino4_dgrad': (_I, [_P, _P, _I, _I, _P, _P]),
    'ct_conv2d_wino4_fwd': (_I, [C.POINTER(ConvDesc), _P, _P]),
    'ct_conv2d_wino4_pool_fwd': (_I, [C.POINTER(ConvDesc), _P, _P, _I, _I, _I, _I, _I, _P]),
    'ct_conv2d_fwd': (_I, [C.POINTER(ConvDesc), _P]),
    'ct_absmax_f32': (_I, [_P, _I, _LL, _LL, _P, _P]),
    'ct_conv_x3_num_configs': (_I, []),
    'ct_conv_x3_config_name': (C.c_char_p, [_I]),
    'ct_conv_x3_config_bk': (_I, [_I]),
    'ct_conv_x3_config_h2': (_I, [_I]),
    'ct_conv_x3_packed_bytes': (_Z, [_I, _I, _I, _I, _I]),
    'ct_conv_x3h_packed_bytes': (_Z, [_I, _I, _I, _I, _I]),
    'ct_conv_pack_weights_x3h': (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _P]),
    'ct_conv_pack_weights_x3': (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _P]),
    'ct_conv_pack_weights_x3_dgrad': (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _P]),
    'ct_conv2d_x3_fwd': (_I, [C.POINTER(ConvDesc), _P, _I, _P]),
    'ct_conv_x3_pack_item_bytes': (_Z, []),
    'ct_conv_x3_pack_item': (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _I, _P]),
    'ct_conv_x3_pack_run': (_I, [_P, _I, _P]),
    'ct_maxpool2d_fwd': (_I, [_P, _P, _L, _I, _I, _I, _I, _I, _I, _I, _P]),
    'ct_preproc_resize': (_I, [_P, _P, _P, _I, _I, _P, _P, _P]),
    'ct_preproc_augment': (_I, [_P, _P, _I, _I, _P, _P, _P]),
    'ct_mixup_blend': (_I, [_P, _P, _P, _I, _L, _P, _P]),
    'ct_ctx_pool_fwd': (_I, [_P, _LL, _P, _LL, _I, _I, _I, _I, _I, _P]),
    'ct_ctx_attention_workspace_bytes': (_Z, [_I, _I, _I, _I]),
    'ct_ctx_attention_piece_products': (_I, []),
    'ct_ctx_attention_fwd': (_I, [_P, _P, _I, _I, _I, C.POINTER(CtxParams), _P, _P, _Z, _P]),
    'ct_ctx_attention_saved_bytes': (_Z, [_I, _I]),
    'ct_ctx_attention_fwd_train': (_I, [_P, _P, _I, _I, _I, C.POINTER(CtxParams), _P, _P, _Z, _P, _Z, _P]),
    'ct_ctx_attention_bwd_workspace_bytes': (_Z, [_I, _I, _I]),
    'ct_ctx_attention_bwd': (_I, [_P, _P, _I, _I, _I, C.POINTER(CtxParams), _P, _P, _P, _P, C.POINTER(CtxGrads),
                                  _P, _Z, _P]),
    'ct_ctx_pool_bwd': (_I, [_P, _LL, _P, _LL, _P, _LL, _I, _I, _I, _I, _I, _P]),
}

_lib = None


def lib():
    """The loaded library (loads on first use; raises CtdetError when it cannot)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise CtdetError(
                'libctdet.so not found at %s -- build it with `python context-transformer_amd/build.py` '
                '(the HIP library is the product; there is no fallback path)' % LIB_PATH)
        # The host glue hands torch's device pointers and streams to the library, so both must sit on ONE HIP
        # runtime: PyTorch-ROCm ships its own libamdhip64; importing torch first makes the dynamic loader resolve
        # libctdet's dependency to that copy instead of a second runtime from /opt/rocm (which would know no device
        # context of torch's allocations: "no ROCm-capable device is detected" on the first launch).
        import torch  # noqa: F401
        try:
            handle = C.CDLL(LIB_PATH)
        except OSError as e:
            raise CtdetError('cannot load %s: %s' % (LIB_PATH, e))
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(handle, name)
            except AttributeError:
                raise CtdetError('%s does not export %s (stale build?)' % (LIB_PATH, name))
            fn.restype = res
            fn.argtypes = args
        if handle.ct_abi_version() != 1:
            raise CtdetError('libctdet ABI version %d, expected 1' % handle.ct_abi_version())
        _lib = handle
    return _lib


def check(status, what=''):
    if status != CT_OK:
        msg = lib().ct_last_error_string().decode(errors='replace')
        raise CtdetError('%s failed (status %d): %s' % (what or 'libctdet call', status, msg))
