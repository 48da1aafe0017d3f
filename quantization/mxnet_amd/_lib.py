"""ctypes binding of libfakequant.so (include/fakequant.h), in the style the reference binds libmxnet
(quantize/freeze/freeze.py:32-34,67-76: `check_call(_LIB.fn(...))`, int status, error text fetched separately).

There is no CPU fallback: if the library cannot be loaded every entry point raises, loudly.
"""
import ctypes
import os

__all__ = ["LIB", "check_call", "load", "library_path", "FakeQuantError", "EXPORTS"]

_HERE = os.path.dirname(os.path.abspath(__file__))
# FQ_LIB_PATH: load another build of the same library (debug / trace builds made by tools/)
_LIB_PATH = os.environ.get("FQ_LIB_PATH") or os.path.join(_HERE, "csrc", "libfakequant.so")

c_f32p = ctypes.POINTER(ctypes.c_float)
c_i32p = ctypes.POINTER(ctypes.c_int32)
c_u32p = ctypes.POINTER(ctypes.c_uint32)
c_u64p = ctypes.POINTER(ctypes.c_uint64)
_vp, _i64, _int, _uint, _f32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_uint, ctypes.c_float

# name -> (restype, argtypes); mirrors include/fakequant.h one to one (tests/test_abi.py checks both directions)
EXPORTS = {
    "fq_last_error": (ctypes.c_char_p, []),
    "fq_version": (_int, []),
    "fq_build_id": (ctypes.c_char_p, []),
    "fq_build_has": (_int, [ctypes.c_char_p]),
    "fq_device_info": (_int, [ctypes.c_char_p, _int, ctypes.POINTER(_int), ctypes.POINTER(_int)]),
    "fq_profile_enable": (_int, [_int]),
    "fq_profile_reset": (_int, []),
    "fq_profile_read": (_int, [_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(_i64),
                               ctypes.POINTER(ctypes.c_double)]),
    "fq_profile_read_moved": (_int, [_int, ctypes.POINTER(ctypes.c_double)]),
    "fq_profile_calibrate": (_int, [_vp, _int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), _vp]),
    "fq_profile_launch_overhead": (_int, [_vp, _int, ctypes.c_double, ctypes.POINTER(ctypes.c_double),
                                          ctypes.POINTER(ctypes.c_double), _vp]),
    "fq_act_workspace_bytes": (ctypes.c_size_t, [_i64]),
    "fq_absmax_per_sample": (_int, [_vp, _i64, _i64, _uint, _vp, _vp]),
    "fq_batch_mean": (_int, [_vp, _i64, _vp, _vp]),
    "fq_batch_mean_gathered": (_int, [_vp, _int, _i64, _vp, _vp]),
    "fq_stat_rows_sum": (_int, [_vp, _i64, _i64, _i64, _vp, _vp]),
    "fq_mean_from_sums": (_int, [_vp, _i64, _vp, _vp]),
    "fq_batch_mean_rows": (_int, [_vp, _i64, _i64, _i64, _vp, _vp]),
    "fq_fake_quant_online": (_int, [_vp, _vp, _i64, _i64, _int, _uint, _vp, _vp, _vp, _vp]),
    "fq_fake_quant_online_prestat": (_int, [_vp, _vp, _i64, _i64, _vp, _int, _uint, _vp, _vp, _vp]),
    "fq_bn_act_stat": (_int, [_vp, _vp, _i64, _i64, _i64, _vp, _vp, _int, _vp, _vp]),
    "fq_bn_act_maxpool_stat": (_int, [_vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _int, _vp, _vp]),
    "fq_add_act_stat": (_int, [_vp, _vp, _vp, _i64, _i64, _int, _vp, _vp]),
    "fq_bn_act_stat_hist": (_int, [_vp, _vp, _i64, _i64, _i64, _vp, _vp, _int, _vp, _vp, _int, _vp, _vp, _vp]),
    "fq_add_act_stat_hist": (_int, [_vp, _vp, _vp, _i64, _i64, _int, _vp, _vp, _int, _vp, _vp, _vp]),
    "fq_bn_add_act_stat": (_int, [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _int, _vp, _vp]),
    "fq_bn_add_act_stat_hist": (_int, [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _int, _vp, _vp, _int, _vp, _vp, _vp]),
    "fq_global_avg_pool_stat": (_int, [_vp, _vp, _i64, _i64, _i64, _int, _vp, _vp]),
    "fq_gemm_i8_codes": (_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _int, _vp]),
    "fq_eval_counters": (_int, [_vp, _vp, _i64, _i64, _vp, _vp]),
    "fq_stem_conv3x3s2": (_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _int, _vp, _vp]),
    "fq_stem_conv7x7s2": (_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _int, _vp, _vp]),
    "fq_stem_conv7x7s2_pool": (_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _int, _vp, _vp]),
    "fq_stem_conv7x7s2_pool_supported": (_int, [_i64, _i64]),
    "fq_stem_conv3x3s2_c16": (_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _int, _vp, _vp, _int, _uint, _vp]),
    "fq_dwconv3x3": (_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _int, _vp, _vp, _int, _uint, _vp, _vp, _vp, _int,
                            _vp, _vp]),
    "fq_dwconv3x3_c16": (_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _int, _vp, _vp, _int, _uint, _vp, _vp, _vp, _int,
                                _vp, _vp, _int, _uint, _vp]),
    "fq_weight_codes": (_int, [_vp, _i64, _i64, _int, _int, _i64, _i64, _vp, _vp, _vp, _vp, _vp]),
    "fq_pwconv_workspace_bytes": (ctypes.c_size_t, [_i64, _i64, _i64]),
    "fq_pwconv_i8": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _int, _uint, _vp, _vp,
                            _vp, _int, _vp, _vp, _vp]),
    "fq_pwconv_i8_stat_supported": (_int, [_i64, _i64, _i64, _i64]),
    "fq_pwconv_i8_stat": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _int, _uint, _vp, _vp,
                                 _vp, _int, _vp, _vp]),
    "fq_pwdw_fused_supported": (_int, [_i64, _i64, _i64, _i64, _i64, _int]),
    # x, wcodes, wscale, wsum, pw_bias, n, cin, cin_pad, cout_pad, cout, h, w, in_stat, in_thr, in_width, in_flags, pw_bn_scale,
    # pw_bn_shift, pw_act, mid_stat, mid_thr, mid_width, mid_flags, mid_cur, dw_w, dw_bias, dw_stride, dw_bn_scale, dw_bn_shift,
    # dw_act, y, stat_out, stream
    "fq_pwdw_fused": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _int, _uint, _vp,
                             _vp, _int, _vp, _vp, _int, _uint, _vp, _vp, _vp, _int, _vp, _vp, _int, _vp, _vp, _vp]),
    "fq_debug_fast_quotient": (_int, [_vp, _i64, _vp, _vp, _vp, _vp]),
    "fq_dense_i8_eval_workspace_bytes": (ctypes.c_size_t, [_i64, _i64]),
    "fq_dense_i8_eval": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _int, _uint, _vp, _vp, _vp,
                                _vp, _vp, _vp]),
    "fq_pwconv_i8_strided": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _int, _vp, _vp, _int,
                                    _uint, _vp, _vp, _vp, _int, _vp, _vp, _vp, _vp]),
    "fq_pwconv_i8_shortcut_supported": (_int, [_i64, _i64, _i64]),
    "fq_pwconv_i8_shortcut": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _int, _uint, _vp, _vp,
                                     _vp, _int, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _int, _uint, _vp, _vp, _vp, _vp]),
    "fq_pwconv_i8_shortcut_c16": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _int, _uint, _vp,
                                         _vp, _vp, _int, _vp, _vp, _int, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _int, _uint, _vp, _vp,
                                         _vp, _vp, _int, _uint, _vp]),
    "fq_pwconv_i8_gap_supported": (_int, [_i64, _i64, _i64, _i64, _int]),
    "fq_pwconv_i8_gap": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _int, _uint, _vp, _vp, _vp,
                                _int, _vp, _vp, _vp, _vp]),
    "fq_pwconv_i8_sub2_supported": (_int, [_i64, _i64]),
    "fq_pwconv_i8_sub2": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _int,
                                 _uint, _vp, _vp, _vp, _int, _vp, _vp, _vp, _vp]),
    "fq_pwconv_i8_c16": (_int, [_vp, _int, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _int, _vp, _vp, _int,
                                _uint, _vp, _vp, _vp, _int, _vp, _vp, _vp, _int, _uint, _vp, _vp]),
    # x, wcodes, wscale, wsum, bias, y, y16, n, cin, cin_pad, cout, h, w, in_stat, in_thr, in_width, in_flags, cur, bn_scale,
    # bn_shift, act, stat_out, residual, out_thr, out_width, out_flags, ws, stream
    "fq_pwconv_i8_c16_dual": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _int,
                                     _uint, _vp, _vp, _vp, _int, _vp, _vp, _vp, _int, _uint, _vp, _vp]),
    "fq_pwconv_i8_c16_dual_sub2": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _int,
                                          _uint, _vp, _vp, _vp, _int, _vp, _vp, _vp, _int, _uint, _vp, _vp]),
    "fq_conv3x3_i8": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _int, _uint, _vp, _vp,
                             _vp, _int, _vp, _vp]),
    "fq_conv3x3_i8_c16": (_int, [_vp, _int, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _int, _uint, _vp,
                                 _vp, _vp, _int, _vp, _vp, _int, _uint, _vp]),
    "fq_conv3x3_i8_sliced": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _int, _uint, _vp,
                                    _vp, _vp, _int, _vp, _vp]),
    "fq_weight_slices": (_int, [_vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp]),
    "fq_fake_quant_offline": (_int, [_vp, _vp, _i64, _i64, _vp, _int, _uint, _vp, _vp, _vp, _vp]),
    "fq_ste_forward": (_int, [_vp, _vp, _i64, _i64, _vp, _int, _f32, _f32, _f32, _vp]),
    "fq_weight_workspace_bytes": (ctypes.c_size_t, [_i64]),
    "fq_weight_fake_quant": (_int, [_vp, _vp, _i64, _i64, _int, _vp, _vp, _vp]),
    "fq_wino_weight_fake_quant": (_int, [_vp, _vp, _i64, _i64, _int, c_f32p, c_f32p, c_f32p, _int, _vp, _vp, _vp]),
    "fq_comm_unique_id": (_int, [_vp]),
    "fq_comm_init": (_int, [_int, _int, _vp]),
    "fq_comm_world": (_int, []),
    "fq_allreduce_f32": (_int, [_vp, _i64, _int, _vp]),
    "fq_allreduce_f64": (_int, [_vp, _i64, _int, _vp]),
    "fq_allreduce_i64": (_int, [_vp, _i64, _int, _vp]),
    "fq_comm_destroy": (_int, []),
    "fq_ema_update": (_int, [_vp, _vp, _i64, ctypes.c_double, _vp]),
    "fq_global_max": (_int, [_vp, _i64, _vp, _vp]),
    "fq_histogram_accumulate": (_int, [_vp, _i64, _vp, _int, _vp, _vp, _vp]),
    "fq_hist_to_float": (_int, [_vp, _vp, _i64, _vp]),
    "fq_kl_workspace_bytes": (ctypes.c_size_t, [_i64, _int]),
    "fq_kl_search": (_int, [_vp, _i64, _int, _int, _int, _vp, _vp, _vp]),
    "fq_quantize_codes": (_int, [_vp, _vp, _i64, _int, _vp, _vp, _vp]),
    "fq_dequantize": (_int, [_vp, _vp, _i64, _vp, _vp]),
    "fq_qconv_weights_bytes": (ctypes.c_size_t, [_i64, _i64, _int, _int, _int, _int, _int, _int, _int]),
    "fq_qconv_kind": (_int, [_i64, _i64, _int, _int, _int, _int, _int, _int, _int]),
    "fq_qconv_weights_prepare": (_int, [_vp, _i64, _i64, _int, _int, _int, _int, _int, _int, _int, _int, _f32, _f32, _vp,
                                        _vp, _vp]),
    "fq_qconv_workspace_bytes": (ctypes.c_size_t, [_i64]),
    "fq_qconv_workspace_init": (_int, [_vp, _vp]),
    "fq_qconv2d_forward": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _int, _int, _int, _int, _int, _int,
                                  _int, _int, _f32, _f32, _vp, _int, _vp, _vp, _vp, _vp, _int, _vp]),
}

FQ_ACT_SIGNED, FQ_ACT_LO_NEG_MAX, FQ_ACT_NO_ABS, FQ_ACT_NO_EPS = 1, 2, 4, 8
FQ_CODES_INT8, FQ_CODES_UINT8, FQ_CODES_RANGE, FQ_CODES_SCALE = 0, 1, 2, 3


class FakeQuantError(RuntimeError):
    """Error raised by the HIP library (the analogue of MXNetError)."""


class _Missing(object):
    def __init__(self, why):
        self._why = why

    def __getattr__(self, name):
        raise FakeQuantError(
            "libfakequant.so is not available (%s). Build it with `python -m quantization.mxnet_amd.csrc.build` "
            "(needs hipcc, target gfx950). There is no CPU fallback for the fake-quant path." % self._why)


def library_path():
    return _LIB_PATH


def load(path=None):
    """dlopen the library and attach prototypes.  torch is imported first so that the HIP runtime already mapped
    into the process (torch's bundled libamdhip64.so.7) is the one libfakequant binds to — two HIP runtimes in
    one process would not share device pointers or streams."""
    import torch  # noqa: F401  (side effect: loads libamdhip64)
    path = path or _LIB_PATH
    if not os.path.exists(path):
        return _Missing("no file at %s" % path)
    try:
        lib = ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
    except OSError as e:            # pragma: no cover
        return _Missing(str(e))
    for name, (restype, argtypes) in EXPORTS.items():
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    return lib


LIB = load()


def check_call(ret):
    if ret != 0:
        raise FakeQuantError(LIB.fq_last_error().decode("utf-8", "replace"))
