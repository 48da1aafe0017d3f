"""ctypes binding of oracle/libfq_host.so (include/fakequant_host.h) over numpy arrays — TEST INFRASTRUCTURE.

The C++/OpenMP restatement is the fast oracle for parity checks at BASELINE's full sizes and the timed CPU baseline of
bench.py.  Same import rule as the rest of oracle/: tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only.
Function names and return values follow oracle/fq_oracle.py where both exist.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "libfq_host.so")

FQ_ACT_SIGNED, FQ_ACT_LO_NEG_MAX, FQ_ACT_NO_ABS, FQ_ACT_NO_EPS = 1, 2, 4, 8
_ACTS = {None: 0, "none": 0, "relu": 1, "relu6": 2}
F32 = np.float32


def build(force=False):
    if force or not os.path.exists(_PATH) or \
            os.path.getmtime(_PATH) < os.path.getmtime(os.path.join(_HERE, "fq_host.cpp")):
        subprocess.check_call(["make", "-s", "-C", _HERE] + (["-B"] if force else []))
    return _PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_PATH)
        _lib.fq_last_error_host.restype = ctypes.c_char_p
        # default team: the physical cores of one socket at most (FQ_HOST_THREADS overrides; set_threads(0) = all)
        _lib.fq_set_threads_host(int(os.environ.get("FQ_HOST_THREADS", min(os.cpu_count() or 1, 64))))
    return _lib


def _p(a):
    return None if a is None else ctypes.c_void_p(a.ctypes.data)


def _f32(a):
    return np.ascontiguousarray(a, dtype=F32)


def _call(name, *args):
    fn = getattr(lib(), name)
    conv = []
    for a in args:
        if isinstance(a, np.ndarray):
            conv.append(_p(a))
        elif isinstance(a, float):
            conv.append(ctypes.c_float(a))
        elif isinstance(a, int):
            conv.append(ctypes.c_int64(a))
        else:
            conv.append(a)
    rc = fn(*conv)
    if rc != 0:
        raise RuntimeError(lib().fq_last_error_host().decode())


def _i(v):
    return ctypes.c_int(int(v))


def _u(v):
    return ctypes.c_uint(int(v))


def threads():
    return int(lib().fq_threads_host())


def set_threads(k):
    lib().fq_set_threads_host(int(k))


def act_flags(signed=False, lo_neg_max=None, no_abs=False, no_eps=False):
    if lo_neg_max is None:
        lo_neg_max = signed
    return (FQ_ACT_SIGNED if signed else 0) | (FQ_ACT_LO_NEG_MAX if lo_neg_max else 0) | \
        (FQ_ACT_NO_ABS if no_abs else 0) | (FQ_ACT_NO_EPS if no_eps else 0)


def _n_inner(x):
    return x.shape[0], x.size // x.shape[0]


def absmax_per_sample(x, no_abs=False):
    x = _f32(x)
    n, inner = _n_inner(x)
    out = np.empty(n, F32)
    _call("fq_absmax_per_sample_host", x, n, inner, _u(FQ_ACT_NO_ABS if no_abs else 0), out, None)
    return out


def batch_mean(v):
    v = _f32(v)
    out = np.empty(1, F32)
    _call("fq_batch_mean_host", v, v.size, out, None)
    return out[0]


def fake_quant_online(x, width=8, flags=0, want_codes=False):
    """-> (y, current_max, codes | None)"""
    x = _f32(x)
    n, inner = _n_inner(x)
    y = np.empty_like(x)
    cur = np.empty(1, F32)
    codes = np.empty(x.shape, np.int32) if want_codes else None
    _call("fq_fake_quant_online_host", x, y, n, inner, _i(width), _u(flags), cur, codes, None, None)
    return y, cur[0], codes


def fake_quant_online_prestat(x, stat, width=8, flags=0, want_codes=False):
    x = _f32(x)
    n, inner = _n_inner(x)
    y = np.empty_like(x)
    cur = np.empty(1, F32)
    codes = np.empty(x.shape, np.int32) if want_codes else None
    _call("fq_fake_quant_online_prestat_host", x, y, n, inner, _f32(stat), _i(width), _u(flags), cur, codes, None)
    return y, cur[0], codes


def fake_quant_offline(x, threshold, width=8, flags=0, want_codes=False, want_stat=True):
    x = _f32(x)
    n, inner = _n_inner(x)
    y = np.empty_like(x)
    cur = np.empty(1, F32) if want_stat else None
    codes = np.empty(x.shape, np.int32) if want_codes else None
    thr = np.asarray([threshold], F32).reshape(1)
    _call("fq_fake_quant_offline_host", x, y, n, inner, thr, _i(width), _u(flags), cur, codes, None, None)
    return y, (cur[0] if want_stat else None), codes


def unfused_chain(x, width=8, flags=0, tmp=None, out=None):
    x = _f32(x)
    n, inner = _n_inner(x)
    y = np.empty_like(x) if out is None else out
    cur = np.empty(1, F32)
    _call("fq_unfused_chain_host", x, y, n, inner, _i(width), _u(flags), cur, tmp, None)
    return y, cur[0]


def bn_act(x, scale, shift, act="relu", want_stat=False):
    x = _f32(x)
    n, c = x.shape[0], x.shape[1]
    hw = x.size // (n * c)
    y = np.empty_like(x)
    stat = np.zeros(n, F32) if want_stat else None
    _call("fq_bn_act_stat_host", x, y, n, c, hw, _f32(scale), _f32(shift), _i(_ACTS[act]), stat, None)
    return (y, stat) if want_stat else y


def bn_act_hist(x, scale, shift, act, max_, bins, hist=None):
    """fq_bn_act_stat_hist_host -> (y, stat, hist uint64, negatives)"""
    x = _f32(x)
    n, c = x.shape[0], x.shape[1]
    y, stat = np.empty_like(x), np.zeros(n, F32)
    hist = np.zeros(bins, np.uint64) if hist is None else hist
    neg = np.zeros(1, np.uint32)
    _call("fq_bn_act_stat_hist_host", x, y, n, c, x.size // (n * c), _f32(scale), _f32(shift), _i(_ACTS[act]), stat,
          np.asarray([max_], F32), _i(bins), hist, neg, None)
    return y, stat, hist, int(neg[0])


def add_act_hist(a, b, act, max_, bins, hist=None):
    """fq_add_act_stat_hist_host -> (y, stat, hist uint64, negatives)"""
    a, b = _f32(a), _f32(b)
    n = a.shape[0]
    y, stat = np.empty_like(a), np.zeros(n, F32)
    hist = np.zeros(bins, np.uint64) if hist is None else hist
    neg = np.zeros(1, np.uint32)
    _call("fq_add_act_stat_hist_host", a, b, y, n, a.size // n, _i(_ACTS[act]), stat, np.asarray([max_], F32), _i(bins),
          hist, neg, None)
    return y, stat, hist, int(neg[0])


def bn_act_maxpool(x, scale, shift, act="relu", want_stat=False):
    x = _f32(x)
    n, c, h, w = x.shape
    y = np.empty((n, c, (h - 1) // 2 + 1, (w - 1) // 2 + 1), F32)
    stat = np.zeros(n, F32) if want_stat else None
    _call("fq_bn_act_maxpool_stat_host", x, y, n, c, h, w, _f32(scale), _f32(shift), _i(_ACTS[act]), stat, None)
    return (y, stat) if want_stat else y


def add_act(a, b, act="relu", want_stat=False):
    a, b = _f32(a), _f32(b)
    n = a.shape[0]
    y = np.empty_like(a)
    stat = np.zeros(n, F32) if want_stat else None
    _call("fq_add_act_stat_host", a, b, y, n, a.size // n, _i(_ACTS[act]), stat, None)
    return (y, stat) if want_stat else y


def global_avg_pool(x, want_stat=False):
    x = _f32(x)
    n, c = x.shape[0], x.shape[1]
    hw = x.size // (n * c)
    y = np.empty((n, c, 1, 1), F32)
    stat = np.zeros(n, F32) if want_stat else None
    _call("fq_global_avg_pool_stat_host", x, y, n, c, hw, _i(0), stat, None)
    return (y, stat) if want_stat else y


def stem_conv_s2(x, w, bias=None, bn_scale=None, bn_shift=None, act=None, want_stat=False, pool=False):
    x, w = _f32(x), _f32(w)
    n, cin, h, wd = x.shape
    cout, ks = w.shape[0], w.shape[2]
    pad = ks // 2
    wt = np.ascontiguousarray(w.transpose(1, 2, 3, 0))
    ho, wo = (h + 2 * pad - ks) // 2 + 1, (wd + 2 * pad - ks) // 2 + 1
    y = np.empty((n, cout, (ho - 1) // 2 + 1, (wo - 1) // 2 + 1) if pool else (n, cout, ho, wo), F32)
    stat = np.zeros(n, F32) if want_stat else None
    name = "fq_stem_conv7x7s2_pool_host" if pool else {3: "fq_stem_conv3x3s2_host", 7: "fq_stem_conv7x7s2_host"}[ks]
    _call(name, x, wt, None if bias is None else _f32(bias), y, n,
          cin, cout, h, wd, None if bn_scale is None else _f32(bn_scale), None if bn_shift is None else _f32(bn_shift),
          _i(_ACTS[act]), stat, None)
    return (y, stat) if want_stat else y


stem_conv3x3s2 = stem_conv_s2


def dwconv3x3(x, w, bias=None, stride=1, in_max=None, in_stat=None, signed=False, width=8, lo_neg_max=None,
              bn_scale=None, bn_shift=None, act=None, want_stat=False):
    x, w = _f32(x), _f32(w)
    n, c, h, wd = x.shape
    y = np.empty((n, c, (h - 1) // stride + 1, (wd - 1) // stride + 1), F32)
    stat = np.zeros(n, F32) if want_stat else None
    thr = None if in_max is None else np.asarray([in_max], F32).reshape(1)
    cur = np.empty(1, F32)
    _call("fq_dwconv3x3_host", x, w, None if bias is None else _f32(bias), y, n, c, h, wd, _i(stride),
          None if in_stat is None else _f32(in_stat), thr, _i(width), _u(act_flags(signed, lo_neg_max)), cur,
          None if bn_scale is None else _f32(bn_scale), None if bn_shift is None else _f32(bn_shift),
          _i(_ACTS[act]), stat, None)
    return (y, stat) if want_stat else y


def weight_codes(w, rows_per_scale, width=8):
    """-> (codes int32 (rows, row_len), scales, rowsum)"""
    w = _f32(w)
    rows = w.shape[0]
    row_len = w.size // rows
    row_pad = (row_len + 63) // 64 * 64
    rows_pad = (rows + 63) // 64 * 64
    codes = np.empty((rows_pad, row_pad), np.int8)
    scales = np.empty(rows_pad, F32)
    rowsum = np.empty(rows_pad, np.int32)
    _call("fq_weight_codes_host", w, rows, row_len, _i(rows_per_scale), _i(width), row_pad, rows_pad, codes, scales,
          rowsum, None, None)
    return codes, scales[:rows].copy(), rowsum[:rows].copy()


def pwconv_i8(x, w, rows_per_scale, wt_width, in_max=None, in_stat=None, signed=False, width=8, lo_neg_max=None,
              bias=None, bn_scale=None, bn_shift=None, act=None, want_stat=False, stride=1, residual=None, subsample=False):
    x = _f32(x)
    if subsample:
        assert stride == 1 and x.ndim == 4
        n, cin, h, wd = x.shape
        codes, scales, rowsum = weight_codes(np.asarray(w).reshape(np.asarray(w).shape[0], -1), rows_per_scale, wt_width)
        cout = np.asarray(w).shape[0]
        y = np.empty((n, cout, (h + 1) // 2, (wd + 1) // 2), F32)
        stat = np.zeros(n, F32) if want_stat else None
        thr = None if in_max is None else np.asarray([in_max], F32).reshape(1)
        cur = np.empty(1, F32)
        _call("fq_pwconv_i8_sub2_host", x, codes, scales, rowsum, None if bias is None else _f32(bias), y, n, cin,
              codes.shape[1], cout, h, wd, None if in_stat is None else _f32(in_stat), thr, _i(width),
              _u(act_flags(signed, lo_neg_max)), cur, None if bn_scale is None else _f32(bn_scale),
              None if bn_shift is None else _f32(bn_shift), _i(_ACTS[act]), stat,
              None if residual is None else _f32(residual), None, None)
        return (y, stat) if want_stat else y
    if stride != 1 or residual is not None:
        return _pwconv_i8_strided(x, w, rows_per_scale, wt_width, in_max, in_stat, signed, width, lo_neg_max, bias,
                                  bn_scale, bn_shift, act, want_stat, stride, residual)
    n, cin = x.shape[0], x.shape[1]
    hw = x.size // (n * cin)
    codes, scales, rowsum = weight_codes(w, rows_per_scale, wt_width)
    cout = np.asarray(w).shape[0]
    y = np.empty((n, cout) + x.shape[2:], F32)
    stat = np.zeros(n, F32) if want_stat else None
    thr = None if in_max is None else np.asarray([in_max], F32).reshape(1)
    cur = np.empty(1, F32)
    _call("fq_pwconv_i8_host", x, codes, scales, rowsum, None if bias is None else _f32(bias), y, n, cin,
          codes.shape[1], cout, hw, None if in_stat is None else _f32(in_stat), thr, _i(width),
          _u(act_flags(signed, lo_neg_max)), cur, None if bn_scale is None else _f32(bn_scale),
          None if bn_shift is None else _f32(bn_shift), _i(_ACTS[act]), stat, None, None)
    return (y, stat) if want_stat else y


def _pwconv_i8_strided(x, w, rows_per_scale, wt_width, in_max, in_stat, signed, width, lo_neg_max, bias, bn_scale,
                       bn_shift, act, want_stat, stride, residual=None):
    n, cin, h, wd = x.shape
    codes, scales, rowsum = weight_codes(w, rows_per_scale, wt_width)
    cout = np.asarray(w).shape[0]
    y = np.empty((n, cout, (h - 1) // stride + 1, (wd - 1) // stride + 1), F32)
    stat = np.zeros(n, F32) if want_stat else None
    thr = None if in_max is None else np.asarray([in_max], F32).reshape(1)
    cur = np.empty(1, F32)
    _call("fq_pwconv_i8_strided_host", x, codes, scales, rowsum, None if bias is None else _f32(bias), y, n, cin,
          codes.shape[1], cout, h, wd, _i(stride), None if in_stat is None else _f32(in_stat), thr, _i(width),
          _u(act_flags(signed, lo_neg_max)), cur, None if bn_scale is None else _f32(bn_scale),
          None if bn_shift is None else _f32(bn_shift), _i(_ACTS[act]), stat,
          None if residual is None else _f32(residual), None, None)
    return (y, stat) if want_stat else y


def conv3x3_i8(x, w, rows_per_scale, wt_width, in_max=None, in_stat=None, signed=False, width=8, lo_neg_max=None,
               bias=None, bn_scale=None, bn_shift=None, act=None, want_stat=False):
    """w: (cout, cin, 3, 3); the codes are taken from the weights permuted to (cout, 3, 3, cin) as the device entry point
    expects (per-row scales and sums do not depend on the order inside a row)."""
    x = _f32(x)
    n, cin, h, wd = x.shape
    wp = np.ascontiguousarray(np.transpose(_f32(w), (0, 2, 3, 1)))
    cout = wp.shape[0]
    rows_pad = (cout + 63) // 64 * 64
    codes = np.zeros((rows_pad, 9 * cin), np.int8)
    scales = np.empty(rows_pad, F32)
    rowsum = np.empty(rows_pad, np.int32)
    _call("fq_weight_codes_host", wp, cout, 9 * cin, _i(rows_per_scale), _i(wt_width), 9 * cin, rows_pad, codes, scales,
          rowsum, None, None)
    y = np.empty((n, cout, h, wd), F32)
    stat = np.zeros(n, F32) if want_stat else None
    thr = None if in_max is None else np.asarray([in_max], F32).reshape(1)
    cur = np.empty(1, F32)
    _call("fq_conv3x3_i8_host", x, codes, scales[:cout].copy(), rowsum[:cout].copy(),
          None if bias is None else _f32(bias), y, n, cin, cout, h, wd, None if in_stat is None else _f32(in_stat), thr,
          _i(width), _u(act_flags(signed, lo_neg_max)), cur, None if bn_scale is None else _f32(bn_scale),
          None if bn_shift is None else _f32(bn_shift), _i(_ACTS[act]), stat, None)
    return (y, stat) if want_stat else y


def conv3x3_i8_sliced(x, w, in_max=None, in_stat=None, signed=False, width=8, lo_neg_max=None, bias=None, bn_scale=None,
                      bn_shift=None, act=None, want_stat=False, want_slices=False):
    """w: (cout, cin, 3, 3), any fp32 filter (under Winograd-domain quantisation: the back-transformed one): three int8
    digit slices of the filter permuted to (cout, 3, 3, cin) (fq_weight_slices_host), then the exact sliced convolution."""
    x = _f32(x)
    n, cin, h, wd = x.shape
    wp = np.ascontiguousarray(np.transpose(_f32(w), (0, 2, 3, 1)))
    cout = wp.shape[0]
    rows_pad = (cout + 63) // 64 * 64
    codes = np.zeros((3, 2 * rows_pad * 9 * cin), np.int8)
    pscale = np.empty(cout, F32)
    rowsum = np.empty((3, cout), np.int32)
    _call("fq_weight_slices_host", wp, cout, 9 * cin, 9 * cin, rows_pad, codes, pscale, rowsum, None, None)
    y = np.empty((n, cout, h, wd), F32)
    stat = np.zeros(n, F32) if want_stat else None
    thr = None if in_max is None else np.asarray([in_max], F32).reshape(1)
    cur = np.empty(1, F32)
    _call("fq_conv3x3_i8_sliced_host", x, codes, pscale, rowsum, None if bias is None else _f32(bias), y, n, cin, cout, h, wd,
          None if in_stat is None else _f32(in_stat), thr, _i(width), _u(act_flags(signed, lo_neg_max)), cur,
          None if bn_scale is None else _f32(bn_scale), None if bn_shift is None else _f32(bn_shift), _i(_ACTS[act]), stat,
          None)
    out = (y, stat) if want_stat else y
    return (out, codes, pscale, rowsum) if want_slices else out


def ste_forward(x, scales, clip_max=None, clip_min=None, eps=1e-10):
    x = _f32(x)
    scales = _f32(scales).reshape(-1)
    rows = scales.size
    y = np.empty_like(x)
    has_clip = clip_max is not None
    lo = 0.0 if clip_min is None else float(clip_min)
    _call("fq_ste_forward_host", x, y, rows, x.size // rows, scales, _i(1 if has_clip else 0),
          ctypes.c_float(lo), ctypes.c_float(float(clip_max) if has_clip else 0.0), ctypes.c_float(eps), None)
    return y


def weight_fake_quant(w, rows, width=8):
    w = _f32(w)
    wq = np.empty_like(w)
    scales = np.empty(rows, F32)
    _call("fq_weight_fake_quant_host", w, wq, rows, w.size // rows, _i(width), scales, None, None)
    return wq, scales


def wino_weight_fake_quant(w, G, GI, GTI, width=8):
    w = _f32(w)
    wq = np.empty_like(w)
    scales = np.empty(w.shape[0], F32)
    _call("fq_wino_weight_fake_quant_host", w, wq, w.shape[0], w.shape[1], _i(np.asarray(G).shape[0]), _f32(G), _f32(GI),
          _f32(GTI), _i(width), scales, None, None)
    return wq, scales


def ema_update(state, current, momentum=0.9):
    st = _f32(state).copy()
    _call("fq_ema_update_host", st, _f32(current), st.size, ctypes.c_double(momentum), None)
    return st


def global_max(x):
    x = _f32(x)
    out = np.empty(1, F32)
    _call("fq_global_max_host", x, x.size, out, None)
    return out[0]


def histogram_accumulate(x, max_, bins=2048, hist=None):
    """-> (hist uint64, negatives)"""
    x = _f32(x)
    if hist is None:
        hist = np.zeros(bins, np.uint64)
    neg = np.zeros(1, np.uint32)
    _call("fq_histogram_accumulate_host", x, x.size, np.asarray([max_], F32), _i(bins), hist, neg, None)
    return hist, int(neg[0])


def discrete_histogram(fm, bins, max_=None):
    """oracle.discrete_histogram's interface: -> (hist fp32, max_)"""
    fm = _f32(fm)
    if max_ is None:
        max_ = global_max(fm)
    h, neg = histogram_accumulate(fm, max_, bins)
    assert neg == 0, "Activation should >=0"
    out = np.empty(bins, F32)
    _call("fq_hist_to_float_host", h, out, bins, None)
    return out, F32(max_)


def kl_search(hist, levels, min_bins):
    hist = _f32(hist)
    if hist.ndim == 1:
        hist = hist.reshape(1, -1)
    L, bins = hist.shape
    out = np.empty(L, np.int32)
    _call("fq_kl_search_host", hist, L, _i(bins), _i(levels), _i(min_bins), out, None, None)
    return out


_MODES = {"int8": 0, "uint8": 1, "range": 2, "scale": 3}


def quantize_codes(x, out_type="int8", range_=None):
    x = _f32(x)
    codes = np.empty(x.shape, np.int32)
    rng = np.zeros(3, F32) if range_ is None else _f32(range_).copy()
    _call("fq_quantize_codes_host", x, codes, x.size, _i(_MODES[out_type]), rng, None, None)
    return codes, rng


def dequantize(codes, scale):
    codes = np.ascontiguousarray(codes, np.int32)
    y = np.empty(codes.shape, F32)
    _call("fq_dequantize_host", codes, y, codes.size, np.asarray([scale], F32), None)
    return y


def gemm_i8_codes(xcodes, wcodes, n, l, zoff=0):
    xc = np.ascontiguousarray(xcodes, np.int8)
    wc = np.ascontiguousarray(wcodes, np.int8)
    cout, k = wc.shape
    wsum = wc.astype(np.int32).sum(axis=1).astype(np.int32)
    out = np.empty((n, cout, l), np.int32)
    _call("fq_gemm_i8_codes_host", xc, wc, wsum, out, n, l, k, cout, _i(zoff), None)
    return out


def dense_i8_eval(x, w, rows_per_scale, wt_width, labels, counters=None, in_max=None, in_stat=None, signed=False, width=8,
                  lo_neg_max=False, bias=None):
    """fq_dense_i8_eval_host: x (n, cin), w (units, cin); returns (logits, counters)."""
    x = _f32(x)
    n, cin = x.shape
    codes, scales, rowsum = weight_codes(w, rows_per_scale, wt_width)
    cout = np.asarray(w).shape[0]
    y = np.empty((n, cout), F32)
    thr = None if in_max is None else np.asarray([in_max], F32).reshape(1)
    cur = np.empty(1, F32)
    out = np.zeros(2 + 2 * cout, F32) if counters is None else _f32(counters).copy()
    _call("fq_dense_i8_eval_host", x, codes, scales, rowsum, None if bias is None else _f32(bias), y, n, cin,
          codes.shape[1], cout, None if in_stat is None else _f32(in_stat), thr, _i(width),
          _u(act_flags(signed, lo_neg_max)), cur, np.ascontiguousarray(labels, np.int64), out, None, None, None)
    return y, out


def eval_counters(logits, labels, counters=None):
    logits = _f32(logits)
    n, classes = logits.shape
    out = np.zeros(2 + 2 * classes, F32) if counters is None else _f32(counters).copy()
    _call("fq_eval_counters_host", logits, np.ascontiguousarray(labels, np.int64), n, classes, out, None)
    return out


_QMODES = {"int8": 0, "uint8": 1}


def qconv2d_forward(x, w, b, stride, padding, groups, input_dtype="uint8", weight_dtype="int8", input_range=None,
                    weight_range=None, act=None, in_stat=None, bn_scale=None, bn_shift=None, want_stat=False):
    """fq_qconv_weights_prepare_host + fq_qconv2d_forward_host: fq_oracle.qconv2d_forward(quantized=True) at full size."""
    x, w = _f32(x), _f32(w)
    n, cin, h, wd = x.shape
    cout, cin_g, kh, kw = w.shape
    sh, sw = stride
    ph, pw = padding
    wbuf = np.zeros(16, F32)
    wm = 2 if weight_range is not None else _QMODES[weight_dtype]
    wlo, whi = (float(weight_range[0]), float(weight_range[1])) if weight_range is not None else (0.0, 0.0)
    _call("fq_qconv_weights_prepare_host", w, cin, cout, _i(kh), _i(kw), _i(sh), _i(sw), _i(ph), _i(pw), _i(groups), _i(wm),
          wlo, whi, wbuf, None, None)
    ho, wo = (h + 2 * ph - kh) // sh + 1, (wd + 2 * pw - kw) // sw + 1
    y = np.empty((n, cout, ho, wo), F32)
    im = 2 if input_range is not None else _QMODES[input_dtype]
    ilo, ihi = (float(input_range[0]), float(input_range[1])) if input_range is not None else (0.0, 0.0)
    stat = np.empty(n, F32) if want_stat else None
    _call("fq_qconv2d_forward_host", x, w, wbuf, None if b is None else _f32(b), y, n, cin, h, wd, cout, _i(kh), _i(kw),
          _i(sh), _i(sw), _i(ph), _i(pw), _i(groups), _i(im), ilo, ihi, None if in_stat is None else _f32(in_stat),
          _i(_ACTS[act]), None if bn_scale is None else _f32(bn_scale), None if bn_shift is None else _f32(bn_shift), stat,
          None, _i(0), None)
    return (y, stat) if want_stat else y
