"""TEST INFRASTRUCTURE: run the product's HOST logic (converters, state machine, calibration loops, CLI) on CPU by
temporarily replacing the HIP entry points of `quantization.mxnet_amd.ops` with the numpy oracle.

Used only by tests/ (host-logic and whole-net parity tests) and by bench.py's `cpu_baseline` leg.  The product never
imports this module, never calls `oracle_ops()` and has no switch that could route it here; without this context
manager every product entry point raises on CPU tensors.
"""
import contextlib

import numpy as np
import torch

from . import fq_oracle as O

F32 = np.float32


def _np(t):
    return t.detach().cpu().numpy()


def _t(a, like=None, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t


def _flags(flags):
    return bool(flags & 1), bool(flags & 2), bool(flags & 4), bool(flags & 8)


def _stat(x, no_abs):
    a = _np(x).astype(F32)
    a = a.reshape(a.shape[0], -1)
    return (a.max(axis=1) if no_abs else np.abs(a).max(axis=1)).astype(F32)


def _apply(x, max_, width, flags):
    signed, lo_neg, no_abs, no_eps = _flags(flags)
    scale = O.act_scale(max_, signed, width)
    lo = F32(-max_) if lo_neg else F32(0)
    codes = O.ste_codes(_np(x), scale, max_, lo, eps=F32(0) if no_eps else O.EPS)
    return (codes * scale).astype(F32), codes


def absmax_per_sample(x, no_abs=False, out=None):
    r = _t(_stat(x, no_abs))
    if out is not None:
        out.copy_(r)
        return out
    return r


def batch_mean(v, out=None):
    r = _t(np.asarray([O.batch_mean(_np(v).reshape(-1))], dtype=F32))
    if out is not None:
        out.copy_(r)
        return out
    return r


def batch_mean_gathered(packs, out=None):
    a = _np(packs)
    vals = np.concatenate([rec[1:1 + int(rec[0])] for rec in a])
    r = _t(np.asarray([O.batch_mean(vals)], dtype=F32))
    if out is not None:
        out.copy_(r)
        return out
    return r


def stat_rows_sum(stats, n, out=None):
    a = _np(stats).astype(np.float64)
    rec = np.zeros(a.shape[0] + 1, np.float64)
    for r in range(a.shape[0]):
        acc = np.float64(0.0)
        for v in a[r, :int(n)]:
            acc += v
        rec[r] = acc
    rec[-1] = float(n)
    r_ = torch.from_numpy(rec)
    if out is not None:
        out.copy_(r_)
        return out
    return r_


def mean_from_sums(sums, out=None):
    a = _np(sums)
    r_ = _t((a[:-1].astype(F32) / F32(a[-1])).astype(F32))
    if out is not None:
        out.copy_(r_)
        return out
    return r_


def batch_mean_rows(v, out=None):
    r = _t(np.asarray([O.batch_mean(row) for row in _np(v)], dtype=F32))
    if out is not None:
        out.copy_(r)
        return out
    return r


def fake_quant_online(x, width=8, flags=0, out=None, cur_out=None, want_codes=False, stat_ws=None):
    per = _stat(x, bool(flags & 4))
    if stat_ws is not None:
        stat_ws[:len(per)].copy_(_t(per))
    cur = O.batch_mean(per)
    y, codes = _apply(x, cur, width, flags)
    yt = _t(y).reshape(x.shape)
    if out is not None:
        out.copy_(yt)
        yt = out
    ct = _t(np.asarray([cur], dtype=F32))
    if cur_out is not None:
        cur_out.copy_(ct)
        ct = cur_out
    return yt, ct, (_t(codes.astype(np.int32)).reshape(x.shape) if want_codes else None)


def fake_quant_online_prestat(x, stat, width=8, flags=0, out=None, cur_out=None, want_codes=False):
    per = _np(stat).reshape(-1)[:x.shape[0]].astype(F32)
    cur = O.batch_mean(per)
    y, codes = _apply(x, cur, width, flags)
    yt = _t(y).reshape(x.shape)
    if out is not None:
        out.copy_(yt)
        yt = out
    ct = _t(np.asarray([cur], dtype=F32))
    if cur_out is not None:
        cur_out.copy_(ct)
        ct = cur_out
    return yt, ct, (_t(codes.astype(np.int32)).reshape(x.shape) if want_codes else None)


def _into_sink(yt, hist):
    """the producers that bin what they store (fq_*_stat_hist): the separate histogram pass over the result, by definition.
    Resolved through the patched module so that the host-twin stand-ins (`host_ops`) take over when they are installed."""
    if hist is not None:
        from quantization.mxnet_amd import ops as _ops
        _ops.histogram_accumulate(yt, hist.fm_max, hist.hist, hist.neg)


def bn_act_stat(x, scale, shift, act="relu", out=None, want_stat=True, hist=None):
    a = _np(x)
    y = O.bn_act(a, _np(scale), _np(shift), act)
    stat = _t(np.abs(y).reshape(y.shape[0], -1).max(axis=1).astype(F32)) if want_stat else None
    yt = _t(y)
    _into_sink(yt, hist)
    return yt, stat


def bn_act_maxpool_stat(x, scale, shift, act="relu", want_stat=True):
    y = O.bn_act_maxpool(_np(x), _np(scale), _np(shift), act)
    stat = _t(np.abs(y).reshape(y.shape[0], -1).max(axis=1).astype(F32)) if want_stat else None
    return _t(y), stat


def add_act_stat(a, b, act="relu", out=None, want_stat=True, hist=None):
    y = (_np(a).astype(F32) + _np(b).astype(F32)).astype(F32)
    if act == "relu":
        y = np.maximum(y, F32(0))
    elif act == "relu6":
        y = np.minimum(np.maximum(y, F32(0)), F32(6))
    yt = _t(y)
    if out is not None:
        out.copy_(yt)
        yt = out
    _into_sink(yt, hist)
    return yt, (_t(_stat(yt, False)) if want_stat else None)


def global_avg_pool_stat(x, want_stat=True):
    y = O.global_avg_pool(_np(x))
    stat = _t(np.abs(y).reshape(y.shape[0], -1).max(axis=1).astype(F32)) if want_stat else None
    return _t(y), stat


def gemm_i8_codes(xcodes, wcodes, n, l, zoff):
    return torch.from_numpy(O.gemm_i8_codes(xcodes.cpu().numpy(), wcodes.cpu().numpy(), n, l, zoff))


def eval_counters(logits, labels, counters):
    counters.copy_(_t(O.eval_counters(_np(logits), labels.cpu().numpy(), _np(counters))))
    return counters


def stem_conv_s2(x, w, bias=None, bn_scale=None, bn_shift=None, act=None, want_stat=True, w_tap_major=None, pool=False):
    assert not pool, "the oracle's first convolution does not pool (quantize.fuse.STEM_POOL = False)"
    y = O.stem_conv_s2(_np(x), _np(w), None if bias is None else _np(bias),
                         None if bn_scale is None else _np(bn_scale), None if bn_shift is None else _np(bn_shift),
                         None if act in (None, "none") else act)
    stat = _t(np.abs(y).reshape(y.shape[0], -1).max(axis=1).astype(F32)) if want_stat else None
    return _t(y), stat


stem_conv3x3s2 = stem_conv_s2


def dwconv3x3(x, w, bias=None, stride=1, in_stat=None, in_thr=None, width=8, flags=0, cur_out=None, bn_scale=None,
              bn_shift=None, act=None, want_stat=True):
    signed, lo_neg, _, _ = _flags(flags)
    in_max = None
    if in_stat is not None:
        in_max = O.batch_mean(_np(in_stat).reshape(-1)[:x.shape[0]])
        if cur_out is not None:
            cur_out.copy_(_t(np.asarray([in_max], dtype=F32)))
    if in_thr is not None:                    # offline (the statistic, when given too, only fed cur_out)
        in_max = F32(_np(in_thr).reshape(-1)[0])
    y = O.dwconv3x3(_np(x), _np(w), None if bias is None else _np(bias), stride, in_max, signed, width, lo_neg,
                    None if bn_scale is None else _np(bn_scale), None if bn_shift is None else _np(bn_shift),
                    None if act in (None, "none") else act)
    stat = _t(np.abs(y).reshape(y.shape[0], -1).max(axis=1).astype(F32)) if want_stat else None
    return _t(y), stat


def weight_codes(w, rows_per_scale, width=8):
    a = _np(w)
    codes, scales = O.weight_codes(a, rows_per_scale, width)
    rows, row_len = codes.shape
    rp, cp = (rows + 63) // 64 * 64, (row_len + 63) // 64 * 64
    padded = np.zeros((rp, cp), np.int8)
    padded[:rows, :row_len] = codes.astype(np.int8)
    return _t(padded), _t(scales), _t(codes.sum(axis=1).astype(np.int32))


def pwconv_strided_supported(cin):
    return True


def pwconv_sub2_supported(cin, cout):
    return cout > 128


def pwconv_gap_supported(xshape, cout, residual=False):
    return False                              # (the oracle pools in a pass of its own)


def pwconv_shortcut_supported(cin, cin2, cout):
    return cout % 256 == 0


def pwconv_i8_shortcut(x, wcodes, wscale, wsum, bias=None, in_stat=None, in_thr=None, width=8, flags=0, cur_out=None,
                       bn_scale=None, bn_shift=None, act=None, x2=None, wcodes2=None, wscale2=None, wsum2=None, in_stat2=None,
                       in_thr2=None, width2=8, flags2=0, cur_out2=None, bn_scale2=None, bn_shift2=None):
    """fq_pwconv_i8_shortcut: the shortcut convolution (+ BatchNorm, no activation), then the closing one with it as residual"""
    s, _ = pwconv_i8(x2, wcodes2, wscale2, wsum2, None, in_stat=in_stat2, in_thr=in_thr2, width=width2, flags=flags2,
                     cur_out=cur_out2, bn_scale=bn_scale2, bn_shift=bn_shift2, act=None, want_stat=False)
    return pwconv_i8(x, wcodes, wscale, wsum, bias, in_stat=in_stat, in_thr=in_thr, width=width, flags=flags, cur_out=cur_out,
                     bn_scale=bn_scale, bn_shift=bn_shift, act=act, residual=s)


def pwconv_i8(x, wcodes, wscale, wsum, bias=None, in_stat=None, in_thr=None, width=8, flags=0, cur_out=None,
              bn_scale=None, bn_shift=None, act=None, want_stat=True, form=None, stride=1, residual=None, subsample=False):
    """(subsample: the statistic of the whole output, then its even pixels of its even rows - fq_pwconv_i8_sub2)"""
    signed, lo_neg, _, _ = _flags(flags)
    if stride != 1:
        x = x[:, :, ::stride, ::stride].contiguous()
    if in_stat is not None:
        in_max = O.batch_mean(_np(in_stat).reshape(-1)[:x.shape[0]])
        if cur_out is not None:
            cur_out.copy_(_t(np.asarray([in_max], dtype=F32)))
    if in_thr is not None:                    # offline (the statistic, when given too, only fed cur_out)
        in_max = F32(_np(in_thr).reshape(-1)[0])
    a = _np(x)
    sx = O.act_scale(in_max, signed, width)
    cx = O.ste_codes(a, sx, in_max, F32(-in_max) if lo_neg else F32(0)).astype(np.int64)
    sw = _np(wscale)
    cout, cin = sw.size, a.shape[1]
    cw = _np(wcodes)[:cout, :cin].astype(np.int64)
    n = a.shape[0]
    isum = np.einsum("oc,ncp->nop", cw, cx.reshape(n, cin, -1))
    y = (isum.astype(F32) * (F32(sx) * sw).astype(F32)[None, :, None]).astype(F32)
    if bias is not None:
        y = (y + _np(bias)[None, :, None]).astype(F32)
    y = y.reshape((n, cout) + a.shape[2:])
    if residual is not None:
        if bn_scale is not None:
            y = O.bn_act(y, _np(bn_scale), _np(bn_shift), "none")
        y = (y + _np(residual)).astype(F32)
        bn_scale = None
    if bn_scale is not None:
        y = O.bn_act(y, _np(bn_scale), _np(bn_shift), act or "none")
    elif act == "relu":
        y = np.maximum(y, F32(0))
    elif act == "relu6":
        y = np.minimum(np.maximum(y, F32(0)), F32(6))
    stat = _t(np.abs(y).reshape(n, -1).max(axis=1).astype(F32)) if want_stat else None
    if subsample:
        y = np.ascontiguousarray(y[:, :, ::2, ::2])
    return _t(y.astype(F32)), stat


def dense_i8_eval(x, wcodes, wscale, wsum, labels, counters, bias=None, in_stat=None, in_thr=None, width=8, flags=0,
                  cur_out=None):
    y, _ = pwconv_i8(x.reshape(x.shape[0], -1, 1, 1), wcodes, wscale, wsum, bias=bias, in_stat=in_stat, in_thr=in_thr,
                     width=width, flags=flags, cur_out=cur_out, want_stat=False)
    y = y.reshape(y.shape[0], -1)
    eval_counters(y, labels, counters)
    return y


def weight_codes_3x3(w, rows_per_scale, width=8):
    return weight_codes(w.permute(0, 2, 3, 1).contiguous(), rows_per_scale, width)


def weight_slices_3x3(w):
    """(codes [3, 2 * rows_pad * row_pad] int8 with the row-major digits in the first half of every slice, pscale, rowsum)"""
    wp = _np(w).transpose(0, 2, 3, 1)
    rows = wp.shape[0]
    flat = wp.reshape(rows, -1)
    row_pad = (flat.shape[1] + 63) // 64 * 64
    rows_pad = (rows + 63) // 64 * 64
    m, p = O.weight_slices(flat)
    codes = np.zeros((3, 2 * rows_pad * row_pad), np.int8)
    rowsum = np.zeros((3, rows), np.int32)
    for sl, d in enumerate(O.slice_digits(m)):
        codes[sl, :rows_pad * row_pad].reshape(rows_pad, row_pad)[:rows, :flat.shape[1]] = d.astype(np.int8)
        rowsum[sl] = d.sum(axis=1)
    return _t(codes), _t(p), _t(rowsum)


def conv3x3_i8(x, wcodes, wscale, wsum, bias=None, in_stat=None, in_thr=None, width=8, flags=0, cur_out=None,
               bn_scale=None, bn_shift=None, act=None, want_stat=True):
    """codes rows are ordered (tap, ci) (weight_codes_3x3); three digit slices when they come from weight_slices_3x3"""
    signed, lo_neg, _, _ = _flags(flags)
    if wcodes.dim() == 2 and wcodes.shape[0] == 3 and wsum.dim() == 2:
        if in_stat is not None:
            in_max = O.batch_mean(_np(in_stat).reshape(-1)[:x.shape[0]])
            if cur_out is not None:
                cur_out.copy_(_t(np.asarray([in_max], dtype=F32)))
        if in_thr is not None:
            in_max = F32(_np(in_thr).reshape(-1)[0])
        a = _np(x)
        n, cin, h, w = a.shape
        p = _np(wscale)
        cout = p.size
        rows_pad = (cout + 63) // 64 * 64
        d = [_np(wcodes)[sl, :rows_pad * 9 * cin].reshape(rows_pad, 9 * cin)[:cout].astype(np.int64) for sl in range(3)]
        m = ((d[0] << 14) + (d[1] << 7) + d[2]).reshape(cout, 3, 3, cin).transpose(0, 3, 1, 2)
        wt = (m.astype(np.float64) * p.astype(np.float64)[:, None, None, None]).astype(F32)      # exact: |m| <= 2^20
        y = O.conv3x3_i8_sliced(a, wt, in_max, signed=signed, width=width, lo_neg_max=lo_neg,
                                bias=None if bias is None else _np(bias),
                                bn_scale=None if bn_scale is None else _np(bn_scale),
                                bn_shift=None if bn_shift is None else _np(bn_shift), act=act)
        stat = _t(np.abs(y).reshape(n, -1).max(axis=1).astype(F32)) if want_stat else None
        return _t(y.astype(F32)), stat
    if in_stat is not None:
        in_max = O.batch_mean(_np(in_stat).reshape(-1)[:x.shape[0]])
        if cur_out is not None:
            cur_out.copy_(_t(np.asarray([in_max], dtype=F32)))
    if in_thr is not None:                    # offline (the statistic, when given too, only fed cur_out)
        in_max = F32(_np(in_thr).reshape(-1)[0])
    a = _np(x)
    n, cin, h, w = a.shape
    sx = O.act_scale(in_max, signed, width)
    cx = O.ste_codes(a, sx, in_max, F32(-in_max) if lo_neg else F32(0)).astype(np.int64)
    sw = _np(wscale)
    cout = sw.size
    cw = _np(wcodes)[:cout, :9 * cin].astype(np.int64).reshape(cout, 3, 3, cin)
    pad = np.zeros((n, cin, h + 2, w + 2), np.int64)
    pad[:, :, 1:-1, 1:-1] = cx
    isum = np.zeros((n, cout, h, w), np.int64)
    for ky in range(3):
        for kx in range(3):
            isum += np.einsum("oc,nchw->nohw", cw[:, ky, kx, :], pad[:, :, ky:ky + h, kx:kx + w])
    y = (isum.astype(F32) * (F32(sx) * sw).astype(F32)[None, :, None, None]).astype(F32)
    if bias is not None:
        y = (y + _np(bias)[None, :, None, None]).astype(F32)
    if bn_scale is not None:
        y = O.bn_act(y, _np(bn_scale), _np(bn_shift), act or "none")
    elif act == "relu":
        y = np.maximum(y, F32(0))
    elif act == "relu6":
        y = np.minimum(np.maximum(y, F32(0)), F32(6))
    stat = _t(np.abs(y).reshape(n, -1).max(axis=1).astype(F32)) if want_stat else None
    return _t(y.astype(F32)), stat


def fake_quant_offline(x, threshold, width=8, flags=0, out=None, cur_out=None, want_stat=True, want_codes=False,
                       stat_ws=None):
    thr = F32(_np(threshold).reshape(-1)[0])
    y, codes = _apply(x, thr, width, flags)
    yt = _t(y).reshape(x.shape)
    if out is not None:
        out.copy_(yt)
        yt = out
    ct = None
    if want_stat:
        per = _stat(x, bool(flags & 4))
        if stat_ws is not None:
            stat_ws[:len(per)].copy_(_t(per))
        ct = _t(np.asarray([O.batch_mean(per)], dtype=F32))
        if cur_out is not None:
            cur_out.copy_(ct)
            ct = cur_out
    return yt, ct, (_t(codes.astype(np.int32)).reshape(x.shape) if want_codes else None)


def ste_forward(x, scales, clip_max=None, clip_min=None, eps=1e-10, out=None):
    a = _np(x)
    s = _np(scales).reshape(-1).astype(F32)
    rows = s.size
    y = O.ste_forward(a.reshape(rows, -1), s.reshape(rows, 1), clip_max, clip_min, eps=F32(eps)).reshape(a.shape)
    return _t(y)


def weight_fake_quant(w, rows, width=8, out=None, want_scales=False):
    a = _np(w)
    wq, sc = O.weight_fake_quant(a.reshape(rows, -1), "layer" if rows == 1 else "channel", width)
    wq = _t(wq.reshape(a.shape))
    return (wq, _t(sc)) if want_scales else wq


def wino_weight_fake_quant(w, variant, width=8, out=None, want_scales=False, GI=None, GTI=None):
    wq, sc, _ = O.wino_weight_fake_quant(_np(w), variant, width, GI, GTI)
    return (_t(wq), _t(sc)) if want_scales else _t(wq)


def ema_update(state, current, momentum=0.9):
    state.copy_(_t(O.ema_update(_np(state), _np(current), momentum)))
    return state


def global_max(x):
    return _t(np.asarray([_np(x).max()], dtype=F32))


def histogram_accumulate(x, max_dev, hist, neg_count=None):
    a = _np(x).reshape(-1)
    if neg_count is not None:
        neg_count += int((a < 0).sum())
    mx = F32(_np(max_dev)[0])
    if mx > 0:          # (the device kernel has no assert; the caller checks max > 0 once, at the end)
        h, _ = O.discrete_histogram(np.maximum(a, 0), hist.numel(), mx)
        hist += _t(h.astype(np.int64))
    return hist


def hist_to_float(hist):
    return hist.to(torch.float32)


def kl_search(hist, levels, min_bins):
    h = _np(hist)
    if h.ndim == 1:
        h = h.reshape(1, -1)
    return _t(np.asarray([O.kl_calibrate(r, levels, min_bins, h.shape[1]) for r in h], dtype=np.int32))


def quantize_codes(x, out_type="int8", range_dev=None):
    a = _np(x)
    if out_type in ("int8", "uint8"):
        codes, scale = O.quantize_codes(a, out_type)
        mx = np.abs(a).max() if out_type == "int8" else a.max()
        mn = -mx if out_type == "int8" else a.min()
        rng = _t(np.asarray([mn, mx, scale], dtype=F32))
    elif out_type == "range":
        r = _np(range_dev)
        codes, scale = O.quantize_codes(a, fixed_range=(r[0], r[1]))
        rng = _t(np.asarray([r[0], r[1], scale], dtype=F32))
    else:
        r = _np(range_dev)
        c = np.minimum(np.maximum(a, r[0]), r[1])
        codes = O.roundf((c / F32(r[2])).astype(F32)).astype(np.int32)
        rng = range_dev
    return _t(codes), rng


def dequantize(codes, scale_dev):
    return _t(O.dequantize(_np(codes), F32(_np(scale_dev).reshape(-1)[0])))


class _QconvWeights(object):
    """stand-in for the opaque device buffer of ops.qconv_weights"""

    def __init__(self, dtype, rng):
        self.dtype, self.rng = dtype, rng


def qconv_weights(w, strides, padding, groups, weight_dtype="int8", weight_range=None):
    if weight_range is None and weight_dtype not in ("int8", "uint8"):
        raise ValueError("unknown out type: %s" % (weight_dtype,))
    return _QconvWeights(weight_dtype, weight_range)


def qconv_workspace(cout, device):
    return torch.zeros(1)


def qconv2d(x, w, wbuf, bias, strides, padding, groups, ws, input_dtype="uint8", input_range=None, act="none", in_stat=None,
            bn_scale=None, bn_shift=None, want_stat=False, force_direct=False, out=None):
    if input_range is None and input_dtype not in ("int8", "uint8"):
        raise ValueError("unknown out type: %s" % (input_dtype,))
    y = O.qconv2d_forward(_np(x), _np(w), None if bias is None else _np(bias), tuple(strides), tuple(padding), groups,
                          input_dtype=input_dtype, weight_dtype=wbuf.dtype, input_range=input_range, weight_range=wbuf.rng,
                          act=None if act == "none" else act, in_stat=None if in_stat is None else _np(in_stat),
                          bn_scale=None if bn_scale is None else _np(bn_scale),
                          bn_shift=None if bn_shift is None else _np(bn_shift))
    if want_stat:
        return _t(y), _t(np.abs(y).reshape(y.shape[0], -1).max(axis=1).astype(F32))
    return _t(y)


def require_hip(device, what="tensor"):
    return None


def default_device(what="this call"):
    return torch.device("cpu")


_REPLACED = ["require_hip", "default_device", "add_act_stat", "stat_rows_sum", "mean_from_sums", "batch_mean_rows", "batch_mean_gathered", "fake_quant_online_prestat",
             "bn_act_stat", "bn_act_maxpool_stat", "eval_counters", "dense_i8_eval", "gemm_i8_codes", "global_avg_pool_stat", "stem_conv3x3s2", "stem_conv_s2", "dwconv3x3", "weight_codes", "pwconv_i8", "pwconv_strided_supported", "pwconv_sub2_supported", "pwconv_gap_supported", "pwconv_shortcut_supported", "pwconv_i8_shortcut", "weight_codes_3x3", "weight_slices_3x3", "conv3x3_i8", "absmax_per_sample", "batch_mean", "fake_quant_online", "fake_quant_offline", "ste_forward",
             "weight_fake_quant", "wino_weight_fake_quant", "ema_update", "global_max", "histogram_accumulate",
             "hist_to_float", "kl_search", "quantize_codes", "dequantize", "qconv_weights", "qconv_workspace", "qconv2d"]


@contextlib.contextmanager
def oracle_ops():
    """`with oracle_ops(): ...` — product host code runs against the CPU oracle inside the block only."""
    from quantization.mxnet_amd import ops
    saved = {name: getattr(ops, name) for name in _REPLACED}
    g = globals()
    for name in _REPLACED:
        setattr(ops, name, g[name])
    try:
        yield
    finally:
        for name, fn in saved.items():
            setattr(ops, name, fn)


# ---- the same stand-in with the C++/OpenMP restatement doing the heavy passes (bench.py's cpu_baseline leg) -------------------
def _host_fns():
    from . import host as H

    def _a(t):
        return t.detach().contiguous().numpy()

    def h_absmax_per_sample(x, no_abs=False, out=None):
        r = torch.from_numpy(H.absmax_per_sample(_a(x), no_abs))
        if out is not None:
            out.copy_(r)
            return out
        return r

    def _finish(y, cur, codes, out, cur_out, want_codes):
        yt = torch.from_numpy(y)
        if out is not None:
            out.copy_(yt)
            yt = out
        ct = torch.tensor([cur], dtype=torch.float32) if cur is not None else None
        if cur_out is not None and ct is not None:
            cur_out.copy_(ct)
        return yt, (cur_out if cur_out is not None else ct), (torch.from_numpy(codes) if want_codes else None)

    def h_fake_quant_online(x, width=8, flags=0, out=None, cur_out=None, want_codes=False, stat_ws=None):
        y, cur, codes = H.fake_quant_online(_a(x), width, flags, want_codes)
        return _finish(y, cur, codes, out, cur_out, want_codes)

    def h_fake_quant_online_prestat(x, stat, width=8, flags=0, out=None, cur_out=None, want_codes=False):
        y, cur, codes = H.fake_quant_online_prestat(_a(x), _a(stat), width, flags, want_codes)
        return _finish(y, cur, codes, out, cur_out, want_codes)

    def h_fake_quant_offline(x, threshold, width=8, flags=0, out=None, cur_out=None, want_stat=True, want_codes=False,
                             stat_ws=None):
        want = cur_out is not None and want_stat
        y, cur, codes = H.fake_quant_offline(_a(x), float(_a(threshold).reshape(-1)[0]), width, flags, want_codes, want)
        return _finish(y, cur, codes, out, cur_out if want else None, want_codes)

    def h_weight_fake_quant(w, rows, width=8, out=None, want_scales=False):
        wq, scales = H.weight_fake_quant(_a(w), int(rows), width)
        wq = torch.from_numpy(wq)
        if out is not None:
            out.copy_(wq)
            wq = out
        return (wq, torch.from_numpy(scales)) if want_scales else wq

    def h_qconv2d(x, w, wbuf, bias, strides, padding, groups, ws, input_dtype="uint8", input_range=None, act="none",
                  in_stat=None, bn_scale=None, bn_shift=None, want_stat=False, force_direct=False, out=None):
        r = H.qconv2d_forward(_a(x), _a(w), None if bias is None else _a(bias), tuple(strides), tuple(padding), groups,
                              input_dtype=input_dtype, weight_dtype=wbuf.dtype, input_range=input_range,
                              weight_range=wbuf.rng, act=None if act == "none" else act,
                              bn_scale=None if bn_scale is None else _a(bn_scale),
                              bn_shift=None if bn_shift is None else _a(bn_shift), want_stat=want_stat)
        if want_stat:
            return torch.from_numpy(r[0]), torch.from_numpy(r[1])
        return torch.from_numpy(r)

    def h_eval_counters(logits, labels, counters):
        counters.copy_(torch.from_numpy(H.eval_counters(_a(logits), _a(labels), _a(counters))))
        return counters

    def h_global_max(x):
        return torch.tensor([H.global_max(_a(x))], dtype=torch.float32)

    def h_histogram_accumulate(x, max_dev, hist, neg_count=None):
        mx = float(_a(max_dev).reshape(-1)[0])
        if mx > 0:
            h, neg = H.histogram_accumulate(_a(x).reshape(-1), mx, hist.numel())
            hist += torch.from_numpy(h.astype(np.int64))
            if neg_count is not None:
                neg_count += int(neg)
        return hist

    def h_kl_search(hist, levels, min_bins):
        return torch.from_numpy(H.kl_search(_a(hist), levels, min_bins))

    return {"global_max": h_global_max, "histogram_accumulate": h_histogram_accumulate, "kl_search": h_kl_search,
            "absmax_per_sample": h_absmax_per_sample, "fake_quant_online": h_fake_quant_online,
            "fake_quant_online_prestat": h_fake_quant_online_prestat, "fake_quant_offline": h_fake_quant_offline,
            "weight_fake_quant": h_weight_fake_quant, "eval_counters": h_eval_counters, "qconv2d": h_qconv2d}


@contextlib.contextmanager
def host_ops():
    """`with host_ops(): ...` — like oracle_ops(), with the per-element passes done by oracle/libfq_host.so on all host
    cores (same results, pinned in tests/test_host_oracle.py); everything else stays with the numpy restatement."""
    from quantization.mxnet_amd import ops
    fast = _host_fns()
    with oracle_ops():
        saved = {name: getattr(ops, name) for name in fast}
        for name, fn in fast.items():
            setattr(ops, name, fn)
        try:
            yield
        finally:
            for name, fn in saved.items():
                setattr(ops, name, fn)
