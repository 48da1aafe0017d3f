"""`gen_conv2d_converter` — reference: quantize/convert/convert_conv2d.py:38-177.

The patched `hybrid_forward` keeps the reference's control flow line for line (fake-BN fold :47-51, activation branch
:53-66, weight branch :68-99 with layer/group/channel and the Winograd-domain variant, the `fixed_params` state machine
:101-105, `origin_forward` :108).  What changes is WHERE the arithmetic runs:

  reference                                                   here
  ---------                                                   ----
  F.max(F.abs(x),axis=(1,2,3)).mean().asscalar()  (:56)       fq_fake_quant_online / _offline: statistic + scale + STE in
  LinearQuantizeSTE(in_scale, max_, min_)(x)      (:66)       two (online) / one (offline) fused HIP passes; the statistic
    = clip, div, round, mul as 4 NDArray passes               stays in a device scalar (no per-layer host sync)
  weight.abs().reshape((num,-1)).max(axis=1) ...  (:70-95)    fq_weight_fake_quant(rows = 1 | G | Cout)
  nd.dot(G, w^T) ..., np.linalg.pinv per call     (:71-83)    fq_wino_weight_fake_quant (pinv cached per variant)
"""
import types
from collections import namedtuple

import torch

from ...mx import nd
from ...mx import autograd
from ...mx.ndarray import NDArray
from ...mx.gluon.nn import Conv2D
from ... import ops
from .._state import DeviceScalar

__all__ = ['gen_conv2d_converter']

QuantizedArgs = namedtuple("ConvQuantizedArgs",
                           "quantize_input in_signed in_width "
                           "wt_width quant_type "
                           "fake_bn wino_quantize")


def _cur_slot(m, like):
    """(1,) device tensor receiving this block's `current_input_max` (a slice of the net's arena once
    `net.update_ema()` has bound one — convert.py)."""
    t = getattr(m, "_fq_cur", None)
    if t is None or t.device != like.device:
        t = torch.zeros(1, dtype=torch.float32, device=like.device)
        m._fq_cur = t
    return t


def _fake_quant_input(m, x, input_max, flags, width):
    """Activation branch shared by Conv2D and Dense (convert_conv2d.py:55-66, convert_dense.py:40-49)."""
    x_in = x
    t = x._t if x._t.is_contiguous() else x._t.contiguous()
    cur = _cur_slot(m, t)
    stat_ws = getattr(m, "_fq_stat_ws", None)
    if stat_ws is not None and (stat_ws.device != t.device or stat_ws.numel() < t.shape[0]):
        stat_ws = None
    gstat = getattr(m, "_fq_global_stat", None)
    # per-sample max|x| already produced by a fused producer (quantize/fuse.py)?  Exactly what the statistic pass
    # would compute, so using it changes no result — it only removes a pass over x.
    hint = x._fq_stat if x._t is t else None
    if hint is not None and (hint.numel() != t.shape[0] or hint.device != t.device):
        hint = None
    if gstat is not None and stat_ws is not None:
        # batch sharded over ranks (dist.py): statistic pass -> all-gather -> GLOBAL batch mean -> apply pass
        n = t.shape[0]
        per_sample = hint if hint is not None else ops.absmax_per_sample(t, out=stat_ws[:n])
        gstat(per_sample, n, cur)
        if m.quantize_input:
            thr = input_max._t if m.quantize_input_offline else cur
            y, _, _ = ops.fake_quant_offline(t, thr, width, flags, want_stat=False)
            x = NDArray(y)
    elif hint is not None:
        if m.quantize_input:
            if m.quantize_input_offline:
                y, _, _ = ops.fake_quant_offline(t, input_max._t, width, flags, want_stat=False)
                ops.batch_mean(hint, out=cur)
            else:
                y, _, _ = ops.fake_quant_online_prestat(t, hint, width, flags, cur_out=cur)
            x = NDArray(y)
        else:
            ops.batch_mean(hint, out=cur)
    elif m.quantize_input:
        if m.quantize_input_offline:
            y, _, _ = ops.fake_quant_offline(t, input_max._t, width, flags, cur_out=cur,
                                             want_stat=getattr(m, "track_input_stat", True), stat_ws=stat_ws)
        else:
            y, _, _ = ops.fake_quant_online(t, width, flags, cur_out=cur, stat_ws=stat_ws)
        x = NDArray(y)
    else:
        # the reference still computes the statistic whenever quantize_args.quantize_input is set (:55-56)
        per_sample = ops.absmax_per_sample(t, out=None if stat_ws is None else stat_ws[:t.shape[0]])
        ops.batch_mean(per_sample, out=cur)
    m._fq_last_n = t.shape[0]
    m.current_input_max = DeviceScalar(cur)
    if x is not x_in:
        # under autograd.record(): straight-through link to the un-quantised input (ste_func.py:43-44, identity)
        x = NDArray(autograd.ste_link(t, x._t))
    return x


def _fused_input_params(m, x, input_max, flags, width):
    """Activation branch when the convolution itself quantises on load (quantize/fuse.py, depthwise 3x3): produce the
    statistic / threshold exactly as `_fake_quant_input` would, but NO apply pass.  Returns kwargs for ops.dwconv3x3."""
    t = x._t if x._t.is_contiguous() else x._t.contiguous()
    n = t.shape[0]
    cur = _cur_slot(m, t)
    stat_ws = getattr(m, "_fq_stat_ws", None)
    if stat_ws is not None and (stat_ws.device != t.device or stat_ws.numel() < n):
        stat_ws = None
    gstat = getattr(m, "_fq_global_stat", None)
    hint = x._fq_stat if x._t is t else None
    if hint is not None and (hint.numel() != n or hint.device != t.device):
        hint = None
    online = m.quantize_input and not m.quantize_input_offline
    need_stat = online or gstat is not None or getattr(m, "track_input_stat", True)
    per_sample = None
    if need_stat:
        per_sample = hint if hint is not None else \
            ops.absmax_per_sample(t, out=None if stat_ws is None else stat_ws[:n])
    kw = {}
    if gstat is not None and stat_ws is not None:
        gstat(per_sample, n, cur)
        if m.quantize_input:
            kw = dict(in_thr=input_max._t if m.quantize_input_offline else cur, width=width, flags=flags)
    elif online:
        kw = dict(in_stat=per_sample, width=width, flags=flags, cur_out=cur)      # the kernel writes cur
    else:
        if per_sample is not None:
            ops.batch_mean(per_sample, out=cur)
        if m.quantize_input:
            kw = dict(in_thr=input_max._t, width=width, flags=flags)
    m._fq_last_n = n
    m.current_input_max = DeviceScalar(cur)
    return kw


def _dw_fused_conv(m, x, weight_q, bias, quant_kw):
    """Depthwise 3x3 through fq_dwconv3x3: quantise-on-load + the BatchNorm / activation that followed this block +
    the per-sample statistic of the output for the next fake-quant."""
    fz = m._fq_dw_fused
    t = x._t if x._t.is_contiguous() else x._t.contiguous()
    w = weight_q._t if weight_q._t.is_contiguous() else weight_q._t.contiguous()
    b = None if bias is None else bias._t
    scale, shift = fz["constants"]() if fz["bn"] is not None else (None, None)
    y, stat = ops.dwconv3x3(t, w, b, stride=m._kwargs["stride"][0], bn_scale=scale, bn_shift=shift, act=fz["act"],
                            **quant_kw)
    out = NDArray(y)
    out._fq_stat = stat
    return out


def _rows_per_scale(m, qa):
    cout = m._kwargs["num_filter"]
    if qa.quant_type == "channel":
        return 1
    if qa.quant_type == "group" and m._kwargs["num_group"] == cout:
        return 1
    return cout


def _pw_fused_conv(m, F, x, weight_raw, weight_q, bias, quant_kw, weights_quantised):
    """1x1 convolution taken over by quantize/fuse.py.  When both operands are quantised to <= 8 bits the convolution
    runs on the integer codes (fq_pwconv_i8: exact int32 sums on the int8 matrix cores, quantise-on-load, BN / activation
    / statistic on store); otherwise the library convolution runs and only BN + activation + statistic are fused."""
    fz = m._fq_pw_fused
    qa = m.quantize_args
    scale, shift = fz["constants"]() if fz["bn"] is not None else (None, None)
    int8_ok = bool(quant_kw) and weights_quantised and qa.in_width <= 8 and qa.wt_width <= 8 \
        and not getattr(m, "_fq_no_int8", False)
    if int8_ok:
        t = x._t if x._t.is_contiguous() else x._t.contiguous()
        cache = getattr(m, "_fq_pw_cache", None)
        if m.fixed_params == 1 and cache is not None and cache[3] == m.weight.data()._t.data_ptr():
            codes, scales, rowsum = cache[:3]
        else:
            # codes of the weights being used by THIS forward: the raw weights while they are (re-)quantised every
            # forward or being frozen right now, the frozen (already fake-quantised) ones afterwards
            src = weight_raw if m.fixed_params != 1 or cache is None else weight_q
            wsrc = src._t if src._t.is_contiguous() else src._t.contiguous()
            codes, scales, rowsum = ops.weight_codes(wsrc, _rows_per_scale(m, qa), qa.wt_width)
            if m.fixed_params == 1:
                m._fq_pw_cache = (codes, scales, rowsum, m.weight.data()._t.data_ptr())
        b = None if bias is None else bias._t
        y, stat = ops.pwconv_i8(t, codes, scales, rowsum, b, bn_scale=scale, bn_shift=shift, act=fz["act"], **quant_kw)
    else:
        if quant_kw:          # input is to be quantised but the integer path does not apply: explicit apply pass
            t = x._t if x._t.is_contiguous() else x._t.contiguous()
            if "in_stat" in quant_kw:
                yq, _, _ = ops.fake_quant_online_prestat(t, quant_kw["in_stat"], quant_kw["width"], quant_kw["flags"],
                                                         cur_out=quant_kw.get("cur_out"))
            else:
                yq, _, _ = ops.fake_quant_offline(t, quant_kw["in_thr"], quant_kw["width"], quant_kw["flags"],
                                                  want_stat=False)
            x = NDArray(yq)
        out = m.origin_forward(F, x, weight_q, bias)
        if fz["bn"] is None and fz["act"] == "none":
            return out
        c = out.shape[1]
        if scale is None:
            scale = torch.ones(c, dtype=torch.float32, device=out._t.device)
            shift = torch.zeros(c, dtype=torch.float32, device=out._t.device)
        y, stat = ops.bn_act_stat(out._t.contiguous(), scale, shift, fz["act"])
    res = NDArray(y)
    res._fq_stat = stat
    return res


def _conv2d_forward(self, F, x, weight, bias=None, input_max=None,
                    gamma=None, beta=None, running_mean=None, running_var=None):
    qa = self.quantize_args
    fz = getattr(self, "_fq_dw_fused", None)
    fzp = getattr(self, "_fq_pw_fused", None)
    if (fz is not None or fzp is not None) and autograd.is_recording():
        raise RuntimeError("this net was rewired by quantize.fuse.fuse_inference (inference only): call "
                           "quantize.fuse.unfuse(net) before recording gradients")
    weight_raw = weight
    quant_kw = {}
    # Fake bn (:47-51)
    if self.fixed_params != 1 and qa.fake_bn:
        w_shape = weight.shape
        cout = w_shape[0]
        weight = (weight.reshape(cout, -1) * gamma.reshape(-1, 1) /
                  F.sqrt(running_var + 1e-10).reshape(-1, 1)).reshape(w_shape)
        bias = gamma * (bias - running_mean) / F.sqrt(running_var + 1e-10) + beta

    if self.enable_quantize:
        # Quantize input (:55-66)
        if qa.quantize_input:
            if fz is None and fzp is None:
                x = _fake_quant_input(self, x, input_max, ops.act_flags(signed=qa.in_signed), qa.in_width)
            else:
                quant_kw = _fused_input_params(self, x, input_max, ops.act_flags(signed=qa.in_signed), qa.in_width)

        # Simulate quantization for weight (:68-99)
        if self.fixed_params != 1:
            wt = weight._t if weight._t.is_contiguous() else weight._t.contiguous()
            if qa.quant_type == 'channel':
                if qa.wino_quantize != 'none' and self._kwargs['kernel'] == (3, 3):
                    wq = ops.wino_weight_fake_quant(wt, qa.wino_quantize, qa.wt_width)
                    wq = autograd.wino_link(wt, wq, *ops.winograd_matrices(qa.wino_quantize))
                    wt = None                                          # already linked (with the transform's gradient)
                else:
                    wq = ops.weight_fake_quant(wt, self._kwargs['num_filter'], qa.wt_width)
            elif qa.quant_type == 'group':
                num = self._kwargs['num_group']
                if num not in (1, wt.shape[0]):
                    # the reference broadcasts a (G,1,1,1) scale against (Cout,Cin/g,kh,kw): MXNet raises here too
                    raise ValueError("group-wise weight quantisation needs num_group in {1, num_filter} "
                                     "(got num_group=%d, num_filter=%d): operands could not be broadcast"
                                     % (num, wt.shape[0]))
                wq = ops.weight_fake_quant(wt, num, qa.wt_width)
            else:
                wq = ops.weight_fake_quant(wt, 1, qa.wt_width)
            # identity backward (ste_func.py:43-44); a no-op unless autograd is recording
            weight_q = NDArray(wq if wt is None else autograd.ste_link(wt, wq))
        else:
            weight_q = weight
    else:
        weight_q = weight

    # Freeze (:101-105)
    if self.fixed_params == 0:
        self.fixed_params = 1
        self.weight.set_data(weight_q)
        if bias is not None:
            self.bias.set_data(bias)

    # Normal convolution (:108) — MIOpen through torch; not the path this project replaces
    if fz is not None:
        return _dw_fused_conv(self, x, weight_q, bias, quant_kw)
    if fzp is not None:
        return _pw_fused_conv(self, F, x, weight_raw, weight_q, bias, quant_kw, bool(self.enable_quantize))
    act = self.origin_forward(F, x, weight_q, bias)

    return act


def _add_quantize_input_params(m):
    m.quantize_input_offline = False
    m.current_input_max = 0.
    m.input_max = m.params.get("input_max",
                               shape=(1,), init="zeros",
                               allow_deferred_init=True,
                               differentiable=False)


def _add_fake_bn_params(m):
    in_channels = m._kwargs['num_filter']
    m.gamma = m.params.get('gamma',
                           shape=(in_channels,), init="ones",
                           allow_deferred_init=True,
                           differentiable=True)
    m.beta = m.params.get('beta',
                          shape=(in_channels,), init="zeros",
                          allow_deferred_init=True,
                          differentiable=True)
    m.running_mean = m.params.get('running_mean',
                                  shape=(in_channels,),
                                  init="zeros",
                                  allow_deferred_init=True,
                                  differentiable=False)
    m.running_var = m.params.get('running_var',
                                 shape=(in_channels,),
                                 init="ones",
                                 allow_deferred_init=True,
                                 differentiable=False)


def _add_fake_bn_ema_hook(m):
    @torch.no_grad()
    def _ema_hook(m, x):
        x = x[0]
        weight = m.weight.data()
        bias = nd.zeros(shape=weight.shape[0], ctx=weight.context) if m.bias is None else m.bias.data()
        y = nd.Convolution(x, weight, bias, **m._kwargs)
        num_samples = y.shape[0] * y.shape[2] * y.shape[3]
        m.current_mean = y.sum(axis=(0, 2, 3)) / num_samples
        diff_square = (y - m.current_mean.reshape(1, -1, 1, 1)) ** 2
        m.current_var = diff_square.sum(axis=(0, 2, 3)) / num_samples
    m.register_forward_pre_hook(_ema_hook)


def gen_conv2d_converter(weight_width=8, quant_type="layer",
                         quantize_input=True, input_signed=False, input_width=8,
                         fake_bn=False, wino_quantize="none"):
    assert wino_quantize in ("none", "F23", "F43", "F63")

    def _converter(m):
        assert isinstance(m, Conv2D)

        if quantize_input:
            _add_quantize_input_params(m)
        if fake_bn:
            _add_fake_bn_params(m)
            _add_fake_bn_ema_hook(m)
        m.origin_forward = m.hybrid_forward
        m.hybrid_forward = types.MethodType(_conv2d_forward, m)
        m.quantize_args = QuantizedArgs(in_signed=input_signed, in_width=input_width, wt_width=weight_width,
                                        quantize_input=quantize_input, fake_bn=fake_bn, quant_type=quant_type,
                                        wino_quantize=wino_quantize)
        m.fixed_params = -1
        m.enable_quantize = True
        m.quantize_input = quantize_input
    return _converter
