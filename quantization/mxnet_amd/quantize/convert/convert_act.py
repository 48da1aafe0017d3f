"""`gen_act_converter(width=8, quantize_act=True)` and `convert_relu_to_relu6(block)` — reference API:
quantize/convert/convert_act.py:32-79.

Activation-OUTPUT fake-quantisation: the statistic is mean_n max_{chw} act WITHOUT abs (:50), the range is unsigned and
the divide has NO epsilon (:54), so an all-zero activation yields NaN exactly as the reference does.  Served by the same
HIP entry points as the convolution inputs with FQ_ACT_NO_ABS | FQ_ACT_NO_EPS.  (The CLI maps `nn.Activation: None`;
this converter exists for API completeness.)"""
from collections import namedtuple

import torch

from ...mx.ndarray import NDArray
from ...mx import autograd
from ...mx.gluon.nn import Activation
from ... import ops
from .._state import DeviceScalar
from ._blocks import OUTPUT_RANGE, contiguous, rebind_forward, scalar_slot

__all__ = ["convert_relu_to_relu6", 'gen_act_converter']

QuantizedArgs = namedtuple("ActQuantizedArgs", "width quantize_act")

_OUTPUT_FLAGS = dict(no_abs=True, no_eps=True)


def _clipped_relu(self, F, x):
    """relu followed by min(., 6)"""
    return F.clip(F.Activation(x, act_type=self._act_type, name='fwd'), 0., 6.)


def convert_relu_to_relu6(m):
    if not (isinstance(m, Activation) and m._act_type == "relu"):
        raise AssertionError("convert_relu_to_relu6 expects a relu Activation")
    rebind_forward(m, _clipped_relu, keep_origin=False)


def _quantised_activation(self, F, x, act_max=None):
    out = self.origin_forward(F, x)
    args = self.quantize_args
    if not (self.enable_quantize and args.quantize_act):
        return out
    t = contiguous(out._t)
    n = t.shape[0]
    cur, side = scalar_slot(self, t)
    flags = ops.act_flags(**_OUTPUT_FLAGS)
    # batch sharded over ranks (dist.py): the per-sample maxima land in this block's row of the net's statistic matrix;
    # strict mode exchanges them before the apply pass, the default mode reads the rows back at `update_ema`
    rows = getattr(self, "_fq_stat_ws", None)
    if rows is not None and (rows.device != t.device or rows.numel() < n):
        rows = None
    exchange = getattr(self, "_fq_global_stat", None) if rows is not None else None
    y = None
    if exchange is not None or (rows is not None and getattr(self, "_fq_keep_rows", False)):
        stat = ops.absmax_per_sample(t, no_abs=True, out=rows[:n])
        if exchange is not None:
            exchange(stat, n, cur)
            if self.quantize_act:
                y = ops.fake_quant_offline(t, act_max._t if self.quantize_act_offline else cur, args.width, flags,
                                           want_stat=False)[0]
        elif self.quantize_act and not self.quantize_act_offline:
            y = ops.fake_quant_online_prestat(t, stat, args.width, flags, cur_out=cur)[0]
        else:
            ops.batch_mean(stat, out=cur)
            if self.quantize_act:
                y = ops.fake_quant_offline(t, act_max._t, args.width, flags, want_stat=False)[0]
    elif not self.quantize_act:                                # statistic only (the reference still computes it, :50)
        ops.batch_mean(ops.absmax_per_sample(t, no_abs=True), out=cur)
    elif self.quantize_act_offline:
        y = ops.fake_quant_offline(t, act_max._t, args.width, flags, cur_out=cur)[0]
    else:
        y = ops.fake_quant_online(t, args.width, flags, cur_out=cur)[0]
    if y is not None:
        out = NDArray(autograd.ste_link(t, y))                 # identity backward; a no-op unless recording
    self._fq_last_n = n
    if not side:
        self.current_act_max = DeviceScalar(cur)
    return out


def gen_act_converter(width=8, quantize_act=True):
    settings = QuantizedArgs(width=width, quantize_act=quantize_act)

    def _converter(m):
        if not isinstance(m, Activation):
            raise AssertionError("gen_act_converter expects an Activation block")
        OUTPUT_RANGE.attach(m)
        rebind_forward(m, _quantised_activation)
        m.quantize_args = settings
        m.quantize_act = quantize_act
        m.enable_quantize = True
    return _converter
