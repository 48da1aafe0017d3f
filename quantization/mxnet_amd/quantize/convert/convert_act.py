"""`gen_act_converter`, `convert_relu_to_relu6` — reference: quantize/convert/convert_act.py:32-79.

Activation-OUTPUT fake-quant: statistic = mean_n max_{chw} act (no abs, :50), unsigned, and NO epsilon in the divide
(:54) — so an all-zero activation yields NaN exactly like the reference.  Served by the same HIP entry points with
FQ_ACT_NO_ABS | FQ_ACT_NO_EPS.  (The CLI maps `nn.Activation: None`; this exists for API completeness.)"""
import types
from collections import namedtuple

import torch

from ...mx.ndarray import NDArray
from ...mx import autograd
from ...mx.gluon.nn import Activation
from ... import ops
from .._state import DeviceScalar

__all__ = ["convert_relu_to_relu6", 'gen_act_converter']

QuantizedArgs = namedtuple("ActQuantizedArgs", "width quantize_act")


def _relu6_forward(self, F, x):
    return F.clip(F.Activation(x, act_type=self._act_type, name='fwd'), 0., 6.)


def convert_relu_to_relu6(m):
    assert isinstance(m, Activation) and m._act_type == "relu"
    m.hybrid_forward = types.MethodType(_relu6_forward, m)


def _act_forward(self, F, x, act_max=None):
    # Normal Activation
    act = self.origin_forward(F, x)

    # Simulate quantization (:49-54)
    if self.enable_quantize and self.quantize_args.quantize_act:
        t = act._t if act._t.is_contiguous() else act._t.contiguous()
        cur = getattr(self, "_fq_cur", None)
        if cur is None or cur.device != t.device:
            cur = torch.zeros(1, dtype=torch.float32, device=t.device)
            self._fq_cur = cur
        flags = ops.act_flags(no_abs=True, no_eps=True)
        if self.quantize_act:
            if self.quantize_act_offline:
                y, _, _ = ops.fake_quant_offline(t, act_max._t, self.quantize_args.width, flags, cur_out=cur)
            else:
                y, _, _ = ops.fake_quant_online(t, self.quantize_args.width, flags, cur_out=cur)
            act = NDArray(autograd.ste_link(t, y))                # identity backward; no-op unless recording
        else:
            ops.batch_mean(ops.absmax_per_sample(t, no_abs=True), out=cur)
        self.current_act_max = DeviceScalar(cur)

    return act


def _add_quantize_act_params(m):
    m.quantize_act_offline = False
    m.current_act_max = 0.
    m.act_max = m.params.get("act_max",
                             shape=(1,), init="zeros",
                             allow_deferred_init=True,
                             differentiable=False)


def gen_act_converter(width=8, quantize_act=True):
    def _converter(m):
        assert isinstance(m, Activation)

        _add_quantize_act_params(m)

        m.origin_forward = m.hybrid_forward
        m.hybrid_forward = types.MethodType(_act_forward, m)
        m.quantize_args = QuantizedArgs(width=width, quantize_act=quantize_act)
        m.enable_quantize = True
        m.quantize_act = quantize_act
    return _converter
