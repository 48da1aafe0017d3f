"""`gen_dense_converter` — reference: quantize/convert/convert_dense.py:34-97.

Same shape as the conv converter with the reference's two Dense quirks kept: the STE is built WITHOUT clip_min, so
inputs clip to [0, max] even when `input_signed` (:49), and there is no `fixed_params`: Dense weights are
re-quantised on every forward (:52-63).  `group` is rewritten to `channel` (:83-84)."""
import types
from collections import namedtuple

from ...mx.ndarray import NDArray
from ...mx import autograd
from ...mx.gluon.nn import Dense
from ... import ops
from .._state import DeviceScalar
from .convert_conv2d import _fake_quant_input, _cur_slot

__all__ = ['gen_dense_converter']

QuantizedArgs = namedtuple("DenseQuantizedArgs", "in_signed in_width wt_width quantize_input quant_type")


def _dense_forward(self, F, x, weight, bias=None, input_max=None):
    qa = self.quantize_args
    if self.enable_quantize:
        # Quantize input (:40-49)
        if qa.quantize_input:
            flags = ops.act_flags(signed=qa.in_signed, lo_neg_max=False)
            inner_tail = 1
            for s in x.shape[2:]:
                inner_tail *= s
            if inner_tail == 1:
                x = _fake_quant_input(self, x, input_max, flags, qa.in_width)
            else:
                # `F.max(F.abs(x), axis=1)` on an un-flattened (N,C,H,W) input reduces over C only, then `.mean()`
                # averages over N*H*W values (:41): statistic via torch amax (exact), ordered mean + apply in HIP.
                t = x._t.contiguous()
                cur = _cur_slot(self, t)
                ops.batch_mean(t.abs().amax(dim=1).reshape(-1).contiguous(), out=cur)
                if self.quantize_input:
                    thr = input_max._t if self.quantize_input_offline else cur
                    y, _, _ = ops.fake_quant_offline(t, thr, qa.in_width, flags, want_stat=False)
                    x = NDArray(autograd.ste_link(t, y))
                self.current_input_max = DeviceScalar(cur)

        # Simulate quantization for weight (:52-63)
        # The reference recomputes this on every forward; the result only depends on the weight, so it is kept until the
        # parameter's storage or in-place version counter changes (an optimiser step / set_data bumps it): identical
        # values, three launches less per forward.
        wt = weight._t if weight._t.is_contiguous() else weight._t.contiguous()
        key = (weight._t.data_ptr(), weight._t._version, qa.quant_type, qa.wt_width, str(wt.device))
        cache = self.__dict__.get("_fq_wq_cache")
        if cache is None or cache[0] != key:
            groups = self._units if qa.quant_type == 'channel' else 1
            cache = (key, ops.weight_fake_quant(wt, groups, qa.wt_width))
            self.__dict__["_fq_wq_cache"] = cache
        weight_q = NDArray(autograd.ste_link(wt, cache[1]))       # identity backward; no-op unless recording
    else:
        weight_q = weight

    # Normal dense (:68) — rocBLAS through torch
    act = self.origin_forward(F, x, weight_q, bias)

    return act


def _add_quantize_input_params(m):
    m.quantize_input_offline = False
    m.current_input_max = 0.
    m.input_max = m.params.get("input_max",
                               shape=(1,), init="zeros",
                               allow_deferred_init=True,
                               differentiable=False)


def gen_dense_converter(weight_width=8, input_signed=False, input_width=8, quantize_input=True, quant_type='layer'):
    if quant_type == "group":
        quant_type = "channel"

    def _converter(m):
        assert isinstance(m, Dense)

        if quantize_input:
            _add_quantize_input_params(m)
        m.origin_forward = m.hybrid_forward
        m.hybrid_forward = types.MethodType(_dense_forward, m)
        m.quantize_args = QuantizedArgs(in_signed=input_signed, in_width=input_width, wt_width=weight_width,
                                        quantize_input=quantize_input, quant_type=quant_type)
        m.enable_quantize = True
        m.quantize_input = quantize_input
    return _converter
