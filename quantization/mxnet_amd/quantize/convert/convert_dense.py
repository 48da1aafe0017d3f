"""`gen_dense_converter(weight_width=8, input_signed=False, input_width=8, quantize_input=True, quant_type='layer')` —
reference API and arithmetic: quantize/convert/convert_dense.py:34-97.

Two reference quirks are kept: the input STE is built WITHOUT clip_min, so inputs clip to [0, max] even when
`input_signed` (:49), and there is no `fixed_params`: Dense weights are quantised on every forward (:52-63; here the
result is reused until the parameter changes — identical values).  `group` means `channel` for a Dense (:83-84)."""
from collections import namedtuple

from ...mx.ndarray import NDArray
from ...mx import autograd
from ...mx.gluon.nn import Dense
from ... import ops
from .._state import DeviceScalar
from ._blocks import INPUT_RANGE, contiguous, rebind_forward, scalar_slot
from .convert_conv2d import fake_quant_block_input, fused_input_plan, current_slot

__all__ = ['gen_dense_converter']

QuantizedArgs = namedtuple("DenseQuantizedArgs", "in_signed in_width wt_width quantize_input quant_type")


def _quantise_unflattened_input(block, x, input_max, flags, width):
    """`F.max(F.abs(x), axis=1)` on an (N, C, H, W) input reduces over C only and `.mean()` then averages N*H*W values
    (:41): statistic through torch's exact amax, ordered mean and apply pass in HIP."""
    t = contiguous(x._t)
    cur, side = scalar_slot(block, t)
    ops.batch_mean(t.abs().amax(dim=1).reshape(-1).contiguous(), out=cur)
    if not side:
        block.current_input_max = DeviceScalar(cur)
    if not block.quantize_input:
        return x
    threshold = input_max._t if block.quantize_input_offline else cur
    y = ops.fake_quant_offline(t, threshold, width, flags, want_stat=False)[0]
    return NDArray(autograd.ste_link(t, y))


def _quantised_weight(block, weight, args):
    """Per-layer or per-unit fake-quantised weight, kept until the parameter's storage or in-place version changes
    (an optimiser step / set_data bumps it)."""
    w = contiguous(weight._t)
    key = (weight._t._version, args.quant_type, args.wt_width)
    held = block.__dict__.get("_fq_wq_cache")
    if held is None or held[0] is not weight._t or held[1] != key:      # (the entry holds its source: no address reuse)
        rows = block._units if args.quant_type == 'channel' else 1
        held = block.__dict__["_fq_wq_cache"] = (weight._t, key, ops.weight_fake_quant(w, rows, args.wt_width))
    return NDArray(autograd.ste_link(w, held[2]))              # identity backward; a no-op unless recording


def _dense_weight_codes(block, weight, args):
    """int8 codes / scales / row sums of the (units, in_units) weight, kept until the parameter changes."""
    w = contiguous(weight._t)
    key = (weight._t._version, args.quant_type, args.wt_width)
    held = block.__dict__.get("_fq_wcodes_cache")
    if held is None or held[0] is not weight._t or held[1] != key:
        rows = 1 if args.quant_type == 'channel' else block._units
        held = block.__dict__["_fq_wcodes_cache"] = (weight._t, key, ops.weight_codes(w, rows, args.wt_width))
    return held[2]


def _dense_on_codes(block, x, weight, bias, input_max, args, flags):
    """quantize/fuse.py: a Dense whose input and weight are both quantised to <= 8 bits IS a 1x1 convolution on a 1x1 plane:
    fq_pwconv_i8 quantises on load (no apply pass) and sums the integer codes exactly (the reference: fp32 FullyConnected of
    the two fake-quantised operands)."""
    plan = fused_input_plan(block, x, input_max, flags, args.in_width)
    codes, scales, rowsum = _dense_weight_codes(block, weight, args)
    t = contiguous(x._t)
    head = block.__dict__.get("_fq_eval_head")
    if head is not None and head.labels is not None and plan and t.shape[1] % 4 == 0:
        # the evaluation loop's counters in the same launch (quantize/fuse.py: EvalHead)
        y = ops.dense_i8_eval(t.reshape(t.shape[0], -1), codes, scales, rowsum, head.labels, head.counters,
                              bias=None if bias is None else bias._t, **plan)
        head.labels, head.counted = None, True
        return NDArray(y)
    y, _ = ops.pwconv_i8(t.reshape(t.shape[0], -1, 1, 1), codes, scales, rowsum, None if bias is None else bias._t,
                         want_stat=False, **plan)
    return NDArray(y.reshape(y.shape[0], -1))


def _quantised_dense(self, F, x, weight, bias=None, input_max=None):
    args = self.quantize_args
    if not self.enable_quantize:
        return self.origin_forward(F, x, weight, bias)
    if getattr(self, "_fq_dense_int8", False) and args.quantize_input and self.quantize_input \
            and args.in_width <= 8 and args.wt_width <= 8 and all(d == 1 for d in x.shape[2:]) \
            and self.act is None and not autograd.is_recording():
        flags = ops.act_flags(signed=args.in_signed, lo_neg_max=False)          # [0, max] always: the clip_min quirk
        return _dense_on_codes(self, x, weight, bias, input_max, args, flags)
    if args.quantize_input:
        flags = ops.act_flags(signed=args.in_signed, lo_neg_max=False)          # [0, max] always: the clip_min quirk
        flattened = all(d == 1 for d in x.shape[2:])
        if flattened:
            x = fake_quant_block_input(self, x, input_max, flags, args.in_width)
        else:
            x = _quantise_unflattened_input(self, x, input_max, flags, args.in_width)
    return self.origin_forward(F, x, _quantised_weight(self, weight, args), bias)    # rocBLAS through torch


def gen_dense_converter(weight_width=8, input_signed=False, input_width=8, quantize_input=True, quant_type='layer'):
    settings = QuantizedArgs(in_signed=input_signed, in_width=input_width, wt_width=weight_width,
                             quantize_input=quantize_input,
                             quant_type="channel" if quant_type == "group" else quant_type)

    def _converter(m):
        if not isinstance(m, Dense):
            raise AssertionError("gen_dense_converter expects a Dense block")
        if quantize_input:
            INPUT_RANGE.attach(m)
        rebind_forward(m, _quantised_dense)
        m.quantize_args = settings
        m.quantize_input = quantize_input
        m.enable_quantize = True
    return _converter
