"""`gen_conv2d_converter(weight_width=8, quant_type="layer", quantize_input=True, input_signed=False, input_width=8,
fake_bn=False, wino_quantize="none")` — reference API and arithmetic: quantize/convert/convert_conv2d.py:38-177.

What the converted block computes is the reference's, step for step: fake-BN fold (:47-51), activation branch (:53-66),
weight branch per layer / group / channel with the Winograd-domain variant (:68-99), the `fixed_params` state machine
-1 -> 0 -> 1 (:101-105, :174) and then the stock convolution (:108).  Where it runs is different:

  reference                                                   here
  ---------                                                   ----
  F.max(F.abs(x),axis=(1,2,3)).mean().asscalar()  (:56)       fq_fake_quant_online / _offline: statistic + scale + STE in
  LinearQuantizeSTE(in_scale, max_, min_)(x)      (:66)       two (online) / one (offline) fused HIP passes; the statistic
    = clip, div, round, mul as 4 NDArray passes               stays in a device scalar (no per-layer host sync)
  weight.abs().reshape((num,-1)).max(axis=1) ...  (:70-95)    fq_weight_fake_quant(rows = 1 | G | Cout)
  nd.dot(G, w^T) ..., np.linalg.pinv per call     (:71-83)    fq_wino_weight_fake_quant (pinv cached per variant)

After `quantize.fuse.fuse_inference` a depthwise 3x3 / pointwise 1x1 block hands its input straight to the convolution
kernel that quantises on load (`depthwise_fused`, `pointwise_fused` below) instead of running an apply pass.
"""
from collections import namedtuple

import os

import torch

from ...mx import nd
from ...mx import autograd
from ...mx.ndarray import NDArray
from ...mx.gluon.nn import Conv2D
from ... import ops
from .._state import DeviceScalar
from ._blocks import INPUT_RANGE, BatchNormTerms, contiguous, rebind_forward, scalar_slot

__all__ = ['gen_conv2d_converter']

QuantizedArgs = namedtuple("ConvQuantizedArgs",
                           "quantize_input in_signed in_width "
                           "wt_width quant_type "
                           "fake_bn wino_quantize")

WINOGRAD_VARIANTS = ("F23", "F43", "F63")


# ---- activation branch ---------------------------------------------------------------------------------------------------
def current_slot(block, like):
    """(1,) device tensor receiving this block's `current_input_max` (`_blocks.scalar_slot`)."""
    return scalar_slot(block, like)[0]


class _InputView(object):
    """What the activation branch needs to know about one input: the contiguous tensor, where the batch statistic goes,
    the per-sample statistic when a fused producer already computed it (bit-identical to the statistic pass, so using it
    changes no result), and the multi-GPU calibration hooks of dist.py."""

    def __init__(self, block, x):
        self.t = contiguous(x._t)
        self.n = self.t.shape[0]
        self.cur, self.side = scalar_slot(block, self.t)
        rows = getattr(block, "_fq_stat_ws", None)
        if rows is not None and (rows.device != self.t.device or rows.numel() < self.n):
            rows = None
        self.rows = rows                                            # this block's row of the net's statistic matrix
        self.exchange = getattr(block, "_fq_global_stat", None)     # strict mode: all-gather before the apply pass
        hint = x._fq_stat if x._t is self.t else None
        if hint is not None and (hint.numel() != self.n or hint.device != self.t.device):
            hint = None
        self.hint = hint

    def per_sample(self):
        """max|x[n]| for every sample; lands in the net's statistic matrix when one is bound (calibration under dist.py
        reads the rows back for its single collective)."""
        if self.hint is None:
            return ops.absmax_per_sample(self.t, out=None if self.rows is None else self.rows[:self.n])
        if self.rows is not None:
            self.rows[:self.n].copy_(self.hint)
            return self.rows[:self.n]
        return self.hint

    def finish(self, block):
        if self.side:                                    # a batch in flight beside others: nothing of the block changes
            return
        block._fq_last_n = self.n
        block.current_input_max = DeviceScalar(self.cur)


def fake_quant_block_input(block, x, input_max, flags, width):
    """Activation branch shared by Conv2D and Dense (convert_conv2d.py:55-66, convert_dense.py:40-49): produces
    `current_input_max` in every mode and the fake-quantised input when the block quantises it."""
    v = _InputView(block, x)
    wanted = block.quantize_input
    offline = block.quantize_input_offline
    y = None
    if v.exchange is not None and v.rows is not None:
        # batch sharded over ranks, strict mode: statistic -> all-gather -> GLOBAL batch mean -> apply pass
        v.exchange(v.per_sample(), v.n, v.cur)
        if wanted:
            y = ops.fake_quant_offline(v.t, input_max._t if offline else v.cur, width, flags, want_stat=False)[0]
    elif v.hint is not None or (v.rows is not None and getattr(block, "_fq_keep_rows", False)):
        stat = v.per_sample()
        if wanted and not offline:
            y = ops.fake_quant_online_prestat(v.t, stat, width, flags, cur_out=v.cur)[0]
        else:
            ops.batch_mean(stat, out=v.cur)
            if wanted:
                y = ops.fake_quant_offline(v.t, input_max._t, width, flags, want_stat=False)[0]
    elif not wanted:
        # the reference still computes the statistic whenever quantize_args.quantize_input is set (:55-56)
        ops.batch_mean(v.per_sample(), out=v.cur)
    elif offline:
        y = ops.fake_quant_offline(v.t, input_max._t, width, flags, cur_out=v.cur, stat_ws=v.rows,
                                   want_stat=getattr(block, "track_input_stat", True))[0]
    else:
        y = ops.fake_quant_online(v.t, width, flags, cur_out=v.cur, stat_ws=v.rows)[0]
    v.finish(block)
    if y is None:
        return x
    # under autograd.record(): straight-through link to the un-quantised input (ste_func.py:43-44, identity)
    return NDArray(autograd.ste_link(v.t, y))


def fused_input_plan(block, x, input_max, flags, width):
    """Activation branch when the convolution itself quantises on load (quantize/fuse.py): the statistic / threshold are
    produced exactly as `fake_quant_block_input` would, but there is NO apply pass.  Returns the keyword arguments that
    tell ops.dwconv3x3 / ops.pwconv_i8 how to quantise ({} = do not)."""
    v = _InputView(block, x)
    wanted = block.quantize_input
    online = wanted and not block.quantize_input_offline
    stat = None
    if online or v.exchange is not None or getattr(block, "track_input_stat", True):
        stat = v.per_sample()
    plan = {}
    if v.exchange is not None and v.rows is not None:
        v.exchange(stat, v.n, v.cur)
        if wanted:
            plan = dict(in_thr=input_max._t if block.quantize_input_offline else v.cur, width=width, flags=flags)
    elif online:
        plan = dict(in_stat=stat, width=width, flags=flags, cur_out=v.cur)         # the kernel writes `cur`
    elif wanted:
        # offline: the stored threshold quantises; the kernel derives `current_input_max` from the statistic on the side
        # (one wavefront of its first workgroup) instead of a separate one-workgroup launch in front of every block
        plan = dict(in_thr=input_max._t, width=width, flags=flags)
        if stat is not None:
            plan.update(in_stat=stat, cur_out=v.cur)
    elif stat is not None:
        ops.batch_mean(stat, out=v.cur)
    v.finish(block)
    return plan


# ---- convolutions taken over by quantize/fuse.py ---------------------------------------------------------------------------
def depthwise_fused(block, x, weight_q, bias, plan):
    """Depthwise 3x3 through fq_dwconv3x3: quantise-on-load + the BatchNorm / activation that followed this block + the
    per-sample statistic of the output for the next fake-quant.  Between two 1x1 convolutions under offline input
    quantisation (a MobileNetV2 unit) it reads the codes the expansion wrote and writes the codes the projection reads
    (fq_dwconv3x3_c16)."""
    fz = block._fq_dw_fused
    scale, shift = fz["constants"]() if fz["bn"] is not None else (None, None)
    c16_in = getattr(x, "_fq_c16", None)
    if c16_in is not None:
        out_codes = handover_target(block, fz) if "in_thr" in plan else None
        if out_codes is not None:
            yc, stat = ops.dwconv3x3_c16(c16_in, contiguous(weight_q._t), None if bias is None else bias._t,
                                         stride=block._kwargs["stride"][0], bn_scale=scale, bn_shift=shift, act=fz["act"],
                                         out_codes=out_codes, **plan)
            out = NDArray(yc.t)
            out._fq_c16 = yc
            out._fq_stat = stat
            return out
        x = NDArray(codes16_to_fake_quant(c16_in))      # (the producer's hand-over cannot be honoured after all)
        plan = {}
    y, stat = ops.dwconv3x3(contiguous(x._t), contiguous(weight_q._t), None if bias is None else bias._t,
                            stride=block._kwargs["stride"][0], bn_scale=scale, bn_shift=shift, act=fz["act"], **plan)
    out = NDArray(y)
    out._fq_stat = stat
    return out


def _weight_rows_per_scale(block, args):
    cout = block._kwargs["num_filter"]
    per_channel = args.quant_type == "channel" or (args.quant_type == "group" and block._kwargs["num_group"] == cout)
    return 1 if per_channel else cout


def _pointwise_weight_codes(block, args, weight_raw, weight_q):
    """int8 codes / scales / row sums of the weights THIS forward uses, or None when the integer path must not be taken.
    While the weights are (re-)quantised every forward, or being frozen right now, they come from the raw weights — the
    same division the fake-quant performs.  Frozen codes are kept until the parameter's storage or in-place version
    changes (set_data / load_parameters / reset_ctx / a second fix_params).  After such a change the block is still frozen
    and its parameter already holds fake-quantised values (the reference convolves with them as they are,
    convert_conv2d.py:96-97): codes re-derived from them are accepted only if code * scale reproduces the parameter bit for
    bit; otherwise the block leaves the integer path (library convolution of the frozen weights)."""
    param_t = block.weight.data()._t
    key = (param_t.data_ptr(), param_t._version)
    held = getattr(block, "_fq_pw_cache", None)
    if block.fixed_params == 1 and held is not None and held[3] == key:
        return held[:3]
    three = block._fq_pw_fused.get("kind") == "3x3"
    if block._fq_pw_fused.get("sliced"):
        # Winograd-domain quantised 3x3 filters (convert_conv2d.py:71-83): the filter the convolution multiplies is
        # weight_q itself (already GI U^ GTI), cut into three int8 digit slices - nothing to verify, the slices ARE the
        # filter to p / 2 <= 2^-20 of its channel maximum (include/fakequant.h at fq_weight_slices)
        codes = ops.weight_slices_3x3(contiguous(weight_q._t))
        if block.fixed_params == 1:
            block._fq_pw_cache = tuple(codes) + (key,)
        return codes
    make = ops.weight_codes_3x3 if three else ops.weight_codes
    rederived = block.fixed_params == 1 and held is not None
    src = contiguous((weight_q if rederived else weight_raw)._t)
    live = None if rederived else block.__dict__.get("_fq_pw_live")
    if live is not None and live[0] is weight_raw._t and live[1] == weight_raw._t._version:
        codes = live[2]                                # not frozen (calibration): the codes of this very parameter version
    else:
        codes = make(src, _weight_rows_per_scale(block, args), args.wt_width)
        if not rederived and block.fixed_params != 1 and weight_raw._t is param_t:
            # kept while the parameter is the same tensor in the same in-place version, as `_fake_quant_weight` does
            block.__dict__["_fq_pw_live"] = (weight_raw._t, weight_raw._t._version, codes)
    if block.fixed_params == 1:
        block.__dict__.pop("_fq_pw_live", None)
    if rederived:
        ref = src.permute(0, 2, 3, 1).contiguous() if three else src
        if not ops.weight_codes_reproduce(ref, codes[0], codes[1]):
            block._fq_no_int8 = True
            block.__dict__.pop("_fq_pw_cache", None)
            return None
    if block.fixed_params == 1:
        block._fq_pw_cache = tuple(codes) + (key,)
    return codes


# Smallest output plane on which a depthwise layer takes (and hands on) codes.  Alone on the GPU fq_dwconv3x3_c16 loses to the
# flat fp32 form below 56x56 (23.0 vs 17.8 us at 14x14: fixed costs, not bytes); with several batches in flight those fixed
# costs are covered by the other batches' kernels and the halved bytes win on every plane (MobileNetV2 offline, three in
# flight: 87.1 k images/s with 3136, 90.7 k with 784, 93.6 k with 196, 94.1 k with 49; one batch at a time: 72.1 -> 71.1 k).
_DW_C16_MIN_PIXELS = int(os.environ.get("FQ_DW_C16_MIN_PIXELS", "1"))


def _consumer_takes_codes(nxt):
    a = nxt.quantize_args
    return (nxt.enable_quantize and a.quantize_input and nxt.quantize_input and nxt.quantize_input_offline
            and a.in_width <= 8 and a.wt_width <= 8 and not getattr(nxt, "_fq_no_int8", False)
            and getattr(nxt, "input_max", None) is not None and getattr(nxt, "_fq_global_stat", None) is None)


def _hooked(b):
    return b is not None and bool(getattr(b, "_forward_hooks", None) or getattr(b, "_forward_pre_hooks", None))


def handover_target(block, fz=None):
    """The consumer this fused convolution may hand integer codes to (quantize/fuse.py links `next`), when the consumer will
    quantise with its STORED threshold in this very forward - offline input quantisation (convert_conv2d.py:58 takes
    `input_max`) - and runs on the integer codes itself.  A depthwise consumer (fq_dwconv3x3_c16 reads AND writes codes)
    counts only while it can hand over to its own consumer.  Returns the keyword `out_codes` of ops.pwconv_i8 /
    conv3x3_i8 / dwconv3x3_c16, or None."""
    from .. import fuse as _fuse
    fz = fz if fz is not None else block._fq_pw_fused          # (the first convolution passes its own record)
    nxt = fz.get("next")
    if nxt is None or not _fuse.HANDOVER or autograd.is_recording() or not _consumer_takes_codes(nxt):
        return None
    # a user hook on the producer, on the consumer or on a block bypassed between them (folded BatchNorm, bypassed
    # activation) would be shown a C16 code tensor where it expects the fp32 activation (collect_feature_maps hooks `x[0]` of
    # every quantised block): no hand-over past a hook
    if any(_hooked(b) for b in (block, nxt, fz.get("bn"), fz.get("act_block")) + tuple(fz.get("via", ()))):
        return None
    dw = getattr(nxt, "_fq_dw_fused", None)
    if dw is not None and (handover_target(nxt, dw) is None or not fz.get("c16_pays", True)):
        return None
    a = nxt.quantize_args
    return dict(thr=nxt.input_max.data()._t, width=a.in_width, flags=ops.act_flags(signed=a.in_signed))


def _same_quantiser(block, other, plan, codes):
    """True when `block` would quantise the tensor `codes` stands for exactly as `other` (for which the codes were made) does:
    same width and signedness, and stored thresholds of EQUAL VALUE - two blocks that always see the same tensor calibrate
    to the same threshold, naive or KL.  The comparison reads two device scalars: it is made once per (tensor, version) pair
    and remembered (never during a graph capture: the eager forwards in front of it have asked already)."""
    from .. import fuse as _fuse
    if not _fuse.SIDE_CODES or codes.width != int(plan["width"]) or (codes.flags & 3) != (int(plan["flags"]) & 3):
        return False
    ta, tb = plan["in_thr"], getattr(other, "input_max", None)
    if tb is None:
        return False
    tb = tb.data()._t
    if tb.data_ptr() != codes.thr.data_ptr():
        return False
    # (`ops.state_epoch()`: update_ema writes the thresholds through raw pointers, which no version counter sees)
    key = (ta.data_ptr(), ta._version, tb.data_ptr(), tb._version, ops.state_epoch())
    memo = block.__dict__.get("_fq_same_thr")
    if memo is None or memo[0] != key:
        if ta.is_cuda and torch.cuda.is_current_stream_capturing():
            return False
        memo = block.__dict__["_fq_same_thr"] = (key, bool(torch.equal(ta.reshape(-1)[:1], tb.reshape(-1)[:1])))
    return memo[1]


_THIN_FORMS = os.environ.get("FQ_PWS_THIN", "1") != "0"      # (the library's A/B switch of the thin streaming instantiations)
_SIDE_MAX_CIN = int(os.environ.get("FQ_SIDE_MAX_CIN", "512"))      # (A/B: 128 = the 56x56 and 28x28 stages only, 0 = off)


def side_target(block, c16_in):
    """The first 1x1 of the NEXT residual unit (quantize/fuse.py links `side_next`) when this closing 1x1 - codes in, residual
    operand, fp32 out - may store its output a second time as that block's codes (fq_pwconv_i8_c16_dual): the consumer will
    quantise with its stored threshold in this very forward and runs on the integer codes.  Hooks are no obstacle: the fp32
    tensor stays what every block and hook is handed; the codes ride beside it."""
    from .. import fuse as _fuse
    nxt = block._fq_pw_fused.get("side_next")
    if nxt is None or c16_in is None or not _fuse.SIDE_CODES or not _fuse.HANDOVER or autograd.is_recording():
        return None
    if not _consumer_takes_codes(nxt) or getattr(nxt, "_fq_pw_fused", None) is None or nxt.quantize_args.in_signed:
        return None
    cin, cout = c16_in.shape[1], block._kwargs["num_filter"]
    if cin not in (64, 128, 256, 512) or cin > _SIDE_MAX_CIN or cout % 32 or cout < 256:
        return None
    return nxt


def sub_target(block, fz, xshape):
    """The link to the two readers of this closing 1x1 convolution's output (quantize/fuse.py: `sub_next`, a stage boundary of
    the v1 bottleneck ResNets) when it may store only what they read: each is a 1x1 convolution with stride 2 and no padding
    (convert_conv2d.py:108 with the block's own kwargs), i.e. it looks at y[:, :, ::2, ::2] and at the statistic of y, which this
    launch still takes over all of y.  Nothing else may observe the tensor: no hooks on the blocks between producer and readers,
    no KL collection, inside the forward of the rewired net only.  Returns the link or None."""
    from .. import fuse as _fuse
    link = fz.get("sub_next")
    if link is None or not _fuse.SUBSAMPLE or autograd.is_recording() or _fuse._collection is not None:
        return None
    if getattr(ops.StatArena._tls, "current", None) is None or len(xshape) != 4:
        return None
    for r in link["readers"]:
        k = r._kwargs
        if k["kernel"] != (1, 1) or k["stride"] != (2, 2) or k["pad"] != (0, 0) or getattr(r, "_fq_pw_fused", None) is None:
            return None
    if any(_hooked(b) for b in (block, fz.get("bn"), fz.get("act_block")) + tuple(link["readers"]) + tuple(link["via"])):
        return None
    for r in link["readers"]:                     # (a BatchNorm or activation folded into a reader is bypassed, not hooked)
        rz = r._fq_pw_fused
        if any(_hooked(b) for b in (rz.get("bn"), rz.get("act_block"))):
            return None
    if not ops.pwconv_sub2_supported(xshape[1], block._kwargs["num_filter"]):
        return None
    return link


def gap_target(block, fz, c16_in, xshape, residual):
    """The GlobalAvgPool2D block behind this fused 1x1 convolution (quantize/fuse.py links `gap_next`) when the convolution may hand
    it the plane means instead of the planes (fq_pwconv_i8_gap): nothing else reads the tensor, nothing observes it (no hooks, no
    KL collection), the call is part of the rewired net's forward and the shape is one the pooling epilogue takes."""
    from .. import fuse as _fuse
    link = fz.get("gap_next")
    if link is None or not _fuse.GAP_FUSE or c16_in is not None or autograd.is_recording() or _fuse._collection is not None:
        return None
    if getattr(ops.StatArena._tls, "current", None) is None or block._kwargs["stride"] != (1, 1):
        return None
    if any(_hooked(b) for b in (block, fz.get("bn"), fz.get("act_block"), link["gap"]) + tuple(link["via"])):
        return None
    if not ops.pwconv_gap_supported(xshape, block._kwargs["num_filter"], residual):
        return None
    return link["gap"]


def defer_shortcut(block, fz, x, x_arg, c16_in, codes3, bias, scale, shift, plan, stride_, extra):
    """The record a residual unit's shortcut convolution answers with instead of launching (quantize/fuse.py sets `_fq_defer_short`
    for the one call from `_residual_unit_forward`): its operands, for the closing 1x1 of the unit to compute it inside its own
    launch (fq_pwconv_i8_shortcut).  None when the call must run as it is."""
    from .. import fuse as _fuse
    if not getattr(block, "_fq_defer_short", False):
        return None
    block._fq_defer_short = False
    if extra or bias is not None or scale is None or fz["act"] != "none" or stride_ != 1:
        return None
    if c16_in is not None and "in_thr" not in plan:
        return None
    if autograd.is_recording() or _fuse._collection is not None or getattr(ops.StatArena._tls, "current", None) is None:
        return None
    if any(_hooked(b) for b in (block, fz.get("bn"))) or not plan:
        return None
    xs = tuple(x_arg.shape)
    dev = x_arg.t.device if isinstance(x_arg, ops.Codes16) else x_arg.device
    out = NDArray(_placeholder((xs[0], block._kwargs["num_filter"], xs[2], xs[3]), dev))
    out._fq_short = dict(x=x_arg, codes=codes3, bn=(scale, shift), plan=dict(plan), block=block)
    return out


def materialise_shortcut_record(d):
    """The tensor a deferred shortcut's record stands for: the launch its convolution would have made."""
    return ops.pwconv_i8(d["x"], *d["codes"], None, bn_scale=d["bn"][0], bn_shift=d["bn"][1], act=None, want_stat=False,
                         **d["plan"])[0]


def materialise_shortcut(s):
    """... as the NDArray the unit would have been handed."""
    return NDArray(materialise_shortcut_record(s._fq_short))


def _convolve(block, F, x, weight, bias):
    """convert_conv2d.py:108 - the block's own convolution; of a subsampled trunk (`_fq_sub2`: this block is one of its
    stride-2 1x1 readers) the stride-1 convolution of what was stored, which is the same values."""
    if getattr(x, "_fq_sub2", None) is None:
        return block.origin_forward(F, x, weight, bias)
    out = F.Convolution(x, weight, bias, name="fwd", **dict(block._kwargs, stride=(1, 1)))
    return out if block.act is None else block.act(out)


_PLACEHOLDERS = {}


def _placeholder(shape, device):
    """An int8 tensor of `shape` without storage of its own (one byte, expanded): what a deferred NDArray carries as `_t` - any
    reader but the linked consumer fails on the dtype."""
    one = _PLACEHOLDERS.get(device)
    if one is None:
        one = _PLACEHOLDERS[device] = torch.zeros(1, dtype=torch.int8, device=device)
    return one.expand(*shape)


def recompute_target(block, fz, x, plan, c16_in):
    """The depthwise block behind this fused 1x1 convolution when the pair may run as statistic pass + ONE recomputing launch
    (fq_pwconv_i8_stat + fq_pwdw_fused; quantize/fuse.py links `pair_dw`): both blocks quantise their inputs ONLINE in this very
    forward (convert_conv2d.py:56-58 - the case in which no codes can be handed over), nothing observes the tensor between them
    (no hooks, no KL collection, no multi-GPU exchange of the depthwise block's statistic), no residual operand, and the shape
    is one both kernels take.  Returns the block or None."""
    from .. import fuse as _fuse
    nxt = fz.get("pair_dw")
    if nxt is None or not _fuse.RECOMPUTE or c16_in is not None or autograd.is_recording() or _fuse._collection is not None:
        return None
    # only inside the forward of the net fuse_inference rewired (its wrapper holds the statistic arena): a caller that runs this
    # block on its own - feature extraction, a test - must get a tensor, not a promise only the linked depthwise block can keep
    if getattr(ops.StatArena._tls, "current", None) is None:
        return None
    if "in_stat" not in plan or "in_thr" in plan or getattr(block, "_fq_residual", None) is not None:
        return None
    a = nxt.quantize_args
    if not (nxt.enable_quantize and a.quantize_input and nxt.quantize_input and not nxt.quantize_input_offline
            and getattr(nxt, "_fq_global_stat", None) is None and not a.fake_bn and nxt._kwargs["num_group"] == block._kwargs["num_filter"]):
        return None
    if any(_hooked(b) for b in (block, nxt, fz.get("bn"), fz.get("act_block"))):
        return None
    t = x._t
    if t.dim() != 4 or t.shape[2] * t.shape[3] < _fuse.RECOMPUTE_MIN_PIXELS or t.shape[1] > _fuse.RECOMPUTE_MAX_CIN:
        return None
    if not ops.pwdw_supported(tuple(t.shape), block._kwargs["num_filter"], nxt._kwargs["stride"][0]):
        return None
    return nxt


def _materialise(x):
    """The tensor a deferred NDArray stands for, computed after all (its consumer turned out not to be the linked depthwise
    block in a state that takes it): the storing launch with the arguments the statistic pass had."""
    d = x._fq_deferred
    y, stat = ops.pwconv_i8(d["x"], *d["codes"], d["bias"], bn_scale=d["bn"][0], bn_shift=d["bn"][1], act=d["act"], **d["plan"])
    out = NDArray(y)
    out._fq_stat = stat
    return out


def depthwise_recompute(block, x, weight_q, bias, flags, width):
    """The second half of a recompute pair: `x` is a deferred NDArray (statistic only); this block's activation branch
    (convert_conv2d.py:53-66) and its convolution (:108) run inside fq_pwdw_fused on the 1x1 output recomputed there."""
    d = x._fq_deferred
    fz = block._fq_dw_fused
    scale, shift = fz["constants"]() if fz["bn"] is not None else (None, None)
    src = d["x"]
    n = src.shape[0]
    cur, side = scalar_slot(block, src)
    stat = x._fq_stat
    rows = getattr(block, "_fq_stat_ws", None)
    if rows is not None and rows.device == src.device and rows.numel() >= n:
        rows[:n].copy_(stat)                              # (calibration under dist.py reads the statistic matrix back)
    p = d["plan"]
    z, zstat = ops.pwdw_fused(src, *d["codes"], contiguous(weight_q._t), pw_bias=d["bias"], in_stat=p["in_stat"],
                              width=p["width"], flags=p["flags"], pw_bn_scale=d["bn"][0], pw_bn_shift=d["bn"][1],
                              pw_act=d["act"], mid_stat=stat, mid_width=width, mid_flags=flags, mid_cur_out=cur,
                              dw_bias=None if bias is None else bias._t, stride=block._kwargs["stride"][0],
                              dw_bn_scale=scale, dw_bn_shift=shift, dw_act=fz["act"])
    if not side:
        block._fq_last_n = n
        block.current_input_max = DeviceScalar(cur)
    out = NDArray(z)
    out._fq_stat = zstat
    return out


def _handed_over(x):
    """ops.Codes16 riding on an NDArray a producer handed over, else None."""
    return getattr(x, "_fq_c16", None)


def codes16_to_fake_quant(c16):
    """Fallback for a consumer that received codes but cannot run on them after all: the fake-quantised fp32 tensor the
    codes stand for, code * scale (torch ops; never taken on the BASELINE configurations)."""
    n, c, h, w = c16.shape
    signed = bool(c16.flags & 1)
    b = (c16.t.to(torch.int16) & 255) ^ 0x80
    codes = (b - 128).to(torch.int8).to(torch.float32) if signed else b.to(torch.float32)
    levels = float((1 << (c16.width - 1)) - 1) if signed else float((1 << c16.width) - 1)
    scale = c16.thr.reshape(()) / levels
    t = codes.reshape(n, -1, h * w, 16).permute(0, 1, 3, 2).reshape(n, -1, h, w)[:, :c]
    return (t * scale).contiguous()


def pointwise_fused(block, F, x, weight_raw, weight_q, bias, plan, weights_quantised):
    """1x1 (or dense 3x3, `kind`) convolution taken over by quantize/fuse.py.  When both operands are quantised to <= 8 bits it runs on the
    integer codes (fq_pwconv_i8: exact int32 sums on the int8 matrix cores, quantise-on-load, BN / activation / statistic
    on store); otherwise the library convolution runs and only BN + activation + statistic are fused."""
    fz = block._fq_pw_fused
    args = block.quantize_args
    scale, shift = fz["constants"]() if fz["bn"] is not None else (None, None)
    on_codes = bool(plan) and weights_quantised and args.in_width <= 8 and args.wt_width <= 8 \
        and not getattr(block, "_fq_no_int8", False)
    held = _pointwise_weight_codes(block, args, weight_raw, weight_q) if on_codes else None
    on_codes = held is not None
    res0 = getattr(block, "_fq_residual", None)
    if res0 is not None and res0.get("short") is not None and not (on_codes and fz.get("kind") == "1x1"):
        res0["t"], res0["short"] = materialise_shortcut_record(res0["short"]), None      # (no integer path here: the tensor after all)
    c16_in = _handed_over(x)
    # (one of the two readers of a subsampled trunk: the stored pixels are the ones this block's stride picks)
    stride_ = 1 if getattr(x, "_fq_sub2", None) is not None else block._kwargs["stride"][0]
    if c16_in is None and on_codes and "in_thr" in plan:
        # the trunk of a ResNet arrives as fp32 (`x._t`, which the unit's shortcut reads) with this block's codes of the same
        # values beside it (`_fq_side`: the previous unit's closing 1x1 stored both, fq_pwconv_i8_c16_dual): read 1 B per element
        side = getattr(x, "_fq_side", None)
        if side is not None and not autograd.is_recording():
            if side[0] is block and side[1].matches(plan["in_thr"], plan["width"], plan["flags"]):
                c16_in = side[1]
            elif side[0] is not block and _same_quantiser(block, side[0], plan, side[1]):
                # another reader of the same trunk (the shortcut convolution of a stage's first unit) whose stored threshold
                # has the very value the codes were made with: the same codes are ITS codes
                c16_in = ops.Codes16(side[1].t, side[1].shape, plan["in_thr"], plan["width"], plan["flags"])
    if c16_in is not None and not (on_codes and "in_thr" in plan):
        x = NDArray(codes16_to_fake_quant(c16_in))      # (cannot run on the codes after all: their fp32 meaning, no apply pass)
        plan, c16_in = {}, None
    if on_codes:
        codes, scales, rowsum = held
        x_arg = c16_in if c16_in is not None else contiguous(x._t)
        if fz.get("kind") == "3x3":
            out_codes = None if fz.get("sliced") else handover_target(block)
            y, stat = ops.conv3x3_i8(x_arg, codes, scales, rowsum, None if bias is None else bias._t,
                                     bn_scale=scale, bn_shift=shift, act=fz["act"],
                                     **({} if fz.get("sliced") or out_codes is None else dict(out_codes=out_codes)), **plan)
        else:
            # the tail of a residual unit (quantize/fuse.py): the shortcut is added in this convolution's epilogue
            res = getattr(block, "_fq_residual", None)
            extra = {}
            xshape = c16_in.shape if c16_in is not None else tuple(x._t.shape)
            side_blk = sub_link = None
            if res is not None and res.get("short") is not None:
                # the unit's shortcut arrives as the record of its convolution: both sums in this launch when the shapes are built
                # (fp32 inputs on both sides, BatchNorm here as there), else the shortcut is computed now and added as a tensor
                d = res["short"]
                # (stored thresholds: this convolution reads the 3x3's codes and leaves the code copy for the next unit's first 1x1 -
                # the dual form's pair of outputs; the shortcut convolution's input may be codes as well)
                side_s = side_target(block, c16_in) if c16_in is not None else None
                d16 = isinstance(d["x"], ops.Codes16)
                if (block._kwargs["stride"][0] == 1 and scale is not None and not _hooked(block)
                        and (c16_in is None and not d16 or (side_s is not None and "in_thr" in plan))
                        and tuple(d["x"].shape[2:]) == tuple(xshape[2:]) and d["x"].shape[0] == xshape[0]
                        and ops.pwconv_shortcut_supported(xshape[1], d["x"].shape[1], block._kwargs["num_filter"])
                        and (c16_in is None or block._kwargs["num_filter"] % 512 == 0 or xshape[1] == 64)
                        and not autograd.is_recording()):
                    p2 = d["plan"]
                    kw2 = dict(x2=d["x"], wcodes2=d["codes"][0], wscale2=d["codes"][1], wsum2=d["codes"][2], in_stat2=p2.get("in_stat"),
                               in_thr2=p2.get("in_thr"), width2=p2["width"], flags2=p2["flags"], cur_out2=p2.get("cur_out"),
                               bn_scale2=d["bn"][0], bn_shift2=d["bn"][1])
                    if side_s is not None:
                        a_ = side_s.quantize_args
                        kw2["side_codes"] = dict(thr=side_s.input_max.data()._t, width=a_.in_width, flags=ops.act_flags(signed=a_.in_signed))
                    out = ops.pwconv_i8_shortcut(x_arg, codes, scales, rowsum, None if bias is None else bias._t, bn_scale=scale,
                                                 bn_shift=shift, act=res["act"], **kw2, **plan)
                    res["used"] = True
                    folded = NDArray(out[0])
                    folded._fq_stat = out[1]
                    if side_s is not None:
                        folded._fq_side = (side_s, out[2])
                    return folded
                res["t"] = materialise_shortcut_record(d)
                res["short"] = None
            if res is not None and block._kwargs["stride"][0] == 1 and tuple(res["t"].shape[2:]) == tuple(xshape[2:]) \
                    and res["t"].shape[1] == block._kwargs["num_filter"]:
                extra = dict(residual=res["t"])
                res["used"] = True
                side_blk = side_target(block, c16_in)
                if side_blk is not None:
                    a_ = side_blk.quantize_args
                    extra["side_codes"] = dict(thr=side_blk.input_max.data()._t, width=a_.in_width,
                                               flags=ops.act_flags(signed=a_.in_signed))
                # (fp32 in and out, or codes in with both outputs of the dual form)
                if side_blk is not None or c16_in is None:
                    sub_link = sub_target(block, fz, xshape)
                    if sub_link is not None:
                        extra["subsample"] = True
            # (a 1x1 convolution that READS codes writes codes too from 256 input channels up - the first 1x1 of a ResNet unit
            # fed by the trunk's code copy - and, on large planes, up to 32: MobileNetV2's 32 -> 16 and 16 -> 96 behind a first
            # convolution that hands its codes over; in between the both-sides instantiations are not built)
            if not extra and (c16_in is None or xshape[1] >= 256 or
                              (xshape[1] <= 32 and xshape[0] * xshape[2] * xshape[3] > 32 * 4096 and _THIN_FORMS)):
                # through a depthwise consumer: on every plane by default, `_DW_C16_MIN_PIXELS` above says why and what a batch
                # alone on the GPU would prefer (fq_dwconv3x3_c16 102 us against 158 at 112x112 stride 2 and 49 against 61 at
                # 56x56 stride 2, but 23 against 18 at 14x14 where the flat fp32 form is at its best) - profiles/r3_handover.txt
                xs = c16_in.shape if c16_in is not None else tuple(x._t.shape)
                s_ = stride_
                fz["c16_pays"] = len(xs) == 4 and ((xs[2] - 1) // s_ + 1) * ((xs[3] - 1) // s_ + 1) >= _DW_C16_MIN_PIXELS
                out_codes = handover_target(block)
                if out_codes is not None and fz.get("via") is not None:
                    # across two MobileNetV2 units (fuse.visit_unit_links): only while the consumer, which will READ codes, can
                    # write codes as well (the rule above, for ITS input) - otherwise its depthwise layer falls back to fp32
                    co = block._kwargs["num_filter"]
                    if not (co >= 256 or (co <= 32 and xs[0] * ((xs[2] - 1) // s_ + 1) * ((xs[3] - 1) // s_ + 1) > 32 * 4096
                                          and _THIN_FORMS)):
                        out_codes = None
                if out_codes is not None:
                    extra = dict(out_codes=out_codes)
            if not extra or (set(extra) == {"residual"} and side_blk is None):
                gap = gap_target(block, fz, c16_in, xshape, "residual" in extra)
                if gap is not None:
                    y, stat = ops.pwconv_i8_gap(x_arg, codes, scales, rowsum, None if bias is None else bias._t, bn_scale=scale,
                                                bn_shift=shift, act=res["act"] if "residual" in extra else fz["act"],
                                                residual=extra.get("residual"), **plan)
                    pooled = NDArray(y)
                    pooled._fq_stat = stat
                    pooled._fq_pooled_by = gap                # (the pooling block behind hands THIS tensor through)
                    return pooled
            short = defer_shortcut(block, fz, x, x_arg, c16_in, (codes, scales, rowsum), bias, scale, shift, plan, stride_, extra)
            if short is not None:
                return short
            pair = recompute_target(block, fz, x, plan, c16_in) if not extra else None
            if pair is not None:
                # statistic only; the depthwise block behind recomputes the values inside its own launch
                b_ = None if bias is None else bias._t
                stat = ops.pwconv_i8_stat(x_arg, codes, scales, rowsum, b_, bn_scale=scale, bn_shift=shift, act=fz["act"], **plan)
                xs = tuple(x_arg.shape)
                res_ = NDArray(_placeholder((xs[0], block._kwargs["num_filter"], xs[2], xs[3]), x_arg.device))
                res_._fq_stat = stat
                res_._fq_deferred = dict(x=x_arg, codes=(codes, scales, rowsum), bias=b_, bn=(scale, shift), act=fz["act"],
                                         plan=dict(in_stat=plan["in_stat"], width=plan["width"], flags=plan["flags"],
                                                   cur_out=plan.get("cur_out")), consumer=pair)
                return res_
            out = ops.pwconv_i8(x_arg, codes, scales, rowsum, None if bias is None else bias._t,
                                bn_scale=scale, bn_shift=shift,
                                act=res["act"] if "residual" in extra else fz["act"],
                                stride=stride_, **extra, **plan)
            y, stat = out[0], out[1]
            if sub_link is not None and side_blk is None:
                trunk = NDArray(y)
                trunk._fq_stat = stat
                trunk._fq_sub2 = {"hw": tuple(xshape[2:]), "readers": sub_link["readers"], "unit": sub_link["unit"]}
                return trunk
            if side_blk is not None:
                trunk = NDArray(y)
                trunk._fq_stat = stat
                trunk._fq_side = (side_blk, out[2])
                if sub_link is not None:
                    trunk._fq_sub2 = {"hw": tuple(xshape[2:]), "readers": sub_link["readers"], "unit": sub_link["unit"]}
                return trunk
    else:
        if plan:          # input is to be quantised but the integer path does not apply: explicit apply pass
            t = contiguous(x._t)
            if "in_thr" not in plan:
                xq = ops.fake_quant_online_prestat(t, plan["in_stat"], plan["width"], plan["flags"],
                                                   cur_out=plan.get("cur_out"))[0]
            else:
                if "in_stat" in plan:
                    ops.batch_mean(plan["in_stat"], out=plan["cur_out"])
                xq = ops.fake_quant_offline(t, plan["in_thr"], plan["width"], plan["flags"], want_stat=False)[0]
            xq_ = NDArray(xq)
            xq_._fq_sub2 = x._fq_sub2
            x = xq_
        out = _convolve(block, F, x, weight_q, bias)
        from .. import fuse as _fuse
        res = getattr(block, "_fq_residual", None)
        if (res is not None and scale is not None and fz["act"] == "none" and res.get("owner") is not None and _fuse.BN_ADD
                and tuple(res["t"].shape) == tuple(out._t.shape) and out._t.is_cuda):
            # the closing convolution of a residual unit that does NOT run on the codes (quantisation switched off: the KL
            # collection's forward, an fp32 evaluation): its BatchNorm, the unit's add and activation, the statistic - and the
            # histogram of the unit's output while feature maps are collected - in ONE pass instead of two (fq_bn_add_act_stat)
            owner = res["owner"]
            sink = _fuse._kl_sink(owner)
            y, stat = ops.bn_act_stat(contiguous(out._t), scale, shift, res["act"], hist=sink, residual=res["t"])
            res["used"] = True
            r_ = NDArray(y)
            r_._fq_stat = stat
            if _fuse._collection is not None:
                r_._fq_kl = (owner, sink)
            return r_
        if fz["bn"] is None and fz["act"] == "none":
            return out
        if scale is None:
            c = out.shape[1]
            scale = torch.ones(c, dtype=torch.float32, device=out._t.device)
            shift = torch.zeros(c, dtype=torch.float32, device=out._t.device)
        sink = _fuse._kl_sink(block)                    # (a KL collection past its first batch: this pass bins what it stores)
        y, stat = ops.bn_act_stat(out._t.contiguous(), scale, shift, fz["act"], hist=sink)
        res = NDArray(y)
        res._fq_stat = stat
        if _fuse._collection is not None:
            res._fq_kl = (block, sink)
        return res
    if isinstance(y, ops.Codes16):
        res = NDArray(y.t)                              # the codes travel on the NDArray; only the linked consumer reads them
        res._fq_c16 = y
    else:
        res = NDArray(y)
    res._fq_stat = stat
    return res


# ---- weight branch -----------------------------------------------------------------------------------------------------------
def _fake_quant_weight(block, args, weight):
    """convert_conv2d.py:68-95: one scale per layer / per group (G in {1, Cout}) / per output channel; per-channel 3x3
    filters go through the Winograd domain when asked to (:71-83).

    The reference recomputes this on every forward; here the result is kept while the source is the same tensor object in
    the same in-place version (an optimiser step, `set_data` or `load_parameters` change one or the other) - identical
    values, ~150 of the ~200 launches of a MobileNetV2 calibration step gone.  Not while gradients are recorded (the
    straight-through link belongs to that forward's tape) and not for folded fake-BN weights (a fresh tensor per forward)."""
    keep = not args.fake_bn and not autograd.is_recording()
    if keep:
        held = block.__dict__.get("_fq_wq_cache")
        if held is not None and held[0] is weight._t and held[1] == weight._t._version:
            return held[2]
    out = _fake_quant_weight_now(block, args, weight)
    if keep:
        block.__dict__["_fq_wq_cache"] = (weight._t, weight._t._version, out)      # (holds the source: its address cannot be reused)
    return out


def _fake_quant_weight_now(block, args, weight):
    w = contiguous(weight._t)
    if args.quant_type == 'channel' and args.wino_quantize != 'none' and block._kwargs['kernel'] == (3, 3):
        wq = ops.wino_weight_fake_quant(w, args.wino_quantize, args.wt_width)
        return NDArray(autograd.wino_link(w, wq, *ops.winograd_matrices(args.wino_quantize)))   # transform's gradient
    if args.quant_type == 'channel':
        rows = block._kwargs['num_filter']
    elif args.quant_type == 'group':
        rows = block._kwargs['num_group']
        if rows not in (1, w.shape[0]):
            # the reference broadcasts a (G,1,1,1) scale against (Cout,Cin/g,kh,kw): MXNet raises here too
            raise ValueError("group-wise weight quantisation needs num_group in {1, num_filter} "
                             "(got num_group=%d, num_filter=%d): operands could not be broadcast" % (rows, w.shape[0]))
    else:
        rows = 1
    # identity backward (ste_func.py:43-44); a no-op unless autograd is recording
    return NDArray(autograd.ste_link(w, ops.weight_fake_quant(w, rows, args.wt_width)))


# ---- the converted forward ---------------------------------------------------------------------------------------------------
def _quantised_conv(self, F, x, weight, bias=None, input_max=None,
                    gamma=None, beta=None, running_mean=None, running_var=None):
    args = self.quantize_args
    dw = getattr(self, "_fq_dw_fused", None)
    pw = getattr(self, "_fq_pw_fused", None)
    taken_over = dw is not None or pw is not None
    if taken_over and autograd.is_recording():
        raise RuntimeError("this net was rewired by quantize.fuse.fuse_inference (inference only): call "
                           "quantize.fuse.unfuse(net) before recording gradients")
    frozen = self.fixed_params == 1
    weight_raw, plan = weight, {}

    if args.fake_bn and not frozen:                                            # :47-51
        terms = BatchNormTerms(gamma, beta, running_mean, running_var)
        weight, bias = terms.fold_weight(F, weight), terms.fold_bias(F, bias)

    sub = getattr(x, "_fq_sub2", None)
    if sub is not None and not any(self is r for r in sub["readers"]):
        raise RuntimeError("a subsampled trunk (fq_pwconv_i8_sub2) reached a block that is not one of its two readers")

    deferred = getattr(x, "_fq_deferred", None)
    if deferred is not None and not (deferred["consumer"] is self and dw is not None and self.enable_quantize
                                     and args.quantize_input and self.quantize_input and not self.quantize_input_offline):
        x, deferred = _materialise(x), None              # (not the consumer the pair was made for: the stored tensor after all)

    weight_q = weight
    if self.enable_quantize:
        if args.quantize_input and deferred is not None:                       # :55-66, inside the recomputing launch
            flags = ops.act_flags(signed=args.in_signed)
        elif args.quantize_input:                                              # :55-66
            flags = ops.act_flags(signed=args.in_signed)
            if taken_over:
                plan = fused_input_plan(self, x, input_max, flags, args.in_width)
            else:
                xq = fake_quant_block_input(self, x, input_max, flags, args.in_width)
                if sub is not None and xq is not x:
                    xq._fq_sub2 = sub
                x = xq
        if not frozen:                                                         # :68-99
            weight_q = _fake_quant_weight(self, args, weight)

    if self.fixed_params == 0:                                                 # :101-105: freeze what was just computed
        self.fixed_params = 1
        self.weight.set_data(weight_q)
        if bias is not None:
            self.bias.set_data(bias)

    if deferred is not None:
        return depthwise_recompute(self, x, weight_q, bias, flags, args.in_width)
    if dw is not None:
        return depthwise_fused(self, x, weight_q, bias, plan)
    if pw is not None:
        return pointwise_fused(self, F, x, weight_raw, weight_q, bias, plan, bool(self.enable_quantize))
    return _convolve(self, F, x, weight_q, bias)                                # :108 — MIOpen through torch


# ---- converter -----------------------------------------------------------------------------------------------------------------
_FAKE_BN_PARAMS = (("gamma", "ones", True), ("beta", "zeros", True),
                   ("running_mean", "zeros", False), ("running_var", "ones", False))


def _attach_fake_bn(block):
    """The four BatchNorm vectors as Parameters of the convolution (:122-141) and the pre-hook that records the batch
    statistics of the un-folded convolution for `update_ema` while training (:144-154)."""
    channels = block._kwargs['num_filter']
    for name, init, trainable in _FAKE_BN_PARAMS:
        setattr(block, name, block.params.get(name, shape=(channels,), init=init, differentiable=trainable,
                                              allow_deferred_init=True))

    @torch.no_grad()
    def record_batch_stats(blk, inputs):
        w = blk.weight.data()
        b = blk.bias.data() if blk.bias is not None else nd.zeros(shape=w.shape[0], ctx=w.context)
        y = nd.Convolution(inputs[0], w, b, **blk._kwargs)
        count = y.shape[0] * y.shape[2] * y.shape[3]
        blk.current_mean = y.sum(axis=(0, 2, 3)) / count
        centred = (y - blk.current_mean.reshape(1, -1, 1, 1)) ** 2
        blk.current_var = centred.sum(axis=(0, 2, 3)) / count
    block.register_forward_pre_hook(record_batch_stats)


def gen_conv2d_converter(weight_width=8, quant_type="layer",
                         quantize_input=True, input_signed=False, input_width=8,
                         fake_bn=False, wino_quantize="none"):
    if wino_quantize != "none" and wino_quantize not in WINOGRAD_VARIANTS:
        raise AssertionError("wino_quantize must be 'none' or one of %s" % (WINOGRAD_VARIANTS,))
    settings = QuantizedArgs(in_signed=input_signed, in_width=input_width, wt_width=weight_width,
                             quantize_input=quantize_input, fake_bn=fake_bn, quant_type=quant_type,
                             wino_quantize=wino_quantize)

    def _converter(m):
        if not isinstance(m, Conv2D):
            raise AssertionError("gen_conv2d_converter expects a Conv2D block")
        if quantize_input:
            INPUT_RANGE.attach(m)
        if fake_bn:
            _attach_fake_bn(m)
        rebind_forward(m, _quantised_conv)
        m.quantize_args = settings
        m.quantize_input = quantize_input
        m.enable_quantize = True
        m.fixed_params = -1              # -1: quantise weights every forward; 0: freeze on the next forward; 1: frozen
    return _converter
