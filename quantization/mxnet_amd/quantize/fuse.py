"""Inference-time producer fusion (new; not in the reference — it only re-schedules work).

Between two quantised convolutions the reference's nets run BatchNorm (inference) and ReLU/ReLU6 as separate Gluon
blocks, then the next layer's fake-quant first makes a statistic pass: BN (r+w) + ReLU (r+w) + statistic (r) = 20 B per
element of pure elementwise traffic.  `fuse_inference(net)` rewires every `[BatchNorm -> relu | RELU6]` pair (and lone
BatchNorms) found as consecutive children of a (Hybrid)Sequential so that ONE HIP pass (`fq_bn_act_stat`, 8 B/elem)
produces the activation AND its per-sample max|y|; the statistic rides along on the NDArray (`_fq_stat`) and the
consumer's fake-quant then runs its apply pass only.

What does not change: every fake-quantised tensor is still exactly `oracle(fake_quant)(its actual input)` — the
statistic handed over is bit-identical to the one the statistic pass would compute.  What does change, in the last bit:
BatchNorm is evaluated as x*scale[c] + shift[c] (scale = gamma/sqrt(var+eps), shift = beta - mean*scale; multiply and
add separately rounded) instead of MIOpen's formula — the same freedom any BN implementation takes.

Use after parameters are final and on their device (`reset_ctx`, `load_parameters`); `unfuse(net)` restores the blocks.
"""
import types

import torch

from ..mx.gluon import nn
from ..mx.ndarray import NDArray
from .. import ops

__all__ = ["fuse_inference", "unfuse", "refresh", "eval_head", "EvalHead"]


def _is_relu6_block(b):
    return type(b).__name__ == "RELU6"


def _act_kind(b):
    if isinstance(b, nn.Activation) and b._act_type == "relu" and not hasattr(b, "quantize_args") \
            and b.hybrid_forward.__func__ is nn.Activation.hybrid_forward:
        return "relu"
    if _is_relu6_block(b):
        return "relu6"
    return None


def _tensors_key(*tensors):
    """identity of parameter storage AND content: `set_data` / `load_parameters` / an optimiser step write in place (same
    pointer), but every in-place write bumps the tensor's version counter"""
    return tuple((t.data_ptr(), t._version) for t in tensors)


def _bn_constants(bn):
    g0, b = bn.gamma.data()._t, bn.beta.data()._t
    mean, var = bn.running_mean.data()._t, bn.running_var.data()._t
    g = torch.ones_like(g0) if bn._kwargs.get("fix_gamma", False) else g0
    scale = g / torch.sqrt(var + bn._kwargs["eps"])
    shift = b - mean * scale
    return scale.contiguous(), shift.contiguous(), _tensors_key(g0, b, mean, var)


# ---- histogram sinks of a feature-map collection (distribution_calibrate.collect_feature_maps) ----------------------------
# From its second batch on a KL collection knows every collected tensor's range, and which fused producer made it: that
# producer then bins what it stores (fq_bn_act_stat_hist / fq_add_act_stat_hist) and the collection skips its own pass over
# the tensor.  {producer block: sink (fm_max, hist, neg)}; None outside a collection.
_collection = None
_collection_calls = {}        # producer -> times it ran in the current forward (a sink is only right for exactly one)


def begin_collection():
    global _collection
    _collection = {}
    _collection_calls.clear()
    return _collection


def end_collection():
    global _collection
    _collection = None
    _collection_calls.clear()


def collection_calls():
    return _collection_calls


def _kl_sink(producer):
    if _collection is None:
        return None
    _collection_calls[producer] = _collection_calls.get(producer, 0) + 1
    return _collection.get(producer)


def _fused_bn_forward(self, F, x, gamma, beta, running_mean, running_var):
    st = self._fq_fused
    key = _tensors_key(gamma._t, beta._t, running_mean._t, running_var._t)
    if st["key"] != key:                       # parameters moved or rewritten (reset_ctx / load / set_data): recompute
        st["scale"], st["shift"], st["key"] = _bn_constants(self)
    t = x._t if x._t.is_contiguous() else x._t.contiguous()
    if st.get("pool") is not None and t.dim() == 4 and t.shape[3] % 4 == 0:
        # BatchNorm -> activation -> MaxPool2D(3, 2, 1) (the head of the ImageNet ResNets) in one pass
        y, stat = ops.bn_act_maxpool_stat(t, st["scale"], st["shift"], st["act"], want_stat=True)
        out = NDArray(y)
        out._fq_stat = stat
        out._fq_pooled_by = st["pool"]               # (the marker rides on the tensor: see _pool_stat_forward)
        return out
    else:
        sink = _kl_sink(self)
        y, stat = ops.bn_act_stat(t, st["scale"], st["shift"], st["act"], want_stat=True, hist=sink)
        out = NDArray(y)
        out._fq_stat = stat
        if _collection is not None:
            out._fq_kl = (self, sink)
        return out


def _pool_after_bn_forward(self, F, x):
    """MaxPool2D whose work the preceding BatchNorm already did (quantize/fuse.py); pools itself when it did not (W % 4)."""
    if getattr(x, "_fq_pooled_by", None) is self:
        x._fq_pooled_by = None
        return x
    return self._fq_pool_fused["orig"](F, x)


def _pool_stat_forward(self, F, x):
    """MaxPool2D(3, 2, 1) behind a fused producer (the ResNets' first convolution): pooling + the per-sample statistic of the
    pooled tensor in one pass (fq_bn_act_maxpool_stat with the identity BatchNorm: x * 1 + 0 is x) - unless the convolution's
    own launch has pooled already (fq_stem_conv7x7s2_pool)."""
    if getattr(x, "_fq_pooled_by", None) is self:    # (the marker rides on the tensor the convolution returned, not on this
        x._fq_pooled_by = None                       #  block: a forward that never reaches this pool leaves nothing behind)
        return x
    t = x._t if x._t.is_contiguous() else x._t.contiguous()
    if t.dim() != 4 or t.shape[3] % 4 or not t.is_cuda:
        return self._fq_pool_fused["orig"](F, x)
    st = self._fq_pool_fused
    c = t.shape[1]
    if st.get("ones") is None or st["ones"].numel() != c or st["ones"].device != t.device:
        st["ones"] = torch.ones(c, dtype=torch.float32, device=t.device)
        st["zeros"] = torch.zeros(c, dtype=torch.float32, device=t.device)
    y, stat = ops.bn_act_maxpool_stat(t, st["ones"], st["zeros"], "none", want_stat=True)
    out = NDArray(y)
    out._fq_stat = stat
    return out


def _is_maxpool_3x3s2p1(b):
    if type(b) is not nn.MaxPool2D or b.hybrid_forward.__func__ is not nn.MaxPool2D.hybrid_forward:
        return False
    k = b._kwargs
    return (k["kernel"] == (3, 3) and k["stride"] == (2, 2) and k["pad"] == (1, 1) and not k["global_pool"]
            and k["pooling_convention"] == "valid")


import os as _os

HANDOVER = _os.environ.get("FQ_HANDOVER", "1") != "0"      # int8 C16 hand-over between fused convolutions under offline input quantisation (tests switch it off to
                     # compare: the logits are bit-equal either way)
# round 4: the closing 1x1 of a ResNet unit stores the trunk twice - fp32 for the shortcut, codes for the next unit's first 1x1
# (fq_pwconv_i8_c16_dual).  Same logits either way (tests/test_gpu_c16.py)
SIDE_CODES = _os.environ.get("FQ_HANDOVER_SIDE", "1") != "0"
# ... and MobileNetV2's first convolution hands its single consumer's codes over (fq_stem_conv3x3s2_c16)
STEM_CODES = _os.environ.get("FQ_HANDOVER_STEM", "1") != "0"
# Recompute pairs (round 6, csrc/fq_pwdw.hip): under ONLINE input quantisation a fused 1x1 convolution followed by a fused
# depthwise 3x3 runs as a statistic-only pass + ONE launch that recomputes the 1x1 output inside the depthwise kernel - the
# tensor between them is never written.  FQ_RECOMPUTE=0 keeps the two storing launches (A/B); FQ_RECOMPUTE_MIN_PIXELS: the
# smallest input plane (pixels) the pair is taken on.
# FQ_WINO_SLICED=0: the 3x3 layers of a Winograd-domain quantised net stay with the tensor library's fp32 convolution of the
# back-transformed filter instead of the three-slice integer form (tools/sliced_effect.py measures what the slices change)
WINO_SLICED = _os.environ.get("FQ_WINO_SLICED", "1") != "0"
# FQ_BN_ADD=0: the closing BatchNorm of a residual unit and the unit's add + activation as two passes again (A/B)
BN_ADD = _os.environ.get("FQ_BN_ADD", "1") != "0"
RECOMPUTE = _os.environ.get("FQ_RECOMPUTE", "1") != "0"
RECOMPUTE_MIN_PIXELS = int(_os.environ.get("FQ_RECOMPUTE_MIN_PIXELS", "3136"))
RECOMPUTE_MAX_CIN = int(_os.environ.get("FQ_RECOMPUTE_MAX_CIN", "128"))      # (input channels of the 1x1: see DESIGN.md for the pairs that pay)
# Subsampled trunk (round 6, fq_pwconv_i8_sub2): the closing 1x1 of a ResNet-v1 stage stores only the pixels its two readers -
# the next stage's first 1x1 and shortcut 1x1, both stride 2 without padding - look at (FQ_SUBSAMPLE=0: the whole tensor; A/B)
SUBSAMPLE = _os.environ.get("FQ_SUBSAMPLE", "1") != "0"
# Pooled producer (round 6, fq_pwconv_i8_gap): the 1x1 convolution in front of a global average pooling - the last 1x1 of the
# MobileNets, the closing 1x1 of ResNet-50's last unit - hands the plane means over instead of the planes (FQ_GAP_FUSE=0: two launches)
GAP_FUSE = _os.environ.get("FQ_GAP_FUSE", "1") != "0"
# Folded shortcut (round 6, fq_pwconv_i8_shortcut): the shortcut convolution of a stage's first unit is computed inside the launch of
# the unit's closing 1x1 - the shortcut tensor is never written (FQ_SHORTCUT_FUSE=0: two launches)
SHORTCUT_FUSE = _os.environ.get("FQ_SHORTCUT_FUSE", "1") != "0"
UNIT_LINKS = _os.environ.get("FQ_HANDOVER_UNITS", "1") != "0"      # hand-over from a MobileNetV2 unit without shortcut to the next block (A/B)


def _identity_forward(self, F, x, *args, **kwargs):
    return x


def _is_dw3x3(b):
    if type(b) is not nn.Conv2D:
        return False
    k = b._kwargs
    cin = b.weight.shape[1] if b.weight.shape else 0
    return (k["kernel"] == (3, 3) and k["pad"] == (1, 1) and k["dilate"] == (1, 1) and k["stride"] in ((1, 1), (2, 2))
            and k["num_group"] == k["num_filter"] and cin == 1 and k["layout"] == "NCHW" and b.act is None
            and not getattr(getattr(b, "quantize_args", None), "fake_bn", False))


def _is_pw1x1(b):
    """A converted 1x1 convolution fq_pwconv_i8 can take: stride 1, or stride 2 (the shortcut / first convolutions of the
    ResNet stages) when the input channel count is one the strided form is built for."""
    if type(b) is not nn.Conv2D or not hasattr(b, "quantize_args"):
        return False
    k = b._kwargs
    if not (k["kernel"] == (1, 1) and k["pad"] == (0, 0) and k["dilate"] == (1, 1) and k["num_group"] == 1
            and k["layout"] == "NCHW" and b.act is None and not b.quantize_args.fake_bn):
        return False
    if k["stride"] == (1, 1):
        return True
    cin = b.weight.shape[1] if b.weight.shape is not None and len(b.weight.shape) == 4 else 0
    return k["stride"] == (2, 2) and cin > 0 and ops.pwconv_strided_supported(cin)


def _is_dense3x3(b):
    """A converted dense 3x3 convolution (stride 1, padding 1) whose integer form fq_conv3x3_i8 takes: the 3x3 layers of the
    ResNet units.  Under Winograd-domain weight quantisation the spatial filter is not an integer multiple of one scale per
    channel: it then goes through the three-slice form (fq_weight_slices + fq_conv3x3_i8_sliced; `_wino_sliced`), which
    needs whole 32-channel tiles."""
    if type(b) is not nn.Conv2D or not hasattr(b, "quantize_args"):
        return False
    k = b._kwargs
    a = b.quantize_args
    cin = b.weight.shape[1] if b.weight.shape is not None and len(b.weight.shape) == 4 else 0
    return (k["kernel"] == (3, 3) and k["pad"] == (1, 1) and k["dilate"] == (1, 1) and k["stride"] == (1, 1)
            and k["num_group"] == 1 and k["layout"] == "NCHW" and b.act is None and not a.fake_bn
            and cin in (64, 128, 256, 512) and k["num_filter"] >= 32
            and (not _wino_sliced(b) or (k["num_filter"] % 32 == 0 and WINO_SLICED)))


def _wino_sliced(b):
    a = b.quantize_args
    return a.quant_type == "channel" and a.wino_quantize != "none"


def _is_stem3x3s2(b):
    """The un-quantised first convolution of the ImageNet nets: Conv2D(3 -> 32, 3x3, stride 2, pad 1) of the MobileNets,
    Conv2D(3 -> 64, 7x7, stride 2, pad 3) of the ResNets."""
    if type(b) is not nn.Conv2D or hasattr(b, "quantize_args"):
        return False
    if b.hybrid_forward.__func__ is not nn.Conv2D.hybrid_forward:
        return False
    k = b._kwargs
    shp = b.weight.shape
    return (len(shp) == 4 and k["num_group"] == 1 and k["dilate"] == (1, 1) and k["layout"] == "NCHW" and b.act is None
            and ops.stem_conv_supported(shp[1], shp[0], k["kernel"], k["stride"], k["pad"]))


def _stem_forward(self, F, x, weight, bias=None):
    st = self._fq_stem_fused
    w = weight._t
    key = (w.data_ptr(), w._version)
    if st["wkey"] != key:                      # tap-major copy of the weights, refreshed when the parameter changes
        st["wt"], st["wkey"] = w.permute(1, 2, 3, 0).contiguous(), key
    scale = shift = None
    if st["constants"] is not None:
        scale, shift = st["constants"]()
    t = x._t if x._t.is_contiguous() else x._t.contiguous()
    out_codes = None
    if st.get("next") is not None and w.shape[2] == 3 and STEM_CODES:
        # offline input quantisation: the single consumer's codes instead of the fp32 tensor (fq_stem_conv3x3s2_c16)
        from .convert.convert_conv2d import handover_target
        out_codes = handover_target(self, st)
    pool = st.get("pool")
    # (a forward hook on the convolution or on a block bypassed behind it would be shown the pooled 56x56 tensor where it
    # expects the 112x112 convolution output: the two-launch form then, as convert_conv2d.handover_target has it)
    hooked = any(bool(getattr(b, "_forward_hooks", None) or getattr(b, "_forward_pre_hooks", None))
                 for b in (self, st.get("bn"), st.get("act_block"), pool) if b is not None)
    pooled = (pool is not None and STEM_POOL and out_codes is None and w.shape[2] == 7 and t.dim() == 4 and not hooked
              and ops.stem_pool_supported(t.shape[2], t.shape[3]))
    y, stat = ops.stem_conv_s2(t, w, None if bias is None else bias._t, bn_scale=scale, bn_shift=shift,
                               act=st["act"], want_stat=True, w_tap_major=st["wt"], pool=pooled,
                               **({} if out_codes is None else dict(out_codes=out_codes)))
    if out_codes is not None:
        out = NDArray(y.t)
        out._fq_c16 = y
    else:
        out = NDArray(y)
    if pooled:
        out._fq_pooled_by = pool                    # the MaxPool2D block behind hands THIS tensor through
    out._fq_stat = stat
    return out


def _gap_stat_forward(self, F, x):
    if getattr(x, "_fq_pooled_by", None) is self:       # (the producer's launch pooled already: fq_pwconv_i8_gap)
        x._fq_pooled_by = None
        return x
    t = x._t if x._t.is_contiguous() else x._t.contiguous()
    y, stat = ops.global_avg_pool_stat(t, want_stat=True)
    out = NDArray(y)
    out._fq_stat = stat
    return out


def _flatten_keep_stat_forward(self, F, x):
    out = NDArray(x._t.reshape(x._t.shape[0], -1))
    out._fq_stat = x._fq_stat                       # a per-sample statistic is unchanged by flattening
    return out


def _plain_dw_forward(self, F, x, weight, bias=None):
    from .convert.convert_conv2d import depthwise_fused
    return depthwise_fused(self, x, weight, bias, {})


def _tail_conv(seq):
    """The 1x1 convolution that ends a unit's body (followed only by the BatchNorm it folded), when fq_pwconv_i8 can take the
    unit's shortcut as its residual operand: taken over by fuse_inference, stride 1, no activation of its own."""
    kids = list(seq._children.values()) if isinstance(seq, (nn.Sequential, nn.HybridSequential)) else []
    if kids and isinstance(kids[-1], (nn.Sequential, nn.HybridSequential)):      # MobileNetV2: the last conv + BN pair
        return _tail_conv(kids[-1])
    if len(kids) < 2 or type(kids[-1]) is not nn.BatchNorm:
        return None
    conv, bn = kids[-2], kids[-1]
    fz = getattr(conv, "_fq_pw_fused", None)
    if fz is None or fz.get("kind") != "1x1" or fz["bn"] is not bn or fz["act"] != "none":
        return None
    return conv if conv._kwargs["stride"] == (1, 1) else None


def _with_residual(conv, shortcut, act, run, owner=None):
    """Runs `run()` with the shortcut handed to `conv` (convert_conv2d.pointwise_fused adds it in the convolution's epilogue
    when that call runs on the integer codes).  Returns (output, True) when it was consumed there.  A shortcut that was not
    computed (`_fq_short`: the record of the shortcut convolution's operands) travels as that record."""
    short = getattr(shortcut, "_fq_short", None)
    t = None if short is not None else (shortcut._t if shortcut._t.is_contiguous() else shortcut._t.contiguous())
    conv._fq_residual = {"t": t, "short": short, "act": act, "used": False, "owner": owner}
    try:
        out = run()
        return out, conv._fq_residual["used"]
    finally:
        conv._fq_residual = None


def _residual_unit_forward(self, x):
    """`(body(x) + shortcut(x)).relu()` of the model zoo's ResNet units.  When the body ends in a 1x1 convolution that runs on
    the integer codes (BottleneckV1), the add and the ReLU happen in THAT convolution's epilogue (fq_pwconv_i8_strided's
    residual operand: the convolution's output is never written and read back); otherwise one pass adds, applies the ReLU
    and takes the per-sample statistic (fq_add_act_stat).  Either way both quantised consumers of the sum (the next unit's
    first convolution and, at a stage boundary, its shortcut convolution) skip their statistic pass."""
    from ..mx import autograd
    if autograd.is_recording():
        raise RuntimeError("this net was rewired by quantize.fuse.fuse_inference (inference only): call "
                           "quantize.fuse.unfuse(net) before recording gradients")
    sub = getattr(x, "_fq_sub2", None)
    if sub is not None and sub["unit"] is not self:
        raise RuntimeError("a subsampled trunk (fq_pwconv_i8_sub2) reached a unit it was not made for")
    tail = _tail_conv(self.body)
    sc = _shortcut_conv(self) if (tail is not None and SHORTCUT_FUSE) else None
    if sc is not None:
        sc._fq_defer_short = True                  # one shot: convert_conv2d.pointwise_fused may answer with a record instead
    try:
        shortcut = x if self.downsample is None else self.downsample(x)
    finally:
        if sc is not None:
            sc._fq_defer_short = False
    if tail is not None:
        h, consumed = _with_residual(tail, shortcut, "relu", lambda: self.body(x), owner=self)
        if consumed:
            return h
    else:
        h = self.body(x)
    if getattr(shortcut, "_fq_short", None) is not None:       # (nobody folded it after all)
        from .convert.convert_conv2d import materialise_shortcut
        shortcut = materialise_shortcut(shortcut)
    a = h._t if h._t.is_contiguous() else h._t.contiguous()
    b = shortcut._t if shortcut._t.is_contiguous() else shortcut._t.contiguous()
    sink = _kl_sink(self)
    y, stat = ops.add_act_stat(a, b, "relu", want_stat=True, hist=sink)
    out = NDArray(y)
    out._fq_stat = stat
    if _collection is not None:
        out._fq_kl = (self, sink)
    return out


def _shortcut_conv(unit):
    """`downsample[0]` of a residual unit when it is a fused 1x1 convolution whose BatchNorm it folded and nothing else follows."""
    ds = getattr(unit, "downsample", None)
    if type(ds) not in (nn.Sequential, nn.HybridSequential) or "forward" in ds.__dict__:
        return None
    kids = list(ds._children.values())
    if len(kids) != 2 or type(kids[0]) is not nn.Conv2D or type(kids[1]) is not nn.BatchNorm:
        return None
    fz = getattr(kids[0], "_fq_pw_fused", None)
    if fz is None or fz.get("kind") != "1x1" or fz["bn"] is not kids[1] or fz["act"] != "none":
        return None
    return kids[0]


def _linear_bottleneck_forward(self, x):
    """MobileNetV2's `out(x) + x`: the add moves into the epilogue of the projection convolution (no activation)."""
    from ..mx import autograd
    if autograd.is_recording():
        raise RuntimeError("this net was rewired by quantize.fuse.fuse_inference (inference only): call "
                           "quantize.fuse.unfuse(net) before recording gradients")
    tail = _tail_conv(self.out) if self.use_shortcut else None
    if tail is None:
        out = self.out(x)
        return out + x if self.use_shortcut else out
    out, consumed = _with_residual(tail, x, "none", lambda: self.out(x))
    return out if consumed else out + x


def _is_linear_bottleneck(b):
    from ..mx.gluon import model_zoo as zoo
    return getattr(type(b), "forward", None) is zoo.LinearBottleneck.forward and hasattr(b, "out") \
        and "forward" not in b.__dict__


def _is_residual_unit(b):
    """a block whose forward IS the zoo's `(x + residual).relu()` (BasicBlockV1 / BottleneckV1 and subclasses)"""
    from ..mx.gluon import model_zoo as zoo
    fwd = getattr(type(b), "forward", None)
    return fwd in (zoo.BasicBlockV1.forward, zoo.BottleneckV1.forward) and hasattr(b, "body") \
        and "forward" not in b.__dict__


def _bn_constants_getter(bn):
    cache = {"key": None, "val": None}

    def get():
        key = _tensors_key(bn.gamma.data()._t, bn.beta.data()._t, bn.running_mean.data()._t,
                           bn.running_var.data()._t) + (bn.__dict__.get("_fq_refresh", 0),)
        if cache["key"] != key:
            scale, shift, _ = _bn_constants(bn)
            cache["key"], cache["val"] = key, (scale, shift)
        return cache["val"]
    return get


def fuse_inference(net, depthwise=True, pointwise_int8=True, stem=True, residual=True, dense_int8=None):
    """Returns the number of blocks fused (BatchNorms folded + depthwise / pointwise convolutions taken over).
    `dense_int8`: also run the quantised classifier on the integer codes (fq_pwconv_i8 on a 1x1 plane: exact sums, no apply
    pass).  On by default since round 3's rows form (csrc/fq_pw_rows.hip; FQ_DENSE_INT8=0 turns the default off): the forms
    made for convolution planes took 23-28 us for the 1024 -> 1000 layer at batch 128 against 16.5 us for the apply pass +
    rocBLAS, which is why it used to be opt-in."""
    from .convert.convert import bump_mode_epoch
    bump_mode_epoch()
    if dense_int8 is None:
        dense_int8 = DENSE_INT8
    fused = [0]

    def bypass(blk):
        blk._fq_bypassed_orig = blk.hybrid_forward
        blk.hybrid_forward = types.MethodType(_identity_forward, blk)

    def visit_dw(container):
        if not isinstance(container, (nn.Sequential, nn.HybridSequential)):
            return
        kids = list(container._children.values())
        for i, b in enumerate(kids):
            if not _is_dw3x3(b) or hasattr(b, "_fq_dw_fused"):
                continue
            converted = hasattr(b, "quantize_args")
            if not converted and b.hybrid_forward.__func__ is not nn.Conv2D.hybrid_forward:
                continue
            bn = kids[i + 1] if i + 1 < len(kids) else None
            if not (type(bn) is nn.BatchNorm and not hasattr(bn, "_fq_fused") and bn._kwargs.get("axis", 1) == 1
                    and bn.hybrid_forward.__func__ is nn.BatchNorm.hybrid_forward):
                bn = None
            nxt = kids[i + 2] if bn is not None and i + 2 < len(kids) else (kids[i + 1] if bn is None and i + 1 < len(kids) else None)
            act = _act_kind(nxt) if nxt is not None else None
            b._fq_dw_fused = {"bn": bn, "act": act or "none", "act_block": nxt if act else None,
                              "constants": _bn_constants_getter(bn) if bn is not None else None,
                              "orig": None if converted else b.hybrid_forward}
            if not converted:
                b.hybrid_forward = types.MethodType(_plain_dw_forward, b)
            if bn is not None:
                bn._fq_fused = {"taken_by_conv": True, "orig": bn.hybrid_forward, "act_block": None}
                bn.hybrid_forward = types.MethodType(_identity_forward, bn)
            if act:
                bypass(nxt)
            fused[0] += 1

    def visit_stem(container):
        if not isinstance(container, (nn.Sequential, nn.HybridSequential)):
            return
        kids = list(container._children.values())
        for i, b in enumerate(kids):
            if not _is_stem3x3s2(b) or hasattr(b, "_fq_stem_fused"):
                continue
            bn = kids[i + 1] if i + 1 < len(kids) else None
            if not (type(bn) is nn.BatchNorm and not hasattr(bn, "_fq_fused") and bn._kwargs.get("axis", 1) == 1
                    and bn.hybrid_forward.__func__ is nn.BatchNorm.hybrid_forward):
                bn = None
            nxt = kids[i + 2] if bn is not None and i + 2 < len(kids) else (kids[i + 1] if bn is None and i + 1 < len(kids) else None)
            act = _act_kind(nxt) if nxt is not None else None
            b._fq_stem_fused = {"bn": bn, "act": act or "none", "act_block": nxt if act else None, "wt": None, "wkey": None,
                                "constants": _bn_constants_getter(bn) if bn is not None else None,
                                "orig": b.hybrid_forward}
            b.hybrid_forward = types.MethodType(_stem_forward, b)
            if bn is not None:
                bn._fq_fused = {"taken_by_conv": True, "orig": bn.hybrid_forward, "act_block": None}
                bn.hybrid_forward = types.MethodType(_identity_forward, bn)
            if act:
                bypass(nxt)
            # the ResNets pool right behind it: pooling + the statistic of the pooled tensor in one pass
            j = i + 1 + (1 if bn is not None else 0) + (1 if act else 0)
            pool = kids[j] if j < len(kids) and _is_maxpool_3x3s2p1(kids[j]) and not hasattr(kids[j], "_fq_pool_fused") \
                else None
            if pool is not None:
                pool._fq_pool_fused = {"orig": pool.hybrid_forward}
                pool.hybrid_forward = types.MethodType(_pool_stat_forward, pool)
                # ... and, where the shape is built, inside the convolution's own launch (fq_stem_conv7x7s2_pool: the
                # convolution output never leaves the CU; STEM_POOL=0 / FQ_STEM_POOL=0 keeps the two launches)
                b._fq_stem_fused["pool"] = pool
            b._fq_stem_fused["stem_follower"] = (container, j) if pool is None else None
            fused[0] += 1

    def link_stem(b):
        """The first convolution's single consumer, when that is a 1x1 convolution taken over on the codes (MobileNetV2: the
        first unit's expansion) - linked after the pointwise pass, honoured under offline input quantisation only."""
        st = getattr(b, "_fq_stem_fused", None)
        if st is None or st.get("stem_follower") is None:
            return
        container, j = st["stem_follower"]
        kids = list(container._children.values())
        nxt = kids[j] if j < len(kids) else None
        # (MobileNetV2's units are blocks of their own: the consumer is the first convolution of the unit's body)
        while nxt is not None and type(nxt) is not nn.Conv2D:
            inner = getattr(nxt, "out", None) or getattr(nxt, "body", None) or nxt
            sub = list(inner._children.values()) if isinstance(inner, (nn.Sequential, nn.HybridSequential)) else []
            if not sub or getattr(nxt, "use_shortcut", False) or sub[0] is nxt:
                nxt = None
            else:
                nxt = sub[0]
        fz = getattr(nxt, "_fq_pw_fused", None) if nxt is not None else None
        if fz is not None and fz.get("kind") == "1x1" and nxt._kwargs["stride"] == (1, 1) and b.weight.shape[2] == 3:
            st["next"] = nxt

    def visit_gap(container):
        """GlobalAvgPool2D [-> Flatten]: pool and statistic in one launch (fq_global_avg_pool_stat), the statistic carried
        through Flatten to a quantised Dense.  Whatever follows (MobileNetV2: an un-quantised 1x1 convolution), the pooling runs
        as this library's kernel - the tensor library's mean over (128, 1280, 7, 7) took 52 us of that step against 9-15."""
        if not isinstance(container, (nn.Sequential, nn.HybridSequential)):
            return
        kids = list(container._children.values()) + after_features.get(id(container), [])
        for i, b in enumerate(kids):
            if type(b) is not nn.GlobalAvgPool2D or hasattr(b, "_fq_gap_fused"):
                continue
            if b.hybrid_forward.__func__ is not nn.GlobalAvgPool2D.hybrid_forward:
                continue
            j = i + 1
            flat = None
            if j < len(kids) and type(kids[j]) is nn.Flatten and \
                    kids[j].hybrid_forward.__func__ is nn.Flatten.hybrid_forward:
                flat = kids[j]
                j += 1
            b._fq_gap_fused = {"orig": b.hybrid_forward, "flatten": flat,
                               "flatten_orig": None if flat is None else flat.hybrid_forward}
            b.hybrid_forward = types.MethodType(_gap_stat_forward, b)
            # the producer of the pooled tensor when that is a fused 1x1 convolution with no other reader: the convolution right in
            # front of the pooling (behind the BatchNorm / activation it folded: the MobileNets), or the closing 1x1 of the last
            # residual unit of the stage in front of it (ResNet-50) - convert_conv2d.gap_target
            own = list(container._children.values())
            if i < len(own) and own[i] is b and type(container) in (nn.Sequential, nn.HybridSequential) \
                    and "forward" not in container.__dict__:
                for k, c in enumerate(own[:i]):
                    fz = getattr(c, "_fq_pw_fused", None)
                    if fz is not None and fz.get("kind") == "1x1" and type(c) is nn.Conv2D and hasattr(c, "quantize_args") and \
                            k + 1 + (1 if fz["bn"] is not None else 0) + (1 if fz["act_block"] is not None else 0) == i:
                        fz["gap_next"] = {"gap": b, "via": ()}
                prev = own[i - 1] if i > 0 else None
                if type(prev) in (nn.Sequential, nn.HybridSequential) and "forward" not in prev.__dict__ and len(prev._children):
                    unit = list(prev._children.values())[-1]
                    if _is_residual_unit(unit):
                        tail = _tail_conv(unit.body)
                        if tail is not None and hasattr(tail, "quantize_args"):
                            tail._fq_pw_fused["gap_next"] = {"gap": b, "via": (unit, prev)}
            if flat is not None:
                flat.hybrid_forward = types.MethodType(_flatten_keep_stat_forward, flat)
            fused[0] += 1

    def visit_pw(container):
        if not isinstance(container, (nn.Sequential, nn.HybridSequential)):
            return
        kids = list(container._children.values())
        for i, b in enumerate(kids):
            dense3 = _is_dense3x3(b)
            if not (_is_pw1x1(b) or dense3) or hasattr(b, "_fq_pw_fused") or hasattr(b, "_fq_dw_fused"):
                continue
            bn = kids[i + 1] if i + 1 < len(kids) else None
            if not (type(bn) is nn.BatchNorm and not hasattr(bn, "_fq_fused") and bn._kwargs.get("axis", 1) == 1
                    and bn.hybrid_forward.__func__ is nn.BatchNorm.hybrid_forward):
                bn = None
            nxt = kids[i + 2] if bn is not None and i + 2 < len(kids) else (kids[i + 1] if bn is None and i + 1 < len(kids) else None)
            act = _act_kind(nxt) if nxt is not None else None
            b._fq_pw_fused = {"bn": bn, "act": act or "none", "act_block": nxt if act else None,
                              "constants": _bn_constants_getter(bn) if bn is not None else None,
                              "kind": "3x3" if dense3 else "1x1", "sliced": bool(dense3 and _wino_sliced(b))}
            if bn is not None:
                bn._fq_fused = {"taken_by_conv": True, "orig": bn.hybrid_forward, "act_block": None}
                bn.hybrid_forward = types.MethodType(_identity_forward, bn)
            if act:
                bypass(nxt)
            fused[0] += 1
        # int8 hand-over (round 3): a fused convolution whose output goes to ONE consumer - the next convolution of the same
        # Sequential, behind the BatchNorm / activation it folded - may write that consumer's integer codes instead of fp32
        # when the consumer quantises with a stored threshold (decided per forward, convert_conv2d.handover_target).
        # 1x1 / dense 3x3 -> 1x1 / dense 3x3 (ResNet bottlenecks) and 1x1 -> depthwise -> 1x1 (MobileNetV2 units; the
        # depthwise kernel on codes reads AND writes them, so its link counts only when it has a consumer of its own).
        kids = list(container._children.values())

        def after(i, fz):
            j = i + 1 + (1 if fz["bn"] is not None else 0) + (1 if fz["act_block"] is not None else 0)
            return kids[j] if j < len(kids) else None

        for i, b in enumerate(kids):
            fz = getattr(b, "_fq_pw_fused", None) or getattr(b, "_fq_dw_fused", None)
            if fz is None or fz.get("sliced") or not hasattr(b, "quantize_args"):
                continue
            nxt = after(i, fz)
            if type(nxt) is not nn.Conv2D or not hasattr(nxt, "quantize_args"):
                continue
            if hasattr(b, "_fq_dw_fused"):
                if getattr(nxt, "_fq_pw_fused", None) is not None and nxt._fq_pw_fused.get("kind") == "1x1":
                    fz["next"] = nxt
            elif getattr(nxt, "_fq_pw_fused", None) is not None and not nxt._fq_pw_fused.get("sliced"):
                fz["next"] = nxt
            elif getattr(nxt, "_fq_dw_fused", None) is not None and fz.get("kind") == "1x1":
                fz["next"] = nxt                  # (honoured only while the depthwise block hands over too)
                # ... and, under online input quantisation, the recompute pair (convert_conv2d.recompute_target)
                if b._kwargs["stride"] == (1, 1) and b._kwargs["num_group"] == 1:
                    fz["pair_dw"] = nxt

    def visit(container):
        kids = list(container._children.values())
        for i, b in enumerate(kids):
            if type(b) is not nn.BatchNorm or hasattr(b, "_fq_fused") or b._kwargs.get("axis", 1) != 1:
                continue
            if b.hybrid_forward.__func__ is not nn.BatchNorm.hybrid_forward:
                continue                                   # already patched by someone else (e.g. bypass_bn)
            nxt = kids[i + 1] if i + 1 < len(kids) and isinstance(container, (nn.Sequential, nn.HybridSequential)) \
                else None
            act = _act_kind(nxt) if nxt is not None else None
            if act and hasattr(nxt, "_fq_bypassed_orig"):
                act = None
            pool = kids[i + 2] if act and i + 2 < len(kids) and _is_maxpool_3x3s2p1(kids[i + 2]) \
                and not hasattr(kids[i + 2], "_fq_pool_fused") else None
            b._fq_fused = {"act": act or "none", "key": None, "scale": None, "shift": None,
                           "orig": b.hybrid_forward, "act_block": nxt if act else None, "pool": pool}
            b.hybrid_forward = types.MethodType(_fused_bn_forward, b)
            if act:
                bypass(nxt)
            if pool is not None:
                pool._fq_pool_fused = {"orig": pool.hybrid_forward}
                pool.hybrid_forward = types.MethodType(_pool_after_bn_forward, pool)
            fused[0] += 1

    # the model zoo's nets compute `output(features(x))`: what follows the last block of `features` is `output`
    after_features = {}
    feats, head = getattr(net, "features", None), getattr(net, "output", None)
    if isinstance(feats, (nn.Sequential, nn.HybridSequential)) and head is not None:
        after_features[id(feats)] = list(head._children.values()) \
            if isinstance(head, (nn.Sequential, nn.HybridSequential)) else [head]
    def visit_residual(b):
        if _is_residual_unit(b) and not hasattr(b, "_fq_residual_fused"):
            b._fq_residual_fused = True
            b.forward = types.MethodType(_residual_unit_forward, b)
            fused[0] += 1
        elif _is_linear_bottleneck(b) and b.use_shortcut and not hasattr(b, "_fq_residual_fused"):
            b._fq_residual_fused = True
            b.forward = types.MethodType(_linear_bottleneck_forward, b)
            fused[0] += 1

    def visit_side_links(container):
        """Consecutive residual units - inside a stage, and the last unit of a stage with the first of the next (whose first 1x1
        is strided; its shortcut convolution keeps reading the fp32 trunk): the closing 1x1 of the first learns the first 1x1 of
        the second (`side_next`), for which it may store the trunk a second time as codes (convert_conv2d.side_target; offline
        input quantisation)."""
        if not isinstance(container, (nn.Sequential, nn.HybridSequential)):
            return
        kids = list(container._children.values())

        def unit(b, last):
            """b itself when it is a fused residual unit, or - a stage of such units - its last / first one"""
            if getattr(b, "_fq_residual_fused", False) and hasattr(b, "body") and hasattr(b, "downsample"):
                return b, False
            if isinstance(b, (nn.Sequential, nn.HybridSequential)) and len(b._children):
                inner = list(b._children.values())[-1 if last else 0]
                if getattr(inner, "_fq_residual_fused", False) and hasattr(inner, "body") and hasattr(inner, "downsample"):
                    return inner, True
            return None, False
        for a_, b_ in zip(kids, kids[1:]):
            (u, ua), (v, va) = unit(a_, True), unit(b_, False)
            if u is None or v is None or ua != va:             # (two units of one stage, or two stages)
                continue
            tail = _tail_conv(u.body)
            body = list(v.body._children.values()) if isinstance(v.body, (nn.Sequential, nn.HybridSequential)) else []
            first = body[0] if body else None
            if tail is None or type(first) is not nn.Conv2D or not hasattr(first, "quantize_args"):
                continue
            fa, fb = getattr(tail, "_fq_pw_fused", None), getattr(first, "_fq_pw_fused", None)
            if fa is None or fb is None or fa.get("kind") != "1x1" or fb.get("kind") != "1x1":
                continue
            if first._kwargs["kernel"] != (1, 1) or first._kwargs["stride"] not in ((1, 1), (2, 2)):
                continue
            fa["side_next"] = first
            # a stage boundary of the v1 bottleneck nets: BOTH readers of the trunk - `first` and the shortcut convolution - are
            # 1x1 with stride 2 and no padding, so only a quarter of it is ever read (convert_conv2d.sub_target)
            ds = v.downsample
            ds_kids = list(ds._children.values()) if isinstance(ds, (nn.Sequential, nn.HybridSequential)) else []
            sc = ds_kids[0] if ds_kids else None

            def strided_1x1(c):
                return (type(c) is nn.Conv2D and c._kwargs["kernel"] == (1, 1) and c._kwargs["stride"] == (2, 2)
                        and c._kwargs["pad"] == (0, 0) and c._kwargs["num_group"] == 1
                        and getattr(c, "_fq_pw_fused", None) is not None and c._fq_pw_fused.get("kind") == "1x1")
            # (the containers the tensor passes through on its way - the two stages, their parent, the shortcut's Sequential - must
            # be the stock ones: a subclass with a forward of its own might look at the tensor between two blocks)
            stock = all(type(c) in (nn.Sequential, nn.HybridSequential) and "forward" not in c.__dict__
                        for c in (container, a_, b_, ds))
            if ua and stock and sc is not None and strided_1x1(first) and strided_1x1(sc) and hasattr(sc, "quantize_args"):
                fa["sub_next"] = {"readers": (first, sc), "unit": v, "via": (u, v, a_, b_, ds)}

    def visit_unit_links(container):
        """MobileNetV2: a unit WITHOUT shortcut followed by another unit without shortcut (32 -> 16 then 16 -> 24: the 16-channel
        tensor at 112 x 112): the projection's only reader is the next unit's first 1x1 - the int8 hand-over link across the two
        containers (`via`: no hand-over past a hook on either of them)."""
        if not isinstance(container, (nn.Sequential, nn.HybridSequential)) or not UNIT_LINKS:
            return
        kids = list(container._children.values())

        def first_conv(seq):
            while isinstance(seq, (nn.Sequential, nn.HybridSequential)) and len(seq._children):
                seq = list(seq._children.values())[0]
            return seq if type(seq) is nn.Conv2D else None
        for a_, b_ in zip(kids, kids[1:]):
            if not (_is_linear_bottleneck(a_) and not a_.use_shortcut):
                continue
            tail = _tail_conv(a_.out)
            # (the 320-channel tensor in front of the last 1x1 has the same property, but its producer - 960 -> 320 at 7 x 7 -
            # has no codes-in-and-out instantiation)
            first = first_conv(b_.out) if _is_linear_bottleneck(b_) and not b_.use_shortcut else None
            fa = getattr(tail, "_fq_pw_fused", None) if tail is not None else None
            fb = getattr(first, "_fq_pw_fused", None) if first is not None else None
            if fa is None or fb is None or fa.get("next") is not None or fb.get("kind") != "1x1" or fb.get("sliced") \
                    or not hasattr(first, "quantize_args") or not hasattr(tail, "quantize_args"):
                continue
            fa["next"] = first
            fa["via"] = (a_, b_)

    def visit_dense(b):
        if type(b) is nn.Dense and hasattr(b, "quantize_args") and dense_int8 and not hasattr(b, "_fq_dense_int8"):
            b._fq_dense_int8 = True                # convert_dense: the classifier on the integer codes
            fused[0] += 1

    net.apply(visit_dense)
    if stem:
        net.apply(visit_stem)
    if depthwise:
        net.apply(visit_dw)
    if pointwise_int8:
        net.apply(visit_pw)
    net.apply(visit_gap)
    net.apply(visit)
    if stem and pointwise_int8:
        net.apply(link_stem)
    if residual:
        net.apply(visit_residual)
        net.apply(visit_side_links)
    if pointwise_int8:
        net.apply(visit_unit_links)
    _install_stat_arena(net, fused[0])
    return fused[0]


def library_gemm_blocks(net):
    """The Dense blocks of `net` whose product runs as a GEMM of the tensor library (rocBLAS / hipBLASLt through torch): those
    with an activation of their own and those `fuse_inference` did not put on the integer codes.  An evaluation loop keeps
    SEVERAL batches in flight only for nets without such blocks: vgg11's forward (Dense(25088 -> 4096, relu) through the library)
    replayed as three hipGraphs on three streams never returned on an MI355X, while one stream and eager launches ran - the
    library's kernels are not ours to vouch for when they compete with each other for CUs (profiles/r6_vgg_streams.txt)."""
    found = []

    def visit(b):
        if type(b) is nn.Dense and (b.act is not None or not getattr(b, "_fq_dense_int8", False)):
            found.append(b)
    net.apply(visit)
    return found


def _install_stat_arena(net, producers):
    """One zeroing launch per forward for all producers' per-sample statistic rows (ops.StatArena)."""
    if producers <= 0 or hasattr(net, "_fq_arena_hooks"):
        return
    state = {}                                          # (device, stream) -> arena: forwards in flight on different streams
                                                        # (an evaluation loop with several batches in flight) do not share rows
    inner = net.forward

    def forward(self, *args):
        """The net's forward between `arena.begin` and `arena.end` - in a try / finally, so that a forward that raises (an ops
        ValueError, out of memory) leaves no arena "current" and no cached stream behind for later block calls."""
        x = args[0]
        dev = x._t.device
        if dev.type != "cuda":
            return inner(*args)
        cur = torch.cuda.current_stream(dev)
        key = (dev.index, cur.cuda_stream)
        arena = state.get(key)
        if arena is None:
            arena = state[key] = ops.StatArena(producers + 8, dev)
        outer = getattr(ops.StatArena._tls, "forward", None)
        arena.begin(x.shape[0])
        # what every block of this forward would otherwise ask torch for again (27 x two stream look-ups per forward);
        # "side" = an evaluation loop declared batches in flight (ops.batches_in_flight) AND this is not the default stream
        ops.StatArena._tls.forward = (dev, key, ops.in_flight() and cur != torch.cuda.default_stream(dev))
        try:
            return inner(*args)
        finally:
            arena.end()
            ops.StatArena._tls.forward = outer
    net.forward = types.MethodType(forward, net)
    net._fq_arena_hooks = True


def refresh(net):
    """Recompute the folded BatchNorm constants (after in-place parameter changes)."""
    from .convert.convert import bump_mode_epoch
    bump_mode_epoch()

    def visit(b):
        if hasattr(b, "_fq_fused"):
            b._fq_fused["key"] = None
            b._fq_refresh = b.__dict__.get("_fq_refresh", 0) + 1
        b.__dict__.pop("_fq_pw_cache", None)
    net.apply(visit)


DENSE_INT8 = _os.environ.get("FQ_DENSE_INT8", "1") != "0"
STEM_POOL = _os.environ.get("FQ_STEM_POOL", "1") != "0"      # A/B: 0 = first convolution and max-pool as two launches


class EvalHead(object):
    """The evaluation counters of the reference CLI (simulate_quantization.py:122-148) riding on the classifier's launch
    (fq_dense_i8_eval).  Per batch: `head.labels = y` before `net(X)`; `head.take()` afterwards says whether that forward
    counted (the caller runs `ops.eval_counters` itself when it did not: Dense not on the integer path in this mode)."""
    __slots__ = ("block", "counters", "labels", "counted")

    def __init__(self, block, counters):
        self.block, self.counters, self.labels, self.counted = block, counters, None, False

    def take(self):
        done, self.counted, self.labels = self.counted, False, None
        return done

    def release(self):
        self.block.__dict__.pop("_fq_eval_head", None)


def eval_head(net, counters):
    """An EvalHead when the logits `net` returns are the output of a Dense that `fuse_inference(dense_int8=True)` put on the
    integer codes (the model zoo's `net.output`: a Dense, or a container ending in one), else None."""
    out = getattr(net, "output", None)
    while isinstance(out, (nn.Sequential, nn.HybridSequential)) and len(out._children):
        out = list(out._children.values())[-1]
    if type(out) is not nn.Dense or not getattr(out, "_fq_dense_int8", False):
        return None
    head = EvalHead(out, counters)
    out.__dict__["_fq_eval_head"] = head
    return head


def unfuse(net):
    from .convert.convert import bump_mode_epoch
    bump_mode_epoch()
    if hasattr(net, "_fq_arena_hooks"):
        del net.forward                                     # the class's own forward again (_install_stat_arena)
        del net._fq_arena_hooks

    def restore_act(blk):
        if blk is not None and hasattr(blk, "_fq_bypassed_orig"):
            blk.hybrid_forward = blk._fq_bypassed_orig
            del blk._fq_bypassed_orig

    def visit(b):
        if hasattr(b, "_fq_pool_fused"):
            b.hybrid_forward = b._fq_pool_fused["orig"]
            del b._fq_pool_fused
            b.__dict__.pop("_fq_pool_done", None)
        if hasattr(b, "_fq_dense_int8"):
            del b._fq_dense_int8
            b.__dict__.pop("_fq_wcodes_cache", None)
            b.__dict__.pop("_fq_eval_head", None)
        if hasattr(b, "_fq_residual_fused"):
            del b.forward                                   # the class's own forward again
            del b._fq_residual_fused
        if hasattr(b, "_fq_gap_fused"):
            st = b._fq_gap_fused
            b.hybrid_forward = st["orig"]
            if st["flatten"] is not None:
                st["flatten"].hybrid_forward = st["flatten_orig"]
            del b._fq_gap_fused
        if hasattr(b, "_fq_stem_fused"):
            st = b._fq_stem_fused
            b.hybrid_forward = st["orig"]
            restore_act(st["act_block"])
            del b._fq_stem_fused
        if hasattr(b, "_fq_dw_fused"):
            st = b._fq_dw_fused
            if st["orig"] is not None:
                b.hybrid_forward = st["orig"]
            restore_act(st["act_block"])
            del b._fq_dw_fused
        if hasattr(b, "_fq_pw_fused"):
            restore_act(b._fq_pw_fused["act_block"])
            del b._fq_pw_fused
            if hasattr(b, "_fq_pw_cache"):
                del b._fq_pw_cache
            b.__dict__.pop("_fq_pw_live", None)
        if hasattr(b, "_fq_fused"):
            st = b._fq_fused
            b.hybrid_forward = st["orig"]
            restore_act(st["act_block"])
            del b._fq_fused
    net.apply(visit)
