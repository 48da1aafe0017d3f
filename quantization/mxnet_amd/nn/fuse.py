"""Producer fusion for nets built from `nn.Conv2D(quantized=True)` (nn/quantized_mobilenet.py; the reference's
tests/models/quantized_mobilenet.py): what sits between two quantised convolutions there is BatchNorm and ReLU as separate
blocks (:57-67), and every quantised convolution starts with a pass over its input for the global range (`quantize`,
nn/quantized_conv.py:63-72).  `fuse_inference(net)` re-schedules that without changing what a layer computes from its input:

  * a quantised convolution followed by BatchNorm [+ ReLU / ReLU6] takes both into its store - the dequantised value times
    the BatchNorm's folded scale plus its shift (multiply and add separately rounded, exactly `fq_bn_act_stat`'s arithmetic),
    then the activation - and leaves the per-sample maximum of what it wrote on the NDArray (`_fq_stat`, `_fq_nonneg`);
  * the next quantised convolution takes its range from that statistic instead of reading its input once more: [0, max]
    with padding (the padding zero IS the minimum), and without padding a scan of the input that stops at its first zero
    (`fq_qconv2d_forward`, `in_stat`) - the range, hence every code, is the one the range pass would have produced;
  * the float first convolution (3 -> 32, 3x3 / stride 2) + BatchNorm + ReLU runs as the fused first-convolution kernel of
    the simulated-quantisation path (`fq_stem_conv3x3s2`), which also leaves the statistic for the first depthwise layer.

Per quantised layer that is one small launch for the range record and one convolution launch that reads its input once and
writes its output once (plus, for unpadded uint8 layers, the conditional exact recomputation that returns at once).
`unfuse(net)` restores the blocks.  `FQ_QCONV_NO_STAT=1` keeps the range passes (A/B runs and tests: identical results).
"""
import os

from ..mx.gluon import nn
from ..mx.ndarray import NDArray
from .. import ops
from .quantized_conv import Conv2D as QConv2D

__all__ = ["fuse_inference", "unfuse"]


def _identity_forward(self, F, x, *args, **kwargs):
    return x


def _act_of(block):
    """"relu" / "relu6" when `block` is such an activation (nn.Activation('relu'), a `RELU6`-style clip block), else None."""
    if isinstance(block, nn.Activation):
        return "relu" if block._act_type == "relu" else None
    if type(block).__name__ == "RELU6":
        return "relu6"
    return None


def _stem_ok(conv):
    shp = conv.weight.shape
    return (not conv._quantized and conv._groups == 1 and conv.bias is None and conv.act is None and len(shp) == 4
            and ops.stem_conv_supported(shp[1], shp[0], conv._kernel_size, conv._strides, conv._padding))


def fuse_inference(net):
    """Returns the number of convolutions that took their BatchNorm / activation over."""
    from ..quantize import fuse as qfuse
    fused = [0]

    def visit(seq):
        if not isinstance(seq, (nn.Sequential, nn.HybridSequential)):
            return
        kids = list(seq._children.values())
        for i, b in enumerate(kids):
            if not isinstance(b, QConv2D) or hasattr(b, "_fq_qfuse") or b.act is not None:
                continue
            if not ((b._quantized and b._fused_ok()) or _stem_ok(b)):
                continue
            bn = kids[i + 1] if i + 1 < len(kids) and type(kids[i + 1]) is nn.BatchNorm else None
            if bn is None or hasattr(bn, "_fq_qfused_by"):
                continue
            nxt = kids[i + 2] if i + 2 < len(kids) else None
            act = _act_of(nxt)
            b._fq_qfuse = {"bn": bn, "constants": qfuse._bn_constants_getter(bn), "act": act or "none",
                           "act_block": nxt if act else None, "stem": not b._quantized, "wt": None, "wkey": None}
            bn._fq_qfused_by = b
            bn._fq_qorig = bn.hybrid_forward
            bn.hybrid_forward = _identity_forward.__get__(bn)
            if act:
                nxt._fq_qorig = nxt.hybrid_forward
                nxt.hybrid_forward = _identity_forward.__get__(nxt)
            fused[0] += 1
    net.apply(visit)
    # one zeroing launch per forward for every layer's per-sample statistic row (ops.StatArena) instead of one memset per
    # layer (27 x 4 us in the MobileNet step, profiles/r4_qconv_kernel_stats.csv)
    qfuse._install_stat_arena(net, fused[0])
    return fused[0]


def unfuse(net):
    if hasattr(net, "_fq_arena_hooks"):
        del net.forward                                     # the class's own forward again (quantize.fuse._install_stat_arena)
        del net._fq_arena_hooks

    def visit(b):
        st = b.__dict__.pop("_fq_qfuse", None)
        if st is None:
            return
        for blk in (st["bn"], st["act_block"]):
            if blk is not None and "_fq_qorig" in blk.__dict__:
                blk.hybrid_forward = blk.__dict__.pop("_fq_qorig")
        st["bn"].__dict__.pop("_fq_qfused_by", None)
    net.apply(visit)


def use_producer_stat():
    return os.environ.get("FQ_QCONV_NO_STAT", "0") != "1"


def stem_forward(conv, x, weight):
    """The float first convolution with its BatchNorm / ReLU and the statistic of the result (fq_stem_conv3x3s2)."""
    st = conv._fq_qfuse
    w = weight._t
    key = (w.data_ptr(), w._version)
    if st["wkey"] != key:
        st["wt"], st["wkey"] = w.permute(1, 2, 3, 0).contiguous(), key
    scale, shift = st["constants"]()
    t = x._t if x._t.is_contiguous() else x._t.contiguous()
    y, stat = ops.stem_conv_s2(t, w, None, bn_scale=scale, bn_shift=shift, act=st["act"], want_stat=True,
                               w_tap_major=st["wt"])
    out = NDArray(y)
    out._fq_stat = stat
    out._fq_nonneg = st["act"] in ("relu", "relu6")
    return out
