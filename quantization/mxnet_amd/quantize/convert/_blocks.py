"""Machinery shared by the block converters, `qparams_init` and `merge_bn` (this project's own structure; the public
names and per-block attributes it produces are the reference's API).

  rebind_forward(block, fn)          swap `hybrid_forward`, keep the stock one as `origin_forward`
  Threshold.attach(block, ...)       the calibrated range of a block as a (1,) non-trainable Gluon Parameter + its flags
  BatchNormTerms                     gamma / beta / mean / var of the BatchNorm that follows a convolution, looked up by the
                                     reference's naming rule, and the fold arithmetic written once
  ensure_bias(conv)                  give a bias-free convolution a zero bias Parameter (a folded BatchNorm needs one)
  as_rows(t)                         contiguous view helper for the HIP entry points
"""
import types

FOLD_EPS = 1e-10      # the reference folds with sqrt(var + 1e-10), not with the BatchNorm's own epsilon


def rebind_forward(block, fn, keep_origin=True):
    if keep_origin:
        block.origin_forward = block.hybrid_forward
    block.hybrid_forward = types.MethodType(fn, block)
    return block


def passthrough(block, F, x, *unused_args, **unused_kwargs):
    """hybrid_forward of a bypassed block"""
    return x


class Threshold(object):
    """Naming scheme of one calibrated range: Parameter `<what>_max`, flag `quantize_<kind>_offline`, running value
    `current_<what>_max`."""

    def __init__(self, what, kind):
        self.param = what + "_max"
        self.offline_flag = "quantize_" + kind + "_offline"
        self.current = "current_" + what + "_max"

    def attach(self, block):
        setattr(block, self.offline_flag, False)
        setattr(block, self.current, 0.)
        setattr(block, self.param, block.params.get(self.param, shape=(1,), init="zeros", differentiable=False,
                                                    allow_deferred_init=True))


INPUT_RANGE = Threshold("input", "input")
OUTPUT_RANGE = Threshold("act", "act")


def ensure_bias(conv, ctx=None):
    if conv.bias is None:
        conv._kwargs['no_bias'] = False
        conv.bias = conv.params.get('bias', shape=(conv.weight.shape[0],), allow_deferred_init=True, init="zeros")
        conv.bias.initialize(ctx=ctx)
    return conv.bias


class BatchNormTerms(object):
    """The four vectors of an inference BatchNorm and what folding them into a convolution means
    (convert_conv2d.py:47-51 == freeze/merge_bn.py:62-73):  w' = w * gamma / sqrt(var + 1e-10) per output channel,
    b' = gamma * (b - mean) / sqrt(var + 1e-10) + beta."""
    FIELDS = ("gamma", "beta", "running_mean", "running_var")

    def __init__(self, gamma, beta, mean, var):
        self.gamma, self.beta, self.mean, self.var = gamma, beta, mean, var

    @classmethod
    def of_sibling(cls, conv, params, conv_name, bn_name):
        """Parameters of the BatchNorm whose name is the convolution's with `conv_name` replaced by `bn_name`
        (initialize.py:51-54); None when the net has no such block."""
        stem = conv.name.replace(conv_name, bn_name) + "_"
        found = [params[stem + f] if (stem + f) in params else None for f in cls.FIELDS]
        if any(p is None for p in found):
            return None
        return cls(*[p.data() for p in found])

    def fold_weight(self, F, weight):
        """(w * gamma) / sqrt(var + 1e-10), in that order (the order decides the last bit)"""
        rows = weight.reshape(weight.shape[0], -1)
        scaled = rows * self.gamma.reshape(-1, 1) / F.sqrt(self.var + FOLD_EPS).reshape(-1, 1)
        return scaled.reshape(weight.shape)

    def fold_bias(self, F, bias):
        return self.gamma * (bias - self.mean) / F.sqrt(self.var + FOLD_EPS) + self.beta


def contiguous(t):
    return t if t.is_contiguous() else t.contiguous()


def _stream_of(t):
    """(key, side) of the stream the launches for `t` go to: key = (device index, stream handle); side = the caller declared
    an evaluation loop with several batches in flight (`with ops.batches_in_flight():`, an explicit opt-in: bench.py, the
    CLI's `evaluate`) AND this is not the device's default stream.  A forward on a non-default stream WITHOUT that
    declaration - a calibration or training loop under `torch.cuda.stream(s)`, a framework that installs per-thread
    streams - is an ordinary forward: it updates the block's `current_*_max`, and `update_ema` sees it.  Inside a fused
    net's forward the answer is the one quantize/fuse.py looked up once for the whole forward."""
    import torch
    from ... import ops
    fwd = getattr(ops.StatArena._tls, "forward", None)
    if fwd is not None and fwd[0] == t.device:
        return fwd[1], fwd[2]
    if not t.is_cuda:
        return None, False
    cur = torch.cuda.current_stream(t.device)
    return (t.device.index, cur.cuda_stream), ops.in_flight() and cur != torch.cuda.default_stream(t.device)


def on_side_stream(t):
    """True when this forward is one of several batches in flight (declared, on a stream of its own).  Such a forward
    writes its batch statistic into a slot of its own (`scalar_slot`), so that forwards in flight do not meet in a block's
    `current_*_max`; calibration (`update_ema`) refuses to run inside such a declaration."""
    return _stream_of(t)[1]


def scalar_slot(block, like):
    """(1,) device tensor receiving this block's current batch statistic (`current_input_max` / `current_act_max`): the
    block's own slot - a slice of the net's calibration arena once `net.update_ema()` has bound one (convert.py) - or, on a
    side stream, a slot private to (block, stream).  Returns (slot, side)."""
    import torch
    key, side = _stream_of(like)
    if side:
        slots = block.__dict__.setdefault("_fq_cur_side", {})
        slot = slots.get(key)
        if slot is None:
            slot = slots[key] = torch.zeros(1, dtype=torch.float32, device=like.device)
        return slot, True
    slot = getattr(block, "_fq_cur", None)
    if slot is None or slot.device != like.device:
        slot = block._fq_cur = torch.zeros(1, dtype=torch.float32, device=like.device)
    return slot, False
