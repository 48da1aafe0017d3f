"""`nn.Conv2D` — the reference's stand-alone "really quantised" convolution (nn/quantized_conv.py:79-175).

Same constructor and semantics: per-tensor quantise input and weight to int32 codes (`quantize` / `_quantize`, :54-72:
global range, scale = max/127 if symmetric else (max-min)/255, NO zero-point), int32 bias at scale in_s*w_s clipped to
+-scale*2^31 (:122-127), integer correlation, optional activation, `dequantize` by in_s*w_s (:74-76, :158).

What differs is the machinery.  With `quantized=True` a forward is ONE call into the library (`ops.qconv2d` ->
`fq_qconv2d_forward`, csrc/fq_qconv.hip): a 4 B/elem pass for the input's global range, one small launch for the range
record / int32 bias codes / per-channel constants, and the convolution with the quantiser ON ITS LOADS - the 1x1 layers on
the int8 matrix cores (the pointwise forms of the model path in range mode), dense 3x3 (Cin 64 ... 512) on the implicit-GEMM
kernel, depthwise 3x3 on the depthwise forms with integer codes in the fp32 chain (exact: 9 taps), every other geometry
on an exact one-output-per-thread kernel - integer sums in wrapping int32 for ANY accumulator size, where the reference's
fp32 `dot` (:140-144) is exact only below 2^24.  No padded copy, no im2col tensor, no int32 code tensor, no casts; ranges and
scales stay in device scalars (no `.asscalar()`).  The weights' codes are kept while the Parameter is the same tensor in the
same in-place version (the reference re-quantises them every forward: identical values).

Two corners keep the round-3 formulation (an im2col of the CODES + `fq_gemm_i8_codes`, or the fp32-held integer product):
an `activation` other than relu (the reference applies it to the int32 tensor before `dequantize`), and
`FQ_QCONV_LEGACY=1` (A/B runs).
"""
import os

import torch
import torch.nn.functional as TF

from ..mx.gluon import nn
from ..mx.ndarray import NDArray
from .. import ops

__all__ = ['Conv2D', 'quantize', 'dequantize']


def _pair(v):
    """an int means the same value for height and width"""
    return (v, v) if isinstance(v, int) else tuple(v)


def quantize(F, x, out_type='int8'):
    """(:63-72) -> (int32 codes NDArray, scale as a (1,) device NDArray)."""
    if out_type not in ('int8', 'uint8'):
        raise ValueError("unknown out type: ", out_type)
    codes, rng = ops.quantize_codes(x._t.contiguous(), out_type)
    return NDArray(codes), NDArray(rng[2:3])


def _quantize(F, x, min_range, max_range):
    """(:54-61) with a caller-fixed range."""
    rng = torch.tensor([float(min_range), float(max_range), 0.0], dtype=torch.float32, device=x._t.device)
    codes, rng = ops.quantize_codes(x._t.contiguous(), "range", rng)
    return NDArray(codes), NDArray(rng[2:3])


def dequantize(F, x, scale):
    """(:74-76)"""
    st = scale._t if isinstance(scale, NDArray) else torch.tensor([float(scale)], dtype=torch.float32,
                                                                  device=x._t.device)
    return NDArray(ops.dequantize(x._t.contiguous(), st.reshape(1)))


class Conv2D(nn.HybridBlock):
    """Constructor signature of the reference's block (nn/quantized_conv.py:80-84).  `_input_range` / `_weight_range`
    (None, or a (min, max) pair) fix the quantisation ranges instead of taking them from the tensors (:112-120)."""

    def __init__(self, channels, kernel_size, strides, padding, in_channels, groups=1,
                 activation=None, use_bias=True, quantized=False,
                 input_dtype='float32', weight_dtype='float32',
                 weight_initializer=None, bias_initializer='zeros',
                 prefix=None, params=None):
        super(Conv2D, self).__init__(prefix, params)
        if in_channels % groups or channels % groups:
            raise AssertionError("groups=%d must divide in_channels=%d and channels=%d" % (groups, in_channels, channels))
        self._groups = groups
        self._kernel_size, self._strides, self._padding = _pair(kernel_size), _pair(strides), _pair(padding)
        self._quantized, self._input_dtype, self._weight_dtype = quantized, input_dtype, weight_dtype
        self._input_range = self._weight_range = None
        with self.name_scope():
            self.weight = self.params.get('weight', init=weight_initializer, allow_deferred_init=True,
                                          shape=(channels, in_channels // groups) + self._kernel_size)
            self.bias = None
            if use_bias:
                self.bias = self.params.get('bias', init=bias_initializer, allow_deferred_init=True, shape=(channels,))
            self.act = None if activation is None else nn.Activation(activation, prefix=activation + '_')

    def _alias(self):
        return "conv2d"

    # ---- quantized=True: one library call per forward ---------------------------------------------------------------------
    def _fused_ok(self):
        act = None if self.act is None else self.act._act_type
        return self._quantized and act in (None, "relu") and os.environ.get("FQ_QCONV_LEGACY", "0") != "1"

    def _prepared_weights(self, w):
        """ops.qconv_weights of the weight, kept until the Parameter's storage or in-place version changes."""
        key = (w._version, self._weight_dtype, None if self._weight_range is None else tuple(self._weight_range))
        held = self.__dict__.get("_fq_qw")
        if held is None or held[0] is not w or held[1] != key:
            buf = ops.qconv_weights(w, self._strides, self._padding, self._groups, self._weight_dtype, self._weight_range)
            held = self.__dict__["_fq_qw"] = (w, key, buf)
        return held[2]

    def _workspace(self, device):
        key = (device.index, torch.cuda.current_stream(device).cuda_stream) if device.type == "cuda" else None
        slots = self.__dict__.setdefault("_fq_qws", {})
        ws = slots.get(key)
        if ws is None:
            ws = slots[key] = ops.qconv_workspace(self.weight.shape[0], device)
        return ws

    def _forward_fused(self, inputs, weight, bias):
        x = inputs._t if inputs._t.is_contiguous() else inputs._t.contiguous()
        w = weight._t if weight._t.is_contiguous() else weight._t.contiguous()
        ops.require_hip(x.device, "nn.Conv2D(quantized=True) input")
        b = None if bias is None else bias._t.contiguous()
        # (uint8 / fixed-range weights are not on one symmetric int8 grid: the exact direct kernel takes them)
        direct = self._weight_dtype != 'int8' or self._weight_range is not None
        fz = self.__dict__.get("_fq_qfuse")            # nn/fuse.py: BatchNorm [+ ReLU] behind this block folded into the store
        kw = {}
        if fz is not None:
            scale, shift = fz["constants"]()
            kw = dict(bn_scale=scale, bn_shift=shift, want_stat=True)
        act = "none" if self.act is None else "relu"
        if fz is not None:
            act = fz["act"]
        in_stat = None
        if self._input_range is None and x is inputs._t and inputs._fq_stat is not None and inputs._fq_nonneg:
            from . import fuse as _qfuse
            if _qfuse.use_producer_stat() and inputs._fq_stat.numel() == x.shape[0]:
                in_stat = inputs._fq_stat               # the producer's per-sample maxima: no range pass
        y = ops.qconv2d(x, w, self._prepared_weights(w), b, self._strides, self._padding, self._groups,
                        self._workspace(x.device), input_dtype=self._input_dtype, input_range=self._input_range,
                        act=act, force_direct=direct, in_stat=in_stat, **kw)
        if fz is None:
            return NDArray(y)
        out = NDArray(y[0])
        out._fq_stat = y[1]
        out._fq_nonneg = act in ("relu", "relu6")
        return out

    def hybrid_forward(self, F, inputs, weight, bias=None):
        if self._fused_ok():
            return self._forward_fused(inputs, weight, bias)
        if not self._quantized and self.__dict__.get("_fq_qfuse") is not None:
            from . import fuse as _qfuse
            return _qfuse.stem_forward(self, inputs, weight)
        # Pad (:108-109)
        ph, pw = self._padding
        x = TF.pad(inputs._t, (pw, pw, ph, ph), mode="constant", value=0.0).contiguous()
        w = weight._t.contiguous()
        b = None if bias is None else bias._t
        if self._quantized:
            # Quantize and cast into int32 (:111-127)
            if self._input_range is None:
                xi, in_scale = quantize(F, NDArray(x), self._input_dtype)
            else:
                xi, in_scale = _quantize(F, NDArray(x), *self._input_range)
            if self._weight_range is None:
                wi, w_scale = quantize(F, NDArray(w), self._weight_dtype)
            else:
                wi, w_scale = _quantize(F, NDArray(w), *self._weight_range)
            b_scale = (in_scale._t * w_scale._t)                     # fp32 product, device scalar
            if b is not None:
                b_max = b_scale * float(2 ** 31)
                rng = torch.cat([-b_max, b_max, b_scale]).contiguous()
                bi, _ = ops.quantize_codes(b.contiguous(), "scale", rng)
                b = bi.to(torch.float32)
            x, w = xi._t.to(torch.float32), wi._t.to(torch.float32)
        int8_codes = self._quantized and self._weight_dtype == 'int8' and self._weight_range is None
        k_group = w.shape[1] * w.shape[2] * w.shape[3]                  # dot length of one output: (Cin / groups) * kh * kw
        # 255 * 127 * K < 2^24: every partial sum of the integer codes is exact in fp32 in any order
        fp32_exact = 255 * 127 * k_group < 2 ** 24
        if int8_codes and (self._groups == 1 or not fp32_exact):
            # Convolution on the int8 matrix cores: im2col of the CODES (the reference's slices, :34-52),
            # then fq_gemm_i8_codes - exact int32 for any accumulator size, where the fp32 formulation below (and the
            # reference's own fp32 `dot`, :140-144) is only exact below 2^24.  Grouped convolutions whose dot length could
            # pass 2^24 go group by group, as the reference's loop does (:129-151).
            n, _, hp, wp = x.shape
            kh, kw = self._kernel_size
            ho, wo = (hp - kh) // self._strides[0] + 1, (wp - kw) // self._strides[1] + 1
            zoff = 128 if self._input_dtype == 'uint8' and self._input_range is None else 0
            if self._input_range is not None:
                zoff = 128 if float(self._input_range[0]) >= 0 else 0   # `_quantize`: codes in [0,255] or [-127,127]
            g = self._groups
            cin_g, cout_g = x.shape[1] // g, w.shape[0] // g
            parts = []
            for gi in range(g):
                xg = x if g == 1 else x[:, gi * cin_g:(gi + 1) * cin_g]
                wg = w if g == 1 else w[gi * cout_g:(gi + 1) * cout_g]
                cols = TF.unfold(xg, (kh, kw), stride=self._strides)     # (n, C*kh*kw, L): small exact integers
                xc = (cols.transpose(1, 2).reshape(n * ho * wo, -1) - float(zoff)).to(torch.int8)
                wc = wg.reshape(wg.shape[0], -1).to(torch.int8)
                parts.append(ops.gemm_i8_codes(xc.contiguous(), wc.contiguous(), n, ho * wo, zoff).reshape(n, -1, ho, wo))
            y = parts[0] if g == 1 else torch.cat(parts, dim=1)
            if b is not None:
                y = y + b.to(torch.int32).reshape(1, -1, 1, 1)
        elif self._quantized:
            # Grouped / depthwise convolutions with short dot products (depthwise: K = 9 - nothing for a matrix core) and
            # uint8 weights: the reference's im2col + dot (:129-151) for all groups at once - the window slices of the integer
            # codes held in fp32 times the weight matrix of each group, as ONE batched matrix product.  Products and partial
            # sums are integers: exact in any summation order while 255 * 127 * K < 2^24 (every int8-weight case that reaches
            # this branch), as exact as the reference's own fp32 dot otherwise.  (Not the library convolution: its Winograd
            # kernels transform the operands and are not exact on integers.)
            n, _, hp, wp = x.shape
            kh, kw = self._kernel_size
            ho, wo = (hp - kh) // self._strides[0] + 1, (wp - kw) // self._strides[1] + 1
            g = self._groups
            cols = TF.unfold(x, (kh, kw), stride=self._strides).reshape(n, g, k_group, ho * wo)
            y = torch.matmul(w.reshape(1, g, w.shape[0] // g, k_group), cols).reshape(n, w.shape[0], ho, wo)
            y = y.to(torch.int32)                                      # (:144) cast back to int32
            if b is not None:
                y = y + b.to(torch.int32).reshape(1, -1, 1, 1)
        else:
            # The float path: correlation with the stride of the reference's window loop (:42-47)
            y = TF.conv2d(x, w, None, stride=self._strides, padding=0, groups=self._groups)
            if b is not None:
                y = y + b.reshape(1, -1, 1, 1)
        y = NDArray(y)
        if self.act is not None:
            y = self.act(y if not self._quantized else NDArray(y._t))
        # Dequantize (:157-158)
        if self._quantized:
            yt = y._t if y._t.dtype == torch.int32 else y._t.to(torch.int32)
            y = NDArray(ops.dequantize(yt.contiguous(), b_scale.reshape(1).contiguous()))
        return y

    def __repr__(self):
        cout, cin_g = self.weight.shape[0], self.weight.shape[1]
        parts = ["%s -> %s" % (cin_g if cin_g else None, cout), "kernel_size=%s" % (self._kernel_size,),
                 "stride=%s" % (self._strides,)]
        if any(self._padding):
            parts.append("padding=%s" % (self._padding,))
        if self._groups != 1:
            parts.append("groups=%d" % self._groups)
        if self.bias is None:
            parts.append("bias=False")
        if self.act:
            parts.append(str(self.act))
        return "%s(%s)" % (type(self).__name__, ", ".join(parts))
