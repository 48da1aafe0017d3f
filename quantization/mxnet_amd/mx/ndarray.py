"""`mx.nd`-shaped array facade over torch tensors (device memory + streams are torch's; nothing more).

Why this exists: the reference's host code is "Python over MXNet-Gluon NDArrays"
(SURVEY.md F2/F3); MXNet is not installable on the build or GPU boxes, PyTorch-ROCm is.  This
module gives the handful of NDArray spellings the reference's call sites use
(`x.asscalar()`, `x.asnumpy()`, `x.as_in_context(ctx)`, `x.clip(lo, hi)`, `F.max(F.abs(x), axis=...)`,
`nd.dot`, `F.Convolution`, ...) so that `quantize.convert.convert_model`, the converter factories,
`qparams_init`, `collect_feature_maps` and `nn.Conv2D` keep their reference signatures.

The fake-quant arithmetic of the product does NOT go through these generic ops: the converters call the
HIP kernels in `csrc/` through the ctypes C-ABI (`..ops`).  The generic ops below exist (a) for
the parts of a network that are not on the hot path (convolution, pooling, BN -> MIOpen/rocBLAS via torch)
and (b) so that, on CPU, this package can stand in for `mxnet` when `tools/gen_golden.py` executes the
reference's own Python to produce golden vectors.  For (b) the primitive semantics matter and are pinned
here to MXNet's documented behaviour:

* all arithmetic in fp32; tensor (op) python/numpy scalar == tensor (op) fp32(scalar), IEEE;
* `round` = C `roundf`, half AWAY from zero (torch.round / np.round are half-to-even: not used);
* `clip(a, lo, hi) = min(max(a, lo), hi)`; `cast(.., 'int32')` truncates toward zero;
* `mean` = fp32( fp64-accumulated sum ) / fp32(N)   (deterministic; see DESIGN.md "batch mean");
* `dot` on small contractions (K <= 64) is evaluated k-sequentially in fp32 with separately rounded
  multiply and add (no FMA), so Winograd transforms are reproducible bit-for-bit.
"""
import numbers
import struct

import numpy as np
import torch
import torch.nn.functional as TF

from .context import Context, cpu

__all__ = ["NDArray", "array", "zeros", "ones", "zeros_like", "ones_like", "uniform", "normal", "dot",
           "concat", "stack", "split", "max", "min", "abs", "sqrt", "round", "cast", "clip", "relu",
           "Convolution", "FullyConnected", "Activation", "BatchNorm", "Pooling", "Flatten", "pad",
           "argmax", "sum", "mean", "waitall", "arange", "broadcast_div", "save", "load"]

_DTYPES = {
    "float32": torch.float32, "float64": torch.float64, "float16": torch.float16,
    "int32": torch.int32, "int64": torch.int64, "int8": torch.int8, "uint8": torch.uint8,
    np.float32: torch.float32, np.float64: torch.float64, np.int32: torch.int32, np.int64: torch.int64,
    np.int8: torch.int8, np.uint8: torch.uint8,
}


def _to_dtype(dtype):
    if isinstance(dtype, torch.dtype):
        return dtype
    if dtype in _DTYPES:
        return _DTYPES[dtype]
    return _DTYPES[np.dtype(dtype).name]


def _roundf(t):
    """C roundf: nearest, ties away from zero.  x - trunc(x) is exact in binary fp."""
    tr = torch.trunc(t)
    frac = t - tr
    return torch.where(frac.abs() >= 0.5, tr + torch.sign(t), tr)


class NDArray(object):
    """Thin wrapper: one torch tensor, MXNet method names."""
    # _fq_stat: optional side channel — per-sample max|x| left by a fused producer (quantize/fuse.py) so that the
    # consuming fake-quant can skip its statistic pass.  Never set by the generic ops.
    # _fq_c16: optional ops.Codes16 — `_t` then holds the int8 codes a fused producer handed to its single fused consumer
    # (offline input quantisation; quantize/convert/convert_conv2d.handover_target).  Never set by the generic ops.
    # _fq_nonneg: only meaningful beside a `_fq_stat`: the producer applied ReLU / ReLU6, so the tensor is non-negative and the
    # statistic is its per-sample maximum (what nn.Conv2D(quantized=True) needs to skip its range pass, nn/fuse.py).
    # _fq_kl: optional (producer block, histogram sink or None) left by a fused producer while the KL calibration collects
    # feature maps (quantize/distribution_calibrate.py): who made this tensor, and whether that pass already binned it.
    # _fq_side: optional (consumer block, ops.Codes16) - `_t` is the fp32 trunk of a ResNet and the codes of the SAME values under
    # that consumer's stored threshold ride beside it (fq_pwconv_i8_c16_dual; quantize/convert/convert_conv2d.pointwise_fused).
    # _fq_deferred: optional dict - this NDArray stands for the output of a fused 1x1 convolution that was NOT stored: only its
    # per-sample statistic was computed (`_fq_stat`, fq_pwconv_i8_stat) and the single consumer, the depthwise convolution
    # linked behind it, recomputes the values inside its own launch (fq_pwdw_fused; quantize/convert/convert_conv2d.py).  `_t`
    # is then an int8 placeholder of the right shape without storage, so that any other reader fails loudly.
    # _fq_pooled_by: optional MaxPool2D block - the first convolution's launch pooled already (fq_stem_conv7x7s2_pool) and that
    # block, when it is handed this very tensor, passes it through (quantize/fuse.py).
    # _fq_sub2: optional dict - `_t` holds only [:, :, ::2, ::2] of the tensor this NDArray stands for (`hw`: its plane), because
    # its only readers (`readers`: two Conv2D blocks, 1x1, stride 2, no padding; `unit`: the residual unit that owns them) never
    # look at the rest; `_fq_stat` is the statistic of the WHOLE tensor (fq_pwconv_i8_sub2; convert_conv2d.sub_target).
    # _fq_short: optional dict - this NDArray stands for the output of a residual unit's shortcut convolution (+ BatchNorm) that was
    # NOT computed: the unit's closing 1x1 computes it inside its own launch (fq_pwconv_i8_shortcut) from the record's operands, or
    # whoever else gets hold of it materialises it (convert_conv2d.materialise_shortcut).  `_t` is a placeholder without storage.
    __slots__ = ("_t", "_fq_stat", "_fq_c16", "_fq_nonneg", "_fq_kl", "_fq_side", "_fq_deferred", "_fq_pooled_by", "_fq_sub2",
                 "_fq_short")
    __array_priority__ = 1000.0
    __array_ufunc__ = None

    def __init__(self, t):
        assert isinstance(t, torch.Tensor), type(t)
        self._t = t
        self._fq_stat = None
        self._fq_c16 = None
        self._fq_nonneg = False
        self._fq_kl = None
        self._fq_side = None
        self._fq_deferred = None
        self._fq_pooled_by = None
        self._fq_sub2 = None
        self._fq_short = None

    # -- plumbing ---------------------------------------------------------------------------
    @property
    def shape(self):
        return tuple(self._t.shape)

    @property
    def size(self):
        return self._t.numel()

    @property
    def ndim(self):
        return self._t.dim()

    @property
    def dtype(self):
        return np.dtype(str(self._t.dtype).replace("torch.", ""))

    @property
    def context(self):
        return Context.from_torch(self._t.device)

    ctx = context

    @property
    def T(self):
        return NDArray(self._t.t().contiguous())

    def asnumpy(self):
        return self._t.detach().cpu().numpy()

    def asscalar(self):
        if self._t.numel() != 1:
            raise ValueError("The current array is not a scalar")
        return self.asnumpy().reshape(-1)[0]          # numpy scalar, as MXNet returns

    def as_in_context(self, ctx):
        dev = ctx.torch_device
        if self._t.device == dev:
            return self
        return NDArray(self._t.to(dev))

    def copyto(self, other):
        if isinstance(other, Context):
            return NDArray(self._t.to(other.torch_device, copy=True))
        other._t.copy_(self._t)
        other._fq_stat = None                   # a fused producer's statistic described the OLD contents
        other._fq_kl = None
        other._fq_side = None
        return other

    def copy(self):
        return NDArray(self._t.clone())

    def astype(self, dtype):
        return cast(self, dtype)

    def wait_to_read(self):
        if self._t.is_cuda:
            torch.cuda.current_stream(self._t.device).synchronize()

    def detach(self):
        return NDArray(self._t.detach())

    def __len__(self):
        return self._t.shape[0]

    def __iter__(self):
        for i in range(self._t.shape[0]):
            yield NDArray(self._t[i])

    def __repr__(self):
        return "\n%s\n<NDArray %s @%s>" % (self.asnumpy(), "x".join(map(str, self.shape)), self.context)

    def __bool__(self):
        if self._t.numel() != 1:
            raise ValueError("The truth value of an NDArray with multiple elements is ambiguous.")
        return bool(self._t.item())

    def __float__(self):
        return float(self._t.item())

    def __int__(self):
        return int(self._t.item())

    def __index__(self):
        return int(self._t.item())

    def __hash__(self):
        return id(self)

    def __getitem__(self, key):
        key = _unwrap_key(key)
        return NDArray(self._t[key])

    def __setitem__(self, key, value):
        key = _unwrap_key(key)
        self._fq_stat = None
        self._fq_kl = None
        self._fq_side = None
        self._t[key] = value._t if isinstance(value, NDArray) else value

    # -- shape ops --------------------------------------------------------------------------
    def reshape(self, *shape, **kwargs):
        if "shape" in kwargs:
            shape = kwargs["shape"]
        if len(shape) == 1 and isinstance(shape[0], (tuple, list)):
            shape = shape[0]
        return NDArray(self._t.reshape(tuple(int(s) for s in shape)))

    def transpose(self, *axes, **kwargs):
        if "axes" in kwargs:
            axes = kwargs["axes"]
        if len(axes) == 1 and isinstance(axes[0], (tuple, list)):
            axes = axes[0]
        if len(axes) == 0:
            axes = tuple(reversed(range(self._t.dim())))
        return NDArray(self._t.permute(*axes).contiguous())

    def swapaxes(self, dim1, dim2):
        return NDArray(self._t.transpose(dim1, dim2).contiguous())

    def flatten(self):
        return NDArray(self._t.reshape(self._t.shape[0], -1))

    def expand_dims(self, axis):
        return NDArray(self._t.unsqueeze(axis))

    def pad(self, mode="constant", constant_value=0, pad_width=()):
        return pad(self, mode=mode, constant_value=constant_value, pad_width=pad_width)

    # -- elementwise ------------------------------------------------------------------------
    def abs(self):
        return NDArray(self._t.abs())

    def sqrt(self):
        return NDArray(self._t.sqrt())

    def round(self):
        return NDArray(_roundf(self._t))

    def clip(self, a_min, a_max):
        return clip(self, a_min, a_max)

    def relu(self):
        return NDArray(torch.relu(self._t))

    def __neg__(self):
        return NDArray(-self._t)

    def __add__(self, o):
        return NDArray(self._t + _operand(o, self._t))

    __radd__ = __add__

    def __sub__(self, o):
        return NDArray(self._t - _operand(o, self._t))

    def __rsub__(self, o):
        return NDArray(_operand(o, self._t) - self._t)

    def __mul__(self, o):
        return NDArray(self._t * _operand(o, self._t))

    __rmul__ = __mul__

    def __truediv__(self, o):
        return NDArray(self._t / _operand(o, self._t))

    def __rtruediv__(self, o):
        return NDArray(_operand(o, self._t) / self._t)

    def __pow__(self, o):
        return NDArray(self._t ** _operand(o, self._t))

    def __iadd__(self, o):
        self._fq_stat = None
        self._fq_kl = None
        self._fq_side = None
        self._t += _operand(o, self._t)
        return self

    def __imul__(self, o):
        self._fq_stat = None
        self._fq_kl = None
        self._fq_side = None
        self._t *= _operand(o, self._t)
        return self

    def _cmp(self, o, fn):
        # MXNet comparisons return 0/1 in the input dtype
        return NDArray(fn(self._t, _operand(o, self._t)).to(self._t.dtype))

    def __eq__(self, o):
        return self._cmp(o, torch.eq)

    def __ne__(self, o):
        return self._cmp(o, torch.ne)

    def __lt__(self, o):
        return self._cmp(o, torch.lt)

    def __le__(self, o):
        return self._cmp(o, torch.le)

    def __gt__(self, o):
        return self._cmp(o, torch.gt)

    def __ge__(self, o):
        return self._cmp(o, torch.ge)

    # -- reductions -------------------------------------------------------------------------
    def max(self, axis=None, keepdims=False):
        return max(self, axis=axis, keepdims=keepdims)

    def min(self, axis=None, keepdims=False):
        return min(self, axis=axis, keepdims=keepdims)

    def sum(self, axis=None, keepdims=False):
        return sum(self, axis=axis, keepdims=keepdims)

    def mean(self, axis=None, keepdims=False):
        return mean(self, axis=axis, keepdims=keepdims)

    def argmax(self, axis=None):
        return argmax(self, axis=axis)

    # ---- autograd (mx.autograd over torch's tape) -------------------------------------------------------------------
    def attach_grad(self, grad_req="write"):
        self._t.requires_grad_(grad_req != "null")

    @property
    def grad(self):
        g = self._t.grad
        return None if g is None else NDArray(g)

    def backward(self, out_grad=None, retain_graph=False, train_mode=True):
        from . import autograd
        autograd.backward([self], None if out_grad is None else [out_grad], retain_graph=retain_graph)


def _unwrap_key(key):
    if isinstance(key, NDArray):
        return key._t.long()
    if isinstance(key, tuple):
        return tuple(_unwrap_key(k) for k in key)
    return key


_scalars = {}          # (device, dtype, bits of the value) -> 0-dim tensor: read-only operands, made once
_SCALARS_MAX = 4096


def _operand(o, like):
    """Right-hand operand with MXNet semantics: scalars become fp32(scalar) on the array's device (a DEVICE scalar, so that
    `x / c` is an IEEE division - with a host scalar the tensor library multiplies by the reciprocal).  Cached per value and
    filled by a kernel, not copied from the host: an arithmetic expression with a constant is then legal inside a hipGraph
    capture (examples/qat_finetune.py --graph)."""
    if isinstance(o, NDArray):
        return o._t
    if isinstance(o, torch.Tensor):
        return o
    if isinstance(o, (numbers.Number, np.generic)):
        if like.dtype.is_floating_point:
            # keyed by the BIT PATTERN (0.0 / -0.0 and the NaNs are different operands: x / -0.0, x * -0.0)
            key = (like.device, like.dtype, struct.pack("<d", float(o)))
            t = _scalars.get(key)
            if t is None:
                t = torch.full((), float(o), dtype=like.dtype, device=like.device)
                # an entry is NEVER dropped: a hipGraph captured earlier may hold its address (update_ema's 0.9 / 0.1, a QAT
                # step), and memory the allocator recycled would make every replay read another constant without an error.
                # A run that keeps producing new scalars (a decaying lr, a loss scale) gets uncached tensors once the table is
                # full - correct, only slower.
                if len(_scalars) < _SCALARS_MAX:
                    _scalars[key] = t
            return t
        return o
    if isinstance(o, np.ndarray):
        return torch.from_numpy(o).to(like.device)
    raise TypeError("unsupported operand %r" % type(o))


def _axes(axis):
    if axis is None:
        return None
    if isinstance(axis, int):
        return (axis,)
    return tuple(axis)


# -- creation ---------------------------------------------------------------------------------
def array(source, ctx=None, dtype=None):
    if isinstance(source, NDArray):
        t = source._t.clone()
    else:
        a = np.asarray(source)
        if dtype is None:
            dtype = a.dtype if isinstance(source, np.ndarray) and a.dtype != np.float64 else "float32"
        t = torch.from_numpy(np.ascontiguousarray(a)).to(_to_dtype(dtype))
    if dtype is not None:
        t = t.to(_to_dtype(dtype))
    return NDArray(t.to((ctx or cpu()).torch_device))


def _shape(shape):
    return (shape,) if isinstance(shape, int) else tuple(shape)


def zeros(shape, ctx=None, dtype="float32"):
    return NDArray(torch.zeros(_shape(shape), dtype=_to_dtype(dtype), device=(ctx or cpu()).torch_device))


def ones(shape, ctx=None, dtype="float32"):
    return NDArray(torch.ones(_shape(shape), dtype=_to_dtype(dtype), device=(ctx or cpu()).torch_device))


def zeros_like(a):
    return NDArray(torch.zeros_like(a._t))


def ones_like(a):
    return NDArray(torch.ones_like(a._t))


def arange(start, stop=None, step=1.0, ctx=None, dtype="float32"):
    if stop is None:
        start, stop = 0, start
    return NDArray(torch.arange(start, stop, step, dtype=_to_dtype(dtype), device=(ctx or cpu()).torch_device))


def uniform(low=0.0, high=1.0, shape=(1,), ctx=None, dtype="float32"):
    a = np.random.uniform(low, high, size=_shape(shape)).astype("float32")    # numpy RNG: seedable like the CLI
    return array(a, ctx=ctx, dtype=dtype)


def normal(loc=0.0, scale=1.0, shape=(1,), ctx=None, dtype="float32"):
    a = np.random.normal(loc, scale, size=_shape(shape)).astype("float32")
    return array(a, ctx=ctx, dtype=dtype)


# -- elementwise functions (the `F.` namespace handed to hybrid_forward) -------------------------
def abs(x):
    return x.abs()


def sqrt(x):
    return x.sqrt()


def round(x):
    return x.round()


def relu(x):
    return x.relu()


def clip(x, a_min, a_max):
    t = x._t
    lo = torch.tensor(float(a_min), dtype=t.dtype, device=t.device) if t.dtype.is_floating_point else a_min
    hi = torch.tensor(float(a_max), dtype=t.dtype, device=t.device) if t.dtype.is_floating_point else a_max
    return NDArray(torch.minimum(torch.maximum(t, lo), hi))


def cast(x, dtype):
    return NDArray(x._t.to(_to_dtype(dtype)))          # float -> int truncates toward zero


def broadcast_div(a, b):
    return a / b


# -- reductions ---------------------------------------------------------------------------------
def max(x, axis=None, keepdims=False):
    ax = _axes(axis)
    if ax is None:
        return NDArray(x._t.max().reshape(1))
    return NDArray(torch.amax(x._t, dim=ax, keepdim=keepdims))


def min(x, axis=None, keepdims=False):
    ax = _axes(axis)
    if ax is None:
        return NDArray(x._t.min().reshape(1))
    return NDArray(torch.amin(x._t, dim=ax, keepdim=keepdims))


def sum(x, axis=None, keepdims=False):
    ax = _axes(axis)
    t = x._t
    if t.dtype == torch.float32:
        # fp64 accumulate, one rounding to fp32 (MXNet CPU uses compensated summation; see module doc)
        r = t.double().sum() if ax is None else t.double().sum(dim=ax, keepdim=keepdims)
        r = r.float()
    else:
        r = t.sum() if ax is None else t.sum(dim=ax, keepdim=keepdims)
    return NDArray(r.reshape(1) if ax is None else r)


def mean(x, axis=None, keepdims=False):
    ax = _axes(axis)
    s = sum(x, axis=axis, keepdims=keepdims)
    if ax is None:
        n = x._t.numel()
    else:
        n = 1
        for a in ax:
            n *= x._t.shape[a]
    return s / float(n)


def argmax(x, axis=None):
    # MXNet returns indices as float32
    if axis is None:
        return NDArray(x._t.reshape(-1).argmax().reshape(1).float())
    return NDArray(x._t.argmax(dim=axis).float())


# -- linear algebra / structure ------------------------------------------------------------------
def dot(lhs, rhs, transpose_a=False, transpose_b=False):
    """MXNet `dot`: contracts the LAST axis of lhs with the FIRST axis of rhs (N-D aware).

    K <= 64: k-sequential fp32 multiply-then-add, each separately rounded (bit-reproducible; this is what the
    Winograd weight transform goes through, convert_conv2d.py:73,83).  Larger K: rocBLAS/BLAS matmul.
    """
    a, b = lhs._t, rhs._t
    if transpose_a:
        a = a.transpose(-1, -2) if a.dim() == 2 else a.permute(*reversed(range(a.dim())))
    if transpose_b:
        b = b.transpose(-1, -2) if b.dim() == 2 else b.permute(*reversed(range(b.dim())))
    K = a.shape[-1]
    assert b.shape[0] == K, "dot shape error: %s x %s" % (tuple(a.shape), tuple(b.shape))
    a2 = a.reshape(-1, K)
    b2 = b.reshape(K, -1)
    if K <= 64 and a2.dtype == torch.float32:
        acc = a2[:, 0:1] * b2[0:1, :]
        for k in range(1, K):
            acc = acc + a2[:, k:k + 1] * b2[k:k + 1, :]
        out = acc
    else:
        out = a2 @ b2
    return NDArray(out.reshape(tuple(a.shape[:-1]) + tuple(b.shape[1:])).contiguous())


def concat(*arrays, dim=1):
    return NDArray(torch.cat([a._t for a in arrays], dim=dim))


def stack(*arrays, axis=0):
    return NDArray(torch.stack([a._t for a in arrays], dim=axis))


def split(x, num_outputs, axis=1, squeeze_axis=False):
    parts = torch.chunk(x._t, num_outputs, dim=axis)
    outs = [NDArray((p.squeeze(axis) if squeeze_axis else p).contiguous()) for p in parts]
    return outs[0] if num_outputs == 1 else outs


def pad(x, mode="constant", constant_value=0, pad_width=()):
    assert mode == "constant"
    pw = list(pad_width)
    # MXNet: (before_0, after_0, before_1, after_1, ...); torch: last dim first
    pairs = [(pw[2 * i], pw[2 * i + 1]) for i in range(len(pw) // 2)]
    flat = []
    for before, after in reversed(pairs):
        flat += [before, after]
    return NDArray(TF.pad(x._t, flat, mode="constant", value=constant_value))


def Flatten(x):
    return x.flatten()


# -- neural-network ops that are NOT on the fake-quant hot path: MIOpen / rocBLAS via torch ---------
def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


def Convolution(data, weight, bias=None, kernel=None, stride=(1, 1), dilate=(1, 1), pad=(0, 0),
                num_filter=None, num_group=1, no_bias=False, layout="NCHW", name=None, **_ignored):
    assert layout == "NCHW"
    b = None if (no_bias or bias is None) else bias._t
    return NDArray(TF.conv2d(data._t, weight._t, b, stride=_pair(stride), padding=_pair(pad),
                             dilation=_pair(dilate), groups=num_group))


def FullyConnected(data, weight, bias=None, num_hidden=None, no_bias=False, flatten=True, name=None, **_ignored):
    t = data._t
    if flatten and t.dim() > 2:
        t = t.reshape(t.shape[0], -1)
    b = None if (no_bias or bias is None) else bias._t
    return NDArray(TF.linear(t, weight._t, b))


def Activation(data, act_type="relu", name=None):
    t = data._t
    if act_type == "relu":
        return NDArray(torch.relu(t))
    if act_type == "sigmoid":
        return NDArray(torch.sigmoid(t))
    if act_type == "tanh":
        return NDArray(torch.tanh(t))
    if act_type == "softrelu":
        return NDArray(TF.softplus(t))
    raise ValueError("unknown act_type %s" % act_type)


def BatchNorm(data, gamma, beta, running_mean, running_var, eps=1e-5, momentum=0.9, fix_gamma=False,
              use_global_stats=False, axis=1, name=None, **_ignored):
    """Inference: running statistics.  Under `autograd.train_mode` (and not use_global_stats): batch statistics,
    differentiable, and MXNet's update of the moving statistics (src/operator/nn/batch_norm.cc):
    moving = moving * momentum + batch * (1 - momentum) with the BIASED batch variance."""
    from . import autograd
    g = torch.ones_like(gamma._t) if fix_gamma else gamma._t
    if autograd.is_training() and not use_global_stats:
        x = data._t
        dims = [d for d in range(x.dim()) if d != axis]
        shape = [1] * x.dim()
        shape[axis] = -1
        mean = x.mean(dim=dims)
        var = ((x - mean.reshape(shape)) ** 2).mean(dim=dims)
        y = (x - mean.reshape(shape)) / torch.sqrt(var.reshape(shape) + eps) * g.reshape(shape) + beta._t.reshape(shape)
        with torch.no_grad():
            running_mean._t.mul_(momentum).add_(mean.detach() * (1.0 - momentum))
            running_var._t.mul_(momentum).add_(var.detach() * (1.0 - momentum))
        return NDArray(y)
    return NDArray(TF.batch_norm(data._t, running_mean._t, running_var._t, g, beta._t, False, 0.0, eps))


def Pooling(data, kernel=(1, 1), pool_type="max", global_pool=False, stride=None, pad=(0, 0),
            pooling_convention="valid", count_include_pad=True, name=None, **_ignored):
    t = data._t
    if global_pool:
        if pool_type == "avg":
            return NDArray(t.mean(dim=(2, 3), keepdim=True))
        return NDArray(torch.amax(t, dim=(2, 3), keepdim=True))
    kernel = _pair(kernel)
    stride = _pair(stride) if stride is not None else kernel
    ceil = pooling_convention == "full"
    if pool_type == "max":
        return NDArray(TF.max_pool2d(t, kernel, stride, _pair(pad), ceil_mode=ceil))
    return NDArray(TF.avg_pool2d(t, kernel, stride, _pair(pad), ceil_mode=ceil,
                                 count_include_pad=count_include_pad))


def waitall():
    if torch.cuda.is_available():
        torch.cuda.synchronize()


# -- files -------------------------------------------------------------------------------------
def save(fname, data):
    """`mx.nd.save`: a list or a {name: NDArray} dict in MXNet's NDArray-list format (mx/ndarray_file.py)."""
    from . import ndarray_file
    if isinstance(data, NDArray):
        data = [data]
    ndarray_file.save(fname, data)


def load(fname):
    """`mx.nd.load`: the list or {name: NDArray} dict of an NDArray-list file (on the CPU, as MXNet loads them)."""
    from . import ndarray_file
    out = ndarray_file.load(fname)
    wrap = lambda a: None if a is None else NDArray(torch.from_numpy(a))
    return {k: wrap(v) for k, v in out.items()} if isinstance(out, dict) else [wrap(a) for a in out]
