"""Gluon-shaped `Parameter` / `ParameterDict` over torch device memory.

The reference keeps every piece of quantisation state as a Gluon Parameter created through
`m.params.get("input_max", shape=(1,), init="zeros", allow_deferred_init=True, differentiable=False)`
(quantize/convert/convert_conv2d.py:116-119) and mutates it with `set_data` / reads it with `data()`
(quantize/convert/convert.py:70; convert_conv2d.py:101-105).  Only that surface is reproduced.
"""
import re
from collections import OrderedDict

import numpy as np
import torch

from .. import initializer as _init
from ..context import Context, cpu
from ..ndarray import NDArray

__all__ = ["Parameter", "ParameterDict", "DeferredInitializationError"]


class DeferredInitializationError(RuntimeError):
    pass


class Parameter(object):
    def __init__(self, name, grad_req="write", shape=None, dtype="float32", init=None,
                 allow_deferred_init=False, differentiable=True, **_ignored):
        self.name = name
        self.shape = None if shape is None else tuple(int(s) for s in ((shape,) if isinstance(shape, int) else shape))
        self.dtype = dtype
        self.init = init
        self.allow_deferred_init = allow_deferred_init
        self.differentiable = differentiable
        self.grad_req = grad_req if differentiable else "null"
        self._data = None            # NDArray
        self._deferred = None        # (init, ctx)
        self._ctx = None

    def __repr__(self):
        return "Parameter %s (shape=%s, dtype=%s)" % (self.name, self.shape, self.dtype)

    def _shape_known(self):
        return self.shape is not None and all(s > 0 for s in self.shape)

    def initialize(self, init=None, ctx=None, default_init=None, force_reinit=False):
        if self._data is not None and not force_reinit:
            return
        if isinstance(ctx, (list, tuple)):
            ctx = ctx[0]
        ctx = ctx or cpu()
        # Gluon rule: an explicit `init` argument wins; otherwise the Parameter's own init; otherwise the
        # dict-wide default (`ParameterDict.initialize(init)` passes its init as `default_init`).
        if init is None:
            init = default_init if self.init is None else self.init
        chosen = _init.create(init, None)
        if not self._shape_known():
            if not self.allow_deferred_init:
                raise ValueError("Cannot initialize Parameter %s: unknown shape %s" % (self.name, self.shape))
            self._deferred = (chosen, ctx)
            return
        self._finish_init(chosen, ctx)

    def _finish_init(self, init, ctx):
        arr = init(self.shape)
        t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32)).to(ctx.torch_device)
        self._data = NDArray(t)
        self._ctx = ctx
        self._deferred = None

    def _finish_deferred_init(self, shape):
        self.shape = tuple(shape)
        if self._deferred is None:
            raise DeferredInitializationError("Parameter %s was never initialized" % self.name)
        init, ctx = self._deferred
        self._finish_init(init, ctx)

    def _check(self):
        if self._data is None:
            if self._deferred is not None:
                raise DeferredInitializationError(
                    "Parameter %s has deferred initialization pending (shape %s)" % (self.name, self.shape))
            raise RuntimeError("Parameter %s has not been initialized" % self.name)

    def data(self, ctx=None):
        self._check()
        if ctx is not None and isinstance(ctx, Context) and ctx != self._ctx:
            raise RuntimeError("Parameter %s was not initialized on context %s (it lives on %s)"
                               % (self.name, ctx, self._ctx))
        return self._data

    def list_data(self):
        return [self.data()]

    def list_ctx(self):
        if self._data is None and self._deferred is not None:
            return [self._deferred[1]]
        self._check()
        return [self._ctx]

    def set_data(self, data):
        _WRITE_EPOCH[0] += 1
        if not isinstance(data, NDArray):
            data = NDArray(torch.as_tensor(np.asarray(data, dtype=np.float32)))
        if self._data is None:
            ctx = self._deferred[1] if self._deferred is not None else data.context
            self.shape = data.shape
            self._data = NDArray(data._t.to(ctx.torch_device, copy=True))
            self._ctx = ctx
            self._deferred = None
            return
        assert tuple(data.shape) == tuple(self._data.shape), \
            "set_data shape mismatch for %s: %s vs %s" % (self.name, data.shape, self._data.shape)
        with torch.no_grad():
            self._data._t.copy_(data._t)

    def _load_init(self, arr, ctx=None):
        """A value read from a parameter file: initialises a deferred Parameter with it, or overwrites the data in place."""
        arr = np.ascontiguousarray(arr, dtype=np.float32)
        _WRITE_EPOCH[0] += 1
        if self._data is None:
            if self._deferred is None:
                self._deferred = (_init.Zero(), ctx or cpu())
            # the declared shape holds wherever it is known (MXNet's Parameter._load_init: 0 = a dimension still to be
            # inferred); a file of another width or class count must not load silently
            declared = tuple(self.shape) if self.shape is not None else None
            if declared is not None and (len(declared) != arr.ndim or
                                         any(d not in (0, a) for d, a in zip(declared, arr.shape))):
                raise AssertionError("Failed loading Parameter '%s' from saved params: shape incompatible expected %s vs "
                                     "saved %s" % (self.name, declared, tuple(arr.shape)))
            self.shape = tuple(arr.shape)
            self._finish_init(_init.Constant(arr), self._deferred[1] if ctx is None else ctx)
        else:
            assert tuple(arr.shape) == tuple(self._data.shape), \
                "Failed loading Parameter '%s' from saved params: shape incompatible expected %s vs saved %s" \
                % (self.name, tuple(self._data.shape), tuple(arr.shape))
            self.set_data(NDArray(torch.from_numpy(arr)))

    def reset_ctx(self, ctx):
        _WRITE_EPOCH[0] += 1
        if isinstance(ctx, (list, tuple)):
            ctx = ctx[0]
        if self._data is not None:
            self._data = NDArray(self._data._t.to(ctx.torch_device))
            self._ctx = ctx
        elif self._deferred is not None:
            self._deferred = (self._deferred[0], ctx)
        else:
            raise ValueError("Cannot reset context for Parameter %s: not initialized" % self.name)

    # ---- gradients (mx.autograd over torch's tape) ---------------------------------------------------------------------
    def _mark_trainable(self):
        """Called by Block.__call__ while recording: the leaf gets a .grad on backward."""
        if self._data is not None and self.grad_req != "null" and not self._data._t.requires_grad:
            self._data._t.requires_grad_(True)

    def grad(self, ctx=None):
        self._check()
        if self.grad_req == "null":
            raise RuntimeError("Cannot get gradient array for Parameter '%s' because grad_req='null'" % self.name)
        g = self._data._t.grad
        if g is None:
            g = torch.zeros_like(self._data._t)
            self._data._t.grad = g
        return NDArray(g)

    def list_grad(self):
        return [self.grad()]

    def zero_grad(self):
        if self._data is not None and self._data._t.grad is not None:
            self._data._t.grad.zero_()

    def cast(self, dtype):
        self.dtype = dtype


# Counts the writes to Parameters through the Gluon surface (set_data, loading a file, moving to another context).  Host-side
# caches of values DERIVED from parameters (folded BatchNorm constants, weight codes) are rebuilt in new tensors after such a
# write; whatever recorded the old tensors' addresses - the CLI's evaluation graphs - compares this number and starts over.
_WRITE_EPOCH = [0]


def write_epoch():
    return _WRITE_EPOCH[0]


def read_parameter_file(filename):
    """{name: numpy array} of a parameter file: an npz archive (what this package writes for `*.npz` names) or MXNet's
    NDArray-list format (`.params`: mx/ndarray_file.py), `arg:` / `aux:` prefixes stripped."""
    from .. import ndarray_file
    if ndarray_file.is_ndarray_file(filename):
        return ndarray_file.load_params(filename)
    with np.load(filename) as z:
        return {k: z[k] for k in z.files}


class ParameterDict(object):
    def __init__(self, prefix="", shared=None):
        self._prefix = prefix
        self._params = OrderedDict()
        self._shared = shared

    @property
    def prefix(self):
        return self._prefix

    def __repr__(self):
        return "%s(\n%s\n)" % (self._prefix, "\n".join("  " + repr(p) for p in self._params.values()))

    def __getitem__(self, key):
        return self._params[key]

    def __iter__(self):
        return iter(self._params)

    def __len__(self):
        return len(self._params)

    def __contains__(self, key):
        return key in self._params

    def items(self):
        return self._params.items()

    def keys(self):
        return self._params.keys()

    def values(self):
        return self._params.values()

    def _get_impl(self, name):
        if name in self._params:
            return self._params[name]
        if self._shared is not None and name in self._shared._params:
            self._params[name] = self._shared._params[name]
            return self._params[name]
        return None

    def get(self, name, **kwargs):
        name = self._prefix + name
        param = self._get_impl(name)
        if param is None:
            param = Parameter(name, **kwargs)
            self._params[name] = param
        else:
            for k, v in kwargs.items():
                if k == "shape" and v is not None and param.shape is not None:
                    v = tuple((v,) if isinstance(v, int) else v)
                    if len(v) == len(param.shape) and all(a == b or a == 0 or b == 0 for a, b in zip(v, param.shape)):
                        param.shape = tuple(a if a != 0 else b for a, b in zip(v, param.shape))
                        continue
                    raise AssertionError("Parameter %s shape mismatch: %s vs %s" % (name, v, param.shape))
        return param

    def update(self, other):
        for k, v in other.items():
            if k in self._params:
                assert self._params[k] is v, "Cannot update self with other because they have different " \
                                             "Parameters with the same name %s" % k
            else:
                self._params[k] = v

    def initialize(self, init=None, ctx=None, verbose=False, force_reinit=False):
        for p in self.values():
            p.initialize(None, ctx, init, force_reinit=force_reinit)

    def reset_ctx(self, ctx):
        for p in self.values():
            p.reset_ctx(ctx)

    def zero_grad(self):
        pass

    def setattr(self, name, value):
        for p in self.values():
            setattr(p, name, value)

    # -- flat {name: array} file: carries thresholds/scales across runs (checkpoint row) --
    def save(self, filename, strip_prefix=""):
        out = {}
        for name, p in self.items():
            key = name[len(strip_prefix):] if strip_prefix and name.startswith(strip_prefix) else name
            out[key] = p.data().asnumpy()
        with open(filename, "wb") as f:
            np.savez(f, **out)

    def load(self, filename, ctx=None, allow_missing=False, ignore_extra=False, restore_prefix=""):
        """Full-name files: the npz this package writes, or an MXNet NDArray-list file (`collect_params().save`, the legacy
        `save_params`, exported `arg:` / `aux:` checkpoints) - told apart by their first bytes."""
        loaded = {restore_prefix + k: v for k, v in read_parameter_file(filename).items()}
        if not allow_missing:
            for name in self.keys():
                assert name in loaded, "Parameter %s is missing in file %s" % (name, filename)
        for name, arr in loaded.items():
            if name not in self._params:
                assert ignore_extra, "Parameter %s loaded from file %s is not present in ParameterDict" % (name, filename)
                continue
            self._params[name]._load_init(arr, ctx)

    def select(self, pattern):
        """`collect_params(select)` filter: regex matched against the full name (Gluon uses re.match)."""
        rx = re.compile(pattern)
        ret = ParameterDict(self._prefix)
        for k, v in self.items():
            if rx.match(k):
                ret._params[k] = v
        return ret
