"""Gluon-shaped `Block` / `HybridBlock`: exactly the hooks the reference's rewriter relies on.

What `quantize.convert.convert_model` needs from a block (quantize/convert/convert.py:58-63,82-89;
convert_conv2d.py:116-119,169-170; distribution_calibrate.py:78-85,111-112):
  * `net.apply(fn)` (children first, then self);
  * per-instance replacement of `hybrid_forward` with `types.MethodType`, the original kept as `origin_forward`;
  * `m.params.get(name, ...)` + attribute assignment registering a Parameter that is then handed to
    `hybrid_forward` as a keyword argument of the same name (`input_max`);
  * `register_forward_hook(fn) -> handle.detach()`, `register_forward_pre_hook`;
  * `collect_params(select)`, `name`, `prefix`, `name_scope()` (Gluon's hierarchical auto-naming — `qparams_init`
    finds a conv's sibling BN by `name.replace("conv", "batchnorm")`, initialize.py:51-54).
Always imperative: `F` is the `nd` module (the reference calls `.asscalar()` inside `hybrid_forward`, so its patched
blocks were never hybridisable either — SURVEY.md 8b).
"""
import threading
from collections import OrderedDict

from .. import ndarray as nd
from .. import autograd as _autograd
from ..ndarray import NDArray
from .parameter import Parameter, ParameterDict, DeferredInitializationError

__all__ = ["Block", "HybridBlock", "HookHandle"]


class _BlockScope(object):
    _current = threading.local()

    def __init__(self, block):
        self._block = block
        self._counter = {}
        self._old_scope = None

    @staticmethod
    def create(prefix, params, hint):
        current = getattr(_BlockScope._current, "value", None)
        if current is None:
            if prefix is None:
                if not hasattr(_BlockScope, "_global_counter"):
                    _BlockScope._global_counter = {}
                count = _BlockScope._global_counter.get(hint, 0)
                _BlockScope._global_counter[hint] = count + 1
                prefix = "%s%d_" % (hint, count)
            if params is None:
                params = ParameterDict(prefix)
            else:
                params = ParameterDict(params.prefix, params)
            return prefix, params
        if prefix is None:
            count = current._counter.get(hint, 0)
            prefix = "%s%d_" % (hint, count)
            current._counter[hint] = count + 1
        if params is None:
            parent = current._block.params
            params = ParameterDict(parent.prefix + prefix, parent._shared)
        else:
            params = ParameterDict(params.prefix, params)
        return current._block.prefix + prefix, params

    def __enter__(self):
        self._old_scope = getattr(_BlockScope._current, "value", None)
        _BlockScope._current.value = self
        return self

    def __exit__(self, ptype, value, trace):
        _BlockScope._current.value = self._old_scope


def reset_naming():
    """Forget the global auto-naming counters (so a freshly built net is again `mobilenet0_...`)."""
    _BlockScope._global_counter = {}


class HookHandle(object):
    def __init__(self, hooks, hid):
        self._hooks, self._id = hooks, hid

    def detach(self):
        self._hooks.pop(self._id, None)


class Block(object):
    def __init__(self, prefix=None, params=None):
        self._empty_prefix = prefix == ""
        self._prefix, self._params = _BlockScope.create(prefix, params, self._alias())
        self._name = self._prefix[:-1] if self._prefix.endswith("_") else self._prefix
        self._scope = _BlockScope(self)
        self._children = OrderedDict()
        self._reg_params = {}
        self._forward_hooks = OrderedDict()
        self._forward_pre_hooks = OrderedDict()
        self._hook_id = 0

    def _alias(self):
        return self.__class__.__name__.lower()

    def __setattr__(self, name, value):
        if hasattr(self, name):
            existing = getattr(self, name)
            if isinstance(existing, (Parameter, Block)) and not isinstance(value, type(existing)):
                raise TypeError("Changing attribute type for %s from %s to %s is not allowed."
                                % (name, type(existing), type(value)))
        if isinstance(value, Block):
            self.register_child(value, name)
        elif isinstance(value, Parameter):
            assert name not in self._reg_params or self._reg_params[name] is value, \
                "Overriding Parameter attribute %s is not allowed." % name
            self._reg_params[name] = value
        super(Block, self).__setattr__(name, value)

    def __repr__(self):
        body = "\n".join("  (%s): %s" % (k, repr(v).replace("\n", "\n  ")) for k, v in self._children.items())
        return "%s(\n%s\n)" % (self.__class__.__name__, body) if body else "%s()" % self.__class__.__name__

    @property
    def prefix(self):
        return self._prefix

    @property
    def name(self):
        return self._name

    @property
    def params(self):
        return self._params

    def name_scope(self):
        return self._scope

    def register_child(self, block, name=None):
        if name is None:
            name = str(len(self._children))
        self._children[name] = block

    def collect_params(self, select=None):
        ret = ParameterDict(self._params.prefix)
        own = self.params if select is None else self.params.select(select)
        ret.update(own)
        for cld in self._children.values():
            ret.update(cld.collect_params(select=select))
        return ret

    def apply(self, fn):
        for cld in list(self._children.values()):
            cld.apply(fn)
        fn(self)
        return self

    def initialize(self, init=None, ctx=None, verbose=False, force_reinit=False):
        self.collect_params().initialize(init, ctx, verbose, force_reinit)

    def hybridize(self, active=True, **kwargs):
        pass        # always imperative (see module docstring)

    def cast(self, dtype):
        pass

    def register_forward_hook(self, hook):
        self._hook_id += 1
        self._forward_hooks[self._hook_id] = hook
        return HookHandle(self._forward_hooks, self._hook_id)

    def register_forward_pre_hook(self, hook):
        self._hook_id += 1
        self._forward_pre_hooks[self._hook_id] = hook
        return HookHandle(self._forward_pre_hooks, self._hook_id)

    def _collect_params_with_prefix(self, prefix=""):
        """{structural name: Parameter} - attribute paths joined by dots (`features.0.weight`, `output.bias`): the names
        MXNet's `save_parameters` writes and the gluoncv model zoo's `.params` checkpoints carry."""
        if prefix:
            prefix += "."
        ret = {prefix + key: val for key, val in self._reg_params.items()}
        for name, child in self._children.items():
            ret.update(child._collect_params_with_prefix(prefix + name))
        return ret

    def save_parameters(self, filename):
        """`*.npz`: an npz archive of the prefix-stripped full names (this package's own checkpoints: --save-qparams); any
        other name: MXNet's NDArray-list format with structural names, as `mxnet.gluon.Block.save_parameters` writes it."""
        if str(filename).endswith(".npz"):
            self.collect_params().save(filename, strip_prefix=self.prefix)
            return
        from .. import ndarray_file
        ndarray_file.save(filename, {k: p.data().asnumpy() for k, p in self._collect_params_with_prefix().items()})

    def load_parameters(self, filename, ctx=None, allow_missing=False, ignore_extra=False):
        """mxnet.gluon.Block.load_parameters: structural names (`features.0.weight`: files written by `save_parameters`, the
        model zoo's checkpoints) when any name holds a dot, else the legacy full-name form through `collect_params().load`."""
        from .parameter import read_parameter_file
        loaded = read_parameter_file(filename)
        params = self._collect_params_with_prefix()
        if not loaded and not params:
            return
        if not any("." in k for k in loaded):
            self.collect_params().load(filename, ctx, allow_missing, ignore_extra, restore_prefix=self.prefix)
            return
        if not allow_missing:
            for name in params:
                assert name in loaded, "Parameter '%s' is missing in file '%s', which contains parameters: %s. Set " \
                    "allow_missing=True to ignore missing parameters." % (name, filename, ", ".join(sorted(loaded)[:8]) + " ...")
        for name, arr in loaded.items():
            if name not in params:
                if not ignore_extra:
                    raise ValueError("Parameter '%s' loaded from file '%s' is not present in the Block, which contains "
                                     "parameters %s. Set ignore_extra=True to ignore." %
                                     (name, filename, ", ".join(sorted(params)[:8]) + " ..."))
                continue
            params[name]._load_init(arr, ctx)

    def __call__(self, *args):
        # torch's grad mode follows mx.autograd's recording flag: outside `autograd.record()` no graph is ever built
        with _autograd.grad_mode():
            for hook in list(self._forward_pre_hooks.values()):
                hook(self, args)
            out = self.forward(*args)
            for hook in list(self._forward_hooks.values()):
                hook(self, args, out)
        return out

    def forward(self, *args):
        raise NotImplementedError

    def summary(self, *inputs):
        print(self)


class HybridBlock(Block):
    def infer_shape(self, *args):
        """Layers with deferred-shape Parameters override `_infer_param_shapes(x)`."""
        self._infer_param_shapes(*args)

    def _infer_param_shapes(self, *args):
        raise DeferredInitializationError("%s cannot infer its parameter shapes" % self.name)

    def forward(self, x, *args):
        assert isinstance(x, NDArray), "HybridBlock input must be an NDArray, got %s" % type(x)
        try:
            params = {k: p.data() for k, p in self._reg_params.items()}
        except DeferredInitializationError:
            self._infer_param_shapes(x, *args)
            params = {k: p.data() for k, p in self._reg_params.items()}
        if _autograd.is_recording():
            for p in self._reg_params.values():
                p._mark_trainable()
        return self.hybrid_forward(nd, x, *args, **params)

    def hybrid_forward(self, F, x, *args, **kwargs):
        raise NotImplementedError

    def export(self, path, epoch=0):
        raise NotImplementedError("symbol export is the (out-of-scope) MKLDNN freeze path; see DESIGN.md")
