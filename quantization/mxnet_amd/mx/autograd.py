"""`mx.autograd` on top of torch's tape (the quantisation-aware-training path).

The facade's NDArrays wrap torch tensors, so torch autograd IS the tape:

* `record(train_mode=True)` / `pause()` set the recording / training flags (mxnet/autograd.py `record`, `pause`,
  `is_recording`, `is_training`).  `gluon.Block.__call__` runs every forward under `torch.set_grad_enabled(is_recording())`
  and flags trainable Parameters `requires_grad` while recording; outside a record scope no graph is ever built.
* The HIP fake-quant kernels are invisible to torch, so each call site links its output back to its input with
  `ste_link(x, y)`: forward returns (an alias of) the kernel's output — bit-exact — and backward is the identity, which
  is the whole of the reference's `LinearQuantizeSTE.backward` (quantize/convert/ste_func.py:43-44: `return dy`, also for
  clipped values; the scale is an attribute, not an input, so nothing flows through it).
* `Function` (base of `LinearQuantizeSTE`, ste_func.py:30) runs `forward` untaped and calls the subclass's `backward`
  from a torch Function, as MXNet's custom `autograd.Function` does.
* `backward(heads, head_grads)`, `NDArray.backward()`, `NDArray.attach_grad()`, `NDArray.grad` as in MXNet.
"""
import contextlib
import threading

import torch

__all__ = ["Function", "record", "pause", "train_mode", "predict_mode", "is_training", "is_recording", "backward",
           "ste_link", "wino_link", "grad_mode"]

_state = threading.local()


def _get(name):
    return getattr(_state, name, False)


def is_recording():
    return _get("recording")


def is_training():
    return _get("training")


@contextlib.contextmanager
def _scope(recording, training):
    prev = (_get("recording"), _get("training"))
    if recording is not None:
        _state.recording = recording
    if training is not None:
        _state.training = training
    try:
        with torch.set_grad_enabled(is_recording()):
            yield
    finally:
        _state.recording, _state.training = prev


def record(train_mode=True):
    return _scope(True, train_mode)


def pause(train_mode=False):
    return _scope(False, train_mode)


def train_mode():
    return _scope(None, True)


def predict_mode():
    return _scope(None, False)


def grad_mode():
    """Context used by Block.__call__: torch grad mode follows the recording flag."""
    return torch.set_grad_enabled(is_recording())


class _StraightThrough(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y):
        return y.view_as(y)              # an alias of the kernel's output: the forward value stays bit-exact

    @staticmethod
    def backward(ctx, g):
        return g, None


def ste_link(x, y):
    """y = fake_quant(x) was computed outside the tape (HIP kernel): make it a function of x with identity gradient.
    x, y: torch tensors of the same shape.  No-op when nothing is being recorded or x does not need a gradient."""
    if is_recording() and torch.is_grad_enabled() and x.requires_grad:
        return _StraightThrough.apply(x, y)
    return y


class _WinogradSTE(torch.autograd.Function):
    """wq = GI . STE(G w G^T) . GTI  (convert_conv2d.py:71-83), computed outside the tape by one HIP kernel.  The chain's
    gradient: dL/dw = G^T (GI^T g GTI^T) G per (cout, cin) 3x3 filter (the STE in the middle is the identity)."""

    @staticmethod
    def forward(ctx, w, wq, G, GI, GTI):
        ctx.save_for_backward(G, GI, GTI)
        return wq.view_as(wq)

    @staticmethod
    def backward(ctx, g):
        G, GI, GTI = ctx.saved_tensors
        gu = torch.einsum("ia,ocij,bj->ocab", GI, g, GTI)            # GI^T g GTI^T  (GI: 3 x t, GTI: t x 3)
        gw = torch.einsum("ai,ocab,bj->ocij", G, gu, G)               # G^T gu G      (G: t x 3)
        return gw, None, None, None, None


def wino_link(w, wq, G, GI, GTI):
    if is_recording() and torch.is_grad_enabled() and w.requires_grad:
        dev = w.device
        as_t = lambda a: torch.as_tensor(a, dtype=torch.float32, device=dev)
        return _WinogradSTE.apply(w, wq, as_t(G), as_t(GI), as_t(GTI))
    return wq


class _CustomBridge(torch.autograd.Function):
    """Runs a `mx.autograd.Function` instance: forward untaped, backward through the instance's `backward`."""

    @staticmethod
    def forward(ctx, fn, n_in, *tensors):
        from .ndarray import NDArray
        ctx.fn = fn
        with torch.no_grad():
            outs = fn.forward(*[NDArray(t) for t in tensors[:n_in]])
        single = not isinstance(outs, (tuple, list))
        ctx.single = single
        outs = (outs,) if single else tuple(outs)
        res = tuple(o._t for o in outs)
        return res[0] if single else res

    @staticmethod
    def backward(ctx, *grads):
        from .ndarray import NDArray
        with torch.no_grad():
            gin = ctx.fn.backward(*[NDArray(g) for g in grads])
        gin = (gin,) if not isinstance(gin, (tuple, list)) else tuple(gin)
        return (None, None) + tuple(None if g is None else g._t for g in gin)


class Function(object):
    """mxnet.autograd.Function: subclass with forward(*NDArray) / backward(*NDArray)."""

    def __init__(self):
        self._used = False

    def __call__(self, *inputs):
        from .ndarray import NDArray
        if is_recording() and torch.is_grad_enabled() and any(isinstance(i, NDArray) and i._t.requires_grad for i in inputs):
            out = _CustomBridge.apply(self, len(inputs), *[i._t for i in inputs])
            if isinstance(out, tuple):
                return tuple(NDArray(o) for o in out)
            return NDArray(out)
        return self.forward(*inputs)

    def forward(self, *inputs):
        raise NotImplementedError

    def backward(self, *output_grads):
        raise NotImplementedError

    def save_for_backward(self, *args):
        self.saved_tensors = args


def backward(heads, head_grads=None, retain_graph=False, train_mode=True):
    """mxnet.autograd.backward: accumulate d(heads)/d(leaf) into the leaves' .grad ("write" semantics are the
    Trainer's business: it zeroes after each step, as gluon.Trainer does for grad_req='write')."""
    from .ndarray import NDArray
    if isinstance(heads, NDArray):
        heads = [heads]
    if head_grads is None:
        head_grads = [None] * len(heads)
    elif isinstance(head_grads, NDArray):
        head_grads = [head_grads]
    ts = [h._t for h in heads]
    gs = [torch.ones_like(t) if g is None else g._t for t, g in zip(ts, head_grads)]
    torch.autograd.backward(ts, gs, retain_graph=retain_graph)
