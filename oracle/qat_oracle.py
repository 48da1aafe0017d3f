"""CPU restatement of ONE quantisation-aware-training step of the reference (TEST INFRASTRUCTURE ONLY — nothing under
quantization/ may import this).

Reference: examples/quantize_aware_training_cifar10.ipynb cell 15 (`with autograd.record(): outputs = net(X); loss =
loss_func(outputs, y)`; `net.update_ema()`; `loss.backward()`; `trainer.step(batch, ignore_stale_grad=True)`), the patched
forwards quantize/convert/convert_conv2d.py:44-110 / convert_dense.py:37-70, the straight-through estimator
quantize/convert/ste_func.py:30-44 (backward = identity, the scale is not an input), `_update_ema` convert.py:66-78,
gluon BatchNorm in train mode (batch statistics, biased variance, moving = moving*momentum + batch*(1-momentum)),
gluon.loss.SoftmaxCrossEntropyLoss and MXNet's Adam (optimizer.py: lr_t = lr*sqrt(1-b2^t)/(1-b1^t);
w -= lr_t*m/(sqrt(v)+eps); gradients rescaled by 1/batch_size).

It is deliberately independent of the facade: a functional torch-CPU graph (torch is only the differentiator of the
float convolution / dense / BN / loss), every fake-quant is the numpy oracle (`fq_oracle`) behind an identity-backward
node.  Parity pinned by: the quantisers' own golden vectors (tests/test_oracle_golden.py) AND, since round 3, the composed
step itself: tests/golden/g11_qat.npz holds three configurations x four steps made by the REFERENCE'S OWN `convert_model` /
`LinearQuantizeSTE` / fake-BN pre-hook / `update_ema` under `autograd.record()` (tools/gen_golden.py::gen_qat; the stand-in
supplies convolution / BatchNorm / loss and the tape), and tests/test_qat_golden.py holds this restatement to it: loss,
logits, moving statistics 2e-6, gradients 2e-5, `input_max` bit for bit.  Not covered by a reference-made fixture: the
optimiser (MXNet's Adam rule is written out below from optimizer.py; the fixture has no optimiser step).

The net is described by a list of layer dicts (see tests/test_qat.py):
  {"op": "conv", "w": name, "stride": s, "pad": p, "groups": g, "quant": True/False}
  {"op": "bn", "gamma": .., "beta": .., "mean": .., "var": .., "momentum": 0.9, "eps": 1e-5}
  {"op": "relu"} {"op": "gap"} {"op": "flatten"} {"op": "dense", "w": .., "b": .., "quant": True}
"""
import math

import numpy as np
import torch
import torch.nn.functional as TF

from . import fq_oracle as O

F32 = np.float32


class _STE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, fn):
        return torch.from_numpy(np.ascontiguousarray(fn(x.detach().numpy()), dtype=F32))

    @staticmethod
    def backward(ctx, g):
        return g, None


def _quant_input(x, state, name, signed, width, offline, dense=False, enabled=True):
    """convert_conv2d.py:55-66 / convert_dense.py:40-49; records current_input_max for update_ema.  The statistic is
    taken whenever the converter was built with quantize_input (:55-56); the quantiser itself only runs when
    `net.quantize_input(enable=True, ...)` (:57)."""
    a = x.detach().numpy()
    cur = O.batch_mean(O.absmax_per_sample(a))
    state["current_input_max"][name] = F32(cur)
    if not enabled:
        return x
    thr = state["input_max"][name] if offline else cur
    if dense:
        fn = lambda v: O.dense_input_fake_quant(v, signed, width, offline_threshold=thr)[0]
    else:
        fn = lambda v: O.conv_input_fake_quant(v, signed, width, offline_threshold=thr)[0]
    return _STE.apply(x, fn)


def _quant_weight(w, quant_type, width, num_group=1):
    def fn(v):
        out = O.weight_fake_quant(v, quant_type, width, num_group)
        return out[0] if isinstance(out, tuple) else out
    return _STE.apply(w, fn)


def forward(layers, params, X, state, train=True, signed=False, in_width=8, wt_width=8, quant_type="layer",
            offline=False, input_quant=True):
    """params: name -> torch tensor (leaves with requires_grad for the trainable ones).  Returns logits."""
    x = X
    qi = 0
    for L in layers:
        op = L["op"]
        if op == "fbconv":
            # fake-BN convolution (convert_conv2d.py:47-51 fold, :144-154 batch-statistic pre-hook)
            w, b = params[L["w"]], params[L["b"]]
            g, be, rm, rv = params[L["gamma"]], params[L["beta"]], params[L["mean"]], params[L["var"]]
            kw = dict(stride=L.get("stride", 1), padding=L.get("pad", 0), groups=L.get("groups", 1))
            with torch.no_grad():
                yh = TF.conv2d(x, w, b, **kw)
                ns = yh.shape[0] * yh.shape[2] * yh.shape[3]
                cm = yh.sum(dim=(0, 2, 3)) / ns
                cv = ((yh - cm.reshape(1, -1, 1, 1)) ** 2).sum(dim=(0, 2, 3)) / ns
                state.setdefault("current_mean", {})[L["mean"]] = cm
                state.setdefault("current_var", {})[L["var"]] = cv
            cout = w.shape[0]
            wf = (w.reshape(cout, -1) * g.reshape(-1, 1) / torch.sqrt(rv + 1e-10).reshape(-1, 1)).reshape(w.shape)
            bf = g * (b - rm) / torch.sqrt(rv + 1e-10) + be
            name = "q%d" % qi
            qi += 1
            x = _quant_input(x, state, name, signed, in_width, offline, enabled=input_quant)
            wq = _quant_weight(wf, quant_type, wt_width, L.get("groups", 1))
            x = TF.conv2d(x, wq, bf, **kw)
        elif op == "conv":
            w = params[L["w"]]
            if L.get("quant", True):
                name = "q%d" % qi
                qi += 1
                x = _quant_input(x, state, name, signed, in_width, offline, enabled=input_quant)
                w = _quant_weight(w, quant_type, wt_width, L.get("groups", 1))
            b = params[L["b"]] if L.get("b") else None
            x = TF.conv2d(x, w, b, stride=L.get("stride", 1), padding=L.get("pad", 0), groups=L.get("groups", 1))
        elif op == "dense":
            w = params[L["w"]]
            if L.get("quant", True):
                name = "q%d" % qi
                qi += 1
                x = _quant_input(x, state, name, signed, in_width, offline, dense=True, enabled=input_quant)
                w = _quant_weight(w, "channel" if quant_type in ("channel", "group") else "layer", wt_width)
            x = TF.linear(x, w, params[L["b"]] if L.get("b") else None)
        elif op == "bn":
            g, b = params[L["gamma"]], params[L["beta"]]
            eps, mom = L.get("eps", 1e-5), L.get("momentum", 0.9)
            if train:
                mean = x.mean(dim=(0, 2, 3))
                var = ((x - mean.reshape(1, -1, 1, 1)) ** 2).mean(dim=(0, 2, 3))
                with torch.no_grad():
                    params[L["mean"]].mul_(mom).add_(mean.detach() * (1.0 - mom))
                    params[L["var"]].mul_(mom).add_(var.detach() * (1.0 - mom))
            else:
                mean, var = params[L["mean"]], params[L["var"]]
            x = (x - mean.reshape(1, -1, 1, 1)) / torch.sqrt(var.reshape(1, -1, 1, 1) + eps) * g.reshape(1, -1, 1, 1) \
                + b.reshape(1, -1, 1, 1)
        elif op == "relu":
            x = torch.relu(x)
        elif op == "gap":
            x = x.mean(dim=(2, 3), keepdim=True)
        elif op == "flatten":
            x = x.reshape(x.shape[0], -1)
        else:
            raise ValueError(op)
    return x


def softmax_ce(logits, y):
    """gluon.loss.SoftmaxCrossEntropyLoss: one value per sample."""
    return -torch.gather(TF.log_softmax(logits, dim=-1), 1, y.long().reshape(-1, 1)).reshape(-1)


def update_ema(state, momentum=0.9, params=None):
    """convert.py:66-78: input_max = (1 - m) * current_input_max + m * input_max (fp32, as `oracle.ema_update`); for
    fake-BN convolutions the same for running_mean / running_var from the pre-hook's batch statistics."""
    for k, cur in state["current_input_max"].items():
        state["input_max"][k] = O.ema_update(state["input_max"][k], cur, momentum)
    if params is not None:
        with torch.no_grad():
            for key in ("current_mean", "current_var"):
                for name, cur in state.get(key, {}).items():
                    params[name].copy_((1 - momentum) * cur + momentum * params[name])


class Adam(object):
    """MXNet Adam with gluon.Trainer's 1/batch_size rescale."""

    def __init__(self, lr, beta1=0.9, beta2=0.999, eps=1e-8):
        self.lr, self.b1, self.b2, self.eps = lr, beta1, beta2, eps
        self.m, self.v, self.t = {}, {}, {}

    def step(self, params, trainable, batch_size):
        with torch.no_grad():
            for k in trainable:
                w = params[k]
                if w.grad is None:
                    continue                                   # ignore_stale_grad=True
                g = w.grad / float(batch_size)
                if k not in self.m:
                    self.m[k], self.v[k], self.t[k] = torch.zeros_like(w), torch.zeros_like(w), 0
                self.t[k] += 1
                t = self.t[k]
                self.m[k].mul_(self.b1).add_((1 - self.b1) * g)
                self.v[k].mul_(self.b2).add_((1 - self.b2) * g * g)
                lr_t = self.lr * math.sqrt(1 - self.b2 ** t) / (1 - self.b1 ** t)
                w.sub_(lr_t * self.m[k] / (self.v[k].sqrt() + self.eps))
                w.grad = None


def train_step(layers, params, trainable, X, y, state, opt, **fwd_kw):
    """One notebook iteration.  Returns (per-sample loss, logits, {name: grad}) and updates params / state in place."""
    for k in trainable:
        params[k].requires_grad_(True)
    logits = forward(layers, params, X, state, train=True, **fwd_kw)
    loss = softmax_ce(logits, y)
    update_ema(state, params=params)
    loss.backward(torch.ones_like(loss))
    grads = {k: (None if params[k].grad is None else params[k].grad.detach().clone()) for k in trainable}
    opt.step(params, trainable, X.shape[0])
    return loss.detach().numpy(), logits.detach().numpy(), grads
