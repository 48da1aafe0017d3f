"""`gluon.Trainer(params, optimizer, optimizer_params)` with MXNet's update rules (mxnet/gluon/trainer.py,
mxnet/optimizer/optimizer.py): `step(batch_size)` rescales gradients by 1/batch_size, applies the optimizer to every
Parameter with grad_req != 'null', and zeroes the gradient buffers (grad_req='write' semantics: the next backward
starts from zero).  `ignore_stale_grad=True` (the QAT notebook) skips Parameters that received no gradient.

  sgd :  g' = rescale*g (+ clip) + wd*w ;  mom = momentum*mom - lr*g' ;  w += mom          (momentum = 0: w -= lr*g')
  adam:  g' = rescale*g (+ clip) + wd*w ;  m = b1*m + (1-b1)*g' ;  v = b2*v + (1-b2)*g'^2 ;
         lr_t = lr*sqrt(1-b2^t)/(1-b1^t) ;  w -= lr_t * m / (sqrt(v) + eps)
(MXNet's Adam keeps epsilon outside the bias correction, torch.optim.Adam does not: written out, not wrapped.)
"""
import math

import torch

from .parameter import ParameterDict

__all__ = ["Trainer"]


class Trainer(object):
    def __init__(self, params, optimizer, optimizer_params=None, kvstore="device", **_ignored):
        if isinstance(params, (dict, ParameterDict)):
            params = list(params.values())
        self._params = [p for p in params if p.grad_req != "null"]
        self._opt = str(optimizer).lower()
        if self._opt not in ("sgd", "adam"):
            raise ValueError("Trainer: optimizer %r is not provided (sgd, adam)" % optimizer)
        op = dict(optimizer_params or {})
        self._lr = float(op.pop("learning_rate", 0.01 if self._opt == "sgd" else 0.001))
        self._wd = float(op.pop("wd", 0.0))
        self._momentum = float(op.pop("momentum", 0.0))
        self._beta1 = float(op.pop("beta1", 0.9))
        self._beta2 = float(op.pop("beta2", 0.999))
        self._eps = float(op.pop("epsilon", 1e-8))
        self._clip = op.pop("clip_gradient", None)
        self._rescale_user = float(op.pop("rescale_grad", 1.0))
        if op:
            raise ValueError("Trainer: unsupported optimizer_params %s" % sorted(op))
        self._state = {}
        self._t = {}

    @property
    def learning_rate(self):
        return self._lr

    def set_learning_rate(self, lr):
        self._lr = float(lr)

    def step(self, batch_size, ignore_stale_grad=False):
        rescale = self._rescale_user / float(batch_size)
        with torch.no_grad():
            for p in self._params:
                if p._data is None:
                    continue
                w = p._data._t
                g = w.grad
                if g is None:
                    if ignore_stale_grad:
                        continue
                    raise UserWarning("Gradient of Parameter `%s` has not been updated by backward since the last "
                                      "`step`; pass ignore_stale_grad=True to skip it" % p.name)
                g = g * rescale
                if self._clip is not None:
                    g = g.clamp(-float(self._clip), float(self._clip))
                if self._wd:
                    g = g + self._wd * w
                key = id(p)
                if self._opt == "sgd":
                    if self._momentum:
                        mom = self._state.get(key)
                        if mom is None:
                            mom = self._state[key] = torch.zeros_like(w)
                        mom.mul_(self._momentum).sub_(self._lr * g)
                        w.add_(mom)
                    else:
                        w.sub_(self._lr * g)
                else:
                    st = self._state.get(key)
                    if st is None:
                        st = self._state[key] = (torch.zeros_like(w), torch.zeros_like(w))
                    t = self._t[key] = self._t.get(key, 0) + 1
                    m, v = st
                    m.mul_(self._beta1).add_((1.0 - self._beta1) * g)
                    v.mul_(self._beta2).add_((1.0 - self._beta2) * g * g)
                    lr_t = self._lr * math.sqrt(1.0 - self._beta2 ** t) / (1.0 - self._beta1 ** t)
                    w.sub_(lr_t * m / (v.sqrt() + self._eps))
                w.grad = None                      # 'write': the next backward starts a fresh buffer

    def allreduce_grads(self):
        from ... import dist as fqdist
        import torch.distributed as td
        if fqdist.world_size() > 1:
            for p in self._params:
                if p._data is not None and p._data._t.grad is not None:
                    td.all_reduce(p._data._t.grad, op=td.ReduceOp.SUM)
