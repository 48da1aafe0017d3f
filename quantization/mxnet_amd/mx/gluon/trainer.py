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
        op_capturable = op.pop("capturable", False)
        if op:
            raise ValueError("Trainer: unsupported optimizer_params %s" % sorted(op))
        self._state = {}
        self._t = {}
        # capturable: the step counter and Adam's bias-corrected rate live on the device and are advanced by device
        # operations inside `step`, so that a whole training step (forward + backward + step) can be captured into a
        # hipGraph and replayed (examples/qat_finetune.py --graph): a replay runs no Python, a host-side `t` would stay frozen
        # The learning rate is a device scalar too (`set_learning_rate` fills it in place): a replay reads it, so a schedule
        # keeps working after capture.  One step counter serves every parameter, which is only MXNet's per-parameter `t` when
        # every parameter takes part in every step - `step` checks that.
        self._capturable = bool(op_capturable)
        self._dev_t = None
        self._dev_lr = None

    @property
    def learning_rate(self):
        return self._lr

    def set_learning_rate(self, lr):
        self._lr = float(lr)
        if self._dev_lr is not None:
            self._dev_lr.fill_(self._lr)           # in place: captured steps hold this tensor's address

    def _lr_operand(self, like):
        """The rate `step` multiplies by: the Python float, or (capturable) a device scalar that replays re-read."""
        if not self._capturable:
            return self._lr
        if self._dev_lr is None:
            self._dev_lr = torch.full((), self._lr, dtype=torch.float64, device=like.device)
        return self._dev_lr

    def step(self, batch_size, ignore_stale_grad=False):
        """One update of every Parameter that received a gradient.  The arithmetic is the per-parameter rule of the module
        docstring, operation by operation and in that order; it is ISSUED through torch's multi-tensor (`_foreach_*`) kernels -
        ~15 launches per step instead of 13 per parameter (800 of the ~2000 launches of a CIFAR ResNet-20 QAT step)."""
        rescale = self._rescale_user / float(batch_size)
        live = []
        for p in self._params:
            if p._data is None:
                continue
            w = p._data._t
            if w.grad is None:
                if ignore_stale_grad:
                    continue
                raise UserWarning("Gradient of Parameter `%s` has not been updated by backward since the last "
                                  "`step`; pass ignore_stale_grad=True to skip it" % p.name)
            live.append((p, w))
        if not live:
            return
        if self._capturable and len(live) != sum(1 for p in self._params if p._data is not None):
            raise RuntimeError("Trainer(capturable=True): every parameter must receive a gradient on every step (one device "
                               "step counter serves them all); %d of %d did" %
                               (len(live), sum(1 for p in self._params if p._data is not None)))
        with torch.no_grad():
            ws = [w for _, w in live]
            gs = torch._foreach_mul([w.grad for w in ws], rescale)                   # g' = rescale * g
            if self._clip is not None:
                gs = [g.clamp_(-float(self._clip), float(self._clip)) for g in gs]
            if self._wd:
                torch._foreach_add_(gs, torch._foreach_mul(ws, self._wd))            # + wd * w
            lr = self._lr_operand(ws[0])
            if self._opt == "sgd":
                lr32 = lr.to(torch.float32) if self._capturable else lr                # fp32(lr), as the scalar overload rounds it
                if self._momentum:
                    moms = []
                    for p, w in live:
                        mom = self._state.get(id(p))
                        if mom is None:
                            mom = self._state[id(p)] = torch.zeros_like(w)
                        moms.append(mom)
                    torch._foreach_mul_(moms, self._momentum)
                    torch._foreach_sub_(moms, torch._foreach_mul(gs, lr32))          # mom = momentum * mom - lr * g'
                    torch._foreach_add_(ws, moms)
                else:
                    torch._foreach_sub_(ws, torch._foreach_mul(gs, lr32))
            else:
                ms, vs, by_t = [], [], {}
                for i, (p, w) in enumerate(live):
                    st = self._state.get(id(p))
                    if st is None:
                        st = self._state[id(p)] = (torch.zeros_like(w), torch.zeros_like(w))
                    t = self._t[id(p)] = self._t.get(id(p), 0) + 1
                    ms.append(st[0])
                    vs.append(st[1])
                    by_t.setdefault(t, []).append(i)
                torch._foreach_mul_(ms, self._beta1)
                torch._foreach_add_(ms, torch._foreach_mul(gs, 1.0 - self._beta1))   # m = b1 m + (1 - b1) g'
                torch._foreach_mul_(vs, self._beta2)
                torch._foreach_add_(vs, torch._foreach_mul(torch._foreach_mul(gs, 1.0 - self._beta2), gs))   # ((1 - b2) g') g'
                den = torch._foreach_sqrt(vs)
                torch._foreach_add_(den, self._eps)
                if self._capturable:
                    # lr_t from a step counter on the device (fp64, the host formula's precision): every parameter of a
                    # captured step takes part in every replay, so one counter serves them all
                    if self._dev_t is None:
                        self._dev_t = torch.zeros((), dtype=torch.float64, device=ws[0].device)
                    self._dev_t += 1.0
                    b1 = torch.full_like(self._dev_t, self._beta1)
                    b2 = torch.full_like(self._dev_t, self._beta2)
                    lr_t = (lr * torch.sqrt(1.0 - torch.pow(b2, self._dev_t)) /
                            (1.0 - torch.pow(b1, self._dev_t))).to(torch.float32)
                    num = torch._foreach_mul(ms, lr_t)
                    torch._foreach_div_(num, den)
                    torch._foreach_sub_(ws, num)
                    by_t = {}
                for t, idx in by_t.items():                          # (one group unless some parameters skipped steps)
                    lr_t = self._lr * math.sqrt(1.0 - self._beta2 ** t) / (1.0 - self._beta1 ** t)
                    num = torch._foreach_mul([ms[i] for i in idx], lr_t)
                    torch._foreach_div_(num, [den[i] for i in idx])                  # (lr_t m) / (sqrt(v) + eps)
                    torch._foreach_sub_([ws[i] for i in idx], num)
            for w in ws:
                w.grad = None                      # 'write': the next backward starts a fresh buffer

    def allreduce_grads(self):
        from ... import dist as fqdist
        import torch.distributed as td
        if fqdist.world_size() > 1:
            for p in self._params:
                if p._data is not None and p._data._t.grad is not None:
                    td.all_reduce(p._data._t.grad, op=td.ReduceOp.SUM)
