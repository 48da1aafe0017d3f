"""`gluon.loss` pieces the QAT notebook uses (mxnet/gluon/loss.py): SoftmaxCrossEntropyLoss (alias
SoftmaxCELoss) and L2Loss.  Plain torch ops on the wrapped tensors, so they are on the tape."""
import torch
import torch.nn.functional as TF

from ..ndarray import NDArray
from .block import HybridBlock

__all__ = ["Loss", "SoftmaxCrossEntropyLoss", "SoftmaxCELoss", "L2Loss"]


class Loss(HybridBlock):
    def __init__(self, weight, batch_axis, **kwargs):
        super(Loss, self).__init__(**kwargs)
        self._weight = weight
        self._batch_axis = batch_axis

    def forward(self, *args):                      # losses take (pred, label[, sample_weight]); no Parameters
        return self.hybrid_forward(None, *args)


def _reduce(loss, batch_axis):
    dims = [d for d in range(loss.dim()) if d != batch_axis]
    return loss.mean(dim=dims) if dims else loss


class SoftmaxCrossEntropyLoss(Loss):
    """-sum_k onehot(label)_k log softmax(pred)_k, mean over all axes but the batch axis; one value per sample."""

    def __init__(self, axis=-1, sparse_label=True, from_logits=False, weight=None, batch_axis=0, **kwargs):
        super(SoftmaxCrossEntropyLoss, self).__init__(weight, batch_axis, **kwargs)
        self._axis = axis
        self._sparse_label = sparse_label
        self._from_logits = from_logits

    def hybrid_forward(self, F, pred, label, sample_weight=None):
        p = pred._t
        logp = p if self._from_logits else TF.log_softmax(p, dim=self._axis)
        if self._sparse_label:
            idx = label._t.long().unsqueeze(self._axis)
            loss = -torch.gather(logp, self._axis, idx)
        else:
            loss = -(logp * label._t).sum(dim=self._axis, keepdim=True)
        if self._weight is not None:
            loss = loss * self._weight
        if sample_weight is not None:
            loss = loss * sample_weight._t
        return NDArray(_reduce(loss, self._batch_axis))


SoftmaxCELoss = SoftmaxCrossEntropyLoss


class L2Loss(Loss):
    def __init__(self, weight=1.0, batch_axis=0, **kwargs):
        super(L2Loss, self).__init__(weight, batch_axis, **kwargs)

    def hybrid_forward(self, F, pred, label, sample_weight=None):
        loss = (pred._t - label._t.reshape(pred._t.shape)) ** 2
        if sample_weight is not None:
            loss = loss * sample_weight._t
        return NDArray(_reduce(loss * (self._weight / 2.0), self._batch_axis))
