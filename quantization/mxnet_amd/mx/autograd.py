"""`mx.autograd` surface used on the path: `autograd.Function` (base of `LinearQuantizeSTE`,
quantize/convert/ste_func.py:30) and the `record()/pause()` scopes.  This round covers the evaluation /
calibration path only (SURVEY.md 8f rank 2 lists the QAT backward as "next"), so `Function.__call__`
runs `forward` and keeps no tape; `backward` stays defined on subclasses (identity for the STE).
"""
import contextlib

__all__ = ["Function", "record", "pause", "is_training", "is_recording"]


class Function(object):
    def __init__(self):
        self._used = False

    def __call__(self, *inputs):
        return self.forward(*inputs)

    def forward(self, *inputs):
        raise NotImplementedError

    def backward(self, *output_grads):
        raise NotImplementedError

    def save_for_backward(self, *args):
        self.saved_tensors = args


@contextlib.contextmanager
def record(train_mode=True):
    yield


@contextlib.contextmanager
def pause(train_mode=False):
    yield


def is_training():
    return False


def is_recording():
    return False
