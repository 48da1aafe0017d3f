"""`LinearQuantizeSTE(scale, clip_max=None, clip_min=None)` — reference: quantize/convert/ste_func.py:30-44.

forward:  (x.clip(clip_min, clip_max) / (scale + 1e-10)).round() * scale   (no clip when clip_max is None;
          clip_min defaults to 0.0), executed by ONE fused HIP kernel (`fq_ste_forward`) instead of four NDArray passes.
backward: identity (straight-through).
`scale` may be a python/numpy scalar or an NDArray shaped (1,), (num,1,1,1) or (num,1) — one scale per leading row.
"""
import numbers

import numpy as np
import torch

from ...mx import autograd
from ...mx.ndarray import NDArray
from ... import ops

__all__ = ['LinearQuantizeSTE']


def _scale_vector(scale, device):
    """one fp32 scale per leading row, as a flat device tensor"""
    if isinstance(scale, NDArray):
        return scale._t.reshape(-1).contiguous()
    if isinstance(scale, torch.Tensor):
        return scale.reshape(-1).contiguous()
    if isinstance(scale, np.ndarray):
        return torch.from_numpy(np.ascontiguousarray(scale, dtype=np.float32).reshape(-1)).to(device)
    if isinstance(scale, (numbers.Number, np.generic)) or hasattr(scale, "asscalar"):
        return torch.tensor([float(np.float32(float(scale)))], dtype=torch.float32, device=device)
    raise TypeError("unsupported scale type %r" % type(scale))


class LinearQuantizeSTE(autograd.Function):
    """Straight-through estimator around the fake-quantiser: the gradient of the output IS the gradient of the input."""

    def __init__(self, scale, clip_max=None, clip_min=None):
        super(LinearQuantizeSTE, self).__init__()
        self.scale, self.clip_max = scale, clip_max
        self.clip_min = 0. if clip_min is None else clip_min          # the reference's default lower bound (:34)

    def forward(self, x):
        t = x._t if isinstance(x, NDArray) else x
        hi = None if self.clip_max is None else float(self.clip_max)
        return NDArray(ops.ste_forward(t.contiguous(), _scale_vector(self.scale, t.device), hi, float(self.clip_min)))

    def backward(self, dy):
        return dy
