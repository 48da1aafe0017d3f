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


class LinearQuantizeSTE(autograd.Function):
    def __init__(self, scale, clip_max=None, clip_min=None):
        super(LinearQuantizeSTE, self).__init__()
        self.clip_max = clip_max
        self.clip_min = clip_min if clip_min is not None else 0.
        self.scale = scale

    def forward(self, x):
        t = x._t if isinstance(x, NDArray) else x
        scale = self.scale
        if isinstance(scale, NDArray):
            st = scale._t.reshape(-1).contiguous()
        elif isinstance(scale, torch.Tensor):
            st = scale.reshape(-1).contiguous()
        elif isinstance(scale, np.ndarray):
            st = torch.from_numpy(np.ascontiguousarray(scale, dtype=np.float32).reshape(-1)).to(t.device)
        elif isinstance(scale, (numbers.Number, np.generic)) or hasattr(scale, "asscalar"):
            st = torch.tensor([float(np.float32(float(scale)))], dtype=torch.float32, device=t.device)
        else:
            raise TypeError("unsupported scale type %r" % type(scale))
        clip_max = None if self.clip_max is None else float(self.clip_max)
        y = ops.ste_forward(t.contiguous(), st, clip_max, float(self.clip_min))
        return NDArray(y)

    def backward(self, dy):
        return dy
