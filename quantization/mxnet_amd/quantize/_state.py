"""Device-resident scalars.  The reference pulls every per-layer statistic to the host with `.asscalar()`
(convert_conv2d.py:56,58 — one device->host sync per quantised layer per batch); here they stay on the GPU and are
only synchronised when somebody actually looks at the value."""
import numpy as np

__all__ = ["DeviceScalar"]


class DeviceScalar(object):
    """A (1,) fp32 device tensor that reads like the numpy scalar the reference stores in `current_input_max`."""
    __slots__ = ("t",)
    __array_priority__ = 2000.0

    def __init__(self, t):
        self.t = t

    def asscalar(self):
        return np.float32(self.t.detach().cpu().numpy().reshape(-1)[0])

    item = asscalar

    def __float__(self):
        return float(self.asscalar())

    def __repr__(self):
        return repr(self.asscalar())

    def __format__(self, spec):
        return format(float(self), spec)

    def __array__(self, dtype=None, copy=None):
        return np.asarray(self.asscalar(), dtype=dtype)

    def _bin(self, other, op):
        o = other.asscalar() if isinstance(other, DeviceScalar) else other
        return op(self.asscalar(), o)

    def __add__(self, o):
        return self._bin(o, lambda a, b: a + b)

    __radd__ = __add__

    def __mul__(self, o):
        return self._bin(o, lambda a, b: a * b)

    __rmul__ = __mul__

    def __sub__(self, o):
        return self._bin(o, lambda a, b: a - b)

    def __rsub__(self, o):
        return self._bin(o, lambda a, b: b - a)

    def __truediv__(self, o):
        return self._bin(o, lambda a, b: a / b)

    def __eq__(self, o):
        return self._bin(o, lambda a, b: a == b)

    def __lt__(self, o):
        return self._bin(o, lambda a, b: a < b)

    def __gt__(self, o):
        return self._bin(o, lambda a, b: a > b)

    def __hash__(self):
        return id(self)
