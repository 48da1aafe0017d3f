"""Initializers with MXNet's names.  Random ones draw from numpy's global RNG so that `np.random.seed(7)`
(the CLI's `--fixed-random-seed`, examples/simulate_quantization.py:99-100,182) makes weights reproducible.
`Constant` is what `qparams_init` uses for `input_max` (quantize/initialize/initialize.py:72-75).
"""
import numpy as np

__all__ = ["Initializer", "Constant", "Zero", "One", "Uniform", "Normal", "Xavier", "MSRAPrelu", "create"]


class Initializer(object):
    def __call__(self, shape):
        raise NotImplementedError


class Constant(Initializer):
    def __init__(self, value):
        self.value = value

    def __call__(self, shape):
        v = self.value
        if hasattr(v, "asnumpy"):
            v = v.asnumpy()
        return np.broadcast_to(np.asarray(v, dtype=np.float32), shape).copy()


class Zero(Constant):
    def __init__(self):
        super(Zero, self).__init__(0.0)


class One(Constant):
    def __init__(self):
        super(One, self).__init__(1.0)


class Uniform(Initializer):
    def __init__(self, scale=0.07):
        self.scale = scale

    def __call__(self, shape):
        return np.random.uniform(-self.scale, self.scale, size=shape).astype(np.float32)


class Normal(Initializer):
    def __init__(self, sigma=0.01):
        self.sigma = sigma

    def __call__(self, shape):
        return np.random.normal(0.0, self.sigma, size=shape).astype(np.float32)


def _fans(shape):
    hw = int(np.prod(shape[2:])) if len(shape) > 2 else 1
    fan_out = shape[0] * hw
    fan_in = (shape[1] if len(shape) > 1 else shape[0]) * hw
    return fan_in, fan_out


class Xavier(Initializer):
    def __init__(self, rnd_type="uniform", factor_type="avg", magnitude=3):
        self.rnd_type, self.factor_type, self.magnitude = rnd_type, factor_type, float(magnitude)

    def __call__(self, shape):
        fan_in, fan_out = _fans(shape)
        factor = {"avg": (fan_in + fan_out) / 2.0, "in": fan_in, "out": fan_out}[self.factor_type]
        scale = np.sqrt(self.magnitude / factor)
        if self.rnd_type == "uniform":
            return np.random.uniform(-scale, scale, size=shape).astype(np.float32)
        return np.random.normal(0, scale, size=shape).astype(np.float32)


class MSRAPrelu(Xavier):
    def __init__(self, factor_type="avg", slope=0.25):
        super(MSRAPrelu, self).__init__("gaussian", factor_type, 2.0 / (1 + slope ** 2))


_ALIASES = {"zeros": Zero, "zero": Zero, "ones": One, "one": One, "uniform": Uniform, "normal": Normal,
            "xavier": Xavier, "msraprelu": MSRAPrelu}


def create(init, default=None):
    if init is None:
        init = default
    if init is None:
        return Uniform()
    if isinstance(init, Initializer):
        return init
    if isinstance(init, str):
        return _ALIASES[init.lower()]()
    raise TypeError("cannot make an initializer from %r" % (init,))
