"""Device contexts with MXNet's spelling (`mx.cpu()`, `mx.gpu(i)`).

The reference selects its device as `gpu(opt.use_gpu) if opt.use_gpu != -1 else cpu()`
(examples/simulate_quantization.py:254).  Here `gpu(i)` is HIP device i of the MI355X node
(PyTorch-ROCm exposes it as `cuda:i`); device memory and streams are torch's, nothing else.
"""
import torch

__all__ = ["Context", "cpu", "gpu", "current_context", "num_gpus"]


class Context(object):
    __slots__ = ("device_type", "device_id")

    def __init__(self, device_type, device_id=0):
        assert device_type in ("cpu", "gpu")
        self.device_type = device_type
        self.device_id = int(device_id)

    @property
    def torch_device(self):
        if self.device_type == "cpu":
            return torch.device("cpu")
        return torch.device("cuda", self.device_id)

    @staticmethod
    def from_torch(dev):
        dev = torch.device(dev)
        if dev.type == "cpu":
            return Context("cpu", 0)
        return Context("gpu", dev.index or 0)

    def __eq__(self, other):
        return isinstance(other, Context) and self.device_type == other.device_type \
            and self.device_id == other.device_id

    def __hash__(self):
        return hash((self.device_type, self.device_id))

    def __repr__(self):
        return "%s(%d)" % (self.device_type, self.device_id)


def cpu(device_id=0):
    return Context("cpu", device_id)


def gpu(device_id=0):
    return Context("gpu", device_id)


def current_context():
    return cpu()


def num_gpus():
    return torch.cuda.device_count()
