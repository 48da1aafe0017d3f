"""`mx` — the MXNet-shaped host facade this project runs on where MXNet itself is absent.

`from quantization.mxnet_amd import mx; mx.nd / mx.gluon / mx.autograd / mx.cpu() / mx.gpu(i)`.
Device memory, streams and the non-hot-path ops are PyTorch-ROCm's; the fake-quant path is the HIP library.
"""
from .context import Context, cpu, gpu, current_context, num_gpus
from . import ndarray
from . import ndarray as nd
from . import initializer
from . import initializer as init
from . import autograd
from . import gluon
