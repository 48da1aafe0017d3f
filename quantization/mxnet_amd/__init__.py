"""quantization.mxnet_amd — MI355X-native simulated-quantisation hot path of hey-yahei/Quantization.MXNet.

Layout:
  csrc/      hand-written HIP kernels for gfx950 + the C-ABI (`libfakequant.so`, declared in include/fakequant.h)
  _lib.py    ctypes loader / `check_call`            ops.py   python entry points over raw device pointers
  mx/        MXNet/Gluon-shaped host facade over torch-ROCm tensors (device memory + streams only)
  quantize/  the reference's `quantize` package surface (convert / initialize / distribution_calibrate / utils)
  nn/        the reference's `nn.Conv2D` (int-code conv)
  dist.py    one-process-per-GPU sharding + the RCCL all-reduces of calibration statistics
"""
import os as _os

# Several evaluation batches in flight (quantize/fuse.py's per-stream state; bench.py --streams, the CLI's --eval-streams) need
# their HIP streams on DIFFERENT hardware queues: the runtime maps streams onto GPU_MAX_HW_QUEUES queues (default 4) and two
# lanes that land on one queue run one after the other - the CLI's three lanes did, next to the streams its captures and the
# tensor library keep (MobileNetV2 evaluation 125 k -> 152 k images/s with 8 queues, profiles/r5_cli_lanes.txt).  The variable
# is read when the HIP runtime initialises, i.e. at the first device call: setting it here, at import, is early enough unless
# the process touched the GPU before importing this package; a value the user exported wins.
_user_set_queues = "GPU_MAX_HW_QUEUES" in _os.environ
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def _hw_queues_took_effect():
    """False when the HIP runtime was already initialised at import (the variable is then ignored without a word and the lanes
    share four hardware queues).  Checked by whoever opens lanes (bench.py, the CLI): they say so once."""
    return _user_set_queues or not _hip_was_up


try:                                             # (torch may not be imported yet: then nothing has touched the GPU through it)
    import sys as _sys
    _t = _sys.modules.get("torch")
    _hip_was_up = bool(_t is not None and _t.cuda.is_initialized())
except Exception:                                # pragma: no cover
    _hip_was_up = False
if _hip_was_up and not _user_set_queues:
    import warnings as _warnings
    _warnings.warn("quantization.mxnet_amd imported after the HIP runtime was initialised: GPU_MAX_HW_QUEUES=%s will not apply "
                   "to this process; several evaluation batches in flight (--streams / --eval-streams) may share hardware queues"
                   % _os.environ.get("GPU_MAX_HW_QUEUES"), RuntimeWarning, stacklevel=2)

__version__ = "0.1.0"
