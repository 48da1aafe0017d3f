"""quantization.mxnet_amd — MI355X-native simulated-quantisation hot path of hey-yahei/Quantization.MXNet.

Layout:
  csrc/      hand-written HIP kernels for gfx950 + the C-ABI (`libfakequant.so`, declared in include/fakequant.h)
  _lib.py    ctypes loader / `check_call`            ops.py   python entry points over raw device pointers
  mx/        MXNet/Gluon-shaped host facade over torch-ROCm tensors (device memory + streams only)
  quantize/  the reference's `quantize` package surface (convert / initialize / distribution_calibrate / utils)
  nn/        the reference's `nn.Conv2D` (int-code conv)
  dist.py    one-process-per-GPU sharding + the RCCL all-reduces of calibration statistics
"""
__version__ = "0.1.0"
