#!/usr/bin/env python3
"""Per-workgroup phase stamps of the first convolution (debug build -DFQ_PW_TRACE, see pw_trace.py)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "quantization", "mxnet_amd", "csrc", "build", "libfakequant_trace.so")


def main():
    import numpy as np
    import torch
    os.environ["FQ_LIB_PATH"] = OUT
    from quantization.mxnet_amd import ops
    raw = ctypes.CDLL(OUT)
    dev = torch.device("cuda", 0)
    n = 128
    torch.manual_seed(3)
    xs = [torch.randn(n, 3, 224, 224, device=dev) for _ in range(4)]
    w = torch.randn(32, 3, 3, 3, device=dev) * 0.3
    sc = torch.rand(32, device=dev) + 0.5
    sh = torch.randn(32, device=dev)
    run = lambda i: ops.stem_conv_s2(xs[i % 4], w, None, bn_scale=sc, bn_shift=sh, act="relu")
    for i in range(4):
        run(i)
    torch.cuda.synchronize()
    buf = torch.zeros(8 * 65536 * 4, dtype=torch.int64, device=dev)
    raw.fq_debug_set_pw_trace(ctypes.c_void_p(buf.data_ptr()))
    run(0)
    torch.cuda.synchronize()
    raw.fq_debug_set_pw_trace(ctypes.c_void_p(0))
    t = buf.cpu().numpy().reshape(-1, 8)
    t = t[t[:, 0] != 0]
    st = (t[:, :6].astype(np.float64) - t[:, 0].min()) / 100.0
    print("first convolution 3x3 -> 32 on (128, 3, 224, 224): %d workgroups, kernel span %.1f us" % (len(t), st[:, 5].max()))
    for a, b, nm in [(0, 1, "constants -> LDS, barrier"), (1, 2, "first tile (wavefront 0)"), (2, 3, "tiles up to the middle of the range"),
                     (3, 4, "second half of the range"), (4, 5, "statistic flush")]:
        d = st[:, b] - st[:, a]
        print("   %-40s median %7.2f  p90 %7.2f us" % (nm, np.median(d), np.percentile(d, 90)))
    print("   start median %.2f max %.2f; end median %.2f max %.2f us" % (np.median(st[:, 0]), st[:, 0].max(), np.median(st[:, 5]), st[:, 5].max()))


if __name__ == "__main__":
    main()
