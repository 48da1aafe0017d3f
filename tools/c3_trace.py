#!/usr/bin/env python3
"""Per-workgroup phase stamps of the dense 3x3 kernel (debug build -DFQ_PW_TRACE, see pw_trace.py) on ResNet-50's four
stages at batch 128.  Build here first (`--build-only`), run on the GPU box:  python tools/c3_trace.py [cin hw]."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
from quantization.mxnet_amd.csrc import build as B  # noqa: E402

OUT = os.path.join(ROOT, "build_tools", "libfakequant_trace.so")


def main():
    if "--build-only" in sys.argv:
        B.build_library(defines=["-DFQ_PW_TRACE=1"], out=OUT, amalgamate=True)
        return
    import numpy as np
    import torch
    os.environ["FQ_LIB_PATH"] = OUT
    from quantization.mxnet_amd import ops
    from kbench import timeit
    raw = ctypes.CDLL(OUT)
    args = [int(a) for a in sys.argv[1:] if a.isdigit()]
    shapes = [tuple(args[:2])] if len(args) >= 2 else [(64, 56), (128, 28), (256, 14), (512, 7)]
    dev = torch.device("cuda", 0)
    n = 128
    for cin, hw in shapes:
        torch.manual_seed(7)
        x = torch.relu(torch.randn(n, cin, hw, hw, device=dev))
        w = torch.randn(cin, cin, 3, 3, device=dev) * 0.1
        sc = torch.rand(cin, device=dev) + 0.5
        sh = torch.randn(cin, device=dev)
        stat = ops.absmax_per_sample(x)
        cur = torch.empty(1, device=dev)
        codes, scales, rowsum = ops.weight_codes_3x3(w, cin, 8)
        run = lambda: ops.conv3x3_i8(x, codes, scales, rowsum, in_stat=stat, width=8, flags=0, cur_out=cur, bn_scale=sc,
                                     bn_shift=sh, act="relu")
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        med, _ = timeit(run, 30)
        buf = torch.zeros(8 * 65536 * 4, dtype=torch.int64, device=dev)
        assert raw.fq_debug_set_pw_trace(ctypes.c_void_p(buf.data_ptr())) == 0
        run()
        torch.cuda.synchronize()
        raw.fq_debug_set_pw_trace(ctypes.c_void_p(0))
        t = buf.cpu().numpy().reshape(-1, 8)
        t = t[t[:, 0] != 0]
        st = (t[:, :6].astype(np.float64) - t[:, 0].min()) / 100.0
        print("3x3 %d -> %d @%dx%d: %.1f us (events); %d workgroups, span %.1f us" % (cin, cin, hw, hw, med * 1e3, len(t),
                                                                                 st[:, 5].max()))
        t6 = (t[:, 6:8].astype(np.float64) - t[:, 0].min()) / 100.0
        st = np.concatenate([st, t6], axis=1)
        for a, b, nm in [(0, 6, "set-up (first loads issued, threshold)"), (6, 7, "quantise -> panel (+ later loads)"),
                         (7, 1, "ring / tap masks"), (1, 2, "barrier wait"), (2, 3, "multiply"),
                         (3, 4, "epilogue + stores"), (4, 5, "statistic flush")]:
            d = st[:, b] - st[:, a]
            print("   %-36s median %7.2f  p90 %7.2f us" % (nm, np.median(d), np.percentile(d, 90)))
        print("   start median %.2f max %.2f; end median %.2f max %.2f us" % (np.median(st[:, 0]), st[:, 0].max(),
                                                                            np.median(st[:, 5]), st[:, 5].max()))


if __name__ == "__main__":
    main()
