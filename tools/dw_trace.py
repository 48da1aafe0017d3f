#!/usr/bin/env python3
"""Per-workgroup phase stamps of the column-walking depthwise kernel (debug build -DFQ_PW_TRACE, see pw_trace.py)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
OUT = os.path.join(ROOT, "build_tools", "libfakequant_trace.so")


def main():
    import numpy as np
    import torch
    os.environ["FQ_LIB_PATH"] = OUT
    from quantization.mxnet_amd import ops
    raw = ctypes.CDLL(OUT)
    dev = torch.device("cuda", 0)
    n = 128
    for c, hw, s, quant in [(512, 14, 1, True), (512, 14, 1, False), (512, 14, 2, True), (1024, 7, 1, True), (256, 28, 1, True)]:
        torch.manual_seed(3)
        x = torch.relu(torch.randn(n, c, hw, hw, device=dev))
        w = torch.randn(c, 1, 3, 3, device=dev) * 0.3
        sc = torch.rand(c, device=dev) + 0.5
        sh = torch.randn(c, device=dev)
        stat = ops.absmax_per_sample(x)
        cur = torch.empty(1, device=dev)
        if quant:
            run = lambda: ops.dwconv3x3(x, w, None, stride=s, in_stat=stat, width=8, flags=0, cur_out=cur, bn_scale=sc,
                                        bn_shift=sh, act="relu")
        else:
            run = lambda: ops.dwconv3x3(x, w, None, stride=s, bn_scale=sc, bn_shift=sh, act="relu")
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        buf = torch.zeros(8 * 65536 * 4, dtype=torch.int64, device=dev)
        raw.fq_debug_set_pw_trace(ctypes.c_void_p(buf.data_ptr()))
        run()
        torch.cuda.synchronize()
        raw.fq_debug_set_pw_trace(ctypes.c_void_p(0))
        t = buf.cpu().numpy().reshape(-1, 8)
        t = t[t[:, 0] != 0]
        st = (t[:, :6].astype(np.float64) - t[:, 0].min()) / 100.0
        print("C=%d %dx%d s%d quant=%s: %d workgroups, kernel span %.1f us" % (c, hw, hw, s, quant, len(t), st[:, 5].max()))
        print("   start: median %.2f  max %.2f us" % (np.median(st[:, 0]), st[:, 0].max()))
        # (the whole-plane form K2o stamps: 1 = first block requested + qparams, 2 = first block done, 3 = all blocks done)
        for a, b, nm in [(0, 1, "batch mean / qparams"), (1, 2, "to the first block (K2o: first block)"),
                         (2, 3, "row loop (K2o: remaining blocks)"), (3, 5, "statistic tail")]:
            ok2 = (t[:, a] != 0) & (t[:, b] != 0)
            d = (st[:, b] - st[:, a])[ok2]
            print("   %-32s median %6.2f  p90 %6.2f us" % (nm, np.median(d), np.percentile(d, 90)))
        if (t[:, 4] != 0).all():
            d = (t[:, 4] - t[:, 1]) / 100.0
            e = (t[:, 2] - t[:, 4]) / 100.0
            ok = t[:, 2] != 0
            print("   K2o: wait for the first block's data  median %6.2f  p90 %6.2f us;  first pass through the block code median %6.2f  p90 %6.2f us"
                  % (np.median(d), np.percentile(d, 90), np.median(e[ok]) if ok.any() else -1, np.percentile(e[ok], 90) if ok.any() else -1))
        if (t[:, 6] != 0).all() and (t[:, 7] != 0).all():
            ok = t[:, 2] != 0
            for a, b, nm in [(0, 6, "arguments + index math"), (6, 7, "first block requested"), (7, 1, "batch statistic -> quantiser parameters")]:
                d = ((t[:, b] - t[:, a]) / 100.0)[ok]
                if len(d):
                    print("   K2o prologue: %-50s median %6.2f  p90 %6.2f us" % (nm, np.median(d), np.percentile(d, 90)))
        print("   whole workgroup                  median %6.2f  p90 %6.2f us" % (np.median(st[:, 5] - st[:, 0]), np.percentile(st[:, 5] - st[:, 0], 90)))


if __name__ == "__main__":
    main()
