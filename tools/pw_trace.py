#!/usr/bin/env python3
"""Per-workgroup phase timeline of the fused pointwise kernel (debug build with -DFQ_PW_TRACE, built by this script into
build_tools/).  Run on the GPU box:  python tools/pw_trace.py [cin cout hw].  Build on the CPU box first: --build-only."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
from quantization.mxnet_amd.csrc import build as B  # noqa: E402

OUT = os.path.join(ROOT, "quantization", "mxnet_amd", "csrc", "build", "libfakequant_trace.so")   # (csrc/build travels to the GPU box)


def build():
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    B.build_library(defines=["-DFQ_PW_TRACE=1"], out=OUT, amalgamate=True)


def main():
    if "--build-only" in sys.argv:
        build()
        return
    import numpy as np
    import torch
    os.environ["FQ_LIB_PATH"] = OUT
    from quantization.mxnet_amd import ops
    raw = ctypes.CDLL(OUT)
    args = [int(a) for a in sys.argv[1:] if a.isdigit() and len(a) > 1]
    shapes = [tuple(args[:3])] if len(args) >= 3 else [(32, 64, 112), (128, 128, 56), (256, 256, 28), (512, 512, 14), (1024, 1024, 7)]
    dev = torch.device("cuda", 0)
    n = 128
    form = sys.argv[sys.argv.index("--form") + 1] if "--form" in sys.argv else None
    for cin, cout, hw in shapes:
        torch.manual_seed(7)
        x = torch.relu(torch.randn(n, cin, hw, hw, device=dev))
        w = torch.randn(cout, cin, 1, 1, device=dev) * 0.1
        sc = torch.rand(cout, device=dev) + 0.5
        sh = torch.randn(cout, device=dev)
        stat = ops.absmax_per_sample(x)
        cur = torch.empty(1, device=dev)
        codes, scales, rowsum = ops.weight_codes(w, cout, 8)
        buf = torch.zeros(8 * 65536 * 4, dtype=torch.int64, device=dev)
        run = lambda: ops.pwconv_i8(x, codes, scales, rowsum, in_stat=stat, width=8, flags=0, cur_out=cur, bn_scale=sc,
                                    bn_shift=sh, act="relu", form=form)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        if "--ablate" in sys.argv:
            from kbench import timeit
            for dbg, nm in [(0, "full"), (4, "full, generic epilogue"), (1, "no output stores"), (2, "no activation loads"), (3, "neither"), (0, "full again")]:
                raw.fq_debug_set_pw_dbg(dbg)
                med, _ = timeit(run, 20)
                print("%d->%d %dx%d  %-22s %7.1f us" % (cin, cout, hw, hw, nm, med * 1e3))
            raw.fq_debug_set_pw_dbg(0)
            continue
        dbg = int(sys.argv[sys.argv.index("--dbg") + 1]) if "--dbg" in sys.argv else 0
        raw.fq_debug_set_pw_dbg(dbg)
        assert raw.fq_debug_set_pw_trace(ctypes.c_void_p(buf.data_ptr())) == 0
        run()
        torch.cuda.synchronize()
        raw.fq_debug_set_pw_trace(ctypes.c_void_p(0))
        t = buf.cpu().numpy().reshape(-1, 8)
        t = t[t[:, 0] != 0]
        st = t[:, :6].astype(np.float64)
        t0 = st[:, 0].min()
        st = (st - t0) / 100.0                                   # wall_clock64 ticks at 100 MHz -> us
        hwid = t[:, 7]
        xcc = (hwid >> 32) & 0xF
        cu = (hwid >> 8) & 0xF
        se = (hwid >> 13) & 0x7
        sh_ = (hwid >> 12) & 1
        cuid = xcc * 64 + se * 16 + sh_ * 8 + cu                 # not dense, just unique
        print("%d->%d %dx%d: %d workgroups on %d distinct CUs; kernel span %.1f us" % (cin, cout, hw, hw, len(t),
              len(set(cuid.tolist())), st[:, 5].max()))
        names = ["start", "setup done", "phase1 issued", "phase1 barrier", "gemm(block0) done", "end"]
        if form == "sample":
            names = ["start", "set-up done", "chunk loop done", "stores issued", "statistic done", "end"]
            tt = t.astype(np.float64)
            for a, b_, nm in [(1, 6, "first chunk quantised + barrier"), (6, 7, "first half of the chunks"), (7, 2, "second half of the chunks")]:
                dd = (tt[:, b_] - tt[:, a]) / 100.0
                print("   sample form: %-36s median %7.2f  p90 %7.2f us" % (nm, np.median(dd), np.percentile(dd, 90)))
        if form == "split":
            names = ["start", "set-up done", "quantise done", "barrier passed", "multiply done", "end"]
        for i, nm in enumerate(names):
            print("   %-18s min %7.2f  median %7.2f  max %7.2f us" % (nm, st[:, i].min(), np.median(st[:, i]), st[:, i].max()))
        d = np.diff(st, axis=1)
        dn = ["setup", "phase1 (load+quant+LDS)", "barrier wait", "gemm block0", "epilogue(+other blocks)"]
        if form == "sample":
            dn = ["set-up (threshold, constants)", "chunk loop", "epilogue + stores", "statistic", "-"]
        if form == "split":
            dn = ["set-up (mean, constants)", "quantise -> panel", "barrier wait", "multiply", "epilogue + statistic"]
        for i, nm in enumerate(dn):
            print("   d %-24s median %7.2f  p90 %7.2f us" % (nm, np.median(d[:, i]), np.percentile(d[:, i], 90)))
        late = (st[:, 0] > 1.0).sum()
        print("   workgroups starting later than 1 us after the first: %d" % late)


if __name__ == "__main__":
    main()
