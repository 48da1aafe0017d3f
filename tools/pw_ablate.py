#!/usr/bin/env python3
"""Where does the sample form of the pointwise convolution spend its time?  Builds fq_pw_sample.hip with one ingredient
removed at a time (-DFQ_PWSMP_ABL=<bits>, results are then wrong) and times every variant on the same shapes in one GPU call.

    python tools/pw_ablate.py build               # here; csrc/build/lib_pwabl_<bits>.so (one compilation each, --only)
    python tools/pw_ablate.py run [bits ...]      # on the GPU box
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
BUILD = os.path.join(ROOT, "quantization", "mxnet_amd", "csrc", "build")
NAMES = {0: "full", 1: "no MFMAs", 2: "no quantiser arithmetic", 4: "no barrier per chunk", 8: "no activation loads",
         16: "no output stores", 32: "no A-fragment loads", 64: "no chunk loop at all", 3: "no MFMAs, no quantiser",
         24: "no loads, no stores", 11: "no MFMAs, no quantiser, no loads", 80: "set-up + epilogue arithmetic only"}
SHAPES = [(512, 512, 14), (256, 512, 14), (256, 256, 28), (128, 256, 28)]


def lib(b):
    return os.path.join(BUILD, "lib_pwabl_%d.so" % b)


def child():
    import torch
    from quantization.mxnet_amd import ops
    dev = torch.device("cuda", 0)
    out = []
    for cin, cout, hw in SHAPES:
        torch.manual_seed(7)
        xs = [torch.relu(torch.randn(128, cin, hw, hw, device=dev)) for _ in range(3)]
        w = torch.randn(cout, cin, 1, 1, device=dev) * 0.1
        sc, sh = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev)
        stat = ops.absmax_per_sample(xs[0])
        cur = torch.empty(1, device=dev)
        codes, scales, rowsum = ops.weight_codes(w, cout, 8)
        def run(k):
            ops.pwconv_i8(xs[k % 3], codes, scales, rowsum, in_stat=stat, width=8, flags=0, cur_out=cur, bn_scale=sc,
                          bn_shift=sh, act="relu", form="sample")
        for k in range(3):
            run(k)
        torch.cuda.synchronize()
        # the host needs ~20 us per call: time the launches replayed from a hipGraph (20 per replay) - GPU time only
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for k in range(20):
                run(k)
        g.replay()
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            g.replay()
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) / 20.0)
        out.append("%6.1f" % (sorted(ts)[3] * 1e3))
    print("  ".join(out), flush=True)


def main():
    cmd = sys.argv[1]
    if cmd == "build":
        from quantization.mxnet_amd.csrc import build
        for b in NAMES:
            build.build_library(defines=["-DFQ_PWSMP_ABL=%d" % b], out=lib(b), verbose=False, only=["fq_pw_sample"])
            print("built", lib(b))
    elif cmd == "child":
        child()
    else:
        bits = [int(a) for a in sys.argv[2:]] or list(NAMES)
        print("%-36s %s   (us per launch, replayed from a hipGraph)" % ("variant", "  ".join("%d->%d@%d" % s for s in SHAPES)))
        for b in bits:
            r = subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, FQ_LIB_PATH=lib(b)),
                               capture_output=True, text=True)
            print("%-36s %s" % (NAMES.get(b, str(b)), r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]),
                  flush=True)


if __name__ == "__main__":
    main()
