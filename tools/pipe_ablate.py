#!/usr/bin/env python3
"""Where does the pipe form of the pointwise convolution (fq_pw_pipe.hip) spend its time?  Builds the unit with one ingredient
removed at a time (-DFQ_PWPIPE_ABL=<bits>, results are then wrong) and times every variant on the same shapes in one GPU call,
next to the sample form of the default library.

    python tools/pipe_ablate.py build [extra -D...]    # here; build_tools/lib_pipeabl_<i>.so (one compilation each, --only)
    python tools/pipe_ablate.py run                    # on the GPU box
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUTDIR = os.path.join(ROOT, "build_tools")
VARIANTS = [("full", []), ("sub-tile loop unrolled", ["-DFQ_PWPIPE_UNROLL=1"]),
            ("no MFMAs", ["-DFQ_PWPIPE_ABL=1"]), ("no quantiser arithmetic", ["-DFQ_PWPIPE_ABL=2"]),
            ("no barriers", ["-DFQ_PWPIPE_ABL=4"]), ("no activation DMA", ["-DFQ_PWPIPE_ABL=8"]),
            ("no output stores", ["-DFQ_PWPIPE_ABL=16"]), ("no A-fragment loads", ["-DFQ_PWPIPE_ABL=32"]),
            ("no quantiser at all", ["-DFQ_PWPIPE_ABL=64"]), ("no multiplication at all", ["-DFQ_PWPIPE_ABL=128"]),
            ("no epilogue at all", ["-DFQ_PWPIPE_ABL=256"]), ("no DMA, no stores", ["-DFQ_PWPIPE_ABL=24"]),
            ("DMA + stores only", ["-DFQ_PWPIPE_ABL=%d" % (64 + 128 + 32)]),
            ("nothing but set-up + barriers", ["-DFQ_PWPIPE_ABL=%d" % (8 + 32 + 64 + 128 + 256)])]
SHAPES = [(512, 512, 14), (256, 512, 14)]


def lib(i):
    return os.path.join(OUTDIR, "lib_pipeabl_%d.so" % i)


def child(form):
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
                          bn_shift=sh, act="relu", form=form)
        for k in range(3):
            run(k)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()                      # GPU time only: 20 launches per replay
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
        os.makedirs(OUTDIR, exist_ok=True)
        extra = [a for a in sys.argv[2:] if a.startswith("-D")]
        for i, (name, defs) in enumerate(VARIANTS):
            build.build_library(defines=(defs + extra) or ["-DFQ_PWPIPE_ABL=0"], out=lib(i), verbose=False, only=["fq_pw_pipe"])
            print("built", lib(i), name)
    elif cmd == "child":
        child(sys.argv[2])
    else:
        print("%-36s %s   (us per launch, replayed from a hipGraph)" % ("variant", "  ".join("%d->%d@%d" % s for s in SHAPES)))
        r = subprocess.run([sys.executable, __file__, "child", "sample"], capture_output=True, text=True)
        print("%-36s %s" % ("(sample form, default library)", r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]))
        for i, (name, _) in enumerate(VARIANTS):
            if not os.path.exists(lib(i)):
                continue
            r = subprocess.run([sys.executable, __file__, "child", "pipe"], env=dict(os.environ, FQ_LIB_PATH=lib(i)),
                               capture_output=True, text=True)
            print("%-36s %s" % (name, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]), flush=True)


if __name__ == "__main__":
    main()
