#!/usr/bin/env python3
"""Reproducer + bisect for the run-to-run irreproducibility of dwconv3x3_flat_kernel<S=2, 28x28> (dropped in round 2,
profiles/r2_dw_planes.txt; cause found in round 3: profiles/r3_dw_flat_race.txt).  Builds the library in several variants -
the unguarded 16-byte buffer stores of round 2 alone and with each bisect knob, and the guarded stores - and runs each at full
size (128 x 256 x 28 x 28, stride 2) REPS times against the four-columns-per-lane form:

    python tools/dw_race_repro.py build            # here (hipcc cross-compiles): csrc/build/lib_dwrace_<variant>.so
    python tools/dw_race_repro.py run [variant..]  # on the GPU box; prints one line per (variant, workgroups per CU)
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
BUILD = os.path.join(ROOT, "quantization", "mxnet_amd", "csrc", "build")
U = "-DFQ_BUFST_NOPS=-1"        # the store-data guard of fq_common.h switched off = the code of round 2
VARIANTS = {"base": [U], "sync": [U, "-DFQ_DWF_SYNC"], "drain": [U, "-DFQ_DWF_DRAIN"], "pad40k": [U, "-DFQ_DWF_PADLDS=40960"],
            "noprefetch": [U, "-DFQ_DWF_NOPREFETCH"], "nodpp": [U, "-DFQ_DWF_NODPP"],
            "nop0": ["-DFQ_BUFST_NOPS=0"], "nop1": ["-DFQ_BUFST_NOPS=1"]}
REPS = int(os.environ.get("FQ_RACE_REPS", "30"))
N, C, HW, S = 128, 256, 28, 2


def lib(v):
    return os.path.join(BUILD, "lib_dwrace_%s.so" % v)


def child(mode):
    import numpy as np
    import torch
    from quantization.mxnet_amd import ops
    dev = torch.device("cuda", 0)
    torch.manual_seed(5)
    x = torch.relu(torch.randn(N, C, HW, HW, device=dev))
    w = torch.randn(C, 1, 3, 3, device=dev) * 0.3
    sc, sh = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
    stat = ops.absmax_per_sample(x)
    run = lambda: ops.dwconv3x3(x, w, None, stride=S, in_stat=stat, width=8, flags=0, cur_out=torch.empty(1, device=dev),
                                bn_scale=sc, bn_shift=sh, act="relu")[0]
    if mode == "ref":
        np.save("/tmp/dwrace_ref.npy", run().cpu().numpy())
        return
    ref = torch.from_numpy(np.load("/tmp/dwrace_ref.npy")).to(dev)
    side = torch.cuda.Stream()
    a = torch.randn(4096, 4096, device=dev)
    counts, first_bad = [], None
    for r in range(REPS):
        if os.environ.get("FQ_RACE_NOISE") == "1":                  # a second stream competing for the CUs
            with torch.cuda.stream(side):
                for _ in range(2):
                    a @ a
        y = run()
        bad = (y != ref)
        counts.append(int(bad.sum().item()))
        if counts[-1] and first_bad is None:
            first_bad = (y.clone(), bad.clone())
    print("  differing outputs per repeat: %s" % counts)
    if first_bad is not None:
        y, bad = first_bad
        idx = bad.reshape(-1).nonzero().reshape(-1).cpu().numpy()
        plane, off = idx // 196, idx % 196
        blk, wave, j = plane // 16, (plane % 16) // 4, plane % 4
        grid = 256 * int(os.environ.get("FQ_DW_FLAT_WG_PER_CU", "3"))
        per, rem = 2048 // grid, 2048 % grid
        wg = np.where(blk < rem * (per + 1), blk // (per + 1), rem + (blk - rem * (per + 1)) // max(per, 1))
        first_blk = np.where(wg < rem, wg * (per + 1), rem * (per + 1) + (wg - rem) * per)
        vals = y.reshape(-1)[torch.from_numpy(idx).to(dev)].cpu().numpy()
        print("  first bad repeat: %d outputs; got==0: %d; plane-in-wave j hist %s; wave hist %s; (off %% 4) hist %s; "
              "block ordinal in its workgroup hist %s; rows %s" % (
                  len(idx), int((vals == 0).sum()), np.bincount(j, minlength=4), np.bincount(wave, minlength=4),
                  np.bincount(off % 4, minlength=4), np.bincount(blk - first_blk)[:6], np.bincount(off // 14, minlength=14)))
        F = j * 196 + off
        u, c = np.unique(F, return_counts=True)
        print("  float index inside the wavefront's output tile -> count: %s" % dict(zip(u.tolist(), c.tolist())))
        ref_v = ref.reshape(-1)[torch.from_numpy(idx).to(dev)].cpu().numpy()
        print("  samples (got, want): %s" % [(float(a), float(b)) for a, b in zip(vals[:6], ref_v[:6])])
        ub, cb = np.unique(blk, return_counts=True)
        print("  blocks hit: %d of 2048; outputs per hit block: min %d max %d; first hit blocks %s" % (
            len(ub), cb.min(), cb.max(), ub[:12].tolist()))
    sys.exit(1 if sum(counts) else 0)


def main():
    cmd = sys.argv[1]
    if cmd == "build":
        from quantization.mxnet_amd.csrc import build
        for v, d in VARIANTS.items():
            build.build_library(defines=list(d), out=lib(v), verbose=False)
            print("built", lib(v))
    elif cmd == "child":
        child(sys.argv[2])
    else:
        subprocess.run([sys.executable, __file__, "child", "ref"], env=dict(os.environ, FQ_DW_FLAT="0"), check=True)
        for v in (sys.argv[2:] or list(VARIANTS)):
            for wg in os.environ.get("FQ_RACE_WG", "1 2 3").split():
                print("variant %-10s workgroups per CU asked for: %s  noise=%s" % (v, wg, os.environ.get("FQ_RACE_NOISE", "0")),
                      flush=True)
                subprocess.run([sys.executable, __file__, "child", "run"],
                               env=dict(os.environ, FQ_LIB_PATH=lib(v), FQ_DW_FLAT_WG_PER_CU=wg))


if __name__ == "__main__":
    main()
