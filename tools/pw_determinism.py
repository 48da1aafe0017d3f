#!/usr/bin/env python3
"""Run-to-run determinism and agreement of the pointwise forms at full size: every shape is computed REPS times by the form
the shape-based choice takes (the sample form for these) and compared bit for bit with the first run and with the split form."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SHAPES = [(128, 512, 512, 14), (128, 256, 512, 14), (128, 256, 256, 28), (128, 128, 256, 28), (128, 1024, 256, 14),
          (96, 256, 1024, 14)]
REPS = 6


def main():
    import torch
    from quantization.mxnet_amd import ops
    dev = torch.device("cuda", 0)
    bad = 0
    for n, cin, cout, hw in SHAPES:
        torch.manual_seed(n + cin + cout + hw)
        x = torch.relu(torch.randn(n, cin, hw, hw, device=dev))
        w = torch.randn(cout, cin, 1, 1, device=dev) * 0.1
        sc = torch.rand(cout, device=dev) + 0.5
        sh = torch.randn(cout, device=dev)
        stat = ops.absmax_per_sample(x)
        codes, scales, rowsum = ops.weight_codes(w, cout, 8)

        def run(form):
            cur = torch.empty(1, device=dev)
            return ops.pwconv_i8(x, codes, scales, rowsum, in_stat=stat, width=8, flags=0, cur_out=cur, bn_scale=sc, bn_shift=sh,
                                 act="relu", form=form)
        y0, st0 = run(None)
        diffs = []
        for _ in range(REPS - 1):
            y, st = run(None)
            diffs.append(int((y != y0).sum().item()) + int((st != st0).sum().item()))
        ys, sts = run("split")
        vs = int((ys != y0).sum().item()) + int((sts != st0).sum().item())
        print("%4d x %4d -> %4d @%2dx%-2d: repeats differ %s, against the split form %d" % (n, cin, cout, hw, hw, diffs, vs))
        bad += sum(diffs) + vs
    print("OK" if bad == 0 else "MISMATCH")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
