#!/usr/bin/env python3
"""Run-to-run determinism and agreement of the depthwise forms at full MobileNet sizes: every shape is computed REPS times by
the form the shape-based choice takes and compared bit for bit with the first run and with the column-walking forms
(FQ_DW_FLAT=0 FQ_DW_PLANES=0 in a child process).  Half of the repeats run beside a second stream that keeps the CUs busy with
matrix products (another occupancy and another memory-pipeline load than the quiet box: the store-data hazard of round 3,
profiles/r3_dw_flat_race.txt, only showed under load)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SHAPES = [(128, 256, 28, 1), (128, 256, 28, 2), (128, 512, 14, 1), (128, 512, 14, 2), (128, 1024, 7, 1), (96, 576, 14, 1),
          (128, 960, 7, 1), (128, 384, 14, 2)]
REPS = 50


def main():
    import numpy as np
    import torch
    from quantization.mxnet_amd import ops
    child = os.environ.get("FQ_DW_DET_CHILD") == "1"
    dev = torch.device("cuda", 0)
    bad = 0
    for n, c, hw, s in SHAPES:
        torch.manual_seed(n + c + hw + s)
        x = torch.relu(torch.randn(n, c, hw, hw, device=dev))
        w = torch.randn(c, 1, 3, 3, device=dev) * 0.3
        sc = torch.rand(c, device=dev) + 0.5
        sh = torch.randn(c, device=dev)
        stat = ops.absmax_per_sample(x)

        def run():
            cur = torch.empty(1, device=dev)
            y, st = ops.dwconv3x3(x, w, None, stride=s, in_stat=stat, width=8, flags=0, cur_out=cur, bn_scale=sc, bn_shift=sh,
                                  act="relu")
            return y, st
        y0, st0 = run()
        path = "/tmp/dwdet_%d_%d_%d_%d.npy" % (n, c, hw, s)
        if child:
            np.save(path, y0.cpu().numpy())
            continue
        diffs = torch.zeros(1, dtype=torch.int64, device=dev)
        side = torch.cuda.Stream()
        a = torch.randn(2048, 2048, device=dev)
        for r in range(REPS - 1):
            if r % 2:                                         # a competing stream during every second repeat
                with torch.cuda.stream(side):
                    for _ in range(3):
                        a @ a
            y, st = run()
            diffs += (y != y0).sum() + (st != st0).sum()
        torch.cuda.synchronize()
        diffs = [int(diffs.item())]
        ref = np.load(path)
        vs = int((ref != y0.cpu().numpy()).sum())
        print("%4d x %4d @%2dx%-2d stride %d: outputs differing over the repeats %s, against the column-walking forms %d" % (n, c, hw, hw, s, diffs, vs))
        bad += sum(diffs) + vs
    if not child:
        print("OK" if bad == 0 else "MISMATCH")
        sys.exit(1 if bad else 0)


if __name__ == "__main__":
    if os.environ.get("FQ_DW_DET_CHILD") != "1":
        env = dict(os.environ, FQ_DW_DET_CHILD="1", FQ_DW_FLAT="0", FQ_DW_PLANES="0")
        subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, check=True)
    main()
