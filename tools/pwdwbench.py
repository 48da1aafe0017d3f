"""The recompute pair alone on the GPU (round 6): fq_pwconv_i8 + fq_dwconv3x3 (two launches, the tensor between them written and
read) against fq_pwconv_i8_stat + fq_pwdw_fused, on MobileNet1.0's pointwise -> depthwise pairs at batch 128.

    python tools/pwdwbench.py [--batch 128] [--reps 30] [--pairs 1,2,3,4,5]

Per pair: time of each launch (HIP events on the launch stream, median over reps, all launches back to back), bytes each form
moves, and the resulting TB/s.  Values are checked bit-equal before timing."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantization.mxnet_amd import ops  # noqa: E402

PAIRS = {1: (32, 64, 112, 2), 2: (64, 128, 56, 1), 3: (128, 128, 56, 2), 4: (128, 256, 28, 1), 5: (256, 256, 28, 2),
         6: (256, 512, 14, 1), 7: (512, 512, 14, 1)}


def timed(fn, reps):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    return float(np.median(ts))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--pairs", default="1,2,3,4,5")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    n = a.batch
    print("pair  shape                          two launches: pw + dw = total (us)   recompute: stat + fused = total (us)   "
          "MB moved two / fused   speed-up")
    tot2 = tot1 = 0.0
    for p in [int(v) for v in a.pairs.split(",")]:
        cin, cout, hw, stride = PAIRS[p]
        x = torch.relu(torch.randn(n, cin, hw, hw, device=dev)) * 1.7
        w1 = torch.randn(cout, cin, device=dev) * 0.2
        w2 = ops.weight_fake_quant(torch.randn(cout, 1, 3, 3, device=dev) * 0.3, cout, 8)
        codes, scales, rowsum = ops.weight_codes(w1, cout, 8)
        sc1, sh1 = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev) * 0.3
        sc2, sh2 = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev) * 0.3
        xstat = ops.absmax_per_sample(x)
        cur1, cur2 = torch.zeros(1, device=dev), torch.zeros(1, device=dev)
        if not ops.pwdw_supported(x.shape, cout, stride):
            print("%4d  %s: shape not taken" % (p, (n, cin, cout, hw, stride)))
            continue
        st = {}

        def pw():
            st["y"], st["ys"] = ops.pwconv_i8(x, codes, scales, rowsum, None, in_stat=xstat, cur_out=cur1, bn_scale=sc1,
                                              bn_shift=sh1, act="relu")

        def dw():
            st["z"], st["zs"] = ops.dwconv3x3(st["y"], w2, None, stride=stride, in_stat=st["ys"], cur_out=cur2, bn_scale=sc2,
                                              bn_shift=sh2, act="relu")

        def sa():
            st["ys1"] = ops.pwconv_i8_stat(x, codes, scales, rowsum, None, in_stat=xstat, cur_out=cur1, bn_scale=sc1,
                                           bn_shift=sh1, act="relu")

        def fb():
            st["z1"], st["zs1"] = ops.pwdw_fused(x, codes, scales, rowsum, w2, in_stat=xstat, pw_bn_scale=sc1, pw_bn_shift=sh1,
                                                 pw_act="relu", mid_stat=st["ys1"], mid_cur_out=cur2, stride=stride,
                                                 dw_bn_scale=sc2, dw_bn_shift=sh2, dw_act="relu")
        pw(); dw(); sa(); fb()
        torch.cuda.synchronize()
        ok = torch.equal(st["z"], st["z1"]) and torch.equal(st["zs"], st["zs1"]) and torch.equal(st["ys"], st["ys1"])
        t_pw, t_dw, t_sa, t_fb = timed(pw, a.reps), timed(dw, a.reps), timed(sa, a.reps), timed(fb, a.reps)
        ho = (hw - 1) // stride + 1
        xb, yb, zb = 4e-6 * n * cin * hw * hw, 4e-6 * n * cout * hw * hw, 4e-6 * n * cout * ho * ho
        mb2, mb1 = xb + 2 * yb + zb, 2 * xb + zb
        tot2 += t_pw + t_dw
        tot1 += t_sa + t_fb
        print("%4d  %3d->%3d @%3dx%-3d dw stride %d   %7.1f + %6.1f = %7.1f (%4.2f TB/s)   %6.1f + %6.1f = %7.1f (%4.2f TB/s)   "
              "%6.0f / %5.0f   %5.2fx  %s" % (p, cin, cout, hw, hw, stride, t_pw, t_dw, t_pw + t_dw, mb2 / (t_pw + t_dw),
                                            t_sa, t_fb, t_sa + t_fb, mb1 / (t_sa + t_fb), mb2, mb1,
                                            (t_pw + t_dw) / (t_sa + t_fb), "bit-equal" if ok else "VALUES DIFFER"))
    print("sum of the listed pairs: two launches %.1f us, recompute %.1f us" % (tot2, tot1))


if __name__ == "__main__":
    main()
