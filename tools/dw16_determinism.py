#!/usr/bin/env python3
"""Run-to-run determinism of fq_dwconv3x3_c16 at full MobileNetV2 sizes (its rows are fetched two output rows ahead through
rotating register sets since round 4): every shape REPS times, bit for bit against the first run - half of the repeats
beside a second stream that keeps the CUs busy - and the first run against the oracle's codes of what the fp32 form
(fq_dwconv3x3 under the same stored threshold) computes."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SHAPES = [(128, 32, 112, 1), (128, 96, 112, 2), (128, 144, 56, 1), (128, 192, 28, 2), (128, 384, 14, 1), (128, 576, 14, 2),
          (128, 960, 7, 1)]
REPS = 50


def main():
    import torch
    from quantization.mxnet_amd import ops
    dev = torch.device("cuda", 0)
    bad = 0
    for n, c, hw, s in SHAPES:
        torch.manual_seed(n + c + hw + s)
        thr = torch.tensor([3.0], device=dev)
        x = torch.relu(torch.randn(n, c, hw, hw, device=dev)) * 2
        w = torch.randn(c, 1, 3, 3, device=dev) * 0.3
        sc, sh = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev) * 0.1
        oc = dict(thr=torch.tensor([2.5], device=dev), width=8, flags=0)
        kw = dict(stride=s, in_thr=thr, width=8, flags=0, bn_scale=sc, bn_shift=sh, act="relu6")
        # the input as a C16 code tensor: the library's own codes of x under `thr`, laid out [n][C/16][pixels][16] and stored
        # as (code + 128 - zoff) ^ 0x80 with zoff = 128 for unsigned codes
        xq, _, codes = ops.fake_quant_offline(x, thr, 8, 0, want_stat=False, want_codes=True)
        cb = (c + 15) // 16
        t = torch.zeros(n, cb * 16, hw * hw, dtype=torch.int32, device=dev)
        t[:, :c] = codes.reshape(n, c, hw * hw)
        x16 = (t.reshape(n, cb, 16, hw * hw).permute(0, 1, 3, 2) ^ 0x80).to(torch.int8).contiguous()
        xc = ops.Codes16(x16, (n, c, hw, hw), thr, 8, 0)

        def run():
            return ops.dwconv3x3_c16(xc, w, None, out_codes=oc, **kw)
        y0, st0 = run()
        want, want_st = ops.dwconv3x3(x, w, None, **kw)
        _, _, wc = ops.fake_quant_offline(want, oc["thr"], 8, 0, want_stat=False, want_codes=True)
        ho = want.shape[2]
        tw = torch.zeros(n, cb * 16, ho * ho, dtype=torch.int32, device=dev)
        tw[:, :c] = wc.reshape(n, c, ho * ho)
        w16 = (tw.reshape(n, cb, 16, ho * ho).permute(0, 1, 3, 2) ^ 0x80).to(torch.int8).contiguous()
        vs = int((w16 != y0.t).sum()) + int((want_st != st0).sum())
        diffs = torch.zeros(1, dtype=torch.int64, device=dev)
        side = torch.cuda.Stream()
        a = torch.randn(2048, 2048, device=dev)
        for r in range(REPS - 1):
            if r % 2:
                with torch.cuda.stream(side):
                    for _ in range(3):
                        a @ a
            y, st = run()
            diffs += (y.t != y0.t).sum() + (st != st0).sum()
        torch.cuda.synchronize()
        d = int(diffs.item())
        print("%4d x %4d @%3dx%-3d stride %d: codes differing over the repeats %d, against the fp32 form's codes %d" % (n, c, hw, hw, s, d, vs))
        bad += d + vs
    print("OK" if bad == 0 else "MISMATCH")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
