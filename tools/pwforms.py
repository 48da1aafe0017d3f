#!/usr/bin/env python3
"""A/B of the pointwise forms on the deep layers (batch 128), each call preceded by a 512 MB fill that evicts the weights and
activations from L2 / Infinity Cache (in the model every layer meets its weights cold).  HIP events around the call only.

    python tools/pwforms.py [--layers mobilenet|resnet|all] [--iters 15]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantization.mxnet_amd import ops  # noqa: E402
from quantization.mxnet_amd._lib import FakeQuantError  # noqa: E402

MOBILENET = [(256, 512, 14), (512, 512, 14), (512, 1024, 7), (1024, 1024, 7)]
MBV2 = [(16, 96, 112), (24, 144, 56), (144, 24, 56), (144, 32, 28), (192, 32, 28), (192, 64, 14), (64, 384, 14), (384, 64, 14),
        (384, 96, 14), (96, 576, 14), (576, 96, 14), (576, 160, 7), (160, 960, 7), (960, 160, 7), (960, 320, 7), (320, 1280, 7)]
MID = [(128, 128, 56), (128, 256, 28), (256, 256, 28)]
RESNET = [(256, 1024, 14), (1024, 256, 14), (1024, 512, 14), (512, 2048, 7), (256, 64, 56), (64, 256, 56), (512, 128, 28),
          (128, 512, 28), (2048, 512, 7)]
VARIANTS = [("auto", None, {}), ("stream", "stream", {}), ("two_kernels", "two_kernels", {}), ("split", "split", {}),
            ("split lb3 cw1", "split", {"FQ_PWS_CFG": "31"}), ("split lb3 cw2", "split", {"FQ_PWS_CFG": "32"}),
            ("split lb3 cw4", "split", {"FQ_PWS_CFG": "34"}), ("split lb4 cw1", "split", {"FQ_PWS_CFG": "41"}),
            ("split lb4 cw2", "split", {"FQ_PWS_CFG": "42"})]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", default="mobilenet")
    ap.add_argument("--iters", type=int, default=15)
    ap.add_argument("--hot", action="store_true", help="no cache flush between calls")
    args = ap.parse_args()
    layers = {"mobilenet": MOBILENET, "resnet": RESNET, "mid": MID, "mbv2": MBV2, "all": MOBILENET + RESNET}[args.layers]
    dev = torch.device("cuda", 0)
    n = 128
    flush = torch.empty(128 * 1024 * 1024, dtype=torch.float32, device=dev)
    for cin, cout, hw in layers:
        torch.manual_seed(7)
        x = torch.relu(torch.randn(n, cin, hw, hw, device=dev))
        w = torch.randn(cout, cin, 1, 1, device=dev) * 0.1
        sc = torch.rand(cout, device=dev) + 0.5
        sh = torch.randn(cout, device=dev)
        stat = ops.absmax_per_sample(x)
        cur = torch.empty(1, device=dev)
        codes, scales, rowsum = ops.weight_codes(w, cout, 8)
        nbytes = 4 * (x.numel() + n * cout * hw * hw)
        ref = None
        print("%4d->%4d @%dx%d  %.1f MB algorithmic" % (cin, cout, hw, hw, nbytes / 1e6))
        for name, form, env in VARIANTS:
            os.environ["FQ_PWS_CFG"] = "0"
            os.environ.update(env)

            def run():
                return ops.pwconv_i8(x, codes, scales, rowsum, in_stat=stat, width=8, flags=0, cur_out=cur, bn_scale=sc,
                                     bn_shift=sh, act="relu", form=form)
            try:
                y, st = run()
            except FakeQuantError as e:
                print("    %-20s n/a (%s)" % (name, str(e)[:60]))
                continue
            if ref is None:
                ref = (y.clone(), st.clone())
            same = torch.equal(y, ref[0]) and torch.equal(st, ref[1])
            ts = []
            for _ in range(args.iters):
                if not args.hot:
                    flush.fill_(1.0)
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                run()
                b.record()
                torch.cuda.synchronize()
                ts.append(a.elapsed_time(b))
            ts.sort()
            med = ts[len(ts) // 2]
            print("    %-20s %8.1f us  %7.1f GB/s  %.3f of 8 TB/s   identical=%s" % (name, med * 1e3, nbytes / med / 1e6,
                                                                                    nbytes / med / 1e6 / 8000.0, same))


if __name__ == "__main__":
    main()
