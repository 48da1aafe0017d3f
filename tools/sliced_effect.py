"""Effect of config 5's sliced 3x3 filters on the logits, at the BASELINE batch (128, 3, 224, 224).

The fused path of a Winograd-domain quantised net multiplies  m * p  with three int8 digit slices m = 2^14 d0 + 2^7 d1 + d2
instead of the fp32 filter g^ = GI U^ GTI (convert_conv2d.py:81-83): |g^ - m p| <= 2^-20 max|g^_c|.  Three nets on the same
input and weights:
  A  un-fused            : the 3x3 layers go through the tensor library's fp32 convolution of g^ (the reference's F.Convolution)
  B  fused, not sliced   : FQ_WINO_SLICED=0 - fused producers, the 3x3 layers still the tensor library's convolution of g^
  C  fused, sliced       : the default
Reported: max |logit difference| / max |logit| and top-1 agreement for A-B (what fusing alone changes: folded BatchNorm and an
integer pointwise path differ from MIOpen in the last bit, which flips a few 8-bit rounding decisions downstream), B-C (what the
sliced filter adds) and A-C.      python tools/sliced_effect.py [--batch 128] [--model resnet50_v1] [--wino F43]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))


def measure(batch=128, model="resnet50_v1", wino="F43"):
    """-> (max|logit| of the un-fused net, {pair: (max |difference|, mean |difference|, top-1 agreements)}, batch)"""
    a = argparse.Namespace(batch=batch, model=model, wino=wino)
    from quantization.mxnet_amd import mx
    from quantization.mxnet_amd.quantize import fuse
    from test_gpu_net import _build as build
    rng = np.random.default_rng(8)
    X = mx.nd.array(rng.standard_normal((a.batch, 3, 224, 224)).astype(np.float32), ctx=mx.gpu(0))

    def run(fused, sliced):
        old = fuse.WINO_SLICED
        fuse.WINO_SLICED = sliced
        try:
            net = build(a.model, 1000, mx.gpu(0), quant_type="channel", wino=a.wino)
            net.fix_params()
            net.quantize_input(enable=True, online=True)
            net(mx.nd.NDArray(X._t[:2].contiguous()))
            if fused:
                fuse.fuse_inference(net)
            return net(X).asnumpy()
        finally:
            fuse.WINO_SLICED = old
    A, B, C = run(False, True), run(True, False), run(True, True)
    scale = float(np.abs(A).max())
    out = {}
    for name, p, q in (("A-B", A, B), ("B-C", B, C), ("A-C", A, C)):
        d = np.abs(p - q)
        out[name] = (float(d.max()), float(d.mean()), int((p.argmax(1) == q.argmax(1)).sum()))
    return scale, out, a.batch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--model", default="resnet50_v1")
    ap.add_argument("--wino", default="F43")
    a = ap.parse_args()
    scale, out, n = measure(a.batch, a.model, a.wino)
    print("%s %s batch %d: max|logit| %.4g" % (a.model, a.wino, a.batch, scale))
    for key, name in (("A-B", "A-B  un-fused vs fused (no slices)"), ("B-C", "B-C  fused: fp32 filter vs sliced"),
                      ("A-C", "A-C  un-fused vs fused + sliced")):
        mx_, mean_, agree = out[key]
        print("  %-38s max %.3e (%.2e of max|logit|)  mean %.3e  top-1 agreement %d / %d" % (name, mx_, mx_ / scale, mean_, agree, n))


if __name__ == "__main__":
    main()
