#!/usr/bin/env python3
"""Quantisation-aware fine-tuning loop of the reference notebook (examples/quantize_aware_training_cifar10.ipynb):

  cells 6-7   converter = {Conv2D: gen_conv2d_converter(quant_type="channel", fake_bn=True, input_width=4,
              weight_width=4), Dense: gen_dense_converter(quant_type="channel", input_width=4, weight_width=4),
              Activation: None, BatchNorm: bypass_bn};  first conv + BN and the first residual conv + BN excluded;
              net.quantize_input(enable=False); qparams_init(net)
  cell 13     SoftmaxCrossEntropyLoss, Trainer(net.collect_params(), 'adam', {'learning_rate': 1e-6})
  cell 15     with autograd.record(): outputs = net(X); loss = loss_func(outputs, y)
              net.update_ema(); loss.backward(); trainer.step(batch, ignore_stale_grad=True)
              at `offline_at`: net.quantize_input(enable=True, online=False)

There is no network here, so the model starts from random weights and the data is synthetic (normalised-CIFAR-shaped
N(0,1) batches generated on the device, or the synthetic `CIFAR10` dataset of the facade with --dataset): the script shows
the loop and measures its throughput, it does not reproduce the notebook's accuracy.  Needs an MI355X (no CPU fallback).

  python examples/qat_finetune.py --steps 60 --offline-at 40
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from quantization.mxnet_amd import mx, ops  # noqa: E402
from quantization.mxnet_amd.mx import autograd, gluon, gpu  # noqa: E402
from quantization.mxnet_amd.mx.gluon import nn  # noqa: E402
from quantization.mxnet_amd.mx.gluon.model_zoo import get_model  # noqa: E402
from quantization.mxnet_amd.quantize import convert  # noqa: E402
from quantization.mxnet_amd.quantize.initialize import qparams_init  # noqa: E402


def build(model, ctx, width):
    np.random.seed(7)
    torch.manual_seed(7)
    net = get_model(model, pretrained=False, classes=10)
    converter = {
        nn.Conv2D: convert.gen_conv2d_converter(quant_type="channel", fake_bn=True, input_width=width, weight_width=width),
        nn.Dense: convert.gen_dense_converter(quant_type="channel", input_width=width, weight_width=width),
        nn.Activation: None,
        nn.BatchNorm: convert.bypass_bn,
    }
    exclude = [net.features[0], net.features[1], net.features[2][0].body[0], net.features[2][0].body[1]]
    convert.convert_model(net, exclude=exclude, convert_fn=converter)
    net.quantize_input(enable=False)
    qparams_init(net)
    net.collect_params().reset_ctx(ctx)                              # cell 7
    return net


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--model", default="cifar_resnet56_v1")
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=5, help="untimed steps before the throughput measurement")
    ap.add_argument("--batch-size", type=int, default=64)            # Config.train_batch_size
    ap.add_argument("--lr", type=float, default=1e-6)                # Config.lr
    ap.add_argument("--offline-at", type=int, default=40, help="switch the input quantisers on, offline (Config.offline_at)")
    ap.add_argument("--width", type=int, default=4)
    ap.add_argument("--graph", type=int, default=0,
                    help="1: capture the whole step (forward + backward + Adam + update_ema) into a hipGraph once per input-"
                         "quantisation mode and replay it (the step is ~3700 launches: launching them from Python is what a "
                         "step costs); the optimiser then keeps its step counter on the device")
    ap.add_argument("--dataset", action="store_true", help="iterate the facade's synthetic CIFAR10 dataset instead of "
                                                           "on-device random batches")
    args = ap.parse_args()
    if not torch.cuda.is_available():
        raise SystemExit("qat_finetune.py needs an MI355X (the fake-quant path has no CPU fallback)")
    ctx = gpu(0)
    dev = ctx.torch_device
    net = build(args.model, ctx, args.width)
    loss_func = gluon.loss.SoftmaxCrossEntropyLoss()
    trainer = gluon.Trainer(net.collect_params(), "adam", {"learning_rate": args.lr, "capturable": bool(args.graph)})

    def batches():
        if args.dataset:
            from quantization.mxnet_amd.mx.gluon.data import DataLoader, vision
            T = vision.transforms
            tf = T.Compose([T.ToTensor(), T.Normalize([0.4914, 0.4822, 0.4465], [0.2023, 0.1994, 0.2010])])
            while True:
                for X, y in DataLoader(vision.CIFAR10(train=True).transform_first(tf), batch_size=args.batch_size,
                                       last_batch="discard"):
                    yield X.as_in_context(ctx), y.as_in_context(ctx)
        g = torch.Generator(device=dev)
        g.manual_seed(7)
        while True:
            X = torch.randn(args.batch_size, 3, 32, 32, device=dev, generator=g)
            y = torch.randint(0, 10, (args.batch_size,), device=dev, generator=g).float()
            yield mx.nd.NDArray(X), mx.nd.NDArray(y)

    def train_step(X, y):
        with autograd.record():
            outputs = net(X)
            loss = loss_func(outputs, y)
        net.update_ema()
        loss.backward()
        trainer.step(args.batch_size, ignore_stale_grad=True)        # bypassed BatchNorms never receive a gradient
        return loss._t.detach().mean()

    stream = batches()
    quantize_offline = False
    losses = []
    t0 = None
    graph, eager_left, captures = None, 2, 0
    Xs = ys = loss_s = None
    side = torch.cuda.Stream(dev) if args.graph else None
    for step in range(1, args.warmup + args.steps + 1):
        if step == args.warmup + 1:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        X, y = next(stream)
        if not args.graph:
            losses.append(train_step(X, y))
        elif graph is not None:
            Xs.copy_(X._t)
            ys.copy_(y._t)
            graph.replay()
            losses.append(loss_s.clone())
        elif eager_left > 0:                                         # the first steps of a mode: eager (lazy state, library warm-up)
            eager_left -= 1
            losses.append(train_step(X, y))
        else:                                                        # capture this step on static buffers; it runs when replayed
            Xs, ys = X._t.clone(), y._t.clone()
            side.wait_stream(torch.cuda.current_stream(dev))
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):
                loss_s = train_step(mx.nd.NDArray(Xs), mx.nd.NDArray(ys))
            torch.cuda.current_stream(dev).wait_stream(side)
            captures += 1
            graph.replay()
            losses.append(loss_s.clone())
        if not quantize_offline and step - args.warmup >= args.offline_at:
            net.quantize_input(enable=True, online=False)
            quantize_offline = True
            graph, eager_left = None, 2                              # another forward: capture again
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ls = torch.stack(losses).cpu().numpy()
    blocks = net.collect_quantized_blocks()
    print(json.dumps({
        "what": "QAT fine-tuning step (forward + backward + Adam) of the reference notebook's configuration",
        "model": args.model, "quantised_blocks": len(blocks), "width": args.width, "batch_size": args.batch_size,
        "steps": args.steps, "offline_at": args.offline_at, "graph": bool(args.graph), "captures": captures, "images_per_sec": round(args.steps * args.batch_size / elapsed, 1),
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "loss_first": float(ls[args.warmup]),
        "loss_last": float(ls[-1]), "loss_finite": bool(np.isfinite(ls).all()),
        "input_max_range": [float(min(b.input_max.data().asnumpy()[0] for b in blocks)),
                            float(max(b.input_max.data().asnumpy()[0] for b in blocks))],
        "data": "synthetic", "device": torch.cuda.get_device_name(0)}))


if __name__ == "__main__":
    main()
