#!/usr/bin/env python3
"""The CLI's replayed evaluation of MobileNetV2 against bare replays of the same graphs: `evaluate` at two loader lengths (the
MARGINAL rate is the steady state; the rest is what a pass pays once: eager first batches, captures), then the same step
captured per (lane, batch) and replayed under loops of increasing resemblance to `evaluate`'s - bare round-robin replays;
+ wait_stream / record_stream per batch; + both context managers.  r4: evaluate 127.3 k images/s marginal (fixed cost 23 ms per
pass, ~100 ms more in a process's first pass), bare replays 126.8 k; one lane 85 k; a host thread per lane changed nothing."""
import importlib.util
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("FQ_SYNTH_VAL_IMAGES", "25600")
os.environ.setdefault("FQ_SYNTH_TRAIN_PER_CLASS", "1")
spec = importlib.util.spec_from_file_location("fq_cli", os.path.join(ROOT, "examples", "simulate_quantization.py"))
cli = importlib.util.module_from_spec(spec)
spec.loader.exec_module(cli)
from quantization.mxnet_amd import mx, ops  # noqa: E402

opt = cli.parse_args(["--model", "mobilenetv2_1.0", "--use-gpu", "0", "--pretrained", "false", "--synthetic-on-device",
                      "--quant-type", "channel", "--weight-bits-width", "4", "--quantize-input-offline", "--calib-epoch", "1",
                      "--num-sample", "1"])
ctx = mx.gpu(0)
dev = ctx.torch_device
sim = cli.Simulation(opt, ctx, 0, 1)
np.random.seed(opt.fixed_random_seed)
sim.build_net()
sim.quantise_net()
sim.make_loaders()
sim.calibrate_naive()
sim.final_evaluation(online=False)                # the CLI's own figure (200 batches)
print("CLI evaluate: %.0f images/s" % cli.evaluate.last_images_per_sec)
for streams, graph in ((3, True), (3, False), (1, True)):
    t = {}
    for images in (25600, 128000):
        os.environ["FQ_SYNTH_VAL_IMAGES"] = str(images)
        sim.make_loaders()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        cli.evaluate(sim.net, 1000, sim.eval_loader, ctx, streams=streams, graph=graph, tqdm_desc="probe")
        t[images] = time.perf_counter() - t0
    print("evaluate(streams=%d, graph=%d): %d images in %.3f s, %d in %.3f s -> marginal %.0f images/s, fixed cost %.0f ms"
          % (streams, graph, 25600, t[25600], 128000, t[128000], (128000 - 25600) / (t[128000] - t[25600]),
             (t[25600] - 25600 * (t[128000] - t[25600]) / (128000 - 25600)) * 1e3))
net = sim.net
counters = torch.zeros(2002, device=dev)
batches = []
for i, b in enumerate(sim.eval_loader):
    batches.append(b)
    if i == 5:
        break
lanes = [torch.cuda.Stream(dev) for _ in range(3)]


def step(x, y):
    out = net(mx.nd.NDArray(x))
    ops.eval_counters(out._t, y, counters)


graphs = {}
with ops.batches_in_flight():
    for li, s in enumerate(lanes):
        with torch.cuda.stream(s):
            step(batches[li][0]._t, batches[li][1]._t.long())
    torch.cuda.synchronize()
    for i, (X, y) in enumerate(batches):
        li = i % 3
        yl = y._t.long()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(lanes[li]):
            g.capture_begin(capture_error_mode="thread_local")
            step(X._t, yl)
            g.capture_end()
        graphs[i] = (g, li, X, yl)
torch.cuda.synchronize()
N = 600
producer = torch.cuda.current_stream(dev)


def run(mode):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(N):
        g, li, X, yl = graphs[i % 6]
        s = lanes[li]
        if mode >= 1:
            s.wait_stream(producer)
            X._t.record_stream(s)
        if mode >= 2:
            with torch.cuda.stream(s), ops.batches_in_flight():
                g.replay()
        else:
            with torch.cuda.stream(s):
                g.replay()
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    total = time.perf_counter() - t0
    print("mode %d: host %.3f ms/batch, total %.3f ms/batch = %.0f images/s" % (mode, host / N * 1e3, total / N * 1e3,
                                                                             N * 128 / total))


for m in (0, 1, 2, 0):
    run(m)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(N):
    g, li, X, yl = graphs[(i % 2) * 3]
    with torch.cuda.stream(lanes[0]):
        g.replay()
torch.cuda.synchronize()
print("one lane: %.0f images/s" % (N * 128 / (time.perf_counter() - t0)))
