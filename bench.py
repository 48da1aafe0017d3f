#!/usr/bin/env python3
"""bench.py — the driver's benchmark contract.

Workload (BASELINE.json configs[1]): int8-simulated **mobilenet1.0**, ImageNet-shaped input (128, 3, 224, 224) per GPU,
per-layer W8A8, ONLINE input quantisation, first conv + BN excluded — i.e. what
`examples/simulate_quantization.py --model=mobilenet1.0` evaluates (reference: simulate_quantization.py:346-348 ->
evaluate :122-148).  One step = one forward of the converted net over one resident batch + the on-device accuracy
counters.  Random-init (seed 7) weights and synthetic N(0,1) images: no network for checkpoints or ImageNet.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Prints ONE JSON line: images/sec for the whole job, plus
  "roofline":     the dominant fake-quant kernel (`act_apply_kernel`, online): algorithmic bytes (8 B/elem: read x, write
                  y — SURVEY.md 8d) / its HIP-event-timed duration inside the timed region, against 8 TB/s;
  "cpu_baseline": the same workload on the host cores through the oracle (a "port": the reference's MXNet path
                  cannot run here), on a bounded sample of images.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E peak (/opt/skills/guides/MI355X_MICROARCH.md)


def build_net(model, classes, ctx, seed=7, fuse=True):
    from quantization.mxnet_amd.mx.gluon import nn
    from quantization.mxnet_amd.mx.gluon.model_zoo import get_model
    from quantization.mxnet_amd.quantize import convert
    from quantization.mxnet_amd.quantize.initialize import qparams_init
    np.random.seed(seed)
    net = get_model(model, pretrained=False, classes=classes)
    convert_fn = {
        nn.Conv2D: convert.gen_conv2d_converter(quantize_input=True, weight_width=8, input_width=8,
                                                input_signed=False, quant_type="layer"),
        nn.Dense: convert.gen_dense_converter(quantize_input=True, weight_width=8, input_width=8,
                                              input_signed=False, quant_type="layer"),
        nn.Activation: None, nn.BatchNorm: None}
    exclude = [net.features[0], net.features[1]]
    convert.convert_model(net, exclude=exclude, convert_fn=convert_fn)
    qparams_init(net)
    net.collect_params().reset_ctx(ctx)
    net.fix_params()
    net.quantize_input(enable=True, online=True)
    if fuse and ctx.device_type == "gpu":
        from quantization.mxnet_amd.quantize import fuse as _fuse
        _fuse.fuse_inference(net)
    return net


def headline_tensor(dev, ops):
    """BASELINE.json's second figure: achieved HBM bandwidth of the fused online fake-quant (statistic pass + apply pass,
    12 algorithmic B/elem: two reads and one write, no credit for Infinity-Cache hits) on the largest
    MobileNet activation, (128, 64, 112, 112) fp32 = 411 MB, measured with HIP events on the launch stream after the
    benchmark's timed region (median of 20).  Reported beside `roofline`, which describes the dominant kernel of the step."""
    import torch
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    x = torch.relu(torch.randn(128, 64, 112, 112, device=dev, generator=g)) * 2.0
    out = torch.empty_like(x)
    cur = torch.zeros(1, device=dev)

    def run():
        ops.fake_quant_online(x, 8, 0, out=out, cur_out=cur)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for a, b in evs:
        a.record()
        run()
        b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in evs)[10]
    nbytes = 12.0 * x.numel()
    return {"what": "online fake-quant (absmax_per_sample + act_apply kernels) on (128,64,112,112) fp32, 12 B/elem",
            "bound": "hbm", "achieved": round(nbytes / (ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "ms": round(ms, 4)}


def cpu_baseline(model, classes, hw, sample_images, budget_s=25.0):
    """Oracle leg: identical converted net on the host, fake-quant through oracle/ (numpy restatement of the
    reference's op chain), conv/FC through torch-CPU.  Bounded sample; reports images/sec."""
    from quantization.mxnet_amd import mx
    from oracle.patch import oracle_ops
    net = build_net(model, classes, mx.cpu())
    rng = np.random.default_rng(7)
    done, t_total = 0, 0.0
    with oracle_ops():
        x = mx.nd.array(rng.standard_normal((2, 3, hw, hw)).astype(np.float32))
        net(x)                                      # first forward freezes the weights (fixed_params 0 -> 1)
        bs = 4
        while done < sample_images and t_total < budget_s:
            x = mx.nd.array(rng.standard_normal((bs, 3, hw, hw)).astype(np.float32))
            t0 = time.perf_counter()
            net(x)
            t_total += time.perf_counter() - t0
            done += bs
    return {"value": round(done / t_total, 3), "unit": "images/sec", "cores": int(torch.get_num_threads()),
            "kind": "port",
            "sample": "%d images (batches of 4) of the same int8-sim %s forward; fake-quant = numpy oracle "
                      "(1 thread), conv/FC = torch-CPU (%d threads); %.1f s" % (done, model, torch.get_num_threads(),
                                                                                t_total)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--model", default="mobilenet1.0")
    ap.add_argument("--batch-size", type=int, default=128, help="per GPU (CLI default, simulate_quantization.py:81)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--event-every", type=int, default=10,
                    help="bracket the library's kernels with HIP events in every n-th timed step (default 10)")
    ap.add_argument("--no-kernel-events", action="store_true",
                    help="do not bracket kernels with HIP events in the timed region (roofline fields become empty)")
    ap.add_argument("--no-headline", action="store_true",
                    help="skip the stand-alone fake-quant measurement on the 411 MB headline tensor (extra JSON object)")
    ap.add_argument("--no-fuse", action="store_true",
                    help="keep BatchNorm / ReLU as separate torch ops (no quantize.fuse.fuse_inference)")
    ap.add_argument("--autotune", action="store_true",
                    help="MIOpen find/benchmark mode (measured: no gain for these shapes, +60 s of search)")
    ap.add_argument("--cpu-sample", type=int, default=64)
    ap.add_argument("--graph", type=int, default=int(os.environ.get("FQ_BENCH_GRAPH", "0")),
                    help="replay the step from a hipGraph (no per-kernel events then; roofline measured in extra steps)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
    distributed = world > 1
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the fake-quant path has no CPU fallback)")
    # test-only knobs for exercising the N > 1 path on a one-GPU box: every rank on device 0, gloo instead of RCCL
    share_gpu = os.environ.get("FQ_BENCH_SHARE_GPU", "0") == "1"
    backend = os.environ.get("FQ_BENCH_BACKEND", "nccl")
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from quantization.mxnet_amd import mx, ops
    # MXNet autotunes convolutions by default (the reference only offers --disable-cudnn-autotune,
    # simulate_quantization.py:184-186); the torch/MIOpen equivalent is benchmark mode.
    torch.backends.cudnn.benchmark = bool(args.autotune)
    classes = 10 if args.model.startswith("cifar") else 1000
    hw = 32 if args.model.startswith("cifar") else 224
    ctx = mx.gpu(local_rank)
    net = build_net(args.model, classes, ctx, fuse=not args.no_fuse)
    nblocks = len(net.collect_quantized_blocks())

    torch.manual_seed(7 + rank)
    X = mx.nd.NDArray(torch.randn(args.batch_size, 3, hw, hw, device=dev))
    y = torch.randint(0, classes, (args.batch_size,), device=dev)
    counters = torch.zeros(2 + 2 * classes, dtype=torch.float32, device=dev)   # n_correct, total, correct[c], label[c]

    def step():
        out = net(X)._t
        ops.eval_counters(out, y, counters)           # the eval loop's argmax + counters (fq_eval_counters, one launch)
        return out

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, 1) if args.graph else args.warmup):
        step()
    torch.cuda.synchronize()

    graph = None
    if args.graph:
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            step()
        graph.replay()
        torch.cuda.synchronize()

    # Kernel events are SAMPLED inside the timed region (every `event_every`-th step): bracketing all ~45 launches of a
    # step costs ~35 % of THAT step (two marker packets per launch; measured 78.4 k images/s with every 4th step
    # bracketed against 85.4 k with none), and the cost is charged to `value`.  Every 10th step keeps >= 3 profiled steps
    # (~40 launches of each producer family) at the default --steps 30.
    event_every = 0 if (args.no_kernel_events or graph is not None) else max(1, args.event_every)
    profiled_steps = 0
    if event_every:
        ops.profile_reset()
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        if graph is not None:
            graph.replay()
        else:
            on = bool(event_every) and (i % event_every == 0)
            if on:
                ops.profile_enable(True)
                profiled_steps += 1
            step()
            if on:
                ops.profile_enable(False)
    barrier()
    elapsed = time.perf_counter() - t0
    if graph is None:
        prof = ops.profile_read()
    else:
        ops.profile_reset()
        ops.profile_enable(True)
        profiled_steps = min(args.steps, 10)
        for _ in range(profiled_steps):
            step()
        torch.cuda.synchronize()
        ops.profile_enable(False)
        prof = ops.profile_read()
    ops.profile_reset()
    # A bracketing event pair adds a fixed cost to every launch it times (two marker packets + dispatch latency).
    # It is measured live around a one-element kernel whose own duration is known from the rocprofv3 trace, and removed.
    null_kernel_us = 3.4          # fill_kernel average in profiles/r1_bench_kernel_stats.csv (rocprofv3)
    ev_overhead_ms = max(ops.profile_event_overhead_ms(dev) - null_kernel_us * 1e-3, 0.0)

    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        dist.all_reduce(counters, op=dist.ReduceOp.SUM)      # the eval counters of simulate_quantization.py:123-147

    if rank == 0:
        images = world * args.batch_size * args.steps
        # every HIP kernel family of the fake-quant path, timed by HIP events inside the timed region
        names = {"apply_online": "act_apply_kernel<ONLINE> (fake-quant apply pass, 8 B/elem)",
                 "apply_offline": "act_apply_kernel<OFFLINE> (8 B/elem)",
                 "stat": "absmax_per_sample_kernel (statistic pass, 4 B/elem)",
                 "dwconv": "dwconv3x3_*_kernel (depthwise 3x3 with fake-quant on load + BN/ReLU/statistic on store, "
                           "4 B/in-elem + 4 B/out-elem)",
                 "bn_act": "stem_conv3x3s2_kernel / bn_act_stat_kernel (un-quantised first conv + BN + ReLU + statistic, "
                           "4 B/in-elem + 4 B/out-elem; lone BN + ReLU + statistic passes, 8 B/elem)",
                 "pwconv": "pwconv_stream_kernel / pwconv_chunk_kernel (+ quant_transpose_i8 + pwconv_i8 for K=1024): 1x1 "
                           "conv on int8 codes, fake-quant on load, exact int32 MFMA sums, BN/ReLU/statistic on store; "
                           "4 B/in-elem + 4 B/out-elem",
                 "weight": "weight fake-quant kernels (8 B/elem)", "histogram": "histogram_kernel (4 B/elem)"}
        kernels = {}
        for key, rec in prof.items():
            if not rec["launches"]:
                continue
            ms = max(rec["ms"] - ev_overhead_ms * rec["launches"], 1e-9)
            gbs = rec["bytes"] / (ms * 1e-3) / 1e9
            kernels[key] = {"kernel": names.get(key, key), "achieved": round(gbs, 1),
                            "frac": round(gbs / HBM_PEAK_GBS, 4), "launches": rec["launches"],
                            "avg_launch_us": round(ms * 1e3 / rec["launches"], 3),
                            "avg_launch_us_raw_events": round(rec["ms"] * 1e3 / rec["launches"], 3),
                            "algorithmic_bytes_per_launch": round(rec["bytes"] / rec["launches"], 1),
                            "ms_per_step": round(ms / max(profiled_steps, 1), 4)}
        dominant = max(kernels, key=lambda k: kernels[k]["ms_per_step"]) if kernels else None
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if dominant and os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("kernels", {}).get(dominant, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        dk = kernels.get(dominant, {"achieved": 0.0, "frac": 0.0, "kernel": None})
        line = {
            "metric": "images/sec int8-sim %s (per-layer W8A8, online input quant)"
                      % ("MobileNet1.0" if args.model == "mobilenet1.0" else args.model),
            "value": round(images / elapsed, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s ImageNet-shaped (%d,3,%d,%d)/GPU, per-layer W8A8, online input quant, first "
                                   "conv excluded, %d fake-quantised blocks, eval forward + accuracy counters"
                                   % (args.model, args.batch_size, hw, hw, nblocks),
                       "global_batch": world * args.batch_size, "parallelism": "dp%d (replicated weights, sharded "
                       "batch, no data-path collective; counters all-reduced once)" % world,
                       "hipgraph": bool(args.graph), "fused_producers": not args.no_fuse},
            "roofline": {"bound": "hbm", "kernel": dk["kernel"], "achieved": dk["achieved"], "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": dk["frac"], "traffic": traffic,
                         "dominant_by": "largest HIP-event time per step among this library's kernels",
                         "event_sampling": "HIP events bracket every library launch in %d of the %d timed steps"
                                           % (profiled_steps, args.steps),
                         "event_pair_overhead_us_removed": round(ev_overhead_ms * 1e3, 3), "kernels": kernels},
        }
        if world == 1 and not args.no_headline:
            line["headline_tensor"] = headline_tensor(dev, ops)
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(args.model, classes, hw, args.cpu_sample)
        elif world == 1:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
