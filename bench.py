#!/usr/bin/env python3
"""bench.py — the driver's benchmark contract.

Default workload (BASELINE.json configs[1], the configuration the metric is quoted on): int8-simulated **mobilenet1.0**,
ImageNet-shaped input (128, 3, 224, 224) per GPU, per-layer W8A8, ONLINE input quantisation, first conv + BN excluded —
what `examples/simulate_quantization.py --model=mobilenet1.0` evaluates (reference: its evaluate(), :122-148, called at
:346-348).  One step = one forward of the converted net over one resident batch + the on-device accuracy counters.
Random-init (seed 7) weights and synthetic N(0,1) images: there is no network for checkpoints or ImageNet.  The other
BASELINE configurations are reachable with flags (they are parity-test cases; their lines are kept under profiles/):

    python bench.py --gpus N --steps K --warmup W
    python bench.py --model resnet50_v1 --quant-type channel [--offline] [--wino F43]
    python bench.py --model mobilenetv2_1.0 --quant-type channel --weight-bits 4 --offline
    python bench.py --phase calib-naive --model mobilenetv2_1.0 --quant-type channel --weight-bits 4    (config 4's calibration:
                    a step = forward with ONLINE scales and the weights re-quantised + net.update_ema())
    python bench.py --phase calib-kl --model resnet50_v1 --quant-type channel                           (config 3's calibration:
                    a step = forward with quantisation disabled + one 2048-bin histogram per quantised block;
                    the threshold search over all layers is timed once, after the steps)
    N > 1: either  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...  (one rank
    per GPU; RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment), or plainly  python bench.py --gpus N:
    with no WORLD_SIZE in the environment the process — before it touches any GPU — starts that launcher itself
    (fresh children, 127.0.0.1 rendezvous on a free port), passes rank 0's JSON line through and exits with the
    children's status.

Prints ONE JSON line: images/sec for the whole job, plus
  "roofline":     the kernel family with the largest share of the step: algorithmic bytes per launch / its
                  HIP-event-timed duration inside the timed region, against 8 TB/s; every family under "kernels", and the
                  whole step's algorithmic bytes over its wall time under "whole_step";
  "cpu_baseline": the same workload on the host cores — the converted net with the C++/OpenMP restatement of the
                  reference's arithmetic (oracle/libfq_host.so) doing the fake-quant on ALL cores and torch-CPU the
                  convolutions — on a bounded sample, with the fake-quant-only figures (the reference's pass-by-pass op
                  chain and the fused form, all cores; the numpy chain on one thread) beside it.
"""
import argparse
import json
import os
import sys
import time

# (the package sets this at import too - quantization/mxnet_amd/__init__.py says why - but this script asks torch for the GPU
# before it imports the package, and the variable is read when the HIP runtime initialises)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E peak (/opt/skills/guides/MI355X_MICROARCH.md)
DEFAULT_STEPS = 500         # ~0.6 s of timed region for the default workload (1.16-1.21 ms per step)

KERNEL_NAMES = {
    "apply_online": "act_apply_kernel<ONLINE> (fake-quant apply pass, 8 B/elem)",
    "apply_offline": "act_apply_kernel<OFFLINE> (8 B/elem)",
    "stat": "absmax_per_sample_kernel (statistic pass, 4 B/elem)",
    "dwconv": "dwconv3x3_{cols4,flat,planes,cols,}_kernel (depthwise 3x3 with fake-quant on load + BN/ReLU/statistic on "
              "store, 4 B/in-elem + 4 B/out-elem)",
    "bn_act": "bn_act_stat_kernel (BatchNorm + ReLU + statistic in one pass, 8 B/elem)",
    "stem": "stem3_rows_kernel / stem7_pool_lds_kernel / stem_mfma_kernel (un-quantised first conv on the fp32 matrix cores + BN + ReLU + statistic, 4 B/in-elem + "
            "4 B/out-elem)",
    "pool": "gap_stat_kernel (global average pool + statistic, 4 B/in-elem + 4 B/out-elem)",
    "pwconv": "pwconv_{stream,sample,split}_kernel (1x1 conv on int8 codes: fake-quant on load, exact int32 MFMA sums, "
              "BN/ReLU/statistic on store; 4 B/in-elem + 4 B/out-elem)",
    "conv3x3": "conv3x3_i8_kernel (dense 3x3 conv on int8 codes: implicit GEMM over (tap, ci), fake-quant on load, exact "
               "int32 MFMA sums, BN/ReLU/statistic on store; 4 B/in-elem + 4 B/out-elem)",
    "dense": "pwconv_rows_kernel (the classifier on the int8 codes + the evaluation counters of its logits; 4 B/in-elem + "
             "4 B/out-elem, latency-bound)",
    "weight": "weight fake-quant kernels (8 B/elem)",
    "histogram": "histogram_kernel (4 B/elem)",
    "global_max": "minmax_kernel (4 B/elem)",
}


def is_qconv_model(model):
    """`quantized_mobilenet1.0` & co: the net the reference builds from nn.Conv2D(quantized=True) blocks
    (tests/models/quantized_mobilenet.py) - real int8 convolutions behind the stand-alone block, not simulated quantisation."""
    return model.startswith("quantized_mobilenet")


def build_qconv_net(model, classes, ctx, seed=7, fuse=True):
    from quantization.mxnet_amd import mx
    from quantization.mxnet_amd.nn import quantized_mobilenet as QM, fuse as qfuse
    np.random.seed(seed)
    v2 = model.startswith("quantized_mobilenetv2_")
    mult = float(model[len("quantized_mobilenetv2_" if v2 else "quantized_mobilenet"):] or 1.0)
    net = (QM.MobileNetV2 if v2 else QM.MobileNet)(mult, classes=classes)
    net.initialize(mx.init.Xavier(magnitude=2.0), ctx=ctx)
    net.collect_quantized_blocks = lambda: [b for b in _blocks_of(net) if getattr(b, "_quantized", False)]
    if fuse and ctx.device_type == "gpu":
        qfuse.fuse_inference(net)
    return net


def _blocks_of(net):
    found = []
    net.apply(found.append)
    return found


def build_net(model, classes, ctx, seed=7, fuse=True, quant_type="layer", weight_bits=8, input_bits=8, signed=False,
              wino="none", freeze=True):
    if is_qconv_model(model):
        return build_qconv_net(model, classes, ctx, seed, fuse)
    from quantization.mxnet_amd.mx.gluon import nn
    from quantization.mxnet_amd.mx.gluon.model_zoo import get_model
    from quantization.mxnet_amd.quantize import convert
    from quantization.mxnet_amd.quantize.initialize import qparams_init
    np.random.seed(seed)
    net = get_model(model, pretrained=False, classes=classes)
    common = dict(quantize_input=True, weight_width=weight_bits, input_width=input_bits, input_signed=signed,
                  quant_type=quant_type)
    convert_fn = {nn.Conv2D: convert.gen_conv2d_converter(wino_quantize=wino, **common),
                  nn.Dense: convert.gen_dense_converter(**common), nn.Activation: None, nn.BatchNorm: None}
    exclude = [net.features[0], net.features[1]]                 # --exclude-first-conv=true, the CLI default
    if model.startswith("mobilenetv2_"):
        exclude.append(net.output[0])
    if model.startswith("cifar_resnet"):
        exclude += [net.features[2][0].body[0], net.features[2][0].body[1]]
    convert.convert_model(net, exclude=exclude, convert_fn=convert_fn)
    qparams_init(net)
    net.collect_params().reset_ctx(ctx)
    if freeze:
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
    # the apply pass alone under a stored threshold (8 B/elem: one read, one write of tensors the Infinity Cache cannot hold):
    # what a streaming kernel reaches on THIS box - the ceiling the whole step's bytes are priced against beside the 8 TB/s
    thr = torch.full((1,), 3.0, device=dev)

    def apply_only():
        ops.fake_quant_offline(x, thr, 8, 0, out=out, want_stat=False)
    for _ in range(3):
        apply_only()
    torch.cuda.synchronize()
    for a, b in evs:
        a.record()
        apply_only()
        b.record()
    torch.cuda.synchronize()
    ms8 = sorted(a.elapsed_time(b) for a, b in evs)[10]
    return {"what": "online fake-quant (absmax_per_sample + act_apply kernels) on (128,64,112,112) fp32, 12 B/elem",
            "bound": "hbm", "achieved": round(nbytes / (ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "ms": round(ms, 4),
            "apply_only": {"what": "offline apply pass alone on the same tensor, 8 B/elem (411 MB read + 411 MB written): the "
                                   "streaming rate of this box",
                           "achieved": round(8.0 * x.numel() / (ms8 * 1e-3) / 1e9, 1), "unit": "GB/s",
                           "frac": round(8.0 * x.numel() / (ms8 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "ms": round(ms8, 4)}}


# the activation tensors entering mobilenet1.0's 27 quantised blocks, per image (C, H, W)
MOBILENET_ACTS = [(32, 112, 112)] * 2 + [(64, 112, 112), (64, 56, 56)] + [(128, 56, 56)] * 3 + [(128, 28, 28)] + \
    [(256, 28, 28)] * 3 + [(256, 14, 14)] + [(512, 14, 14)] * 11 + [(512, 7, 7)] + [(1024, 7, 7)] * 2 + [(1024, 1, 1)]


def cpu_baseline(args, classes, hw, budget_s=20.0):
    """Oracle leg (test infrastructure, never the product path): (1) the same converted net on the host, fake-quant by the
    C++/OpenMP restatement on all cores, convolutions by torch-CPU — images/sec on a bounded sample; (2) the fake-quant
    work alone over the 27 MobileNet activation shapes: the reference's pass-by-pass op chain and the fused form on all
    cores, and the numpy op chain on one thread."""
    from quantization.mxnet_amd import mx
    from oracle.patch import host_ops
    from oracle import host as H
    from oracle import fq_oracle as O
    cores = os.cpu_count() or 1
    # one thread per physical core of the machine, SMT siblings left idle (measured on the 2 x 64-core / 256-thread host:
    # 256 OpenMP threads are several times SLOWER than 64-128 on these passes)
    threads = int(os.environ.get("FQ_HOST_THREADS", max(1, cores // 2)))
    torch.set_num_threads(threads)
    H.set_threads(threads)
    phase = getattr(args, "phase", "eval")
    net = build_net(args.model, classes, mx.cpu(), quant_type=args.quant_type, weight_bits=args.weight_bits,
                    input_bits=args.input_bits, signed=args.input_signed, wino=args.wino, freeze=phase == "eval")
    rng = np.random.default_rng(7)
    bs, done, t_total = (32 if phase == "eval" and not is_qconv_model(args.model) else 8), 0, 0.0
    with host_ops():
        if phase == "eval":
            net(mx.nd.array(rng.standard_normal((2, 3, hw, hw)).astype(np.float32)))   # freezes the weights (0 -> 1)
        x = mx.nd.array(rng.standard_normal((bs, 3, hw, hw)).astype(np.float32))
        if phase == "calib-kl":
            # the same calibration on the host: forward with quantisation disabled, every block's input histogrammed by the
            # C++/OpenMP twin of the device kernel (the reference: single-threaded numpy over a host copy)
            from quantization.mxnet_amd.quantize.distribution_calibrate import collect_feature_maps
            net.disable_quantize()
            collect_feature_maps(net, 2048, [(x, None)], mx.cpu())                   # warm
            while t_total < budget_s * 0.5 and done < 1024:
                t0 = time.perf_counter()
                collect_feature_maps(net, 2048, [(x, None), (x, None)], mx.cpu())
                t_total += time.perf_counter() - t0
                done += 2 * bs
        else:
            net(x)                                                                   # warm (allocator, thread pools)
            while t_total < budget_s * 0.5 and done < 4096:
                t0 = time.perf_counter()
                net(x)
                if phase == "calib-naive":
                    net.update_ema()
                t_total += time.perf_counter() - t0
                done += bs
    out = {"value": round(done / t_total, 3), "unit": "images/sec", "cores": threads, "kind": "port",
           "host_logical_cpus": cores,
           "threads": {"openmp_fake_quant": H.threads(), "torch_conv": torch.get_num_threads()},
           "sample": "%d images (batches of %d) of the same %s %s %s on the host: %s%s = C++/OpenMP "
                     "restatement of the reference's arithmetic (oracle/libfq_host.so, %d threads), %s = torch-CPU "
                     "(%d threads); %.1f s" % (done, bs, "real-int8" if is_qconv_model(args.model) else "int8-sim", args.model,
                                               {"eval": "forward", "calib-naive": "naive-EMA calibration step",
                                                "calib-kl": "KL histogram collection"}[phase],
                                               "the whole nn.Conv2D(quantized=True) block (range, codes, integer convolution, "
                                               "dequantise)" if is_qconv_model(args.model) else "fake-quant",
                                               " / histograms" if phase == "calib-kl" else "", H.threads(),
                                               "first conv / BatchNorm / ReLU / FC" if is_qconv_model(args.model)
                                               else "conv/FC", torch.get_num_threads(), t_total)}
    if phase != "eval" or is_qconv_model(args.model):
        return out
    # (2) fake-quant only, 27-layer sweep at batch 16, buffers allocated and touched before timing
    n = 16
    biggest = max(c * h * w for c, h, w in MOBILENET_ACTS) * n
    src = np.maximum(rng.standard_normal(biggest, dtype=np.float32), 0) * np.float32(1.7)
    y = np.zeros(biggest, np.float32)
    tmp = np.zeros(2 * biggest, np.float32)
    t_chain = t_fused = 0.0
    for c, h, w in MOBILENET_ACTS:
        k = n * c * h * w
        xs, ys = src[:k].reshape(n, c, h, w), y[:k].reshape(n, c, h, w)
        t0 = time.perf_counter()
        H.unfused_chain(xs, tmp=tmp, out=ys)
        t1 = time.perf_counter()
        H._call("fq_fake_quant_online_host", xs, ys, n, c * h * w, H._i(8), H._u(0), np.empty(1, np.float32), None, None,
                None)
        t2 = time.perf_counter()
        t_chain += t1 - t0
        t_fused += t2 - t1
    xs = src[:2 * 64 * 112 * 112].reshape(2, 64, 112, 112)
    t0 = time.perf_counter()
    O.unfused_reference_chain(xs)
    t_np = time.perf_counter() - t0
    per_img = sum(c * h * w for c, h, w in MOBILENET_ACTS)
    out["fake_quant_only"] = {
        "what": "online fake-quant of the 27 mobilenet1.0 activation tensors (4.99 M elements per image), batch 16",
        "reference_op_chain_all_cores_images_per_sec": round(n / t_chain, 2),
        "fused_all_cores_images_per_sec": round(n / t_fused, 2),
        "numpy_op_chain_1_thread_images_per_sec": round(xs.size / t_np / per_img, 3),
        "threads": H.threads()}
    return out


def launch_ranks(n):
    """`python bench.py --gpus N` typed without a launcher: run N ranks of this same command under
    torch.distributed.run as CHILD processes.  Nothing in this (parent) process has touched a GPU — `import torch` and
    `torch.cuda.device_count()` do not initialise HIP — and nothing that has is ever re-exec'ed."""
    import socket
    import subprocess
    share = os.environ.get("FQ_BENCH_SHARE_GPU", "0") == "1"
    have = torch.cuda.device_count()
    if have < n and not share:
        raise SystemExit("bench.py: --gpus %d asked for but this node shows %d GPU(s)" % (n, have))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def _gloo_sum(dist, t, op):
    """Test-only transport (FQ_BENCH_BACKEND=gloo, ranks sharing one GPU): stage the tensor through the host."""
    h = t.detach().cpu()
    dist.all_reduce(h, op=op)
    t.copy_(h)


def box_probe_start():
    """rocm-smi started in the background right before the timed region, so that it reads the clocks of a BUSY GPU."""
    import subprocess
    # (not under rocprofv3: its preloaded library initialises the GPU in every child process, and rocm-smi - a script behind
    # `#!/usr/bin/env python3` - would then exec from such a process, which the boxes of this pool refuse)
    if "rocprof" in os.environ.get("LD_PRELOAD", "") or any(k.startswith("ROCPROF") for k in os.environ):
        return None
    try:
        return subprocess.Popen(["rocm-smi", "--showclocks", "--showpower", "--showmaxpower", "--showperflevel", "--showtemp"],
                                stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    except Exception:
        return None


def box_info(proc=None):
    """Which box is this?  Clocks, power cap and temperature as rocm-smi printed them during the first timed block (the pool's
    boxes differ by up to 10 % on the same tree: the line should say what it ran on), the hardware-queue setting the lanes
    depend on.  Best effort: only the environment where rocm-smi is absent."""
    import re
    out = {"GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES")}
    try:
        txt = proc.communicate(timeout=30)[0] if proc is not None else ""
    except Exception:
        return out
    for key, pat in (("sclk_mhz", r"GPU\[0\].*sclk clock level.*\((\d+)Mhz\)"), ("mclk_mhz", r"GPU\[0\].*mclk clock level.*\((\d+)Mhz\)"),
                     ("fclk_mhz", r"GPU\[0\].*fclk clock level.*\((\d+)Mhz\)"),
                     ("power_cap_w", r"GPU\[0\].*Max Graphics Package Power \(W\): ([\d.]+)"),
                     ("power_w", r"GPU\[0\].*Current Socket Graphics Package Power \(W\): ([\d.]+)"),
                     ("temp_junction_c", r"GPU\[0\].*Temperature \(Sensor junction\) \(C\): ([\d.]+)"),
                     ("temp_memory_c", r"GPU\[0\].*Temperature \(Sensor memory\) \(C\): ([\d.]+)")):
        m = re.search(pat, txt)
        if m:
            out[key] = float(m.group(1))
    m = re.search(r"GPU\[0\].*Performance Level: (\w+)", txt)
    if m:
        out["perf_level"] = m.group(1)
    out["what"] = "rocm-smi started right before the first timed block (read while the GPU is busy)"
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=DEFAULT_STEPS)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--model", default="mobilenet1.0")
    ap.add_argument("--batch-size", type=int, default=128, help="per GPU (CLI default, simulate_quantization.py:81)")
    ap.add_argument("--quant-type", default="layer", choices=["layer", "group", "channel"])
    ap.add_argument("--weight-bits", type=int, default=8)
    ap.add_argument("--input-bits", type=int, default=8)
    ap.add_argument("--input-signed", action="store_true")
    ap.add_argument("--wino", default="none", choices=["none", "F23", "F43", "F63"])
    ap.add_argument("--offline", action="store_true",
                    help="offline input quantisation: two naive-EMA calibration steps fix the thresholds, then the timed "
                         "steps run with them (the evaluation phase of --quantize-input-offline)")
    ap.add_argument("--phase", default="eval", choices=["eval", "calib-naive", "calib-kl"],
                    help="eval (default): the evaluation forward; calib-naive: naive-EMA calibration steps "
                         "(simulate_quantization.py:317-334); calib-kl: histogram collection of the KL calibration "
                         "(:296-315, quantize/distribution_calibrate.py:50-114)")
    ap.add_argument("--rotate", type=int, default=4, help="distinct resident input batches cycled through the steps")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--min-region-s", type=float, default=float(os.environ.get("FQ_BENCH_MIN_REGION_S", "8.0")),
                    help="blocks of --steps steps are repeated until they cover this much time (8 s: long enough for a "
                         "once-per-few-seconds utilisation sampler beside the run to see the GPU busy); `value` is the median "
                         "block")
    ap.add_argument("--max-repeats", type=int, default=4000)
    ap.add_argument("--event-every", type=int, default=None,
                    help="bracket the library's kernels with HIP events in every n-th timed step (default 50 - calibration "
                         "phases, whose blocks are a dozen batches with a special first one: 7 -: 10 steps of the "
                         "default 500 = 130 launches per family; a bracketed step is launched eagerly and, with --streams > 1, "
                         "runs alone - at every 25th step that cost `value` 4 %%: 126 k against 132 k images/s without events).  "
                         "Only every third block (1, 4, 7 ...; FQ_BENCH_EVENT_BLOCK_EVERY, 1 = every block) carries bracketed "
                         "steps, so the median block has none; a run that needs a single block gets a second one for them "
                         "(one more than --max-repeats 1 allows)")
    ap.add_argument("--no-kernel-events", action="store_true",
                    help="do not bracket kernels with HIP events in the timed region (roofline fields become empty)")
    ap.add_argument("--no-headline", action="store_true",
                    help="skip the stand-alone fake-quant measurement on the 411 MB headline tensor (extra JSON object)")
    ap.add_argument("--no-fuse", action="store_true",
                    help="keep BatchNorm / ReLU as separate torch ops (no quantize.fuse.fuse_inference)")
    ap.add_argument("--autotune", action="store_true",
                    help="MIOpen find/benchmark mode (measured: no gain for these shapes, +60 s of search)")
    ap.add_argument("--graph", type=int, default=int(os.environ.get("FQ_BENCH_GRAPH", "1")),
                    help="1 (default): replay the evaluation step from hipGraphs, one per (stream, resident input batch); the "
                         "steps that carry kernel events still run eagerly.  The host needs ~1.05 ms to launch the ~32 kernels "
                         "of a step from Python - as long as the GPU needs for them: on ONE stream replay changes nothing (the "
                         "host runs ahead of a GPU-bound stream: 108.0 vs 109.3 k images/s), with several batches in flight the "
                         "eager host becomes the limit (100.7 k on a box with a slow host) and replay lifts it (127.7-128.8 k in "
                         "the same call).  0: every step launched from the host")
    ap.add_argument("--streams", type=int, default=int(os.environ.get("FQ_BENCH_STREAMS", "0")),
                    help="evaluation steps in flight: step i runs on HIP stream i %% S (same net; per-forward device state is "
                         "kept per stream), so that the ramp and the tail of one batch's ~30 kernels fill with the other batch's work "
                         "(independent batches; every step's kernels, results and counters are what they are with S = 1).  "
                         "1: one stream, the figure of rounds 1-3; the line reports that too (`single_stream`).  0 (default): 4 where the "
                         "layers hand integer codes over (--offline: short, vector-unit-bound kernels; 4 against 3 +2.6 %% on "
                         "MobileNetV2 W4, 5 and 6 lose 8-12 %%), else 3 (fp32 tensors between the layers: 4 against 3 was +0.6 / "
                         "+0.8 %% on two boxes and -1.1 %% (sd 0.05) on a third; profiles/r5_lanes4_ab.txt)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus))         # decided before any GPU call; children do the work
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s)" % (args.gpus, world))
    distributed = world > 1
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the fake-quant path has no CPU fallback)")
    # test-only knobs for exercising the N > 1 path on a one-GPU box: every rank on device 0, optionally gloo
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
    calib = args.phase != "eval"
    qconv = is_qconv_model(args.model)
    if qconv and (calib or args.offline or args.quant_type != "layer" or args.wino != "none"):
        raise SystemExit("bench.py: %s is built from nn.Conv2D(quantized=True) blocks (per-tensor ranges taken from every "
                         "batch): evaluation phase only, no calibration / threshold / Winograd options" % args.model)
    if calib and args.offline:
        raise SystemExit("bench.py: --offline describes the evaluation phase; a calibration phase produces the thresholds")
    torch.manual_seed(7 + rank)
    rotate = max(1, args.rotate)
    batches = [mx.nd.NDArray(torch.randn(args.batch_size, 3, hw, hw, device=dev)) for _ in range(rotate)]
    labels = [torch.randint(0, classes, (args.batch_size,), device=dev) for _ in range(rotate)]
    counters = torch.zeros(2 + 2 * classes, dtype=torch.float32, device=dev)   # n_correct, total, correct[c], label[c]

    def prepare_net():
        """One replica of the net in the state the timed steps need (same seed: the same weights in every replica)."""
        net = build_net(args.model, classes, ctx, fuse=not args.no_fuse and not args.offline, quant_type=args.quant_type,
                        weight_bits=args.weight_bits, input_bits=args.input_bits, signed=args.input_signed, wino=args.wino,
                        freeze=not args.offline and not calib)
        if args.offline:
            for i in range(2):                            # naive-EMA calibration (simulate_quantization.py:320-323)
                net(batches[i % rotate])
                net.update_ema()
            net.fix_params()
            net.quantize_input(enable=True, online=False)
            if not args.no_fuse:
                net(batches[0])                           # the freezing forward
                from quantization.mxnet_amd.quantize import fuse as _fuse
                _fuse.fuse_inference(net)
        return net

    # Steps in flight (--streams): evaluation only, eager launches only.  ONE net: a forward on a side stream keeps its
    # per-forward device state (statistic arena, batch-statistic slots, workspaces) per stream (quantize/fuse.py,
    # quantize/convert/_blocks.py: scalar_slot), so forwards of independent batches may overlap on the device
    if args.streams <= 0:
        args.streams = 4 if args.offline else 3
    n_streams = args.streams if args.phase == "eval" else 1
    net = prepare_net()
    from quantization.mxnet_amd.quantize import fuse as _fuse_mod
    lib_gemm = _fuse_mod.library_gemm_blocks(net)
    if n_streams > 1 and lib_gemm:
        # (Dense layers through the tensor library - vgg: one batch at a time, quantize.fuse.library_gemm_blocks says why)
        n_streams = args.streams = 1
    nets = [net] * n_streams
    streams = [torch.cuda.Stream(dev) for _ in range(n_streams)] if n_streams > 1 else [None]
    if n_streams > 1:
        # the explicit declaration the blocks ask for before a forward on a non-default stream may keep its batch statistic
        # to itself (quantize/convert/_blocks.py: _stream_of); held for the rest of the process: evaluation only
        ops.batches_in_flight().__enter__()
    nblocks = len(net.collect_quantized_blocks())
    from quantization.mxnet_amd import dist as fqdist
    if args.phase == "calib-naive" and distributed:
        # ONE all-reduce of L + 1 doubles per calibration step (dist.py; north_star's collective)
        fqdist.attach_calibration_sync(net, args.batch_size)
    if args.phase == "calib-kl":
        net.disable_quantize()                        # fp32 inputs and weights while collecting (:298)
    kl_extra = {}
    from quantization.mxnet_amd.quantize import fuse as _fuse_mod
    use_head = args.phase == "eval" and not args.no_fuse and os.environ.get("FQ_BENCH_HEAD", "1") != "0"
    heads = [_fuse_mod.eval_head(net, counters) if use_head else None] * n_streams

    def eval_step(i, lane):
        head = heads[lane]
        if head is not None:
            head.labels = labels[i % rotate]          # the classifier's launch counts as well (fq_dense_i8_eval)
        out = nets[lane](batches[i % rotate])._t
        if head is None or not head.take():
            ops.eval_counters(out, labels[i % rotate], counters)      # the eval loop's argmax + counters, one launch
        return out

    def step(i, lane=0):
        if args.phase == "calib-naive":               # evaluate(..., update_ema=True): forward, then the EMA of the thresholds
            out = net(batches[i % rotate])._t
            net.update_ema()
            return out
        if streams[lane] is None:
            return eval_step(i, lane)
        with torch.cuda.stream(streams[lane]):
            return eval_step(i, lane)

    class _KLBatches(object):
        """The loader `collect_feature_maps` walks: `count` resident batches; kernel events are switched on for every
        `event_every`-th of them, exactly as in the other phases."""

        def __init__(self, first, count, sample_events):
            self.first, self.count, self.sample_events = first, count, sample_events
            self.profiled = 0

        def __len__(self):
            return self.count

        def __iter__(self):
            for i in range(self.first, self.first + self.count):
                on = self.sample_events and ((i - self.first) % self.sample_events == 0)
                ops.profile_enable(bool(on))
                self.profiled += 1 if on else 0
                yield batches[i % rotate], None
            ops.profile_enable(False)

    def kl_block(first, count, sample_events):
        """`count` calibration batches through the product's own collect_feature_maps (hooks, first-batch ranges, exact
        uint64 histograms on the device, one synchronisation and the fp32 conversion at the end)."""
        from quantization.mxnet_amd.quantize.distribution_calibrate import collect_feature_maps
        loader = _KLBatches(first, count, sample_events)
        hists, ranges = collect_feature_maps(net, 2048, loader, ctx, sync=fqdist.kl_sync if distributed else None)
        kl_extra["hists"], kl_extra["ranges"] = hists, ranges
        return loader.profiled

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    if n_streams > 1:
        torch.cuda.synchronize()                      # (the lanes are non-blocking streams: nothing orders them behind the default one)
        step(0, 0)                                    # set-up: the weight codes are made once, by the first forward ...
        torch.cuda.synchronize()
        for lane in range(1, n_streams):              # ... and every stream gets its arena, slots and workspaces
            step(0, lane)
    if args.phase == "calib-kl":
        if args.warmup:
            kl_block(0, args.warmup, 0)
    else:
        for i in range(args.warmup):
            step(i, i % n_streams)
    torch.cuda.synchronize()

    def replay(lane, b):
        if streams[lane] is None:
            graphs[lane][b].replay()
        else:
            with torch.cuda.stream(streams[lane]):
                graphs[lane][b].replay()

    # hipGraph replay of the step (one captured graph per resident input batch): the ~45 launches of a step are
    # launch-latency-sensitive (1.3 ms of kernels); replay removes the host from the loop.  The steps that carry the kernel
    # events of the roofline leg (every `event_every`-th) still run eagerly INSIDE the timed region.
    graphs, graph_error = None, None
    if args.graph and args.phase == "calib-naive" and not distributed:
        # one process: the calibration step (forward with online scales + net.update_ema()) holds no collective and no host
        # synchronisation, so it replays from a hipGraph like the evaluation step does - launching its ~100 kernels from
        # Python takes longer than running them (2.5 ms against ~1.5 ms for MobileNetV2).  With several ranks the step's
        # all-reduce stays outside any capture (dist.py refuses one inside) and the step launches eagerly.
        try:
            graphs = [[None] * rotate]
            pool = torch.cuda.graph_pool_handle()
            for b in range(rotate):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=pool):
                    step(b)
                graphs[0][b] = g
            for b in range(rotate):
                graphs[0][b].replay()
            torch.cuda.synchronize()
        except Exception as e:
            graphs, graph_error = None, "%s: %s" % (type(e).__name__, str(e)[:200])
            torch.cuda.synchronize()
    if args.graph and args.phase == "eval":
        try:
            # one graph per (stream, resident batch): a forward's per-stream buffers are baked into its graph, so a graph is
            # replayed on the stream it was captured on; the graphs of a stream share one memory pool (they never overlap)
            graphs = [[None] * rotate for _ in range(n_streams)]
            for lane in range(n_streams):
                pool = torch.cuda.graph_pool_handle()
                for b in range(rotate):
                    g = torch.cuda.CUDAGraph()
                    if streams[lane] is None:
                        with torch.cuda.graph(g, pool=pool):
                            eval_step(b, lane)
                    else:
                        with torch.cuda.graph(g, pool=pool, stream=streams[lane]):
                            eval_step(b, lane)
                    graphs[lane][b] = g
            for lane in range(n_streams):
                for b in range(rotate):
                    replay(lane, b)
            torch.cuda.synchronize()
        except Exception as e:                       # capture is an optimisation: fall back to eager launches, and say so
            graphs, graph_error = None, "%s: %s" % (type(e).__name__, str(e)[:200])
            torch.cuda.synchronize()

    # Kernel events are SAMPLED inside the timed region (every `event_every`-th step of every third block): bracketing all
    # ~45 launches of a step costs ~35 % of THAT step (two marker packets per launch), and the cost is charged to its block.
    if args.event_every is None:
        args.event_every = 50 if args.phase == "eval" else 7
    event_every = 0 if args.no_kernel_events else max(1, args.event_every)
    profiled_steps = 0
    if event_every:
        ops.profile_reset()

    def all_reduce(t, op):
        if backend == "nccl":
            dist.all_reduce(t, op=op)
        else:
            _gloo_sum(dist, t, op)

    def timed_block(first_step, with_events=True):
        """EXACTLY `--steps` steps between barrier + synchronize on both sides; returns the MAX over ranks (seconds).
        `with_events`: this block carries the steps whose kernels are bracketed with HIP events (every third block does)."""
        nonlocal profiled_steps
        every = event_every if with_events else 0
        barrier()
        t0 = time.perf_counter()
        if args.phase == "calib-kl":
            profiled_steps += kl_block(first_step, args.steps, every)
        else:
            sampled = []
            for i in range(first_step, first_step + args.steps):
                on = bool(every) and ((i - first_step) % every == 0)
                if on and n_streams > 1:
                    sampled.append(i)                 # two steps in flight: the sampled steps run at the END of the block, alone
                elif on:
                    ops.profile_enable(True)
                    profiled_steps += 1
                    step(i)
                    ops.profile_enable(False)
                elif graphs is not None:
                    replay(i % n_streams, i % rotate)
                else:
                    step(i, i % n_streams)
            if sampled:
                # ONE drain, then the steps that carry kernel events one after the other on one stream - still inside the
                # timed region and charged to `value` (a drain in front of and behind every sampled step cost more than the
                # second stream gained)
                torch.cuda.synchronize()
                ops.profile_enable(True)
                for i in sampled:
                    profiled_steps += 1
                    step(i, 0)
                ops.profile_enable(False)
        barrier()
        dt = time.perf_counter() - t0
        own_blocks.append(dt)
        if distributed:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            all_reduce(t, dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    # A block shorter than MIN_BLOCK_S (the driver's `--steps 20` is 24 ms) is REPEATED — same steps, same warm-up, each
    # block bracketed as above — until MIN_REGION_S has been timed; `value` is then the median block.  Every rank takes
    # the same decision: it is made on the rank-maximum.
    own_blocks = []
    fqdist.collective_stats(reset=True)              # (what warm-up and set-up exchanged is not the timed region's)
    # The bracketed steps (eager, alone, two marker packets per launch) cost their block 1.5-2 %: they ride in every THIRD block
    # (1, 4, 7 ... - not in block 0, the first after the warm-up, whose steps run up to 5 % slower on some boxes), so that
    # with three or more blocks the median block - `value` - is one without them, while the roofline figures still come from
    # events inside the timed region.  A run of a single block gets a second one that carries them.  Both kinds of block
    # are reported (`consistency`).
    EVENT_BLOCK_EVERY = max(1, int(os.environ.get("FQ_BENCH_EVENT_BLOCK_EVERY", "3")))       # (1: every block, as rounds 1-4)

    def carries_events(k):
        return EVENT_BLOCK_EVERY == 1 or k % EVENT_BLOCK_EVERY == 1

    box_probe = box_probe_start() if rank == 0 else None
    blocks = [timed_block(0, with_events=carries_events(0))]
    while sum(blocks) < args.min_region_s and len(blocks) < args.max_repeats:
        blocks.append(timed_block(len(blocks) * args.steps, with_events=carries_events(len(blocks))))
    if event_every and not any(carries_events(k) for k in range(len(blocks))):
        blocks.append(timed_block(len(blocks) * args.steps, with_events=True))
        event_idx = {len(blocks) - 1}
    else:
        event_idx = {k for k in range(len(blocks)) if carries_events(k)}
    elapsed = float(np.median(blocks))
    event_blocks = [b for k, b in enumerate(blocks) if k in event_idx]
    plain_blocks = [b for k, b in enumerate(blocks) if k not in event_idx]
    # what each rank's own clock says about the same blocks (the line's figures are the per-block MAXIMUM over ranks)
    rank_ms = None
    if distributed:
        mine = torch.tensor([float(np.median(own_blocks)) / args.steps * 1e3], dtype=torch.float64, device=dev)
        lo, hi = mine.clone(), mine.clone()
        all_reduce(lo, dist.ReduceOp.MIN)
        all_reduce(hi, dist.ReduceOp.MAX)
        rank_ms = (float(lo.item()), float(hi.item()))
    coll = fqdist.collective_stats()
    # the same steps on ONE stream (the figure of rounds 1-3), no kernel events: one block, reported beside `value`
    single_s = None
    if n_streams > 1:
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            if graphs is not None:
                replay(0, i % rotate)
            else:
                step(i, 0)
        barrier()
        single_s = time.perf_counter() - t0
        if distributed:
            t = torch.tensor([single_s], dtype=torch.float64, device=dev)
            all_reduce(t, dist.ReduceOp.MAX)
            single_s = float(t.item())
    kl_search_ms, thresholds = None, None
    if args.phase == "calib-kl":
        # the threshold search over all layers: ONE launch (reference: ~1.4 s of Python per layer); once per calibration
        from quantization.mxnet_amd.quantize.distribution_calibrate import kl_calibrate_many
        blocks_q = net.collect_quantized_blocks()
        levels = 2 ** (args.input_bits - 1 if args.input_signed else args.input_bits)
        hs = [kl_extra["hists"][b] for b in blocks_q]
        kl_calibrate_many(hs, levels=levels, min_bins=levels, bins=2048, device=dev)          # warm
        torch.cuda.synchronize()
        t_kl = time.perf_counter()
        best = kl_calibrate_many(hs, levels=levels, min_bins=levels, bins=2048, device=dev)
        torch.cuda.synchronize()
        kl_search_ms = (time.perf_counter() - t_kl) * 1e3
        thresholds = [float((bb + 0.5) * (kl_extra["ranges"][b] / 2048)) for bb, b in zip(best, blocks_q)]
    # calibration phases on several ranks: every rank must end with the SAME thresholds, bit for bit (the collectives exist for
    # exactly that) - a digest of each rank's vector, the minimum and maximum over the ranks compared on rank 0
    thresholds_equal = None
    if distributed and args.phase != "eval":
        if args.phase == "calib-naive":
            vec = torch.cat([b.input_max.data()._t.reshape(-1).float() for b in net.collect_quantized_blocks()])
        else:
            vec = torch.tensor(thresholds, dtype=torch.float32, device=dev)
        bits = vec.contiguous().view(torch.int32).to(torch.float64)
        digest = torch.stack([bits.sum(), (bits * torch.arange(1, bits.numel() + 1, device=dev, dtype=torch.float64)).sum()])
        lo_d, hi_d = digest.clone(), digest.clone()
        all_reduce(lo_d, dist.ReduceOp.MIN)
        all_reduce(hi_d, dist.ReduceOp.MAX)
        thresholds_equal = bool(torch.equal(lo_d, hi_d))
    prof = ops.profile_read()
    ops.profile_reset()
    # A bracketing event pair adds a fixed cost to every launch it times (two marker packets + dispatch latency): the
    # median pair around a one-element kernel minus that kernel's own back-to-back cost, both measured NOW on this device.
    ev_overhead_ms, null_kernel_ms = ops.profile_event_overhead_ms(dev)
    launch_overhead_ms, spin_ms = ops.profile_launch_overhead_ms(dev)

    if distributed:
        all_reduce(counters, dist.ReduceOp.SUM)              # the eval counters of simulate_quantization.py:123-147
    torch.cuda.synchronize()

    if rank == 0:
        images = world * args.batch_size * args.steps
        ms_per_step = elapsed / args.steps * 1e3
        kernels = {}
        step_bytes = 0.0
        step_moved = 0.0
        for key, rec in prof.items():
            if not rec["launches"]:
                continue
            # `frac` / `achieved` / `avg_launch_us` / `ms_per_step`: HIP-event time of the family's launches MINUS, per launch, what
            # bracketing a launch with an event pair costs - measured in THIS process on a self-timing kernel
            # (fq_profile_launch_overhead: median pair time of 200 bracketed 30 us launches - the same launches inside one pair;
            # 2.3-2.6 us, 4.7 us inside a rocprofv3 process).  Validated where it can be: in a process that runs under rocprofv3 the
            # figure equals the profiler's own kernel table of that process to 0.3 % (depthwise 474.6 vs 476.7 us per step,
            # pointwise 514.9 vs 516.2; another box: 516.5 vs 518.1, 543.4 vs 543.0; the nn.Conv2D net: +1.7 %) while the raw event
            # time sits 11-15 % above it there (profiles/r4_events_vs_rocprof.txt; tools/check_events_vs_rocprof.py gates the
            # dominant family at 3 %).  Un-profiled line against a profiled table is a comparison of two PROCESSES: they differ
            # by up to 10 % on this pool's boxes whatever is measured.  The raw figures are kept as `*_raw_events` (a lower
            # bound of the fraction).  (Round 3 removed pair(null kernel) - back-to-back(null kernel) = 4.6 us: a one-element
            # kernel hides its dispatch, that figure over-corrected by 2 us per launch and sat 10 % above the tables.)
            ms_raw = max(rec["ms"], 1e-9)
            ms_k = max(rec["ms"] - launch_overhead_ms * rec["launches"], 1e-9)
            gbs_raw = rec["bytes"] / (ms_raw * 1e-3) / 1e9
            gbs_k = rec["bytes"] / (ms_k * 1e-3) / 1e9
            step_bytes += rec["bytes"] / max(profiled_steps, 1)
            moved = rec.get("bytes_moved", rec["bytes"])
            step_moved += moved / max(profiled_steps, 1)
            gbs_moved = moved / (ms_k * 1e-3) / 1e9
            # `achieved` / `frac`: the bytes the launches MOVED over time - equal to the algorithmic bytes (4 B per input and per
            # output element) wherever every tensor crosses HBM as fp32 (the default workload), 1 B per element where a side is a
            # C16 code tensor; the 4-B-per-element figure of such a run is kept as `frac_algorithmic` (it says what the hand-over
            # saves and can exceed 1: not a roofline fraction)
            kernels[key] = {"kernel": KERNEL_NAMES.get(key, key), "achieved": round(gbs_moved, 1),
                            "frac": round(gbs_moved / HBM_PEAK_GBS, 4),
                            "achieved_algorithmic": round(gbs_k, 1), "frac_algorithmic": round(gbs_k / HBM_PEAK_GBS, 4),
                            "frac_raw_events": round(gbs_raw / HBM_PEAK_GBS, 4), "launches": rec["launches"],
                            # bytes the launches really moved (1 B per element of a C16 code tensor): THE roofline fraction of a
                            # code-hand-over run - `frac` there measures what the hand-over saves and may exceed 1
                            "achieved_actual": round(gbs_moved, 1), "frac_actual": round(gbs_moved / HBM_PEAK_GBS, 4),
                            "moved_bytes_per_launch": round(moved / rec["launches"], 1),
                            "avg_launch_us": round(ms_k * 1e3 / rec["launches"], 3),
                            "avg_launch_us_raw_events": round(ms_raw * 1e3 / rec["launches"], 3),
                            "algorithmic_bytes_per_launch": round(rec["bytes"] / rec["launches"], 1),
                            "ms_per_step": round(ms_k / max(profiled_steps, 1), 4),
                            "ms_per_step_raw_events": round(ms_raw / max(profiled_steps, 1), 4)}
        dominant = max(kernels, key=lambda k: kernels[k]["ms_per_step"]) if kernels else None
        # `traffic` would be HBM bytes of THIS run's dominant kernel, which only a rocprofv3 --pmc pass around the process can
        # give: this line never carries it (null).  What the committed PMC passes measured for the same command is quoted under
        # its own key, with where it came from.
        traffic_from_profiles = None
        default_workload = (args.model == "mobilenet1.0" and args.quant_type == "layer" and not args.offline
                            and args.weight_bits == 8 and args.input_bits == 8 and not args.no_fuse)
        # (the default workload's passes: profiles/pmc_traffic.json; another configuration's: profiles/pmc_traffic_<key>.json,
        # tools/refresh_r5.sh writes them for BASELINE configurations 3 and 4)
        cfg_key = "%s_%s_w%da%d%s" % (args.model, args.quant_type, args.weight_bits, args.input_bits,
                                      "_offline" if args.offline else "")
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json" if default_workload else "pmc_traffic_%s.json" % cfg_key)
        if dominant and args.phase == "eval" and not args.no_fuse and args.wino == "none" and os.path.exists(tpath):
            try:
                rec = json.load(open(tpath))
                krec = rec.get("kernels", {}).get(dominant, {})
                if krec.get("hbm_bytes_per_launch") is not None:
                    traffic_from_profiles = {
                        "kernel": dominant, "hbm_bytes_per_launch": krec["hbm_bytes_per_launch"],
                        "read_bytes_per_launch": krec.get("read_bytes_per_launch"),
                        "write_bytes_per_launch": krec.get("write_bytes_per_launch"),
                        "source": rec.get("source", ""), "commit": rec.get("commit"), "box": rec.get("box"),
                        "file": "profiles/" + os.path.basename(tpath),
                        "what": "NOT measured in this run: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes around this "
                                "command (tools/pmc_run.py), gfx950 corrections of MI355X_MICROARCH.md applied"}
            except Exception:
                traffic_from_profiles = None
        dk = kernels.get(dominant, {"achieved": 0.0, "frac": 0.0, "frac_raw_events": 0.0, "kernel": None})
        whole = step_bytes / (ms_per_step * 1e-3) / 1e9 if step_bytes else 0.0
        whole_moved = step_moved / (ms_per_step * 1e-3) / 1e9 if step_moved else 0.0
        flavour = "%s W%dA%d, %s input quant" % ({"layer": "per-layer", "group": "per-group", "channel": "per-channel"}
                                                  [args.quant_type], args.weight_bits, args.input_bits,
                                                  "offline" if args.offline else "online")
        if qconv:
            flavour = "nn.Conv2D(quantized=True) blocks: per-tensor uint8 inputs / int8 weights from every batch's own range, " \
                      "exact int32 sums, %s" % ("BatchNorm + ReLU folded into the stores, ranges from the producers' "
                                                "statistics (nn/fuse.py)" if not args.no_fuse else "separate BatchNorm / "
                                                "ReLU blocks, one range pass per layer")
        if args.wino != "none":
            flavour += ", Winograd-domain %s weights" % args.wino
        what_step = {"eval": "eval forward + accuracy counters",
                     "calib-naive": "naive-EMA calibration step: forward with ONLINE scales, weights re-quantised every "
                                    "forward (fixed_params = -1; the library keeps the result while the parameter is "
                                    "unchanged), then net.update_ema()" + (" - replayed from a hipGraph" if graphs else ""),
                     "calib-kl": "KL calibration batch: forward with quantisation disabled + one 2048-bin histogram per "
                                 "quantised block (collect_feature_maps, ranges fixed by the first batch)"}[args.phase]
        metric_head = {"eval": "images/sec int8-sim", "calib-naive": "images/sec naive-EMA calibration of int8-sim",
                       "calib-kl": "images/sec KL-calibration histogram collection of int8-sim"}[args.phase]
        if qconv:
            metric_head = "images/sec real-int8 (reference tests/models/quantized_mobilenet.py)"
        line = {
            "metric": "%s %s (%s)" % (metric_head, "MobileNet1.0" if args.model == "mobilenet1.0" else args.model, flavour),
            "value": round(images / elapsed, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s ImageNet-shaped (%d,3,%d,%d)/GPU, %s, first conv %s, %d %s "
                                   "blocks, %s, %d resident input batches cycled"
                                   % (args.model, args.batch_size, hw, hw, flavour, "in fp32" if qconv else "excluded", nblocks,
                                      "quantised-convolution" if qconv else "fake-quantised", what_step, rotate),
                       "phase": args.phase,
                       "global_batch": world * args.batch_size, "parallelism": "dp%d (replicated weights, sharded "
                       "batch; %s)" % (world, {"eval": "no data-path collective; counters all-reduced once",
                                               "calib-naive": "ONE all-reduce of L+1 doubles per calibration step",
                                               "calib-kl": "ranges broadcast after the first batch, ONE all-reduce of the "
                                                           "exact histograms at the end"}[args.phase]),
                       "hipgraph": graphs is not None, "hipgraph_error": graph_error,
                       "fused_producers": not args.no_fuse,
                       "streams": n_streams,
                       "streams_what": "step i runs on HIP stream i % S (one net; per-forward device state per stream): "
                                       "independent batches, S in flight; the steps that carry kernel events run at the end "
                                       "of their block, alone on one stream (inside the timed region)" if n_streams > 1
                                       else "one stream"},
            "roofline": {"bound": "hbm", "kernel": dk["kernel"], "achieved": dk["achieved"], "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": dk["frac"], "frac_actual": dk.get("frac_actual", dk["frac"]),
                         "frac_actual_what": "= frac: bytes the launches really moved - 1 B per element where a side is a C16 code "
                                             "tensor (offline hand-over configurations); for a recompute pair (online thresholds, "
                                             "round 6: fq_pwconv_i8_stat + fq_pwdw_fused) the statistic pass moves its input only "
                                             "and the fused launch the 1x1 input + the depthwise output, while frac_algorithmic keeps "
                                             "counting the 4 B per input and output element of the two layers they stand for",
                         "frac_algorithmic": dk.get("frac_algorithmic", dk["frac"]),
                         "frac_what": "algorithmic bytes / (HIP-event time of the family's launches - per launch the cost of the "
                                      "event pair, measured in this process on a self-timing kernel: launch_overhead_us_measured); in "
                                      "an un-profiled process (this line) the family times equal rocprofv3's kernel table of the same box "
                                      "to 0.1-2.4 % (round 6: dwconv 0.992 / 1.001, pwconv 0.984 / 1.024; tools/check_events_vs_"
                                      "rocprof.py, profiles/r6_events_vs_rocprof.txt); the line a process prints UNDER rocprofv3 matched "
                                      "its own table to 0.3-1.7 % in rounds 4-5 and sits 3-7 % above it in round 6's runs (the "
                                      "profiler's gap between two dispatches, which the calibration's back-to-back launches count as "
                                      "kernel time, grew: null-kernel launch 8.5 us there against 6.1 us in round 5 and 1.6 us "
                                      "un-profiled); frac_raw_events keeps the event pairs' cost in (a lower bound; 11-25 % low "
                                      "inside a profiled process, 2-10 % otherwise)",
                         "frac_raw_events": dk["frac_raw_events"],
                         "traffic": None, "traffic_from_profiles": traffic_from_profiles,
                         "frac_method": "fixed since round 4 (events minus the measured pair cost); last validated against "
                                        "profiles/r6_bench_kernel_stats.csv by tools/check_events_vs_rocprof.py "
                                        "(profiles/r6_events_vs_rocprof.txt)",
                         "dominant_by": "largest HIP-event time per step among this library's kernels",
                         "event_sampling": "HIP events bracket every library launch in %d of the %d timed steps (every %d-th "
                                           "step of every third block)%s"
                                           % (profiled_steps, args.steps * len(blocks), max(event_every, 1),
                                              "; these steps run ALONE on one stream at the end of their block, so the "
                                              "per-kernel figures are those of kernels that do not share the GPU with another "
                                              "batch (compare with a rocprofv3 table of --streams 1 --graph 0)"
                                              if n_streams > 1 or graphs is not None else ""),
                         "launch_overhead_us_measured": round(launch_overhead_ms * 1e3, 3),
                         "launch_overhead_what": "a %.1f us one-wavefront kernel, 200 launches: median event-pair time of the "
                                                 "bracketed launches - (the same launches inside ONE pair) / 200" % (spin_ms * 1e3),
                         "event_pair_minus_null_kernel_us": round(ev_overhead_ms * 1e3, 3),
                         "null_kernel_us_measured": round(null_kernel_ms * 1e3, 3),
                         "whole_step": {"algorithmic_bytes_per_step": round(step_bytes, 1), "achieved": round(whole_moved, 1),
                                        "frac": round(whole_moved / HBM_PEAK_GBS, 4),
                                        "achieved_algorithmic": round(whole, 1), "frac_algorithmic": round(whole / HBM_PEAK_GBS, 4),
                                        "moved_bytes_per_step": round(step_moved, 1), "achieved_actual": round(whole_moved, 1),
                                        "frac_actual": round(whole_moved / HBM_PEAK_GBS, 4),
                                        "what": "sum of the algorithmic bytes of every library launch of one step / "
                                                "ms_per_step (library convolutions and launch gaps included in the time)"},
                         "kernels": kernels},
            "repeats": len(blocks),
            "ranks": {"world": world, "backend": backend if distributed else None,
                      "rccl_world": (dist.get_world_size() if distributed and backend == "nccl" else None),
                      "ms_per_step_min_over_ranks": None if rank_ms is None else round(rank_ms[0], 4),
                      "ms_per_step_max_over_ranks": None if rank_ms is None else round(rank_ms[1], 4),
                      "thresholds_equal_on_all_ranks": thresholds_equal,
                      "collectives_in_timed_region": coll,
                      "collectives_per_step": {k: round(v["calls"] / float(args.steps * len(blocks)), 4) for k, v in coll.items()},
                      "bytes_per_collective": {k: round(v["bytes"] / float(max(v["calls"], 1)), 1) for k, v in coll.items()},
                      "what": "data-path collectives issued through quantization.mxnet_amd.dist between the first and the last "
                              "timed step (the timing's own rank-maximum reductions and barriers are not in here): eval - none; "
                              "calib-naive - one all-reduce of L + 1 doubles per step; calib-kl - one range broadcast + one "
                              "histogram all-reduce per collection"},
            "consistency": {"timed_block_s": round(elapsed, 5), "timed_region_s": round(sum(blocks), 4),
                            "blocks": len(blocks),
                            "what": "each block = exactly --steps steps between barrier + synchronize; ms_per_step / "
                                    "value are the MEDIAN block (max over ranks per block)",
                            "ms_per_step_min": round(min(blocks) / args.steps * 1e3, 4),
                            "ms_per_step_max": round(max(blocks) / args.steps * 1e3, 4),
                            "ms_per_step_first_block": round(blocks[0] / args.steps * 1e3, 4),
                            "event_blocks": {"which": "blocks %s carry the steps bracketed with HIP events (eager, alone on "
                                                      "one stream); the other blocks are plain" % sorted(event_idx),
                                             "n": len(event_blocks), "ms_per_step_median": round(
                                                 float(np.median(event_blocks)) / args.steps * 1e3, 4) if event_blocks
                                             else None},
                            "plain_blocks": {"n": len(plain_blocks), "ms_per_step_median": round(
                                float(np.median(plain_blocks)) / args.steps * 1e3, 4) if plain_blocks else None}},
            "eval_counters": {"images": float(counters[1].item()), "top1_correct": float(counters[0].item()),
                              "what": "fq_eval_counters over every step run so far (warm-up included), summed over "
                                      "the ranks in ONE all-reduce after the timed region"},
        }
        if single_s is not None:
            line["single_stream"] = {"value": round(images / single_s, 2), "ms_per_step": round(single_s / args.steps * 1e3, 4),
                                     "lanes_gain": round(line["value"] / max(images / single_s, 1e-9), 4),
                                     "what": "the same %d steps launched on ONE stream (--streams 1), one block, no kernel "
                                             "events; lanes_gain = value / this" % args.steps}
        line["box"] = box_info(box_probe)
        line["consistency"]["value_what"] = ("median over ALL blocks; since round 5 only every third block carries the "
                                             "event-bracketed steps (which run alone on one stream and cost their block 1-2 %), so "
                                             "with three or more blocks `value` is a plain block's; rounds 1-4 charged that cost to "
                                             "every block (FQ_BENCH_EVENT_BLOCK_EVERY=1 reproduces it)")
        if args.phase != "eval":
            del line["eval_counters"]
        if args.phase == "calib-kl":
            line["kl_search"] = {"ms": round(kl_search_ms, 3), "layers": nblocks,
                                 "what": "fq_kl_search over all layers' histograms in one launch + the read-back of "
                                         "best_bins, once per calibration (outside the per-batch figure)",
                                 "thresholds_first_last": [round(thresholds[0], 6), round(thresholds[-1], 6)]}
            from quantization.mxnet_amd.quantize import distribution_calibrate as _dc
            lib_ms = sum(k["ms_per_step"] for k in line["roofline"]["kernels"].values())
            line["split"] = {"this_library_ms_per_step": round(lib_ms, 4),
                             "tensor_library_and_host_ms_per_step": round(line["ms_per_step"] - lib_ms, 4),
                             "histograms_in_the_producers_pass": bool(_dc.FUSED_HISTOGRAMS),
                             "what": "this library: BatchNorm / residual passes (which bin what they store from a collection's "
                                     "second batch on, FQ_KL_FUSED_HIST=0 switches that off), remaining histogram passes, first-batch "
                                     "ranges, first convolution, pooling - HIP events on sampled batches; the rest of the batch: "
                                     "the fp32 convolutions of the un-quantised forward (MIOpen / rocBLAS) and launch gaps"}
        if world == 1 and not args.no_headline and args.phase == "eval":
            line["headline_tensor"] = headline_tensor(dev, ops)
            stream_gbs = line["headline_tensor"]["apply_only"]["achieved"]
            line["roofline"]["whole_step"]["frac_of_streaming_rate"] = round(line["roofline"]["whole_step"]["achieved"] / stream_gbs, 4)
            line["roofline"]["whole_step"]["streaming_rate"] = stream_gbs
            line["roofline"]["whole_step"]["streaming_rate_what"] = ("headline_tensor.apply_only of this run: the library's own "
                                                                     "read + write pass on 822 MB")
        if world == 1:
            line["cpu_baseline"] = None if args.no_cpu_baseline else cpu_baseline(args, classes, hw)
        print(json.dumps(line), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
