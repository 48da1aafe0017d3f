#!/usr/bin/env python3
"""Simulated-quantisation CLI — same flags, flow and printed banners as the reference's
examples/simulate_quantization.py (:45-119 flags, :178-353 flow), running on MI355X through libfakequant.

    python examples/simulate_quantization.py --model=mobilenet1.0 --use-gpu=0
    python examples/simulate_quantization.py --model=resnet50_v1 --quant-type=channel --quantize-input-offline \
           --calib-mode=kl --use-gpu=0
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/simulate_quantization.py \
           --model=mobilenetv2_1.0 --quant-type=channel --weight-bits-width=4 --quantize-input-offline   # 8 x MI355X

Differences that are deliberate: no `.asscalar()` per layer, accuracy counters on the device instead of a
per-sample Python loop (:139-142), KL histograms + search on the device, optional one-process-per-GPU sharding
(`torchrun`), synthetic datasets / seeded weights when ImageNet / gluoncv checkpoints are absent (no network here).
"""
import argparse
import os
import sys
import time

import numpy as np
from tqdm import tqdm

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402
from quantization.mxnet_amd import mx, ops, dist as fqdist  # noqa: E402
from quantization.mxnet_amd.mx import cpu, gpu, nd  # noqa: E402
from quantization.mxnet_amd.mx.gluon import nn  # noqa: E402
from quantization.mxnet_amd.mx.gluon.data import Sampler, DataLoader, vision  # noqa: E402
from quantization.mxnet_amd.mx.gluon.model_zoo import get_model, get_model_list  # noqa: E402
from quantization.mxnet_amd.quantize import convert  # noqa: E402
from quantization.mxnet_amd.quantize.initialize import qparams_init  # noqa: E402
from quantization.mxnet_amd.quantize.distribution_calibrate import kl_calibrate_many, collect_feature_maps  # noqa: E402

T = vision.transforms
CIFAR10, ImageNet = vision.CIFAR10, vision.ImageNet


def parse_args(argv=None):
    parser = argparse.ArgumentParser(description='Simulate for quantization.')
    parser.add_argument('--model', type=str, default=None,
                        help='type of model to use. see vision_model for options. (required)')
    parser.add_argument('--print-model', action='store_true',
                        help='print the architecture of model.')
    parser.add_argument('--list-models', action='store_true',
                        help='list all models supported for --model.')
    parser.add_argument('--use-gpu', type=int, default=-1,
                        help='run model on gpu. (default: cpu — which this build refuses: HIP device required)')
    parser.add_argument('--dataset', type=str, default="imagenet",
                        choices=['imagenet', 'cifar10'],
                        help='dataset to evaluate (default: imagenet)')
    parser.add_argument('--use-gn', action='store_true',
                        help='whether to use group norm.')
    parser.add_argument('--batch-norm', action='store_true',
                        help='enable batch normalization or not in vgg. default is false.')
    parser.add_argument('--use-se', action='store_true',
                        help='use SE layers or not in resnext. default is false.')
    parser.add_argument('--last-gamma', action='store_true',
                        help='whether to init gamma of the last BN layer in each bottleneck to 0.')
    parser.add_argument('--merge-bn', action='store_true',
                        help='merge batchnorm into convolution or not. (default: False)')
    parser.add_argument('--weight-bits-width', type=int, default=8,
                        help='bits width of weight to quantize into.')
    parser.add_argument('--input-signed', type=str, default="false",
                        help='quantize inputs into int(true) or uint(fasle). (default: false)')
    parser.add_argument('--input-bits-width', type=int, default=8,
                        help='bits width of input to quantize into.')
    parser.add_argument('--quant-type', type=str, default="layer",
                        choices=['layer', 'group', 'channel'],
                        help='quantize weights on layer/group/channel. (default: layer)')
    parser.add_argument('-j', '--num-data-workers', dest='num_workers', default=4, type=int,
                        help='number of preprocessing workers (default: 4)')
    parser.add_argument('--batch-size', type=int, default=128,
                        help='evaluate batch size per device (CPU/GPU). (default: 128)')
    parser.add_argument('--num-sample', type=int, default=5,
                        help='number of samples for every class in trainset. (default: 5)')
    parser.add_argument('--quantize-input-offline', action='store_true',
                        help='calibrate via EMA on trainset and quantize input offline.')
    parser.add_argument('--calib-mode', type=str, default="naive",
                        choices=['naive', 'kl'],
                        help='how to calibrate inputs. (default: naive)')
    parser.add_argument('--calib-epoch', type=int, default=3,
                        help='number of epoches to calibrate via EMA on trainset. (default: 3)')
    parser.add_argument('--disable-cudnn-autotune', action='store_true',
                        help='disable MIOpen find/benchmark mode to pick the best convolution algorithm.')
    parser.add_argument('--eval-per-calib', action='store_true',
                        help='evaluate once after every calibration.')
    parser.add_argument('--exclude-first-conv', type=str, default="true",
                        choices=['false', 'true'],
                        help='exclude first convolution layer when quantize. (default: true)')
    parser.add_argument('--fixed-random-seed', type=int, default=7,
                        help='set random_seed for numpy to provide reproducibility. (default: 7)')
    parser.add_argument('--wino_quantize', type=str, default="none",
                        choices=['none', 'F23', 'F43', 'F63'],
                        help='quantize weights for Conv2D in Winograd domain (default: none)')
    # additions (not in the reference)
    parser.add_argument('--pretrained', type=str, default="true",
                        help="'true' (look for a local checkpoint, else seeded weights), 'false', or a parameter file")
    parser.add_argument('--save-qparams', type=str, default=None,
                        help='write the calibrated parameters (incl. every input_max) to this file')
    parser.add_argument('--no-fuse', action='store_true',
                        help='keep BatchNorm / ReLU / depthwise convolution as separate library ops '
                             '(default: quantize.fuse.fuse_inference folds them into the fake-quant kernels)')
    parser.add_argument('--synthetic-on-device', action='store_true',
                        help='(synthetic datasets only) generate normalised image batches directly on the GPU instead '
                             'of image by image on the host, so the reported speed is the network\'s, not the data '
                             'pipeline\'s; sample values differ from the host pipeline\'s (both are random)')
    parser.add_argument('--export-scale-table', type=str, default=None,
                        help='after calibration write an ncnn-style int8 scale table (per-channel weight scales after '
                             'BN folding, one input scale per layer; quantize/freeze/scale_table.py) to this file')
    parser.add_argument('--load-qparams', type=str, default=None,
                        help='load thresholds written by --save-qparams instead of calibrating')
    opt = parser.parse_args(argv)

    if opt.list_models:
        for key in get_model_list():
            print(key)
        exit(0)
    elif opt.model is None:
        print("error: --model is required")
        exit(2)

    if fqdist.rank() == 0 and int(os.environ.get("RANK", "0")) == 0:
        print()
        print('*'*25 + ' Settings ' + '*'*25)
        for k, v in opt.__dict__.items():
            print("{0: <25}: {1}".format(k, v))
        print('*'*(25*2+len(' Setting ')))
        print()
    return opt


class DeviceSyntheticLoader(object):
    """Stand-in for DataLoader over the synthetic datasets: N(0,1) "normalised images" and uniform labels generated on
    the device, one generator seed per batch index (so every rank count / batch size sees a well-defined sequence);
    batches are strided across ranks like the real loader's."""

    def __init__(self, n_images, batch_size, shape, classes, ctx, seed, rank=0, world_size=1):
        self._n, self._b, self._shape, self._classes = int(n_images), int(batch_size), tuple(shape), int(classes)
        self._dev, self._seed, self._rank, self._world = ctx.torch_device, int(seed), int(rank), int(world_size)
        self._batches = [i for i in range((self._n + self._b - 1) // self._b) if i % self._world == self._rank]

    def __len__(self):
        return len(self._batches)

    def __iter__(self):
        g = torch.Generator(device=self._dev)
        for i in self._batches:
            b = min(self._b, self._n - i * self._b)
            g.manual_seed(self._seed * 1000003 + i)
            X = torch.randn((b,) + self._shape, device=self._dev, generator=g)
            y = torch.randint(0, self._classes, (b,), device=self._dev, generator=g).float()
            yield mx.nd.NDArray(X), mx.nd.NDArray(y)


def evaluate(net, num_class, dataloader, ctx, update_ema=False, tqdm_desc="Eval"):
    """reference :122-148.  Counters live on the device: [n_correct, total, correct[c], label[c]]; one all-reduce."""
    dev = ctx.torch_device
    counters = torch.zeros(2 + 2 * num_class, dtype=torch.float32, device=dev)
    n_images, t0 = 0, time.perf_counter()
    with tqdm(total=len(dataloader), desc=tqdm_desc, disable=fqdist.rank() != 0) as pbar:
        for i, (X, y) in enumerate(dataloader):
            X = X.as_in_context(ctx)
            y = y.as_in_context(ctx)._t.long()
            outputs = net(X)
            if update_ema:
                net.update_ema()
            ops.eval_counters(outputs._t, y, counters)       # argmax + per-class counters, one launch (fq_eval_counters)
            n_images += int(y.numel())
            pbar.update(1)
    torch.cuda.synchronize(dev) if dev.type == "cuda" else None
    elapsed = time.perf_counter() - t0
    fqdist.allreduce_eval_counters(counters)
    c = counters.cpu().numpy()
    eval_acc = float(c[0] / max(c[1], 1.0))
    eval_acc_avg = float((c[2:2 + num_class] / (c[2 + num_class:] + 1e-10)).mean())
    evaluate.last_images_per_sec = n_images * fqdist.world_size() / max(elapsed, 1e-9)
    return eval_acc, eval_acc_avg


class UniformSampler(Sampler):
    """`num_per_class` indices per class from one shuffled pass (reference :151-175); numpy's global RNG, so every
    rank (same seed) draws the same sequence."""

    def __init__(self, classes, num_per_class, labels):
        self._classes = classes
        self._num_per_class = num_per_class
        self._labels = labels

    def __iter__(self):
        sample_indices = []
        label_counter = np.zeros(self._classes)
        shuffle_indices = np.arange(len(self._labels))
        np.random.shuffle(shuffle_indices)
        for idx in shuffle_indices:
            label = self._labels[idx]
            if label_counter[label] < self._num_per_class:
                sample_indices.append(idx)
                label_counter[label] += 1
            if label_counter.sum() == self._classes * self._num_per_class:
                break
        for idx, cnt in enumerate(label_counter):
            if cnt < self._num_per_class:
                raise ValueError("Number of samples for class {} is {} < {}".format(idx, cnt, self._num_per_class))
        return iter(sample_indices)

    def __len__(self):
        return self._classes * self._num_per_class


def banner(title, lines=(), rank0=True):
    if not rank0:
        return
    print('*' * 25 + ' ' + title + ' ' + '*' * 25)
    for line in lines:
        print(line)
    print('*' * (25 * 2 + 2 + len(title)))
    print()


def main(argv=None):
    rank, local_rank, world = fqdist.init()
    opt = parse_args(argv)
    if world > 1:
        ctx = gpu(local_rank)
    else:
        ctx = gpu(opt.use_gpu) if opt.use_gpu != -1 else cpu()
    if ctx.device_type == "cpu":
        raise SystemExit("error: this build runs the fake-quant path on an MI355X only (no CPU fallback); "
                         "pass --use-gpu=<id>")
    try:
        return run(opt, ctx, rank, world)
    finally:
        fqdist.shutdown()


def run(opt, ctx, rank=0, world=1):
    r0 = rank == 0

    # set random_seed for numpy (identical on every rank: same weights, same sampler draws)
    np.random.seed(opt.fixed_random_seed)
    torch.backends.cudnn.benchmark = not opt.disable_cudnn_autotune and False   # measured: no gain, +60 s

    # get model (:188-204)
    model_name = opt.model
    classes = 10 if opt.dataset == 'cifar10' else 1000
    pretrained = {"true": True, "false": False}.get(opt.pretrained.lower(), opt.pretrained)
    kwargs = {'pretrained': pretrained, 'classes': classes}
    if opt.use_gn:
        raise NotImplementedError("--use-gn needs gluoncv.nn.GroupNorm, which is not part of this build")
    if model_name.startswith('vgg'):
        kwargs['batch_norm'] = opt.batch_norm
    elif model_name.startswith('resnext'):
        kwargs['use_se'] = opt.use_se
    if opt.last_gamma:
        kwargs['last_gamma'] = True
    net = get_model(model_name, **kwargs)

    if opt.print_model and r0:
        banner(opt.model, [repr(net)])

    # convert model to quantization version (:213-250)
    convert_fn = {
        nn.Conv2D: convert.gen_conv2d_converter(
            quantize_input=True,
            wino_quantize=opt.wino_quantize,
            fake_bn=opt.merge_bn,
            input_signed=opt.input_signed == 'true',
            weight_width=opt.weight_bits_width,
            input_width=opt.input_bits_width,
            quant_type=opt.quant_type
        ),
        nn.Dense: convert.gen_dense_converter(
            quantize_input=True,
            input_signed=opt.input_signed == 'true',
            weight_width=opt.weight_bits_width,
            input_width=opt.input_bits_width,
            quant_type=opt.quant_type
        ),
        nn.Activation: None,
        nn.BatchNorm: convert.bypass_bn if opt.merge_bn else None
    }
    exclude_blocks = []
    if opt.exclude_first_conv == 'true':
        exclude_blocks.extend([net.features[0], net.features[1]])
    if model_name.startswith('mobilenetv2_'):
        exclude_blocks.append(net.output[0])
    if model_name.startswith('cifar_resnet'):
        exclude_blocks.extend([net.features[2][0].body[0], net.features[2][0].body[1]])
    banner('Exclude blocks', [b.name for b in exclude_blocks], r0)
    convert.convert_model(net, exclude=exclude_blocks, convert_fn=convert_fn)

    # initialize for quantization parameters and reset context (:253-255)
    qparams_init(net)
    net.collect_params().reset_ctx(ctx)
    if not opt.no_fuse and ctx.device_type == "gpu":
        from quantization.mxnet_amd.quantize import fuse
        n_fused = fuse.fuse_inference(net)
        if r0:
            print("[fuse] %d BatchNorm / depthwise blocks folded into fused HIP passes (--no-fuse to disable)" % n_fused)

    # construct transformer (:258-269)
    if opt.dataset == 'imagenet':
        eval_transformer = T.Compose([
            T.Resize(256, keep_ratio=True),
            T.CenterCrop(224),
            T.ToTensor(),
            T.Normalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])
        ])
    else:
        eval_transformer = T.Compose([
            T.ToTensor(),
            T.Normalize([0.4914, 0.4822, 0.4465], [0.2023, 0.1994, 0.2010])
        ])

    # fetch dataset and dataloader (:272-292); batches are strided across ranks
    dataset = ImageNet if opt.dataset == 'imagenet' else CIFAR10
    shard = fqdist.shard_loader_kwargs()
    eval_dataset = dataset(train=False).transform_first(eval_transformer)
    eval_loader = DataLoader(dataset=eval_dataset, batch_size=opt.batch_size, num_workers=opt.num_workers,
                             last_batch='keep', **shard)
    if opt.synthetic_on_device:
        hw_in = 224 if opt.dataset == 'imagenet' else 32
        eval_loader = DeviceSyntheticLoader(len(eval_dataset), opt.batch_size, (3, hw_in, hw_in), classes, ctx, 7, **shard)
        if opt.quantize_input_offline and not opt.load_qparams:
            train_loader = DeviceSyntheticLoader(classes * opt.num_sample, opt.batch_size, (3, hw_in, hw_in), classes, ctx,
                                                 11, **shard)
    if opt.quantize_input_offline and not opt.load_qparams and not opt.synthetic_on_device:
        train_dataset = dataset(train=True).transform_first(eval_transformer)
        if opt.dataset == 'imagenet':
            train_labels = [item[1] for item in train_dataset._data.items]
        elif opt.dataset == 'cifar10':
            train_labels = train_dataset._data._label
        train_loader = DataLoader(dataset=train_dataset, batch_size=opt.batch_size,
                                  sampler=UniformSampler(classes, opt.num_sample, train_labels),
                                  num_workers=opt.num_workers, last_batch='keep', **shard)

    def report(acc, avg_acc):
        if r0:
            print('{0: <8}: {1:2.2f}%'.format('acc', acc * 100))
            print('{0: <8}: {1:2.2f}%'.format('avg_acc', avg_acc * 100))
            print('{0: <8}: {1:.1f} images/sec on {2} GPU(s)'.format('speed', evaluate.last_images_per_sec, world))

    # calibrate for input ranges and evaluate for simulation (:294-353)
    if opt.quantize_input_offline:
        if opt.load_qparams:
            net.load_parameters(opt.load_qparams, ctx=ctx, allow_missing=True, ignore_extra=True)
        elif opt.calib_mode == "kl":
            if r0:
                print('*' * 25 + ' KL Calibration ' + '*' * 25)
            net.disable_quantize()      # calibrate with fp32_input and fp32_weight inference
            input_levels = 2 ** ((opt.input_bits_width - 1) if opt.input_signed == "true" else opt.input_bits_width)
            min_bins, bins = input_levels, 2048
            # collect feature maps: histograms accumulate on the device; ranks exchange max (first batch) + counts (end)
            hist_collector, fm_max_collector = collect_feature_maps(net, bins=bins, loader=train_loader, ctx=ctx,
                                                                    sync=fqdist.kl_sync if world > 1 else None)
            # do calibration: all layers in one launch
            quantized_blocks = net.collect_quantized_blocks()
            n_quantized_blocks = len(quantized_blocks)
            best = kl_calibrate_many([hist_collector[m] for m in quantized_blocks], levels=input_levels,
                                     min_bins=min_bins, bins=bins, device=ctx.torch_device)
            thresholds = {}
            for i, (m, best_bins) in enumerate(zip(quantized_blocks, best)):
                thresholds[m] = (best_bins + 0.5) * (fm_max_collector[m] / bins)
                if r0:
                    print(f"({i+1}/{n_quantized_blocks})\tBest threshold for {m.name}: {thresholds[m]}")
            # update input_max
            for m, th in thresholds.items():
                m.input_max.set_data(nd.array([th], ctx=ctx))
            net.enable_quantize()
            if r0:
                print('*' * (25 * 2 + len(' KL Calibration ')))
                print()
        else:
            if r0:
                print('*' * 25 + ' Naive Calibration ' + '*' * 25)
            if world > 1:
                fqdist.attach_calibration_sync(net, opt.batch_size)
            for i in range(opt.calib_epoch):
                net.quantize_input(enable=True, online=True)    # calibrate with int_input and int_weight inference
                _ = evaluate(net, classes, train_loader, ctx=ctx, update_ema=True,
                             tqdm_desc="Calib[{}/{}]".format(i+1, opt.calib_epoch))
                if opt.eval_per_calib:
                    net.quantize_input(enable=True, online=False)
                    acc, avg_acc = evaluate(net, classes, eval_loader, ctx=ctx, update_ema=False,
                                            tqdm_desc="Eval[{}/{}]".format(i + 1, opt.calib_epoch))
                    report(acc, avg_acc)
                    if r0:
                        print()
            if world > 1:
                fqdist.detach_calibration_sync(net)
            if r0:
                for m in net.collect_quantized_blocks():
                    print(f"Best threshold for {m.name}: {m.input_max.data().asscalar()}")
                print('*' * (25 * 2 + len(' Naive Calibration ')))
                print()
        if opt.save_qparams and r0:
            net.save_parameters(opt.save_qparams)
        if opt.export_scale_table and r0:
            from quantization.mxnet_amd.quantize.freeze import export_scale_table
            export_scale_table(net, opt.export_scale_table, weight_width=8, input_width=8,
                               json_path=opt.export_scale_table + ".json")
        if not opt.eval_per_calib:
            net.fix_params()
            net.quantize_input(enable=True, online=False)
            acc, avg_acc = evaluate(net, classes, eval_loader, ctx=ctx, update_ema=False)
            if r0:
                print('*' * 25 + ' Result ' + '*' * 25)
            report(acc, avg_acc)
            if r0:
                print('*' * (25 * 2 + len(' Result ')))
                print()
    else:
        net.fix_params()
        net.quantize_input(enable=True, online=True)
        acc, avg_acc = evaluate(net, classes, eval_loader, ctx=ctx, update_ema=False)
        if r0:
            print('*'*25 + ' Result ' + '*'*25)
        report(acc, avg_acc)
        if r0:
            print('*'*(25*2 + len(' Result ')))
            print()
    return acc, avg_acc, net


if __name__ == "__main__":
    main()
