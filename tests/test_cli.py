"""examples/simulate_quantization.py: flags identical to the reference's CLI (:49-103) and its three flows end to end.
CPU runs use the oracle stand-in for the HIP entry points (host logic only); GPU runs are marked."""
import importlib.util
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT
from oracle.patch import oracle_ops


def _cli():
    spec = importlib.util.spec_from_file_location("fq_cli", os.path.join(ROOT, "examples", "simulate_quantization.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


REFERENCE_FLAGS = ["--model", "--print-model", "--list-models", "--use-gpu", "--dataset", "--use-gn", "--batch-norm",
                   "--use-se", "--last-gamma", "--merge-bn", "--weight-bits-width", "--input-signed",
                   "--input-bits-width", "--quant-type", "-j", "--num-data-workers", "--batch-size", "--num-sample",
                   "--quantize-input-offline", "--calib-mode", "--calib-epoch", "--disable-cudnn-autotune",
                   "--eval-per-calib", "--exclude-first-conv", "--fixed-random-seed", "--wino_quantize"]


def test_cli_keeps_every_reference_flag_and_default():
    cli = _cli()
    src = open(os.path.join(ROOT, "examples", "simulate_quantization.py")).read()
    for flag in REFERENCE_FLAGS:
        assert re.search(r"['\"]%s['\"]" % re.escape(flag), src), flag
    opt = cli.parse_args(["--model", "mobilenet1.0"])
    assert (opt.use_gpu, opt.dataset, opt.weight_bits_width, opt.input_signed, opt.input_bits_width, opt.quant_type,
            opt.num_workers, opt.batch_size, opt.num_sample, opt.calib_mode, opt.calib_epoch, opt.exclude_first_conv,
            opt.fixed_random_seed, opt.wino_quantize, opt.quantize_input_offline, opt.merge_bn) == \
        (-1, "imagenet", 8, "false", 8, "layer", 4, 128, 5, "naive", 3, "true", 7, "none", False, False)


def test_cli_refuses_cpu_context():
    cli = _cli()
    with pytest.raises(SystemExit, match="no CPU fallback"):
        cli.main(["--model", "cifar_resnet20_v1", "--dataset", "cifar10"])


def test_uniform_sampler_matches_reference_semantics():
    cli = _cli()
    labels = np.arange(200) % 10
    np.random.seed(7)
    s = cli.UniformSampler(10, 3, labels)
    idx = list(iter(s))
    assert len(idx) == len(s) == 30
    assert all((labels[idx] == c).sum() == 3 for c in range(10))
    np.random.seed(7)
    assert idx == list(iter(cli.UniformSampler(10, 3, labels)))         # same seed -> same draw on every rank
    with pytest.raises(ValueError, match="Number of samples for class"):
        list(iter(cli.UniformSampler(10, 30, labels)))


@pytest.fixture()
def tiny_data(monkeypatch):
    monkeypatch.setenv("FQ_SYNTH_VAL_IMAGES", "16")
    monkeypatch.setenv("FQ_SYNTH_TRAIN_PER_CLASS", "2")


BASE = ["--model", "cifar_resnet20_v1", "--dataset", "cifar10", "--batch-size", "8", "--num-sample", "1"]


@pytest.mark.parametrize("extra", [[], ["--quantize-input-offline", "--calib-epoch", "1", "--quant-type", "channel",
                                        "--weight-bits-width", "4"],
                                   ["--input-signed", "true", "--quant-type", "channel", "--wino_quantize", "F23"]],
                         ids=["online", "naive_ema_w4", "signed_wino"])
def test_cli_flows_on_cpu_with_oracle(tiny_data, capsys, extra):
    from quantization.mxnet_amd import mx
    from quantization.mxnet_amd.mx.gluon import nn
    cli = _cli()
    opt = cli.parse_args(BASE + extra)
    with oracle_ops():
        acc, avg_acc, net = cli.run(opt, mx.cpu())
    out = capsys.readouterr().out
    assert "Exclude blocks" in out and "Result" in out and "acc" in out
    assert 0.0 <= acc <= 1.0 and 0.0 <= avg_acc <= 1.0
    blocks = net.collect_quantized_blocks()
    assert len(blocks) == 20
    assert all(b.fixed_params == 1 for b in blocks if isinstance(b, nn.Conv2D))
    if "--quantize-input-offline" in extra:
        assert out.count("Best threshold for") == 20
        assert all(b.quantize_input_offline and float(b.input_max.data().asscalar()) > 0 for b in blocks)


def test_cli_runs_vgg_with_the_batch_norm_flag_on_cpu_with_oracle(tiny_data, capsys):
    """`--batch-norm` goes to the zoo for vgg names only (simulate_quantization.py:197); with `--merge-bn` the thirteen BatchNorms are
    folded into their convolutions (convert_conv2d.py:47-51).  vgg11: 7 quantised convolutions (the first is excluded) + 3 Dense."""
    from quantization.mxnet_amd import mx
    from quantization.mxnet_amd.mx.gluon import nn
    cli = _cli()
    opt = cli.parse_args(["--model", "vgg11", "--batch-norm", "--dataset", "cifar10", "--batch-size", "8", "--num-sample", "1",
                          "--quantize-input-offline", "--calib-epoch", "1"])
    with oracle_ops():
        acc, avg_acc, net = cli.run(opt, mx.cpu())
    out = capsys.readouterr().out
    assert "Result" in out and 0.0 <= acc <= 1.0
    blocks = net.collect_quantized_blocks()
    assert sum(isinstance(b, nn.Conv2D) for b in blocks) == 7 and sum(isinstance(b, nn.Dense) for b in blocks) == 3
    assert sum(type(b) is nn.BatchNorm for b in net.features._children.values()) == 8
    assert all(float(b.input_max.data().asscalar()) > 0 for b in blocks)


def test_cli_kl_flow_and_qparams_roundtrip_on_cpu_with_oracle(tiny_data, capsys, tmp_path):
    from quantization.mxnet_amd import mx
    cli = _cli()
    path = str(tmp_path / "q.npz")
    opt = cli.parse_args(BASE + ["--quantize-input-offline", "--calib-mode", "kl", "--input-bits-width", "4",
                                 "--save-qparams", path])
    with oracle_ops():
        acc, _, net = cli.run(opt, mx.cpu())
        out = capsys.readouterr().out
        assert "KL Calibration" in out and out.count("Best threshold for") == 20
        thr = [b.input_max.data().asscalar() for b in net.collect_quantized_blocks()]
        assert all(t > 0 for t in thr)
        opt2 = cli.parse_args(BASE + ["--quantize-input-offline", "--load-qparams", path])
        acc2, _, net2 = cli.run(opt2, mx.cpu())
        thr2 = [b.input_max.data().asscalar() for b in net2.collect_quantized_blocks()]
    assert thr == thr2 and acc == acc2          # thresholds survive the parameter file (checkpoint/resume row)


# The reference's four canned invocations (examples/scripts/simulate_quantization.md:5-23), flag for flag; only the data
# volume is cut down (tiny_data: 16 validation images, 2 training images per class; --num-sample / --batch-size / one
# calibration epoch) - the scripts themselves ask for 500 samples per class of the real CIFAR-10.
CANNED = {
    "c10_r56_uint4_int4_layer_merge_naive": ["--model=cifar_resnet56_v1", "--input-bits-width=4", "--weight-bits-width=4",
                                             "--dataset=cifar10", "--merge-bn", "--quantize-input-offline",
                                             "--calib-mode=naive"],
    "c10_r56_uint4_int4_layer_merge_kl": ["--model=cifar_resnet56_v1", "--input-bits-width=4", "--weight-bits-width=4",
                                          "--dataset=cifar10", "--merge-bn", "--quantize-input-offline", "--calib-mode=kl"],
    "c10_r56_uint4_int4_channel_merge_naive": ["--model=cifar_resnet56_v1", "--quant-type=channel", "--input-bits-width=4",
                                               "--weight-bits-width=4", "--dataset=cifar10", "--merge-bn",
                                               "--quantize-input-offline", "--calib-mode=naive"],
    "c10_r56_uint4_int4_channel_merge_kl": ["--model=cifar_resnet56_v1", "--quant-type=channel", "--input-bits-width=4",
                                            "--weight-bits-width=4", "--dataset=cifar10", "--merge-bn",
                                            "--quantize-input-offline", "--calib-mode=kl"],
}
CANNED_CUT = ["--num-sample=1", "--batch-size=8", "--calib-epoch=1"]


def _check_canned(name, out, acc, net):
    from quantization.mxnet_amd.mx.gluon import nn
    assert "Result" in out and 0.0 <= acc <= 1.0
    blocks = net.collect_quantized_blocks()
    convs = [b for b in blocks if isinstance(b, nn.Conv2D)]
    assert len(convs) >= 54 and out.count("Best threshold for") == len(blocks)
    assert all(b.quantize_input_offline and float(b.input_max.data().asscalar()) > 0 for b in blocks)
    assert all(b.fixed_params == 1 for b in convs)
    # --merge-bn: every converted convolution folded its BatchNorm (fake_bn) and carries the folded bias
    assert all(b.quantize_args.fake_bn and b.bias is not None for b in convs)
    assert all(b.quantize_args.wt_width == 4 and b.quantize_args.in_width == 4 for b in convs)
    assert all(b.quantize_args.quant_type == ("channel" if "channel" in name else "layer") for b in convs)
    assert ("KL Calibration" in out) == name.endswith("_kl")


@pytest.mark.parametrize("name", sorted(CANNED))
def test_canned_reference_invocations_on_cpu_with_oracle(tiny_data, capsys, name):
    from quantization.mxnet_amd import mx
    cli = _cli()
    opt = cli.parse_args(CANNED[name] + CANNED_CUT)
    with oracle_ops():
        acc, _, net = cli.run(opt, mx.cpu())
    _check_canned(name, capsys.readouterr().out, acc, net)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CANNED))
def test_canned_reference_invocations_on_gpu(tiny_data, capsys, name):
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    cli = _cli()
    acc, _, net = cli.main(CANNED[name] + CANNED_CUT + ["--use-gpu=0"])
    _check_canned(name, capsys.readouterr().out, acc, net)


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--quantize-input-offline", "--calib-epoch", "2", "--quant-type", "channel",
                                        "--weight-bits-width", "4"],
                                   ["--quantize-input-offline", "--calib-mode", "kl", "--quant-type", "channel"],
                                   ["--quant-type", "channel", "--wino_quantize", "F43"], ["--merge-bn"]],
                         ids=["online", "naive_ema_w4", "kl", "wino_f43", "merge_bn"])
def test_cli_flows_on_gpu(tiny_data, capsys, extra):
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    cli = _cli()
    acc, avg_acc, net = cli.main(BASE + ["--use-gpu", "0"] + extra)
    out = capsys.readouterr().out
    assert "Result" in out and "images/sec" in out
    assert 0.0 <= acc <= 1.0


@pytest.mark.gpu
def test_scale_table_export_on_gpu(tiny_data, tmp_path):
    """--export-scale-table after a naive-EMA calibration on the device: the ncnn-layout table (one `<layer>_param_0` line
    of per-channel weight multipliers per quantised layer, then one `<layer>` line with the input multiplier) against an
    independent numpy computation from the net's own parameters and thresholds.  (The notebook the reference's README
    points to for this table, examples/mobilenet_gluon2ncnn.ipynb, is not part of the reference tree: no reference-made
    fixture exists; the format is ncnn's int8 calibration table.)"""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from quantization.mxnet_amd.mx.gluon import nn
    cli = _cli()
    path = str(tmp_path / "scales.table")
    _, _, net = cli.main(BASE + ["--use-gpu", "0", "--quantize-input-offline", "--calib-epoch", "1", "--quant-type",
                                 "channel", "--export-scale-table", path])
    lines = [l.split() for l in open(path).read().strip().split("\n")]
    blocks = [b for b in net.collect_quantized_blocks() if isinstance(b, (nn.Conv2D, nn.Dense))]
    assert [l[0] for l in lines] == [b.name + "_param_0" for b in blocks] + [b.name for b in blocks]
    by = {l[0]: np.asarray(l[1:], np.float64) for l in lines}
    for b in blocks:
        w = b.weight.data().asnumpy().astype(np.float64)
        want = 127.0 / np.abs(w.reshape(w.shape[0], -1)).max(axis=1)
        np.testing.assert_allclose(by[b.name + "_param_0"], want, rtol=1e-6)
        thr = float(b.input_max.data().asscalar())
        assert thr > 0
        np.testing.assert_allclose(by[b.name], [127.0 / thr], rtol=1e-6)
    import json
    js = json.load(open(path + ".json"))
    assert len(js) == 2 * len(blocks) and {e["kind"] for e in js} == {"weight", "input"}


@pytest.mark.gpu
def test_evaluation_state_is_reused_across_passes_and_dropped_when_the_mode_changes(tiny_data, monkeypatch):
    """Round 5: `evaluate` keeps its lanes, captured graphs and counters on the net.  A second pass over the same (resident)
    batches replays the first pass's graphs - no new capture - and counts the same; eager launches count the same; a change
    of mode (`fix_params`: new weight-code tensors) or a parameter written through the Gluon surface drops the graphs."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from quantization.mxnet_amd import mx
    cli = _cli()
    opt = cli.parse_args(["--model", "mobilenet1.0", "--use-gpu", "0", "--batch-size", "4", "--pretrained", "false",
                          "--synthetic-on-device", "--synthetic-resident", "6"])
    ctx = mx.gpu(0)
    sim = cli.Simulation(opt, ctx, 0, 1)
    np.random.seed(opt.fixed_random_seed)
    sim.build_net()
    sim.quantise_net()
    monkeypatch.setenv("FQ_SYNTH_VAL_IMAGES", "50")      # 12 full batches of 4 + a ragged one of 2
    sim.make_loaders()
    sim.net.fix_params()
    sim.net.quantize_input(enable=True, online=True)
    first = cli.evaluate(sim.net, 1000, sim.eval_loader, ctx, streams=3, graph=1)
    su1 = dict(cli.evaluate.last_setup)
    assert su1["captures"] == 6 and cli.evaluate.last_replayed == 9          # 3 eager firsts, ragged last, 9 replays
    second = cli.evaluate(sim.net, 1000, sim.eval_loader, ctx, streams=3, graph=1)
    su2 = dict(cli.evaluate.last_setup)
    assert su2["captures"] == 0 and cli.evaluate.last_replayed == 12 and su2["pass_of_this_state"] == 2
    eager = cli.evaluate(sim.net, 1000, sim.eval_loader, ctx, streams=1, graph=0)
    assert first == second == eager
    sim.net.fix_params()                                 # the next forward freezes the weights again, into NEW tensors
    third = cli.evaluate(sim.net, 1000, sim.eval_loader, ctx, streams=3, graph=1)
    assert cli.evaluate.last_setup["captures"] == 6 and cli.evaluate.last_setup["pass_of_this_state"] == 1 and third == first
    blk = sim.net.collect_quantized_blocks()[3]
    blk.weight.set_data(blk.weight.data() * 0.5)          # a parameter written through Gluon: derived caches are rebuilt
    sim.net.fix_params()
    fourth = cli.evaluate(sim.net, 1000, sim.eval_loader, ctx, streams=3, graph=1)
    assert cli.evaluate.last_setup["pass_of_this_state"] == 1
    fifth = cli.evaluate(sim.net, 1000, sim.eval_loader, ctx, streams=1, graph=0)
    assert fourth == fifth
