"""Real inputs for the drop-in CLI (missing 1): MXNet's NDArray-list `.params` format behind `load_parameters` /
`get_model(pretrained=True)` with gluoncv's structural names, gluoncv's ImageFolder-layout ImageNet and MXNet's CIFAR-10 batch
files behind `gluon.data.vision.{ImageNet, CIFAR10}` (reference examples/simulate_quantization.py:188-204, 259-279) - synthetic
only when the root is absent, and the CLI says which it used.  The files are written here byte by byte from the format's
description (mx/ndarray_file.py), not with the package's own writer, so reader and writer are checked against the layout and
not against each other."""
import os
import pickle
import struct

import numpy as np
import pytest

from oracle.patch import oracle_ops
from quantization.mxnet_amd import mx
from quantization.mxnet_amd.mx import ndarray_file
from quantization.mxnet_amd.mx.gluon import data as gdata
from quantization.mxnet_amd.mx.gluon.model_zoo import find_checkpoint, get_model

FLAGS = {np.dtype(np.float32): 0, np.dtype(np.float64): 1, np.dtype(np.float16): 2, np.dtype(np.uint8): 3,
         np.dtype(np.int32): 4, np.dtype(np.int8): 5, np.dtype(np.int64): 6}


def pack_v2(arr, magic=0xF993FAC9, ctx=(2, 3)):
    """One NDArray record as mxnet 1.x NDArray::Save writes it (saved from gpu(3): the context is ignored on load)."""
    out = struct.pack("<I", magic)
    if magic != 0xF993FAC8:
        out += struct.pack("<i", 0)                                   # dense storage
    out += struct.pack("<I", arr.ndim) + b"".join(struct.pack("<q", d) for d in arr.shape)
    out += struct.pack("<ii", *ctx) + struct.pack("<i", FLAGS[arr.dtype]) + arr.tobytes()
    return out


def pack_list(named, record=pack_v2):
    out = struct.pack("<QQ", 0x112, 0) + struct.pack("<Q", len(named))
    out += b"".join(record(a) for _, a in named)
    names = [k for k, _ in named if k is not None]
    out += struct.pack("<Q", len(names))
    for k in names:
        out += struct.pack("<Q", len(k.encode())) + k.encode()
    return out


def test_ndarray_list_format_is_read_as_mxnet_writes_it(tmp_path):
    rng = np.random.default_rng(0)
    named = [("features.0.weight", rng.standard_normal((4, 3, 3, 3)).astype(np.float32)),
             ("features.1.running_var", rng.random(4).astype(np.float32)),
             ("arg:fc_weight", rng.standard_normal((2, 5)).astype(np.float64)),
             ("aux:bn_moving_mean", rng.integers(-9, 9, 7).astype(np.int32)),
             ("codes", rng.integers(-128, 127, (2, 3)).astype(np.int8)),
             ("half", rng.standard_normal(3).astype(np.float16)),
             ("bytes", rng.integers(0, 255, (2, 2, 2)).astype(np.uint8)),
             ("long", np.arange(3, dtype=np.int64))]
    f = tmp_path / "hand.params"
    f.write_bytes(pack_list(named))
    assert ndarray_file.is_ndarray_file(str(f))
    got = ndarray_file.load(str(f))
    assert list(got) == [k for k, _ in named]
    for k, a in named:
        assert got[k].dtype == a.dtype and np.array_equal(got[k], a), k
    stripped = ndarray_file.load_params(str(f))
    assert "fc_weight" in stripped and "bn_moving_mean" in stripped and "arg:fc_weight" not in stripped
    # V3 (numpy shape semantics) and V1 (no storage-type word) records; an unnamed list
    for magic in (0xF993FACA, 0xF993FAC8):
        f.write_bytes(pack_list(named[:2], lambda a: pack_v2(a, magic)))
        got = ndarray_file.load(str(f))
        assert all(np.array_equal(got[k], a) for k, a in named[:2])
    f.write_bytes(pack_list([(None, named[0][1]), (None, named[1][1])]))
    got = ndarray_file.load(str(f))
    assert isinstance(got, list) and np.array_equal(got[0], named[0][1]) and np.array_equal(got[1], named[1][1])
    # the oldest layout: no magic, the record starts with ndim, uint32 dims
    a = named[0][1]
    legacy = struct.pack("<I", a.ndim) + b"".join(struct.pack("<I", d) for d in a.shape) + struct.pack("<iii", 1, 0, 0) + a.tobytes()
    f.write_bytes(struct.pack("<QQQ", 0x112, 0, 1) + legacy + struct.pack("<Q", 1) + struct.pack("<Q", 1) + b"w")
    assert np.array_equal(ndarray_file.load(str(f))["w"], a)
    # damage is reported, not read past
    blob = pack_list(named)
    f.write_bytes(blob[:len(blob) // 2])
    with pytest.raises(ValueError, match="truncated"):
        ndarray_file.load(str(f))
    f.write_bytes(b"PK\x03\x04 not an ndarray file at all ....")
    assert not ndarray_file.is_ndarray_file(str(f))
    with pytest.raises(ValueError, match="0x112"):
        ndarray_file.load(str(f))


def test_writer_produces_the_same_bytes_as_the_hand_packed_file(tmp_path):
    rng = np.random.default_rng(1)
    named = [("a.weight", rng.standard_normal((3, 2)).astype(np.float32)), ("a.bias", rng.standard_normal(3).astype(np.float32))]
    f = tmp_path / "w.params"
    ndarray_file.save(str(f), dict(named))
    assert f.read_bytes() == pack_list(named, lambda a: pack_v2(a, ctx=(1, 0)))
    mx.nd.save(str(f), {k: mx.nd.array(v) for k, v in named})
    back = mx.nd.load(str(f))
    assert all(np.array_equal(back[k].asnumpy(), v) for k, v in named)


def test_model_zoo_loads_a_gluoncv_style_checkpoint(tmp_path, monkeypatch):
    """`get_model(name, pretrained=True)`: `<name>-<hash>.params` under $MXNET_HOME/models, structural names, MXNet byte layout."""
    np.random.seed(3)
    donor = get_model("cifar_resnet20_v1", classes=10)
    named = [(k, p.data().asnumpy()) for k, p in donor._collect_params_with_prefix().items()]
    assert {"features.0.weight", "features.1.gamma", "features.1.running_mean", "output.weight", "output.bias"} <= {k for k, _ in named}
    home = tmp_path / "mxhome"
    (home / "models").mkdir(parents=True)
    (home / "models" / "cifar_resnet20_v1-0a1b2c3d.params").write_bytes(pack_list(named))
    monkeypatch.setenv("MXNET_HOME", str(home))
    assert find_checkpoint("cifar_resnet20_v1").endswith("cifar_resnet20_v1-0a1b2c3d.params")
    np.random.seed(99)                                             # other random weights underneath: the file must replace them all
    net = get_model("cifar_resnet20_v1", classes=10, pretrained=True)
    x = mx.nd.array(np.random.default_rng(5).standard_normal((2, 3, 32, 32)).astype(np.float32))
    assert np.array_equal(net(x).asnumpy(), donor(x).asnumpy())
    # the legacy full-name form (collect_params().save / save_params) through the same entry point
    legacy = [(k, p.data().asnumpy()) for k, p in donor.collect_params().items()]
    f = tmp_path / "legacy.params"
    f.write_bytes(pack_list([(k[len(donor.prefix):], a) for k, a in legacy]))
    np.random.seed(98)
    net2 = get_model("cifar_resnet20_v1", classes=10)
    net2.load_parameters(str(f))
    assert np.array_equal(net2(x).asnumpy(), donor(x).asnumpy())
    # missing / extra names fail the way Gluon's do
    f.write_bytes(pack_list(named[:-1]))
    with pytest.raises(AssertionError, match="is missing in file"):
        net2.load_parameters(str(f))
    f.write_bytes(pack_list(named + [("features.99.weight", np.zeros(1, np.float32))]))
    with pytest.raises(ValueError, match="not present"):
        net2.load_parameters(str(f))
    net2.load_parameters(str(f), ignore_extra=True)
    # save_parameters('x.params') writes what MXNet would read back: structural names in the NDArray-list layout
    out = tmp_path / "saved.params"
    donor.save_parameters(str(out))
    assert out.read_bytes() == pack_list(named, lambda a: pack_v2(a, ctx=(1, 0)))


def _write_png(path, arr):
    from PIL import Image
    Image.fromarray(arr).save(path)


def test_imagenet_reads_the_image_folder_layout(tmp_path, monkeypatch, capsys):
    rng = np.random.default_rng(2)
    root = tmp_path / "imagenet"
    want = []
    for li, wnid in enumerate(["n01440764", "n01443537", "n01484850"]):
        d = root / "val" / wnid
        d.mkdir(parents=True)
        for k in range(2):
            img = rng.integers(0, 256, (40 + 8 * k, 50, 3), dtype=np.uint8)
            name = "ILSVRC2012_val_%08d.%s" % (li * 2 + k, "png" if k else "PNG")
            _write_png(str(d / name), img)
            want.append((str(d / name), li, img))
    (root / "val" / "README.txt").write_text("not a class directory")
    (root / "val" / "n01440764" / "notes.txt").write_text("not an image")
    monkeypatch.setenv("FQ_IMAGENET_ROOT", str(root))
    ds = gdata.vision.ImageNet(train=False)
    assert ds.source == "disk:" + str(root / "val") and len(ds) == 6
    assert ds.synsets == ["n01440764", "n01443537", "n01484850"]
    assert sorted(ds.items) == sorted((p, l) for p, l, _ in want)
    by_path = {p: (l, img) for p, l, img in want}
    for i in range(len(ds)):
        img, label = ds[i]
        l, raw = by_path[ds.items[i][0]]
        assert label == l and img.dtype == np.uint8 and np.array_equal(img.asnumpy(), raw)
    assert "6 images in 3 classes" in capsys.readouterr().out
    # the CLI's transform chain on a decoded file, and the loader with decoding threads
    T = gdata.vision.transforms
    tf = T.Compose([T.Resize(32, keep_ratio=True), T.CenterCrop(24), T.ToTensor(), T.Normalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])])
    batches = list(gdata.DataLoader(ds.transform_first(tf), batch_size=4, num_workers=3, last_batch="keep"))
    assert [tuple(b[0].shape) for b in batches] == [(4, 3, 24, 24), (2, 3, 24, 24)]
    assert np.array_equal(np.concatenate([b[1].asnumpy() for b in batches]), [l for _, l in ds.items])
    serial = list(gdata.DataLoader(ds.transform_first(tf), batch_size=4, num_workers=0, last_batch="keep"))
    assert all(np.array_equal(a[0].asnumpy(), b[0].asnumpy()) for a, b in zip(batches, serial))
    # no train split on disk: synthetic, and it says so
    tr = gdata.vision.ImageNet(train=True)
    assert tr.source == "synthetic" and "not available" in capsys.readouterr().out


@pytest.mark.parametrize("kind", ["bin", "bin_subdir", "py"])
def test_cifar10_reads_the_batch_files(tmp_path, monkeypatch, kind):
    rng = np.random.default_rng(4)
    root = tmp_path / "cifar10"
    sub = {"bin": root, "bin_subdir": root / "cifar-10-batches-bin", "py": root / "cifar-10-batches-py"}[kind]
    sub.mkdir(parents=True)
    data, labels = {}, {}
    for name, n in [("data_batch_%d" % i, 7) for i in range(1, 6)] + [("test_batch", 9)]:
        chw = rng.integers(0, 256, (n, 3, 32, 32), dtype=np.uint8)
        lab = rng.integers(0, 10, n).astype(np.uint8)
        data[name], labels[name] = chw, lab
        if kind == "py":
            with open(sub / name, "wb") as f:
                pickle.dump({b"data": chw.reshape(n, 3072), b"labels": [int(v) for v in lab], b"batch_label": b"x"}, f)
        else:
            rec = np.concatenate([lab[:, None], chw.reshape(n, 3072)], axis=1).astype(np.uint8)
            rec.tofile(str(sub / (name + ".bin")))
    monkeypatch.setenv("FQ_CIFAR10_ROOT", str(root))
    test = gdata.vision.CIFAR10(train=False)
    assert test.source.startswith("disk:") and len(test) == 9 and test._label.dtype == np.int32
    assert np.array_equal(test._label, labels["test_batch"])
    assert np.array_equal(test._data.asnumpy(), data["test_batch"].transpose(0, 2, 3, 1))
    img, lab = test[3]
    assert tuple(img.shape) == (32, 32, 3) and lab == labels["test_batch"][3]
    train = gdata.vision.CIFAR10(train=True)
    assert len(train) == 35
    assert np.array_equal(train._label, np.concatenate([labels["data_batch_%d" % i] for i in range(1, 6)]))
    assert np.array_equal(train._data.asnumpy()[7:14], data["data_batch_2"].transpose(0, 2, 3, 1))
    # the reference reads the labels through the transformed dataset (simulate_quantization.py:281)
    T = gdata.vision.transforms
    wrapped = train.transform_first(T.Compose([T.ToTensor()]))
    assert wrapped._data._label is train._label and tuple(wrapped[0][0].shape) == (3, 32, 32)
    # an incomplete set is not silently half-read
    os.remove(str(sub / ("data_batch_3" + ("" if kind == "py" else ".bin"))))
    assert gdata.vision.CIFAR10(train=True).source == "synthetic"


def test_cli_on_disk_dataset_and_params_file_equals_the_in_memory_run(tmp_path, monkeypatch, capsys):
    """The CLI on a tiny on-disk CIFAR-10 + a `.params` checkpoint in MXNet's layout gives the accuracy of the same images
    and weights handed over in memory, and its result block names both sources."""
    import importlib.util
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("fq_cli_ri", os.path.join(ROOT, "examples", "simulate_quantization.py"))
    cli = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cli)
    rng = np.random.default_rng(6)
    root = tmp_path / "cifar10"
    root.mkdir()
    n = 24
    chw = rng.integers(0, 256, (n, 3, 32, 32), dtype=np.uint8)
    lab = (np.arange(n) % 10).astype(np.uint8)
    np.concatenate([lab[:, None], chw.reshape(n, 3072)], axis=1).astype(np.uint8).tofile(str(root / "test_batch.bin"))
    for i in range(1, 6):
        np.concatenate([lab[:, None], chw.reshape(n, 3072)], axis=1).astype(np.uint8).tofile(str(root / ("data_batch_%d.bin" % i)))
    np.random.seed(21)
    donor = get_model("cifar_resnet20_v1", classes=10)
    home = tmp_path / "mxhome"
    (home / "models").mkdir(parents=True)
    ckpt = home / "models" / "cifar_resnet20_v1-deadbeef.params"
    ckpt.write_bytes(pack_list([(k, p.data().asnumpy()) for k, p in donor._collect_params_with_prefix().items()]))
    monkeypatch.setenv("FQ_CIFAR10_ROOT", str(root))
    monkeypatch.setenv("MXNET_HOME", str(home))
    args = ["--model", "cifar_resnet20_v1", "--dataset", "cifar10", "--batch-size", "8", "--num-sample", "1",
            "--quantize-input-offline", "--calib-epoch", "1"]
    with oracle_ops():
        acc, avg_acc, net = cli.run(cli.parse_args(args), mx.cpu())
    out = capsys.readouterr().out
    assert "data    : disk:" + str(root) in out and "weights : " + str(ckpt) in out
    assert "24 images from" in out and "parameters from " + str(ckpt) in out
    # the same run with the weights as an npz of this package and the images as the synthetic-equivalent arrays in memory:
    # (a) weights - the npz checkpoint gives the same accuracy
    npz = tmp_path / "w.params.npz"
    donor.save_parameters(str(npz))
    with oracle_ops():
        acc2, avg2, _ = cli.run(cli.parse_args(args + ["--pretrained", str(npz)]), mx.cpu())
    assert (acc2, avg2) == (acc, avg_acc)
    # (b) data - the evaluation counters recomputed here from the decoded arrays and the loaded net
    T = gdata.vision.transforms
    tf = T.Compose([T.ToTensor(), T.Normalize([0.4914, 0.4822, 0.4465], [0.2023, 0.1994, 0.2010])])
    x = mx.nd.array(np.stack([tf(mx.nd.array(im, dtype="uint8")).asnumpy() for im in chw.transpose(0, 2, 3, 1)]))
    with oracle_ops():
        pred = net(x).asnumpy().argmax(axis=1)
    assert acc == pytest.approx(float((pred == lab).mean()))
