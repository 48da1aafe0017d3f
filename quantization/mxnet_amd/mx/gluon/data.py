"""`gluon.data` surface of the CLI (examples/simulate_quantization.py:24-30,151-175,257-292): `Sampler`, `DataLoader`,
`vision.transforms`, and ImageNet / CIFAR10 datasets.

ImageNet (gluoncv's ImageFolder layout) and CIFAR-10 (MXNet's binary batches or the python pickles) are read from disk
when their root holds them.  No dataset can be downloaded here; when the root holds no data the vision datasets are
SYNTHETIC: deterministic uint8 images of the final crop size (seeded per index) with uniformly cycling labels, so the
evaluation and calibration loops, the `UniformSampler`, and rank-sharding run end to end.  Accuracy on them is
meaningless; every dataset says where its images came from (`source`) and the CLI prints it.
"""
import os

import numpy as np
import torch

from ..ndarray import NDArray

__all__ = ["Dataset", "SimpleDataset", "ArrayDataset", "Sampler", "SequentialSampler", "RandomSampler",
           "BatchSampler", "DataLoader", "vision"]


class Dataset(object):
    def __getitem__(self, idx):
        raise NotImplementedError

    def __len__(self):
        raise NotImplementedError

    def transform(self, fn, lazy=True):
        return _LazyTransformDataset(self, fn)

    def transform_first(self, fn, lazy=True):
        return self.transform(_TransformFirstClosure(fn), lazy)


class SimpleDataset(Dataset):
    def __init__(self, data):
        self._data = data

    def __len__(self):
        return len(self._data)

    def __getitem__(self, idx):
        return self._data[idx]


class ArrayDataset(Dataset):
    def __init__(self, *args):
        self._length = len(args[0])
        self._data = list(args)

    def __getitem__(self, idx):
        if len(self._data) == 1:
            return self._data[0][idx]
        return tuple(d[idx] for d in self._data)

    def __len__(self):
        return self._length


class _LazyTransformDataset(Dataset):
    def __init__(self, data, fn):
        self._data = data
        self._fn = fn

    def __len__(self):
        return len(self._data)

    def __getitem__(self, idx):
        item = self._data[idx]
        if isinstance(item, tuple):
            return self._fn(*item)
        return self._fn(item)


class _TransformFirstClosure(object):
    def __init__(self, fn):
        self._fn = fn

    def __call__(self, x, *args):
        if args:
            return (self._fn(x),) + args
        return self._fn(x)


class Sampler(object):
    def __iter__(self):
        raise NotImplementedError

    def __len__(self):
        raise NotImplementedError


class SequentialSampler(Sampler):
    def __init__(self, length):
        self._length = length

    def __iter__(self):
        return iter(range(self._length))

    def __len__(self):
        return self._length


class RandomSampler(Sampler):
    def __init__(self, length):
        self._length = length

    def __iter__(self):
        indices = np.arange(self._length)
        np.random.shuffle(indices)
        return iter(indices)

    def __len__(self):
        return self._length


class BatchSampler(Sampler):
    def __init__(self, sampler, batch_size, last_batch="keep"):
        self._sampler, self._batch_size, self._last_batch = sampler, batch_size, last_batch
        self._prev = []

    def __iter__(self):
        batch, self._prev = self._prev, []
        for i in self._sampler:
            batch.append(i)
            if len(batch) == self._batch_size:
                yield batch
                batch = []
        if batch:
            if self._last_batch == "keep":
                yield batch
            elif self._last_batch == "discard":
                return
            elif self._last_batch == "rollover":
                self._prev = batch
            else:
                raise ValueError("last_batch must be one of 'keep', 'discard', or 'rollover'")

    def __len__(self):
        n = len(self._sampler)
        if self._last_batch == "keep":
            return (n + self._batch_size - 1) // self._batch_size
        if self._last_batch == "discard":
            return n // self._batch_size
        return (len(self._prev) + n) // self._batch_size


def _batchify(items):
    first = items[0]
    if isinstance(first, tuple):
        return tuple(_batchify([it[i] for it in items]) for i in range(len(first)))
    if isinstance(first, NDArray):
        return NDArray(torch.stack([it._t for it in items], dim=0))
    arr = np.asarray(items)
    if arr.dtype == np.float64:
        arr = arr.astype(np.float32)
    return NDArray(torch.from_numpy(arr))


def _base_dataset(ds):
    """The dataset under `transform` / `transform_first` wrappers."""
    while isinstance(ds, _LazyTransformDataset):
        ds = ds._data
    return ds


class DataLoader(object):
    """Loader in the calling process; `num_workers` threads decode a batch's image files side by side when the dataset is
    read from disk (synthetic data costs nothing to produce: no threads).

    `rank`/`world_size` (new; the reference is single-device) stride the batch LIST across one-process-per-GPU
    ranks: every rank draws the identical sampler sequence (same numpy seed), then keeps batches i with
    i % world_size == rank.
    """

    def __init__(self, dataset, batch_size=None, shuffle=False, sampler=None, last_batch=None, batch_sampler=None,
                 batchify_fn=None, num_workers=0, rank=0, world_size=1, **_ignored):
        self._dataset = dataset
        if batch_sampler is None:
            if batch_size is None:
                raise ValueError("batch_size must be specified unless batch_sampler is specified")
            if sampler is None:
                sampler = RandomSampler(len(dataset)) if shuffle else SequentialSampler(len(dataset))
            elif shuffle:
                raise ValueError("shuffle must not be specified if sampler is specified")
            batch_sampler = BatchSampler(sampler, batch_size, last_batch if last_batch else "keep")
        self._batch_sampler = batch_sampler
        self._batchify_fn = batchify_fn or _batchify
        self._rank, self._world = rank, world_size
        # worker threads only where an item costs something to make (files on disk); synthetic items are seeded arrays
        self._workers = int(num_workers) if getattr(_base_dataset(dataset), "source", "synthetic") != "synthetic" else 0

    def __iter__(self):
        pool = None
        if self._workers > 0:
            # decoding an image file releases the interpreter lock (PIL): `num_workers` threads fetch a batch's items side by side
            from concurrent.futures import ThreadPoolExecutor
            pool = ThreadPoolExecutor(max_workers=self._workers)
        try:
            for i, batch in enumerate(self._batch_sampler):
                if i % self._world != self._rank:
                    continue
                if pool is not None:
                    items = list(pool.map(lambda idx: self._dataset[int(idx)], batch))
                else:
                    items = [self._dataset[int(idx)] for idx in batch]
                yield self._batchify_fn(items)
        finally:
            if pool is not None:
                pool.shutdown(wait=True)

    def __len__(self):
        n = len(self._batch_sampler)
        return (n - self._rank + self._world - 1) // self._world


# ----------------------------------------------------------------------------------------------------
class _Transforms(object):
    """`gluon.data.vision.transforms` subset used by the CLI (`:259-269`)."""

    class Compose(object):
        def __init__(self, transforms):
            self._transforms = transforms

        def __call__(self, x):
            for t in self._transforms:
                x = t(x)
            return x

    class Resize(object):
        def __init__(self, size, keep_ratio=False, interpolation=1):
            self._size, self._keep = size, keep_ratio

        def __call__(self, x):
            t = x._t
            h, w = t.shape[0], t.shape[1]
            if isinstance(self._size, int):
                if self._keep:
                    if h < w:
                        nh, nw = self._size, int(round(w * self._size / h))
                    else:
                        nh, nw = int(round(h * self._size / w)), self._size
                else:
                    nh = nw = self._size
            else:
                nw, nh = self._size
            if (nh, nw) == (h, w):
                return x
            y = torch.nn.functional.interpolate(t.permute(2, 0, 1)[None].float(), size=(nh, nw), mode="bilinear",
                                                align_corners=False)[0].permute(1, 2, 0)
            return NDArray(y.round().clamp(0, 255).to(torch.uint8))

    class CenterCrop(object):
        def __init__(self, size, interpolation=1):
            self._size = (size, size) if isinstance(size, int) else size

        def __call__(self, x):
            t = x._t
            h, w = t.shape[0], t.shape[1]
            cw, ch = self._size
            if (h, w) == (ch, cw) or h < ch or w < cw:
                return x
            y0, x0 = (h - ch) // 2, (w - cw) // 2
            return NDArray(t[y0:y0 + ch, x0:x0 + cw].contiguous())

    class ToTensor(object):
        def __call__(self, x):
            return NDArray((x._t.permute(2, 0, 1).float() / 255.0).contiguous())

    class Normalize(object):
        def __init__(self, mean=0.0, std=1.0):
            self._mean = torch.tensor(np.asarray(mean, dtype=np.float32)).reshape(-1, 1, 1)
            self._std = torch.tensor(np.asarray(std, dtype=np.float32)).reshape(-1, 1, 1)

        def __call__(self, x):
            return NDArray((x._t - self._mean) / self._std)

    class Cast(object):
        def __init__(self, dtype="float32"):
            self._dtype = dtype

        def __call__(self, x):
            return x.astype(self._dtype)


class _Items(object):
    """Lazy `[(path, label), ...]` list (`train_dataset._data.items`, simulate_quantization.py:279)."""

    def __init__(self, labels):
        self._labels = labels

    def __len__(self):
        return len(self._labels)

    def __getitem__(self, i):
        return ("synthetic/%08d.jpg" % i, int(self._labels[i]))

    def __iter__(self):
        for i in range(len(self._labels)):
            yield self[i]


class _SyntheticImages(Dataset):
    def __init__(self, n, classes, hw, seed):
        self._n, self._classes, self._hw, self._seed = n, classes, hw, seed
        self._label = (np.arange(n) * 7919 % classes).astype(np.int32)     # every class appears n/classes times
        self.items = _Items(self._label)
        self.synsets = ["class%d" % i for i in range(classes)]

    def __len__(self):
        return self._n

    def __getitem__(self, idx):
        rng = np.random.default_rng(self._seed * 1000003 + int(idx))
        img = rng.integers(0, 256, size=(self._hw, self._hw, 3), dtype=np.uint8)
        return NDArray(torch.from_numpy(img)), int(self._label[idx])


def _synthetic_count(train, classes, default_per_class):
    env = os.environ.get("FQ_SYNTH_TRAIN_PER_CLASS" if train else "FQ_SYNTH_VAL_IMAGES")
    if env:
        return int(env) * (classes if train else 1)
    return default_per_class * classes if train else None


def _datasets_dir(name, root=None, env=None):
    """`root`, else $<env>, else MXNet's `~/.mxnet/datasets/<name>` ($MXNET_HOME/datasets/<name> when that is set, as
    `mxnet.base.data_dir()` has it)."""
    if root is None:
        root = os.environ.get(env) if env else None
    if root is None:
        root = os.path.join(os.environ.get("MXNET_HOME", os.path.join("~", ".mxnet")), "datasets", name)
    return os.path.expanduser(root)


def _decode_image(path):
    """An image file as an (H, W, 3) uint8 array, RGB - what `mx.image.imread(path, flag=1)` hands the transforms."""
    from PIL import Image
    with Image.open(path) as im:
        return np.array(im.convert("RGB"), dtype=np.uint8)          # (a writable copy)


class ImageFolderDataset(Dataset):
    """`mxnet.gluon.data.vision.ImageFolderDataset`: `root/<category>/<image>`; categories are the sorted sub-directories
    (`synsets`), `items` the sorted (path, label) list; an item is (H x W x 3 uint8 NDArray, label)."""
    _exts = (".jpg", ".jpeg", ".png")

    def __init__(self, root, flag=1, transform=None):
        self._root = os.path.expanduser(root)
        self._flag = flag
        self._transform = transform
        self.synsets, self.items = [], []
        for folder in sorted(os.listdir(self._root)):
            path = os.path.join(self._root, folder)
            if not os.path.isdir(path):
                continue
            label = len(self.synsets)
            self.synsets.append(folder)
            for filename in sorted(os.listdir(path)):
                if os.path.splitext(filename)[1].lower() in self._exts:
                    self.items.append((os.path.join(path, filename), label))

    def __len__(self):
        return len(self.items)

    def __getitem__(self, idx):
        img = NDArray(torch.from_numpy(_decode_image(self.items[idx][0])))
        label = self.items[idx][1]
        if self._transform is not None:
            return self._transform(img, label)
        return img, label


def _has_class_dirs(path):
    try:
        return any(os.path.isdir(os.path.join(path, d)) for d in os.listdir(path))
    except OSError:
        return False


class _ImageNet(Dataset):
    """`gluoncv.data.ImageNet(root='~/.mxnet/datasets/imagenet', train=True, transform=None)` (reference
    examples/simulate_quantization.py:272-279): the ImageFolder layout gluoncv's set-up script leaves - `root/train/<wnid>/*.JPEG`,
    `root/val/<wnid>/*.JPEG` - read from disk when that split exists (root: the argument, $FQ_IMAGENET_ROOT, or MXNet's
    default).  Without it: SYNTHETIC images of the crop size (there is no network to fetch the dataset) - `source` says which,
    and the CLI prints it beside every result."""

    def __init__(self, root=None, train=True, transform=None):
        base = _datasets_dir("imagenet", root, "FQ_IMAGENET_ROOT")
        split = os.path.join(base, "train" if train else "val")
        if _has_class_dirs(split):
            self._impl = ImageFolderDataset(split, 1, transform)
            self.source = "disk:" + split
            print("[data] ImageNet %s: %d images in %d classes from %s"
                  % ("train" if train else "val", len(self._impl), len(self._impl.synsets), split))
        else:
            n = _synthetic_count(train, 1000, 6) or 2048
            self._impl = _SyntheticImages(n, 1000, 224, 7 if train else 77)
            self.source = "synthetic"
            print("[data] ImageNet not available (no network / no %s): %d synthetic %s images"
                  % (split, n, "train" if train else "val"))
        self.items, self.synsets = self._impl.items, self._impl.synsets

    def __len__(self):
        return len(self._impl)

    def __getitem__(self, idx):
        return self._impl[idx]


def _cifar10_files(base, train):
    """The batch files of one split, as (kind, [paths]): MXNet's binary records (`data_batch_1..5.bin` / `test_batch.bin`, in
    the root or in `cifar-10-batches-bin/`) or the python pickles of `cifar-10-batches-py/`; None when neither is complete."""
    names = ["data_batch_%d" % i for i in range(1, 6)] if train else ["test_batch"]
    for sub, ext, kind in (("", ".bin", "bin"), ("cifar-10-batches-bin", ".bin", "bin"), ("cifar-10-batches-py", "", "py"),
                           ("", "", "py")):
        paths = [os.path.join(base, sub, nme + ext) for nme in names]
        if all(os.path.isfile(q) for q in paths):
            return kind, paths
    return None


def _read_cifar_batch(kind, path):
    if kind == "bin":           # mxnet/gluon/data/vision/datasets.py CIFAR10._read_batch: 1 label byte + 3 x 32 x 32 bytes, CHW
        rec = np.fromfile(path, dtype=np.uint8).reshape(-1, 3072 + 1)
        return rec[:, 1:].reshape(-1, 3, 32, 32).transpose(0, 2, 3, 1), rec[:, 0].astype(np.int32)
    import pickle
    with open(path, "rb") as f:
        d = pickle.load(f, encoding="bytes")
    data = np.asarray(d[b"data"], dtype=np.uint8).reshape(-1, 3, 32, 32).transpose(0, 2, 3, 1)
    return data, np.asarray(d[b"labels"], dtype=np.int32)


class _CIFAR10(Dataset):
    """`mxnet.gluon.data.vision.CIFAR10(root='~/.mxnet/datasets/cifar10', train=True, transform=None)`: the batch files read
    from disk when they are there (root: the argument, $FQ_CIFAR10_ROOT, or MXNet's default) into `_data` (N x 32 x 32 x 3
    uint8) and `_label` (int32; the reference's UniformSampler reads it, simulate_quantization.py:281); synthetic otherwise."""

    def __init__(self, root=None, train=True, transform=None):
        base = _datasets_dir("cifar10", root, "FQ_CIFAR10_ROOT")
        found = _cifar10_files(base, train)
        self._transform = transform
        if found is not None:
            parts = [_read_cifar_batch(found[0], q) for q in found[1]]
            self._data = NDArray(torch.from_numpy(np.ascontiguousarray(np.concatenate([a for a, _ in parts]))))
            self._label = np.concatenate([l for _, l in parts])
            self._impl = None
            self.source = "disk:" + os.path.dirname(found[1][0])
            print("[data] CIFAR10 %s: %d images from %s" % ("train" if train else "test", len(self._label), self.source[5:]))
        else:
            n = _synthetic_count(train, 10, 100) or 2000
            self._impl = _SyntheticImages(n, 10, 32, 11 if train else 111)
            self._label = self._impl._label
            self.source = "synthetic"
            print("[data] CIFAR10 not available (no network / no batch files under %s): %d synthetic %s images"
                  % (base, n, "train" if train else "val"))
        self.items = _Items(self._label)
        self.synsets = ["airplane", "automobile", "bird", "cat", "deer", "dog", "frog", "horse", "ship", "truck"]

    def __len__(self):
        return len(self._label)

    def __getitem__(self, idx):
        if self._impl is not None:
            item = self._impl[idx]
        else:
            item = (NDArray(self._data._t[idx]), int(self._label[idx]))
        if self._transform is not None:
            return self._transform(*item)
        return item


class _Vision(object):
    transforms = _Transforms
    ImageNet = _ImageNet
    CIFAR10 = _CIFAR10
    ImageFolderDataset = ImageFolderDataset


vision = _Vision
