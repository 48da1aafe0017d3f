"""MXNet's NDArray-list file format - what `mx.nd.save` / `mx.nd.load`, `Block.save_parameters` / `load_parameters` and the
gluoncv model zoo's `.params` checkpoints use (`get_model(..., pretrained=True)`, reference
examples/simulate_quantization.py:188-204).  MXNet itself is a third-party dependency that is absent from the reference tree
(requirements: mxnet 1.x); the layout below restates its published serialisation (mxnet 1.x `src/ndarray/ndarray.cc`:
`NDArray::Save` / `NDArray::Load` and `MXNDArraySave`'s list wrapper), little-endian throughout:

    uint64  0x112                       list magic (kMXAPINDArrayListMagic)
    uint64  0                           reserved
    uint64  n                           number of arrays, then n records:
        uint32  0xF993FAC9              NDARRAY_V2_MAGIC   (0xF993FACA = V3: same layout, numpy shape semantics;
                                                            0xF993FAC8 = V1: no storage-type word)
        int32   storage type            0 = dense (sparse arrays are refused here)
        uint32  ndim ; int64 dims[ndim] shape            (files older than V1 start with ndim itself and hold uint32 dims)
        int32   dev_type ; int32 dev_id context the array was saved from (1 = cpu; ignored on load)
        int32   type flag               0 float32, 1 float64, 2 float16, 3 uint8, 4 int32, 5 int8, 6 int64
        raw data, C order
    uint64  m                           number of names (0, or n), then m records:  uint64 length ; bytes

A list saved from a dict carries the names; from a list, m = 0.  Checkpoints written by `Module` / `export` prefix every name
with `arg:` or `aux:`; Gluon strips those on load and so does `load_params` below.
"""
import struct

import numpy as np

__all__ = ["save", "load", "load_params", "is_ndarray_file"]

LIST_MAGIC = 0x112
V1_MAGIC, V2_MAGIC, V3_MAGIC = 0xF993FAC8, 0xF993FAC9, 0xF993FACA
_TYPE_FLAGS = {0: np.float32, 1: np.float64, 2: np.float16, 3: np.uint8, 4: np.int32, 5: np.int8, 6: np.int64}
_FLAG_OF = {np.dtype(v): k for k, v in _TYPE_FLAGS.items()}


def is_ndarray_file(path):
    """True when `path` starts with the list magic (a `.params` file), False for anything else (e.g. the npz files this
    package writes for `*.npz` names)."""
    try:
        with open(path, "rb") as f:
            head = f.read(8)
    except OSError:
        return False
    return len(head) == 8 and struct.unpack("<Q", head)[0] == LIST_MAGIC


class _Reader(object):
    def __init__(self, blob, name):
        self._b, self._p, self._name = blob, 0, name

    def take(self, fmt):
        size = struct.calcsize(fmt)
        if self._p + size > len(self._b):
            raise ValueError("%s: truncated NDArray file (wanted %d bytes at offset %d of %d)"
                             % (self._name, size, self._p, len(self._b)))
        out = struct.unpack_from(fmt, self._b, self._p)
        self._p += size
        return out if len(out) > 1 else out[0]

    def raw(self, size):
        if self._p + size > len(self._b):
            raise ValueError("%s: truncated NDArray file (array data of %d bytes at offset %d of %d)"
                             % (self._name, size, self._p, len(self._b)))
        out = self._b[self._p:self._p + size]
        self._p += size
        return out


def _read_array(r):
    magic = r.take("<I")
    if magic in (V2_MAGIC, V3_MAGIC):
        stype = r.take("<i")
        if stype != 0:
            raise ValueError("sparse NDArray (storage type %d) in a parameter file: not supported" % stype)
        ndim = r.take("<I")
        shape = tuple(r.take("<%dq" % ndim)) if ndim > 1 else ((r.take("<q"),) if ndim == 1 else ())
    elif magic == V1_MAGIC:
        ndim = r.take("<I")
        shape = tuple(r.take("<%dq" % ndim)) if ndim > 1 else ((r.take("<q"),) if ndim == 1 else ())
    else:                                   # the oldest layout: the word just read IS ndim, dims are uint32
        ndim = magic
        if ndim > 32:
            raise ValueError("not an NDArray record (magic 0x%08X)" % magic)
        shape = tuple(r.take("<%dI" % ndim)) if ndim > 1 else ((r.take("<I"),) if ndim == 1 else ())
    if ndim == 0 and magic != V3_MAGIC:
        return None                         # (an empty NDArray: nothing follows it)
    r.take("<ii")                           # context: where it was saved from
    flag = r.take("<i")
    if flag not in _TYPE_FLAGS:
        raise ValueError("unknown NDArray type flag %d" % flag)
    dt = np.dtype(_TYPE_FLAGS[flag]).newbyteorder("<")
    count = int(np.prod(shape, dtype=np.int64)) if shape else 1
    return np.frombuffer(r.raw(count * dt.itemsize), dtype=dt).reshape(shape).copy()


def load(fname):
    """`mx.nd.load`: a dict {name: numpy array} when the file carries names, else a list of numpy arrays."""
    with open(fname, "rb") as f:
        blob = f.read()
    r = _Reader(blob, fname)
    if len(blob) < 24 or r.take("<Q") != LIST_MAGIC:
        raise ValueError("%s is not an MXNet NDArray file (no 0x112 list magic)" % fname)
    r.take("<Q")
    n = r.take("<Q")
    arrays = [_read_array(r) for _ in range(n)]
    m = r.take("<Q")
    if m not in (0, n):
        raise ValueError("%s: %d names for %d arrays" % (fname, m, n))
    names = []
    for _ in range(m):
        ln = r.take("<Q")
        names.append(r.raw(ln).decode("utf-8"))
    return dict(zip(names, arrays)) if m else arrays


def save(fname, data):
    """`mx.nd.save`: `data` is a dict {name: array} or a list of arrays (anything `np.asarray` takes; NDArrays through
    `.asnumpy()`).  Written as V2 dense records saved from cpu(0)."""
    if isinstance(data, dict):
        names, arrays = list(data.keys()), list(data.values())
    else:
        names, arrays = [], list(data)
    out = [struct.pack("<QQQ", LIST_MAGIC, 0, len(arrays))]
    for a in arrays:
        a = a.asnumpy() if hasattr(a, "asnumpy") else np.asarray(a)
        if a.dtype not in _FLAG_OF:
            raise TypeError("dtype %s has no MXNet type flag" % a.dtype)
        a = np.ascontiguousarray(a, dtype=a.dtype.newbyteorder("<"))
        out.append(struct.pack("<IiI", V2_MAGIC, 0, a.ndim))
        out.append(struct.pack("<%dq" % a.ndim, *a.shape))
        out.append(struct.pack("<iii", 1, 0, _FLAG_OF[np.dtype(a.dtype.name)]))
        out.append(a.tobytes())
    out.append(struct.pack("<Q", len(names)))
    for nme in names:
        raw = nme.encode("utf-8")
        out.append(struct.pack("<Q", len(raw)))
        out.append(raw)
    with open(fname, "wb") as f:
        f.write(b"".join(out))


def load_params(fname):
    """{name: array} of a parameter file with the `arg:` / `aux:` prefixes of exported checkpoints stripped (Gluon's
    `ParameterDict.load` does the same)."""
    loaded = load(fname)
    if not isinstance(loaded, dict):
        raise ValueError("%s holds an unnamed NDArray list, not parameters" % fname)
    return {(k[4:] if k.startswith(("arg:", "aux:")) else k): v for k, v in loaded.items()}
