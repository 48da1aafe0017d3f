"""Python entry points over the C ABI: raw device pointers of torch-ROCm tensors + torch's current HIP stream.

Each function names the reference op chain it stands in for.  Inputs must be fp32, contiguous and resident on a
HIP device; anything else raises (there is deliberately no CPU path here — the oracle under oracle/ is test
infrastructure and is never imported by this package).
"""
import ctypes
import threading

import numpy as np
import torch

from . import _lib
from ._lib import FakeQuantError

__all__ = ["comm_unique_id", "comm_init", "comm_world", "comm_allreduce", "comm_destroy", "add_act_stat", "bn_act_maxpool_stat", "stat_rows_sum", "mean_from_sums", "fake_quant_online_prestat", "bn_act_stat", "stem_conv3x3s2", "stem_conv_s2", "stem_conv_supported", "eval_counters", "dense_i8_eval", "gemm_i8_codes", "global_avg_pool_stat", "dwconv3x3", "dwconv3x3_c16", "weight_codes", "pwconv_i8", "pwconv_strided_supported", "weight_codes_3x3", "weight_slices_3x3", "conv3x3_i8", "Codes16", "batch_mean_rows", "batch_mean_gathered", "ste_forward", "absmax_per_sample", "batch_mean", "fake_quant_online", "fake_quant_offline", "weight_fake_quant",
           "wino_weight_fake_quant", "ema_update", "global_max", "histogram_accumulate", "hist_to_float",
           "kl_search", "quantize_codes", "dequantize", "qconv_kind", "qconv_weights", "qconv_workspace", "qconv2d",
           "batches_in_flight", "in_flight", "winograd_matrices", "device_info", "act_flags"]

_WS = {}


def _lib_():
    return _lib.LIB


_DEVICE_TLS = threading.local()


def _stream(t):
    """HIP stream handle for a launch on `t`'s device.  The library launches on whatever device is CURRENT for the
    calling thread (kernels, memsets and the cached compute-unit count follow it), so the tensor's device is made current
    here — every entry point evaluates `_stream(x)` as an argument of its C call, i.e. before the call happens — and
    `check_call`, which every entry point wraps around that call, puts the caller's device back.  A process that only
    ever uses one GPU (the normal case: one process per GPU) never takes the branch."""
    dev = t.device
    now = torch.cuda.current_device()
    if dev.index is not None and dev.index != now:
        if getattr(_DEVICE_TLS, "restore", None) is None:
            _DEVICE_TLS.restore = now
        torch.cuda.set_device(dev)
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def check_call(ret):
    """Status check of _lib.check_call + the device switch of `_stream` undone (the launch is already enqueued)."""
    prev = getattr(_DEVICE_TLS, "restore", None)
    if prev is not None:
        _DEVICE_TLS.restore = None
        torch.cuda.set_device(prev)
    _lib.check_call(ret)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _check(t, name, dtype=torch.float32):
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch tensor (got %s)" % (name, type(t)))
    if not t.is_cuda:
        raise FakeQuantError("%s lives on %s: the fake-quant path runs on a HIP device only (no CPU fallback)"
                             % (name, t.device))
    if t.dtype != dtype:
        raise TypeError("%s must be %s (got %s)" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)
    return t


def require_hip(device, what="tensor"):
    if torch.device(device).type != "cuda":
        raise FakeQuantError("%s lives on %s: the fake-quant path runs on a HIP device only (no CPU fallback)"
                             % (what, device))


def default_device(what="this call"):
    if not torch.cuda.is_available():
        raise FakeQuantError("%s needs a HIP device (no CPU fallback)" % what)
    return torch.device("cuda", torch.cuda.current_device())


def _workspace(dev, nbytes):
    """Stream-ordered scratch: one growing buffer per (device, stream)."""
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 16), dtype=torch.uint8, device=dev)
        _WS[key] = buf
    return buf


def device_info():
    arch = ctypes.create_string_buffer(64)
    cu, wf = ctypes.c_int(0), ctypes.c_int(0)
    check_call(_lib_().fq_device_info(arch, 64, ctypes.byref(cu), ctypes.byref(wf)))
    return {"arch": arch.value.decode(), "compute_units": cu.value, "wavefront": wf.value}


KERNEL_IDS = {"stat": 0, "apply_online": 1, "apply_offline": 2, "weight": 3, "histogram": 4, "bn_act": 5,
              "dwconv": 6, "pwconv": 7, "stem": 8, "pool": 9, "global_max": 10, "conv3x3": 11, "dense": 12}


def profile_enable(on=True):
    """Bracket every streaming-kernel launch with HIP events on its launch stream (include/fakequant.h)."""
    check_call(_lib_().fq_profile_enable(1 if on else 0))


def profile_reset():
    check_call(_lib_().fq_profile_reset())


def profile_read():
    """{kernel: {"ms": total, "launches": n, "bytes": algorithmic bytes, "bytes_moved": bytes really moved (1 B per element of
    a C16 code tensor)}} since the last reset (synchronises)."""
    out = {}
    for name, kid in KERNEL_IDS.items():
        ms, n, b = ctypes.c_double(0), ctypes.c_int64(0), ctypes.c_double(0)
        check_call(_lib_().fq_profile_read(kid, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(b)))
        mv = ctypes.c_double(0)
        check_call(_lib_().fq_profile_read_moved(kid, ctypes.byref(mv)))
        out[name] = {"ms": ms.value, "launches": n.value, "bytes": b.value, "bytes_moved": mv.value}
    return out


def profile_event_overhead_ms(device=None, repeats=200):
    """(overhead_ms, null_kernel_ms): the fixed cost a bracketing event pair adds to a launch = median pair around a
    trivial kernel minus that kernel's own back-to-back cost, both measured now on this device (fq_profile_calibrate)."""
    device = default_device() if device is None else device
    scratch = torch.zeros(4, dtype=torch.float32, device=device)
    pair, null = ctypes.c_double(0), ctypes.c_double(0)
    check_call(_lib_().fq_profile_calibrate(_ptr(scratch), int(repeats), ctypes.byref(pair), ctypes.byref(null),
                                            _stream(scratch)))
    return max(pair.value - null.value, 0.0), null.value


def profile_launch_overhead_ms(device=None, repeats=200, spin_us=30.0):
    """(overhead_ms, spin_ms): what an event pair measures beyond a kernel's own begin -> end time, measured now on this
    device with a kernel that times itself on the wall clock (fq_profile_launch_overhead)."""
    device = default_device() if device is None else device
    scratch = torch.zeros(4 * int(repeats), dtype=torch.float32, device=device)
    over, own = ctypes.c_double(0), ctypes.c_double(0)
    check_call(_lib_().fq_profile_launch_overhead(_ptr(scratch), int(repeats), float(spin_us), ctypes.byref(over),
                                                  ctypes.byref(own), _stream(scratch)))
    return max(over.value, 0.0), own.value


def act_flags(signed=False, lo_neg_max=None, no_abs=False, no_eps=False):
    """`lo_neg_max` defaults to `signed` (conv: convert_conv2d.py:59-63); Dense passes False (convert_dense.py:49)."""
    if lo_neg_max is None:
        lo_neg_max = signed
    f = 0
    if signed:
        f |= _lib.FQ_ACT_SIGNED
    if lo_neg_max:
        f |= _lib.FQ_ACT_LO_NEG_MAX
    if no_abs:
        f |= _lib.FQ_ACT_NO_ABS
    if no_eps:
        f |= _lib.FQ_ACT_NO_EPS
    return f


def _n_inner(x):
    if x.dim() < 1 or x.numel() == 0:
        raise ValueError("empty activation tensor %s" % (tuple(x.shape),))
    n = x.shape[0]
    return n, x.numel() // n


def absmax_per_sample(x, no_abs=False, out=None):
    """`F.max(F.abs(x), axis=(1,2,3))` (convert_conv2d.py:56) -> (N,) device tensor."""
    _check(x, "x")
    n, inner = _n_inner(x)
    out = torch.empty(n, dtype=torch.float32, device=x.device) if out is None else _check(out, "out")
    check_call(_lib_().fq_absmax_per_sample(_ptr(x), n, inner, act_flags(no_abs=no_abs), _ptr(out), _stream(x)))
    return out


def batch_mean(v, out=None):
    """`.mean()` of the per-sample maxima (convert_conv2d.py:56) -> (1,) device tensor."""
    _check(v, "v")
    out = torch.empty(1, dtype=torch.float32, device=v.device) if out is None else _check(out, "out")
    check_call(_lib_().fq_batch_mean(_ptr(v), v.numel(), _ptr(out), _stream(v)))
    return out


def batch_mean_gathered(packs, out=None):
    """Batch mean over all-gathered per-rank records {n_w, v_w[0..n_w)} in rank order (dist.py); packs: (W, stride)."""
    _check(packs, "packs")
    if packs.dim() != 2:
        raise ValueError("batch_mean_gathered wants a (world, stride) tensor")
    out = torch.empty(1, dtype=torch.float32, device=packs.device) if out is None else _check(out, "out")
    check_call(_lib_().fq_batch_mean_gathered(_ptr(packs), packs.shape[0], packs.shape[1], _ptr(out), _stream(packs)))
    return out


def stat_rows_sum(stats, n, out=None):
    """One rank's record for the calibration-step collective: per layer the fp64 sum of its first `n` per-sample maxima,
    then `n` itself.  stats: (L, width) fp32 -> (L + 1,) fp64."""
    _check(stats, "stats")
    rows, width = stats.shape
    if out is None:
        out = torch.empty(rows + 1, dtype=torch.float64, device=stats.device)
    _check(out, "out", torch.float64)
    check_call(_lib_().fq_stat_rows_sum(_ptr(stats), rows, int(n), width, _ptr(out), _stream(stats)))
    return out


def mean_from_sums(sums, out=None):
    """(L + 1,) fp64 [sum_0 .. sum_{L-1}, count] (summed over the ranks) -> (L,) fp32 batch means of the global batch."""
    _check(sums, "sums", torch.float64)
    rows = sums.numel() - 1
    if out is None:
        out = torch.empty(rows, dtype=torch.float32, device=sums.device)
    _check(out, "out")
    check_call(_lib_().fq_mean_from_sums(_ptr(sums), rows, _ptr(out), _stream(sums)))
    return out


def batch_mean_rows(v, out=None):
    """Row-wise `batch_mean`: v (rows, n) contiguous -> (rows,)."""
    _check(v, "v")
    if v.dim() != 2:
        raise ValueError("batch_mean_rows wants a 2-D tensor")
    out = torch.empty(v.shape[0], dtype=torch.float32, device=v.device) if out is None else _check(out, "out")
    check_call(_lib_().fq_batch_mean_rows(_ptr(v), v.shape[0], v.shape[1], v.stride(0), _ptr(out), _stream(v)))
    return out


def _stat_ws(x, n, stat_ws):
    """Scratch for the per-sample statistic.  A caller-owned `stat_ws` (>= n floats) keeps the per-sample maxima
    readable after the call (the multi-GPU calibration all-gathers them: dist.py)."""
    if stat_ws is None:
        return _workspace(x.device, _lib_().fq_act_workspace_bytes(n))
    _check(stat_ws, "stat_ws")
    if stat_ws.numel() < n:
        raise ValueError("stat_ws holds %d floats, need %d" % (stat_ws.numel(), n))
    return stat_ws


def fake_quant_online(x, width=8, flags=0, out=None, cur_out=None, want_codes=False, stat_ws=None):
    """convert_conv2d.py:56-66 + ste_func.py:41, threshold = this batch's statistic.
    Returns (y, current_max (1,) device tensor, codes or None)."""
    _check(x, "x")
    n, inner = _n_inner(x)
    y = torch.empty_like(x) if out is None else _check(out, "out")
    cur = torch.empty(1, dtype=torch.float32, device=x.device) if cur_out is None else _check(cur_out, "cur_out")
    codes = torch.empty(x.shape, dtype=torch.int32, device=x.device) if want_codes else None
    ws = _stat_ws(x, n, stat_ws)
    check_call(_lib_().fq_fake_quant_online(_ptr(x), _ptr(y), n, inner, int(width), int(flags), _ptr(cur),
                                            _ptr(codes), _ptr(ws), _stream(x)))
    return y, cur, codes


def fake_quant_online_prestat(x, stat, width=8, flags=0, out=None, cur_out=None, want_codes=False):
    """Online fake-quant when the per-sample statistic of x was already produced (by `bn_act_stat`): apply pass only."""
    _check(x, "x")
    _check(stat, "stat")
    n, inner = _n_inner(x)
    if stat.numel() < n:
        raise ValueError("stat holds %d values for %d samples" % (stat.numel(), n))
    y = torch.empty_like(x) if out is None else _check(out, "out")
    cur = torch.empty(1, dtype=torch.float32, device=x.device) if cur_out is None else _check(cur_out, "cur_out")
    codes = torch.empty(x.shape, dtype=torch.int32, device=x.device) if want_codes else None
    check_call(_lib_().fq_fake_quant_online_prestat(_ptr(x), _ptr(y), n, inner, _ptr(stat), int(width), int(flags),
                                                    _ptr(cur), _ptr(codes), _stream(x)))
    return y, cur, codes


_ACTS = {None: 0, "none": 0, "relu": 1, "relu6": 2}


_PREZEROED = 0x100


class StatArena(object):
    """Per-forward pool of pre-zeroed per-sample statistic rows: ONE zeroing launch per forward instead of one memset
    per producer kernel (~27 per mobilenet forward).  quantize/fuse.py owns the instance; producers call `take(n)`.
    The forward under way is tracked per thread (two nets evaluated from two threads keep their own arenas)."""
    _tls = threading.local()

    def __init__(self, slots, device):
        self.slots, self.device = slots, device
        self.width = 0
        self.buf = None
        self.next = 0

    def begin(self, n):
        if self.buf is None or n > self.width:
            self.width = int(n)
            self.buf = torch.zeros(self.slots, self.width, dtype=torch.float32, device=self.device)
        else:
            self.buf.zero_()
        self.next = 0
        StatArena._tls.current = self

    def end(self):
        if getattr(StatArena._tls, "current", None) is self:
            StatArena._tls.current = None

    @staticmethod
    def take(n, device):
        a = getattr(StatArena._tls, "current", None)
        if a is None or a.next >= a.slots or n > a.width or a.device != device:
            return None
        row = a.buf[a.next, :n]
        a.next += 1
        return row


def in_flight():
    """True inside `batches_in_flight()` on this thread."""
    return getattr(StatArena._tls, "in_flight", 0) > 0


class batches_in_flight(object):
    """`with ops.batches_in_flight():` - the caller runs EVALUATION forwards of independent batches on several HIP streams
    of one net at the same time.  Inside, a forward on a non-default stream keeps its per-forward device state per stream
    and leaves the blocks' `current_*_max` alone (quantize/convert/_blocks.py: scalar_slot); outside - whatever the
    current stream - every forward is an ordinary one that calibration (`net.update_ema()`) may follow."""

    def __enter__(self):
        StatArena._tls.in_flight = getattr(StatArena._tls, "in_flight", 0) + 1
        return self

    def __exit__(self, *exc):
        StatArena._tls.in_flight = getattr(StatArena._tls, "in_flight", 1) - 1
        return False


def _stat_target(n, device, want_stat):
    """(tensor or None, act-flag): a pre-zeroed arena row when a forward is under way, else a fresh tensor."""
    if not want_stat:
        return None, 0
    row = StatArena.take(n, device)
    if row is not None:
        return row, _PREZEROED
    return torch.empty(n, dtype=torch.float32, device=device), 0


HIST_FUSED_MAX_BINS = 4096        # fq_*_stat_hist keep four private copies of the histogram in 64 KiB of LDS


def _hist_args(hist, want_stat):
    """(max, bins, counts, negatives) of a histogram sink (`fm_max` float32[1], `hist` int64[bins], `neg` int32[1]: what
    distribution_calibrate keeps per collected block) for the producers that bin what they store."""
    if not want_stat:
        raise ValueError("a producer bins its output only together with the per-sample statistic")
    _check(hist.fm_max, "hist.fm_max")
    _check(hist.hist, "hist.hist", torch.int64)
    _check(hist.neg, "hist.neg", torch.int32)
    if not 0 < hist.hist.numel() <= HIST_FUSED_MAX_BINS:
        raise ValueError("%d bins: the fused form takes 1..%d" % (hist.hist.numel(), HIST_FUSED_MAX_BINS))
    return _ptr(hist.fm_max), int(hist.hist.numel()), _ptr(hist.hist), _ptr(hist.neg)


def bn_act_stat(x, scale, shift, act="relu", out=None, want_stat=True, hist=None, residual=None):
    """Fused inference BatchNorm (per-channel scale/shift) + activation + per-sample max|y| in one pass.
    x: (N, C, ...) ; returns (y, stat (N,) or None).  `hist` (KL collection, distribution_calibrate.py): a sink whose counts
    also receive y's histogram in the same pass (fq_bn_act_stat_hist) - what `histogram_accumulate(y, ...)` would add.
    `residual` (x's shape; fq_bn_add_act_stat): added after BatchNorm, before the activation - the values of this call without
    activation followed by `add_act_stat`, in one pass; the statistic is part of that form."""
    _check(x, "x")
    _check(scale, "scale")
    _check(shift, "shift")
    if x.dim() < 2 or x.numel() == 0:
        raise ValueError("bn_act_stat wants (N, C, ...) input")
    n, c = x.shape[0], x.shape[1]
    hw = x.numel() // (n * c)
    if scale.numel() != c or shift.numel() != c:
        raise ValueError("scale/shift must have %d elements" % c)
    if act not in _ACTS:
        raise ValueError("unknown activation %r" % (act,))
    y = torch.empty_like(x) if out is None else _check(out, "out")
    if residual is not None:
        _check(residual, "residual")
        if tuple(residual.shape) != tuple(x.shape) or not want_stat:
            raise ValueError("the residual must have x's shape %s (got %s) and the statistic is part of this form"
                             % (tuple(x.shape), tuple(residual.shape)))
        stat, zflag = _stat_target(n, x.device, True)
        if hist is not None:
            check_call(_lib_().fq_bn_add_act_stat_hist(_ptr(x), _ptr(residual), _ptr(y), n, c, hw, _ptr(scale), _ptr(shift),
                                                       _ACTS[act] | zflag, _ptr(stat), *_hist_args(hist, True), _stream(x)))
        else:
            check_call(_lib_().fq_bn_add_act_stat(_ptr(x), _ptr(residual), _ptr(y), n, c, hw, _ptr(scale), _ptr(shift),
                                                  _ACTS[act] | zflag, _ptr(stat), _stream(x)))
        return y, stat
    stat, zflag = _stat_target(n, x.device, want_stat)
    if hist is not None:
        check_call(_lib_().fq_bn_act_stat_hist(_ptr(x), _ptr(y), n, c, hw, _ptr(scale), _ptr(shift), _ACTS[act] | zflag,
                                               _ptr(stat), *_hist_args(hist, want_stat), _stream(x)))
        return y, stat
    check_call(_lib_().fq_bn_act_stat(_ptr(x), _ptr(y), n, c, hw, _ptr(scale), _ptr(shift), _ACTS[act] | zflag,
                                      _ptr(stat), _stream(x)))
    return y, stat


def bn_act_maxpool_stat(x, scale, shift, act="relu", want_stat=True):
    """BatchNorm (per-channel scale/shift) + activation + MaxPool2D(3, stride 2, padding 1) + per-sample max|y| in one
    pass.  x: (N, C, H, W) with W % 4 == 0; returns (y (N, C, (H-1)//2+1, W//2), stat (N,) or None)."""
    _check(x, "x")
    _check(scale, "scale")
    _check(shift, "shift")
    if x.dim() != 4 or x.shape[3] % 4:
        raise ValueError("bn_act_maxpool_stat wants (N, C, H, W) with W a multiple of 4, got %s" % (tuple(x.shape),))
    n, c, h, w = x.shape
    if scale.numel() != c or shift.numel() != c:
        raise ValueError("scale/shift must have %d elements" % c)
    y = torch.empty((n, c, (h - 1) // 2 + 1, w // 2), dtype=torch.float32, device=x.device)
    stat, zflag = _stat_target(n, x.device, want_stat)
    check_call(_lib_().fq_bn_act_maxpool_stat(_ptr(x), _ptr(y), n, c, h, w, _ptr(scale), _ptr(shift), _ACTS[act] | zflag,
                                              _ptr(stat), _stream(x)))
    return y, stat


def add_act_stat(a, b, act="relu", out=None, want_stat=True, hist=None):
    """`(a + b).relu()` — the residual tail of the model zoo's ResNet units — with the per-sample max|y| of the result for
    the quantised consumers (fq_add_act_stat).  Returns (y, stat or None).  `hist`: as in `bn_act_stat`."""
    _check(a, "a")
    _check(b, "b")
    if a.shape != b.shape:
        raise ValueError("shape mismatch: %s vs %s" % (tuple(a.shape), tuple(b.shape)))
    n, inner = _n_inner(a)
    y = torch.empty_like(a) if out is None else _check(out, "out")
    stat, zflag = _stat_target(n, a.device, want_stat)
    if hist is not None:
        check_call(_lib_().fq_add_act_stat_hist(_ptr(a), _ptr(b), _ptr(y), n, inner, _ACTS[act] | zflag, _ptr(stat),
                                                *_hist_args(hist, want_stat), _stream(a)))
        return y, stat
    check_call(_lib_().fq_add_act_stat(_ptr(a), _ptr(b), _ptr(y), n, inner, _ACTS[act] | zflag, _ptr(stat), _stream(a)))
    return y, stat


def global_avg_pool_stat(x, want_stat=True):
    """Global average pooling of (N, C, H, W) -> (N, C, 1, 1) plus the per-sample max|y| for the consumer's input
    quantiser (fq_global_avg_pool_stat).  Returns (y, stat (N,) or None)."""
    _check(x, "x")
    if x.dim() != 4:
        raise ValueError("global_avg_pool_stat wants (N, C, H, W)")
    n, c, h, w = x.shape
    y = torch.empty((n, c, 1, 1), dtype=torch.float32, device=x.device)
    stat, zflag = _stat_target(n, x.device, want_stat)
    check_call(_lib_().fq_global_avg_pool_stat(_ptr(x), _ptr(y), n, c, h * w, int(zflag), _ptr(stat), _stream(x)))
    return y, stat


def gemm_i8_codes(xcodes, wcodes, n, l, zoff):
    """Exact int32 GEMM of int8 code matrices on the matrix cores (fq_gemm_i8_codes).
    xcodes: (n*l, K) int8 im2col rows (already re-centred by `zoff` if unsigned); wcodes: (Cout, K) int8.
    Returns (n, Cout, l) int32 = sum_k x*w + zoff * rowsum(w)."""
    require_hip(xcodes.device, "xcodes")
    require_hip(wcodes.device, "wcodes")
    if xcodes.dtype != torch.int8 or wcodes.dtype != torch.int8 or xcodes.dim() != 2 or wcodes.dim() != 2 \
            or xcodes.shape[1] != wcodes.shape[1] or xcodes.shape[0] != n * l:
        raise ValueError("gemm_i8_codes wants xcodes (n*l, K) and wcodes (Cout, K), both int8")
    k, cout = xcodes.shape[1], wcodes.shape[0]
    kp, rp, cp = (k + 31) // 32 * 32, (cout + 31) // 32 * 32, (n * l + 31) // 32 * 32
    xc = torch.zeros((cp, kp), dtype=torch.int8, device=xcodes.device)
    xc[:n * l, :k] = xcodes
    wc = torch.zeros((rp, kp), dtype=torch.int8, device=xcodes.device)
    wc[:cout, :k] = wcodes
    wsum = wcodes.to(torch.int32).sum(dim=1).to(torch.int32).contiguous()
    out = torch.empty((n, cout, l), dtype=torch.int32, device=xcodes.device)
    check_call(_lib_().fq_gemm_i8_codes(xc.data_ptr(), wc.data_ptr(), wsum.data_ptr(), out.data_ptr(), n, l, kp, cout,
                                        int(zoff), _stream(xcodes)))
    return out


_EVAL_WS = {}


def dense_i8_eval(x, wcodes, wscale, wsum, labels, counters, bias=None, in_stat=None, in_thr=None, width=8, flags=0,
                  cur_out=None):
    """The quantised classifier on the integer codes AND the evaluation counters of its logits in one launch
    (fq_dense_i8_eval): `pwconv_i8` on planes of one pixel followed by `eval_counters`, same values.  x: (N, Cin[, 1, 1])
    fp32; labels (N,) int64; counters as `eval_counters`.  Returns the logits (N, units)."""
    _check(x, "x")
    _check(wcodes, "wcodes", torch.int8)
    _check(wscale, "wscale")
    _check(wsum, "wsum", torch.int32)
    _check(counters, "counters")
    require_hip(labels.device, "labels")
    for name, t in (("bias", bias), ("in_stat", in_stat), ("in_thr", in_thr), ("cur_out", cur_out)):
        if t is not None:
            _check(t, name)
    n, cin = x.shape[0], x.shape[1]
    cout, cin_pad = wscale.numel(), wcodes.shape[1]
    if x.numel() != n * cin:
        raise ValueError("dense_i8_eval wants (N, Cin) or (N, Cin, 1, 1) activations, got %s" % (tuple(x.shape),))
    if (cin + 63) // 64 * 64 != cin_pad:
        raise ValueError("x has %d channels but the weight codes were made for a row length that pads to %d"
                         % (cin, cin_pad))
    if labels.dim() != 1 or labels.shape[0] != n or labels.dtype != torch.int64:
        raise ValueError("dense_i8_eval wants labels (N,) int64")
    if counters.numel() != 2 + 2 * cout:
        raise ValueError("counters must hold 2 + 2 * %d floats" % cout)
    if in_stat is not None and cur_out is None:
        cur_out = torch.empty(1, dtype=torch.float32, device=x.device)
    nbytes = _lib_().fq_dense_i8_eval_workspace_bytes(n, cout)
    key = (str(x.device), int(nbytes), int(torch.cuda.current_stream(x.device).cuda_stream))
    ews = _EVAL_WS.get(key)
    if ews is None:
        ews = _EVAL_WS[key] = torch.zeros(nbytes // 8 + 1, dtype=torch.int64, device=x.device)   # zeroed once, kept zeroed
    y = torch.empty((n, cout), dtype=torch.float32, device=x.device)
    ws = torch.empty(_lib_().fq_pwconv_workspace_bytes(n, cin_pad, 1), dtype=torch.uint8, device=x.device)
    lb = labels.contiguous()
    check_call(_lib_().fq_dense_i8_eval(_ptr(x), _ptr(wcodes), _ptr(wscale), _ptr(wsum), _ptr(bias), _ptr(y), n, cin,
                                        cin_pad, cout, _ptr(in_stat), _ptr(in_thr), int(width), int(flags), _ptr(cur_out),
                                        lb.data_ptr(), _ptr(counters), ews.data_ptr(), _ptr(ws), _stream(x)))
    return y


def eval_counters(logits, labels, counters):
    """counters ([n_correct, total, correct[c], label[c]], float32, 2 + 2*classes) += the batch's evaluation counts
    (reference CLI :122-148).  logits (N, classes) fp32, labels (N,) int64.  Returns `counters`."""
    _check(logits, "logits")
    _check(counters, "counters")
    require_hip(labels.device, "labels")
    if logits.dim() != 2 or labels.dim() != 1 or labels.shape[0] != logits.shape[0] or labels.dtype != torch.int64:
        raise ValueError("eval_counters wants logits (N, classes) float32 and labels (N,) int64")
    n, classes = logits.shape
    if counters.numel() != 2 + 2 * classes or not counters.is_contiguous():
        raise ValueError("counters must hold 2 + 2*classes contiguous floats")
    lg = logits if logits.is_contiguous() else logits.contiguous()
    lb = labels if labels.is_contiguous() else labels.contiguous()
    check_call(_lib_().fq_eval_counters(_ptr(lg), lb.data_ptr(), n, classes, _ptr(counters), _stream(logits)))
    return counters


def stem_conv_s2(x, w, bias=None, bn_scale=None, bn_shift=None, act=None, want_stat=True, w_tap_major=None, out_codes=None,
                 pool=False):
    """The un-quantised first convolution - 3x3 / stride 2 / pad 1 / 3 -> 32 channels (MobileNets) or 7x7 / stride 2 / pad 3 /
    3 -> 64 (ResNets) - with fused BatchNorm / activation / per-sample statistic.  w: (Cout, 3, K, K) as the Conv2D parameter
    holds it; pass `w_tap_major` (= w.permute(1,2,3,0) contiguous) to skip the permutation.  Returns (y, stat (N,) or None).
    `out_codes=dict(thr=..., width=8, flags=0)` (3x3 form; fq_stem_conv3x3s2_c16): y is a `Codes16` of the consumer's codes.
    `pool=True` (7x7 form; fq_stem_conv7x7s2_pool): MaxPool2D(3, 2, 1) of the result in the same launch - y is the pooled tensor,
    stat its statistic (`stem_pool_supported(h, w)` says whether the shape is built)."""
    _check(x, "x")
    _check(w, "w")
    if x.dim() != 4 or w.dim() != 4 or w.shape[2] != w.shape[3] or w.shape[2] not in (3, 7) or w.shape[1] != x.shape[1]:
        raise ValueError("stem_conv_s2 wants x (N,C,H,W) and w (Cout,C,K,K), K = 3 or 7; got %s and %s"
                         % (tuple(x.shape), tuple(w.shape)))
    for name, t in (("bias", bias), ("bn_scale", bn_scale), ("bn_shift", bn_shift), ("w_tap_major", w_tap_major)):
        if t is not None:
            _check(t, name)
    act = act or "none"
    if act not in _ACTS:
        raise ValueError("unknown activation %r" % (act,))
    n, cin, h, wd = x.shape
    cout, ks = w.shape[0], w.shape[2]
    pad = ks // 2
    wt = w_tap_major if w_tap_major is not None else w.permute(1, 2, 3, 0).contiguous()
    y = torch.empty((n, cout, (h + 2 * pad - ks) // 2 + 1, (wd + 2 * pad - ks) // 2 + 1), dtype=torch.float32,
                    device=x.device)
    stat, zflag = _stat_target(n, x.device, want_stat)
    if out_codes is not None:
        if ks != 3:
            raise ValueError("the first convolution hands codes over in its 3x3 form only")
        othr = _check(out_codes["thr"], "out_codes['thr']")
        yc = Codes16.empty(tuple(y.shape), x.device, othr, out_codes.get("width", 8), out_codes.get("flags", 0))
        check_call(_lib_().fq_stem_conv3x3s2_c16(_ptr(x), _ptr(wt), _ptr(bias), _ptr(yc.t), n, cin, cout, h, wd, _ptr(bn_scale),
                                                 _ptr(bn_shift), _ACTS[act] | zflag, _ptr(stat), _ptr(othr), int(yc.width),
                                                 int(yc.flags), _stream(x)))
        return yc, stat
    if pool:
        if ks != 7 or not stem_pool_supported(h, wd):
            raise ValueError("the pooled form is built for the 7x7 first convolution on planes whose output rows cut into four "
                             "tiles of at most 32 columns; got K = %d on %d x %d" % (ks, h, wd))
        ho, wo = y.shape[2], y.shape[3]
        yp = torch.empty((n, cout, (ho - 1) // 2 + 1, (wo - 1) // 2 + 1), dtype=torch.float32, device=x.device)
        check_call(_lib_().fq_stem_conv7x7s2_pool(_ptr(x), _ptr(wt), _ptr(bias), _ptr(yp), n, cin, cout, h, wd, _ptr(bn_scale),
                                                  _ptr(bn_shift), _ACTS[act] | zflag, _ptr(stat), _stream(x)))
        return yp, stat
    entry = _lib_().fq_stem_conv3x3s2 if ks == 3 else _lib_().fq_stem_conv7x7s2
    check_call(entry(_ptr(x), _ptr(wt), _ptr(bias), _ptr(y), n, cin, cout, h, wd, _ptr(bn_scale), _ptr(bn_shift),
                     _ACTS[act] | zflag, _ptr(stat), _stream(x)))
    return y, stat


stem_conv3x3s2 = stem_conv_s2          # the 3x3 -> 32 case had its own name first


def stem_pool_supported(h, w):
    return bool(_lib_().fq_stem_conv7x7s2_pool_supported(int(h), int(w)))


def stem_conv_supported(cin, cout, kernel, stride, pad):
    k, s, p = tuple(kernel), tuple(stride), tuple(pad)
    return cin == 3 and s == (2, 2) and ((cout == 32 and k == (3, 3) and p == (1, 1)) or
                                         (cout == 64 and k == (7, 7) and p == (3, 3)))


def dwconv3x3(x, w, bias=None, stride=1, in_stat=None, in_thr=None, width=8, flags=0, cur_out=None, bn_scale=None,
              bn_shift=None, act=None, want_stat=True):
    """Depthwise 3x3 (pad 1) with optional quantise-on-load of x (in_stat: online, in_thr: offline) and fused
    BatchNorm/activation/per-sample-statistic epilogue.  Returns (y, stat (N,) or None)."""
    _check(x, "x")
    _check(w, "w")
    if x.dim() != 4 or w.dim() != 4 or tuple(w.shape[1:]) != (1, 3, 3) or w.shape[0] != x.shape[1]:
        raise ValueError("dwconv3x3 wants x (N,C,H,W) and w (C,1,3,3); got %s and %s" % (tuple(x.shape), tuple(w.shape)))
    for name, t in (("bias", bias), ("in_stat", in_stat), ("in_thr", in_thr), ("bn_scale", bn_scale),
                    ("bn_shift", bn_shift), ("cur_out", cur_out)):
        if t is not None:
            _check(t, name)
    n, c, h, wd = x.shape
    ho, wo = (h - 1) // stride + 1, (wd - 1) // stride + 1
    y = torch.empty((n, c, ho, wo), dtype=torch.float32, device=x.device)
    stat, zflag = _stat_target(n, x.device, want_stat)
    check_call(_lib_().fq_dwconv3x3(_ptr(x), _ptr(w), _ptr(bias), _ptr(y), n, c, h, wd, int(stride), _ptr(in_stat),
                                    _ptr(in_thr), int(width), int(flags), _ptr(cur_out), _ptr(bn_scale),
                                    _ptr(bn_shift), _ACTS[act] | zflag, _ptr(stat), _stream(x)))
    return y, stat


def dwconv3x3_c16(x, w, bias=None, stride=1, in_stat=None, in_thr=None, width=8, flags=0, cur_out=None, bn_scale=None,
                  bn_shift=None, act=None, want_stat=True, out_codes=None):
    """Depthwise 3x3 between two `Codes16` tensors (fq_dwconv3x3_c16): x was quantised with in_thr / width / flags, the
    result carries the consumer's codes (`out_codes=dict(thr=..., width=8, flags=0)`).  Returns (Codes16, stat or None)."""
    if not isinstance(x, Codes16) or in_thr is None or not x.matches(in_thr, width, flags):
        raise ValueError("dwconv3x3_c16 wants a Codes16 input quantised with this call's in_thr / width / signedness")
    if out_codes is None:
        raise ValueError("dwconv3x3_c16 writes codes: give out_codes")
    _check(x.t, "x", torch.int8)
    _check(w, "w")
    n, c, h, wd = x.shape
    if w.dim() != 4 or tuple(w.shape[1:]) != (1, 3, 3) or w.shape[0] != c:
        raise ValueError("dwconv3x3_c16 wants w (C,1,3,3); got %s for C = %d" % (tuple(w.shape), c))
    for name, t in (("bias", bias), ("in_stat", in_stat), ("in_thr", in_thr), ("bn_scale", bn_scale),
                    ("bn_shift", bn_shift), ("cur_out", cur_out)):
        if t is not None:
            _check(t, name)
    dev = x.t.device
    ho, wo = (h - 1) // stride + 1, (wd - 1) // stride + 1
    othr = _check(out_codes["thr"], "out_codes['thr']")
    y = Codes16.empty((n, c, ho, wo), dev, othr, out_codes.get("width", 8), out_codes.get("flags", 0))
    if in_stat is not None and cur_out is None:
        cur_out = torch.empty(1, dtype=torch.float32, device=dev)
    stat, zflag = _stat_target(n, dev, want_stat)
    check_call(_lib_().fq_dwconv3x3_c16(_ptr(x.t), _ptr(w), _ptr(bias), _ptr(y.t), n, c, h, wd, int(stride), _ptr(in_stat),
                                        _ptr(in_thr), int(width), int(flags), _ptr(cur_out), _ptr(bn_scale), _ptr(bn_shift),
                                        _ACTS[act] | zflag, _ptr(stat), _ptr(othr), y.width, y.flags, _stream(w)))
    return y, stat


def pwdw_supported(xshape, cout, stride):
    """True when the pair (1x1 to `cout` channels, depthwise 3x3 of stride `stride`) on an (N, Cin, H, W) input is a shape both
    fq_pwconv_i8_stat and fq_pwdw_fused take."""
    n, cin, h, w = (int(v) for v in xshape)
    lib = _lib_()
    return bool(lib.fq_pwconv_i8_stat_supported(n, cin, int(cout), h * w)) and \
        bool(lib.fq_pwdw_fused_supported(n, cin, int(cout), h, w, int(stride)))


def pwconv_i8_stat(x, wcodes, wscale, wsum, bias=None, in_stat=None, in_thr=None, width=8, flags=0, cur_out=None,
                   bn_scale=None, bn_shift=None, act=None):
    """The statistic-only pass of a fused 1x1 convolution (fq_pwconv_i8_stat): what `pwconv_i8` computes without storing it.
    Returns the per-sample maxima max|y[n]| (N,)."""
    _check(x, "x")
    _check(wcodes, "wcodes", torch.int8)
    _check(wscale, "wscale")
    _check(wsum, "wsum", torch.int32)
    for name, t in (("bias", bias), ("in_stat", in_stat), ("in_thr", in_thr), ("bn_scale", bn_scale),
                    ("bn_shift", bn_shift), ("cur_out", cur_out)):
        if t is not None:
            _check(t, name)
    n, cin = x.shape[0], x.shape[1]
    cout = wscale.numel()
    cin_pad = wcodes.shape[1]
    if (cin + 63) // 64 * 64 != cin_pad:
        raise ValueError("x has %d channels but the weight codes were made for a row length that pads to %d" % (cin, cin_pad))
    if in_stat is not None and cur_out is None:
        cur_out = torch.empty(1, dtype=torch.float32, device=x.device)
    stat, zflag = _stat_target(n, x.device, True)
    hw = x.numel() // (n * cin)
    check_call(_lib_().fq_pwconv_i8_stat(_ptr(x), _ptr(wcodes), _ptr(wscale), _ptr(wsum), _ptr(bias), n, cin, cin_pad,
                                         wcodes.shape[0], cout, hw, _ptr(in_stat), _ptr(in_thr), int(width), int(flags), _ptr(cur_out),
                                         _ptr(bn_scale), _ptr(bn_shift), _ACTS[act] | zflag, _ptr(stat), _stream(x)))
    return stat


def pwdw_fused(x, wcodes, wscale, wsum, dw_w, pw_bias=None, in_stat=None, in_thr=None, width=8, flags=0, pw_bn_scale=None,
               pw_bn_shift=None, pw_act=None, mid_stat=None, mid_thr=None, mid_width=8, mid_flags=0, mid_cur_out=None,
               dw_bias=None, stride=1, dw_bn_scale=None, dw_bn_shift=None, dw_act=None, want_stat=True):
    """A fused 1x1 convolution and the depthwise 3x3 behind it in one launch (fq_pwdw_fused): the values of `pwconv_i8`
    followed by `dwconv3x3` with quantise-on-load, without the tensor between them.  `mid_stat`: the per-sample maxima of the
    1x1 output (`pwconv_i8_stat`), or `mid_thr`: a stored threshold.  Returns (z, stat or None)."""
    _check(x, "x")
    _check(wcodes, "wcodes", torch.int8)
    _check(wscale, "wscale")
    _check(wsum, "wsum", torch.int32)
    _check(dw_w, "dw_w")
    for name, t in (("pw_bias", pw_bias), ("in_stat", in_stat), ("in_thr", in_thr), ("pw_bn_scale", pw_bn_scale),
                    ("pw_bn_shift", pw_bn_shift), ("mid_stat", mid_stat), ("mid_thr", mid_thr), ("mid_cur_out", mid_cur_out),
                    ("dw_bias", dw_bias), ("dw_bn_scale", dw_bn_scale), ("dw_bn_shift", dw_bn_shift)):
        if t is not None:
            _check(t, name)
    if x.dim() != 4:
        raise ValueError("pwdw_fused wants x (N, Cin, H, W); got %s" % (tuple(x.shape),))
    n, cin, h, w = x.shape
    cout = wscale.numel()
    if dw_w.dim() != 4 or tuple(dw_w.shape) != (cout, 1, 3, 3):
        raise ValueError("pwdw_fused wants dw_w (%d,1,3,3); got %s" % (cout, tuple(dw_w.shape)))
    cout_pad, cin_pad = wcodes.shape
    if (cin + 63) // 64 * 64 != cin_pad:
        raise ValueError("x has %d channels but the weight codes were made for a row length that pads to %d" % (cin, cin_pad))
    ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
    y = torch.empty((n, cout, ho, wo), dtype=torch.float32, device=x.device)
    stat, zflag = _stat_target(n, x.device, want_stat)
    check_call(_lib_().fq_pwdw_fused(_ptr(x), _ptr(wcodes), _ptr(wscale), _ptr(wsum), _ptr(pw_bias), n, cin, cin_pad, cout_pad,
                                     cout, h, w, _ptr(in_stat), _ptr(in_thr), int(width), int(flags), _ptr(pw_bn_scale),
                                     _ptr(pw_bn_shift), _ACTS[pw_act], _ptr(mid_stat), _ptr(mid_thr), int(mid_width),
                                     int(mid_flags), _ptr(mid_cur_out), _ptr(dw_w), _ptr(dw_bias), int(stride),
                                     _ptr(dw_bn_scale), _ptr(dw_bn_shift), _ACTS[dw_act] | zflag, _ptr(y), _ptr(stat),
                                     _stream(x)))
    return y, stat


def weight_codes(w, rows_per_scale, width=8):
    """Integer codes of the weight fake-quant (convert_conv2d.py:70-95): code = roundf(w / (s + 1e-10)) with one scale
    per `rows_per_scale` leading rows.  Returns (codes int8 [rows_pad, row_pad] zero padded to multiples of 64,
    scales (rows,), rowsum int32 (rows,))."""
    _check(w, "w")
    rows = w.shape[0]
    row_len = w.numel() // rows
    row_pad = (row_len + 63) // 64 * 64
    rows_pad = (rows + 63) // 64 * 64
    # row-major codes followed by the fragment-major copy the tiled pointwise kernel streams (include/fakequant.h)
    buf = torch.empty(2 * rows_pad * row_pad, dtype=torch.int8, device=w.device)
    codes = buf[:rows_pad * row_pad].view(rows_pad, row_pad)
    scales = torch.empty(rows, dtype=torch.float32, device=w.device)
    rowsum = torch.empty(rows, dtype=torch.int32, device=w.device)
    ws = _workspace(w.device, _lib_().fq_weight_workspace_bytes(rows))
    check_call(_lib_().fq_weight_codes(_ptr(w), rows, row_len, int(rows_per_scale), int(width), row_pad, rows_pad,
                                       _ptr(codes), _ptr(scales), _ptr(rowsum), _ptr(ws), _stream(w)))
    return codes, scales, rowsum


def weight_codes_reproduce(w, codes, scales):
    """True when code * scale gives back `w` bit for bit (w: the (rows, ...) fp32 weight the codes were derived from).
    Used after re-deriving codes from already fake-quantised (frozen) weights: fl(fl(127 s) / 127) need not equal s, and
    the convolution must multiply what the reference would — the frozen weights themselves.  Synchronises; rare path."""
    rows = w.shape[0]
    w2 = w.reshape(rows, -1)
    back = codes[:rows, :w2.shape[1]].to(torch.float32) * scales.reshape(rows, 1)
    return bool(torch.equal(back, w2))


class Codes16(object):
    """An (n, C, h, w) activation handed over as integer codes in the C16 layout of include/fakequant.h: int8
    [n][ceil(C / 16)][h * w][16], byte = (code + 128 - zoff) ^ 0x80.  `thr` / `width` / `flags` describe the quantiser that
    made the codes - the CONSUMER's stored threshold (offline input quantisation)."""
    __slots__ = ("t", "shape", "thr", "width", "flags")

    def __init__(self, t, shape, thr, width, flags):
        self.t, self.shape, self.thr, self.width, self.flags = t, tuple(int(d) for d in shape), thr, int(width), int(flags)

    @staticmethod
    def empty(shape, device, thr, width, flags):
        n, c, h, w = shape
        t = torch.empty((n, (c + 15) // 16, h * w, 16), dtype=torch.int8, device=device)
        return Codes16(t, shape, thr, width, flags)

    def matches(self, thr, width, flags):
        return self.thr.data_ptr() == thr.data_ptr() and self.width == int(width) and \
            (self.flags & 3) == (int(flags) & 3)


PW_FORMS = {None: 0, "auto": 0, "two_kernels": 1, "stream": 3, "split": 6, "sample": 7, "rows": 8, "pipe": 9}


SPLIT_KT = (2, 4, 6, 8, 10, 12, 16, 18, 30, 32, 64)     # padded Cin / 32 the split form of fq_pwconv_i8 is built for


def pwconv_strided_supported(cin):
    """Strided 1x1 convolutions run on the integer codes only through the split form (include/fakequant.h)."""
    return ((int(cin) + 63) // 64 * 64) // 32 in SPLIT_KT


def pwconv_shortcut_supported(cin, cin2, cout):
    """Shapes `pwconv_i8_shortcut` takes (fq_pwconv_i8_shortcut_supported): the stage heads of the v1 bottleneck ResNets."""
    return bool(_lib_().fq_pwconv_i8_shortcut_supported(int(cin), int(cin2), int(cout)))


def pwconv_i8_shortcut(x, wcodes, wscale, wsum, bias=None, in_stat=None, in_thr=None, width=8, flags=0, cur_out=None,
                       bn_scale=None, bn_shift=None, act=None, x2=None, wcodes2=None, wscale2=None, wsum2=None, in_stat2=None,
                       in_thr2=None, width2=8, flags2=0, cur_out2=None, bn_scale2=None, bn_shift2=None, side_codes=None):
    """The closing 1x1 convolution of a residual unit and the unit's shortcut convolution (`..2` arguments: 1x1 on x2, BatchNorm,
    no activation, no bias) in one launch (fq_pwconv_i8_shortcut): what `pwconv_i8(x, ..., residual=pwconv_i8(x2, ...)[0])` returns,
    bit for bit, without the shortcut tensor.  Returns (y, stat).

    Stored thresholds (fq_pwconv_i8_shortcut_c16): `x` a `Codes16` made with in_thr, `side_codes=dict(thr=, width=8, flags=0)` - y
    leaves as fp32 and, third value, as the `Codes16` of y under that threshold (as `pwconv_i8(..., side_codes=)`); `x2` fp32 or a
    `Codes16` made with in_thr2."""
    a16, b16 = isinstance(x, Codes16), isinstance(x2, Codes16)
    if a16 != (side_codes is not None) or (b16 and not a16):
        raise ValueError("pwconv_i8_shortcut: fp32 on every side, or a Codes16 x with side_codes (x2 then fp32 or Codes16)")
    if a16 and (in_thr is None or not x.matches(in_thr, width, flags)):
        raise ValueError("the C16 input was quantised with another threshold / width / signedness than this call names")
    if b16 and (in_thr2 is None or not x2.matches(in_thr2, width2, flags2)):
        raise ValueError("the C16 shortcut input was quantised with another threshold / width / signedness than this call names")
    xt, x2t = (x.t if a16 else x), (x2.t if b16 else x2)
    xs, x2s = (x.shape if a16 else tuple(x.shape)), (x2.shape if b16 else tuple(x2.shape))
    for name, t, dt in (("x", xt, torch.int8 if a16 else None), ("wcodes", wcodes, torch.int8), ("wscale", wscale, None),
                        ("wsum", wsum, torch.int32), ("x2", x2t, torch.int8 if b16 else None), ("wcodes2", wcodes2, torch.int8),
                        ("wscale2", wscale2, None), ("wsum2", wsum2, torch.int32), ("bn_scale2", bn_scale2, None),
                        ("bn_shift2", bn_shift2, None)):
        if dt is None:
            _check(t, name)
        else:
            _check(t, name, dt)
    for name, t in (("bias", bias), ("in_stat", in_stat), ("in_thr", in_thr), ("bn_scale", bn_scale), ("bn_shift", bn_shift),
                    ("cur_out", cur_out), ("in_stat2", in_stat2), ("in_thr2", in_thr2), ("cur_out2", cur_out2)):
        if t is not None:
            _check(t, name)
    if len(xs) != 4 or len(x2s) != 4 or tuple(xs[2:]) != tuple(x2s[2:]) or xs[0] != x2s[0]:
        raise ValueError("pwconv_i8_shortcut wants x (N, Cin, H, W) and x2 (N, Cin2, H, W); got %s and %s" % (tuple(xs), tuple(x2s)))
    n, cin, h, w = xs
    cin2 = x2s[1]
    dev = xt.device
    cout = wscale.numel()
    if wscale2.numel() != cout:
        raise ValueError("the two convolutions must have the same number of filters (%d and %d)" % (cout, wscale2.numel()))
    if wcodes.shape[1] != cin or wcodes2.shape[1] != cin2 or not pwconv_shortcut_supported(cin, cin2, cout):
        raise ValueError("pwconv_i8_shortcut: %d / %d -> %d channels is not a shape it takes" % (cin, cin2, cout))
    if in_stat is not None and cur_out is None:
        cur_out = torch.empty(1, dtype=torch.float32, device=dev)
    if in_stat2 is not None and cur_out2 is None:
        cur_out2 = torch.empty(1, dtype=torch.float32, device=dev)
    y = torch.empty((n, cout, h, w), dtype=torch.float32, device=dev)
    stat, zflag = _stat_target(n, dev, True)
    if not a16:
        check_call(_lib_().fq_pwconv_i8_shortcut(_ptr(xt), _ptr(wcodes), _ptr(wscale), _ptr(wsum), _ptr(bias), _ptr(y), n, cin, cin,
                                                 cout, h * w, _ptr(in_stat), _ptr(in_thr), int(width), int(flags), _ptr(cur_out),
                                                 _ptr(bn_scale), _ptr(bn_shift), _ACTS[act] | zflag, _ptr(stat), _ptr(x2t),
                                                 _ptr(wcodes2), _ptr(wscale2), _ptr(wsum2), cin2, cin2, _ptr(in_stat2), _ptr(in_thr2),
                                                 int(width2), int(flags2), _ptr(cur_out2), _ptr(bn_scale2), _ptr(bn_shift2),
                                                 _stream(xt)))
        return y, stat
    sthr = _check(side_codes["thr"], "side_codes['thr']")
    y16 = Codes16.empty((n, cout, h, w), dev, sthr, side_codes.get("width", 8), side_codes.get("flags", 0))
    check_call(_lib_().fq_pwconv_i8_shortcut_c16(_ptr(xt), _ptr(wcodes), _ptr(wscale), _ptr(wsum), _ptr(bias), _ptr(y), _ptr(y16.t), n,
                                                 cin, cin, cout, h * w, _ptr(in_stat), _ptr(in_thr), int(width), int(flags),
                                                 _ptr(cur_out), _ptr(bn_scale), _ptr(bn_shift), _ACTS[act] | zflag, _ptr(stat),
                                                 _ptr(x2t), 1 if b16 else 0, _ptr(wcodes2), _ptr(wscale2), _ptr(wsum2), cin2, cin2,
                                                 _ptr(in_stat2), _ptr(in_thr2), int(width2), int(flags2), _ptr(cur_out2),
                                                 _ptr(bn_scale2), _ptr(bn_shift2), _ptr(sthr), int(y16.width), int(y16.flags),
                                                 _stream(wcodes)))
    return y, stat, y16


def pwconv_gap_supported(xshape, cout, residual=False):
    """Shapes `pwconv_i8_gap` takes (fq_pwconv_i8_gap_supported): whole planes of 45 .. 64 pixels."""
    if len(xshape) != 4:
        return False
    return bool(_lib_().fq_pwconv_i8_gap_supported(int(xshape[0]), int(xshape[1]), int(cout), int(xshape[2]) * int(xshape[3]),
                                                    1 if residual else 0))


def pwconv_i8_gap(x, wcodes, wscale, wsum, bias=None, in_stat=None, in_thr=None, width=8, flags=0, cur_out=None,
                  bn_scale=None, bn_shift=None, act=None, residual=None):
    """`pwconv_i8` and the global average pooling behind it in one launch (fq_pwconv_i8_gap): returns (means (N, Cout, 1, 1), stat)
    - what `global_avg_pool_stat(pwconv_i8(...)[0])` returns, bit for bit, without the tensor between."""
    _check(x, "x")
    _check(wcodes, "wcodes", torch.int8)
    _check(wscale, "wscale")
    _check(wsum, "wsum", torch.int32)
    for name, t in (("bias", bias), ("in_stat", in_stat), ("in_thr", in_thr), ("bn_scale", bn_scale), ("bn_shift", bn_shift),
                    ("cur_out", cur_out), ("residual", residual)):
        if t is not None:
            _check(t, name)
    if x.dim() != 4:
        raise ValueError("pwconv_i8_gap wants x (N, Cin, H, W); got %s" % (tuple(x.shape),))
    n, cin, h, w = x.shape
    cout = wscale.numel()
    cin_pad = wcodes.shape[1]
    if cin != cin_pad:
        raise ValueError("pwconv_i8_gap: Cin = %d must be a multiple of 64" % cin)
    if residual is not None and tuple(residual.shape) != (n, cout, h, w):
        raise ValueError("the residual must have the convolution output's shape %s, got %s" % ((n, cout, h, w), tuple(residual.shape)))
    if not pwconv_gap_supported(tuple(x.shape), cout, residual is not None):
        raise ValueError("pwconv_i8_gap: shape %s -> %d channels is not taken (fq_pwconv_i8_gap_supported)" % (tuple(x.shape), cout))
    if in_stat is not None and cur_out is None:
        cur_out = torch.empty(1, dtype=torch.float32, device=x.device)
    y = torch.empty((n, cout, 1, 1), dtype=torch.float32, device=x.device)
    stat, zflag = _stat_target(n, x.device, True)
    ws = torch.empty(_lib_().fq_pwconv_workspace_bytes(n, cin_pad, h * w), dtype=torch.uint8, device=x.device)
    check_call(_lib_().fq_pwconv_i8_gap(_ptr(x), _ptr(wcodes), _ptr(wscale), _ptr(wsum), _ptr(bias), _ptr(y), n, cin, cin_pad, cout,
                                        h * w, _ptr(in_stat), _ptr(in_thr), int(width), int(flags), _ptr(cur_out), _ptr(bn_scale),
                                        _ptr(bn_shift), _ACTS[act] | zflag, _ptr(stat), _ptr(residual), _ptr(ws), _stream(x)))
    return y, stat


def pwconv_sub2_supported(cin, cout):
    """Shapes `pwconv_i8(..., subsample=True)` takes (fq_pwconv_i8_sub2_supported)."""
    return bool(_lib_().fq_pwconv_i8_sub2_supported(int(cin), int(cout)))


def pwconv_i8(x, wcodes, wscale, wsum, bias=None, in_stat=None, in_thr=None, width=8, flags=0, cur_out=None,
              bn_scale=None, bn_shift=None, act=None, want_stat=True, form=None, stride=1, residual=None, out_codes=None,
              side_codes=None, subsample=False):
    """1x1 convolution on the integer codes (int8 MFMA, exact int32 accumulation) with quantise-on-load and fused
    BatchNorm / activation / statistic.  x: (N, Cin, H, W) raw activations; stride 1 or 2 (no padding); `residual` (the
    output's shape) is added after BatchNorm and before the activation.  `form` names one of PW_FORMS ("stream", "sample",
    "split", "two_kernels") instead of the library's shape-based choice - for parity tests and tuning; a shape the named form
    does not take raises.  Returns (y, stat or None).

    Offline hand-over (fq_pwconv_i8_c16): `x` may be a `Codes16` (made with this call's in_thr / width / flags), and
    `out_codes=dict(thr=<consumer's threshold tensor>, width=8, flags=0)` makes `y` a `Codes16` of the consumer's codes.
    `side_codes=dict(thr=..., width=8, flags=0)` (fq_pwconv_i8_c16_dual; a `Codes16` input with a residual operand, stride 1):
    y stays fp32 and a third value is returned, the `Codes16` of y under that threshold - the trunk of a ResNet stored twice.

    `subsample=True` (fq_pwconv_i8_sub2; fp32 x (N, Cin, H, W), stride 1): the returned y is y[:, :, ::2, ::2] of the tensor
    the call stands for - (N, Cout, ceil(H/2), ceil(W/2)), all its stride-2 readers need - while `stat` and `residual` keep the
    whole planes."""
    in16 = isinstance(x, Codes16)
    if subsample:
        if (in16 and side_codes is None) or out_codes is not None or stride != 1 or len(x.shape) != 4:
            raise ValueError("subsample=True goes with (N, Cin, H, W) activations, stride 1 and fp32 output (codes in: with "
                             "side_codes, fq_pwconv_i8_c16_dual_sub2)")
        if not pwconv_sub2_supported(x.shape[1], wscale.numel()):
            raise ValueError("subsample=True: %d -> %d channels is not a shape fq_pwconv_i8_sub2 takes" % (x.shape[1], wscale.numel()))
    if in16:
        if in_thr is None or not x.matches(in_thr, width, flags):
            raise ValueError("the C16 input was quantised with another threshold / width / signedness than this call names")
        xs = x.shape
        _check(x.t, "x", torch.int8)
    else:
        _check(x, "x")
        xs = tuple(x.shape)
    _check(wcodes, "wcodes", torch.int8)
    _check(wscale, "wscale")
    _check(wsum, "wsum", torch.int32)
    for name, t in (("bias", bias), ("in_stat", in_stat), ("in_thr", in_thr), ("bn_scale", bn_scale),
                    ("bn_shift", bn_shift), ("cur_out", cur_out), ("residual", residual)):
        if t is not None:
            _check(t, name)
    n, cin = xs[0], xs[1]
    dev = x.t.device if in16 else x.device
    cout = wscale.numel()
    cin_pad = wcodes.shape[1]
    if (cin + 63) // 64 * 64 != cin_pad:
        raise ValueError("x has %d channels but the weight codes were made for a row length that pads to %d"
                         % (cin, cin_pad))
    if in_stat is not None and cur_out is None:
        cur_out = torch.empty(1, dtype=torch.float32, device=dev)
    stat, zflag = _stat_target(n, dev, want_stat)
    if side_codes is not None:
        if not in16 or residual is None or stride != 1 or out_codes is not None or len(xs) != 4:
            raise ValueError("side_codes goes with a Codes16 input, a residual operand, stride 1 and fp32 output")
        h, w = xs[2], xs[3]
        hs, ws_ = ((h + 1) // 2, (w + 1) // 2) if subsample else (h, w)
        y = torch.empty((n, cout, hs, ws_), dtype=torch.float32, device=dev)
        if tuple(residual.shape) != (n, cout, h, w):
            raise ValueError("the residual must have the whole output's shape %s, got %s" % ((n, cout, h, w), tuple(residual.shape)))
        sthr = _check(side_codes["thr"], "side_codes['thr']")
        y16 = Codes16.empty((n, cout, hs, ws_), dev, sthr, side_codes.get("width", 8), side_codes.get("flags", 0))
        ws = torch.empty(_lib_().fq_pwconv_workspace_bytes(n, cin_pad, h * w), dtype=torch.uint8, device=dev)
        check_call((_lib_().fq_pwconv_i8_c16_dual_sub2 if subsample else _lib_().fq_pwconv_i8_c16_dual)(_ptr(x.t), _ptr(wcodes), _ptr(wscale), _ptr(wsum), _ptr(bias), _ptr(y),
                                                 _ptr(y16.t), n, cin, cin_pad, cout, h, w, _ptr(in_stat), _ptr(in_thr),
                                                 int(width), int(flags), _ptr(cur_out), _ptr(bn_scale), _ptr(bn_shift),
                                                 _ACTS[act] | zflag, _ptr(stat), _ptr(residual), _ptr(sthr), int(y16.width),
                                                 int(y16.flags), _ptr(ws), _stream(wcodes)))
        return y, stat, y16
    if in16 or out_codes is not None:
        if len(xs) != 4:
            raise ValueError("a C16 hand-over needs (N, Cin, H, W) activations, got %s" % (xs,))
        h, w = xs[2], xs[3]
        ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
        if out_codes is not None:
            if residual is not None:
                raise ValueError("a residual operand goes with fp32 output")
            othr = _check(out_codes["thr"], "out_codes['thr']")
            y = Codes16.empty((n, cout, ho, wo), dev, othr, out_codes.get("width", 8), out_codes.get("flags", 0))
            yp, ow, of = _ptr(y.t), y.width, y.flags
        else:
            y = torch.empty((n, cout, ho, wo), dtype=torch.float32, device=dev)
            if residual is not None and tuple(residual.shape) != tuple(y.shape):
                raise ValueError("the residual must have the output's shape %s, got %s" % (tuple(y.shape), tuple(residual.shape)))
            othr, yp, ow, of = None, _ptr(y), 8, 0
        ws = torch.empty(_lib_().fq_pwconv_workspace_bytes(n, cin_pad, ho * wo), dtype=torch.uint8, device=dev)
        check_call(_lib_().fq_pwconv_i8_c16(_ptr(x.t if in16 else x), 1 if in16 else 0, _ptr(wcodes), _ptr(wscale), _ptr(wsum),
                                            _ptr(bias), yp, n, cin, cin_pad, cout, h, w, int(stride), _ptr(in_stat),
                                            _ptr(in_thr), int(width), int(flags), _ptr(cur_out), _ptr(bn_scale),
                                            _ptr(bn_shift), _ACTS[act] | zflag, _ptr(stat), _ptr(residual), _ptr(othr),
                                            int(ow), int(of), _ptr(ws), _stream(wcodes)))
        return y, stat
    if subsample:
        h, w = x.shape[2], x.shape[3]
        y = torch.empty((n, cout, (h + 1) // 2, (w + 1) // 2), dtype=torch.float32, device=x.device)
        if residual is not None and tuple(residual.shape) != (n, cout, h, w):
            raise ValueError("the residual must have the whole output's shape %s, got %s" % ((n, cout, h, w), tuple(residual.shape)))
        ws = torch.empty(_lib_().fq_pwconv_workspace_bytes(n, cin_pad, h * w), dtype=torch.uint8, device=x.device)
        check_call(_lib_().fq_pwconv_i8_sub2(_ptr(x), _ptr(wcodes), _ptr(wscale), _ptr(wsum), _ptr(bias), _ptr(y), n, cin, cin_pad,
                                             cout, h, w, _ptr(in_stat), _ptr(in_thr), int(width), int(flags), _ptr(cur_out),
                                             _ptr(bn_scale), _ptr(bn_shift), _ACTS[act] | zflag, _ptr(stat), _ptr(residual),
                                             _ptr(ws), _stream(x)))
        return y, stat
    if stride != 1 or residual is not None:
        if x.dim() != 4:
            raise ValueError("a strided 1x1 convolution needs (N, Cin, H, W) activations, got %s" % (tuple(x.shape),))
        h, w = x.shape[2], x.shape[3]
        ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
        y = torch.empty((n, cout, ho, wo), dtype=torch.float32, device=x.device)
        if residual is not None and tuple(residual.shape) != tuple(y.shape):
            raise ValueError("the residual must have the output's shape %s, got %s" % (tuple(y.shape), tuple(residual.shape)))
        ws = torch.empty(_lib_().fq_pwconv_workspace_bytes(n, cin_pad, ho * wo), dtype=torch.uint8, device=x.device)
        check_call(_lib_().fq_pwconv_i8_strided(_ptr(x), _ptr(wcodes), _ptr(wscale), _ptr(wsum), _ptr(bias), _ptr(y), n,
                                                cin, cin_pad, cout, h, w, int(stride), _ptr(in_stat), _ptr(in_thr),
                                                int(width), int(flags), _ptr(cur_out), _ptr(bn_scale), _ptr(bn_shift),
                                                _ACTS[act] | zflag | (PW_FORMS[form] << 12), _ptr(stat), _ptr(residual),
                                                _ptr(ws), _stream(x)))
        return y, stat
    hw = x.numel() // (n * cin)
    y = torch.empty((n, cout) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device)
    ws = torch.empty(_lib_().fq_pwconv_workspace_bytes(n, cin_pad, hw), dtype=torch.uint8, device=x.device)
    check_call(_lib_().fq_pwconv_i8(_ptr(x), _ptr(wcodes), _ptr(wscale), _ptr(wsum), _ptr(bias), _ptr(y), n, cin,
                                    cin_pad, cout, hw, _ptr(in_stat), _ptr(in_thr), int(width), int(flags),
                                    _ptr(cur_out), _ptr(bn_scale), _ptr(bn_shift),
                                    _ACTS[act] | zflag | (PW_FORMS[form] << 12), _ptr(stat), _ptr(ws), _stream(x)))
    return y, stat


def weight_codes_3x3(w, rows_per_scale, width=8):
    """Codes of a dense 3x3 weight (Cout, Cin, 3, 3) in the K order `conv3x3_i8` multiplies in: (tap, ci), i.e. the codes of
    the weight permuted to (Cout, 3, 3, Cin).  Scales and row sums do not depend on the order inside a row."""
    _check(w, "w")
    if w.dim() != 4 or tuple(w.shape[2:]) != (3, 3):
        raise ValueError("expected a (Cout, Cin, 3, 3) weight, got %s" % (tuple(w.shape),))
    return weight_codes(w.permute(0, 2, 3, 1).contiguous(), rows_per_scale, width)


def weight_slices_3x3(w):
    """Three int8 slices of a dense 3x3 filter that is not on one integer grid per channel (Winograd-domain quantised weights,
    include/fakequant.h at fq_weight_slices), in conv3x3_i8's K order.  Returns (codes [3, 2 * rows_pad * row_pad] int8,
    pscale (cout,), rowsum (3, cout) int32)."""
    _check(w, "w")
    if w.dim() != 4 or tuple(w.shape[2:]) != (3, 3):
        raise ValueError("expected a (Cout, Cin, 3, 3) weight, got %s" % (tuple(w.shape),))
    wp = w.permute(0, 2, 3, 1).contiguous()
    rows = wp.shape[0]
    row_len = wp.numel() // rows
    row_pad = (row_len + 63) // 64 * 64
    rows_pad = (rows + 63) // 64 * 64
    codes = torch.empty((3, 2 * rows_pad * row_pad), dtype=torch.int8, device=w.device)
    pscale = torch.empty(rows, dtype=torch.float32, device=w.device)
    rowsum = torch.empty((3, rows), dtype=torch.int32, device=w.device)
    ws = _workspace(w.device, max(4 * rows, _lib_().fq_weight_workspace_bytes(rows)))
    check_call(_lib_().fq_weight_slices(_ptr(wp), rows, row_len, row_pad, rows_pad, _ptr(codes), _ptr(pscale), _ptr(rowsum),
                                        _ptr(ws), _stream(wp)))
    return codes, pscale, rowsum


def conv3x3_i8(x, wcodes, wscale, wsum, bias=None, in_stat=None, in_thr=None, width=8, flags=0, cur_out=None,
               bn_scale=None, bn_shift=None, act=None, want_stat=True, out_codes=None):
    """Dense 3x3 convolution (stride 1, padding 1) on the integer codes (int8 MFMA, exact int32 accumulation) with
    quantise-on-load and fused BatchNorm / activation / statistic.  x: (N, Cin, H, W) raw activations; wcodes / wscale / wsum
    from `weight_codes_3x3` - or from `weight_slices_3x3` (three int8 slices of a filter that is not on one integer grid:
    Winograd-domain quantised weights; fq_conv3x3_i8_sliced).  Offline hand-over as in `pwconv_i8`: `x` may be a `Codes16`,
    `out_codes=dict(thr=..., width=8, flags=0)` makes `y` one (fq_conv3x3_i8_c16).  Returns (y, stat or None)."""
    in16 = isinstance(x, Codes16)
    if in16:
        if in_thr is None or not x.matches(in_thr, width, flags):
            raise ValueError("the C16 input was quantised with another threshold / width / signedness than this call names")
        xs = x.shape
        _check(x.t, "x", torch.int8)
    else:
        _check(x, "x")
        xs = tuple(x.shape)
    _check(wcodes, "wcodes", torch.int8)
    _check(wscale, "wscale")
    _check(wsum, "wsum", torch.int32)
    for name, t in (("bias", bias), ("in_stat", in_stat), ("in_thr", in_thr), ("bn_scale", bn_scale),
                    ("bn_shift", bn_shift), ("cur_out", cur_out)):
        if t is not None:
            _check(t, name)
    if len(xs) != 4:
        raise ValueError("expected (N, Cin, H, W) activations, got %s" % (xs,))
    n, cin, h, w = xs
    dev = x.t.device if in16 else x.device
    cout = wscale.numel()
    sliced = wcodes.dim() == 2 and wcodes.shape[0] == 3 and wsum.dim() == 2        # from `weight_slices_3x3`
    if sliced:
        rows_pad = (cout + 63) // 64 * 64
        if wcodes.shape[1] != 2 * rows_pad * 9 * cin or (9 * cin) % 64:
            raise ValueError("weight slices do not match Cin = %d, Cout = %d" % (cin, cout))
    elif wcodes.shape[1] != 9 * cin:
        raise ValueError("weight codes have rows of %d, expected 9 * Cin = %d" % (wcodes.shape[1], 9 * cin))
    stat, zflag = _stat_target(n, dev, want_stat)
    if in_stat is not None and cur_out is None:
        cur_out = torch.empty(1, dtype=torch.float32, device=dev)
    if in16 or out_codes is not None:
        if sliced:
            raise ValueError("the three-slice form takes fp32 tensors on both sides")
        if out_codes is not None:
            othr = _check(out_codes["thr"], "out_codes['thr']")
            y = Codes16.empty((n, cout, h, w), dev, othr, out_codes.get("width", 8), out_codes.get("flags", 0))
            yp, ow, of = _ptr(y.t), y.width, y.flags
        else:
            y = torch.empty((n, cout, h, w), dtype=torch.float32, device=dev)
            othr, yp, ow, of = None, _ptr(y), 8, 0
        check_call(_lib_().fq_conv3x3_i8_c16(_ptr(x.t if in16 else x), 1 if in16 else 0, _ptr(wcodes), _ptr(wscale),
                                             _ptr(wsum), _ptr(bias), yp, n, cin, cout, h, w, _ptr(in_stat), _ptr(in_thr),
                                             int(width), int(flags), _ptr(cur_out), _ptr(bn_scale), _ptr(bn_shift),
                                             _ACTS[act] | zflag, _ptr(stat), _ptr(othr), int(ow), int(of), _stream(wcodes)))
        return y, stat
    y = torch.empty((n, cout, h, w), dtype=torch.float32, device=dev)
    if sliced:
        check_call(_lib_().fq_conv3x3_i8_sliced(_ptr(x), _ptr(wcodes), _ptr(wscale), _ptr(wsum), _ptr(bias), _ptr(y), n, cin,
                                                cout, h, w, _ptr(in_stat), _ptr(in_thr), int(width), int(flags),
                                                _ptr(cur_out), _ptr(bn_scale), _ptr(bn_shift), _ACTS[act] | zflag,
                                                _ptr(stat), _stream(x)))
        return y, stat
    check_call(_lib_().fq_conv3x3_i8(_ptr(x), _ptr(wcodes), _ptr(wscale), _ptr(wsum), _ptr(bias), _ptr(y), n, cin, cout,
                                     h, w, _ptr(in_stat), _ptr(in_thr), int(width), int(flags), _ptr(cur_out),
                                     _ptr(bn_scale), _ptr(bn_shift), _ACTS[act] | zflag, _ptr(stat), _stream(x)))
    return y, stat


def fake_quant_offline(x, threshold, width=8, flags=0, out=None, cur_out=None, want_stat=True, want_codes=False,
                       stat_ws=None):
    """Same arithmetic with the stored threshold `input_max` (convert_conv2d.py:58).  `want_stat` also yields the
    batch statistic the reference computes in every mode (:56), fused into the same pass."""
    _check(x, "x")
    _check(threshold, "threshold")
    n, inner = _n_inner(x)
    y = torch.empty_like(x) if out is None else _check(out, "out")
    cur = None
    if want_stat:
        cur = torch.empty(1, dtype=torch.float32, device=x.device) if cur_out is None else _check(cur_out, "cur_out")
    codes = torch.empty(x.shape, dtype=torch.int32, device=x.device) if want_codes else None
    ws = _stat_ws(x, n, stat_ws)
    check_call(_lib_().fq_fake_quant_offline(_ptr(x), _ptr(y), n, inner, _ptr(threshold), int(width), int(flags),
                                             _ptr(cur), _ptr(codes), _ptr(ws), _stream(x)))
    return y, cur, codes


def ste_forward(x, scales, clip_max=None, clip_min=None, eps=1e-10, out=None):
    """Generic `LinearQuantizeSTE.forward` (ste_func.py:37-41): `scales` is a device tensor with one scale per
    leading row of x (1 element = scalar scale)."""
    _check(x, "x")
    _check(scales, "scales")
    rows = scales.numel()
    if rows <= 0 or x.numel() == 0 or x.numel() % rows:
        raise ValueError("cannot broadcast %d scales over %s" % (rows, tuple(x.shape)))
    y = torch.empty_like(x) if out is None else _check(out, "out")
    has_clip = clip_max is not None
    lo = 0.0 if clip_min is None else float(clip_min)           # ste_func.py:34
    check_call(_lib_().fq_ste_forward(_ptr(x), _ptr(y), rows, x.numel() // rows, _ptr(scales), int(has_clip), lo,
                                      float(clip_max) if has_clip else 0.0, float(eps), _stream(x)))
    return y


def weight_fake_quant(w, rows, width=8, out=None, want_scales=False):
    """convert_conv2d.py:70-95 / convert_dense.py:52-63 with w viewed as (rows, numel/rows)."""
    _check(w, "w")
    if w.numel() == 0 or rows <= 0 or w.numel() % rows:
        raise ValueError("weight of %d elements cannot be viewed as %d rows" % (w.numel(), rows))
    wq = torch.empty_like(w) if out is None else _check(out, "out")
    scales = torch.empty(rows, dtype=torch.float32, device=w.device) if want_scales else None
    ws = _workspace(w.device, _lib_().fq_weight_workspace_bytes(rows))
    check_call(_lib_().fq_weight_fake_quant(_ptr(w), _ptr(wq), rows, w.numel() // rows, int(width), _ptr(scales),
                                            _ptr(ws), _stream(w)))
    return (wq, scales) if want_scales else wq


_G = {
    "F23": [[1, 0, 0], [1 / 2, 1 / 2, 1 / 2], [1 / 2, -1 / 2, 1 / 2], [0, 0, 1]],
    "F43": [[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6],
            [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]],
    "F63": [[1, 0, 0], [-2 / 9, -2 / 9, -2 / 9], [-2 / 9, 2 / 9, -2 / 9], [1 / 90, 1 / 45, 2 / 45],
            [1 / 90, -1 / 45, 2 / 45], [32 / 45, 16 / 45, 8 / 45], [32 / 45, -16 / 45, 8 / 45], [0, 0, 1]],
}
_WINO_CACHE = {}


def winograd_matrices(variant):
    """(G, pinv(G), pinv(G^T)) as fp32 numpy arrays.  G as in wino_matrix.py:29-54; the pseudo-inverses are computed
    on the host with numpy exactly like convert_conv2d.py:81-82, once per variant instead of on every forward."""
    if variant not in _WINO_CACHE:
        G = np.asarray(_G[variant], dtype=np.float32)
        _WINO_CACHE[variant] = (G, np.ascontiguousarray(np.linalg.pinv(G), dtype=np.float32),
                                np.ascontiguousarray(np.linalg.pinv(G.T), dtype=np.float32))
    return _WINO_CACHE[variant]


def wino_weight_fake_quant(w, variant, width=8, out=None, want_scales=False, GI=None, GTI=None):
    """convert_conv2d.py:71-83: per-out-channel fake-quant of 3x3 filters in the Winograd domain."""
    _check(w, "w")
    if w.dim() != 4 or tuple(w.shape[2:]) != (3, 3):
        raise ValueError("Winograd-domain quantisation needs (Cout, Cin/g, 3, 3) weights, got %s" % (tuple(w.shape),))
    G, GI0, GTI0 = winograd_matrices(variant)
    GI = GI0 if GI is None else np.ascontiguousarray(GI, dtype=np.float32)
    GTI = GTI0 if GTI is None else np.ascontiguousarray(GTI, dtype=np.float32)
    t = G.shape[0]
    wq = torch.empty_like(w) if out is None else _check(out, "out")
    cout, cin_g = w.shape[0], w.shape[1]
    scales = torch.empty(cout, dtype=torch.float32, device=w.device) if want_scales else None
    fp = ctypes.POINTER(ctypes.c_float)
    check_call(_lib_().fq_wino_weight_fake_quant(_ptr(w), _ptr(wq), cout, cin_g, t, G.ctypes.data_as(fp),
                                                 GI.ctypes.data_as(fp), GTI.ctypes.data_as(fp), int(width),
                                                 _ptr(scales), ctypes.c_void_p(0), _stream(w)))
    return (wq, scales) if want_scales else wq


# ---- RCCL collectives of the C ABI (for integrators without torch.distributed; dist.py itself uses torch.distributed) ------
COMM_SUM, COMM_MAX = 0, 1


def comm_unique_id():
    """128 opaque bytes made by rank 0 (ncclGetUniqueId); ship them to every other rank, then `comm_init` everywhere."""
    buf = ctypes.create_string_buffer(128)
    check_call(_lib_().fq_comm_unique_id(buf))
    return buf.raw


def comm_init(rank, world, unique_id):
    if len(unique_id) != 128:
        raise ValueError("the unique id is 128 bytes (got %d)" % len(unique_id))
    check_call(_lib_().fq_comm_init(int(rank), int(world), ctypes.create_string_buffer(bytes(unique_id), 128)))


def comm_world():
    return int(_lib_().fq_comm_world())


def comm_allreduce(t, op=COMM_SUM):
    """In-place all-reduce of a contiguous fp32 / fp64 / int64 device tensor over the library's RCCL communicator."""
    fn = {torch.float32: "fq_allreduce_f32", torch.float64: "fq_allreduce_f64", torch.int64: "fq_allreduce_i64"}.get(t.dtype)
    if fn is None:
        raise TypeError("comm_allreduce takes fp32, fp64 or int64 tensors (got %s)" % t.dtype)
    _check(t, "t", t.dtype)
    check_call(getattr(_lib_(), fn)(_ptr(t), t.numel(), int(op), _stream(t)))
    return t


def comm_destroy():
    check_call(_lib_().fq_comm_destroy())


_STATE_EPOCH = [0]


def state_epoch():
    """Counts the library's own in-place updates of calibration state (they go through raw pointers, so the tensor library's
    version counters do not see them): host-side memos about threshold VALUES key on it (quantize/convert/convert_conv2d.py)."""
    return _STATE_EPOCH[0]


def ema_update(state, current, momentum=0.9):
    """`_update_ema` (convert.py:66-79) over a contiguous vector of per-block scalars, in place on `state`."""
    _STATE_EPOCH[0] += 1
    _check(state, "state")
    _check(current, "current")
    if state.numel() != current.numel():
        raise ValueError("state/current size mismatch")
    check_call(_lib_().fq_ema_update(_ptr(state), _ptr(current), state.numel(), float(momentum), _stream(state)))
    return state


def global_max(x):
    """`np.max(feature_maps)` (distribution_calibrate.py:33-34) -> (1,) device tensor."""
    _check(x, "x")
    out = torch.empty(1, dtype=torch.float32, device=x.device)
    check_call(_lib_().fq_global_max(_ptr(x), x.numel(), _ptr(out), _stream(x)))
    return out


def histogram_accumulate(x, max_dev, hist, neg_count=None):
    """`_discrete_histogram` + `hist_collector[m] += hist` (distribution_calibrate.py:39-45,103-104).
    `hist`: int64 device tensor of `bins` exact counts, accumulated in place."""
    _check(x, "x")
    _check(max_dev, "max_dev")
    _check(hist, "hist", torch.int64)
    if neg_count is not None:
        _check(neg_count, "neg_count", torch.int32)
    check_call(_lib_().fq_histogram_accumulate(_ptr(x), x.numel(), _ptr(max_dev), hist.numel(), _ptr(hist),
                                               _ptr(neg_count), _stream(x)))
    return hist


def hist_to_float(hist):
    _check(hist, "hist", torch.int64)
    out = torch.empty(hist.shape, dtype=torch.float32, device=hist.device)
    check_call(_lib_().fq_hist_to_float(_ptr(hist), _ptr(out), hist.numel(), _stream(hist)))
    return out


def kl_search(hist, levels, min_bins):
    """`kl_calibrate` (distribution_calibrate.py:117-171) for L histograms: hist (L, bins) fp32 -> (L,) int32."""
    _check(hist, "hist")
    if hist.dim() == 1:
        hist = hist.reshape(1, -1)
    L, bins = hist.shape
    out = torch.empty(L, dtype=torch.int32, device=hist.device)
    ws = _workspace(hist.device, _lib_().fq_kl_workspace_bytes(L, bins))
    check_call(_lib_().fq_kl_search(_ptr(hist), L, bins, int(levels), int(min_bins), _ptr(out), _ptr(ws),
                                    _stream(hist)))
    return out


_CODE_MODES = {"int8": _lib.FQ_CODES_INT8, "uint8": _lib.FQ_CODES_UINT8, "range": _lib.FQ_CODES_RANGE,
               "scale": _lib.FQ_CODES_SCALE}


def quantize_codes(x, out_type="int8", range_dev=None):
    """`quantize`/`_quantize` (nn/quantized_conv.py:54-72).  Returns (codes int32, range_dev = [min, max, scale])."""
    _check(x, "x")
    if out_type not in _CODE_MODES:
        raise ValueError("unknown out type: %s" % (out_type,))
    mode = _CODE_MODES[out_type]
    if range_dev is None:
        if mode >= _lib.FQ_CODES_RANGE:
            raise ValueError("range_dev is required for out_type=%s" % out_type)
        range_dev = torch.empty(3, dtype=torch.float32, device=x.device)
    _check(range_dev, "range_dev")
    codes = torch.empty(x.shape, dtype=torch.int32, device=x.device)
    ws = _workspace(x.device, _lib_().fq_act_workspace_bytes(1))
    check_call(_lib_().fq_quantize_codes(_ptr(x), _ptr(codes), x.numel(), mode, _ptr(range_dev), _ptr(ws),
                                         _stream(x)))
    return codes, range_dev


def dequantize(codes, scale_dev):
    """`dequantize` (nn/quantized_conv.py:74-76)."""
    _check(codes, "codes", torch.int32)
    _check(scale_dev, "scale_dev")
    y = torch.empty(codes.shape, dtype=torch.float32, device=codes.device)
    check_call(_lib_().fq_dequantize(_ptr(codes), _ptr(y), codes.numel(), _ptr(scale_dev), _stream(codes)))
    return y


# ---- nn.Conv2D(quantized=True): one call per forward (include/fakequant.h at fq_qconv2d_forward) -------------------------------
QCONV_KINDS = {0: "direct", 1: "pointwise", 2: "dense3x3", 3: "depthwise3x3"}


def _geom(shape_w, strides, padding, groups):
    cout, cin_g, kh, kw = (int(v) for v in shape_w)
    return (cin_g * int(groups), cout, kh, kw, int(strides[0]), int(strides[1]), int(padding[0]), int(padding[1]), int(groups))


def qconv_kind(shape_w, strides, padding, groups):
    """Which kernel family a geometry takes ("direct", "pointwise", "dense3x3", "depthwise3x3")."""
    return QCONV_KINDS[_lib_().fq_qconv_kind(*_geom(shape_w, strides, padding, groups))]


def qconv_weights(w, strides, padding, groups, weight_dtype="int8", weight_range=None):
    """`quantize(F, weight, weight_dtype)` / `_quantize(F, weight, *_weight_range)` (nn/quantized_conv.py:117-120) once per
    weight version: the weights' range record, and - for the geometries with a fused kernel - their int8 codes in the layout
    that kernel reads.  Returns an opaque device buffer for `qconv2d`."""
    _check(w, "weight")
    if w.dim() != 4:
        raise ValueError("weight must be (cout, cin / groups, kh, kw)")
    if weight_range is None and weight_dtype not in ("int8", "uint8"):
        raise ValueError("unknown out type: %s" % (weight_dtype,))
    g = _geom(w.shape, strides, padding, groups)
    nbytes = _lib_().fq_qconv_weights_bytes(*g)
    if nbytes == 0:
        raise ValueError("bad convolution geometry %r" % (g,))
    buf = torch.empty(int(nbytes), dtype=torch.uint8, device=w.device)
    ws = _workspace(w.device, 64)
    mode = _lib.FQ_CODES_RANGE if weight_range is not None else _CODE_MODES[weight_dtype]
    lo, hi = (float(weight_range[0]), float(weight_range[1])) if weight_range is not None else (0.0, 0.0)
    check_call(_lib_().fq_qconv_weights_prepare(_ptr(w), g[0], g[1], g[2], g[3], g[4], g[5], g[6], g[7], g[8], mode, lo, hi,
                                                _ptr(buf), _ptr(ws), _stream(w)))
    return buf


def qconv_workspace(cout, device):
    """Per-(block, stream) state of `qconv2d`: running {min, max}, the range record, bias codes, per-channel constants.
    Initialised once; every forward leaves it initialised."""
    ws = torch.empty(int(_lib_().fq_qconv_workspace_bytes(int(cout))), dtype=torch.uint8, device=device)
    check_call(_lib_().fq_qconv_workspace_init(_ptr(ws), _stream(ws)))
    return ws


def qconv2d(x, w, wbuf, bias, strides, padding, groups, ws, input_dtype="uint8", input_range=None, act="none", in_stat=None,
            bn_scale=None, bn_shift=None, want_stat=False, force_direct=False, out=None):
    """`Conv2D.hybrid_forward` with quantized=True (nn/quantized_conv.py:106-159) on the UNPADDED fp32 input: range of the
    padded tensor, quantise-on-load integer convolution, int32 bias, activation on the integers, dequantise - one call.
    Returns y, or (y, per-sample max|y|) with want_stat."""
    _check(x, "x")
    _check(w, "weight")
    if x.dim() != 4 or w.dim() != 4:
        raise ValueError("x must be (n, cin, h, w) and weight (cout, cin / groups, kh, kw)")
    if input_range is None and input_dtype not in ("int8", "uint8"):
        raise ValueError("unknown out type: %s" % (input_dtype,))
    if act not in ("none", "relu") and not (act == "relu6" and bn_scale is not None):
        raise ValueError("activation %r: the block applies its activation to the int32 sums (none or relu; relu6 only behind "
                         "a folded BatchNorm)" % (act,))
    g = _geom(w.shape, strides, padding, groups)
    n, cin, h, wd = (int(v) for v in x.shape)
    if cin != g[0]:
        raise ValueError("input has %d channels, the weight wants %d" % (cin, g[0]))
    ho = (h + 2 * g[6] - g[2]) // g[4] + 1
    wo = (wd + 2 * g[7] - g[3]) // g[5] + 1
    if ho <= 0 or wo <= 0:
        raise ValueError("kernel larger than the padded input")
    y = torch.empty((n, g[1], ho, wo), dtype=torch.float32, device=x.device) if out is None else _check(out, "out")
    stat, zflag = _stat_target(n, x.device, want_stat)
    mode = _lib.FQ_CODES_RANGE if input_range is not None else _CODE_MODES[input_dtype]
    lo, hi = (float(input_range[0]), float(input_range[1])) if input_range is not None else (0.0, 0.0)
    if bias is not None:
        _check(bias, "bias")
        if bias.numel() != g[1]:
            raise ValueError("bias must have %d elements" % g[1])
    if (bn_scale is None) != (bn_shift is None):
        raise ValueError("bn_scale and bn_shift go together")
    for name, t in (("bn_scale", bn_scale), ("bn_shift", bn_shift)):
        if t is not None:
            _check(t, name)
            if t.numel() != g[1]:
                raise ValueError("%s must have %d elements (got %d)" % (name, g[1], t.numel()))
    if in_stat is not None:
        _check(in_stat, "in_stat")
        if in_stat.numel() != n:
            raise ValueError("in_stat must have one entry per sample")
    check_call(_lib_().fq_qconv2d_forward(_ptr(x), _ptr(w), _ptr(wbuf), _ptr(bias), _ptr(y), n, cin, h, wd, g[1], g[2], g[3],
                                          g[4], g[5], g[6], g[7], g[8], mode, lo, hi, _ptr(in_stat), _ACTS[act] | zflag,
                                          _ptr(bn_scale), _ptr(bn_shift), _ptr(stat), _ptr(ws), 1 if force_direct else 0,
                                          _stream(x)))
    return (y, stat) if want_stat else y
