"""CPU ORACLE — test infrastructure, NOT product code.

A plain-numpy restatement of the reference's fake-quantisation / calibration algorithm, written from the reference's
Python (file:line cited per function).  Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import this package; the shipped path (`quantization.mxnet_amd`) never does and fails loudly when
its HIP library is missing.

Parity status (DESIGN.md "Oracle"):
  * histogram / collect / KL functions: PINNED — checked in tests/test_oracle_golden.py against outputs of the
    reference's real `quantize/distribution_calibrate.py` executed in the build container (tests/golden/g1..g3).
  * everything that needs MXNet in the reference: pinned for COMPOSITION only — the goldens g4..g9 were produced by
    running the reference's unchanged Python over a CPU stand-in for `mxnet` whose primitive ops encode MXNet's
    documented semantics (the reference has no golden vectors or asserting tests of its own for this boundary:
    SURVEY.md F6).  Third-party arithmetic holder: Apache MXNet (unpinned in the reference, ~1.5.x) and numpy
    (this container: 2.2.x; scalar promotion follows NEP 50).

Primitive semantics restated here: fp32 everywhere unless stated; `roundf` = half AWAY from zero;
tensor / scalar = IEEE fp32 divide by fp32(scalar); `clip(a, lo, hi) = min(max(a, lo), hi)`; cast-to-int32
truncates toward zero; batch mean = fp32(sequential fp64 sum) / fp32(N).
"""
import numpy as np

F32 = np.float32
EPS = F32(1e-10)          # ste_func.py:39,41  `scale + 1e-10`  (fp32 add: numpy-2 weak python float / MXNet _plus_scalar)

__all__ = ["roundf", "absmax_per_sample", "batch_mean", "act_scale", "ste_codes", "ste_forward",
           "conv_input_fake_quant", "dense_input_fake_quant", "act_output_fake_quant", "weight_fake_quant",
           "winograd_G", "wino_weight_fake_quant", "ema_update", "discrete_histogram", "kl_calibrate",
           "kl_threshold", "quantize_codes", "dequantize", "qconv2d_forward", "unfused_reference_chain", "bn_act", "dwconv3x3", "weight_codes", "pwconv_i8", "conv3x3_i8", "bn_act_maxpool", "stem_conv_s2"]


def roundf(x):
    """C roundf (MXNet `round`): nearest, ties away from zero.  x - trunc(x) is exact."""
    x = np.asarray(x, dtype=F32)
    t = np.trunc(x)
    frac = x - t
    return np.where(np.abs(frac) >= F32(0.5), t + np.sign(x).astype(F32), t).astype(F32)


# ---- activations --------------------------------------------------------------------------------------
def absmax_per_sample(x):
    """`F.max(F.abs(x), axis=(1,2,3))` (convert_conv2d.py:56) / `axis=1` (convert_dense.py:41): one value per sample."""
    x = np.asarray(x, dtype=F32)
    return np.abs(x).reshape(x.shape[0], -1).max(axis=1).astype(F32)


def batch_mean(per_sample):
    """`.mean()` of the N per-sample maxima (convert_conv2d.py:56).  Defined here as fp32(sum in fp64, n = 0..N-1)
    divided in fp32 by N — MXNet's CPU reducer is compensated, its GPU reducer a tree: both agree with this to the
    last bit for the N <= 1024 non-negative values seen here except at double-rounding ties."""
    acc = np.float64(0.0)
    for v in np.asarray(per_sample, dtype=F32):
        acc += np.float64(v)
    return F32(F32(acc) / F32(len(per_sample)))


def act_scale(max_, signed, width):
    """convert_conv2d.py:59-64: fp32 `max_ / (2^(w-1)-1)` (signed) or `max_ / (2^w-1)` (unsigned)."""
    levels = (2 ** (width - 1) - 1) if signed else (2 ** width - 1)
    return F32(F32(max_) / F32(levels))


def ste_codes(x, scale, clip_max=None, clip_min=None, eps=EPS):
    """Integer stage of LinearQuantizeSTE.forward (ste_func.py:39,41): round(clip(x) / (scale + 1e-10)), as fp32."""
    x = np.asarray(x, dtype=F32)
    scale = np.asarray(scale, dtype=F32)
    denom = (scale + F32(eps)).astype(F32)
    if clip_max is not None:
        lo = F32(0.0) if clip_min is None else F32(clip_min)          # ste_func.py:34
        x = np.minimum(np.maximum(x, lo), F32(clip_max))
    with np.errstate(divide="ignore", invalid="ignore"):
        return roundf((x / denom).astype(F32))


def ste_forward(x, scale, clip_max=None, clip_min=None, eps=EPS):
    """LinearQuantizeSTE.forward (ste_func.py:37-41): codes * scale (scale WITHOUT epsilon)."""
    scale = np.asarray(scale, dtype=F32)
    return (ste_codes(x, scale, clip_max, clip_min, eps) * scale).astype(F32)


def conv_input_fake_quant(x, signed=False, width=8, offline_threshold=None):
    """Activation branch of `_conv2d_forward` (convert_conv2d.py:53-66).
    Returns (x_q, current_input_max, scale, codes)."""
    cur = batch_mean(absmax_per_sample(x))
    max_ = F32(offline_threshold) if offline_threshold is not None else cur
    scale = act_scale(max_, signed, width)
    min_ = F32(-max_) if signed else F32(0.0)
    codes = ste_codes(x, scale, max_, min_)
    return (codes * scale).astype(F32), cur, scale, codes


def dense_input_fake_quant(x, signed=False, width=8, offline_threshold=None):
    """`_dense_forward` (convert_dense.py:39-49): as conv but STE gets no clip_min => clips to [0, max] even when
    signed (reference quirk, kept)."""
    a = np.asarray(x, dtype=F32)
    if a.ndim > 2:
        # `F.max(F.abs(x), axis=1).mean()` (:41) on an un-flattened (N, C, H, W) input - vgg's first Dense - reduces over C only;
        # the mean then averages N * H * W values
        cur = batch_mean(np.abs(a).max(axis=1).reshape(-1))
    else:
        cur = batch_mean(absmax_per_sample(x))
    max_ = F32(offline_threshold) if offline_threshold is not None else cur
    scale = act_scale(max_, signed, width)
    codes = ste_codes(x, scale, max_, None)
    return (codes * scale).astype(F32), cur, scale, codes


def act_output_fake_quant(act, width=8, offline_threshold=None):
    """`_act_forward` (convert_act.py:49-54): per-sample max WITHOUT abs, no epsilon, unsigned."""
    a = np.asarray(act, dtype=F32)
    cur = batch_mean(a.reshape(a.shape[0], -1).max(axis=1))
    max_ = F32(offline_threshold) if offline_threshold is not None else cur
    scale = F32(max_ / F32(2 ** width - 1))
    codes = ste_codes(a, scale, max_, 0.0, eps=F32(0.0))
    return (codes * scale).astype(F32), cur, scale, codes


# ---- weights ------------------------------------------------------------------------------------------
def weight_fake_quant(w, quant_type="layer", width=8, num_group=1):
    """Weight branch of `_conv2d_forward` (convert_conv2d.py:68-95) and `_dense_forward` (convert_dense.py:52-63).
    layer: one scale; channel: one per w.shape[0]; group: one per `num_group` rows of reshape((G,-1)), broadcast as
    (G,1,1,1) — only shape-valid when G in {1, Cout}, as in the reference.  No clipping; scale array so the epsilon
    add is an fp32 tensor add.  Returns (w_q, scales)."""
    w = np.asarray(w, dtype=F32)
    levels = F32(2 ** (width - 1) - 1)
    if quant_type == "layer":
        rows = 1
    elif quant_type == "channel":
        rows = w.shape[0]
    elif quant_type == "group":
        rows = num_group
        if rows not in (1, w.shape[0]):
            raise ValueError("group quantisation broadcasts (G,1,1,1) against (Cout,...): needs G in {1, Cout}")
    else:
        raise ValueError(quant_type)
    max_ = np.abs(w).reshape(rows, -1).max(axis=1).astype(F32)
    scale = (max_ / levels).astype(F32)
    sc = scale.reshape((rows,) + (1,) * (w.ndim - 1)) if rows > 1 else scale.reshape((1,) * w.ndim)
    return ste_forward(w, sc), scale


_G = {
    "F23": [[1, 0, 0], [1 / 2, 1 / 2, 1 / 2], [1 / 2, -1 / 2, 1 / 2], [0, 0, 1]],
    "F43": [[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6],
            [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]],
    "F63": [[1, 0, 0], [-2 / 9, -2 / 9, -2 / 9], [-2 / 9, 2 / 9, -2 / 9], [1 / 90, 1 / 45, 2 / 45],
            [1 / 90, -1 / 45, 2 / 45], [32 / 45, 16 / 45, 8 / 45], [32 / 45, -16 / 45, 8 / 45], [0, 0, 1]],
}


def winograd_G(variant):
    """Winograd weight-transform matrices F(2,3) 4x3, F(4,3) 6x3, F(6,3) 8x3 (wino_matrix.py:29-60), fp32."""
    return np.asarray(_G[variant], dtype=F32)


def _seq_dot_last_first(a, b):
    """MXNet `dot`: contract last axis of a with first axis of b; k-sequential fp32, multiply and add separately
    rounded (no FMA) — the order the HIP kernel uses."""
    K = a.shape[-1]
    a2 = a.reshape(-1, K).astype(F32)
    b2 = b.reshape(K, -1).astype(F32)
    acc = (a2[:, 0:1] * b2[0:1, :]).astype(F32)
    for k in range(1, K):
        acc = (acc + (a2[:, k:k + 1] * b2[k:k + 1, :]).astype(F32)).astype(F32)
    return acc.reshape(a.shape[:-1] + b.shape[1:])


def wino_weight_fake_quant(w, variant, width=8, GI=None, GTI=None):
    """Winograd-domain per-channel weight fake-quant (convert_conv2d.py:71-83): U = G g G^T per (co,ci);
    per-out-channel abs-max over U; STE; back with pinv(G), pinv(G^T) (host numpy SVD, fp32)."""
    w = np.asarray(w, dtype=F32)
    G = winograd_G(variant)
    if GI is None:
        GI = np.linalg.pinv(G)
    if GTI is None:
        GTI = np.linalg.pinv(G.T)
    t1 = _seq_dot_last_first(G, w.transpose(2, 3, 0, 1)).transpose(2, 3, 0, 1)       # (co, ci, t, 3)
    U = _seq_dot_last_first(np.ascontiguousarray(t1), np.ascontiguousarray(G.T))      # (co, ci, t, t)
    cout = w.shape[0]
    levels = F32(2 ** (width - 1) - 1)
    max_ = np.abs(U).reshape(cout, -1).max(axis=1).astype(F32)
    scale = (max_ / levels).astype(F32)
    Uq = ste_forward(U, scale.reshape(cout, 1, 1, 1))
    t2 = _seq_dot_last_first(GI.astype(F32), Uq.transpose(2, 3, 0, 1)).transpose(2, 3, 0, 1)
    wq = _seq_dot_last_first(np.ascontiguousarray(t2), GTI.astype(F32))
    return wq.astype(F32), scale, U


# ---- EMA ----------------------------------------------------------------------------------------------
def ema_update(state, current, momentum=0.9):
    """`_update_ema` (convert.py:70): input_max <- (1-m)*current + m*input_max, every product/sum rounded to fp32."""
    a = (F32(1 - momentum) * np.asarray(current, dtype=F32)).astype(F32)
    b = (np.asarray(state, dtype=F32) * F32(momentum)).astype(F32)
    return (a + b).astype(F32)


# ---- KL calibration -----------------------------------------------------------------------------------
def discrete_histogram(fm, bins, max_=None):
    """`_discrete_histogram` (distribution_calibrate.py:31-47).  Deviation, documented in DESIGN.md: an index equal
    to `bins` (possible once max_ >= 256, where fp32 `max_ + 1e-5 == max_`) is clamped into the last bin; the
    reference would return a (bins+1)-long histogram there."""
    fm = np.asarray(fm, dtype=F32)
    if max_ is None:
        max_ = np.max(fm)
    max_ = F32(max_)
    assert np.min(fm) >= 0.0, "Activation should >=0"
    assert max_ > 0, "Bad distribution: all zero-value"
    v = np.minimum(np.maximum(fm.reshape(-1), F32(0)), max_)
    v = v[v != 0]
    scales = F32(F32(bins) / F32(max_ + F32(1e-5)))
    idx = (v * scales).astype(F32).astype(np.int32)
    idx = np.minimum(idx, bins - 1)
    return np.bincount(idx, minlength=bins).astype(F32), max_


def _seq_sum(a, dtype):
    """Python's builtin `sum()` over a numpy array = strictly sequential accumulation in the array's dtype
    (numpy 2 promotion).  `np.cumsum` is the same left-to-right recurrence, vectorised."""
    a = np.asarray(a, dtype=dtype)
    if a.size == 0:
        return dtype(0)
    return np.cumsum(a, dtype=dtype)[-1]


def kl_calibrate(data, levels, min_bins, bins):
    """`kl_calibrate` (distribution_calibrate.py:117-171), same arithmetic and summation order:
    P: fp32, tail mass folded with a sequential fp32 sum, normalised by a sequential fp32 sum;
    Q: fp64 merge into `levels`, linear-interpolated expansion, masked where P == 0, normalised by a sequential fp64
    sum; KL = sequential fp64 sum of P*log(P/Q) over Q != 0; strict `<` keeps the first minimum."""
    assert min_bins >= levels
    data = np.asarray(data, dtype=F32)
    best, best_div = min_bins, np.inf
    for i in range(min_bins, bins):
        p = data[:i].copy()
        tail = _seq_sum(data[i:], F32)
        p[i - 1] = F32(p[i - 1] + tail)
        s = _seq_sum(p, F32)
        with np.errstate(divide="ignore", invalid="ignore"):
            p = (p / s).astype(F32)
        b = np.arange(i) * levels / i
        fl = b.astype(np.int32)
        q = np.zeros(levels, np.float64)
        np.add.at(q, fl, data[:i].astype(np.float64))            # sequential in j, like the reference loop
        ce = np.clip(np.ceil(b), 0, levels - 1).astype(np.int32)
        qe = (q[ce] - q[fl]) * (b - fl) + q[fl]
        qe = qe * (p != 0)
        qs = _seq_sum(qe, np.float64)
        with np.errstate(divide="ignore", invalid="ignore"):
            qe = qe / qs
            m = qe != 0
            terms = p[m] * np.log(p[m] / qe[m])
        div = _seq_sum(terms, np.float64)
        if div < best_div:
            best_div, best = div, i
    return best


def kl_threshold(best_bins, fm_max, bins):
    """Caller-side threshold (examples/simulate_quantization.py:310): (best + 0.5) * (fm_max / bins)."""
    return F32((best_bins + 0.5) * (F32(fm_max) / bins))


# ---- int-code path (nn/quantized_conv.py) ---------------------------------------------------------------
def quantize_codes(x, out_type="int8", fixed_range=None):
    """`quantize` + `_quantize` (nn/quantized_conv.py:54-72): global range, clip, scale = max/127 if symmetric else
    (max-min)/255 (no zero-point), round(x/scale) (no epsilon), int32.  Returns (codes int32, scale fp32)."""
    x = np.asarray(x, dtype=F32)
    if fixed_range is not None:
        mn, mx = F32(fixed_range[0]), F32(fixed_range[1])
    elif out_type == "int8":
        mx = np.abs(x).max().astype(F32)
        mn = F32(-mx)
    elif out_type == "uint8":
        mx, mn = x.max().astype(F32), x.min().astype(F32)
    else:
        raise ValueError("unknown out type: " + str(out_type))
    xc = np.minimum(np.maximum(x, mn), mx)
    scale = F32(mx / F32(127)) if mx == -mn else F32(F32(mx - mn) / F32(255))
    with np.errstate(divide="ignore", invalid="ignore"):
        codes = roundf((xc / scale).astype(F32)).astype(np.int32)
    return codes, scale


def dequantize(codes, scale):
    """`dequantize` (nn/quantized_conv.py:74-76)."""
    return (np.asarray(codes).astype(F32) * F32(scale)).astype(F32)


def qconv2d_forward(x, w, b, stride, padding, groups, input_dtype="uint8", weight_dtype="int8", quantized=True,
                    input_range=None, weight_range=None, act=None, in_stat=None, bn_scale=None, bn_shift=None):
    """`Conv2D.hybrid_forward` (nn/quantized_conv.py:106-159) with the im2col + dot done as an exact integer
    correlation (the reference's fp32 dot is exact while |acc| < 2^24, :140-144): pad, per-tensor codes of the PADDED
    input and of the weight (`quantize`, or `_quantize` with the fixed `_input_range` / `_weight_range`, :112-120), int32
    bias codes (:122-127), per-group correlation, `act` ("relu") on the integers (:154-155), dequantise (:157-158).
    Project additions (None in the reference's block): `in_stat` - per-sample maxima of a non-negative input (checked; the
    device library then skips its range pass, the result is unchanged); `bn_scale` / `bn_shift` - an inference BatchNorm folded behind the
    dequantisation, multiply and add separately rounded, the activation then applies to that value."""
    x = np.asarray(x, dtype=F32)
    ph, pw = padding
    x = np.pad(x, ((0, 0), (0, 0), (ph, ph), (pw, pw)))
    if quantized:
        if in_stat is not None:
            # the per-sample maxima of a non-negative x as a fused producer left them: they spare the library its range pass
            # and change no result - the range is the tensor's, as always
            assert input_range is None and (x >= 0).all() and len(in_stat) == x.shape[0]
            assert np.array_equal(np.asarray(in_stat, dtype=F32), x.reshape(x.shape[0], -1).max(axis=1))
        xi, in_s = quantize_codes(x, input_dtype, fixed_range=input_range)
        wi, w_s = quantize_codes(w, weight_dtype, fixed_range=weight_range)
        bi = None
        if b is not None:
            b_scale = F32(in_s * w_s)
            b_max = F32(F32(in_s * w_s) * F32(2 ** 31))
            bc = np.minimum(np.maximum(np.asarray(b, dtype=F32), -b_max), b_max)
            with np.errstate(divide="ignore", invalid="ignore"):
                bi = roundf((bc / b_scale).astype(F32)).astype(np.int64)
        xa, wa = xi.astype(np.int64), wi.astype(np.int64)
    else:
        xa, wa, bi = x.astype(np.float64), np.asarray(w, dtype=np.float64), None if b is None else np.asarray(b, np.float64)
    N, C, H, W = xa.shape
    Co, Cg, kh, kw = wa.shape
    sh, sw = stride
    # NB the reference computes out_h = (H - kh + 1) // sh but collects range(0, H-kh+1, sh) columns (:42-49);
    # they agree for stride 1, the only stride its tests use.
    oh, ow = len(range(0, H - kh + 1, sh)), len(range(0, W - kw + 1, sw))
    y = np.zeros((N, Co, oh, ow), dtype=xa.dtype)
    cpg_out = Co // groups
    for g in range(groups):
        xs = xa[:, g * Cg:(g + 1) * Cg]
        wg = wa[g * cpg_out:(g + 1) * cpg_out].reshape(cpg_out, -1)
        if oh * ow > 64:            # larger planes: tap by tap over whole planes (same integer sums, any order)
            acc = np.zeros((N, cpg_out, oh, ow), dtype=xa.dtype)
            wg4 = wa[g * cpg_out:(g + 1) * cpg_out]
            for ky in range(kh):
                for kx in range(kw):
                    win = xs[:, :, ky:ky + (oh - 1) * sh + 1:sh, kx:kx + (ow - 1) * sw + 1:sw]
                    acc += np.einsum("nchw,oc->nohw", win, wg4[:, :, ky, kx])
            y[:, g * cpg_out:(g + 1) * cpg_out] = acc
            continue
        for i in range(oh):
            for j in range(ow):
                win = xs[:, :, i * sh:i * sh + kh, j * sw:j * sw + kw].reshape(N, -1)
                y[:, g * cpg_out:(g + 1) * cpg_out, i, j] = win @ wg.T
    if bi is not None:
        y = y + bi.reshape(1, -1, 1, 1)
    if quantized:
        yi = y.astype(np.int32)                      # (:144) wraps like the cast
        if act == "relu" and bn_scale is None:
            yi = np.maximum(yi, 0)
        out = dequantize(yi, F32(in_s * w_s))
        if bn_scale is not None:
            out = (out * np.asarray(bn_scale, dtype=F32).reshape(1, -1, 1, 1)).astype(F32)
            out = (out + np.asarray(bn_shift, dtype=F32).reshape(1, -1, 1, 1)).astype(F32)
            if act == "relu":
                out = np.maximum(out, F32(0))
            elif act == "relu6":
                out = np.minimum(np.maximum(out, F32(0)), F32(6))
        return out
    out = y.astype(F32)
    return np.maximum(out, F32(0)) if act == "relu" else out


# ---- fused producer (project addition; quantize/fuse.py) ---------------------------------------------------------
def bn_act(x, scale, shift, act="relu"):
    """Inference BatchNorm folded to per-channel scale/shift, multiply and add separately rounded, then the activation:
    the arithmetic of `fq_bn_act_stat` (x is (N, C, ...))."""
    x = np.asarray(x, dtype=F32)
    bshape = (1, -1) + (1,) * (x.ndim - 2)
    y = (x * np.asarray(scale, dtype=F32).reshape(bshape)).astype(F32)
    y = (y + np.asarray(shift, dtype=F32).reshape(bshape)).astype(F32)
    if act == "relu":
        y = np.maximum(y, F32(0))
    elif act == "relu6":
        y = np.minimum(np.maximum(y, F32(0)), F32(6))
    return y.astype(F32)


def bn_act_maxpool(x, scale, shift, act="relu"):
    """`bn_act` followed by MaxPool2D(3, stride 2, padding 1) (padding never wins): the arithmetic of
    `fq_bn_act_maxpool_stat`."""
    y = bn_act(x, scale, shift, act)
    N, C, H, W = y.shape
    ho, wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    pad = np.full((N, C, 2 * ho + 1, 2 * wo + 1), -np.inf, F32)
    pad[:, :, 1:H + 1, 1:W + 1] = y
    out = np.full((N, C, ho, wo), -np.inf, F32)
    for ky in range(3):
        for kx in range(3):
            out = np.maximum(out, pad[:, :, ky:ky + 2 * ho:2, kx:kx + 2 * wo:2])
    return out.astype(F32)


def global_avg_pool(x):
    """Arithmetic of `fq_global_avg_pool_stat` (gluon GlobalAvgPool2D): fp32(fp64 sum over the plane, in order) / fp32(hw)."""
    x = np.asarray(x, dtype=F32)
    n, c = x.shape[:2]
    flat = x.reshape(n, c, -1).astype(np.float64)
    acc = np.zeros((n, c), np.float64)
    for i in range(flat.shape[2]):
        acc = acc + flat[:, :, i]
    return (acc.astype(F32) / F32(flat.shape[2])).astype(F32).reshape(n, c, 1, 1)


def gemm_i8_codes(xcodes, wcodes, n, l, zoff=0):
    """Arithmetic of `fq_gemm_i8_codes` (the integer accumulation of nn/quantized_conv.py:134-151):
    out[n][co][p] = sum_k xcodes[n*l + p][k] * wcodes[co][k] + zoff * sum_k wcodes[co][k], exact in int64 -> int32."""
    x = np.asarray(xcodes).astype(np.int64)
    w = np.asarray(wcodes).astype(np.int64)
    acc = x @ w.T + int(zoff) * w.sum(axis=1)[None, :]
    return acc.reshape(n, l, w.shape[0]).transpose(0, 2, 1).astype(np.int32)


def eval_counters(logits, labels, counters=None):
    """The evaluation loop's bookkeeping (reference examples/simulate_quantization.py:122-148): pred = argmax(axis=1)
    (first index on ties), n_correct, total, correct_counter[gt], label_counter[gt] -> [n_correct, total, correct[c], label[c]]."""
    logits = np.asarray(logits, dtype=F32)
    labels = np.asarray(labels).astype(np.int64)
    n, classes = logits.shape
    out = np.zeros(2 + 2 * classes, dtype=F32) if counters is None else np.asarray(counters, dtype=F32).copy()
    pred = np.argmax(logits, axis=1)
    for p_, gt in zip(pred, labels):
        out[1] += 1
        if 0 <= gt < classes:
            out[2 + classes + gt] += 1
            if p_ == gt:
                out[0] += 1
                out[2 + gt] += 1
    return out


def stem_conv_s2(x, w, bias=None, bn_scale=None, bn_shift=None, act=None):
    """Arithmetic of `fq_stem_conv3x3s2` / `fq_stem_conv7x7s2` (the un-quantised first convolution; reference: mxnet
    F.Convolution called by gluon/nn/conv_layers.py, then the separate BatchNorm / Activation blocks): dense K x K (K from
    w), stride 2, pad K // 2, acc = fmaf(w[co][ci][ky][kx], x, acc) over ci, ky, kx in that order (fmaf emulated as in
    `dwconv3x3`), + bias, folded BN (separately rounded mul, add), activation."""
    x = np.asarray(x, dtype=F32)
    w = np.asarray(w, dtype=F32)
    N, C, H, W = x.shape
    Co, ks = w.shape[0], w.shape[2]
    pad = ks // 2
    Ho, Wo = (H + 2 * pad - ks) // 2 + 1, (W + 2 * pad - ks) // 2 + 1
    xp = np.zeros((N, C, H + 2 * pad + 1, W + 2 * pad + 1), F32)
    xp[:, :, pad:pad + H, pad:pad + W] = x
    acc = np.zeros((N, Co, Ho, Wo), F32)
    for ci in range(C):
        for ky in range(ks):
            for kx in range(ks):
                tap = xp[:, ci, ky:ky + (Ho - 1) * 2 + 1:2, kx:kx + (Wo - 1) * 2 + 1:2]
                acc = (w[None, :, ci, ky, kx, None, None].astype(np.float64) * tap[:, None].astype(np.float64)
                       + acc.astype(np.float64)).astype(F32)
    if bias is not None:
        acc = (acc + np.asarray(bias, dtype=F32).reshape(1, Co, 1, 1)).astype(F32)
    if bn_scale is not None:
        return bn_act(acc, bn_scale, bn_shift, act or "none")
    if act == "relu":
        acc = np.maximum(acc, F32(0))
    elif act == "relu6":
        acc = np.minimum(np.maximum(acc, F32(0)), F32(6))
    return acc.astype(F32)


stem_conv3x3s2 = stem_conv_s2          # (the 3x3 -> 32 case had its own name first)


def dwconv3x3(x, w, bias=None, stride=1, in_max=None, signed=False, width=8, lo_neg_max=None, bn_scale=None,
              bn_shift=None, act=None):
    """Arithmetic of `fq_dwconv3x3`: optional fake-quant of x with threshold `in_max` (exactly `ste_forward`), then
    acc = fmaf(w[ky][kx], xq, acc) over ky, kx row-major with zero padding (fmaf emulated as fp32(fp64 product + fp64
    acc): the fp64 product of two fp32 is exact), + bias, folded BN (separately rounded mul, add), activation."""
    x = np.asarray(x, dtype=F32)
    if in_max is not None:
        lo_neg = signed if lo_neg_max is None else lo_neg_max
        scale = act_scale(in_max, signed, width)
        x = ste_forward(x, scale, in_max, F32(-F32(in_max)) if lo_neg else F32(0))
    N, C, H, W = x.shape
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    xp = np.zeros((N, C, H + 2, W + 2), F32)
    xp[:, :, 1:-1, 1:-1] = x
    w = np.asarray(w, dtype=F32).reshape(C, 3, 3)
    acc = np.zeros((N, C, Ho, Wo), F32)
    for ky in range(3):
        for kx in range(3):
            tap = xp[:, :, ky:ky + (Ho - 1) * stride + 1:stride, kx:kx + (Wo - 1) * stride + 1:stride]
            acc = (w[None, :, ky, kx, None, None].astype(np.float64) * tap.astype(np.float64)
                   + acc.astype(np.float64)).astype(F32)
    if bias is not None:
        acc = (acc + np.asarray(bias, dtype=F32).reshape(1, C, 1, 1)).astype(F32)
    if bn_scale is not None:
        return bn_act(acc, bn_scale, bn_shift, act or "none")
    if act == "relu":
        acc = np.maximum(acc, F32(0))
    elif act == "relu6":
        acc = np.minimum(np.maximum(acc, F32(0)), F32(6))
    return acc.astype(F32)


def weight_codes(w, rows_per_scale, width=8):
    """Integer codes of the weight fake-quant: code = roundf(w / (s + 1e-10)), s = max|w| over the scale group /
    (2^(w-1)-1) (convert_conv2d.py:70-95).  Returns (codes int32 (rows, row_len), scales (rows,))."""
    w = np.asarray(w, dtype=F32)
    rows = w.shape[0]
    w2 = w.reshape(rows, -1)
    groups = rows // rows_per_scale
    gmax = np.abs(w2).reshape(groups, -1).max(axis=1).astype(F32)
    s = np.repeat((gmax / F32(2 ** (width - 1) - 1)).astype(F32), rows_per_scale)
    with np.errstate(divide="ignore", invalid="ignore"):
        codes = roundf((w2 / (s + EPS).astype(F32)[:, None]).astype(F32)).astype(np.int32)
    return codes, s


def pwconv_i8(x, w, rows_per_scale, wt_width, in_max, signed=False, width=8, lo_neg_max=None, bias=None,
              bn_scale=None, bn_shift=None, act=None, stride=1, residual=None):
    """Arithmetic of `fq_pwconv_i8` (`_strided`): integer codes of x and w, EXACT integer dot products, one fp32 multiply
    by sx*sw[co], bias, folded BN, [+ residual,] activation.  A stride only subsamples the input (in_max is the statistic of
    the WHOLE input, as the reference's fake-quant in front of the convolution sees it)."""
    x = np.asarray(x, dtype=F32)
    if stride != 1:
        x = np.ascontiguousarray(x[:, :, ::stride, ::stride])
    lo_neg = signed if lo_neg_max is None else lo_neg_max
    sx = act_scale(in_max, signed, width)
    cx = ste_codes(x, sx, in_max, F32(-F32(in_max)) if lo_neg else F32(0)).astype(np.int64)
    cw, sw = weight_codes(w, rows_per_scale, wt_width)
    N, C = x.shape[0], x.shape[1]
    isum = np.einsum("oc,ncp->nop", cw.astype(np.int64), cx.reshape(N, C, -1))
    sxw = (F32(sx) * sw).astype(F32)
    y = (isum.astype(F32) * sxw[None, :, None]).astype(F32)
    if bias is not None:
        y = (y + np.asarray(bias, dtype=F32)[None, :, None]).astype(F32)
    y = y.reshape((N, cw.shape[0]) + x.shape[2:])
    if residual is not None:                 # the tail of a residual unit: BN, then the shortcut, then the activation
        if bn_scale is not None:
            y = bn_act(y, bn_scale, bn_shift, "none")
        y = (y + np.asarray(residual, dtype=F32)).astype(F32)
        bn_scale = None
    if bn_scale is not None:
        return bn_act(y, bn_scale, bn_shift, act or "none")
    if act == "relu":
        y = np.maximum(y, F32(0))
    elif act == "relu6":
        y = np.minimum(np.maximum(y, F32(0)), F32(6))
    return y.astype(F32)


def conv3x3_i8(x, w, rows_per_scale, wt_width, in_max, signed=False, width=8, lo_neg_max=None, bias=None,
               bn_scale=None, bn_shift=None, act=None):
    """Arithmetic of `fq_conv3x3_i8` (dense 3x3, stride 1, zero padding 1): integer codes of x and w, EXACT integer sums
    over (ci, ky, kx) - the integer form nn/quantized_conv.py:134-151 spells out -, one fp32 multiply by sx*sw[co], bias,
    folded BN, activation.  w: (cout, cin, 3, 3)."""
    x = np.asarray(x, dtype=F32)
    lo_neg = signed if lo_neg_max is None else lo_neg_max
    sx = act_scale(in_max, signed, width)
    cx = ste_codes(x, sx, in_max, F32(-F32(in_max)) if lo_neg else F32(0)).astype(np.int64)
    cw, sw = weight_codes(w, rows_per_scale, wt_width)
    N, C, H, W = x.shape
    cw = cw.reshape(-1, C, 3, 3).astype(np.int64)
    pad = np.zeros((N, C, H + 2, W + 2), np.int64)
    pad[:, :, 1:-1, 1:-1] = cx
    isum = np.zeros((N, cw.shape[0], H, W), np.int64)
    for ky in range(3):
        for kx in range(3):
            isum += np.einsum("oc,nchw->nohw", cw[:, :, ky, kx], pad[:, :, ky:ky + H, kx:kx + W])
    sxw = (F32(sx) * sw).astype(F32)
    y = (isum.astype(F32) * sxw[None, :, None, None]).astype(F32)
    if bias is not None:
        y = (y + np.asarray(bias, dtype=F32)[None, :, None, None]).astype(F32)
    if bn_scale is not None:
        return bn_act(y, bn_scale, bn_shift, act or "none")
    if act == "relu":
        y = np.maximum(y, F32(0))
    elif act == "relu6":
        y = np.minimum(np.maximum(y, F32(0)), F32(6))
    return y.astype(F32)


def to_c16(codes, zoff):
    """Integer codes (n, C, h, w) -> the C16 code tensor of include/fakequant.h: int8 [n][ceil(C/16)][h*w][16], byte =
    (code + 128 - zoff) ^ 0x80 (unsigned codes re-centred by 128, signed ones as they are); channels past C hold code 0."""
    codes = np.asarray(codes).astype(np.int64)
    n, c, h, w = codes.shape
    cb = (c + 15) // 16
    full = np.zeros((n, cb * 16, h * w), np.int64)
    full[:, :c] = codes.reshape(n, c, h * w)
    b = ((full + 128 - zoff) & 255) ^ 0x80
    return b.astype(np.uint8).view(np.int8).reshape(n, cb, 16, h * w).transpose(0, 1, 3, 2).copy()


def from_c16(t, c, h, w, zoff):
    """The inverse of `to_c16`: -> integer codes (n, c, h, w)."""
    t = np.asarray(t).view(np.uint8).astype(np.int64)
    n, cb = t.shape[:2]
    b = t ^ 0x80                                               # = (code + 128 - zoff) mod 256
    if zoff == 128:
        codes = b                                              # unsigned codes 0 .. 255
    else:
        v = (b - 128) & 255                                    # signed: the two's complement byte of the code
        codes = np.where(v >= 128, v - 256, v)
    return codes.reshape(n, cb, h * w, 16).transpose(0, 1, 3, 2).reshape(n, cb * 16, h, w)[:, :c]


def weight_slices(w):
    """Arithmetic of `fq_weight_slices` for a (rows, ...) fp32 filter: per row p = 2^e (smallest power of two with
    max|w| <= p * 2^20), m = rint(w / p) as int64 (|m| <= 2^20).  Returns (m, p); the three int8 digits of m in balanced base
    128 are `slice_digits(m)`."""
    w = np.asarray(w, dtype=F32)
    rows = w.shape[0]
    flat = w.reshape(rows, -1)
    mx = np.abs(flat).max(axis=1)
    p = np.ones(rows, F32)
    nz = mx > 0
    f, e = np.frexp(mx[nz])                                    # mx = f * 2^e, 0.5 <= f < 1
    e = np.where(f == 0.5, e - 1, e)                           # a power of two itself
    p[nz] = np.ldexp(F32(1), e - 20).astype(F32)
    m = np.rint((flat / p[:, None]).astype(F32)).astype(np.int64)
    return m.reshape(w.shape), p


def slice_digits(m):
    d3 = ((m + 64) & 127) - 64
    m1 = (m - d3) >> 7
    d2 = ((m1 + 64) & 127) - 64
    d1 = (m1 - d2) >> 7
    return d1, d2, d3


def conv3x3_i8_sliced(x, w, in_max, signed=False, width=8, lo_neg_max=None, bias=None, bn_scale=None, bn_shift=None,
                      act=None):
    """Arithmetic of `fq_conv3x3_i8_sliced`: integer codes of x, the filter as m * p per output channel (`weight_slices`),
    T = sum m * cx EXACTLY (= (S1 << 14) + (S2 << 7) + S3 of the three digit slices), y = fp32(fp64(T) * fp64(sx * p)), then
    bias / folded BN / activation.  w: (cout, cin, 3, 3) - the filter the convolution is to multiply (under Winograd-domain
    quantisation: `wino_weight_fake_quant`'s output, convert_conv2d.py:79-83)."""
    x = np.asarray(x, dtype=F32)
    lo_neg = signed if lo_neg_max is None else lo_neg_max
    sx = act_scale(in_max, signed, width)
    cx = ste_codes(x, sx, in_max, F32(-F32(in_max)) if lo_neg else F32(0)).astype(np.int64)
    m, p = weight_slices(w)
    d1, d2, d3 = slice_digits(m)
    assert np.array_equal((d1 << 14) + (d2 << 7) + d3, m) and max(np.abs(d).max() for d in (d1, d2, d3)) <= 65
    N, C, H, W = x.shape
    pad = np.zeros((N, C, H + 2, W + 2), np.int64)
    pad[:, :, 1:-1, 1:-1] = cx
    T = np.zeros((N, m.shape[0], H, W), np.int64)
    for ky in range(3):
        for kx in range(3):
            T += np.einsum("oc,nchw->nohw", m[:, :, ky, kx], pad[:, :, ky:ky + H, kx:kx + W])
    sxp = (F32(sx) * p).astype(F32)
    y = (T.astype(np.float64) * sxp.astype(np.float64)[None, :, None, None]).astype(F32)
    if bias is not None:
        y = (y + np.asarray(bias, dtype=F32)[None, :, None, None]).astype(F32)
    if bn_scale is not None:
        return bn_act(y, bn_scale, bn_shift, act or "none")
    if act == "relu":
        y = np.maximum(y, F32(0))
    elif act == "relu6":
        y = np.minimum(np.maximum(y, F32(0)), F32(6))
    return y.astype(F32)


# ---- the reference's UNFUSED op chain, pass by pass (used as the CPU baseline workload) -------------------
def unfused_reference_chain(x, signed=False, width=8):
    """What `_conv2d_forward` + `LinearQuantizeSTE.forward` execute on the reference's CPU path for one activation
    tensor, one full-tensor pass per NDArray op (convert_conv2d.py:56-66, ste_func.py:41):
    abs -> per-sample max -> mean -> clip -> divide -> round -> multiply."""
    a = np.abs(x)
    m = a.reshape(a.shape[0], -1).max(axis=1)
    cur = batch_mean(m)
    scale = act_scale(cur, signed, width)
    lo = F32(-cur) if signed else F32(0)
    c = np.clip(x, lo, cur)
    d = c / F32(scale + EPS)
    t = np.trunc(d)
    r = np.where(np.abs(d - t) >= F32(0.5), t + np.sign(d), t).astype(F32)
    return r * scale, cur
